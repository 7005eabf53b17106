"""Reads the gfx950 code object embedded in libcholmi.so (clang offload bundle in .hip_fatbin)
and returns, per kernel, what the hardware will allocate: VGPRs (granulated, arch + acc) and LDS.
Test infrastructure: lets the CPU suite check the co-residency budget the design relies on."""
import struct


def _sections(elf: bytes):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2 and elf[5] == 1
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    raw = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        raw.append(dict(name_off=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    strtab = raw[shstrndx]
    for s in raw:
        end = elf.index(b"\0", strtab["off"] + s["name_off"])
        s["name"] = elf[strtab["off"] + s["name_off"]:end].decode()
    return raw


def device_elf(so_path: str, arch: str = "gfx950") -> bytes:
    host = open(so_path, "rb").read()
    fat = next(s for s in _sections(host) if s["name"] == ".hip_fatbin")
    blob = host[fat["off"]:fat["off"] + fat["size"]]
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    assert blob.startswith(magic)
    n, = struct.unpack_from("<Q", blob, len(magic))
    p = len(magic) + 8
    for _ in range(n):
        off, size, tlen = struct.unpack_from("<QQQ", blob, p)
        triple = blob[p + 24:p + 24 + tlen].decode()
        p += 24 + tlen
        if arch in triple:
            return blob[off:off + size]
    raise KeyError(arch)


def kernel_resources(so_path: str) -> dict:
    """{mangled kernel name: {"vgprs": allocated VGPRs per lane (arch+acc), "lds": bytes}}"""
    elf = device_elf(so_path)
    secs = _sections(elf)
    out = {}
    for symtab in (s for s in secs if s["type"] in (2, 11)):  # SYMTAB, DYNSYM
        strs = secs[symtab["link"]]
        for i in range(symtab["size"] // 24):
            name_off, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", elf, symtab["off"] + 24 * i)
            end = elf.index(b"\0", strs["off"] + name_off)
            name = elf[strs["off"] + name_off:end].decode()
            if not name.endswith(".kd") or shndx == 0 or shndx >= len(secs):
                continue
            sec = secs[shndx]
            kd = elf[sec["off"] + value - sec["addr"]:][:64]
            lds, = struct.unpack_from("<I", kd, 0)
            rsrc1, = struct.unpack_from("<I", kd, 48)
            out[name[:-3]] = {"vgprs": ((rsrc1 & 0x3F) + 1) * 8, "lds": lds}
    return out


def disassemble(so_path: str, arch: str = "gfx950") -> dict:
    """{mangled kernel name: [instruction text, ...]} from llvm-objdump of the embedded code object."""
    import os
    import re
    import subprocess
    import tempfile

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(device_elf(so_path, arch))
        path = f.name
    try:
        text = subprocess.run([objdump, "-d", f"--mcpu={arch}", "--no-show-raw-insn", path], check=True, capture_output=True, text=True).stdout
    finally:
        os.unlink(path)
    out, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is not None and ln.startswith("\t"):
            ins = ln.split("//")[0].strip()
            if ins:
                cur.append(ins)
    return out


def vgprs_of(ins: str) -> set:
    """VGPR numbers an instruction names (v7, v[4:5])."""
    import re

    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", ins):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", ins))
    return regs
