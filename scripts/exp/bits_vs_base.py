#!/usr/bin/env python3
"""Is the factor bit-identical between two builds of the library?  bits_vs_base.py dump <file>  (run once per build via
LIBCHOLMI_PATH), then bits_vs_base.py cmp <a> <b>.  Tile POTRF + inverse-based TRSM on single tiles, and whole matrices."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
import numpy as np

def dump(path):
    from dense_linear_app_amd import chameleon as ch
    ch.CHAMELEON_Init(1, 1)
    out = {}
    rng = np.random.default_rng(3)
    for dt, npd in ((ch.ChamRealDouble, np.float64), (ch.ChamRealFloat, np.float32)):
        for B in (128, 256, 384, 512, 1024):
            G = rng.standard_normal((B, B))
            A = np.asfortranarray((G @ G.T + B * np.eye(B)).astype(npd))
            d = ch.CHAMELEON_Desc_Create(A, dt, B, B, B * B, B, B, 0, 0, B, B, 1, 1)
            info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
            out[f"tile {npd.__name__} {B}"] = [int(info), hashlib.sha256(np.tril(A).tobytes()).hexdigest()]
    for N, B in ((2048, 512), (4096, 512), (3072, 384), (4096, 1024), (8192, 512)):
        d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
        L = np.tril(d.to_lapack())
        out[f"matrix {N} {B}"] = [int(info), hashlib.sha256(L.tobytes()).hexdigest()]
        ch.CHAMELEON_Desc_Destroy(d)
    json.dump(out, open(path, "w"), indent=1)

if sys.argv[1] == "dump":
    dump(sys.argv[2])
else:
    a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
    same = 0
    for k in a:
        eq = a[k] == b[k]
        same += eq
        print(f"{k:28s} {'identical' if eq else 'DIFFERENT'}  info {a[k][0]} / {b[k][0]}")
    print(f"{same} of {len(a)} identical")
