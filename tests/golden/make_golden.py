#!/usr/bin/env python3
"""Generates the committed golden vectors in tests/golden/.  Run in the build
container only (needs /root/reference for oracle/_ref and scipy for the BLAS
stand-in); the GPU box consumes the committed files and never runs this.

What is produced, and from what:
  golden_inputs.json  -- outputs of the REFERENCE'S OWN functions
      (client_distrib.cpp:41-93, 224-264, 280-321 compiled by oracle/build_ref.sh):
      known answers and SHA-256 of the raw bytes of the generated matrix and of
      every lower tile for (N,B) = (12,4), (10,4) [ragged: zero-padded edge],
      (1024,256); load_params / parse_int_str cases; block ids.
  spd_N12.npy         -- the full 12x12 reference input (C1/C2 default N=12,B=4).
  tileops_B64.npz     -- inputs + outputs of the four tile ops with the worker's
      flag sets (worker_distrib.cpp:238,323,416,511) computed by scipy's bundled
      OpenBLAS (the library family the reference's CPU path ends in; the
      reference itself ships no vectors -> numerics are "parity unpinned").
  dag_N12_B4.npz / dag_N1024_B256.npz -- the factor produced by replaying the
      C1/C2 wave order (20 tasks at Nb=4: 4/6/6/4) with those scipy calls:
      full L for N=12; diag(L), per-tile Frobenius norms and 64 probe entries
      for N=1024.
  reference_vm_rel_error.json -- DATA recorded by the reference itself: the distinct
      (N, NB) -> rel_error values of Cholesky_chameleon_VM/cho/benchmark_results_plots/
      bench.csv (column 12, written by benchmark.c:282-285 from v6_test.c:86's "%.2e"
      line; seed 42 from benchmark.c:131, bump = N from v6_test.c:46).  These are the only
      numerical outputs the reference holds; they pin the dplgsy generator restatement and
      the literal V6 validation sequence (oracle.v6_literal_validation) to three digits.
"""
import hashlib
import json
import os
import sys

import numpy as np
from scipy.linalg import blas, lapack

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import oracle as orc  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.asfortranarray(a).tobytes(order="F")).hexdigest()


def replay_dag_scipy(A: np.ndarray, B: int):
    """C2:506-565 wave order with the W2 flag sets, on scipy BLAS. Returns tiles dict + task counts."""
    N = A.shape[0]
    Nb = (N + B - 1) // B
    ref = orc.RefClient()
    T = {(i, j): ref.extract_block(A, B, i, j) for i in range(Nb) for j in range(i + 1)}
    cnt = dict(POTRF=0, TRSM=0, SYRK=0, GEMM=0)
    for k in range(Nb):
        c, info = lapack.dpotrf(T[k, k], lower=1, clean=0)
        assert info == 0, info
        T[k, k] = np.asfortranarray(c)
        cnt["POTRF"] += 1
        for i in range(k + 1, Nb):
            T[i, k] = np.asfortranarray(blas.dtrsm(1.0, T[k, k], T[i, k], side=1, lower=1, trans_a=1, diag=0))
            cnt["TRSM"] += 1
        for i in range(k + 1, Nb):
            for j in range(k + 1, i + 1):
                if i == j:
                    T[i, i] = np.asfortranarray(blas.dsyrk(-1.0, T[i, k], beta=1.0, c=T[i, i], trans=0, lower=1))
                    cnt["SYRK"] += 1
                else:
                    T[i, j] = np.asfortranarray(blas.dgemm(-1.0, T[i, k], T[j, k], beta=1.0, c=T[i, j], trans_b=1))
                    cnt["GEMM"] += 1
    return T, cnt


def assemble_lower(T, N, B):
    Nb = (N + B - 1) // B
    L = np.zeros((Nb * B, Nb * B), order="F")
    for (i, j), t in T.items():
        L[i * B:(i + 1) * B, j * B:(j + 1) * B] = t
    return np.tril(L)[:N, :N]


def reference_vm_rel_error():
    import csv

    root = os.environ.get("REFERENCE_ROOT", "/root/reference")
    rel = "Cholesky_chameleon_VM/cho/benchmark_results_plots/bench.csv"
    seen = {}
    with open(os.path.join(root, rel), newline="") as f:
        for row in csv.DictReader(f):
            if int(row["exit_code"]) != 0:
                continue
            seen.setdefault((int(row["N"]), int(row["NB"])), set()).add(row["rel_error"])
    vals = []
    for (N, NB), v in sorted(seen.items()):
        assert len(v) == 1, (N, NB, v)  # deterministic across runs, schedulers and mappings
        vals.append({"N": N, "NB": NB, "rel_error": v.pop()})
    out = {"source": rel + " column 12 (rel_error), rows with exit_code 0", "seed": 42, "bump": "N",
           "printed_as": "%.2e by v6_test.c:86, re-printed %.6e by benchmark.c:283", "values": vals}
    with open(os.path.join(HERE, "reference_vm_rel_error.json"), "w") as f:
        json.dump(out, f, indent=1)
    return len(vals)


def main():
    print("reference_vm_rel_error.json:", reference_vm_rel_error(), "cases")
    ref = orc.RefClient()
    out = {"source": "reference functions compiled by oracle/build_ref.sh", "cases": {}}
    for N, B in ((12, 4), (10, 4), (1024, 256)):
        A = ref.reference_input(N)
        raw = ref.reference_input(N, dominance=False)
        Nb = (N + B - 1) // B
        case = {
            "N": N, "B": B, "Nb": Nb,
            "A00": A[0, 0], "A10": A[1, 0], "Alast": A[N - 1, N - 1], "sum": float(A.sum()),
            "sha_A": sha(A), "sha_A_before_dominance": sha(raw),
            "blk_1_0_first4": [float(x) for x in ref.extract_block(A, B, 1, 0).ravel(order="F")[:4]],
            "tiles_sha": {ref.block_id_from_ij(i, j): sha(ref.extract_block(A, B, i, j))
                          for i in range(Nb) for j in range(i + 1)},
        }
        out["cases"][f"N{N}_B{B}"] = case
        if N == 12:
            np.save(os.path.join(HERE, "spd_N12.npy"), A)
    # upper-variant of the generator (C2:243-249)
    out["upper_N12_sha"] = sha(ref.reference_input(12, uplo="U", dominance=False))
    # parameter parsing (C2:46-93).  env handled by the caller: keep it clean here
    os.environ.pop("CHOLESKY_N", None)
    os.environ.pop("CHOLESKY_B", None)
    pcases = [[], ["--N=1024", "--B=256"], ["64", "16"], ["--N=abc"], ["--N=0", "--B=-3"], ["--B=8"],
              ["100"], ["--N=12x"], ["--N=1073741824"], ["--N=1073741825"], ["-x", "20", "5", "7"]]
    out["load_params"] = [{"argv": a, "NB": list(ref.load_params(a))} for a in pcases]
    os.environ["CHOLESKY_N"] = "48"
    os.environ["CHOLESKY_B"] = "junk"
    out["load_params_env"] = {"env": {"CHOLESKY_N": "48", "CHOLESKY_B": "junk"},
                              "cases": [{"argv": a, "NB": list(ref.load_params(a))} for a in ([], ["--B=6"], ["7"])]}
    os.environ.pop("CHOLESKY_N")
    os.environ.pop("CHOLESKY_B")
    out["parse_int_str"] = [{"s": s, "fallback": 5, "value": ref.parse_int_str(s, 5)}
                            for s in ["7", "007", " 9", "9 ", "+3", "-1", "0", "1e3", "", "4294967296", "1073741824"]]
    out["block_ids"] = {f"{i},{j}": ref.block_id_from_ij(i, j) for i, j in ((0, 0), (3, 1), (12, 10))}
    with open(os.path.join(HERE, "golden_inputs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)

    # ---- tile ops on scipy OpenBLAS (stand-in for the reference's OpenBLAS path)
    B = 64
    A = ref.reference_input(4 * B)
    Akk = ref.extract_block(A, B, 0, 0)
    A10 = ref.extract_block(A, B, 1, 0)
    A20 = ref.extract_block(A, B, 2, 0)
    A21 = ref.extract_block(A, B, 2, 1)
    A11 = ref.extract_block(A, B, 1, 1)
    Lkk, info = lapack.dpotrf(Akk, lower=1, clean=0)
    assert info == 0
    X10 = blas.dtrsm(1.0, Lkk, A10, side=1, lower=1, trans_a=1, diag=0)
    X20 = blas.dtrsm(1.0, Lkk, A20, side=1, lower=1, trans_a=1, diag=0)
    C11 = blas.dsyrk(-1.0, X10, beta=1.0, c=A11.copy(order="F"), trans=0, lower=1)
    C21 = blas.dgemm(-1.0, X20, X10, beta=1.0, c=A21.copy(order="F"), trans_b=1)
    np.savez_compressed(os.path.join(HERE, "tileops_B64.npz"), Akk=Akk, A10=A10, A20=A20, A21=A21, A11=A11,
                        potrf_out=Lkk, trsm10_out=X10, trsm20_out=X20, syrk11_out=C11, gemm21_out=C21)

    # ---- whole DAG
    A12 = ref.reference_input(12)
    T, cnt = replay_dag_scipy(A12, 4)
    L12 = assemble_lower(T, 12, 4)
    np.savez_compressed(os.path.join(HERE, "dag_N12_B4.npz"), L=L12, counts=json.dumps(cnt))
    A1k = ref.reference_input(1024)
    T, cnt = replay_dag_scipy(A1k, 256)
    assert cnt == dict(POTRF=4, TRSM=6, SYRK=6, GEMM=4), cnt
    L = assemble_lower(T, 1024, 256)
    res = np.linalg.norm(L @ L.T - A1k) / np.linalg.norm(A1k)
    rng = np.random.default_rng(7)
    pi = rng.integers(0, 1024, 64)
    pj = rng.integers(0, 1024, 64)
    pi, pj = np.maximum(pi, pj), np.minimum(pi, pj)
    fro = np.array([[np.linalg.norm(np.tril(L)[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256]) for j in range(4)]
                    for i in range(4)])
    np.savez_compressed(os.path.join(HERE, "dag_N1024_B256.npz"), diag=np.diag(L).copy(), tile_fro=fro,
                        probe_i=pi, probe_j=pj, probe_v=L[pi, pj], residual=res, counts=json.dumps(cnt))
    print("golden written; N=1024 residual", res, "counts", cnt)


if __name__ == "__main__":
    main()
