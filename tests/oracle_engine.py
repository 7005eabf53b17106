"""Test-only worker backend on the CPU oracle (the product has no CPU path)."""
import numpy as np

from oracle import oracle as orc


class OracleTileBackend:
    """Test-only worker backend: the four tile routines on the CPU oracle."""

    def potrf(self, A, B):
        out, info = orc.dpotrf(A.reshape((B, B), order="F"))
        A[:] = out.ravel(order="F")
        return info

    def trsm(self, L, A, B):
        A[:] = orc.dtrsm(L.reshape((B, B), order="F"), A.reshape((B, B), order="F")).ravel(order="F")
        return 0

    def syrk(self, A, Cm, B):
        Cm[:] = orc.dsyrk(A.reshape((B, B), order="F"), Cm.reshape((B, B), order="F")).ravel(order="F")
        return 0

    def gemm(self, Ai, Aj, Cm, B):
        Cm[:] = orc.dgemm(Ai.reshape((B, B), order="F"), Aj.reshape((B, B), order="F"),
                          Cm.reshape((B, B), order="F")).ravel(order="F")
        return 0
