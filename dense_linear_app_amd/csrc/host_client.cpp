// Host-side (client) helpers of libcholmi.so: the reference client's input
// construction, restated on the C++ standard library exactly as the reference
// does it (client_distrib.cpp:224-264, 280-309).  Pure host code: callable
// without a GPU.  The independent C restatement in oracle/chol_oracle.c and the
// reference's own functions (oracle/build_ref.sh) are the checkers for this file.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/cholmi.h"

extern "C" {

// C2:224-252: fill one triangle column by column from mt19937_64(seed) through
// uniform_real_distribution(-0.5, 0.5), mirror it, then add `bump` to the diagonal.
// libstdc++ maps one 64-bit draw r to  u = (double)((long double)r / 2^64), clamps u == 1 to
// nextafter(1, 0), and returns u * (b - a) + a; r -> double rounds to nearest-even exactly like
// the long-double quotient does, so the same bits come out of plain double arithmetic (the
// golden SHA-256 of the reference's own output pin this: tests/test_host_api.py).
void chol_make_spd_like_chameleon(double *A, int N, int LDA, double bump, char uplo,
                                  unsigned long long seed) {
  std::mt19937_64 gen(seed);
  const size_t ld = (size_t)LDA;
  const bool lower = (uplo == 'L' || uplo == 'l');
  for (int j = 0; j < N; ++j) {
    const int i0 = lower ? j : 0, i1 = lower ? N : j + 1;
    double *col = A + j * ld;
    for (int i = i0; i < i1; ++i) {
      double u = (double)gen() * 0x1p-64;
      if (u >= 1.0) u = std::nextafter(1.0, 0.0);
      col[i] = u * 1.0 + -0.5;
    }
  }
  // mirror, in cache-sized blocks (the values do not depend on the order)
  constexpr int TB = 64;
  for (int jb = 0; jb < N; jb += TB)
    for (int ib = 0; ib <= jb; ib += TB) {
      const int je = jb + TB < N ? jb + TB : N, ie = ib + TB < N ? ib + TB : N;
      for (int j = jb; j < je; ++j)
        for (int i = ib; i < ie && i < j; ++i) {
          // (i, j) is strictly upper: i < j
          if (lower)
            A[i + j * ld] = A[j + i * ld];
          else
            A[j + i * ld] = A[i + j * ld];
        }
    }
  for (int i = 0; i < N; ++i) A[i + i * ld] += bump;
}

// C2:255-264: raise each diagonal entry to (sum of |off-diagonal| of its row) + eps
// when it is smaller; rows are summed in column order j = 0..N-1.  Swept column by column
// (unit stride) with one running sum per row: every row still adds its terms in the order
// j = 0..N-1, so the sums are bit-identical to the reference's row-by-row loop, and a
// row's sum never involves a diagonal entry, so the order of the fix-ups does not matter.
void chol_enforce_strict_diag_dominance(double *A, int N, int LDA, double eps) {
  const size_t ld = (size_t)LDA;
  std::vector<double> s((size_t)N, 0.0);
  for (int j = 0; j < N; ++j) {
    const double *col = A + j * ld;
    for (int i = 0; i < j; ++i) s[i] += std::abs(col[i]);
    for (int i = j + 1; i < N; ++i) s[i] += std::abs(col[i]);
  }
  for (int i = 0; i < N; ++i) {
    const double need = s[i] + eps - A[i + i * ld];
    if (need > 0.0) A[i + i * ld] += need;
  }
}

// C2:280-309: tile (bi,bj) of a column-major N x N matrix as a zero-padded B x B
// column-major block.
void chol_extract_block(const double *A, int N, int LDA, int B, int bi, int bj, double *block) {
  std::memset(block, 0, sizeof(double) * (size_t)B * (size_t)B);
  const int r0 = bi * B, c0 = bj * B;
  const int rows = (N - r0 < B) ? N - r0 : B;
  for (int jj = 0; jj < B && c0 + jj < N; ++jj)
    if (rows > 0)
      std::memcpy(block + (size_t)jj * B, A + r0 + (size_t)(c0 + jj) * LDA, sizeof(double) * (size_t)rows);
}

}  // extern "C"
