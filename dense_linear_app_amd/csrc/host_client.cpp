// Host-side (client) helpers of libcholmi.so: the reference client's input
// construction, restated on the C++ standard library exactly as the reference
// does it (client_distrib.cpp:224-264, 280-309).  Pure host code: callable
// without a GPU.  The independent C restatement in oracle/chol_oracle.c and the
// reference's own functions (oracle/build_ref.sh) are the checkers for this file.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>

#include "../../include/cholmi.h"

extern "C" {

// C2:224-252: fill one triangle column by column from mt19937_64(seed) through
// uniform_real_distribution(-0.5, 0.5), mirror it, then add `bump` to the diagonal.
void chol_make_spd_like_chameleon(double *A, int N, int LDA, double bump, char uplo,
                                  unsigned long long seed) {
  std::mt19937_64 gen(seed);
  std::uniform_real_distribution<double> dist(-0.5, 0.5);
  const size_t ld = (size_t)LDA;
  const bool lower = (uplo == 'L' || uplo == 'l');
  for (int j = 0; j < N; ++j) {
    const int i0 = lower ? j : 0, i1 = lower ? N : j + 1;
    for (int i = i0; i < i1; ++i) A[i + j * ld] = dist(gen);
  }
  for (int j = 0; j < N; ++j) {
    if (lower)
      for (int i = 0; i < j; ++i) A[i + j * ld] = A[j + i * ld];
    else
      for (int i = j + 1; i < N; ++i) A[i + j * ld] = A[j + i * ld];
  }
  for (int i = 0; i < N; ++i) A[i + i * ld] += bump;
}

// C2:255-264: raise each diagonal entry to (sum of |off-diagonal| of its row) + eps
// when it is smaller; rows are summed in column order j = 0..N-1.
void chol_enforce_strict_diag_dominance(double *A, int N, int LDA, double eps) {
  const size_t ld = (size_t)LDA;
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
    for (int j = 0; j < N; ++j)
      if (j != i) s += std::abs(A[i + j * ld]);
    const double need = s + eps - A[i + i * ld];
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
