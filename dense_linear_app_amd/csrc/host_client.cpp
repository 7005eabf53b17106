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

// handle_json (worker_distrib.cpp:47-69) for the n payloads of a wave in one call.  buf: the payloads back to
// back, payload t = buf[off[t] .. off[t+1]).  Per payload: op[t] = 1 TRSM, 2 SYRK, 3 GEMM, 4 POTRF; B[t]; and for the
// op's tile ids, in the order the grouped launch takes them -- TRSM {inA, inL}, SYRK {inC, inA}, GEMM {inC, inAi,
// inAj}, POTRF {in} -- where the id string sits: id_off[3 t + r] (into buf), id_len[3 t + r].
// op[t] = 0: anything this flat reader does not take (another op, a missing field, an escape inside a string, a
// number that is not a plain integer, nested values, trailing text): the caller hands THAT payload to its general
// JSON parser, whose verdict -- including the exception text -- is then the reference's.  Returns 0.
static inline const char *skip_ws(const char *p, const char *e) {
  while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
  return p;
}
int chol_parse_payloads(const char *buf, const long long *off, int n, int *op, int *B, long long *id_off, int *id_len) {
  if (!buf || !off || !op || !B || !id_off || !id_len || n < 0) return -1;
  for (int t = 0; t < n; ++t) {
    op[t] = 0;
    B[t] = 0;
    for (int r = 0; r < 3; ++r) id_off[3 * t + r] = 0, id_len[3 * t + r] = -1;
    const char *p = buf + off[t], *e = buf + off[t + 1];
    // fields: 0 op, 1 B, 2 in, 3 inL, 4 inA, 5 inC, 6 inAi, 7 inAj
    const char *vs[8] = {nullptr};
    int vl[8] = {0};
    bool have[8] = {false};
    long long bval = 0;
    bool ok = true;
    p = skip_ws(p, e);
    if (p >= e || *p != '{') continue;
    p = skip_ws(p + 1, e);
    if (p < e && *p == '}') ok = false;
    while (ok) {
      if (p >= e || *p != '"') { ok = false; break; }
      const char *k = ++p;
      while (p < e && *p != '"' && *p != '\\') ++p;
      if (p >= e || *p != '"') { ok = false; break; }
      const int kl = (int)(p - k);
      p = skip_ws(p + 1, e);
      if (p >= e || *p != ':') { ok = false; break; }
      p = skip_ws(p + 1, e);
      int f = -1;
      static const char *names[8] = {"op", "B", "in", "inL", "inA", "inC", "inAi", "inAj"};
      for (int q = 0; q < 8; ++q)
        if ((int)strlen(names[q]) == kl && !memcmp(names[q], k, kl)) f = q;
      if (p < e && *p == '"') {
        const char *v = ++p;
        while (p < e && *p != '"' && *p != '\\' && (unsigned char)*p >= 0x20) ++p;
        if (p >= e || *p != '"') { ok = false; break; }
        if (f == 1) { ok = false; break; }  // B as a string: the general parser's TypeError
        if (f >= 0) vs[f] = v, vl[f] = (int)(p - v), have[f] = true;
        ++p;
      } else if (p < e && (*p == '-' || (*p >= '0' && *p <= '9'))) {
        const char *v = p;
        bool neg = *p == '-';
        if (neg) ++p;
        if (p >= e || *p < '0' || *p > '9') { ok = false; break; }
        if (*p == '0' && p + 1 < e && p[1] >= '0' && p[1] <= '9') { ok = false; break; }  // leading zero: not JSON
        long long x = 0;
        int digits = 0;
        while (p < e && *p >= '0' && *p <= '9') x = x * 10 + (*p - '0'), ++p, ++digits;
        if (digits > 9 || (p < e && (*p == '.' || *p == 'e' || *p == 'E'))) { ok = false; break; }
        if (f == 1) bval = neg ? -x : x, have[1] = true;
        else if (f >= 0) { ok = false; break; }  // an id that is a number: str(d[...]) in the general parser
        (void)v;
      } else {
        ok = false;  // true / false / null / nested: the general parser decides
        break;
      }
      p = skip_ws(p, e);
      if (p < e && *p == ',') { p = skip_ws(p + 1, e); continue; }
      if (p < e && *p == '}') { p = skip_ws(p + 1, e); ok = p == e; break; }
      ok = false;
    }
    if (!ok || !have[0] || !have[1]) continue;
    int code = 0;
    int need[3] = {-1, -1, -1};
    if (vl[0] == 4 && !memcmp(vs[0], "TRSM", 4)) code = 1, need[0] = 4, need[1] = 3;
    else if (vl[0] == 4 && !memcmp(vs[0], "SYRK", 4)) code = 2, need[0] = 5, need[1] = 4;
    else if (vl[0] == 4 && !memcmp(vs[0], "GEMM", 4)) code = 3, need[0] = 5, need[1] = 6, need[2] = 7;
    else if (vl[0] == 5 && !memcmp(vs[0], "POTRF", 5)) code = 4, need[0] = 2;
    else continue;
    bool all = true;
    for (int r = 0; r < 3; ++r)
      if (need[r] >= 0 && !have[need[r]]) all = false;
    if (!all) continue;
    for (int r = 0; r < 3; ++r)
      if (need[r] >= 0) id_off[3 * t + r] = vs[need[r]] - buf, id_len[3 * t + r] = vl[need[r]];
    op[t] = code;
    B[t] = (int)bval;
  }
  return 0;
}

}  // extern "C"
