// The whole-matrix auxiliary tile operations the reference driver's validation block
// calls around the factorisation (v6_test.c:51 dlacpy, :74/:85 dlange, :79-80 dlacpy +
// dlauum, :83 dgeadd), on a device-resident single-process descriptor.
//
// All four are HBM-bound passes over the stored tile image except dlauum (a TN product,
// N^3/3 flops on the fp64/fp32 16x16x4 MFMA, operands read straight from L2/HBM: it is
// used for validation only).  A stored tile may be larger than the
// caller's (mbu) and the last tile row/column ragged: only positions inside the
// m x n matrix take part.
#include "cholmi_internal.h"

namespace cholmi {

namespace {

typedef double vd4_t __attribute__((ext_vector_type(4)));
typedef float vf4_t __attribute__((ext_vector_type(4)));

template <typename T>
struct Mf;
template <>
struct Mf<double> {
  using acc_t = vd4_t;
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <>
struct Mf<float> {
  using acc_t = vf4_t;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

// stored index -> (global row, global col, inside the matrix?)
struct Pos {
  long gi, gj;
  bool inside;
};
__device__ __forceinline__ Pos locate(const TileGeo &g, long idx) {
  const long per = (long)g.mbs * g.mbs;
  const long tl = idx / per, e = idx - tl * per;
  const int il = (int)(tl % g.lmt), jl = (int)(tl / g.lmt);
  const int ii = (int)(e % g.mbs), jj = (int)(e / g.mbs);
  Pos p;
  p.gi = (long)il * g.mbu + ii;
  p.gj = (long)jl * g.mbu + jj;
  p.inside = ii < g.mbu && jj < g.mbu && p.gi < g.m && p.gj < g.n;
  return p;
}
__device__ __forceinline__ bool on_side(int side, long gi, long gj) {
  return side == 0 || (side == 1 && gi >= gj) || (side == 2 && gi <= gj);
}

template <typename T>
__global__ __launch_bounds__(256) void k_lacpy(TileGeo g, int side, const T *__restrict__ A, T *__restrict__ B) {
  const long total = (long)g.lmt * g.lnt * g.mbs * g.mbs;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const Pos p = locate(g, idx);
    if (p.inside && on_side(side, p.gi, p.gj)) B[idx] = A[idx];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_geadd(TileGeo g, T alpha, const T *__restrict__ A, T beta,
                                               T *__restrict__ B) {
  const long total = (long)g.lmt * g.lnt * g.mbs * g.mbs;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const Pos p = locate(g, idx);
    if (p.inside) B[idx] = alpha * A[idx] + beta * B[idx];
  }
}

__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v) {
  // non-negative doubles order like their bit patterns
  atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(v));
}

// One thread per (tile, row): the row's |a| sum over the tile's columns, added to rowsum[gi].
// Adjacent threads read adjacent rows of the same column: coalesced.
template <typename T>
__global__ __launch_bounds__(256) void k_rowsums(TileGeo g, const T *__restrict__ A, double *rowsum) {
  const long lines = (long)g.lmt * g.lnt * g.mbs;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < lines; q += (long)gridDim.x * 256) {
    const long tl = q / g.mbs;
    const int ii = (int)(q - tl * g.mbs);
    const int il = (int)(tl % g.lmt), jl = (int)(tl / g.lmt);
    const long gi = (long)il * g.mbu + ii;
    if (ii >= g.mbu || gi >= g.m) continue;
    const long cols = min((long)g.mbu, g.n - (long)jl * g.mbu);
    const T *p = A + tl * (long)g.mbs * g.mbs + ii;
    double s = 0.0;
    for (long jj = 0; jj < cols; ++jj) s += fabs((double)p[jj * g.mbs]);
    atomicAdd(&rowsum[gi], s);
  }
}

// One wave per (tile, column): |a| sum -> colsum[gj]; max|a| -> scal[0]; sum a^2 -> scal[1].
template <typename T>
__global__ __launch_bounds__(256) void k_colstats(TileGeo g, const T *__restrict__ A, double *colsum,
                                                   double *scal) {
  const int lane = threadIdx.x & 63;
  const long lines = (long)g.lmt * g.lnt * g.mbs;
  for (long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6); q < lines; q += (long)gridDim.x * 4) {
    const long tl = q / g.mbs;
    const int jj = (int)(q - tl * g.mbs);
    const int il = (int)(tl % g.lmt), jl = (int)(tl / g.lmt);
    const long gj = (long)jl * g.mbu + jj;
    if (jj >= g.mbu || gj >= g.n) continue;
    const long rows = min((long)g.mbu, g.m - (long)il * g.mbu);
    const T *p = A + tl * (long)g.mbs * g.mbs + (long)jj * g.mbs;
    double s = 0.0, mx = 0.0, ss = 0.0;
    for (long ii = lane; ii < rows; ii += 64) {
      const double v = fabs((double)p[ii]);
      s += v;
      mx = fmax(mx, v);
      ss += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_down(s, o, 64);
      mx = fmax(mx, __shfl_down(mx, o, 64));
      ss += __shfl_down(ss, o, 64);
    }
    if (lane == 0) {
      atomicAdd(&colsum[gj], s);
      atomic_max_nonneg(&scal[0], mx);
      atomicAdd(&scal[1], ss);
    }
  }
}

__global__ __launch_bounds__(256) void k_vecmax(const double *v, long n, double *out) {
  double mx = 0.0;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n; q += (long)gridDim.x * 256) mx = fmax(mx, v[q]);
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
  if ((threadIdx.x & 63) == 0) atomic_max_nonneg(out, mx);
}

// out(I,J) = sum_{K >= I} L(K,I)^T L(K,J) for the tiles I >= J, L lower triangular (the strict
// upper triangle of the diagonal tiles is not read).  One workgroup = a 64 x 64 block of one
// output tile, one wave = 32 x 32 of it = 2 x 2 MFMA tiles.  Operand lane l holds column
// (l & 15) and, per 16-row chunk, the four rows 4 (l >> 4) .. +3 (one 32/16-byte load); MFMA
// step s multiplies row 4 (l >> 4) + s of both operands, so the k order inside a chunk is
// permuted identically on both sides.  Swapped operands: accumulator register r of lane l is
// out(i0 + (l & 15), j0 + drow(l, r)), i.e. consecutive lanes write consecutive rows.
template <typename T>
__global__ __launch_bounds__(256) void k_lauum_lower(const T *__restrict__ L, T *__restrict__ out, int nt,
                                                     int mbs) {
  using acc_t = typename Mf<T>::acc_t;
  const int nb = mbs / 64, per = nb * nb;
  const int tix = blockIdx.x / per, blk = blockIdx.x % per;
  int I = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
  while ((long)I * (I + 1) / 2 > tix) --I;
  while ((long)(I + 1) * (I + 2) / 2 <= tix) ++I;
  const int J = tix - I * (I + 1) / 2;
  const int bi = blk % nb, bj = blk / nb;
  if (I == J && bi < bj) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i0 = bi * 64 + (w & 1) * 32, j0 = bj * 64 + (w >> 1) * 32;
  const int c = lane & 15, g4 = (lane >> 4) * 4;
  const long bsiz = (long)mbs * mbs;
  acc_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = T(0);
  for (int K = I; K < nt; ++K) {
    const T *Li = L + ((long)K + (long)I * nt) * bsiz;
    const T *Lj = L + ((long)K + (long)J * nt) * bsiz;
    const bool tri_i = (K == I), tri_j = (K == J);
    // rows above the wave's first column are zero in a triangular tile
    const int kbeg = tri_i ? (i0 & ~15) : 0;
    for (int k0 = kbeg; k0 < mbs; k0 += 16) {
      T xi[2][4], xj[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int ci = i0 + a * 16 + c, cj = j0 + a * 16 + c;
        const T *pi = Li + (long)ci * mbs + k0 + g4;
        const T *pj = Lj + (long)cj * mbs + k0 + g4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          xi[a][s] = pi[s];
          xj[a][s] = pj[s];
          if (tri_i && k0 + g4 + s < ci) xi[a][s] = T(0);
          if (tri_j && k0 + g4 + s < cj) xj[a][s] = T(0);
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = Mf<T>::mfma(xj[b][s], xi[a][s], acc[a][b]);
    }
  }
  T *O = out + ((long)I + (long)J * nt) * bsiz;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + a * 16 + c, j = j0 + b * 16 + Mf<T>::drow(lane, r);
        if (I > J || i >= j) O[(long)i + (long)j * mbs] = acc[a][b][r];
      }
}

// out = in^T for `count` independent mb x mb tiles (tile q of `in` at in + q*istride, of `out` at
// out + q*ostride, elements); mb a multiple of 64.  64 x 64 sub-blocks through LDS, both sides
// coalesced.
template <typename T>
__global__ __launch_bounds__(256) void k_tiles_transpose(const T *__restrict__ in, long istride, T *__restrict__ out,
                                                         long ostride, int mb) {
  __shared__ T sa[64][65];
  const int nsub = mb / 64;
  const int q = blockIdx.x / (nsub * nsub), sub = blockIdx.x % (nsub * nsub);
  const int si = sub % nsub, sj = sub / nsub;
  const T *ta = in + q * istride + si * 64 + (long)sj * 64 * mb;
  T *tb = out + q * ostride + sj * 64 + (long)si * 64 * mb;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int c = ty; c < 64; c += 4) sa[c][tx] = ta[tx + (long)c * mb];
  __syncthreads();
  for (int c = ty; c < 64; c += 4) tb[tx + (long)c * mb] = sa[tx][c];
}

inline int grid_for_elems(long total) { return (int)std::min<long>(65536, (total + 255) / 256); }

}  // namespace

template <typename T>
void launch_lacpy(hipStream_t s, const TileGeo &g, int side, const T *A, T *B) {
  const long total = (long)g.lmt * g.lnt * g.mbs * g.mbs;
  if (total) k_lacpy<T><<<grid_for_elems(total), 256, 0, s>>>(g, side, A, B);
}

template <typename T>
void launch_geadd(hipStream_t s, const TileGeo &g, double alpha, const T *A, double beta, T *B) {
  const long total = (long)g.lmt * g.lnt * g.mbs * g.mbs;
  if (total) k_geadd<T><<<grid_for_elems(total), 256, 0, s>>>(g, (T)alpha, A, (T)beta, B);
}

// work: max(m, n) + 2 doubles, zeroed by this call.  Result in work[0] afterwards:
// kind 0 = max |a|, 1 = one norm, 2 = infinity norm, 3 = Frobenius norm squared.
template <typename T>
void launch_lange(hipStream_t s, const TileGeo &g, int kind, const T *A, double *work) {
  const long vec = std::max(g.m, g.n);
  (void)hipMemsetAsync(work, 0, (size_t)(vec + 2) * sizeof(double), s);
  const long lines = (long)g.lmt * g.lnt * g.mbs;
  if (!lines) return;
  double *scal = work, *v = work + 2;
  if (kind == 2) {
    k_rowsums<T><<<grid_for_elems(lines), 256, 0, s>>>(g, A, v);
    k_vecmax<<<grid_for_elems(g.m), 256, 0, s>>>(v, g.m, scal);
  } else {
    k_colstats<T><<<(int)std::min<long>(65536, (lines + 3) / 4), 256, 0, s>>>(g, A, v, scal);
    if (kind == 1) {
      (void)hipMemsetAsync(scal, 0, sizeof(double), s);
      k_vecmax<<<grid_for_elems(g.n), 256, 0, s>>>(v, g.n, scal);
    } else if (kind == 3) {
      (void)hipMemcpyAsync(scal, scal + 1, sizeof(double), hipMemcpyDeviceToDevice, s);
    }
  }
}

template <typename T>
void launch_lauum_lower(hipStream_t s, const T *L, T *out, int nt, int mbs) {
  const long blocks = (long)nt * (nt + 1) / 2 * (mbs / 64) * (mbs / 64);
  if (blocks) k_lauum_lower<T><<<(unsigned)blocks, 256, 0, s>>>(L, out, nt, mbs);
}

template <typename T>
void launch_tiles_transpose(hipStream_t s, const T *in, long istride, T *out, long ostride, int mb, int count) {
  const int nsub = mb / 64;
  if (count > 0) k_tiles_transpose<T><<<(unsigned)(count * nsub * nsub), 256, 0, s>>>(in, istride, out, ostride, mb);
}

#define INSTANTIATE_V(T)                                                                     \
  template void launch_lacpy<T>(hipStream_t, const TileGeo &, int, const T *, T *);          \
  template void launch_geadd<T>(hipStream_t, const TileGeo &, double, const T *, double, T *); \
  template void launch_lange<T>(hipStream_t, const TileGeo &, int, const T *, double *);     \
  template void launch_lauum_lower<T>(hipStream_t, const T *, T *, int, int);                \
  template void launch_tiles_transpose<T>(hipStream_t, const T *, long, T *, long, int, int);
INSTANTIATE_V(double)
INSTANTIATE_V(float)

}  // namespace cholmi
