// wave_linalg.h -- one-wavefront dense solvers on matrices staged in LDS (device only).
//
// wave_pinv_solve: x = pinv(A) b with small singular values zeroed, by one-sided (Hestenes)
// Jacobi -- what the reference gets from vnl_matrix_inverse + zero_out_absolute
// (DenseLinearEquationSystemParametersEstimator.hxx:38-45, SinglePointTarget...cxx:192-201).
// One wave64 per system: the n/2 disjoint column pairs of a round-robin round are rotated in
// parallel, two lanes per pair (each takes every other row); matrices are column-major in LDS
// with an odd leading dimension so the pair/row interleave is bank-conflict free.
// wave_pinv_solve: workgroup = one wave; block_pinv_solve<256>: four waves per system.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace lsqr {

__device__ inline double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ inline double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) {
    double t = __shfl_xor(v, o);
    v = t > v ? t : v;
  }
  return v;
}

// A: m x n (m >= n, n <= 64) column-major, leading dimension lda; destroyed (becomes U*S).
// V: n x n column-major (ldv).  b: m.  x: n (output).  cwork: n scratch doubles.
// Singular values <= max(tol_abs, tol_rel * sigma_max) are zeroed.  Returns the rank.
// T = threads of the calling workgroup (64 or 256): T/32 lanes share one column pair, each taking
// every (T/32)-th row; the lanes of a pair are consecutive, so their partial dot products combine
// with wave shuffles.
template <int T>
__device__ inline int block_pinv_solve(int m, int n, double *A, int lda, double *V, int ldv,
                                       const double *b, double tol_abs, double tol_rel, double *x,
                                       double *cwork) {
  constexpr int LPP = T / 32;  // lanes per pair
  __shared__ int s_rank;
  __shared__ double s_smax;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < n * n; idx += T) {
    int r = idx % n, c = idx / n;
    V[c * ldv + r] = (r == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int nn = (n + 1) & ~1;  // even number of players; index n is a dummy when n is odd
  const int p = tid / LPP, sub = tid % LPP;
  for (int sweep = 0; sweep < 60; sweep++) {
    int any_rot = 0;
    for (int r = 0; r < nn - 1; r++) {
      int i = 0, j = 0;
      bool active = p < nn / 2;
      if (active) {
        if (p == 0) {
          i = nn - 1;
          j = r;
        } else {
          i = (r + p) % (nn - 1);
          j = (r - p + (nn - 1)) % (nn - 1);
        }
        if (i > j) {
          int t = i;
          i = j;
          j = t;
        }
        active = j < n;
      }
      double al = 0, be = 0, ga = 0;
      if (active)
        for (int k = sub; k < m; k += LPP) {
          double ui = A[i * lda + k], uj = A[j * lda + k];
          al = fma(ui, ui, al);
          be = fma(uj, uj, be);
          ga = fma(ui, uj, ga);
        }
#pragma unroll
      for (int o = 1; o < LPP; o <<= 1) {
        al += __shfl_xor(al, o);
        be += __shfl_xor(be, o);
        ga += __shfl_xor(ga, o);
      }
      // columns count as orthogonal once their cosine is at the rounding floor of an m-term dot
      // product (a tighter bound only re-rotates noise until the sweep limit)
      bool rot = active && ga != 0.0 && fabs(ga) > 4e-15 * sqrt(al * be);
      if (rot) {
        double zeta = (be - al) / (2.0 * ga);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int k = sub; k < m; k += LPP) {
          double ui = A[i * lda + k], uj = A[j * lda + k];
          A[i * lda + k] = c * ui - sn * uj;
          A[j * lda + k] = sn * ui + c * uj;
        }
        for (int k = sub; k < n; k += LPP) {
          double vi = V[i * ldv + k], vj = V[j * ldv + k];
          V[i * ldv + k] = c * vi - sn * vj;
          V[j * ldv + k] = sn * vi + c * vj;
        }
      }
      any_rot |= __syncthreads_or(rot ? 1 : 0);  // also the barrier between rounds
    }
    if (!any_rot) break;
  }
  // singular values and projections of b
  double s2 = 0, d = 0;
  if (tid < n)
    for (int k = 0; k < m; k++) {
      double a = A[tid * lda + k];
      s2 = fma(a, a, s2);
      d = fma(a, b[k], d);
    }
  double sig = sqrt(s2);
  if (tid < 64) {  // n <= 64: the first wave holds every column
    double smax = wave_max(tid < n ? sig : 0.0);
    if (tid == 0) s_smax = smax;
  }
  __syncthreads();
  double tol = tol_rel * s_smax;
  if (tol_abs > tol) tol = tol_abs;
  bool keep = tid < n && sig > tol;
  if (tid < 64) {
    int rank = __builtin_popcountll(__ballot(keep));
    if (tid == 0) s_rank = rank;
  }
  if (tid < n) cwork[tid] = keep ? d / s2 : 0.0;  // (u.b)/sigma with u = a/sigma
  __syncthreads();
  if (tid < n) {
    double t = 0;
    for (int jj = 0; jj < n; jj++) t = fma(V[jj * ldv + tid], cwork[jj], t);
    x[tid] = t;
  }
  __syncthreads();
  return s_rank;
}

__device__ inline int wave_pinv_solve(int m, int n, double *A, int lda, double *V, int ldv,
                                      const double *b, double tol_abs, double tol_rel, double *x,
                                      double *cwork) {
  return block_pinv_solve<64>(m, n, A, lda, V, ldv, b, tol_abs, tol_rel, x, cwork);
}

}  // namespace lsqr
