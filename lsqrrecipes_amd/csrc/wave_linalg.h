// wave_linalg.h -- one-wavefront dense solvers on matrices staged in LDS (device only).
//
// wave_pinv_solve: x = pinv(A) b with small singular values zeroed, by one-sided (Hestenes)
// Jacobi -- what the reference gets from vnl_matrix_inverse + zero_out_absolute
// (DenseLinearEquationSystemParametersEstimator.hxx:38-45, SinglePointTarget...cxx:192-201).
// One wave64 per system: the n/2 disjoint column pairs of a round-robin round are rotated in
// parallel, two lanes per pair (each takes every other row); matrices are column-major in LDS
// with an odd leading dimension so the pair/row interleave is bank-conflict free.
// The workgroup calling these must be exactly one wave (64 threads).
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
__device__ inline int wave_pinv_solve(int m, int n, double *A, int lda, double *V, int ldv,
                                      const double *b, double tol_abs, double tol_rel, double *x,
                                      double *cwork) {
  const int lane = threadIdx.x & 63;
  for (int idx = lane; idx < n * n; idx += 64) {
    int r = idx % n, c = idx / n;
    V[c * ldv + r] = (r == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int nn = (n + 1) & ~1;  // even number of players; index n is a dummy when n is odd
  const int p = lane >> 1, half = lane & 1;
  for (int sweep = 0; sweep < 60; sweep++) {
    bool any_rot = false;
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
        for (int k = half; k < m; k += 2) {
          double ui = A[i * lda + k], uj = A[j * lda + k];
          al = fma(ui, ui, al);
          be = fma(uj, uj, be);
          ga = fma(ui, uj, ga);
        }
      al += __shfl_xor(al, 1);
      be += __shfl_xor(be, 1);
      ga += __shfl_xor(ga, 1);
      // columns count as orthogonal once their cosine is at the rounding floor of an m-term dot
      // product (a tighter bound only re-rotates noise until the sweep limit)
      bool rot = active && ga != 0.0 && fabs(ga) > 4e-15 * sqrt(al * be);
      if (rot) {
        double zeta = (be - al) / (2.0 * ga);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int k = half; k < m; k += 2) {
          double ui = A[i * lda + k], uj = A[j * lda + k];
          A[i * lda + k] = c * ui - s * uj;
          A[j * lda + k] = s * ui + c * uj;
        }
        for (int k = half; k < n; k += 2) {
          double vi = V[i * ldv + k], vj = V[j * ldv + k];
          V[i * ldv + k] = c * vi - s * vj;
          V[j * ldv + k] = s * vi + c * vj;
        }
      }
      any_rot = any_rot || __any(rot);
      __syncthreads();
    }
    if (!any_rot) break;
  }
  // singular values and projections of b
  double s2 = 0, d = 0;
  if (lane < n)
    for (int k = 0; k < m; k++) {
      double a = A[lane * lda + k];
      s2 = fma(a, a, s2);
      d = fma(a, b[k], d);
    }
  double sig = sqrt(s2);
  double smax = wave_max(lane < n ? sig : 0.0);
  double tol = tol_rel * smax;
  if (tol_abs > tol) tol = tol_abs;
  bool keep = lane < n && sig > tol;
  int rank = __builtin_popcountll(__ballot(keep));
  if (lane < n) cwork[lane] = keep ? d / s2 : 0.0;  // (u.b)/sigma with u = a/sigma
  __syncthreads();
  if (lane < n) {
    double t = 0;
    for (int jj = 0; jj < n; jj++) t = fma(V[jj * ldv + lane], cwork[jj], t);
    x[lane] = t;
  }
  __syncthreads();
  return rank;
}

}  // namespace lsqr
