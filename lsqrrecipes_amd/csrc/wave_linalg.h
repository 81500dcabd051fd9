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
// One-sided Jacobi sweeps (see block_pinv_solve): A (m x n, column-major) becomes U*S, V (n x n)
// accumulates the rotations.
template <int T>
__device__ inline int block_jacobi_svd(int m, int n, double *A, int lda, double *V, int ldv) {
  constexpr int LPP = T / 32;  // lanes per pair
  const int tid = threadIdx.x;
  for (int idx = tid; idx < n * n; idx += T) {
    int r = idx % n, c = idx / n;
    V[c * ldv + r] = (r == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int nn = (n + 1) & ~1;  // even number of players; index n is a dummy when n is odd
  const int p = tid / LPP, sub = tid % LPP;
  int sweep = 0;
  for (; sweep < 60; sweep++) {
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
  return sweep;
}

// The same sweeps for compile-time sizes: both columns of a pair (and of V) are fetched into registers in
// one batch of LDS reads, the rotation works on the registers and writes back -- one LDS round trip per
// round instead of one per element (the generic loops above wait for every dependent read).  Same
// arithmetic in the same order as block_jacobi_svd.
constexpr int pow2_floor(int x) { return x < 2 ? 1 : 2 * pow2_floor(x / 2); }

template <int T, int M, int N>
__device__ inline int block_jacobi_svd_fixed(double *A, int lda, double *V, int ldv) {
  // lanes per column pair: as many as the workgroup affords for the (N + 1) / 2 pairs of a round (a power of
  // two <= 64, so that a pair's lanes are consecutive inside one wave and combine with shuffles); 31 columns
  // on one wave: 16 pairs x 4 lanes, 8 rows per lane
  constexpr int NPAIR = ((N + 1) & ~1) / 2;
  constexpr int LPP = pow2_floor(T / NPAIR) > 64 ? 64 : pow2_floor(T / NPAIR);
  constexpr int RA = (M + LPP - 1) / LPP, RV = (N + LPP - 1) / LPP;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < N * N; idx += T) {
    int r = idx % N, c = idx / N;
    V[c * ldv + r] = (r == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  constexpr int nn = (N + 1) & ~1;
  const int p = tid / LPP, sub = tid % LPP;
  int sweep = 0;
  for (; sweep < 60; sweep++) {
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
        active = j < N;
      }
      const double *ai = A + (active ? i : 0) * lda, *aj = A + (active ? j : 0) * lda;
      const double *vi = V + (active ? i : 0) * ldv, *vj = V + (active ? j : 0) * ldv;
      double ui[RA], uj[RA], wi[RV], wj[RV];
#pragma unroll
      for (int q = 0; q < RA; q++) {
        const int k = sub + q * LPP;
        const bool ok = active && k < M;
        ui[q] = ok ? ai[k] : 0.0;
        uj[q] = ok ? aj[k] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < RV; q++) {
        const int k = sub + q * LPP;
        const bool ok = active && k < N;
        wi[q] = ok ? vi[k] : 0.0;
        wj[q] = ok ? vj[k] : 0.0;
      }
      double al = 0, be = 0, ga = 0;
#pragma unroll
      for (int q = 0; q < RA; q++) {
        al = fma(ui[q], ui[q], al);
        be = fma(uj[q], uj[q], be);
        ga = fma(ui[q], uj[q], ga);
      }
#pragma unroll
      for (int o = 1; o < LPP; o <<= 1) {
        al += __shfl_xor(al, o);
        be += __shfl_xor(be, o);
        ga += __shfl_xor(ga, o);
      }
      // |ga| > 4e-15 sqrt(al be), squared (no square root on the critical path; al be <= 1e300 for any
      // system the callers accept)
      const bool rot = active && ga != 0.0 && ga * ga > 1.6e-29 * (al * be);
      if (rot) {
        double zeta = (be - al) / (2.0 * ga);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        double *bi = A + i * lda, *bj = A + j * lda, *xi = V + i * ldv, *xj = V + j * ldv;
#pragma unroll
        for (int q = 0; q < RA; q++) {
          const int k = sub + q * LPP;
          if (k < M) {
            bi[k] = c * ui[q] - sn * uj[q];
            bj[k] = sn * ui[q] + c * uj[q];
          }
        }
#pragma unroll
        for (int q = 0; q < RV; q++) {
          const int k = sub + q * LPP;
          if (k < N) {
            xi[k] = c * wi[q] - sn * wj[q];
            xj[k] = sn * wi[q] + c * wj[q];
          }
        }
      }
      if constexpr (T == 64) {  // one wave: a ballot instead of the workgroup reduction
        __syncthreads();
        any_rot |= __any(rot) ? 1 : 0;
      } else {
        any_rot |= __syncthreads_or(rot ? 1 : 0);  // also the barrier between rounds
      }
    }
    if (!any_rot) break;
  }
  return sweep;
}

// M, N > 0: compile-time sizes (m == M, n == N), register-batched rounds (block_jacobi_svd_fixed)
template <int T, int M = 0, int N = 0>
__device__ inline int block_pinv_solve(int m, int n, double *A, int lda, double *V, int ldv,
                                       const double *b, double tol_abs, double tol_rel, double *x,
                                       double *cwork) {
  __shared__ int s_rank;
  __shared__ double s_smax;
  const int tid = threadIdx.x;
  if constexpr (M > 0) block_jacobi_svd_fixed<T, M, N>(A, lda, V, ldv);
  else block_jacobi_svd<T>(m, n, A, lda, V, ldv);
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

// Right singular vector of the smallest singular value of A (m x n, n <= 64; destroyed) -> x (n),
// unit length; *n_positive = number of singular values > 0 (vnl_svd's default rank: only exact zeros
// are dropped).  All T threads call it.
template <int T, int M = 0, int N = 0>
__device__ inline void block_null_vector(int m, int n, double *A, int lda, double *V, int ldv,
                                         double *x, int *n_positive) {
  __shared__ int s_jmin, s_npos;
  const int tid = threadIdx.x;
  if constexpr (M > 0) block_jacobi_svd_fixed<T, M, N>(A, lda, V, ldv);
  else block_jacobi_svd<T>(m, n, A, lda, V, ldv);
  double s2 = 0;
  if (tid < n)
    for (int k = 0; k < m; k++) {
      double a = A[tid * lda + k];
      s2 = fma(a, a, s2);
    }
  if (tid < 64) {
    double v = tid < n ? s2 : INFINITY;
    int idx = tid;
    for (int o = 32; o > 0; o >>= 1) {
      double ov = __shfl_xor(v, o);
      int oi = __shfl_xor(idx, o);
      if (ov < v || (ov == v && oi < idx)) v = ov, idx = oi;
    }
    int npos = __builtin_popcountll(__ballot(tid < n && s2 > 0.0));
    if (tid == 0) {
      s_jmin = idx;
      s_npos = npos;
    }
  }
  __syncthreads();
  if (tid < n) x[tid] = V[s_jmin * ldv + tid];
  if (tid == 0) *n_positive = s_npos;
  __syncthreads();
}

// Fast path of the n x n minimal solves: Gaussian elimination with partial pivoting on the system in
// LDS (A column-major, element (row i, col j) at A[j*lda + i]; A and b destroyed), T threads.  The
// reference solves through an SVD pseudo-inverse and declares the system singular when a singular value
// is <= 2.2e-16 (DenseLinear...Estimator.hxx:38-45); for a well-conditioned system both give the
// same solution to rounding (cond * eps).  The fast path only ACCEPTS when every pivot exceeds
// 1e-8 * max(1, max|A|) -- far from the rank decision -- and reports false otherwise, in which case the
// caller reloads the system and takes the SVD path, which makes the reference's rank decision.
template <int T>
__device__ inline bool block_gepp_solve(int n, double *A, int lda, double *b, double *x, double piv_rel = 1e-8) {
  __shared__ int s_p, s_fail;
  __shared__ double s_piv, s_amax, s_red[T / 64];
  const int tid = threadIdx.x;
  double am = 0.0;
  for (int idx = tid; idx < n * n; idx += T) {
    double v = fabs(A[(idx / n) * lda + idx % n]);
    am = v > am ? v : (v == v ? am : INFINITY);  // NaN poisons the fast path
  }
  am = wave_max(am);
  if ((tid & 63) == 0) s_red[tid >> 6] = am;
  if (tid == 0) s_fail = 0;
  __syncthreads();
  if (tid == 0) {
    double m = s_red[0];
    for (int w = 1; w < T / 64; w++) m = s_red[w] > m ? s_red[w] : m;
    s_amax = m;
  }
  __syncthreads();
  const double tol = piv_rel * (s_amax > 1.0 ? s_amax : 1.0);
  if (!(s_amax <= 1e150)) return false;
  for (int k = 0; k < n; k++) {
    if (tid < 64) {  // pivot search in column k (n <= 64: one element per lane)
      double v = (tid >= k && tid < n) ? fabs(A[k * lda + tid]) : -1.0;
      int idx = tid;
      for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_xor(v, o);
        int oi = __shfl_xor(idx, o);
        if (ov > v || (ov == v && oi < idx)) v = ov, idx = oi;
      }
      if (tid == 0) {
        s_p = idx;
        s_piv = A[k * lda + idx];
        if (!(v > tol)) s_fail = 1;
      }
    }
    __syncthreads();
    if (s_fail) return false;
    const int p = s_p;
    const double rpiv = 1.0 / s_piv;
    if (p != k) {
      for (int j = k + tid; j < n; j += T) {
        double t = A[j * lda + k];
        A[j * lda + k] = A[j * lda + p];
        A[j * lda + p] = t;
      }
      if (tid == T - 1) {
        double t = b[k];
        b[k] = b[p];
        b[p] = t;
      }
      __syncthreads();
    }
    for (int i = k + 1 + tid; i < n; i += T) A[k * lda + i] *= rpiv;  // multipliers
    __syncthreads();
    const int w = n - k - 1;
    for (int idx = tid; idx < w * w; idx += T) {
      const int i = k + 1 + idx % w, j = k + 1 + idx / w;
      A[j * lda + i] = fma(-A[k * lda + i], A[j * lda + k], A[j * lda + i]);
    }
    for (int i = k + 1 + tid; i < n; i += T) b[i] = fma(-A[k * lda + i], b[k], b[i]);
    __syncthreads();
  }
  for (int k = n - 1; k >= 0; k--) {  // back substitution
    if (tid == 0) x[k] = b[k] / A[k * lda + k];
    __syncthreads();
    const double xk = x[k];
    for (int i = tid; i < k; i += T) b[i] = fma(-A[k * lda + i], xk, b[i]);
    __syncthreads();
  }
  return true;
}

// The same elimination by ONE wave (r04): the system in LDS is private to the wave, lane = row (n <= 64), no
// workgroup barriers -- the LDS serves a wave's requests in order, a compiler barrier between dependent phases is all
// it takes.  block_gepp_solve<256> spends its time in barriers (five per pivot step, 64 steps: 0.30 ms per 64 x 64
// system whatever the chip is doing); here a pivot step is a butterfly over the column, a row swap, and n - k - 1
// updates of one column each (one broadcast read of the pivot row's element, one read, one fma, one write per lane).
// Same pivot rule, same multipliers, same fma updates, same back substitution: the result is BIT-IDENTICAL to
// block_gepp_solve's (tests/test_gpu_dense_fused.py compares them).  Four systems per workgroup of 256 threads.
__device__ inline bool wave_gepp_solve(int n, double *A, int lda, double *b, double *x, double piv_rel = 1e-8,
                                       double piv_abs = 0.0) {
  const int lane = threadIdx.x & 63;
  double am = 0.0;
  for (int idx = lane; idx < n * n; idx += 64) {
    const double v = fabs(A[(idx / n) * lda + idx % n]);
    am = v > am ? v : (v == v ? am : INFINITY);  // NaN poisons the fast path
  }
  am = wave_max(am);
  if (!(am <= 1e150)) return false;
  double tol = piv_rel * (am > 1.0 ? am : 1.0);
  tol = tol > piv_abs ? tol : piv_abs;  // (callers whose rank decision is an absolute threshold: k_estimate_us)
  for (int k = 0; k < n; k++) {
    // pivot search in column k: the butterfly of block_gepp_solve (largest |a|, ties to the lower row)
    double v = (lane >= k && lane < n) ? fabs(A[k * lda + lane]) : -1.0;
    int idx = lane;
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(v, o);
      const int oi = __shfl_xor(idx, o);
      if (ov > v || (ov == v && oi < idx)) v = ov, idx = oi;
    }
    const int p = __builtin_amdgcn_readfirstlane(idx);
    if (!(v > tol)) return false;               // (v is the same in every lane after the butterfly)
    const double rpiv = 1.0 / A[k * lda + p];
    if (p != k) {
      if (lane >= k && lane < n) {
        const double t = A[lane * lda + k];
        A[lane * lda + k] = A[lane * lda + p];
        A[lane * lda + p] = t;
      }
      if (lane == 0) {
        const double t = b[k];
        b[k] = b[p];
        b[p] = t;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const bool below = lane > k && lane < n;
    double mult = 0.0;
    if (below) {
      mult = A[k * lda + lane] * rpiv;
      A[k * lda + lane] = mult;
    }
    __builtin_amdgcn_wave_barrier();
    const double bk = b[k];
    // eight columns per round: all their reads are issued before the first fma (a column at a time ran at the LDS
    // latency: read, fma, write, read ...)
    int j = k + 1;
    for (; j + 8 <= n; j += 8) {
      double akj[8], aij[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        akj[q] = A[(j + q) * lda + k];          // the pivot row's element: one address for the wave
        aij[q] = A[(j + q) * lda + (below ? lane : k)];
      }
#pragma unroll
      for (int q = 0; q < 8; q++)
        if (below) A[(j + q) * lda + lane] = fma(-mult, akj[q], aij[q]);
    }
    for (; j < n; j++) {
      const double akj = A[j * lda + k];
      if (below) A[j * lda + lane] = fma(-mult, akj, A[j * lda + lane]);
    }
    if (below) b[lane] = fma(-mult, bk, b[lane]);
    __builtin_amdgcn_wave_barrier();
  }
  for (int k = n - 1; k >= 0; k--) {  // back substitution
    const double xk = b[k] / A[k * lda + k];
    if (lane == 0) x[k] = xk;
    __builtin_amdgcn_wave_barrier();
    if (lane < k) b[lane] = fma(-A[k * lda + lane], xk, b[lane]);
    __builtin_amdgcn_wave_barrier();
  }
  return true;
}


// The same elimination for n = 64 with the system IN REGISTERS: lane = row (its 64 coefficients and right-hand side in
// 130 registers), the pivot row's elements broadcast with v_readlane as scalar operands of the update's fma -- no LDS
// round trip per column, 64 steps of ~(64 - k) fmas and as many pairs of v_readlane instead of 64 steps of dependent LDS
// reads and writes (~2 us each).  Rows are never moved: a row keeps its lane and a POSITION that the pivot steps
// exchange, which is all wave_gepp_solve's swap does to the arithmetic -- same pivot rule (largest |a|, ties to the
// lower position), same multipliers, same fma updates, same back substitution: BIT-IDENTICAL results
// (tests/test_gpu_dense_fused.py).  x[k] is returned in lane k.
__device__ __forceinline__ double wave_readlane_f64(double v, int l) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ inline bool wave_gepp_solve_reg64(double (&a)[64], double bv, double &xv, double piv_rel = 1e-8) {
  const int lane = threadIdx.x & 63;
  double am = 0.0;
#pragma unroll
  for (int c = 0; c < 64; c++) {
    const double v = fabs(a[c]);
    am = v > am ? v : (v == v ? am : INFINITY);  // NaN poisons the fast path
  }
  am = wave_max(am);
  if (!(am <= 1e150)) return false;
  const double tol = piv_rel * (am > 1.0 ? am : 1.0);
  int pos = lane;  // the position of my row (wave_gepp_solve's row index after its swaps)
  bool solved = true;
#pragma unroll
  for (int k = 0; k < 64; k++) {
    double v = pos >= k ? fabs(a[k]) : -1.0;
    int idx = pos;
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(v, o);
      const int oi = __shfl_xor(idx, o);
      if (ov > v || (ov == v && oi < idx)) v = ov, idx = oi;
    }
    const int p = __builtin_amdgcn_readfirstlane(idx);
    // (v is the same in every lane after the butterfly.  A refused pivot ends wave_gepp_solve there; here the steps
    // run on -- the loop has to be free of exits for the compiler to unroll it and keep the rows in registers -- and
    // the result is discarded)
    solved = solved && v > tol;
    const int lp = __builtin_ctzll(__ballot(pos == p) | (1ull << 63));  // the lane that holds the pivot row
    const double rpiv = 1.0 / wave_readlane_f64(a[k], lp);
    if (p != k) pos = pos == p ? k : pos == k ? p : pos;
    const bool below = pos > k;
    double mult = 0.0;
    if (below) {
      mult = a[k] * rpiv;
      a[k] = mult;
    }
    const double bk = wave_readlane_f64(bv, lp);
#pragma unroll
    for (int j = k + 1; j < 64; j++) {
      const double akj = wave_readlane_f64(a[j], lp);
      if (below) a[j] = fma(-mult, akj, a[j]);
      // (the scheduler would fetch a whole step's pivot row first: 126 scalar registers, 394 spilled)
      if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    if (below) bv = fma(-mult, bk, bv);
  }
  xv = 0.0;
#pragma unroll
  for (int k = 63; k >= 0; k--) {  // back substitution
    const int lk = __builtin_ctzll(__ballot(pos == k) | (1ull << 63));
    const double xk = wave_readlane_f64(bv, lk) / wave_readlane_f64(a[k], lk);
    if (lane == k) xv = xk;
    if (pos < k) bv = fma(-a[k], xk, bv);
  }
  return solved;
}

__device__ inline int wave_pinv_solve(int m, int n, double *A, int lda, double *V, int ldv,
                                      const double *b, double tol_abs, double tol_rel, double *x,
                                      double *cwork) {
  return block_pinv_solve<64>(m, n, A, lda, V, ldv, b, tol_abs, tol_rel, x, cwork);
}

}  // namespace lsqr
