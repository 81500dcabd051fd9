// lm_persist.h -- a whole Levenberg-Marquardt fit in ONE launch (r05).
//
// The iterative final fits of the US calibrations (SinglePointTargetUSCalibrationParametersEstimator.cxx:272-329,
// :926-971) take thousands of function evaluations at the reference's 1e-15 tolerances.  r02-r04 ran every evaluation
// as two launches (k_lm_pass_mfma_t + k_lm_publish) with the host's MINPACK step between them: 28 us per evaluation
// for a fit alone on the device, of which the pass itself is 15.  Here the workgroups stay resident for the whole fit:
//
//   every workgroup    pass over its share of the compacted consensus set (tiles of 64 frames, field-major:
//                      kernels.h: k_compact_write_tiles), (J | f)^T (J | f) on the fp64 matrix cores exactly as
//                      k_lm_pass_mfma_t forms it, block partials written through (sc1 stores), one arrival per workgroup
//   workgroup 0        waits for the arrivals, sums the partials in k_lm_publish's order, and then EITHER runs MINPACK's
//                      step itself (lm_core.h: lm_advance on one lane, state in LDS -- `host_step` 0) OR hands the
//                      moment block to the host through tagged granules in pinned memory and takes the next trial
//                      point's coefficients back the same way (`host_step` 1); the coefficients of the next evaluation
//                      go to every workgroup as tagged 8-byte granules {payload word | evaluation tag} in device memory
//   every workgroup    polls those granules (a granule is valid the moment its tag matches: no flag, no fence) and starts
//                      the next pass
//
// No kernel boundary, no launch latency and no k_lm_publish per evaluation.  The SUMS ARE THE LAUNCH PATH'S BIT FOR BIT:
// the work is cut into the same nb "virtual" blocks of four waves (tile t belongs to virtual wave t mod 4 nb), each
// virtual wave accumulates its tiles in order from zero, the four waves of a block are folded in order, the blocks are
// summed lane-strided and by the same shuffle tree -- whatever the number of resident workgroups G.  The trial points'
// coefficients come from lsqr_sincos (small_linalg.h: the same bits on the host and the device), so the host-stepped
// launches, the persistent kernel with the host's step and the persistent kernel with its own step walk through the
// same iterates and stop with the same info / nfev (tests/test_gpu_lm_persist.py).
//
// Residency: G workgroups must be resident together.  The host side (lsqr_hip.hip: run_lm_persist) draws G "compute
// unit tokens" from a per-device pool before it launches, so the persistent kernels of one process never ask for more
// than the chip holds; every wait in the kernel is bounded all the same (another process may share the device): a
// workgroup whose wait expires sets `abort`, every workgroup leaves at its next poll, and the host falls back to the
// launch path.  The dynamic LDS request (> 80 KB) keeps it to one persistent workgroup per compute unit.
#pragma once
#include "kernels.h"

namespace lsqr {

enum { LMP_EVAL = 1, LMP_FIN = 2, LMP_ABORT = 3 };
enum { LMP_MAXCOEF = 48, LMP_TRACE = 64, LMP_NBMAX = 512 };  // (NBMAX: the launch path cuts a pass into <= 512 blocks)

struct LmpCtl {                    // device memory; zeroed (stream-ordered) before every launch
  unsigned int arrive;             // workgroups 1 .. G-1 add one per evaluation
  unsigned int pad0[31];
  unsigned int abort;              // set by a workgroup whose bounded wait expired
  unsigned int pad1[31];
  unsigned long long gran[2 * LMP_MAXCOEF + 2];  // broadcast: coefficient word w as {word | tag}; last: {command | tag}
  unsigned long long pad2[30];
  // workgroup 0's clock (100 MHz) per evaluation: own pass done, all arrived, partials summed, step done / reply in
  unsigned long long trace[LMP_TRACE][4];
  unsigned long long t_begin, t_end;
  unsigned int evals, status;      // evaluations run; LMP_FIN / LMP_ABORT
};

struct LmpInit {
  int n, maxfev;
  double ftol, xtol, gtol, factor;
  double x0[LM_NMAX];
};

__device__ inline unsigned long long lmp_clock() { return wall_clock64(); }

// one bounded poll of an agent-scope word: true when (uint32)value == tag
template <int SCOPE>
__device__ inline bool lmp_wait_tag(const unsigned long long *p, uint32_t tag, const unsigned int *abort_flag,
                                    unsigned long long timeout, unsigned long long *val) {
  unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE);
  if ((uint32_t)v == tag) {
    *val = v;
    return true;
  }
  const unsigned long long t0 = lmp_clock();
  for (unsigned spins = 1;; spins++) {
    __builtin_amdgcn_s_sleep(1);
    v = __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE);
    if ((uint32_t)v == tag) {
      *val = v;
      return true;
    }
    if ((spins & 31) == 0) {
      if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
      if (lmp_clock() - t0 > timeout) return false;
    }
  }
}

// RT: tiles of 64 frames every wave keeps IN REGISTERS for the whole fit (0: every evaluation streams its tiles).  With
// one round of virtual blocks per evaluation (G * slots >= nb: the fit has the device to itself) a wave owns the same
// <= ceil(ntiles / 4 nb) tiles in every evaluation -- at BASELINE config 5 four tiles of 14 doubles per lane = 112 of the
// 256 registers a wave of a 512-thread workgroup may use -- and the pass stops reading memory at all.
template <class M, int RT>
__global__ __launch_bounds__(RT ? 512 : 1024) void k_lm_persist(const double *__restrict__ tiles, size_t n, int nb,
                                                                LmpCtl *__restrict__ ctl, double *__restrict__ partials,
                                                                unsigned long long *__restrict__ h_res,
                                                                const unsigned long long *__restrict__ h_cmd,
                                                                LmState *__restrict__ st_out, SolveOut *__restrict__ out,
                                                                LmpInit init, uint32_t seq0, int host_step,
                                                                unsigned long long timeout, uint32_t test_abort_at) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef typename M::LmCoef Coef;
  constexpr int NL = M::NLM, NMOM = M::NMOM_LM, P = 17, REC = M::REC;
  constexpr int NCW = (int)(sizeof(Coef) / 8), NG = 2 * NCW + 1;  // coefficient doubles; granules (+ the command)
  static_assert(NCW <= LMP_MAXCOEF && sizeof(Coef) % 8 == 0, "coefficient block");
  static_assert(NL + 1 <= 16, "one 16 x 16 accumulator tile");
  extern __shared__ double s_dyn[];  // one 64 x P row tile per wave (the fold of a virtual block reuses it)
  __shared__ double s_coef[LMP_MAXCOEF];
  __shared__ double s_mom[LM_MOM_MAX];
  __shared__ LmState s_st;
  __shared__ int s_cmd, s_bail;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = (int)(blockDim.x >> 6), slots = NW >> 2;
  const int k4 = lane >> 4, c16 = lane & 15;
  const int g = blockIdx.x, G = gridDim.x;
  // workgroups 0 .. R-1 sum the block partials, each its share of the moments: ONE compute unit takes in ~35-45 GB/s, and
  // the 356 KB of 489 blocks x 91 moments were 11 us of an evaluation on workgroup 0 alone.  With the host's step every
  // reducer publishes its moments to the host itself; the device's step needs them in one place (R = 1).
  const int R = host_step ? (G < 8 ? G : 8) : 1;
  double *tile = s_dyn + (size_t)wave * (64 * P);
  const size_t ntiles = (n + 63) / 64, W = (size_t)nb * 4;
  if (tid == 0) {
    s_bail = 0;
    if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      s_cmd = LMP_ABORT;  // (a workgroup that became resident after the others gave up)
    } else {
      Coef k;
      M::lm_coef(init.x0, k);  // every workgroup forms the first trial point's coefficients itself
      const double *kd = (const double *)&k;
      for (int i = 0; i < NCW; i++) s_coef[i] = kd[i];
      s_cmd = LMP_EVAL;
      if (g == 0) ctl->t_begin = lmp_clock();
    }
  }
  // the resident tiles (RT > 0: one round, so this wave's virtual wave never changes)
  double rx[RT ? RT : 1][REC];
  if constexpr (RT > 0) {
    const int vb = g * slots + (wave >> 2);
    const size_t vw = (size_t)vb * 4 + (wave & 3);
#pragma unroll
    for (int i = 0; i < RT; i++) {
      const size_t t = vw + (size_t)i * W;
      const bool have = vb < nb && t < ntiles;
      const double *src = tiles + (have ? t : 0) * (size_t)(REC * 64) + lane;
#pragma unroll
      for (int j = 0; j < REC; j++) rx[i][j] = (j == 12 && M::IS_US) || !have ? 0.0 : src[(size_t)j * 64];
    }
  }
  __syncthreads();
  uint32_t e = 0;
  while (s_cmd == LMP_EVAL) {
    e++;
    const uint32_t tag = seq0 + e;
    Coef coef;  // wave-uniform: scalar registers
    {
      uint32_t *d = (uint32_t *)&coef;
      const uint32_t *s = (const uint32_t *)s_coef;
#pragma unroll
      for (int i = 0; i < 2 * NCW; i++) d[i] = __builtin_amdgcn_readfirstlane(s[i]);
    }
    // ---- the pass: virtual blocks g * slots + s, + G * slots, ... --------------------------------------------------
    for (int vb0 = g * slots; vb0 < nb; vb0 += G * slots) {
      const int vb = vb0 + (wave >> 2);
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
      auto tile_step = [&](const double *x, size_t t) {  // k_lm_pass_mfma_t's body: rows -> LDS -> 16 matrix instructions
        double z[16];
#pragma unroll
        for (int j = 0; j < 16; j++) z[j] = 0.0;
        if (t * 64 + lane < n) M::lm_row(x, coef, z);
#pragma unroll
        for (int j = 0; j < 16; j++) tile[lane * P + j] = z[j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 16; s++) {
          const double v = tile[(4 * s + k4) * P + c16];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        // (the resident tiles' steps are unrolled: without this the scheduler interleaves four of them and their
        // temporaries spill next to the 112 resident registers)
        __builtin_amdgcn_sched_barrier(0);
      };
      if (vb < nb) {
        const size_t vw = (size_t)vb * 4 + (wave & 3);
        size_t t0 = vw;
        if constexpr (RT > 0) {
#pragma unroll
          for (int i = 0; i < RT; i++) {
            const size_t t = vw + (size_t)i * W;
            if (t < ntiles) tile_step(rx[i], t);
          }
          t0 = vw + (size_t)RT * W;
        }
        if (t0 < ntiles) {  // streamed tiles (all of them when RT == 0), the next one in flight
          double nx[REC];
          auto fetch = [&](size_t t) {
            const double *src = tiles + t * (size_t)(REC * 64) + lane;
#pragma unroll
            for (int j = 0; j < REC; j++) nx[j] = (j == 12 && M::IS_US) ? 0.0 : src[(size_t)j * 64];
          };
          fetch(t0);
          for (size_t t = t0; t < ntiles; t += W) {
            double x[REC];
#pragma unroll
            for (int j = 0; j < REC; j++) x[j] = nx[j];
            if (t + W < ntiles) fetch(t + W);
            tile_step(x, t);
          }
        }
      }
      // fold the four virtual waves of each block in order (k_lm_pass_mfma_t: fold = ((0 + a0) + a1) + a2) + a3);
      // D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
      for (int rg = 0; rg < 4; rg++) tile[(k4 + 4 * rg) * 16 + c16] = acc[rg];
      __syncthreads();
      {
        const int s = tid >> 8, pos = tid & 255, vbs = vb0 + s;
        if (vbs < nb) {
          const double *a0 = s_dyn + (size_t)(4 * s) * (64 * P);
          double f = 0.0 + a0[pos];
          f += a0[64 * P + pos];
          f += a0[2 * 64 * P + pos];
          f += a0[3 * 64 * P + pos];
          const int r = pos >> 4, cc = pos & 15;
          int idx = -1;
          if (r < NL && cc < NL && r <= cc) idx = 1 + r * NL - r * (r - 1) / 2 + (cc - r);
          if (r < NL && cc == NL) idx = 1 + NL * (NL + 1) / 2 + r;
          if (r == NL && cc == NL) idx = 0;
          if (idx >= 0)  // written through: the reducers read it with sc1 loads after the arrivals; MOMENT-major, so
                         // that their sums read 64 consecutive blocks of one moment per load
            __hip_atomic_store(partials + (size_t)idx * LMP_NBMAX + vbs, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
    }
    // ---- arrival: every storing wave has drained its stores before the one lane signals ----------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && G > 1) __hip_atomic_fetch_add(&ctl->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool tr = g == 0 && e <= (uint32_t)LMP_TRACE;
    if (g < R) {
      // ---- a reducer: wait for every workgroup, then this workgroup's share of the moments -------------------------
      if (tid == 0) {
        if (tr) ctl->trace[e - 1][0] = lmp_clock();
        if (G > 1) {
          const uint32_t want = e * (uint32_t)G;
          const unsigned long long t0 = lmp_clock();
          for (unsigned spins = 1; __hip_atomic_load(&ctl->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want;
               spins++) {
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 31) == 0 && (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
                                      lmp_clock() - t0 > timeout)) {
              s_bail = 1;
              break;
            }
          }
        }
        if (tr) ctl->trace[e - 1][1] = lmp_clock();
        if (g == 0 && e == test_abort_at) s_bail = 1;  // (tests: as if a wait had expired at this evaluation)
      }
      __syncthreads();
      if (!s_bail) {
        // moment k = g + R i: the blocks' partials summed as k_lm_publish sums them -- lane l takes blocks l, l + 64, ...
        // in order, then the shuffle tree; all eight loads of a moment in flight
        for (int k = g + R * wave; k < NMOM; k += R * NW) {
          const double *pk = partials + (size_t)k * LMP_NBMAX + lane;
          double a[LMP_NBMAX / 64];
#pragma unroll
          for (int j = 0; j < LMP_NBMAX / 64; j++)
            a[j] = lane + 64 * j < nb ? __hip_atomic_load(pk + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
          double t = 0.0;
#pragma unroll
          for (int j = 0; j < LMP_NBMAX / 64; j++)
            if (lane + 64 * j < nb) t += a[j];
          for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
          if (lane == 0) {
            if (host_step) {
              // to the host (pinned memory): the two tagged granules of the moment (k_lm_publish's format) in one 16-byte store
              const unsigned long long bits = __builtin_bit_cast(unsigned long long, t);
              const u32x4 v = {tag, (unsigned int)(bits >> 32), tag, (unsigned int)bits};
              asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(h_res + 2 * k), "v"(v) : "memory");
            } else {
              s_mom[k] = t;
            }
          }
        }
      }
    }
    if (g == 0) {
      if (!host_step) {
        __syncthreads();
        if constexpr (RT == 0)  // (the resident variant is launched with the host's step only: MINPACK's locals next to
                                // 112 registers of resident tiles would spill into the pass)
        if (!s_bail) {
          if (tid == 0) {
            if (tr) ctl->trace[e - 1][2] = lmp_clock();
            if (e == 1) lm_init(s_st, init.n, init.x0, init.ftol, init.xtol, init.gtol, init.maxfev, init.factor);
            const bool cont = lm_advance(s_st, s_mom);
            if (cont) {
              Coef k;
              M::lm_coef(s_st.xtrial, k);
              const double *kd = (const double *)&k;
              for (int i = 0; i < NCW; i++) s_coef[i] = kd[i];
            } else {
              const bool ok = s_st.info >= 1 && s_st.info <= 4;  // vnl_levenberg_marquardt::minimize -> true
              out->cont = 0;
              out->lm_info = s_st.info;
              out->lm_nfev = s_st.nfev;
              out->pad = s_st.stall;
              out->ok = ok ? 1 : 0;
              out->cost = s_st.fnorm * s_st.fnorm;
              const int np = M::lm_finalize(s_st.x, out->params);
              out->n_params = ok ? np : 0;
            }
            s_cmd = cont ? LMP_EVAL : LMP_FIN;
            if (tr) ctl->trace[e - 1][3] = lmp_clock();
          }
          __syncthreads();
          if (tid < NG) {
            const uint32_t w = tid < 2 * NCW ? ((const uint32_t *)s_coef)[tid ^ 1] : (uint32_t)s_cmd;
            __hip_atomic_store(&ctl->gran[tid], ((unsigned long long)w << 32) | tag, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      } else if (wave == 0 && !s_bail) {
        // the host's reply: ONE lane polls the command granule (the host writes it last; a poll by 85 lanes is 85 PCIe
        // reads), then one coalesced 16-byte-per-lane read of all granules, every tag checked (a granule whose tag is not
        // this evaluation's yet is simply read again); forwarded to the other workgroups as it stands
        if (lane == 0 && tr) ctl->trace[e - 1][2] = lmp_clock();
        unsigned long long v0 = 0;
        bool ok = true;
        if (lane == 0) ok = lmp_wait_tag<__HIP_MEMORY_SCOPE_SYSTEM>(&h_cmd[2 * NCW], tag, &ctl->abort, timeout, &v0);
        ok = __shfl((int)ok, 0) != 0;
        constexpr int NPAIR = (NG + 1) / 2;  // lanes reading two granules each
        u32x4 r = {0, 0, 0, 0};
        for (int tries = 0; ok; tries++) {
          if (lane < NPAIR)
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=v"(r)
                         : "v"(h_cmd + 2 * lane)
                         : "memory");
          const bool mine = lane >= NPAIR || (r.x == tag && (2 * lane + 1 >= NG || r.z == tag));
          if (__ballot(mine) == ~0ULL) break;
          if (tries > 1000) ok = false;
          __builtin_amdgcn_s_sleep(2);
        }
        if (ok) {
          if (lane < NPAIR) {
            const unsigned long long g0 = ((unsigned long long)r.y << 32) | r.x, g1 = ((unsigned long long)r.w << 32) | r.z;
            __hip_atomic_store(&ctl->gran[2 * lane], g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (2 * lane < 2 * NCW) ((uint32_t *)s_coef)[(2 * lane) ^ 1] = r.y;
            else s_cmd = (int)r.y;
            if (2 * lane + 1 < NG) {
              __hip_atomic_store(&ctl->gran[2 * lane + 1], g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (2 * lane + 1 < 2 * NCW) ((uint32_t *)s_coef)[(2 * lane + 1) ^ 1] = r.w;
              else s_cmd = (int)r.w;
            }
          }
          if (lane == 0 && tr) ctl->trace[e - 1][3] = lmp_clock();
        } else if (lane == 0) {
          s_bail = 1;
        }
      }
    } else if (!(g < R && s_bail)) {
      // the next trial point (or the end): thread i < NG polls granule i
      if (tid < NG) {
        unsigned long long v;
        if (lmp_wait_tag<__HIP_MEMORY_SCOPE_AGENT>(&ctl->gran[tid], tag, &ctl->abort, timeout, &v)) {
          if (tid < 2 * NCW) ((uint32_t *)s_coef)[tid ^ 1] = (uint32_t)(v >> 32);
          else s_cmd = (int)(v >> 32);
        } else {
          s_bail = 1;
        }
      }
    }
    __syncthreads();
    if (s_bail) {  // a bounded wait expired (or another workgroup said so): tell everyone, leave
      if (tid == 0) {
        __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_cmd = LMP_ABORT;
      }
      __syncthreads();
    }
  }
  if (g == 0 && tid == 0) {
    ctl->evals = e;
    ctl->status = (unsigned)s_cmd;
    ctl->t_end = lmp_clock();
    if (!host_step && s_cmd == LMP_FIN) *st_out = s_st;
  }
}

}  // namespace lsqr
