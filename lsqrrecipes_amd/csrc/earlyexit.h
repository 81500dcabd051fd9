// earlyexit.h -- chunked early exit of the agree() scans that have no spatial index (dense system, US calibrations):
// the batched form of RANSAC.hxx:94.
//
// The serial loop abandons a hypothesis as soon as  votes so far + observations still to come  can no longer exceed
// the best count seen (RANSAC.hxx:94), and only a hypothesis that becomes the best-so-far changes the loop's state
// (strict '>', :100).  Batched, over the N observations of the upload:
//   A   the first sixteenth of the observations for every hypothesis of the batch;
//   B   (only when the best count of A says that the best model agrees with >= 40 % of the observations -- else the
//       second pass over the observations costs more than it saves, see k_ee_split)
//       the CANDIDATES -- hypotheses with at least half the largest count of A -- are counted to the end first: their
//       final counts are what gives the bound below its teeth (after one sixteenth nobody's partial count is large).
//       With Lg = the best final count (or the best of earlier batches), a hypothesis without votes can be abandoned
//       once R = N - b <= Lg observations remain: the plan (k_ee_plan, on the device -- no host round trip) puts the
//       next boundary at b* = N - Lg + N / 64 and halves the remainder twice;
//   C   the others run [N/16, b*), [b*, mid), [mid, N); after each chunk hypothesis h stays alive only if
//           votes[h] + R > L[h],    R = observations not yet scanned,
//           L[h] = max(best of earlier batches, max over ALL h' < h of votes[h'])
//       -- votes[h'] being final for a candidate and partial otherwise: either way a lower bound of the final count.
//       The scan kernels take their observation range and their hypothesis count from device memory.
// Exactness.  A dropped h has  final[h] <= votes[h] + R <= L[h]: some earlier hypothesis (or an earlier batch) already
// holds at least as many votes.  If that earlier hypothesis is itself dropped later, the same argument applies to it
// with a still earlier one; the chain ends at a hypothesis counted to the end.  So when the serial loop reaches h its
// running maximum is >= final[h] and the strict '>' does not fire: winner, iteration count (numTries is only updated
// on a new maximum) and consensus set are unchanged; a dropped hypothesis reports its partial count, which is <= L[h]
// and therefore inert in the replay as well.  lsqr_scan (the explicit entry point) always counts everything.
// What it can save is bounded by the data: a wrong model is abandoned when R <= (best count) -- with the best model
// agreeing with a fraction f of the observations, after (1 - f) N of them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsqr {

struct EeState {
  uint32_t n_cand, n_alive;  // candidates (pass B) / other hypotheses still alive
  uint32_t n_drop_first;     // hypotheses dropped at the first selection (diagnostics)
  uint32_t chunks;           // selections done
  unsigned long long work;   // (observation, hypothesis) pairs handed to the scan kernels so far
  uint32_t rng[4][2];        // observation ranges [begin, end): 0 = pass B, 1..3 = the chunks of pass C
  uint32_t pad[2];
};
static_assert(sizeof(EeState) == 64, "copied to a 64-byte pinned area");
constexpr int kEeChunksC = 3;

// block-wide helpers (1024 threads): exclusive scan of one value per thread through s[1024]
__device__ inline uint32_t ee_scan_add(uint32_t *s, uint32_t v, int t, uint32_t *total) {
  s[t] = v;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const uint32_t a = t >= o ? s[t - o] : 0u;
    __syncthreads();
    s[t] += a;
    __syncthreads();
  }
  const uint32_t incl = s[t];
  *total = s[1023];
  __syncthreads();
  return incl - v;
}
__device__ inline uint32_t ee_scan_max(uint32_t *s, uint32_t v, int t) {  // exclusive prefix maximum
  s[t] = v;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const uint32_t a = t >= o ? s[t - o] : 0u;
    __syncthreads();
    s[t] = a > s[t] ? a : s[t];
    __syncthreads();
  }
  const uint32_t excl = t ? s[t - 1] : 0u;
  __syncthreads();
  return excl;
}

// after pass A (observations [0, b1), all H hypotheses): candidates and the others, both in index order
__global__ __launch_bounds__(1024) void k_ee_split(const uint32_t *__restrict__ votes, const uint8_t *__restrict__ valid,
                                                   uint32_t H, uint32_t *__restrict__ sel_c, uint32_t *__restrict__ sel_o,
                                                   EeState *__restrict__ st, uint32_t b1, uint32_t n) {
  __shared__ uint32_t s_red[16], s_scan[1024];
  const int t = threadIdx.x;
  if (H > kSelCap) return;  // 1024 threads x 8 hypotheses (models.h)
  uint32_t mx = 0;
  for (uint32_t h = t; h < H; h += 1024) mx = valid[h] && votes[h] > mx ? votes[h] : mx;
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t a = __shfl_down(mx, o);
    mx = a > mx ? a : mx;
  }
  if ((t & 63) == 0) s_red[t >> 6] = mx;
  __syncthreads();
  mx = 0;
  for (int w = 0; w < 16; w++) mx = s_red[w] > mx ? s_red[w] : mx;
  const uint32_t thr = mx - mx / 2;  // ceil(mx / 2)
  // Pass B only pays when the best model agrees with a large share of the observations (it is a second pass over the
  // observations): with a share f the final count lets a wrong model go after (1 - f) N observations, the partial
  // counts alone after N / (1 + f) -- for f = 0.3 that is 0.70 N against 0.77 N.  Below f = 0.4 (estimated from
  // pass A) there are no candidates and pass C runs on partial counts.
  const bool pass_b = (unsigned long long)mx * 5 >= (unsigned long long)b1 * 2;
  uint32_t fc[8], fo[8], nc = 0, no = 0;
  for (int k = 0; k < 8; k++) {
    const uint32_t h = t * 8 + k;
    const bool v = h < H && valid[h];
    fc[k] = (v && pass_b && mx > 0 && votes[h] >= thr) ? 1u : 0u;
    fo[k] = (v && !fc[k]) ? 1u : 0u;
    nc += fc[k];
    no += fo[k];
  }
  uint32_t tc, to;
  uint32_t pc = ee_scan_add(s_scan, nc, t, &tc);
  uint32_t po = ee_scan_add(s_scan, no, t, &to);
  for (int k = 0; k < 8; k++) {
    if (fc[k]) sel_c[pc++] = t * 8 + k;
    if (fo[k]) sel_o[po++] = t * 8 + k;
  }
  if (t == 0) {
    st->n_cand = tc;
    st->n_alive = to;
    st->n_drop_first = 0;
    st->chunks = 0;
    st->rng[0][0] = b1;
    st->rng[0][1] = n;
    st->pad[0] = mx;  // the largest count of pass A (k_ee_plan projects it when there was no pass B)
    st->work = (unsigned long long)H * b1 + (unsigned long long)tc * (n - b1);
  }
}

// after pass B: the observation ranges of pass C from the best count known (see the header comment)
__global__ __launch_bounds__(1024) void k_ee_plan(const uint32_t *__restrict__ votes, const uint32_t *__restrict__ sel_c,
                                                  EeState *__restrict__ st, uint32_t best_before, uint32_t b1, uint32_t n,
                                                  uint32_t align) {
  __shared__ uint32_t s_red[16];
  const int t = threadIdx.x;
  uint32_t mx = best_before;
  const uint32_t nc = st->n_cand;
  for (uint32_t j = t; j < nc; j += 1024) {
    const uint32_t v = votes[sel_c[j]];
    mx = v > mx ? v : mx;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t a = __shfl_down(mx, o);
    mx = a > mx ? a : mx;
  }
  if ((t & 63) == 0) s_red[t >> 6] = mx;
  __syncthreads();
  if (t == 0) {
    uint32_t Lg = 0;
    for (int w = 0; w < 16; w++) Lg = s_red[w] > Lg ? s_red[w] : Lg;
    auto up = [&](unsigned long long v) -> uint32_t {
      v = (v + align - 1) / align * align;
      return (uint32_t)(v < n ? v : n);
    };
    // R = n - b <= Lg - (a few votes): b* = n - Lg + n / 64; nothing known (Lg = 0): one chunk to the end
    uint32_t bs = Lg ? up((unsigned long long)(n - (Lg < n ? Lg : n)) + n / 64) : n;
    if (nc == 0 && st->pad[0] > 0) {
      // no pass B: the best partial count grows like f b (f = pad[0] / b1, taken 10 % low), and a hypothesis without
      // votes goes when n - b <= f b:  b* = n / (1 + 0.9 f) + n / 64
      const unsigned long long den = (unsigned long long)b1 * 10 + (unsigned long long)st->pad[0] * 9;
      const uint32_t bp = up((unsigned long long)n * b1 * 10 / den + n / 64);
      bs = bp < bs ? bp : bs;
    }
    if (bs < b1) bs = b1;
    const uint32_t mid = up(((unsigned long long)bs + n) / 2);
    st->rng[1][0] = b1, st->rng[1][1] = bs;
    st->rng[2][0] = bs, st->rng[2][1] = mid;
    st->rng[3][0] = mid, st->rng[3][1] = n;
    st->work += (unsigned long long)st->n_alive * (bs - b1);
  }
}

// after chunk k (1 or 2) of pass C: who stays alive (R = n - end of that chunk observations remain)
__global__ __launch_bounds__(1024) void k_ee_select(const uint32_t *__restrict__ votes, const uint8_t *__restrict__ valid,
                                                    uint32_t H, uint32_t n, int k, uint32_t best_before,
                                                    const uint32_t *__restrict__ sel_in, uint32_t *__restrict__ sel_out,
                                                    EeState *__restrict__ st) {
  __shared__ uint32_t s_scan[1024], s_alive[kSelCap / 32];  // alive bitmap of up to kSelCap hypotheses
  const int t = threadIdx.x;
  if (H > kSelCap) return;
  const uint32_t n_in = st->n_alive;
  const uint32_t R = n - st->rng[k][1];
  if (t < (int)(kSelCap / 32)) s_alive[t] = 0;
  __syncthreads();
  for (uint32_t j = t; j < n_in; j += 1024) {
    const uint32_t h = sel_in[j];
    atomicOr(&s_alive[h >> 5], 1u << (h & 31));
  }
  __syncthreads();
  uint32_t v[8], lm = 0, pre[8];
  for (int q = 0; q < 8; q++) {
    const uint32_t h = t * 8 + q;
    v[q] = (h < H && valid[h]) ? votes[h] : 0u;
    pre[q] = lm;  // maximum over this thread's earlier entries
    lm = v[q] > lm ? v[q] : lm;
  }
  const uint32_t before = ee_scan_max(s_scan, lm, t);
  uint32_t f[8], cnt = 0;
  for (int q = 0; q < 8; q++) {
    const uint32_t h = t * 8 + q;
    uint32_t L = before > pre[q] ? before : pre[q];
    L = best_before > L ? best_before : L;
    const bool alive = h < H && ((s_alive[h >> 5] >> (h & 31)) & 1u);
    f[q] = (alive && (unsigned long long)v[q] + R > L) ? 1u : 0u;
    cnt += f[q];
  }
  uint32_t total;
  uint32_t pos = ee_scan_add(s_scan, cnt, t, &total);
  for (int q = 0; q < 8; q++)
    if (f[q]) sel_out[pos++] = t * 8 + q;
  if (t == 0) {
    if (st->chunks == 0) st->n_drop_first = n_in - total;
    st->n_alive = total;
    st->chunks += 1;
    st->work += (unsigned long long)total * (st->rng[k + 1][1] - st->rng[k + 1][0]);
  }
}

// rows of a selection -> compact fp32 rows (rows past the selection up to `cap` are zero-filled)
__global__ __launch_bounds__(256) void k_ee_gather_f32(const uint32_t *__restrict__ sel, const uint32_t *__restrict__ n_sel,
                                                       uint32_t cap, const float *__restrict__ src, int row,
                                                       float *__restrict__ dst, const float *__restrict__ src2, int row2,
                                                       float *__restrict__ dst2, float fill2) {
  const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= cap) return;
  const bool live = j < *n_sel;
  const uint32_t h = live ? sel[j] : 0;
  for (int k = lane; k < row; k += 64) dst[(size_t)j * row + k] = live ? src[(size_t)h * row + k] : 0.0f;
  for (int k = lane; k < row2; k += 64) dst2[(size_t)j * row2 + k] = live ? src2[(size_t)h * row2 + k] : fill2;
}

}  // namespace lsqr
