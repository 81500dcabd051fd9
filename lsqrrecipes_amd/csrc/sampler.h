// sampler.h -- counter-based minimal-subset sampler.  Replaces RANSAC.hxx:51-68 (k draws with
// libc rand(), each followed by an O(N) scan for "the selectedIndex-th datum not yet chosen")
// by the same selection rule evaluated in O(k^2): draw l of hypothesis h takes
//   u = SplitMix64-finalise(seed + GOLDEN * (64 h + l + 1)),  rank = floor(u (N-l) / 2^64)
// and picks the rank-th index not chosen by draws 0..l-1.  Draw order is preserved (RANSAC.hxx:65:
// the first drawn datum becomes the plane's point / the sphere's radius anchor).  Stateless in h,
// so any rank / GPU can generate any slice of the hypothesis stream.
#pragma once
#include <stdint.h>

#include "small_linalg.h"

namespace lsqr {

LSQR_HD uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

LSQR_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

// idx[0..k) in draw order; sorted[0..k) ascending scratch (k <= 64)
LSQR_HD void ctr_subset(uint64_t seed, uint64_t h, uint64_t n, int k, uint32_t *idx,
                        uint32_t *sorted) {
  for (int l = 0; l < k; l++) {
    uint64_t u = mix64(seed + 0x9E3779B97F4A7C15ULL * (h * 64ULL + (uint64_t)l + 1ULL));
    uint32_t v = (uint32_t)mulhi64(u, n - (uint64_t)l);
    int i = 0;
    while (i < l && sorted[i] <= v) {
      v++;
      i++;
    }
    for (int j = l; j > i; j--) sorted[j] = sorted[j - 1];
    sorted[i] = v;
    idx[l] = v;
  }
}

}  // namespace lsqr
