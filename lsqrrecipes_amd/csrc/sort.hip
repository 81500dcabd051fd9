// sort.hip -- the one library primitive of the index build: a stable LSD radix sort of (Morton key, record index)
// pairs (rocPRIM device radix sort).  Its own translation unit: rocPRIM's templates compile in parallel with
// lsqr_hip.hip and stay out of its way.  Everything around the sort -- bounds, keys, gather, cell boxes -- is in
// cells.h.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "sort.h"

namespace lsqr {

hipError_t sort_pairs_u32(void *tmp, size_t *tmp_bytes, const uint32_t *keys_in, uint32_t *keys_out,
                          const uint32_t *vals_in, uint32_t *vals_out, size_t n, unsigned end_bit,
                          hipStream_t stream) {
  return rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit,
                                   stream);
}

}  // namespace lsqr
