// sort.h -- stable radix sort of (uint32 key, uint32 value) pairs by key bits [0, end_bit) on `stream`
// (sort.hip: rocPRIM).  Call with tmp == nullptr to obtain the temporary storage size in *tmp_bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace lsqr {
hipError_t sort_pairs_u32(void *tmp, size_t *tmp_bytes, const uint32_t *keys_in, uint32_t *keys_out,
                          const uint32_t *vals_in, uint32_t *vals_out, size_t n, unsigned end_bit,
                          hipStream_t stream);
}
