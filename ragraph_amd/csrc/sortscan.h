// Device prefix sums and a stable LSD radix sort (csrc/sortscan.hip): the integer bookkeeping of graph ingestion (edge list ->
// CSR, csrc/ingest.hip) and of the duplicate grouping of a bank (csrc/dedup.hip).  Own kernels: nothing on any path of the
// library is a call into another library.
#pragma once
#include "common.h"

namespace ragraph {

// out[i] = sum of in[0 .. i) (inclusive = false) or in[0 .. i] (true); int32 sums (callers bound their totals by INT_MAX).
// in == out allowed.  temp: scan_temp_bytes(n) bytes.
size_t scan_temp_bytes(int64_t n);
int scan_sum_i32(const int* in, int* out, int64_t n, bool inclusive, void* temp, size_t temp_bytes, hipStream_t st);

// Stable sort of 64-bit keys by their low `bits` bits (8 bits per pass, least significant first), with optional values of
// val_bytes = 0 / 4 / 8 bytes each.  The inputs are left as they are; the result lands in keys_out / vals_out.
// temp: radix_sort_temp_bytes(n, val_bytes) bytes.
size_t radix_sort_temp_bytes(int64_t n, int val_bytes);
int radix_sort_u64(const uint64_t* keys_in, uint64_t* keys_out, const void* vals_in, void* vals_out, int val_bytes, int64_t n,
                   int bits, void* temp, size_t temp_bytes, hipStream_t st);

}  // namespace ragraph
