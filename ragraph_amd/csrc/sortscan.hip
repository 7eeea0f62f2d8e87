// Prefix sums and a stable LSD radix sort on the device (see sortscan.h).  Build-time / ingestion-time work: written for
// clarity and determinism (no atomics in anything that decides an output position), not for the last GB/s.
//
//   scan   tiles of 1024 ints per workgroup: tile sums -> (recursively) their exclusive prefix -> per-tile scan + offset.
//   sort   per pass (8 bits): (1) per-tile digit histograms, stored digit-major [256][tiles]; (2) their exclusive prefix
//          = where the keys of (digit, tile) start in the output; (3) scatter: a tile's keys are ranked STABLY inside the
//          tile -- the tile is cut into 32 segments of 64 keys (one wave each, in key order), a lane's rank among the lanes
//          of its wave that hold the same digit comes from eight ballots, the segments' per-digit counts are prefix-summed
//          by one thread per digit -- and written to start(digit, tile) + rank.  Equal keys keep their order.
#include "sortscan.h"

namespace ragraph {

constexpr int SCAN_TILE = 1024;

__global__ void __launch_bounds__(256) scan_tile_sums_kernel(const int* __restrict__ in, int64_t n, int* __restrict__ sums) {
  __shared__ int wsum[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  int s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += base + j < n ? in[base + j] : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// out = scan of the tile + offs[tile] (offs == nullptr: a single tile)
__global__ void __launch_bounds__(256) scan_tiles_kernel(const int* __restrict__ in, int64_t n, const int* __restrict__ offs,
                                                         int inclusive, int* __restrict__ out) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  int v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = base + j < n ? in[base + j] : 0;
  const int mine = v[0] + v[1] + v[2] + v[3];
  int incl = mine;  // inclusive scan of the threads' totals inside the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(incl, off);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int run = (offs ? offs[blockIdx.x] : 0) + incl - mine;
  for (int ww = 0; ww < w; ++ww) run += wsum[ww];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (base + j < n) out[base + j] = inclusive ? run + v[j] : run;
    run += v[j];
  }
}

static size_t scan_level_sizes(int64_t n, int64_t* levels, int max_levels) {
  // numbers of tiles per level until one tile remains
  int nl = 0;
  int64_t m = n;
  while (m > SCAN_TILE && nl < max_levels) {
    m = cdiv(m, (int64_t)SCAN_TILE);
    levels[nl++] = m;
  }
  return (size_t)nl;
}

size_t scan_temp_bytes(int64_t n) {
  int64_t lv[8];
  const size_t nl = scan_level_sizes(n, lv, 8);
  size_t bytes = 256;
  for (size_t i = 0; i < nl; ++i) bytes += align_up((size_t)lv[i] * sizeof(int), 256);
  return bytes;
}

int scan_sum_i32(const int* in, int* out, int64_t n, bool inclusive, void* temp, size_t temp_bytes, hipStream_t st) {
  if (n <= 0) return RAGRAPH_OK;
  RG_REQUIRE(temp_bytes >= scan_temp_bytes(n), RAGRAPH_EWORKSPACE, "scan: temp %zu < %zu", temp_bytes, scan_temp_bytes(n));
  if (n <= SCAN_TILE) {
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(256), 0, st, in, n, (const int*)nullptr, inclusive ? 1 : 0, out);
    RG_CHECK_LAUNCH("scan");
    return RAGRAPH_OK;
  }
  const int64_t tiles = cdiv(n, (int64_t)SCAN_TILE);
  int* sums = static_cast<int*>(temp);
  char* rest = static_cast<char*>(temp) + align_up((size_t)tiles * sizeof(int), 256);
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)tiles), dim3(256), 0, st, in, n, sums);
  RG_CHECK_LAUNCH("scan(tile sums)");
  const int rc = scan_sum_i32(sums, sums, tiles, false, rest, temp_bytes - align_up((size_t)tiles * sizeof(int), 256), st);
  if (rc != RAGRAPH_OK) return rc;
  hipLaunchKernelGGL(scan_tiles_kernel, dim3((unsigned)tiles), dim3(256), 0, st, in, n, (const int*)sums, inclusive ? 1 : 0, out);
  RG_CHECK_LAUNCH("scan(tiles)");
  return RAGRAPH_OK;
}

// ---- radix sort -----------------------------------------------------------------------------------------------------
constexpr int RS_ITEMS = 8, RS_TILE = 256 * RS_ITEMS;  // keys per workgroup; key i of a tile: item i / 256, thread i % 256

__global__ void __launch_bounds__(256) rs_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift, unsigned dmask,
                                                      int64_t tiles, int* __restrict__ hist) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & dmask)], 1);  // (an integer count: exact whatever the order)
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * tiles + blockIdx.x] = h[threadIdx.x];
}

template <typename V>
__global__ void __launch_bounds__(256) rs_scatter_kernel(const uint64_t* __restrict__ kin, uint64_t* __restrict__ kout,
                                                         const V* __restrict__ vin, V* __restrict__ vout, int64_t n, int shift,
                                                         unsigned dmask, int64_t tiles, const int* __restrict__ start) {
  __shared__ int seg[RS_ITEMS * 4][256];  // per segment (item, wave) and digit: count, then the count of earlier segments
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < RS_ITEMS * 4 * 256; i += 256) (&seg[0][0])[i] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  uint64_t key[RS_ITEMS];
  int dig[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    const bool ok = i < n;
    key[it] = ok ? kin[i] : 0ull;
    const int d = (int)((key[it] >> shift) & dmask);
    dig[it] = ok ? d : -1;
    // the lanes of this wave that hold the same digit
    unsigned long long same = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long bal = __ballot((d >> b) & 1);
      same &= ((d >> b) & 1) ? bal : ~bal;
    }
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    rank[it] = __popcll(same & lt);
    if (ok && (same & lt) == 0ull) seg[it * 4 + w][d] = __popcll(same);  // (the first lane of each digit writes its count)
  }
  __syncthreads();
  {  // thread d: exclusive prefix of digit d's counts over the segments, in key order
    int run = 0;
#pragma unroll
    for (int s = 0; s < RS_ITEMS * 4; ++s) {
      const int c = seg[s][threadIdx.x];
      seg[s][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    if (dig[it] < 0) continue;
    const int64_t pos = (int64_t)start[(int64_t)dig[it] * tiles + blockIdx.x] + seg[it * 4 + w][dig[it]] + rank[it];
    kout[pos] = key[it];
    if constexpr (sizeof(V) > 1) vout[pos] = vin[base + it * 256 + threadIdx.x];
  }
}

size_t radix_sort_temp_bytes(int64_t n, int val_bytes) {
  if (n <= 0) return 256;
  const int64_t tiles = cdiv(n, (int64_t)RS_TILE);
  return align_up((size_t)n * 8, 256) + align_up((size_t)n * (size_t)(val_bytes > 0 ? val_bytes : 0), 256) +
         align_up((size_t)tiles * 256 * sizeof(int), 256) + scan_temp_bytes(tiles * 256) + 256;
}

int radix_sort_u64(const uint64_t* keys_in, uint64_t* keys_out, const void* vals_in, void* vals_out, int val_bytes, int64_t n,
                   int bits, void* temp, size_t temp_bytes, hipStream_t st) {
  RG_REQUIRE(val_bytes == 0 || val_bytes == 4 || val_bytes == 8, RAGRAPH_EINVAL, "radix sort: values of %d bytes", val_bytes);
  RG_REQUIRE(n >= 0 && n < (int64_t)INT_MAX && bits >= 1 && bits <= 64, RAGRAPH_EINVAL, "radix sort: n=%lld bits=%d", (long long)n, bits);
  if (n == 0) return RAGRAPH_OK;
  RG_REQUIRE(temp_bytes >= radix_sort_temp_bytes(n, val_bytes), RAGRAPH_EWORKSPACE, "radix sort: temp %zu < %zu", temp_bytes,
             radix_sort_temp_bytes(n, val_bytes));
  const int64_t tiles = cdiv(n, (int64_t)RS_TILE);
  char* t = static_cast<char*>(temp);
  uint64_t* ktmp = reinterpret_cast<uint64_t*>(t);
  t += align_up((size_t)n * 8, 256);
  char* vtmp = t;
  t += align_up((size_t)n * (size_t)val_bytes, 256);
  int* hist = reinterpret_cast<int*>(t);
  t += align_up((size_t)tiles * 256 * sizeof(int), 256);
  const size_t scan_bytes = scan_temp_bytes(tiles * 256);
  const int passes = (bits + 7) / 8;
  const uint64_t* ksrc = keys_in;
  const char* vsrc = static_cast<const char*>(vals_in);
  for (int p = 0; p < passes; ++p) {
    // the last pass writes the caller's output; the passes before it alternate between the temporaries and the output
    const bool to_out = ((passes - 1 - p) & 1) == 0;
    uint64_t* kdst = to_out ? keys_out : ktmp;
    char* vdst = to_out ? static_cast<char*>(vals_out) : vtmp;
    const unsigned dmask = bits - 8 * p >= 8 ? 0xFFu : ((1u << (bits - 8 * p)) - 1u);   // (the last pass may be narrower)
    hipLaunchKernelGGL(rs_hist_kernel, dim3((unsigned)tiles), dim3(256), 0, st, ksrc, n, 8 * p, dmask, tiles, hist);
    RG_CHECK_LAUNCH("radix sort(histogram)");
    const int rc = scan_sum_i32(hist, hist, tiles * 256, false, t, scan_bytes, st);
    if (rc != RAGRAPH_OK) return rc;
    if (val_bytes == 8)
      hipLaunchKernelGGL(rs_scatter_kernel<int64_t>, dim3((unsigned)tiles), dim3(256), 0, st, ksrc, kdst, reinterpret_cast<const int64_t*>(vsrc),
                         reinterpret_cast<int64_t*>(vdst), n, 8 * p, dmask, tiles, (const int*)hist);
    else if (val_bytes == 4)
      hipLaunchKernelGGL(rs_scatter_kernel<int32_t>, dim3((unsigned)tiles), dim3(256), 0, st, ksrc, kdst, reinterpret_cast<const int32_t*>(vsrc),
                         reinterpret_cast<int32_t*>(vdst), n, 8 * p, dmask, tiles, (const int*)hist);
    else
      hipLaunchKernelGGL(rs_scatter_kernel<char>, dim3((unsigned)tiles), dim3(256), 0, st, ksrc, kdst, (const char*)nullptr, (char*)nullptr, n,
                         8 * p, dmask, tiles, (const int*)hist);
    RG_CHECK_LAUNCH("radix sort(scatter)");
    ksrc = kdst;
    vsrc = vdst;
  }
  return RAGRAPH_OK;
}

}  // namespace ragraph

using namespace ragraph;

// Exported for the tests (tests/test_gpu_ingest.py): the two primitives against numpy's stable sort / cumsum.
extern "C" size_t ragraph_radix_sort_workspace_bytes(int64_t n, int val_bytes) { return radix_sort_temp_bytes(n, val_bytes); }
extern "C" int ragraph_radix_sort_u64(const uint64_t* keys_in, uint64_t* keys_out, const void* vals_in, void* vals_out, int val_bytes,
                                      int64_t n, int bits, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(keys_in && keys_out && ws && (val_bytes == 0 || (vals_in && vals_out)), RAGRAPH_EINVAL, "radix_sort: null pointer");
  return radix_sort_u64(keys_in, keys_out, vals_in, vals_out, val_bytes, n, bits, ws, ws_bytes, as_stream(stream));
}
extern "C" size_t ragraph_scan_workspace_bytes(int64_t n) { return scan_temp_bytes(n); }
extern "C" int ragraph_scan_sum_i32(const int* in, int* out, int64_t n, int inclusive, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(in && out && ws, RAGRAPH_EINVAL, "scan_sum: null pointer");
  return scan_sum_i32(in, out, n, inclusive != 0, ws, ws_bytes, as_stream(stream));
}
