// Graph ingestion on the device: edge lists -> the hot path's input formats.
//
//   csr_sym_normalized   D^-1/2 (A + I) D^-1/2 as CSR                  RAGraph_node/ragraph_utils/utility.py:19-26 (normalize_adj),
//                                                                       :45-66 (coo_matrix of ones -> dense: duplicate edges SUM;
//                                                                       + eye; torch.FloatTensor)
//   binorm_edges         bipartite D^-1/2 A D^-1/2 as a sorted edge list RAGraph_edge/modules/base_model.py:34-52 (_make_binorm_adj:
//                                                                       binarised, symmetric, .tocoo() order), edge times of
//                                                                       utils/dataloader.py:94,108-113 (the LAST time of a
//                                                                       repeated (user, item) pair wins: dict assignment)
//
// The reference builds both on the host (a numpy row_stack loop that densifies the block-diagonal adjacency -- quadratic
// in the batch's node count and the reason a 100k-node graph cannot be ingested at all -- and Python loops over every
// edge).  Here: 64-bit keys, the library's own stable radix sort and prefix sums (csrc/sortscan.hip), run-length heads, a prefix
// sum, and kernels that count multiplicities, accumulate integer degrees (exact whatever the atomic order) and emit the
// normalised values in double precision cast to fp32 last -- as scipy (float64) followed by torch.FloatTensor does.
#include "sortscan.h"

namespace ragraph {

__global__ void __launch_bounds__(256) sym_keys_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                                                       int64_t E, int64_t n, uint64_t* __restrict__ keys) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < E) keys[e] = (uint64_t)col[e] * (uint64_t)n + (uint64_t)row[e];  // normalize_adj returns (A D)^T D: transposed pattern
  else if (e < E + n) keys[e] = (uint64_t)(e - E) * (uint64_t)n + (uint64_t)(e - E);  // + sp.eye(n)
}

// head[s] = 1 where a run of equal keys starts (LAST = 0) or ends (LAST = 1)
template <bool LAST>
__global__ void __launch_bounds__(256) run_flags_kernel(const uint64_t* __restrict__ keys, int64_t M, int* __restrict__ flag) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= M) return;
  if (LAST) flag[s] = (s == M - 1 || keys[s] != keys[s + 1]) ? 1 : 0;
  else flag[s] = (s == 0 || keys[s] != keys[s - 1]) ? 1 : 0;
}

// pos[u] = sorted position of the u-th flagged element; count = number of flagged elements; pos[count] = M
__global__ void __launch_bounds__(256) run_positions_kernel(const int* __restrict__ flag, const int* __restrict__ slot, int64_t M,
                                                            int64_t* __restrict__ pos, int64_t* __restrict__ count) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= M) return;
  if (flag[s]) pos[slot[s]] = s;
  if (s == M - 1) {
    const int64_t c = (int64_t)slot[s] + flag[s];
    *count = c;
    pos[c] = M;
  }
}

// unique (out_row = original col, out_col = original row) entries with multiplicities; integer degrees of the ORIGINAL rows
__global__ void __launch_bounds__(256) sym_entries_kernel(const uint64_t* __restrict__ keys, const int64_t* __restrict__ pos,
                                                          const int64_t* __restrict__ nnz, int64_t n,
                                                          int64_t* __restrict__ rowptr, int32_t* __restrict__ out_col,
                                                          unsigned* __restrict__ cnt, unsigned* __restrict__ deg) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nn = *nnz;
  if (u == 0) rowptr[n] = nn;
  if (u >= nn) return;
  const uint64_t key = keys[pos[u]];
  const int64_t orow = (int64_t)(key / (uint64_t)n), r = (int64_t)(key % (uint64_t)n);
  const unsigned c = (unsigned)(pos[u + 1] - pos[u]);
  out_col[u] = (int32_t)r;
  cnt[u] = c;
  atomicAdd(deg + r, c);  // row sum of A + I over the original row r (integers: exact in any order)
  if (u == 0 || (int64_t)(keys[pos[u - 1]] / (uint64_t)n) != orow) rowptr[orow] = u;  // every row holds its self loop
}

__global__ void __launch_bounds__(256) sym_values_kernel(const uint64_t* __restrict__ keys, const int64_t* __restrict__ pos,
                                                         const int64_t* __restrict__ nnz, int64_t n,
                                                         const unsigned* __restrict__ cnt, const unsigned* __restrict__ deg,
                                                         float* __restrict__ val) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= *nnz) return;
  const uint64_t key = keys[pos[u]];
  const int64_t c = (int64_t)(key / (uint64_t)n), r = (int64_t)(key % (uint64_t)n);
  const double dr = deg[r] ? pow((double)deg[r], -0.5) : 0.0, dc = deg[c] ? pow((double)deg[c], -0.5) : 0.0;
  val[u] = (float)(((double)cnt[u] * dr) * dc);  // (A[r][c] * d_r) * d_c in float64, cast last
}

// ---- bipartite ----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pair_keys_kernel(const int64_t* __restrict__ u, const int64_t* __restrict__ it, int64_t E,
                                                        int64_t num_items, uint64_t* __restrict__ keys, int64_t* __restrict__ idx) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  keys[e] = (uint64_t)u[e] * (uint64_t)num_items + (uint64_t)it[e];
  idx[e] = e;
}

// unique (user, item) pairs: the time of the LAST occurrence; both directed edges' keys (dst * n + src) and degrees
__global__ void __launch_bounds__(256) pair_edges_kernel(const uint64_t* __restrict__ keys, const int64_t* __restrict__ idx,
                                                         const int64_t* __restrict__ pos /* last positions */,
                                                         const int64_t* __restrict__ npairs, const int64_t* __restrict__ step,
                                                         int64_t num_users, int64_t num_items, uint64_t* __restrict__ ekeys,
                                                         int64_t* __restrict__ etime, unsigned* __restrict__ deg) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t np_ = *npairs;
  if (p >= np_) return;
  const int64_t s = pos[p];
  const uint64_t key = keys[s];
  const int64_t uu = (int64_t)(key / (uint64_t)num_items), ii = (int64_t)(key % (uint64_t)num_items) + num_users;
  const uint64_t n = (uint64_t)(num_users + num_items);
  const int64_t t = step[idx[s]];
  ekeys[p] = (uint64_t)ii * n + (uint64_t)uu;        // edge uu -> ii, ordered by (dst, src)
  ekeys[np_ + p] = (uint64_t)uu * n + (uint64_t)ii;  // edge ii -> uu
  etime[p] = t;
  etime[np_ + p] = t;
  atomicAdd(deg + uu, 1u);
  atomicAdd(deg + ii, 1u);
}

__global__ void __launch_bounds__(256) pair_emit_kernel(const uint64_t* __restrict__ ekeys_sorted, const int64_t* __restrict__ etime_sorted,
                                                        const int64_t* __restrict__ npairs, int64_t n,
                                                        const unsigned* __restrict__ deg, int64_t* __restrict__ edges,
                                                        float* __restrict__ norm, int64_t* __restrict__ times,
                                                        int64_t* __restrict__ nedges) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t ne = 2 * *npairs;
  if (e == 0) *nedges = ne;
  if (e >= ne) return;
  const uint64_t key = ekeys_sorted[e];
  const int64_t dst = (int64_t)(key / (uint64_t)n), src = (int64_t)(key % (uint64_t)n);
  edges[2 * e] = src;
  edges[2 * e + 1] = dst;
  const double ds = deg[src] ? pow((double)deg[src], -0.5) : 0.0, dd = deg[dst] ? pow((double)deg[dst], -0.5) : 0.0;
  norm[e] = (float)(ds * dd);  // mat.data.astype(np.float32) of the float64 product
  times[e] = etime_sorted[e];
}

// ---- COO -> CSR (stable) ------------------------------------------------------------------------------------------
// key = row (sort_cols = 0: the given edge order survives inside a row -- the order scatter_add_ accumulates in,
// RAGraph_edge/modules/utils.py:17-32) or row * n + col (columns ascending inside a row); the value is the edge's position.
__global__ void __launch_bounds__(256) coo_keys_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t E,
                                                       int64_t n, int sort_cols, uint64_t* __restrict__ keys,
                                                       int64_t* __restrict__ idx) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  keys[e] = sort_cols ? (uint64_t)row[e] * (uint64_t)n + (uint64_t)col[e] : (uint64_t)row[e];
  idx[e] = e;
}

// rowptr[r] = the first sorted slot whose row is >= r: slot s writes the rows in (row(s - 1), row(s)] -- its own row and
// the empty rows before it --, the last slot also the rows behind it.  O(E + n) stores, no histogram and no prefix sum.
// out_col (optional): the columns in CSR order.
__global__ void __launch_bounds__(256) coo_rowptr_kernel(const uint64_t* __restrict__ keys, const int64_t* __restrict__ perm,
                                                         const int64_t* __restrict__ col, int64_t E, int64_t n, int sort_cols,
                                                         int64_t* __restrict__ rowptr, int32_t* __restrict__ out_col) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= E) return;
  const uint64_t un = (uint64_t)n;
  const int64_t r = (int64_t)(sort_cols ? keys[s] / un : keys[s]);
  const int64_t rp = s == 0 ? -1 : (int64_t)(sort_cols ? keys[s - 1] / un : keys[s - 1]);
  for (int64_t x = rp + 1; x <= r; ++x) rowptr[x] = s;
  if (s == E - 1)
    for (int64_t x = r + 1; x <= n; ++x) rowptr[x] = E;
  if (out_col) out_col[s] = (int32_t)(sort_cols ? (int64_t)(keys[s] % un) : col[perm[s]]);
}

__global__ void __launch_bounds__(256) fill_i64_kernel(int64_t* __restrict__ p, int64_t n, int64_t v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

// rows[e] = the row whose range [rowptr[r], rowptr[r + 1]) holds e (binary search: the last r with rowptr[r] <= e)
__global__ void __launch_bounds__(256) csr_row_ids_kernel(const int64_t* __restrict__ rowptr, int64_t n, int64_t nnz,
                                                          int64_t* __restrict__ rows) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nnz) return;
  int64_t lo = 0, hi = n;   // invariant: rowptr[lo] <= e < rowptr[hi]
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] <= e) lo = mid;
    else hi = mid;
  }
  rows[e] = lo;
}

// ---- positions of the set elements of a byte mask (boolean indexing without another library's select) --------------
__global__ void __launch_bounds__(256) mask_flags_kernel(const unsigned char* __restrict__ mask, int64_t E, int* __restrict__ flag) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < E) flag[e] = mask[e] ? 1 : 0;
}

__global__ void __launch_bounds__(256) mask_positions_kernel(const unsigned char* __restrict__ mask, const int* __restrict__ slot,
                                                             int64_t E, int64_t* __restrict__ pos, int64_t* __restrict__ count) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  if (mask[e]) pos[slot[e]] = e;
  if (e == E - 1) *count = (int64_t)slot[e] + (mask[e] ? 1 : 0);
}

static int key_bits(uint64_t max_key) {
  int b = 1;
  while (b < 64 && (max_key >> b) != 0) ++b;
  return b;
}

struct IngestWs {
  uint64_t *keys_a, *keys_b;
  int64_t *vals_a, *vals_b, *pos, *count;
  int *flag, *slot;
  unsigned *cnt, *deg;
  void* temp;
  size_t temp_bytes;
};

static size_t ingest_carve(char* w, int64_t M, int64_t n, IngestWs* out) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = w ? w + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  IngestWs f;
  f.keys_a = reinterpret_cast<uint64_t*>(take((size_t)M * 8));
  f.keys_b = reinterpret_cast<uint64_t*>(take((size_t)M * 8));
  f.vals_a = reinterpret_cast<int64_t*>(take((size_t)M * 8));
  f.vals_b = reinterpret_cast<int64_t*>(take((size_t)M * 8));
  f.pos = reinterpret_cast<int64_t*>(take((size_t)(M + 1) * 8));
  f.count = reinterpret_cast<int64_t*>(take(8));
  f.flag = reinterpret_cast<int*>(take((size_t)M * 4));
  f.slot = reinterpret_cast<int*>(take((size_t)M * 4));
  f.cnt = reinterpret_cast<unsigned*>(take((size_t)M * 4));
  f.deg = reinterpret_cast<unsigned*>(take((size_t)n * 4));
  const size_t t1 = radix_sort_temp_bytes(M, 8), t3 = scan_temp_bytes(M);
  f.temp_bytes = t1 > t3 ? t1 : t3;
  f.temp = take(f.temp_bytes + 256);
  if (out) *out = f;
  return off;
}

}  // namespace ragraph

using namespace ragraph;

extern "C" size_t ragraph_ingest_workspace_bytes(int64_t max_keys, int64_t n) {
  if (max_keys < 1 || max_keys >= (int64_t)INT_MAX || n < 1) return 0;
  return ingest_carve(nullptr, max_keys, n, nullptr);
}

#define RG_HIPCUB(call, what)                                                  \
  do {                                                                         \
    hipError_t e__ = (call);                                                   \
    if (e__ != hipSuccess) {                                                   \
      set_error("%s: %s", (what), hipGetErrorString(e__));                     \
      return RAGRAPH_EDEVICE;                                                  \
    }                                                                          \
  } while (0)

extern "C" int ragraph_csr_sym_normalized_f32(const int64_t* row, const int64_t* col, int64_t E, int64_t n, int64_t* rowptr,
                                              int32_t* out_col, float* out_val, int64_t* nnz, void* ws, size_t ws_bytes,
                                              void* stream) {
  RG_REQUIRE((row && col) || E == 0, RAGRAPH_EINVAL, "csr_sym_normalized: null edge list");
  RG_REQUIRE(rowptr && out_col && out_val && nnz && ws, RAGRAPH_EINVAL, "csr_sym_normalized: null pointer");
  RG_REQUIRE(E >= 0 && n >= 1 && n < ((int64_t)1 << 31), RAGRAPH_EINVAL, "csr_sym_normalized: bad E/n");
  const int64_t M = E + n;
  RG_REQUIRE(M < (int64_t)INT_MAX, RAGRAPH_EUNSUPPORTED, "csr_sym_normalized: E + n must fit int32");
  RG_REQUIRE(ws_bytes >= ragraph_ingest_workspace_bytes(M, n), RAGRAPH_EWORKSPACE, "csr_sym_normalized: workspace too small");
  hipStream_t st = as_stream(stream);
  IngestWs f;
  ingest_carve(static_cast<char*>(ws), M, n, &f);
  const unsigned g = (unsigned)cdiv(M, 256);
  RG_HIPCUB(hipMemsetAsync(f.deg, 0, (size_t)n * 4, st), "csr_sym_normalized(memset)");
  hipLaunchKernelGGL(sym_keys_kernel, dim3(g), dim3(256), 0, st, row, col, E, n, f.keys_a);
  int rc = radix_sort_u64(f.keys_a, f.keys_b, nullptr, nullptr, 0, M, key_bits((uint64_t)n * (uint64_t)n), f.temp, f.temp_bytes, st);
  if (rc != RAGRAPH_OK) return rc;
  hipLaunchKernelGGL(run_flags_kernel<false>, dim3(g), dim3(256), 0, st, f.keys_b, M, f.flag);
  rc = scan_sum_i32(f.flag, f.slot, M, false, f.temp, f.temp_bytes, st);
  if (rc != RAGRAPH_OK) return rc;
  hipLaunchKernelGGL(run_positions_kernel, dim3(g), dim3(256), 0, st, f.flag, f.slot, M, f.pos, nnz);
  hipLaunchKernelGGL(sym_entries_kernel, dim3(g), dim3(256), 0, st, f.keys_b, f.pos, nnz, n, rowptr, out_col, f.cnt, f.deg);
  hipLaunchKernelGGL(sym_values_kernel, dim3(g), dim3(256), 0, st, f.keys_b, f.pos, nnz, n, f.cnt, f.deg, out_val);
  RG_CHECK_LAUNCH("csr_sym_normalized");
  return RAGRAPH_OK;
}

extern "C" int ragraph_binorm_edges_f32(const int64_t* users, const int64_t* items, const int64_t* step, int64_t E,
                                        int64_t num_users, int64_t num_items, int64_t* edges, float* norm, int64_t* times,
                                        int64_t* nedges, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(users && items && step && edges && norm && times && nedges && ws, RAGRAPH_EINVAL, "binorm_edges: null pointer");
  RG_REQUIRE(E >= 1 && num_users >= 1 && num_items >= 1, RAGRAPH_EINVAL, "binorm_edges: bad sizes");
  const int64_t n = num_users + num_items, M = 2 * E;
  RG_REQUIRE(M < (int64_t)INT_MAX && n < ((int64_t)1 << 31), RAGRAPH_EUNSUPPORTED, "binorm_edges: 2E and n must fit int32");
  RG_REQUIRE(ws_bytes >= ragraph_ingest_workspace_bytes(M, n), RAGRAPH_EWORKSPACE, "binorm_edges: workspace too small");
  hipStream_t st = as_stream(stream);
  IngestWs f;
  ingest_carve(static_cast<char*>(ws), M, n, &f);
  const unsigned gE = (unsigned)cdiv(E, 256), gM = (unsigned)cdiv(M, 256);
  RG_HIPCUB(hipMemsetAsync(f.deg, 0, (size_t)n * 4, st), "binorm_edges(memset)");
  hipLaunchKernelGGL(pair_keys_kernel, dim3(gE), dim3(256), 0, st, users, items, E, num_items, f.keys_a, f.vals_a);
  // (the radix sort is stable: among equal (user, item) pairs the original order survives, so a run's last element is the
  // pair's last occurrence)
  int rc = radix_sort_u64(f.keys_a, f.keys_b, f.vals_a, f.vals_b, 8, E, key_bits((uint64_t)num_users * (uint64_t)num_items), f.temp,
                          f.temp_bytes, st);
  if (rc != RAGRAPH_OK) return rc;
  hipLaunchKernelGGL(run_flags_kernel<true>, dim3(gE), dim3(256), 0, st, f.keys_b, E, f.flag);
  rc = scan_sum_i32(f.flag, f.slot, E, false, f.temp, f.temp_bytes, st);
  if (rc != RAGRAPH_OK) return rc;
  hipLaunchKernelGGL(run_positions_kernel, dim3(gE), dim3(256), 0, st, f.flag, f.slot, E, f.pos, f.count);
  // directed edges keyed by (dst, src); keys_a / vals_a are free again
  RG_HIPCUB(hipMemsetAsync(f.keys_a, 0xFF, (size_t)M * 8, st), "binorm_edges(memset keys)");  // unused tail sorts last
  hipLaunchKernelGGL(pair_edges_kernel, dim3(gE), dim3(256), 0, st, f.keys_b, f.vals_b, f.pos, f.count, step, num_users, num_items,
                     f.keys_a, f.vals_a, f.deg);
  // (vals_a beyond 2 * npairs is garbage paired with 0xFF.. keys: sorted behind every real edge)
  rc = radix_sort_u64(f.keys_a, f.keys_b, f.vals_a, f.vals_b, 8, M, 64, f.temp, f.temp_bytes, st);
  if (rc != RAGRAPH_OK) return rc;
  hipLaunchKernelGGL(pair_emit_kernel, dim3(gM), dim3(256), 0, st, f.keys_b, f.vals_b, f.count, n, f.deg, edges, norm, times,
                     nedges);
  RG_CHECK_LAUNCH("binorm_edges");
  return RAGRAPH_OK;
}

// COO -> CSR, stable (what torch.sort(stable) + bincount + cumsum did on the per-step paths until round 5: the edge flavour's
// destination-sorted CSR of every re-drawn edge set, modules/RAGraph.py:337-343 + utils.py:40-53; the transposed pattern of a
// training SpMM; the hit lists of a gather's backward).  perm[s] = the input position of CSR slot s.
extern "C" size_t ragraph_coo_to_csr_workspace_bytes(int64_t E, int64_t n) {
  if (E < 0 || E >= (int64_t)INT_MAX || n < 1) return 0;
  return ingest_carve(nullptr, E > 0 ? E : 1, 1, nullptr);
}

extern "C" int ragraph_coo_to_csr_i64(const int64_t* row, const int64_t* col, int64_t E, int64_t n, int sort_cols, int64_t* rowptr,
                                      int64_t* perm, int32_t* out_col, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(rowptr && ws && (E == 0 || (row && col && perm)), RAGRAPH_EINVAL, "coo_to_csr: null pointer");
  RG_REQUIRE(E >= 0 && E < (int64_t)INT_MAX && n >= 1 && n < ((int64_t)1 << 31), RAGRAPH_EINVAL, "coo_to_csr: bad E/n");
  RG_REQUIRE(ws_bytes >= ragraph_coo_to_csr_workspace_bytes(E, n), RAGRAPH_EWORKSPACE, "coo_to_csr: workspace too small");
  hipStream_t st = as_stream(stream);
  if (E == 0) {
    hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, st, rowptr, n + 1, (int64_t)0);
    RG_CHECK_LAUNCH("coo_to_csr");
    return RAGRAPH_OK;
  }
  IngestWs f;
  ingest_carve(static_cast<char*>(ws), E, 1, &f);
  const unsigned g = (unsigned)cdiv(E, 256);
  hipLaunchKernelGGL(coo_keys_kernel, dim3(g), dim3(256), 0, st, row, col, E, n, sort_cols, f.keys_a, f.vals_a);
  const uint64_t max_key = sort_cols ? (uint64_t)n * (uint64_t)n - 1 : (uint64_t)(n - 1);
  const int rc = radix_sort_u64(f.keys_a, f.keys_b, f.vals_a, perm, 8, E, key_bits(max_key), f.temp, f.temp_bytes, st);
  if (rc != RAGRAPH_OK) return rc;
  hipLaunchKernelGGL(coo_rowptr_kernel, dim3(g), dim3(256), 0, st, f.keys_b, perm, col, E, n, sort_cols, rowptr, out_col);
  RG_CHECK_LAUNCH("coo_to_csr");
  return RAGRAPH_OK;
}

extern "C" int ragraph_csr_row_ids_i64(const int64_t* rowptr, int64_t n, int64_t nnz, int64_t* rows, void* stream) {
  RG_REQUIRE(rowptr && (rows || nnz == 0), RAGRAPH_EINVAL, "csr_row_ids: null pointer");
  RG_REQUIRE(n >= 1 && nnz >= 0, RAGRAPH_EINVAL, "csr_row_ids: bad n/nnz");
  if (nnz == 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(csr_row_ids_kernel, dim3((unsigned)cdiv(nnz, 256)), dim3(256), 0, as_stream(stream), rowptr, n, nnz, rows);
  RG_CHECK_LAUNCH("csr_row_ids");
  return RAGRAPH_OK;
}

// pos[j] = the position of the j-th non-zero byte of mask, ascending; *count = how many (the edge dropout of a training step,
// modules/utils.py:40-53: edges[mask] through the library's own prefix sums).  ws: ragraph_mask_positions_workspace_bytes(E).
extern "C" size_t ragraph_mask_positions_workspace_bytes(int64_t E) {
  if (E < 0 || E >= (int64_t)INT_MAX) return 0;
  const int64_t M = E > 0 ? E : 1;
  return 2 * align_up((size_t)M * 4, 256) + scan_temp_bytes(M) + 256;
}

extern "C" int ragraph_mask_positions_i64(const unsigned char* mask, int64_t E, int64_t* pos, int64_t* count, void* ws,
                                          size_t ws_bytes, void* stream) {
  RG_REQUIRE(count && ws && (E == 0 || (mask && pos)), RAGRAPH_EINVAL, "mask_positions: null pointer");
  RG_REQUIRE(E >= 0 && E < (int64_t)INT_MAX, RAGRAPH_EINVAL, "mask_positions: bad E");
  RG_REQUIRE(ws_bytes >= ragraph_mask_positions_workspace_bytes(E), RAGRAPH_EWORKSPACE, "mask_positions: workspace too small");
  hipStream_t st = as_stream(stream);
  if (E == 0) {
    hipLaunchKernelGGL(fill_i64_kernel, dim3(1), dim3(256), 0, st, count, (int64_t)1, (int64_t)0);
    RG_CHECK_LAUNCH("mask_positions");
    return RAGRAPH_OK;
  }
  char* w = static_cast<char*>(ws);
  int* flag = reinterpret_cast<int*>(w);
  int* slot = reinterpret_cast<int*>(w + align_up((size_t)E * 4, 256));
  void* temp = w + 2 * align_up((size_t)E * 4, 256);
  const unsigned g = (unsigned)cdiv(E, 256);
  hipLaunchKernelGGL(mask_flags_kernel, dim3(g), dim3(256), 0, st, mask, E, flag);
  const int rc = scan_sum_i32(flag, slot, E, false, temp, scan_temp_bytes(E) + 256, st);
  if (rc != RAGRAPH_OK) return rc;
  hipLaunchKernelGGL(mask_positions_kernel, dim3(g), dim3(256), 0, st, mask, slot, E, pos, count);
  RG_CHECK_LAUNCH("mask_positions");
  return RAGRAPH_OK;
}
