// Exact-duplicate collapsing of a key bank, and the expansion of a top-k over the unique rows back to the bank's rows.
//
// The reference's own banks are mostly duplicates: ToyGraphBase._build_toy_graph_base appends 1 + num_augment_scale passes
// per resource graph (RAGraph_node/ragraph_utils/ToyGraphBase.py:91-119), the rows of a pass are drawn WITH replacement
// (:98 torch.multinomial(..., replacement=True)), and the augmented passes multiply the features by
// bernoulli(sample_prob * 0.01) (Augmentation.py:9-20) -- zero for practically every node -- so every row of those
// passes is the same vector, normalize(PReLU(bias)).  Three of four bank rows are bit-identical.  A filter that keeps
// "every key within eps of the k-th best" keeps whole groups of such rows (hundreds of thousands of candidates); scoring
// one representative per group and expanding the winners is the same result for a fraction of the work.
//
//   ragraph_dedup_rows_f32         groups the BIT-identical rows of Kn (64-bit row hash -> stable radix sort of (hash, row) (csrc/sortscan.hip)
//                                  -> neighbours compared bit by bit, so a hash collision only splits a group, never
//                                  merges two), numbers the groups by their lowest row (ascending) and lists every
//                                  group's rows in ascending order
//   ragraph_topk_expand_groups_f32 canonical top-k of the bank from the canonical top-min(k,U) of the unique rows: identical
//                                  rows have identical scores (the same fmaf chain on the same bits), so the bank's
//                                  canonical order (score descending, row ascending) walks the winners' groups in list
//                                  order, each group's rows ascending; groups whose scores TIE are merged by row.  Why
//                                  the top-k of the unique rows suffices: a group outside it scores no better than the
//                                  k-th listed group, and when it ties with listed groups those have lower
//                                  representatives -- at least as many rows below every row of the unlisted group as
//                                  the tie can still place.
#include "sortscan.h"

namespace ragraph {

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// One wave per row: lane l hashes the 32-bit words l, l + 64, ... of the row together with their positions; the row's
// hash is the (wrapping) sum of the word hashes, mixed once more.  Bit patterns, not values: -0 and +0 differ.
__global__ void __launch_bounds__(256) row_hash_kernel(const uint32_t* __restrict__ X, int64_t n, int D,
                                                       uint64_t* __restrict__ hash, int32_t* __restrict__ iota) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const uint32_t* row = X + r * (int64_t)D;
  uint64_t h = 0;
  for (int c = lane; c < D; c += 64)
    h += mix64((uint64_t)row[c] * 0x9E3779B97F4A7C15ull + (uint64_t)(c + 1) * 0xC2B2AE3D27D4EB4Full);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)h, off, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(h >> 32), off, 64);
    h += ((uint64_t)hi << 32) | lo;
  }
  if (lane == 0) {
    hash[r] = mix64(h);
    iota[r] = (int32_t)r;
  }
}

// flag[p] = 1 where sorted position p starts a group: another hash than its predecessor, or other bits under the same
// hash.  One wave per position (only equal hashes read rows).
__global__ void __launch_bounds__(256) dedup_flags_kernel(const uint32_t* __restrict__ X, int64_t n, int D,
                                                          const uint64_t* __restrict__ hs, const int32_t* __restrict__ perm,
                                                          int32_t* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  int head = 1;
  if (p > 0 && hs[p] == hs[p - 1]) {
    const uint32_t* a = X + (int64_t)perm[p] * D;
    const uint32_t* b = X + (int64_t)perm[p - 1] * D;
    int diff = 0;
    for (int c = lane; c < D; c += 64) diff |= (a[c] != b[c]);
    head = __any(diff) ? 1 : 0;
  }
  if (lane == 0) flag[p] = head;
}

// group g (sorted order) starts at gstart[g]; its lowest row (the stable sort keeps rows ascending inside a hash run) is
// grep[g]; isrep marks those rows in bank order
__global__ void __launch_bounds__(256) dedup_heads_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ gid1,
                                                          const int32_t* __restrict__ perm, int64_t n,
                                                          int32_t* __restrict__ gstart, int32_t* __restrict__ grep,
                                                          int32_t* __restrict__ isrep) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n || !flag[p]) return;
  const int g = gid1[p] - 1;
  gstart[g] = (int32_t)p;
  grep[g] = perm[p];
  isrep[perm[p]] = 1;
}

// unique id u of group g = the number of representatives below its own (groups numbered by lowest row, ascending)
__global__ void __launch_bounds__(256) dedup_counts_kernel(const int32_t* __restrict__ gid1, const int32_t* __restrict__ gstart,
                                                           const int32_t* __restrict__ grep, const int32_t* __restrict__ repscan,
                                                           int64_t n, int32_t* __restrict__ cnt, int64_t* __restrict__ uniq_row,
                                                           int64_t* __restrict__ stats) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t U = gid1[n - 1];
  if (g >= U) return;
  const int end = (g + 1 < U) ? gstart[g + 1] : (int)n;
  const int c = end - gstart[g];
  const int u = repscan[grep[g]];
  cnt[u] = c;
  uniq_row[u] = grep[g];
  atomicMax((unsigned long long*)&stats[1], (unsigned long long)c);   // (integer max: exact whatever the order)
  if (g == 0) stats[0] = U;
}

__global__ void __launch_bounds__(256) dedup_members_kernel(const int32_t* __restrict__ gid1, const int32_t* __restrict__ gstart,
                                                            const int32_t* __restrict__ grep, const int32_t* __restrict__ repscan,
                                                            const int32_t* __restrict__ perm, const int32_t* __restrict__ group_ptr,
                                                            int64_t n, int32_t* __restrict__ members) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int g = gid1[p] - 1;
  const int u = repscan[grep[g]];
  members[group_ptr[u] + ((int)p - gstart[g])] = perm[p];
}

struct DedupWs {
  uint64_t *hash_a, *hash_b;
  int32_t *iota, *perm, *flag, *gid1, *gstart, *grep, *isrep, *repscan, *cnt;
  void* temp;
  size_t temp_bytes;
};

static size_t dedup_carve(char* w, int64_t N, DedupWs* out) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = w ? w + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  DedupWs f;
  f.hash_a = reinterpret_cast<uint64_t*>(take((size_t)N * 8));
  f.hash_b = reinterpret_cast<uint64_t*>(take((size_t)N * 8));
  int32_t** const arrays[] = {&f.iota, &f.perm, &f.flag, &f.gid1, &f.gstart, &f.grep, &f.isrep, &f.repscan};
  for (int32_t** a : arrays) *a = reinterpret_cast<int32_t*>(take((size_t)N * 4));
  f.cnt = reinterpret_cast<int32_t*>(take((size_t)(N + 1) * 4));
  const size_t t1 = radix_sort_temp_bytes(N, 4), t2 = scan_temp_bytes(N + 1);
  f.temp_bytes = t1 > t2 ? t1 : t2;
  f.temp = take(f.temp_bytes + 256);
  if (out) *out = f;
  return off;
}

// One wave per query; lane j holds listed group j.
__global__ void __launch_bounds__(256) topk_expand_groups_kernel(const float* __restrict__ su, const int64_t* __restrict__ iu,
                                                                 int ku, const int32_t* __restrict__ group_ptr,
                                                                 const int32_t* __restrict__ members, int64_t U, int64_t B,
                                                                 int k, int64_t idx_base_u, int64_t idx_base,
                                                                 float* __restrict__ out_s, int64_t* __restrict__ out_i) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= B) return;
  float s = RG_NEG_INF;
  int base = 0, c = 0;
  if (lane < ku) {
    s = su[q * ku + lane];
    const int64_t u = iu[q * ku + lane] - idx_base_u;
    if (u >= 0 && u < U) {   // (sharded lists pad with -inf / INT64_MAX: an empty group)
      base = group_ptr[u];
      c = group_ptr[u + 1] - base;
      c = c < k ? c : k;
    }
  }
  // the list is in canonical order (score descending): rows of strictly better groups come first, then the tie run
  int before = 0, run_lo = lane, run_hi = lane, total = 0;
  for (int j = 0; j < ku; ++j) {
    const float sj = __shfl(s, j, 64);
    const int cj = __shfl(c, j, 64);
    total += cj;
    if (cj == 0) continue;
    if (sj > s) before += cj;
    else if (sj == s) {
      run_lo = j < run_lo ? j : run_lo;
      run_hi = j > run_hi ? j : run_hi;
    }
  }
  float* os = out_s + q * k;
  int64_t* oi = out_i + q * k;
  if (c > 0) {
    if (run_lo == run_hi) {   // no other group ties: the group's rows in order
      for (int m = 0; m < c && before + m < k; ++m) {
        os[before + m] = s;
        oi[before + m] = (int64_t)members[base + m] + idx_base;
      }
    } else {                  // tied groups: merged by row (every list ascending, at most k rows of each matter)
      for (int m = 0; m < c; ++m) {
        const int x = members[base + m];
        int pos = before + m;
        for (int j = run_lo; j <= run_hi && pos < k; ++j) {
          if (j == lane) continue;
          const int64_t uj = iu[q * ku + j] - idx_base_u;
          if (su[q * ku + j] != s || uj < 0 || uj >= U) continue;
          const int bj = group_ptr[uj];
          int cj = group_ptr[uj + 1] - bj;
          cj = cj < k ? cj : k;
          for (int t = 0; t < cj && members[bj + t] < x; ++t) ++pos;
        }
        if (pos >= k) break;  // (later rows of this group only rank lower)
        os[pos] = s;
        oi[pos] = (int64_t)x + idx_base;
      }
    }
  }
  for (int p = total + lane; p < k; p += 64) {  // a shard's list may hold fewer than k rows in all
    os[p] = RG_NEG_INF;
    oi[p] = INT64_MAX;
  }
}

}  // namespace ragraph

using namespace ragraph;

#define RG_HIPCUB(call, what)                                                  \
  do {                                                                         \
    hipError_t e__ = (call);                                                   \
    if (e__ != hipSuccess) {                                                   \
      set_error("%s: %s", (what), hipGetErrorString(e__));                     \
      return RAGRAPH_EDEVICE;                                                  \
    }                                                                          \
  } while (0)

extern "C" size_t ragraph_dedup_rows_workspace_bytes(int64_t N) {
  if (N < 1 || N >= (int64_t)INT_MAX) return 0;
  return dedup_carve(nullptr, N, nullptr);
}

extern "C" int ragraph_dedup_rows_f32(const float* Kn, int64_t N, int D, int64_t* stats, int64_t* uniq_row, int32_t* group_ptr,
                                      int32_t* members, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(Kn && stats && uniq_row && group_ptr && members && ws, RAGRAPH_EINVAL, "dedup_rows: null pointer");
  RG_REQUIRE(N >= 1 && N < (int64_t)INT_MAX && D >= 1, RAGRAPH_EINVAL, "dedup_rows: N=%lld D=%d", (long long)N, D);
  RG_REQUIRE(ws_bytes >= ragraph_dedup_rows_workspace_bytes(N), RAGRAPH_EWORKSPACE, "dedup_rows: workspace too small");
  hipStream_t st = as_stream(stream);
  DedupWs f;
  dedup_carve(static_cast<char*>(ws), N, &f);
  const uint32_t* X = reinterpret_cast<const uint32_t*>(Kn);
  const unsigned gw = (unsigned)cdiv(N, 4), gt = (unsigned)cdiv(N, 256);
  hipLaunchKernelGGL(row_hash_kernel, dim3(gw), dim3(256), 0, st, X, N, D, f.hash_a, f.iota);
  // (the radix sort is stable: rows of one hash stay in ascending order, so a group's first row is its lowest)
  int rc = radix_sort_u64(f.hash_a, f.hash_b, f.iota, f.perm, 4, N, 64, f.temp, f.temp_bytes, st);
  if (rc != RAGRAPH_OK) return rc;
  hipLaunchKernelGGL(dedup_flags_kernel, dim3(gw), dim3(256), 0, st, X, N, D, f.hash_b, f.perm, f.flag);
  rc = scan_sum_i32(f.flag, f.gid1, N, true, f.temp, f.temp_bytes, st);
  if (rc != RAGRAPH_OK) return rc;
  RG_HIPCUB(hipMemsetAsync(f.isrep, 0, (size_t)N * 4, st), "dedup_rows(memset)");
  RG_HIPCUB(hipMemsetAsync(f.cnt, 0, (size_t)(N + 1) * 4, st), "dedup_rows(memset)");
  RG_HIPCUB(hipMemsetAsync(stats, 0, 16, st), "dedup_rows(memset)");
  hipLaunchKernelGGL(dedup_heads_kernel, dim3(gt), dim3(256), 0, st, f.flag, f.gid1, f.perm, N, f.gstart, f.grep, f.isrep);
  rc = scan_sum_i32(f.isrep, f.repscan, N, false, f.temp, f.temp_bytes, st);
  if (rc != RAGRAPH_OK) return rc;
  hipLaunchKernelGGL(dedup_counts_kernel, dim3(gt), dim3(256), 0, st, f.gid1, f.gstart, f.grep, f.repscan, N, f.cnt, uniq_row, stats);
  rc = scan_sum_i32(f.cnt, group_ptr, N + 1, false, f.temp, f.temp_bytes, st);
  if (rc != RAGRAPH_OK) return rc;
  hipLaunchKernelGGL(dedup_members_kernel, dim3(gt), dim3(256), 0, st, f.gid1, f.gstart, f.grep, f.repscan, f.perm, group_ptr, N,
                     members);
  RG_CHECK_LAUNCH("dedup_rows");
  return RAGRAPH_OK;
}

extern "C" int ragraph_topk_expand_groups_f32(const float* scores_u, const int64_t* idx_u, int ku, int64_t idx_base_u,
                                              const int32_t* group_ptr, const int32_t* members, int64_t U, int64_t B, int k,
                                              int64_t idx_base, float* out_scores, int64_t* out_idx, void* stream) {
  RG_REQUIRE(scores_u && idx_u && group_ptr && members && out_scores && out_idx, RAGRAPH_EINVAL, "topk_expand_groups: null pointer");
  RG_REQUIRE(B >= 0 && U >= 1 && ku >= 1 && ku <= 64 && k >= 1 && k <= RAGRAPH_TOPK_MAX, RAGRAPH_EINVAL,
             "topk_expand_groups: B=%lld U=%lld ku=%d k=%d", (long long)B, (long long)U, ku, k);
  if (B == 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(topk_expand_groups_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, as_stream(stream), scores_u, idx_u, ku,
                     group_ptr, members, U, B, k, idx_base_u, idx_base, out_scores, out_idx);
  RG_CHECK_LAUNCH("topk_expand_groups");
  return RAGRAPH_OK;
}
