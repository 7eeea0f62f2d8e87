// Top-k over a MATERIALISED score matrix, history masking, and the few-shot position codes.
//
//   topk_rows        torch.topk(scores, k)                    few-shot retrieve: 0.001*structure + 0.999*semantic
//                                                             (RAGraph_node_fewshot/ragraph_utils/ToyGraphBase.py:58-64);
//                                                             edge evaluation: top-20 of user x item ratings
//                                                             (RAGraph_edge/utils/metrics.py:112-118)
//   topk_select_rows the canonical top-k SET for large k       RAGraph_edge/modules/RAGraph.py:57,73,308-321 (vanilla phase:
//                                                             retrieve_num in the thousands, only the winners' mean is used)
//   scatter_fill     batch_pred[i, pos_list] = -1e8           RAGraph_edge/utils/metrics.py:210-214 (_mask_history_pos)
//   floyd_warshall   min-plus all-pairs closure               ragraph_utils/PositionAwareEncoder.py:27-48
//   position_code    1/(d+1) if d < dis_q else 0 to anchors   PositionAwareEncoder.py:6-24
//
// topk_rows is HBM streaming (4 B per score): one 256-thread workgroup per row, 16 B/lane coalesced loads, a wave-level
// threshold, and the wave-cooperative sorted insert shared with the fused kernels; the 4 wave lists are merged at the
// end.  Canonical order (score desc, index asc), so it agrees with the fused cosine top-k on the same scores.
#include "common.h"

namespace ragraph {

__device__ __forceinline__ float rows_insert_coop(float* ls, int* li, int k, float s, int idx, int lane) {
  const bool mine = lane < k;
  float es = mine ? ls[lane] : 0.f;
  int ei = mine ? li[lane] : 0;
  const unsigned long long ahead = __ballot(mine && cand_better(es, ei, s, idx));
  const int pos = __popcll(ahead);
  const float us = __shfl_up(es, 1);
  const int ui = __shfl_up(ei, 1);
  if (pos < k) {
    if (lane == pos) {
      es = s;
      ei = idx;
    } else if (lane > pos) {
      es = us;
      ei = ui;
    }
    if (mine && lane >= pos) {
      ls[lane] = es;
      li[lane] = ei;
    }
  }
  return __shfl(es, k - 1);
}

__global__ void __launch_bounds__(256) topk_rows_kernel(const float* __restrict__ S, int64_t N, int64_t ld, int k,
                                                        float* __restrict__ out_s, int64_t* __restrict__ out_i) {
  __shared__ float ls[4][64];
  __shared__ int li[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.x;
  const float* row = S + b * ld;
  if (lane < 64) {
    ls[wave][lane] = RG_NEG_INF;
    li[wave][lane] = RG_IDX_NONE;
  }
  float thr = RG_NEG_INF;
  const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
  const int64_t nvec = vec ? (N >> 2) : 0;  // float4 chunks; the tail (and unaligned rows) go scalar
  // four 16-B loads per lane in flight: a row is streamed by only four waves, one load at a time would pay the memory
  // latency per 1 KiB
  for (int64_t c00 = (int64_t)wave * 64; c00 < nvec; c00 += 1024) {
    float4 vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t c = c00 + 256 * u + lane;
      vv[u] = make_float4(RG_NEG_INF, RG_NEG_INF, RG_NEG_INF, RG_NEG_INF);
      if (c < nvec) vv[u] = reinterpret_cast<const float4*>(row)[c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t c0 = c00 + 256 * u;
      const int64_t c = c0 + lane;
      const float4 v = vv[u];
      const float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
      if (__any(m >= thr)) {
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          unsigned long long pend = __ballot(e[r] >= thr && c < nvec);
          while (pend) {
            const int src = __ffsll((long long)pend) - 1;
            pend &= pend - 1;
            const float sc = __shfl(e[r], src);
            thr = rows_insert_coop(ls[wave], li[wave], k, sc, (int)(4 * (c0 + src) + r), lane);
            pend &= __ballot(e[r] >= thr);
          }
        }
      }
    }
  }
  for (int64_t e0 = 4 * nvec + (int64_t)wave * 64; e0 < N; e0 += 256) {
    const int64_t e = e0 + lane;
    const float v = (e < N) ? row[e] : RG_NEG_INF;
    unsigned long long pend = __ballot(v >= thr && e < N);
    while (pend) {
      const int src = __ffsll((long long)pend) - 1;
      pend &= pend - 1;
      thr = rows_insert_coop(ls[wave], li[wave], k, __shfl(v, src), (int)(e0 + src), lane);
      pend &= __ballot(v >= thr);
    }
  }
  __syncthreads();
  if (wave == 0) {  // merge the 4 sorted wave lists: k rounds of wave argmax over <= 4*k candidates
    float prev_s = __builtin_huge_valf();
    int prev_i = -1;
    for (int r = 0; r < k; ++r) {
      float best_s = RG_NEG_INF;
      int best_i = RG_IDX_NONE;
      for (int c = lane; c < 4 * k; c += 64) {
        const float s = ls[c / k][c % k];
        const int i = li[c / k][c % k];
        const bool after_prev = (s < prev_s) || (s == prev_s && i > prev_i);
        if (after_prev && cand_better(s, i, best_s, best_i)) {
          best_s = s;
          best_i = i;
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const float os = __shfl_xor(best_s, off);
        const int oi = __shfl_xor(best_i, off);
        if (cand_better(os, oi, best_s, best_i)) {
          best_s = os;
          best_i = oi;
        }
      }
      if (lane == 0) {
        out_s[b * k + r] = best_s;
        out_i[b * k + r] = best_i;
      }
      prev_s = best_s;
      prev_i = best_i;
    }
  }
}

// ---- large k: the canonical top-k SET of a materialised score row (k up to N) --------------------------------------
// The edge flavour's vanilla phase retrieves with retrieve_num in the thousands and consumes only the MEAN of the
// winners' values (RAGraph_edge/modules/RAGraph.py:57,73,308-321): sorted lists are not needed, the set is.  One
// workgroup per row: an exact radix select of the k-th largest score (four 8-bit passes over an order-preserving
// integer image of the floats, 256-bin histograms in LDS), then one ordered pass that writes the indices of every score
// above it and of the first (k - #above) scores equal to it -- the canonical tie rule (score descending, index
// ascending) -- in ASCENDING index order.  Deterministic: no atomics on global memory, positions by prefix sums.
__device__ __forceinline__ unsigned select_key(float f) {  // larger float -> larger unsigned
  unsigned b = __float_as_uint(f);
  if (b == 0x80000000u) b = 0;  // -0 and +0 are one score (they compare equal: a tie, broken by index)
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ void __launch_bounds__(256) topk_select_rows_kernel(const float* __restrict__ S, int64_t N, int64_t ld, int64_t k,
                                                               float* __restrict__ out_kth, int64_t* __restrict__ out_idx) {
  __shared__ unsigned hist[256];
  __shared__ unsigned sh_prefix, sh_need;
  __shared__ unsigned wsum_gt[4], wsum_eq[4];
  __shared__ unsigned long long sh_sel_before;
  __shared__ unsigned sh_eq_before;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = S + (int64_t)blockIdx.x * ld;
  // ---- radix select: after the pass over digit d (most significant first), `prefix` holds the top digits of the k-th
  // largest key and `need` how many keys with that prefix are still to be taken
  unsigned prefix = 0, mask = 0;
  unsigned need = (unsigned)k;  // k <= N < 2^31
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[tid] = 0;
    __syncthreads();
    for (int64_t e = tid; e < N; e += 256) {
      const unsigned key = select_key(row[e]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned acc = 0;
      int d = 255;
      for (; d > 0; --d) {
        if (acc + hist[d] >= need) break;
        acc += hist[d];
      }
      sh_prefix = prefix | ((unsigned)d << shift);
      sh_need = need - acc;
    }
    __syncthreads();
    prefix = sh_prefix;
    need = sh_need;
    mask |= 255u << shift;
    __syncthreads();
  }
  const unsigned kth_key = prefix;  // the k-th largest key; `need` of the keys equal to it are taken (lowest indices)
  if (tid == 0) {
    const unsigned b = (kth_key & 0x80000000u) ? (kth_key & 0x7FFFFFFFu) : ~kth_key;
    out_kth[blockIdx.x] = __uint_as_float(b);
    sh_sel_before = 0;
    sh_eq_before = 0;
  }
  __syncthreads();
  // ---- ordered compaction, 256 scores per round
  int64_t* out = out_idx + (int64_t)blockIdx.x * k;
  for (int64_t e0 = 0; e0 < N; e0 += 256) {
    const int64_t e = e0 + tid;
    unsigned key = 0;
    bool gt = false, eq = false;
    if (e < N) {
      key = select_key(row[e]);
      gt = key > kth_key;
      eq = key == kth_key;
    }
    const unsigned long long bg = __ballot(gt), be = __ballot(eq);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) {
      wsum_gt[wave] = (unsigned)__popcll(bg);
      wsum_eq[wave] = (unsigned)__popcll(be);
    }
    __syncthreads();
    unsigned gt_before = 0, eq_before_w = 0;
    for (int w = 0; w < wave; ++w) {
      gt_before += wsum_gt[w];
      eq_before_w += wsum_eq[w];
    }
    const unsigned eq_rank = sh_eq_before + eq_before_w + (unsigned)__popcll(be & below);  // among ALL equals so far
    const bool sel = gt || (eq && eq_rank < need);
    // selected equals before this element (in this round) = equals before it whose rank < need
    const unsigned eq_taken_before_round = sh_eq_before < need ? sh_eq_before : need;
    const unsigned eq_before_here = sh_eq_before + eq_before_w + (unsigned)__popcll(be & below);
    const unsigned eq_taken_before_here = (eq_before_here < need ? eq_before_here : need) - eq_taken_before_round;
    const unsigned long long pos = sh_sel_before + gt_before + (unsigned)__popcll(bg & below) + eq_taken_before_here;
    if (sel) out[pos] = e;
    __syncthreads();
    if (tid == 0) {
      unsigned tg = 0, te = 0;
      for (int w = 0; w < 4; ++w) {
        tg += wsum_gt[w];
        te += wsum_eq[w];
      }
      const unsigned eq_after = sh_eq_before + te;
      const unsigned taken = (eq_after < need ? eq_after : need) - eq_taken_before_round;
      sh_sel_before += tg + taken;
      sh_eq_before = eq_after;
    }
    __syncthreads();
  }
}

// ---- the same selection for LONG rows (N >= SELECT_CHUNK * 4), with a workspace: one workgroup per row walks 4 M
// scores five times alone (41 ms for 64 rows of a 4 M-key bank: 64 of 256 CUs busy, and 15 000 barrier rounds of
// compaction each).  Here a row is cut into chunks of SELECT_CHUNK scores and every pass is a launch over (chunk, row):
//   4 x { select_hist: LDS histogram of the chunk's keys that match the prefix found so far, integer atomicAdd into the
//         row's 256 global bins (exact whatever the order);  select_pick: the digit that holds the k-th largest }
//   select_count: per chunk, how many keys are above / equal to the k-th largest
//   select_scan:  exclusive prefix over the chunks -> every chunk's first output position and its share of the ties
//   select_compact: every chunk writes its winners in ascending index order at its offset.
// Same result as topk_select_rows_kernel, bit for bit (the canonical set, ascending indices).
constexpr int SELECT_CHUNK = 16384;

struct SelectState {  // per row
  unsigned prefix, mask, need, kth_key;
};

__global__ void __launch_bounds__(256) select_init_kernel(SelectState* __restrict__ st, unsigned* __restrict__ hist, int64_t B,
                                                          int64_t k) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < B) st[i] = SelectState{0u, 0u, (unsigned)k, 0u};
  if (i < B * 256) hist[i] = 0u;
}

__global__ void __launch_bounds__(256) select_hist_kernel(const float* __restrict__ S, int64_t N, int64_t ld, int shift,
                                                          const SelectState* __restrict__ st, unsigned* __restrict__ hist) {
  __shared__ unsigned h[256];
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.y;
  const unsigned prefix = st[b].prefix, mask = st[b].mask;
  h[tid] = 0;
  __syncthreads();
  const float* row = S + b * ld;
  const int64_t e0 = (int64_t)blockIdx.x * SELECT_CHUNK;
  const int64_t e1 = e0 + SELECT_CHUNK < N ? e0 + SELECT_CHUNK : N;
  for (int64_t e = e0 + tid; e < e1; e += 256) {
    const unsigned key = select_key(row[e]);
    if ((key & mask) == prefix) atomicAdd(&h[(key >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[tid]) atomicAdd(hist + b * 256 + tid, h[tid]);
}

__global__ void __launch_bounds__(256) select_pick_kernel(SelectState* __restrict__ st, unsigned* __restrict__ hist, int shift) {
  __shared__ unsigned h[256];
  const int64_t b = blockIdx.x;
  h[threadIdx.x] = hist[b * 256 + threadIdx.x];
  hist[b * 256 + threadIdx.x] = 0u;  // ready for the next pass
  __syncthreads();
  if (threadIdx.x == 0) {
    SelectState s = st[b];
    unsigned acc = 0;
    int d = 255;
    for (; d > 0; --d) {
      if (acc + h[d] >= s.need) break;
      acc += h[d];
    }
    s.prefix |= (unsigned)d << shift;
    s.need -= acc;
    s.mask |= 255u << shift;
    if (shift == 0) s.kth_key = s.prefix;
    st[b] = s;
  }
}

__global__ void __launch_bounds__(256) select_count_kernel(const float* __restrict__ S, int64_t N, int64_t ld,
                                                           const SelectState* __restrict__ st, unsigned* __restrict__ counts,
                                                           int nchunks) {
  __shared__ unsigned wg[4], we[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.y;
  const unsigned kth = st[b].kth_key;
  const float* row = S + b * ld;
  const int64_t e0 = (int64_t)blockIdx.x * SELECT_CHUNK;
  const int64_t e1 = e0 + SELECT_CHUNK < N ? e0 + SELECT_CHUNK : N;
  unsigned g = 0, q = 0;
  for (int64_t e = e0 + tid; e < e1; e += 256) {
    const unsigned key = select_key(row[e]);
    g += key > kth;
    q += key == kth;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    g += __shfl_xor(g, off);
    q += __shfl_xor(q, off);
  }
  if (lane == 0) {
    wg[wave] = g;
    we[wave] = q;
  }
  __syncthreads();
  if (tid == 0) {
    counts[(b * nchunks + blockIdx.x) * 2] = wg[0] + wg[1] + wg[2] + wg[3];
    counts[(b * nchunks + blockIdx.x) * 2 + 1] = we[0] + we[1] + we[2] + we[3];
  }
}

// counts[b][c] = (above, equal) of chunk c  ->  (first output position of the chunk, equals before the chunk)
__global__ void __launch_bounds__(64) select_scan_kernel(const SelectState* __restrict__ st, unsigned* __restrict__ counts,
                                                         int nchunks, float* __restrict__ out_kth, int64_t B) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const unsigned need = st[b].need, kth_key = st[b].kth_key;
  out_kth[b] = __uint_as_float((kth_key & 0x80000000u) ? (kth_key & 0x7FFFFFFFu) : ~kth_key);
  unsigned pos = 0, eq_before = 0;
  for (int c = 0; c < nchunks; ++c) {
    unsigned* p = counts + (b * nchunks + c) * 2;
    const unsigned g = p[0], q = p[1];
    p[0] = pos;
    p[1] = eq_before;
    const unsigned taken_before = eq_before < need ? eq_before : need;
    const unsigned eq_after = eq_before + q;
    pos += g + ((eq_after < need ? eq_after : need) - taken_before);
    eq_before = eq_after;
  }
}

__global__ void __launch_bounds__(256) select_compact_kernel(const float* __restrict__ S, int64_t N, int64_t ld, int64_t k,
                                                             const SelectState* __restrict__ st,
                                                             const unsigned* __restrict__ counts, int nchunks,
                                                             int64_t* __restrict__ out_idx) {
  __shared__ unsigned wsum_gt[4], wsum_eq[4];
  __shared__ unsigned sh_sel_before, sh_eq_before;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.y;
  const unsigned kth_key = st[b].kth_key, need = st[b].need;
  const float* row = S + b * ld;
  int64_t* out = out_idx + b * k;
  const int64_t c0 = (int64_t)blockIdx.x * SELECT_CHUNK;
  const int64_t c1 = c0 + SELECT_CHUNK < N ? c0 + SELECT_CHUNK : N;
  if (tid == 0) {
    sh_sel_before = counts[(b * nchunks + blockIdx.x) * 2];
    sh_eq_before = counts[(b * nchunks + blockIdx.x) * 2 + 1];
  }
  __syncthreads();
  for (int64_t e0 = c0; e0 < c1; e0 += 256) {  // (the ordered compaction of topk_select_rows_kernel, from the chunk's offsets)
    const int64_t e = e0 + tid;
    bool gt = false, eq = false;
    if (e < c1) {
      const unsigned key = select_key(row[e]);
      gt = key > kth_key;
      eq = key == kth_key;
    }
    const unsigned long long bg = __ballot(gt), be = __ballot(eq);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) {
      wsum_gt[wave] = (unsigned)__popcll(bg);
      wsum_eq[wave] = (unsigned)__popcll(be);
    }
    __syncthreads();
    unsigned gt_before = 0, eq_before_w = 0;
    for (int w = 0; w < wave; ++w) {
      gt_before += wsum_gt[w];
      eq_before_w += wsum_eq[w];
    }
    const unsigned eq_before_here = sh_eq_before + eq_before_w + (unsigned)__popcll(be & below);
    const bool sel = gt || (eq && eq_before_here < need);
    const unsigned eq_taken_before_round = sh_eq_before < need ? sh_eq_before : need;
    const unsigned eq_taken_before_here = (eq_before_here < need ? eq_before_here : need) - eq_taken_before_round;
    const unsigned pos = sh_sel_before + gt_before + (unsigned)__popcll(bg & below) + eq_taken_before_here;
    if (sel) out[pos] = e;
    __syncthreads();
    if (tid == 0) {
      unsigned tg = 0, te = 0;
      for (int w = 0; w < 4; ++w) {
        tg += wsum_gt[w];
        te += wsum_eq[w];
      }
      const unsigned eq_after = sh_eq_before + te;
      sh_sel_before += tg + ((eq_after < need ? eq_after : need) - eq_taken_before_round);
      sh_eq_before = eq_after;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) scatter_fill_kernel(float* __restrict__ S, int64_t ld,
                                                           const int64_t* __restrict__ rowptr,
                                                           const int64_t* __restrict__ col, float value) {
  const int64_t b = blockIdx.x;
  for (int64_t e = rowptr[b] + threadIdx.x; e < rowptr[b + 1]; e += 256) S[b * ld + col[e]] = value;
}

// One step k of Floyd-Warshall over the whole matrix: d[i][j] = min(d[i][j], d[i][k] + d[k][j]).  Row k and column k
// are fixed points of step k (d[k][k] = 0), so the in-place update is race-free.
__global__ void __launch_bounds__(256) fw_step_kernel(float* __restrict__ d, int n, int kk) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= n) return;
  const float via = __fadd_rn(d[(int64_t)i * n + kk], d[(int64_t)kk * n + j]);
  const float cur = d[(int64_t)i * n + j];
  if (via < cur) d[(int64_t)i * n + j] = via;
}

// dist init: PositionAwareEncoder.py:38-41: dist = adj; dist[adj == 0] = inf; diagonal = 0
__global__ void __launch_bounds__(256) fw_init_kernel(const float* __restrict__ adj, float* __restrict__ d, int n) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)n * n) return;
  const int i = (int)(e / n), j = (int)(e % n);
  const float a = adj[e];
  d[e] = (i == j) ? 0.f : (a == 0.f ? __builtin_huge_valf() : a);
}

__global__ void __launch_bounds__(256) position_code_kernel(const float* __restrict__ d, int n,
                                                            const int64_t* __restrict__ anchors, int A, float dis_q,
                                                            float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)n * A) return;
  const int u = (int)(e / A), a = (int)(e % A);
  const float dist = d[(int64_t)u * n + anchors[a]];
  out[e] = (dist < dis_q) ? 1.f / (dist + 1.f) : 0.f;
}

// ---- distances to the anchors only, on the CSR (few-shot query graphs, per forward) ---------------------------------------
// PositionAwareEncoder.py:6-24 needs dist[u, anchor] for 10 anchors; its all-pairs Floyd-Warshall (:27-48) over the dense
// block-diagonal batch adjacency is O(n^3) in n dependent steps.  Here: one workgroup per anchor keeps d[0..n) in LDS and
// relaxes every row against its out-edges, d[u] = min(d[u], val[u,v] + d[v]), until a round changes nothing (<= n rounds).
// fp32 addition and min are monotone, so the rounds converge to ONE fixpoint whatever the order in which rows see each
// other's updates: per node the minimum over all walks to the anchor of the sum w1 + (w2 + (w3 + ...)) -- the value
// oracle/ragraph_oracle.c computes sequentially, bit for bit.  (Floyd-Warshall associates a path's sum differently:
// equal to ~1 ulp, pinned at 1e-6 on the codes against the reference's golden vectors.)  Entries with val == 0 are "no
// edge", as the reference's `dist[adj == 0] = inf`; the diagonal is 0.
__global__ void __launch_bounds__(1024) anchor_dist_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const float* __restrict__ val, int n,
                                                           const int64_t* __restrict__ anchors, int A, float dis_q,
                                                           float* __restrict__ codes, float* __restrict__ dist_out) {
  extern __shared__ float anchor_d[];
  __shared__ int changed;
  const int a = blockIdx.x, tid = threadIdx.x;
  const int src = (int)anchors[a];
  for (int u = tid; u < n; u += 1024) anchor_d[u] = (u == src) ? 0.f : __builtin_huge_valf();
  __syncthreads();
  for (int round = 0; round < n; ++round) {
    if (tid == 0) changed = 0;
    __syncthreads();
    for (int u = tid; u < n; u += 1024) {
      const float cur = anchor_d[u];
      float best = cur;
      const int64_t e1 = rowptr[u + 1];
      for (int64_t e = rowptr[u]; e < e1; ++e) {
        const float w = val[e];
        const int v = col[e];
        if (w != 0.f && v != u) best = fminf(best, __fadd_rn(w, anchor_d[v]));
      }
      if (best < cur) {
        anchor_d[u] = best;
        changed = 1;
      }
    }
    __syncthreads();
    const int again = changed;
    __syncthreads();
    if (!again) break;
  }
  for (int u = tid; u < n; u += 1024) {
    const float d = anchor_d[u];
    if (dist_out) dist_out[(int64_t)u * A + a] = d;
    codes[(int64_t)u * A + a] = (d < dis_q) ? 1.f / (d + 1.f) : 0.f;
  }
}

}  // namespace ragraph

using namespace ragraph;

extern "C" int ragraph_topk_rows_f32(const float* S, int64_t B, int64_t N, int64_t ld, int k, float* out_scores,
                                     int64_t* out_idx, void* stream) {
  RG_REQUIRE(S && out_scores && out_idx, RAGRAPH_EINVAL, "topk_rows: null pointer");
  RG_REQUIRE(B >= 0 && N >= 1 && ld >= N, RAGRAPH_EINVAL, "topk_rows: bad shape");
  RG_REQUIRE(k >= 1 && k <= N, RAGRAPH_EINVAL, "topk_rows: k=%d out of range for N=%lld", k, (long long)N);
  RG_REQUIRE(k <= 64, RAGRAPH_EUNSUPPORTED, "topk_rows: k=%d > 64", k);
  RG_REQUIRE(N < (int64_t)INT_MAX, RAGRAPH_EUNSUPPORTED, "topk_rows: N must fit int32");
  if (B == 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)B), dim3(256), 0, as_stream(stream), S, N, ld, k, out_scores,
                     out_idx);
  RG_CHECK_LAUNCH("topk_rows");
  return RAGRAPH_OK;
}

extern "C" int ragraph_topk_select_rows_f32(const float* S, int64_t B, int64_t N, int64_t ld, int64_t k, float* out_kth,
                                            int64_t* out_idx, void* stream) {
  RG_REQUIRE(S && out_kth && out_idx, RAGRAPH_EINVAL, "topk_select_rows: null pointer");
  RG_REQUIRE(B >= 0 && N >= 1 && ld >= N, RAGRAPH_EINVAL, "topk_select_rows: bad shape");
  RG_REQUIRE(k >= 1 && k <= N, RAGRAPH_EINVAL, "topk_select_rows: k=%lld out of range for N=%lld", (long long)k, (long long)N);
  RG_REQUIRE(N < (int64_t)INT_MAX, RAGRAPH_EUNSUPPORTED, "topk_select_rows: N must fit int32");
  if (B == 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(topk_select_rows_kernel, dim3((unsigned)B), dim3(256), 0, as_stream(stream), S, N, ld, k, out_kth,
                     out_idx);
  RG_CHECK_LAUNCH("topk_select_rows");
  return RAGRAPH_OK;
}

extern "C" size_t ragraph_topk_select_rows_workspace_bytes(int64_t B, int64_t N) {
  if (B < 1 || N < 1) return 0;
  const int64_t nchunks = cdiv(N, (int64_t)SELECT_CHUNK);
  return align_up((size_t)B * sizeof(SelectState), 256) + align_up((size_t)B * 256 * sizeof(unsigned), 256) +
         align_up((size_t)B * nchunks * 2 * sizeof(unsigned), 256);
}

extern "C" int ragraph_topk_select_rows_ws_f32(const float* S, int64_t B, int64_t N, int64_t ld, int64_t k, float* out_kth,
                                               int64_t* out_idx, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(S && out_kth && out_idx, RAGRAPH_EINVAL, "topk_select_rows: null pointer");
  RG_REQUIRE(B >= 0 && N >= 1 && ld >= N, RAGRAPH_EINVAL, "topk_select_rows: bad shape");
  RG_REQUIRE(k >= 1 && k <= N, RAGRAPH_EINVAL, "topk_select_rows: k=%lld out of range for N=%lld", (long long)k, (long long)N);
  RG_REQUIRE(N < (int64_t)INT_MAX, RAGRAPH_EUNSUPPORTED, "topk_select_rows: N must fit int32");
  if (B == 0) return RAGRAPH_OK;
  const int64_t nchunks = cdiv(N, (int64_t)SELECT_CHUNK);
  if (!ws || nchunks < 4)  // short rows: one workgroup per row does it all
    return ragraph_topk_select_rows_f32(S, B, N, ld, k, out_kth, out_idx, stream);
  RG_REQUIRE(ws_bytes >= ragraph_topk_select_rows_workspace_bytes(B, N), RAGRAPH_EWORKSPACE, "topk_select_rows: workspace too small");
  RG_REQUIRE(B <= 65535, RAGRAPH_EUNSUPPORTED, "topk_select_rows(ws): B=%lld rows per call", (long long)B);
  char* w = static_cast<char*>(ws);
  SelectState* st = reinterpret_cast<SelectState*>(w);
  w += align_up((size_t)B * sizeof(SelectState), 256);
  unsigned* hist = reinterpret_cast<unsigned*>(w);
  w += align_up((size_t)B * 256 * sizeof(unsigned), 256);
  unsigned* counts = reinterpret_cast<unsigned*>(w);
  hipStream_t sm = as_stream(stream);
  const dim3 grid((unsigned)nchunks, (unsigned)B);
  hipLaunchKernelGGL(select_init_kernel, dim3((unsigned)cdiv(B * 256, 256)), dim3(256), 0, sm, st, hist, B, k);
  for (int shift = 24; shift >= 0; shift -= 8) {
    hipLaunchKernelGGL(select_hist_kernel, grid, dim3(256), 0, sm, S, N, ld, shift, st, hist);
    hipLaunchKernelGGL(select_pick_kernel, dim3((unsigned)B), dim3(256), 0, sm, st, hist, shift);
  }
  hipLaunchKernelGGL(select_count_kernel, grid, dim3(256), 0, sm, S, N, ld, st, counts, (int)nchunks);
  hipLaunchKernelGGL(select_scan_kernel, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, sm, st, counts, (int)nchunks, out_kth, B);
  hipLaunchKernelGGL(select_compact_kernel, grid, dim3(256), 0, sm, S, N, ld, k, st, counts, (int)nchunks, out_idx);
  RG_CHECK_LAUNCH("topk_select_rows(ws)");
  return RAGRAPH_OK;
}

extern "C" int ragraph_scatter_fill_f32(float* S, int64_t B, int64_t N, int64_t ld, const int64_t* rowptr,
                                        const int64_t* col, float value, void* stream) {
  RG_REQUIRE(S && rowptr && (col || B == 0), RAGRAPH_EINVAL, "scatter_fill: null pointer");
  RG_REQUIRE(B >= 0 && N >= 1 && ld >= N, RAGRAPH_EINVAL, "scatter_fill: bad shape");
  if (B == 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(scatter_fill_kernel, dim3((unsigned)B), dim3(256), 0, as_stream(stream), S, ld, rowptr, col, value);
  RG_CHECK_LAUNCH("scatter_fill");
  return RAGRAPH_OK;
}

extern "C" int ragraph_floyd_warshall_f32(const float* adj, int n, float* dist, void* stream) {
  RG_REQUIRE(adj && dist, RAGRAPH_EINVAL, "floyd_warshall: null pointer");
  RG_REQUIRE(n >= 1 && n <= 46340, RAGRAPH_EINVAL, "floyd_warshall: n=%d", n);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(fw_init_kernel, dim3((unsigned)cdiv((int64_t)n * n, 256)), dim3(256), 0, st, adj, dist, n);
  for (int kk = 0; kk < n; ++kk)
    hipLaunchKernelGGL(fw_step_kernel, dim3((unsigned)cdiv(n, 256), (unsigned)n), dim3(256), 0, st, dist, n, kk);
  RG_CHECK_LAUNCH("floyd_warshall");
  return RAGRAPH_OK;
}

extern "C" int ragraph_position_codes_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n,
                                              const int64_t* anchors, int A, float dis_q, float* codes, float* dist,
                                              void* stream) {
  RG_REQUIRE(rowptr && anchors && codes, RAGRAPH_EINVAL, "position_codes_csr: null pointer");
  RG_REQUIRE(n >= 1 && A >= 1, RAGRAPH_EINVAL, "position_codes_csr: bad shape");
  RG_REQUIRE(n <= 40000, RAGRAPH_EUNSUPPORTED, "position_codes_csr: n=%lld: the distance vector of one anchor must fit the "
             "160 KB of LDS (few-shot query batches have a few hundred nodes)", (long long)n);
  const size_t lds = (size_t)n * sizeof(float);
  static DeviceOnce lds_once;
  if (hipError_t e = raise_dynamic_lds(lds_once, &anchor_dist_kernel, 160 * 1024 - 64); e != hipSuccess) {
    set_error("position_codes_csr: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    return RAGRAPH_EDEVICE;
  }
  hipLaunchKernelGGL(anchor_dist_kernel, dim3((unsigned)A), dim3(1024), lds, as_stream(stream), rowptr, col, val, (int)n, anchors,
                     A, dis_q, codes, dist);
  RG_CHECK_LAUNCH("position_codes_csr");
  return RAGRAPH_OK;
}

extern "C" int ragraph_position_code_f32(const float* dist, int n, const int64_t* anchors, int A, float dis_q,
                                         float* out, void* stream) {
  RG_REQUIRE(dist && anchors && out, RAGRAPH_EINVAL, "position_code: null pointer");
  RG_REQUIRE(n >= 1 && A >= 1, RAGRAPH_EINVAL, "position_code: bad shape");
  hipLaunchKernelGGL(position_code_kernel, dim3((unsigned)cdiv((int64_t)n * A, 256)), dim3(256), 0, as_stream(stream),
                     dist, n, anchors, A, dis_q, out);
  RG_CHECK_LAUNCH("position_code");
  return RAGRAPH_OK;
}
