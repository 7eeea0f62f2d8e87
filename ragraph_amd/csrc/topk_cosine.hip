// Fused cosine-score + top-k over a key bank (the north-star kernel).
//
// Replaces  normalize(Q) @ normalize(K).T  ->  torch.topk(k)   (RAGraph_node/ragraph_utils/SimilarityFunctions.py:6-16,
// ToyGraphBase.py:66-67; slab loop RAGraph_edge/modules/RAGraph.py:298-311) without ever writing the B x N scores.
//
// Mapping onto CDNA4 (MI355X):
//   * the contraction is a dense fp32 GEMM with a tiny K (= D <= 256), so it runs on v_mfma_f32_32x32x2_f32 with the
//     KEYS as the A operand (streamed) and 32 QUERIES per wave as the B operand, resident in D/2 VGPRs for the whole
//     stream.  With keys on the rows, the 32x32 result puts one query on each lane column: a lane owns 16 scores of
//     ONE query per tile, so the running top-k needs no cross-lane traffic (lanes l and l+32 share a query).
//   * a workgroup = 8 waves (2 per SIMD, so one wave's list update hides under the other's MFMAs) = 256 queries; it
//     streams its key range in 32 KiB stages: global_load_dwordx4 (16 B/lane, one 1 KiB key row per wave-instruction
//     at D=256) issued one stage ahead into registers, written to a double-buffered, 16-B-padded LDS image after the
//     MFMAs of the current stage (issue-early / write-late), one barrier per stage.
//   * the grid is (query tiles) x (key splits); every (tile, split) leaves an UNSORTED k-candidate partial list and
//     ragraph::topk_merge selects the final canonical top-k.  HBM traffic is ~the bank once per 32 concurrently
//     resident query tiles; the kernel is MFMA-bound (arithmetic intensity = 128 flop/B), see DESIGN.md.
//   * every score is one fmaf chain in natural k order (MFMA 32x32x2 semantics), so results do not depend on the
//     split count, the query batch size, or how the bank is sharded across GPUs.
#include "common.h"

namespace ragraph {

// ------------------------------------------------------------------------------------------------------------------
// Per-query candidate list in LDS, unsorted, replace-the-worst policy.
//   ls/li: [k][QT] (position-major so the 32 lanes of a half-wave hit 32 banks), thr_q[QT] = score of the worst kept.
// Offered to by at most one lane per query at a time (the two half-waves that share a query are serialised).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void list_offer(float* ls, int* li, float* thr_q, int q, int k, int QT, float s, int idx) {
  // one scan: worst and second-worst of the current list
  float w_s = ls[q];
  int w_i = li[q];
  int w_p = 0;
  float w2_s = __builtin_huge_valf();  // "better than anything" until a second element is seen
  int w2_i = -1;
  for (int p = 1; p < k; ++p) {
    const float cs = ls[p * QT + q];
    const int ci = li[p * QT + q];
    if (cand_better(w_s, w_i, cs, ci)) {  // (cs,ci) is worse than the current worst -> new worst
      w2_s = w_s;
      w2_i = w_i;
      w_s = cs;
      w_i = ci;
      w_p = p;
    } else if (cand_better(w2_s, w2_i, cs, ci)) {
      w2_s = cs;
      w2_i = ci;
    }
  }
  if (!cand_better(s, idx, w_s, w_i)) return;  // not better than the k-th best: list unchanged
  ls[w_p * QT + q] = s;
  li[w_p * QT + q] = idx;
  // new worst = worse of (second worst, the new element)
  thr_q[q] = (k == 1) ? s : (cand_better(s, idx, w2_s, w2_i) ? w2_s : s);
}

struct TopkParams {
  const float* Qn;   // [B,D] normalised queries
  const float* Kn;   // [N,D] normalised keys
  int64_t B, N;
  int k;
  int nsplit;
  int64_t keys_per_split;  // multiple of the stage size
  float* part_s;           // [B][nsplit][k]
  int* part_i;
};

template <int D>
struct TopkCfg {
  static constexpr int WAVES = 8;
  static constexpr int THREADS = WAVES * 64;
  static constexpr int QT = WAVES * 32;                 // queries per workgroup
  static constexpr int TILES = 256 / D;                 // 32-key MFMA tiles per stage (1 at D=256)
  static constexpr int STAGE_KEYS = 32 * TILES;         // 32 KiB of keys per stage
  static constexpr int ROW = D + 4;                     // padded LDS row (floats): conflict-free ds_read_b128
  static constexpr int STAGE_FLOATS = STAGE_KEYS * ROW;
  static constexpr int CHUNKS = STAGE_KEYS * (D / 4);   // float4 chunks per stage = 2048
  static constexpr int LOADS = CHUNKS / THREADS;        // float4 loads per thread per stage = 4
  static size_t lds_bytes(int k) { return sizeof(float) * (2 * STAGE_FLOATS + QT) + (size_t)k * QT * 8; }
};

template <int D>
__global__ void __launch_bounds__(512, 2) topk_stream_kernel(TopkParams p) {
  using C = TopkCfg<D>;
  // ONE __shared__ object; everything below is an offset from it so every access stays a ds_* instruction.
  extern __shared__ float4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  constexpr int OFF_THR = 2 * C::STAGE_FLOATS;  // [QT]
  constexpr int OFF_LS = OFF_THR + C::QT;       // [k][QT] scores, then [k][QT] indices
  float* thr_q = smem + OFF_THR;
  float* ls = smem + OFF_LS;
  int* li = reinterpret_cast<int*>(smem + OFF_LS + p.k * C::QT);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int k = p.k;

  // block -> (query tile, key split); consecutive blocks differ in split, so the blocks that share an XCD
  // (blockIdx % 8 equal) stream the same key range while they are co-resident (speed only).
  const int split = blockIdx.x % p.nsplit;
  const int64_t qtile = blockIdx.x / p.nsplit;
  const int64_t q0 = qtile * C::QT;
  const int64_t n_begin = (int64_t)split * p.keys_per_split;
  const int64_t n_end = min(p.N, n_begin + p.keys_per_split);
  const int nstages = (int)((n_end - n_begin + C::STAGE_KEYS - 1) / C::STAGE_KEYS);

  // ---- B operand: this lane's query (row q0 + wave*32 + j), k-slots h, h+2, h+4, ... ------------------------
  float breg[D / 2];
  {
    int64_t q = q0 + wave * 32 + j;
    if (q > p.B - 1) q = p.B - 1;  // clamp: results of padded queries are never written
    const float4* qp = reinterpret_cast<const float4*>(p.Qn + q * D);
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
      const float4 v = qp[c];
      breg[2 * c] = h ? v.y : v.x;
      breg[2 * c + 1] = h ? v.w : v.z;
    }
  }

  for (int i = tid; i < k * C::QT; i += C::THREADS) {
    ls[i] = RG_NEG_INF;
    li[i] = RG_IDX_NONE;
  }
  for (int i = tid; i < C::QT; i += C::THREADS) thr_q[i] = RG_NEG_INF;

  // ---- staging: thread t owns float4 chunks t, t+512, ... of the stage (row = chunk / (D/4)) ----------------
  // The loads are unconditional (the stage index is clamped, the write is skipped) so the four float4 stay in VGPRs.
  static_assert(C::LOADS == 4, "staging is written for 4 float4 per thread per stage");
  float4 pre0, pre1, pre2, pre3;
  const int srow = tid / (D / 4);                 // row of chunk i inside the stage: srow + i * (THREADS / (D/4))
  const int scol = 4 * (tid % (D / 4));
  constexpr int SROWS = C::THREADS / (D / 4);     // rows covered by one pass of the 512 threads
#define RG_STAGE_LOAD(s_)                                                                         \
  do {                                                                                            \
    const int64_t key0_ = n_begin + (int64_t)(s_) * C::STAGE_KEYS + srow;                         \
    const int64_t last_ = p.N - 1; /* tail: duplicate the last key, masked by index later */      \
    const int64_t r0_ = min(key0_, last_), r1_ = min(key0_ + SROWS, last_);                       \
    const int64_t r2_ = min(key0_ + 2 * SROWS, last_), r3_ = min(key0_ + 3 * SROWS, last_);       \
    pre0 = *reinterpret_cast<const float4*>(p.Kn + r0_ * D + scol);                               \
    pre1 = *reinterpret_cast<const float4*>(p.Kn + r1_ * D + scol);                               \
    pre2 = *reinterpret_cast<const float4*>(p.Kn + r2_ * D + scol);                               \
    pre3 = *reinterpret_cast<const float4*>(p.Kn + r3_ * D + scol);                               \
  } while (0)
#define RG_STAGE_WRITE(buf_)                                                                      \
  do {                                                                                            \
    float* d_ = smem + (buf_) * C::STAGE_FLOATS + srow * C::ROW + scol;                           \
    *reinterpret_cast<float4*>(d_) = pre0;                                                        \
    *reinterpret_cast<float4*>(d_ + SROWS * C::ROW) = pre1;                                       \
    *reinterpret_cast<float4*>(d_ + 2 * SROWS * C::ROW) = pre2;                                   \
    *reinterpret_cast<float4*>(d_ + 3 * SROWS * C::ROW) = pre3;                                   \
  } while (0)

  RG_STAGE_LOAD(0);
  RG_STAGE_WRITE(0);
  __syncthreads();

  const int ql = wave * 32 + j;  // query slot inside the workgroup
  float thr = RG_NEG_INF;

  for (int s = 0; s < nstages; ++s) {
    const bool more = (s + 1 < nstages);
    RG_STAGE_LOAD(more ? s + 1 : s);  // in flight under this stage's MFMAs
    __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks the loads below the MFMA block (to save VGPRs)
    const int cur = (s & 1) * C::STAGE_FLOATS;

#pragma unroll 1
    for (int t = 0; t < C::TILES; ++t) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* arow = smem + cur + (t * 32 + j) * C::ROW;
#pragma unroll
      for (int c = 0; c < D / 4; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(arow + 4 * c);
        const float a0 = h ? v.y : v.x;  // key[j][4c + h]
        const float a1 = h ? v.w : v.z;  // key[j][4c + 2 + h]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, breg[2 * c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, breg[2 * c + 1], acc, 0, 0, 0);
      }

      // ---- epilogue: acc[r] = score(key row (r&3) + 8*(r>>2) + 4*h of the tile, query j) ------------------
      float m = acc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
      if (__any(m >= thr)) {
        // rare path (~k ln(n/k) times per query over the stream): offer candidates one per lane per round, lowest
        // key index first; the two half-waves (same queries) take turns so a list has one writer at a time.
        const int key_base = (int)(n_begin + (int64_t)s * C::STAGE_KEYS + t * 32) + 4 * h;
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int idx = key_base + (r & 3) + 8 * (r >> 2);
          if (acc[r] >= thr && idx < (int)n_end) mask |= 1u << r;
        }
        while (__any(mask != 0)) {
          const int r0 = __ffs(mask) - 1;  // -1 when this lane has nothing left
          float sc = acc[0];
#pragma unroll
          for (int r = 1; r < 16; ++r) sc = (r0 == r) ? acc[r] : sc;
          const int idx = key_base + (r0 & 3) + 8 * (r0 >> 2);
#pragma unroll 1
          for (int hh = 0; hh < 2; ++hh) {
            if (mask != 0 && h == hh) list_offer(ls, li, thr_q, ql, k, C::QT, sc, idx);
          }
          mask &= mask - 1;
          thr = thr_q[ql];
          // candidates that no longer reach the tightened threshold can be dropped without an offer
        }
      }
    }

    if (more) RG_STAGE_WRITE((s + 1) & 1);
    __syncthreads();
    thr = thr_q[ql];  // the partner half-wave may have tightened it
  }

#undef RG_STAGE_LOAD
#undef RG_STAGE_WRITE

  // ---- write this (tile, split)'s unsorted candidates; empty slots stay (-inf, IDX_NONE) --------------------
  for (int i = tid; i < k * C::QT; i += C::THREADS) {
    const int pos = i / C::QT, q = i % C::QT;
    const int64_t qg = q0 + q;
    if (qg < p.B) {
      const int64_t o = (qg * p.nsplit + split) * k + pos;
      p.part_s[o] = ls[i];
      p.part_i[o] = li[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Select the canonical top-k among M candidates per query (partials of the splits, or the per-GPU lists).
// One wave per query; round r picks the best candidate strictly worse than round r-1's winner, so no marking is
// needed and unsorted input is fine.  Candidate c of query b, list g: element (g*gs + b*bs + c), c in [0, k).
// ------------------------------------------------------------------------------------------------------------------
template <typename IdxT>
__global__ void __launch_bounds__(256) topk_select_kernel(const float* __restrict__ cs, const IdxT* __restrict__ ci,
                                                          int G, int64_t B, int k, int64_t gs, int64_t bs,
                                                          int64_t idx_base, float* __restrict__ out_s,
                                                          int64_t* __restrict__ out_i) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int M = G * k;
  float prev_s = __builtin_huge_valf();
  int64_t prev_i = -1;  // everything is worse than (+inf, -1)
  for (int r = 0; r < k; ++r) {
    float best_s = RG_NEG_INF;
    int64_t best_i = INT64_MAX;
    for (int c = lane; c < M; c += 64) {
      const int g = c / k, e = c % k;
      const int64_t o = g * gs + b * bs + e;
      const float s = cs[o];
      const int64_t i = (int64_t)ci[o];
      const bool after_prev = (s < prev_s) || (s == prev_s && i > prev_i);
      const bool beats = (s > best_s) || (s == best_s && i < best_i);
      if (after_prev && beats) {
        best_s = s;
        best_i = i;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float os = __shfl_xor(best_s, off);
      const int64_t oi = __shfl_xor(best_i, off);
      if ((os > best_s) || (os == best_s && oi < best_i)) {
        best_s = os;
        best_i = oi;
      }
    }
    if (lane == 0) {
      out_s[b * k + r] = best_s;
      out_i[b * k + r] = best_i + idx_base;
    }
    prev_s = best_s;
    prev_i = best_i;
  }
}

// ---- split planning ----------------------------------------------------------------------------------------------
// Work items = qtiles x nsplit equal-length workgroups, one per CU at a time (LDS-limited).  Pick the split count that
// fills whole rounds of 256 CUs while keeping each stream long enough that the list warm-up (~k ln(n/k) offers per
// query) stays small next to its MFMA time.
struct TopkPlan {
  int nsplit;
  int64_t keys_per_split;
  size_t qn_bytes, part_s_bytes, part_i_bytes;
};

static TopkPlan plan_topk(int64_t B, int64_t N, int D, int k) {
  const int QT = 256, CUS = 256;
  const int stage_keys = 32 * (256 / D);
  const int64_t qtiles = cdiv(B, QT);
  const int64_t nstages = cdiv(N, stage_keys);
  int64_t max_split = nstages < 256 ? nstages : 256;
  const int64_t min_keys = 8192;  // below this the warm-up dominates
  int best = 1;
  double best_cost = 1e300;
  for (int64_t s = 1; s <= max_split; ++s) {
    const int64_t per = cdiv(nstages, s) * stage_keys;
    if (s > 1 && per < min_keys) break;
    const int64_t real = cdiv(N, per);  // splits that actually hold keys
    if (real != s) continue;
    const int64_t wgs = qtiles * s;
    const double rounds = (double)cdiv(wgs, CUS);
    const double warm = 1.0 + 6.0 * (double)k * log((double)per / k + 1.0) / (double)per * 32.0 / 16.0;
    const double cost = rounds * (double)per * warm;
    if (cost < best_cost * 0.999) {
      best_cost = cost;
      best = (int)s;
    }
  }
  TopkPlan pl;
  pl.nsplit = best;
  pl.keys_per_split = cdiv(nstages, best) * stage_keys;
  pl.qn_bytes = align_up((size_t)B * D * sizeof(float), 256);
  pl.part_s_bytes = align_up((size_t)B * best * k * sizeof(float), 256);
  pl.part_i_bytes = align_up((size_t)B * best * k * sizeof(int), 256);
  return pl;
}

template <int D>
static int launch_topk(const TopkParams& p, int64_t qtiles, hipStream_t st) {
  using C = TopkCfg<D>;
  const size_t lds = C::lds_bytes(p.k);
  static bool attr_set = false;  // raising the dynamic-LDS cap is idempotent; racing setters write the same value
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_stream_kernel<D>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("topk_cosine: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
      return RAGRAPH_EDEVICE;
    }
    attr_set = true;
  }
  const int64_t grid = qtiles * p.nsplit;
  hipLaunchKernelGGL(topk_stream_kernel<D>, dim3((unsigned)grid), dim3(C::THREADS), lds, st, p);
  RG_CHECK_LAUNCH("topk_cosine");
  return RAGRAPH_OK;
}

}  // namespace ragraph

using namespace ragraph;

extern "C" size_t ragraph_topk_cosine_workspace_bytes(int64_t B, int64_t N, int D, int k) {
  if (B < 1 || N < 1 || k < 1 || (D != 64 && D != 128 && D != 256)) return 0;
  TopkPlan pl = plan_topk(B, N, D, k);
  return pl.qn_bytes + pl.part_s_bytes + pl.part_i_bytes;
}

extern "C" int ragraph_topk_cosine_f32(const float* Q, int64_t B, const float* Kn, int64_t N, int D, int k,
                                       int64_t idx_base, float* out_scores, int64_t* out_idx, void* ws,
                                       size_t ws_bytes, void* stream) {
  RG_REQUIRE(Q && Kn && out_scores && out_idx && ws, RAGRAPH_EINVAL, "topk_cosine: null pointer");
  RG_REQUIRE(B >= 1 && N >= 1, RAGRAPH_EINVAL, "topk_cosine: B=%lld N=%lld must be >= 1", (long long)B, (long long)N);
  RG_REQUIRE(k >= 1 && k <= N, RAGRAPH_EINVAL, "topk_cosine: k=%d out of range for N=%lld (torch.topk raises too)", k,
             (long long)N);
  RG_REQUIRE(D == 64 || D == 128 || D == 256, RAGRAPH_EUNSUPPORTED, "topk_cosine: D=%d not in {64,128,256}", D);
  RG_REQUIRE(k <= 32, RAGRAPH_EUNSUPPORTED, "topk_cosine: k=%d > 32 not supported by the fused kernel yet", k);
  RG_REQUIRE(N < (int64_t)INT_MAX - 1024, RAGRAPH_EUNSUPPORTED, "topk_cosine: shard rows must fit int32");
  RG_REQUIRE(aligned16(Q) && aligned16(Kn) && aligned16(ws), RAGRAPH_EINVAL, "topk_cosine: Q, Kn, ws must be 16-B aligned");
  TopkPlan pl = plan_topk(B, N, D, k);
  RG_REQUIRE(ws_bytes >= pl.qn_bytes + pl.part_s_bytes + pl.part_i_bytes, RAGRAPH_EWORKSPACE,
             "topk_cosine: workspace %zu < %zu", ws_bytes, pl.qn_bytes + pl.part_s_bytes + pl.part_i_bytes);
  hipStream_t st = as_stream(stream);
  char* w = static_cast<char*>(ws);
  float* Qn = reinterpret_cast<float*>(w);
  float* part_s = reinterpret_cast<float*>(w + pl.qn_bytes);
  int* part_i = reinterpret_cast<int*>(w + pl.qn_bytes + pl.part_s_bytes);

  int rc = ragraph_normalize_rows_f32(Q, B, D, Qn, stream);
  if (rc != RAGRAPH_OK) return rc;

  TopkParams p;
  p.Qn = Qn;
  p.Kn = Kn;
  p.B = B;
  p.N = N;
  p.k = k;
  p.nsplit = pl.nsplit;
  p.keys_per_split = pl.keys_per_split;
  p.part_s = part_s;
  p.part_i = part_i;
  const int64_t qtiles = cdiv(B, 256);
  if (D == 256)
    rc = launch_topk<256>(p, qtiles, st);
  else if (D == 128)
    rc = launch_topk<128>(p, qtiles, st);
  else
    rc = launch_topk<64>(p, qtiles, st);
  if (rc != RAGRAPH_OK) return rc;

  // partial layout [B][nsplit][k]: list g of query b at b*(nsplit*k) + g*k
  const int wpb = 4;
  hipLaunchKernelGGL(topk_select_kernel<int>, dim3((unsigned)cdiv(B, wpb)), dim3(wpb * 64), 0, st, part_s, part_i,
                     pl.nsplit, B, k, (int64_t)k, (int64_t)pl.nsplit * k, idx_base, out_scores, out_idx);
  RG_CHECK_LAUNCH("topk_select");
  return RAGRAPH_OK;
}

extern "C" int ragraph_topk_merge_f32(const float* scores, const int64_t* idx, int G, int64_t B, int k,
                                      float* out_scores, int64_t* out_idx, void* stream) {
  RG_REQUIRE(scores && idx && out_scores && out_idx, RAGRAPH_EINVAL, "topk_merge: null pointer");
  RG_REQUIRE(G >= 1 && B >= 1 && k >= 1, RAGRAPH_EINVAL, "topk_merge: G,B,k must be >= 1");
  RG_REQUIRE((int64_t)G * k <= 4096, RAGRAPH_EUNSUPPORTED, "topk_merge: G*k=%lld > 4096", (long long)G * k);
  const int wpb = 4;
  hipLaunchKernelGGL(topk_select_kernel<int64_t>, dim3((unsigned)cdiv(B, wpb)), dim3(wpb * 64), 0, as_stream(stream),
                     scores, idx, G, B, k, (int64_t)B * k, (int64_t)k, (int64_t)0, out_scores, out_idx);
  RG_CHECK_LAUNCH("topk_merge");
  return RAGRAPH_OK;
}
