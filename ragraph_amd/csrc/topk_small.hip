// The exact cosine top-k of UP TO 32 QUERIES against a large bank in ONE LAUNCH (SimilarityFunctions.py:6-16 +
// ToyGraphBase.py:66-67 at the reference's smallest batch sizes: RAGraph_graph/RAGraph.py:48-60 retrieves ONE query per
// forward).
//
// The multi-launch filtered call (topk_filter.hip) spends, for one query against 1M x 256 keys, 42 us streaming the int8
// copy and as long again on everything around it: prepare 5 us, bound pass 9 us, rescoring 18 us, and ~4.5 us of idle
// chip at each of the four launch boundaries.  Here every workgroup does all of it, and nothing waits for a launch:
//
//   0. prepare (every workgroup, redundantly -- 32 rows): normalised queries (normalize_rows' tree: the same bits), their
//      bf16 / int8 rounding errors and int8 scales, into LDS; each wave builds its MFMA B operands from there.
//   1. bound: the workgroup's share of a bf16 pass over a PREFIX of the bank; per query the best approximate score of each
//      of G = min(4k, 64) parts of the prefix goes to global memory by atomicMax (order-preserving unsigned; 0 = nothing
//      published).  Each part's best key has an exact score >= its approximate one - eps, the parts' best keys are distinct,
//      so the k-th largest of the PUBLISHED maxima, minus eps, is a lower bound theta of the query's final k-th best score
//      -- with all parts published or only some of them (-inf when fewer than k are).
//   2. a BOUNDED wait: the workgroup announces its bound units (a counter) and waits until every workgroup has, or a time
//      limit passes (workgroups of another process may hold the CUs of ours: RAGRAPH_SMALL_WAIT_TICKS).  No grid
//      barrier: whatever is published at that moment is a valid bound; a workgroup that had to move on early refreshes its
//      thresholds from the published maxima while it streams.
//   3. filter: the workgroup's share of ONE pass over the int8 copy (D bytes per key; bf16 when the bank's int8 copy is not
//      accurate enough), the register-fed stream of topk_filter_direct.hip; keys whose integer sum reaches the query's
//      threshold T = floor((theta - eps_bf16 - eps_int8) / (s_q s_k)) - 2 are kept in a wave-private LDS buffer.
//   4. exact rescoring AT THE SOURCE: each wave scores its own candidates with the k = 0..D-1 fmaf chain from +0 (rows
//      fetched cooperatively, 16 at a time), the workgroup reserves list space with ONE atomic per query and stores the
//      exact (score, key) pairs.  There are no candidate lists of approximate hits, hence no sub-lists and no list
//      overflow from the approximate filter; a query whose exact-pair list passes SMALL_LIST_CAP (a bank of near-
//      duplicates) stops passing keys and is answered by an exact scan (6 below).
//   5. the LAST workgroup to finish (a ticket) selects every query's canonical top-k from its pair list (a lane-maxima bound
//      prunes the list to <= 128 pairs, ranks by counting), answers zero queries (scores +0, rows in order), LISTS the
//      overflowed ones and -- under a speculative first bound -- the queries the bound was too high for, writes the call's
//      statistics words, and leaves the state buffer zeroed for the next call.
//   6. behind the kernel, every call: topk_overflow_fixup_kernel (topk_filter.hip) -- exact scans of the listed queries, cut
//      into key slices over the whole chip; the list is empty as a rule and the launch returns at once.  (Until late in
//      round 5 the last workgroup scanned overflowed queries itself: 25 ms a query.)
//
// The result has the bits of ragraph_topk_cosine_f32: every output score is the fp32 chain, the selection the canonical
// one; the approximate phases only decide which keys are scored.
#include "rescore_common.h"
#include <type_traits>

namespace ragraph {

constexpr int SMALL_MAX_B = 32;
// exact (score, key) pairs per query in the workspace.  A workgroup contributes at most its own k best per query from its
// LDS list (pruned before the reservation) plus what a flood appended directly while its bound was still rising -- 256
// workgroups x 32 + slack; beyond the cap (every workgroup moving on without a bound AND passing its whole share, for
// several queries) the query is answered by the exact scan of the fixup launch.
#ifndef RG_SMALL_LIST_CAP
#define RG_SMALL_LIST_CAP 16384
#endif
constexpr int SMALL_LIST_CAP = RG_SMALL_LIST_CAP;
constexpr int SMALL_PRUNE_MIN = 64;      // pairs in a workgroup's LDS list from which it looks for queries to prune
constexpr int SMALL_PARTS_MAX = 256;     // parts of the bound prefix: ONE PER WORKGROUP that has bound units (four per lane of the selecting wave)
constexpr int SMALL_WG_LIST = 1024;      // exact pairs a workgroup collects in LDS before its one reservation per query
constexpr int SMALL_PAIRBUF = 512;       // (key, query) pairs a wave expands at a time
// state buffer (ints; ZERO before the first call, left zero by every call): [1] workgroups done, cnt[q] at 32 + 32 q (a
// 128-byte line each), part maxima [32][256] from 32 + 32 * 32
constexpr int SMALL_STATE_CNT0 = 32;
constexpr int SMALL_STATE_GMAX0 = SMALL_STATE_CNT0 + 32 * SMALL_MAX_B;
constexpr int SMALL_STATE_INTS = SMALL_STATE_GMAX0 + SMALL_MAX_B * SMALL_PARTS_MAX;

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct SmallParams {
  const float* Q;          // [B,D] raw queries
  const float* Kn;         // [N,D] normalised keys (fp32: the exact chains)
  const uint16_t* Kb;      // bf16 copy, fragment order (filter_common.h)
  const signed char* Kb8;  // int8 copy behind it
  const unsigned* max_kerr2;  // bf16 copy's tail: max |dk|^2
  const unsigned* tail8;      // int8 copy's tail: max |dk|^2, s_k of the NORMAL / HEAVY granules (filter_common.h)
  const unsigned* cls8;       // its class bits
  int64_t N, idx_base;
  int B, k;
  int64_t bound_units;     // bound pass: units [0, bound_units) of the bf16 copy ...
  int parts;               // ... dealt over the first `parts` workgroups (= the parts of the bound)
  int64_t nunits;          // filter pass: units of the copy it streams
  unsigned wait_ticks;     // 10 ns ticks a workgroup waits for the others' bound units at most
  int* state;
  int list_cap;            // pairs a query's list may hold: SMALL_LIST_CAP (RAGRAPH_SMALL_LIST_CAP: fewer -- the tests' way to an overflow)
  float* list_s;           // [B][SMALL_LIST_CAP] exact scores ...
  int* list_k;             // ... and keys
  float* out_s;
  int64_t* out_i;
  int* overflow;           // out: queries answered by an exact scan (a list beyond its cap; under a prior: a miss)
  // a SPECULATIVE first bound (this thread's prior, ragraph_topk_cosine_filtered_set_prior): no bound units, no wait, theta =
  // prior for every query; the last workgroup lists the queries whose k-th best pair scores below it for the sliced exact
  // scan of the fixup launch behind this one (qn_out: their normalised rows for it)
  int use_prior;
  float prior;
  float* qn_out;           // [B,D] normalised queries (written under a prior only)
  int* miss_count;         // [1]
  int* miss_list;          // [B]
  int* fix_done;           // [>= B] tickets of the fixup launch: left zero
  int* stats;              // [32] the call's statistics words (include/ragraph_hip.h: [0] magic, [14] queries, [16] speculative,
                           // [17] misses, [18] / [19] smallest / largest final k-th best score)
};

template <int D_>
struct SmallGeo {  // unit geometry of a fragment-order copy whose keys take 2 D_ bytes (the int8 copy of D: D_ = D / 2)
  static constexpr int KS = D_ / 32;                 // MFMA k-steps per sub-tile
  static constexpr int KSTEPS = D_ / 16;             // 1-KiB blocks per 32-key sub-tile
  static constexpr int SUBS = 16 / KSTEPS;           // sub-tiles per 16-KiB unit
  static constexpr int UNIT_KEYS = 32 * SUBS;
};

__device__ __forceinline__ unsigned small_f2u(float f) { return (unsigned)f2ord(f) ^ 0x80000000u; }  // order-preserving, > 0
__device__ __forceinline__ float small_u2f(unsigned u) { return ord2f((int)(u ^ 0x80000000u)); }

template <int D, bool I8>
struct SmallLds {
  static constexpr int QLD = D + 4;  // floats per query row in LDS (16-B pad: a column of 16 rows spreads over the banks)
  static constexpr int CAND_BUF = 512;
  static constexpr size_t qn = 0;
  static constexpr size_t wbuf = qn + (size_t)SMALL_MAX_B * QLD * 4;
  static constexpr size_t pairbuf = wbuf + (size_t)8 * CAND_BUF * 8;
  static constexpr size_t tile = pairbuf + (size_t)8 * SMALL_PAIRBUF * 8;
  static constexpr size_t wg_s = tile + (size_t)8 * 16 * RESCORE_LD * 4;
  static constexpr size_t wg_k = wg_s + (size_t)SMALL_WG_LIST * 4;
  static constexpr size_t wg_q = wg_k + (size_t)SMALL_WG_LIST * 4;
  static constexpr size_t small = wg_q + (size_t)SMALL_WG_LIST * 4;   // per-query scalars
  static constexpr size_t bytes_stream = small + 32 * 4 * 12;
  static constexpr size_t bytes = bytes_stream;
};

// canonical 64-bit key of a pair (rescore_common.h: wave_select's order)
__device__ __forceinline__ unsigned long long small_key64(float s, int key) {
  return ((unsigned long long)select_ord(s) << 32) | (unsigned)~(unsigned)key;
}

// Canonical top-k of a list held as NPL canonical keys per lane (0 = empty): the k-th largest of the 64 lanes' best keys
// bounds the k-th best of all from below, the keys at or above it (a few dozen) are compacted into `surv` and ranked by
// counting (wave_select<2>).  false: more than 128 keys survive (ties) -- the caller takes the chunked path.
template <int NPL>
__device__ __forceinline__ bool small_select_held(const unsigned long long (&held)[NPL], int k, int lane, int64_t idx_base,
                                                  unsigned long long* surv, float* os, int64_t* oi, float* kth_out) {
  unsigned long long best = 0ull;
#pragma unroll
  for (int u = 0; u < NPL; ++u) best = held[u] > best ? held[u] : best;
  int rank = 0;
  for (int o = 0; o < 64; ++o) {
    const unsigned long long x = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(best >> 32), o) << 32) |
                                 (unsigned)__builtin_amdgcn_readlane((int)(unsigned)best, o);
    rank += (x > best || (x == best && o < lane)) ? 1 : 0;
  }
  const unsigned long long kb = __ballot(rank == k - 1);
  const int src = __ffsll((long long)kb) - 1;
  const unsigned long long bound = ((unsigned long long)(unsigned)__shfl((int)(best >> 32), src) << 32) | (unsigned)__shfl((int)(unsigned)best, src);
  int ns = 0;  // wave-uniform
#pragma unroll
  for (int u = 0; u < NPL; ++u) {
    const bool keep = held[u] != 0ull && held[u] >= bound;
    const unsigned long long bal = __ballot(keep);
    const int pos = ns + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
    if (keep && pos < 128) surv[pos] = held[u];
    ns += __popcll(bal);
  }
  if (ns > 128) return false;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float s2[2];
  int id2[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    s2[u] = RG_NEG_INF;
    id2[u] = INT_MAX;
    if (lane + 64 * u < ns) {
      const unsigned long long key = surv[lane + 64 * u];
      s2[u] = select_unord((unsigned)(key >> 32));
      id2[u] = (int)~(unsigned)key;
    }
  }
  __builtin_amdgcn_wave_barrier();
  wave_select<2>(s2, id2, k, lane, idx_base, os, oi, kth_out);
  return true;
}

#ifdef RG_SMALL_TIMING  // diagnostic build: wall-clock stamps (10 ns ticks) of workgroup 0 and of the last workgroup to finish
__device__ unsigned long long g_small_t[2][16];
#define RG_SSTAMP(i_) \
  if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); if (blockIdx.x == 0) g_small_t[0][i_] = now_; st_[i_] = now_; }
#else
#define RG_SSTAMP(i_)
#endif

#define RG_LIKELY(x) __builtin_expect(!!(x), 1)
#define RG_UNLIKELY(x) __builtin_expect(!!(x), 0)
template <int D, bool I8>
__global__ void __launch_bounds__(512, 2) topk_small_kernel(SmallParams p) {
#ifdef RG_SMALL_TIMING
  unsigned long long st_[16] = {};
#endif
  RG_SSTAMP(0);
  using GB = SmallGeo<D>;                 // the bf16 copy (bound pass; the filter pass when !I8)
  using GF = SmallGeo<I8 ? D / 2 : D>;    // the copy the filter pass streams
  using L = SmallLds<D, I8>;
  using acc_t = typename std::conditional<I8, i32x4, f32x4>::type;
  constexpr int QLD = L::QLD;
  extern __shared__ float4 small_smem4[];
  char* smem = reinterpret_cast<char*>(small_smem4);
  float* qn = reinterpret_cast<float*>(smem + L::qn);
  uint2* wbuf_all = reinterpret_cast<uint2*>(smem + L::wbuf);
  int2* pair_all = reinterpret_cast<int2*>(smem + L::pairbuf);
  float* tile_all = reinterpret_cast<float*>(smem + L::tile);
  float* wg_s = reinterpret_cast<float*>(smem + L::wg_s);
  int* wg_k = reinterpret_cast<int*>(smem + L::wg_k);
  int* wg_q = reinterpret_cast<int*>(smem + L::wg_q);
  float* sc_eqb = reinterpret_cast<float*>(smem + L::small);  // [32] |dq| of the bf16 rounding
  float* sc_eq8 = sc_eqb + 32;                                // [32] |dq| of the int8 rounding
  float* sc_qs = sc_eq8 + 32;                                 // [32] int8 scale of the query
  int* sc_flag = reinterpret_cast<int*>(sc_qs + 32);          // [32] 2: zero query
  float* thr_lds = reinterpret_cast<float*>(sc_flag + 32);    // [32] pass thresholds (int8: integer bits)
  float* theta_lds = thr_lds + 32;                            // [32] theta itself: an exact score below it cannot be in the top-k
  int* qcnt = reinterpret_cast<int*>(theta_lds + 32);         // [32] this workgroup's exact pairs per query
  int* qbase = qcnt + 32;                                     // [32] their place in the query's list
  int* misc = qbase + 32;                                     // [0] wg_n, [1] all bound units in, [2] last workgroup, [3] overflowed queries,
                                                              // [4] rounds of a flood (0: none), [8..24) compaction counts

  float* kth_lds = reinterpret_cast<float*>(misc + 32);       // [32] the queries' final k-th best scores (the last workgroup)
  float* thrh_lds = kth_lds + 32;                             // [32] int8: pass thresholds for keys of HEAVY granules (filter_common.h)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int B = p.B, k = p.k;
  const int G = (int)gridDim.x;
  const int ngq = B > 16 ? 2 : 1;  // query groups of 16 (block-uniform)
  int* cnt_g = p.state + SMALL_STATE_CNT0;
  unsigned* gmax_g = reinterpret_cast<unsigned*>(p.state + SMALL_STATE_GMAX0);

  const unsigned lane16 = (unsigned)lane * 16u;
  f32x4 A0[16], A1[16];
#define RG_SLOAD(buf_, base_, u_)                                                                                  \
  {                                                                                                                \
    const char* ub_ = reinterpret_cast<const char*>(base_) + (uint64_t)(u_) * (16 * 1024) + lane16;                \
    _Pragma("unroll") for (int b_ = 0; b_ < 16; ++b_)                                                              \
      buf_[b_] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ub_ + b_ * 1024));                      \
  }
  // (the copies' error rows: requested now, used by the thresholds)
  const unsigned tail_b = *p.max_kerr2, tail_8e = I8 ? p.tail8[0] : 0u, tail_8s = I8 ? p.tail8[1] : 0u;
  const unsigned tail_8eh = I8 ? p.tail8[3] : 0u, tail_8sh = I8 ? p.tail8[4] : 0u;   // (the HEAVY granules' error and scale)
  // The bound pass's units are dealt over the first G_b workgroups -- at least 64 of them when there are that many units
  // (parts = workgroups: the bound wants >= 4 k of them), one unit per wave before a workgroup takes a second round: unit u
  // belongs to workgroup u % G_b, wave (u / G_b) % 8.  This wave's first unit: its loads need no query -- in flight during
  // the prepare phase.
  const int G_b = p.parts;
  const int64_t u_first = (int)blockIdx.x < G_b ? (int64_t)blockIdx.x + (int64_t)G_b * wave : p.bound_units;
  if (u_first < p.bound_units) RG_SLOAD(A0, p.Kb, u_first);

  // ---- 0. prepare: one wave per query row (filter_prep_kernel's arithmetic) ----------------------------------------
  if (tid < 32) {
    qcnt[tid] = 0;
    thr_lds[tid] = I8 ? __int_as_float(tid < p.B ? INT_MIN : INT_MAX) : (tid < p.B ? RG_NEG_INF : __builtin_huge_valf());
    thrh_lds[tid] = thr_lds[tid];
    theta_lds[tid] = RG_NEG_INF;
  }
  if (tid < 32) {
    misc[tid] = 0;
    kth_lds[tid] = RG_NEG_INF;
  }
  constexpr int NCH = D / 4;
  float4 vrow[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {  // (a wave's rows: wave, wave + 8, ... -- all requested before the first is used)
    vrow[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < NCH && wave + 8 * c < B) vrow[c] = reinterpret_cast<const float4*>(p.Q + (int64_t)(wave + 8 * c) * D)[lane];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int q = wave + 8 * c;
    if (q >= 16 * ngq) break;
    float4 v = vrow[c];
    float pp = 0.f;
    pp = fmaf(v.x, v.x, pp);
    pp = fmaf(v.y, v.y, pp);
    pp = fmaf(v.z, v.z, pp);
    pp = fmaf(v.w, v.w, pp);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) pp = __fadd_rn(pp, __shfl_xor(pp, off));
    const float d = fmaxf(sqrtf(pp), 1e-12f);
    v.x = v.x / d; v.y = v.y / d; v.z = v.z / d; v.w = v.w / d;
    if (lane < NCH) *reinterpret_cast<float4*>(qn + q * QLD + 4 * lane) = v;
    if (p.qn_out && blockIdx.x == 0 && q < B && lane < NCH) reinterpret_cast<float4*>(p.qn_out + (int64_t)q * D)[lane] = v;
    const float x[4] = {v.x, v.y, v.z, v.w};
    float e2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dd = x[e] - (float)(__bf16)x[e];
      e2 = fmaf(dd, dd, e2);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) e2 += __shfl_xor(e2, off);
    unsigned am = max(max(__float_as_uint(fabsf(v.x)), __float_as_uint(fabsf(v.y))), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w))));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) am = max(am, (unsigned)__shfl_xor((int)am, off));
    const float sq = __uint_as_float(am) / 127.f;
    float e8 = 0.f;
    if (I8 && sq > 0.f) {
      const float inv = 1.f / sq;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int qi = quantize_i8(x[e], inv);
        const float dd = fmaf(sq, (float)qi, -x[e]);
        e8 = fmaf(dd, dd, e8);
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) e8 += __shfl_xor(e8, off);
    if (lane == 0) {
      sc_eqb[q] = sqrtf(e2) * 1.0000002f;
      sc_eq8[q] = sqrtf(e8) * 1.000001f;
      sc_qs[q] = sq;
      sc_flag[q] = (q < B && am == 0u) ? 2 : 0;
    }
  }
  __syncthreads();
  RG_SSTAMP(1);

  // the bound pass's B operands (bf16): lane j + 16 g of k-step t of group gq = elements 32 t + 8 g .. + 7 of query 16 gq + j
  // (under a speculative first bound there is no bound pass: the int8 filter pass then needs none of this)
  const bool with_bound = !p.use_prior;   // (block-uniform)
  bf16x8 bqb[2 * GB::KS];
  if (with_bound || !I8)
#pragma unroll
  for (int gq = 0; gq < 2; ++gq)
#pragma unroll
    for (int t = 0; t < GB::KS; ++t) {
      const float* src = qn + (16 * gq + j) * QLD + 32 * t + 8 * g;
      const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
      bf16x8 o;
      o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
      o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
      bqb[gq * GB::KS + t] = gq < ngq ? o : bf16x8{};
    }

  // ---- 1. bound: units 8 b + w (+ 8 G i) of the prefix belong to wave w of workgroup b, and workgroup b's units ARE part b:
  // its waves' maxima are combined in LDS and leave as ONE relaxed agent-scope store per query (written through, visible
  // to every XCD; non-zero = published).  No atomic, no counter, no wait on the publishing side: a reader that finds all
  // parts non-zero has the final maxima in the same round trip.  (The first version published by atomicMax, announced
  // itself on a counter after waiting for its atomics, and read the maxima after the counter filled: five agent-scope
  // round trips of 2 - 3 us each between a workgroup's bound unit and its thresholds; now two.) ------------------------
  if (RG_UNLIKELY(with_bound)) {   // (the steady state of a bank runs under its prior: cold blocks are laid out behind the hot path)
    float* wmax = tile_all;                                       // [8][32] the wave's maxima over its units
    if (lane < 32) wmax[wave * 32 + lane] = RG_NEG_INF;
    for (int64_t u = u_first; u < p.bound_units; u += (int64_t)G_b * 8) {
      if (u != u_first) RG_SLOAD(A0, p.Kb, u);                    // (the first unit's loads were issued before the prepare phase)
#pragma unroll
      for (int sub = 0; sub < GB::SUBS; ++sub) {
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          if (gq >= ngq) break;
          f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
          for (int t = 0; t < GB::KS; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h)
              acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A0[(sub * GB::KS + t) * 2 + h]),
                                                               bqb[gq * GB::KS + t], acc[h], 0, 0, 0);
          float m = acc[0][0];
#pragma unroll
          for (int r = 1; r < 4; ++r) m = fmaxf(m, acc[0][r]);
#pragma unroll
          for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[1][r]);
          m = fmaxf(m, __shfl_xor(m, 16));
          m = fmaxf(m, __shfl_xor(m, 32));
          if (g == 0) wmax[wave * 32 + 16 * gq + j] = fmaxf(wmax[wave * 32 + 16 * gq + j], m);
        }
      }
    }
    __syncthreads();
    if (tid < B && (int)blockIdx.x < G_b) {
      float m = wmax[tid];
#pragma unroll
      for (int w = 1; w < 8; ++w) m = fmaxf(m, wmax[w * 32 + tid]);
      __hip_atomic_store(gmax_g + tid * SMALL_PARTS_MAX + blockIdx.x, small_f2u(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  RG_SSTAMP(2);
  // the filter pass's B operands
  i32x4 bq8[I8 ? 2 * GF::KS : 1];
  if constexpr (I8) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
#pragma unroll
      for (int t = 0; t < GF::KS; ++t) {  // elements 64 t + 16 g .. + 15 of query 16 gq + j
        if (gq >= ngq) {  // (block-uniform: up to 16 queries have no second group to build)
          bq8[gq * GF::KS + t] = i32x4{0, 0, 0, 0};
          continue;
        }
        const int q = 16 * gq + j;
        const float* src = qn + q * QLD + 64 * t + 16 * g;
        const float sq = sc_qs[q];
        const float inv = sq > 0.f ? 1.f / sq : 0.f;
        i32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 a = *reinterpret_cast<const float4*>(src + 4 * c);
          o[c] = (int)(((unsigned)quantize_i8(a.x, inv) & 0xFFu) | (((unsigned)quantize_i8(a.y, inv) & 0xFFu) << 8) |
                       (((unsigned)quantize_i8(a.z, inv) & 0xFFu) << 16) | (((unsigned)quantize_i8(a.w, inv) & 0xFFu) << 24));
        }
        bq8[gq * GF::KS + t] = (gq < ngq && sq > 0.f) ? o : i32x4{0, 0, 0, 0};
      }
  }

  // this wave's units of the filter pass: gw, gw + W, ...; the first one's loads fly during the wait
  const int64_t W = (int64_t)G * 8;
  const int64_t gw = (int64_t)blockIdx.x * 8 + wave;
  const int64_t n_mine = gw < p.nunits ? (p.nunits - gw + W - 1) / W : 0;
  const char* fbase = I8 ? reinterpret_cast<const char*>(p.Kb8) : reinterpret_cast<const char*>(p.Kb);
  // (under a speculative first bound the loads could fly from the kernel's first instruction on -- measured: 51.7 -> 53.0 us
  // for one query; sixteen loads in front of the query rows' delay the prepare phase by more than they hide)
  if (n_mine > 0) RG_SLOAD(A0, fbase, gw);
  // (int8) the class word of unit u's granule (granule = u / 2: filter_common.h), requested a unit ahead like the unit's blocks
  auto cls_word = [&](int64_t u) -> unsigned {
    if constexpr (I8) return p.cls8[__builtin_amdgcn_readfirstlane((int)(u >> 6))];
    else return 0u;
  };
  [[maybe_unused]] unsigned cw0 = n_mine > 0 ? cls_word(gw) : 0u, cw1 = 0u, cwP[2] = {0u, 0u};
  // Up to 16 queries (one MFMA query group): the first TWO units of the stream are multiplied right here, while the other
  // workgroups' part maxima are still on their way -- their accumulators wait in registers for the thresholds, their
  // epilogues run behind phase 2.  Two rounds of the stream (2 x 32 MB chip-wide, ~7 us of HBM time) move under the
  // wait instead of behind it.
  const bool pre = !p.use_prior && ngq == 1 && n_mine >= 2;   // (wave-uniform)
  acc_t accP[2][GF::SUBS][2];
  auto unit_mfma_g0 = [&](f32x4 (&A)[16], acc_t (&out)[GF::SUBS][2]) {
#pragma unroll
    for (int sub = 0; sub < GF::SUBS; ++sub) {
      out[sub][0] = out[sub][1] = acc_t{0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < GF::KS; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if constexpr (I8)
            out[sub][h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, A[(sub * GF::KS + t) * 2 + h]), bq8[t],
                                                                out[sub][h], 0, 0, 0);
          else
            out[sub][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[(sub * GF::KS + t) * 2 + h]), bqb[t],
                                                                  out[sub][h], 0, 0, 0);
        }
    }
  };

  // ---- 2. thresholds from the published part maxima, after a BOUNDED wait for the missing ones -----------------------------
  // Cross-workgroup traffic of this kernel is agent-scope ATOMICS only (relaxed atomic stores / loads, which go past the
  // XCD's L2, and read-modify-writes for list space and the ticket): a __threadfence() would write the whole L2 back
  // (buffer_wbl2: 10 - 30 us with 256 workgroups at it -- measured: it was most of a 175-us first version).
  RG_SSTAMP(3);
  RG_SSTAMP(4);
  const float ek_b = sqrtf(__uint_as_float(tail_b));
  const float ek_8 = I8 ? sqrtf(__uint_as_float(tail_8e)) : 0.f;
  const float sk_8 = I8 ? __uint_as_float(tail_8s) : 0.f;
  const float ek_8h = I8 ? sqrtf(__uint_as_float(tail_8eh)) : 0.f;
  const float sk_8h = I8 ? __uint_as_float(tail_8sh) : 0.f;
  // threshold of query q from the part maxima published so far (one wave; every lane returns with thr_lds[q] written)
  // this lane's parts of query q as published so far: parts lane, lane + 64, ... (0: nothing yet); the largest of them is the
  // maximum of a coarser part (64 of them: one per lane); *missing = some part of this lane is still unpublished
  auto issue_parts = [&](int q, unsigned (&w4)[4]) {   // the loads only: consumed by parts_of
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
      w4[c4] = (q < B && lane + 64 * c4 < G_b)
                   ? __hip_atomic_load(gmax_g + q * SMALL_PARTS_MAX + lane + 64 * c4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                   : 1u;
  };
  auto parts_of = [&](int q, const unsigned (&w4)[4], bool* missing) -> unsigned {
    unsigned u = 0u;
    bool miss = false;
    if (q < B) {
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        miss = miss || w4[c4] == 0u;
        if (lane + 64 * c4 < G_b) u = w4[c4] > u ? w4[c4] : u;
      }
    }
    *missing = miss;
    return u;
  };
  auto load_part = [&](int q, bool* missing) -> unsigned {
    unsigned w4[4];
    issue_parts(q, w4);
    return parts_of(q, w4, missing);
  };
  const int nparts = G_b < 64 ? G_b : 64;
  // theta = a PROVEN lower bound of query q's final k-th best exact score (k distinct keys score at least that): the pass
  // threshold of the filter follows from it.  One lane.  Every value ever written is a valid bound and the words only
  // rise, so concurrent writers (a wave's refresh, a flooded wave's ratchet) at worst leave the lower of two valid bounds.
  auto raise_theta = [&](int q, float theta) {
    theta_lds[q] = fmaxf(theta_lds[q], theta);
    float out;
    if (sc_flag[q]) {
      out = I8 ? __int_as_float(INT_MAX) : __builtin_huge_valf();   // a zero query passes nothing (answered at the end)
      thrh_lds[q] = out;
    } else if constexpr (I8) {
      // (never below a threshold already in force: a list that passed its cap has raised it to INT_MAX)
      const int t = filter_threshold_i8_of(theta, sc_eq8[q], ek_8, sc_qs[q] * sk_8);
      const int old = __float_as_int(thr_lds[q]);
      out = __int_as_float(t > old ? t : old);
      const int th = filter_threshold_i8_of(theta, sc_eq8[q], ek_8h, sc_qs[q] * sk_8h);   // keys of HEAVY granules
      const int oldh = __float_as_int(thrh_lds[q]);
      thrh_lds[q] = __int_as_float(th > oldh ? th : oldh);
    } else {
      const float e = sc_eqb[q];
      const float eps_b = fmaf(fmaf(e, ek_b, e + ek_b), 1.0009765625f, FILTER_EPS_SLACK);
      const float t = __fsub_rn(theta, eps_b);
      out = fmaxf(t, thr_lds[q]);
    }
    thr_lds[q] = out;
  };
  auto make_threshold = [&](int q, unsigned u) {
    const float e = sc_eqb[q];
    const float eps_b = fmaf(fmaf(e, ek_b, e + ek_b), 1.0009765625f, FILTER_EPS_SLACK);
    float v = RG_NEG_INF;
    if (u != 0u) v = __fsub_rn(small_u2f(u), eps_b);
    int rank = 0;
    for (int o = 0; o < nparts; ++o) {  // (o is wave-uniform: v_readlane, not a ds_bpermute round trip per part)
      const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), o));
      rank += (x > v || (x == v && o < lane)) ? 1 : 0;
    }
    const unsigned long long kth = __ballot(lane < nparts && rank == k - 1);
    const float theta = kth ? __shfl(v, __ffsll((long long)kth) - 1) : RG_NEG_INF;
    if (lane == 0) raise_theta(q, theta);
  };
  if (RG_LIKELY(p.use_prior)) {  // a speculative first bound: every query starts from it, nothing to wait for
    if (lane < 4 && wave + 8 * lane < B) raise_theta(wave + 8 * lane, p.prior);   // (a wave's queries: wave, wave + 8, ...)
  } else {  // a wave's queries: wave, wave + 8, ... -- all their part maxima are requested in one batch, again until none is missing
     // or the time limit has passed (workgroups of another process may hold the CUs some of ours still need)
    unsigned up[4];
    const unsigned long long t0 = wall_clock64();
    bool missing;
    {
      // the first request of the part maxima is ISSUED here and consumed behind the two stream units that are multiplied
      // ahead of their thresholds: an agent-scope load is 3 - 7 us under the stream's load, which is what this phase cost
      unsigned raw[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) issue_parts(wave + 8 * c, raw[c]);
      if (pre) {
        RG_SLOAD(A1, fbase, gw + W);
        cwP[0] = cw0;
        cwP[1] = cls_word(gw + W);
        unit_mfma_g0(A0, accP[0]);
        if (n_mine > 2) {
          RG_SLOAD(A0, fbase, gw + 2 * W);
          cw0 = cls_word(gw + 2 * W);
        }
        unit_mfma_g0(A1, accP[1]);
      }
      bool mc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) up[c] = parts_of(wave + 8 * c, raw[c], &mc[c]);
      missing = __any(mc[0] || mc[1] || mc[2] || mc[3]);
    }
    while (missing && (unsigned)(wall_clock64() - t0) < p.wait_ticks) {
      bool mc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) up[c] = load_part(wave + 8 * c, &mc[c]);
      missing = __any(mc[0] || mc[1] || mc[2] || mc[3]);
    }
    if (missing && lane == 0) atomicOr(misc + 1, 1);   // somebody moved on early: wave 0 refreshes while it streams
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (wave + 8 * c < B) make_threshold(wave + 8 * c, up[c]);
  }
  __syncthreads();
  RG_SSTAMP(5);
  bool refresh = wave == 0 && misc[1] != 0;   // (wave-uniform) some workgroup's bound units were still missing

  // ---- 3. filter pass ---------------------------------------------------------------------------------------------
  uint2* wbuf = wbuf_all + wave * L::CAND_BUF;
  int wcnt = 0;  // wave-uniform
  // exact score of (query q, key) by this lane alone (the slow paths: a flood of candidates in mid-stream)
  auto lane_score = [&](int q, int key) {
    const float4* kr = reinterpret_cast<const float4*>(p.Kn + (int64_t)key * D);
    const float4* qr = reinterpret_cast<const float4*>(qn + q * QLD);
    float acc = 0.f;
#pragma unroll 8
    for (int d4 = 0; d4 < D / 4; ++d4) {
      const float4 kv = kr[d4], qv = qr[d4];
      acc = fmaf(qv.x, kv.x, acc);
      acc = fmaf(qv.y, kv.y, acc);
      acc = fmaf(qv.z, kv.z, acc);
      acc = fmaf(qv.w, kv.w, acc);
    }
    return acc;
  };
  auto append_global = [&](int q, int key, float s) {  // one atomic per pair: only where a workgroup's LDS list is full
    const int pos = atomicAdd(cnt_g + 32 * q, 1);
    if (pos < p.list_cap) {
      __hip_atomic_store(p.list_s + (int64_t)q * SMALL_LIST_CAP + pos, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.list_k + (int64_t)q * SMALL_LIST_CAP + pos, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      thr_lds[q] = I8 ? __int_as_float(INT_MAX) : __builtin_huge_valf();   // the exact scan answers this query: pass nothing more
      thrh_lds[q] = thr_lds[q];
    }
  };
  // mid-stream flush (the buffer filled up: a flood): every lane scores its entries' keys itself
  // A flood (a bound that is still -inf because this workgroup moved on early, or thousands of near-duplicates of a query)
  // must not fill the query's list: after every chunk of 64 entries the wave RATCHETS the bound -- among the lanes whose
  // entry belongs to query q0, the k-th largest of the lanes' best exact scores is the score of k distinct keys, hence a
  // valid theta -- and the raised pass threshold ends the flood at its source (each ratchet lifts theta to about the 85th
  // percentile of what still passes: a few chunks per decade).
  auto flush_slow = [&]() {
    for (int i0 = 0; i0 < wcnt; i0 += 64) {
      const int i = i0 + lane;
      float best = RG_NEG_INF;
      int q = -1;
      if (i < wcnt) {
        const uint2 e = wbuf[i];
        q = (int)((e.y >> 8) & 0xFFu);
        unsigned mk = e.y & 0xFFu;
        while (mk) {
          const int r = __ffs(mk) - 1;
          mk &= mk - 1;
          const int key = (int)e.x + (r & 3) + 16 * (r >> 2);
          const float s = lane_score(q, key);
          best = fmaxf(best, s);
          if (s >= theta_lds[q]) append_global(q, key, s);
        }
      }
      unsigned long long todo = __ballot(q >= 0);
      while (todo) {  // (wave-uniform: one round per query present in the chunk)
        const int q0 = __shfl(q, __ffsll((long long)todo) - 1);
        const unsigned long long part = __ballot(q == q0);
        todo &= ~part;
        if (__popcll(part) < k) continue;
        int rank = 0;
        for (int o = 0; o < 64; ++o) {
          if (!((part >> o) & 1ull)) continue;
          const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best), o));
          rank += (x > best || (x == best && o < lane)) ? 1 : 0;
        }
        const unsigned long long kth = __ballot(q == q0 && rank == k - 1);
        if (kth && lane == __ffsll((long long)kth) - 1) raise_theta(q0, best);
      }
    }
    wcnt = 0;
  };
  [[maybe_unused]] int cls_unit = 0;   // (int8) class of the granule the unit in process lies in (wave-uniform; granule = unit / 2)
  auto epilogue = [&](const acc_t (&a)[2], int gq, int64_t unit, int sub) {
    const int64_t key_base = (unit * GF::SUBS + sub) * 32 + 4 * g;  // the lane's keys: + r + 16 h  (mask bit 4 h + r)
    unsigned mk = 0;
    bool any;
    if constexpr (I8) {
      int m = a[0][0];
#pragma unroll
      for (int r = 1; r < 4; ++r) m = max(m, a[0][r]);
#pragma unroll
      for (int r = 0; r < 4; ++r) m = max(m, a[1][r]);
      const int th = __float_as_int((cls_unit ? thrh_lds : thr_lds)[16 * gq + j]);
      any = __any(m >= th);
      if (any) {
        const unsigned tm1 = (unsigned)(max(-(1 << 24), min(1 << 24, th)) - 1);
#pragma unroll
        for (int b = 7; b >= 0; --b) mk = __builtin_amdgcn_alignbit(mk, tm1 - (unsigned)a[b >> 2][b & 3], 31);
        if (th == INT_MAX) mk = 0;  // (the clamp would let |I| >= 2^24 through: impossible, but the sentinel must pass nothing)
      }
    } else {
      float m = a[0][0];
#pragma unroll
      for (int r = 1; r < 4; ++r) m = fmaxf(m, a[0][r]);
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, a[1][r]);
      const float th = thr_lds[16 * gq + j];
      any = __any(m >= th);
      if (any) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) mk |= (a[h][r] >= th) ? (1u << (4 * h + r)) : 0u;
      }
    }
    if (!any) return;
    if (key_base + 32 > p.N) {
      unsigned vm = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) vm |= (key_base + (r & 3) + 16 * (r >> 2) < p.N) ? (1u << r) : 0u;
      mk &= vm;
    }
    const unsigned long long bal = __ballot(mk != 0);
    if (bal) {
      const int pos = wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
      if (mk) wbuf[pos] = make_uint2((unsigned)key_base, mk | ((unsigned)(16 * gq + j) << 8));
      wcnt += __popcll(bal);
    }
  };
  auto set_class = [&]([[maybe_unused]] unsigned cw, [[maybe_unused]] int64_t unit) {
    if constexpr (I8) cls_unit = (int)((cw >> ((unit >> 1) & 31)) & 1u);
  };
  auto process = [&](f32x4 (&A)[16], int64_t unit, unsigned cw) {
    set_class(cw, unit);
    if (wcnt > L::CAND_BUF - 128 * GF::SUBS) flush_slow();  // two groups x SUBS sub-tiles x <= 64 entries
#pragma unroll
    for (int sub = 0; sub < GF::SUBS; ++sub) {
      acc_t acc[2][2];  // [group][half]
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) acc[gq][0] = acc[gq][1] = acc_t{0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < GF::KS; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int gq = 0; gq < 2; ++gq) {
            if (gq >= ngq) continue;
            if constexpr (I8)
              acc[gq][h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, A[(sub * GF::KS + t) * 2 + h]),
                                                                 bq8[gq * GF::KS + t], acc[gq][h], 0, 0, 0);
            else
              acc[gq][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[(sub * GF::KS + t) * 2 + h]),
                                                                   bqb[gq * GF::KS + t], acc[gq][h], 0, 0, 0);
          }
#pragma unroll
      for (int gq = 0; gq < 2; ++gq)
        if (gq < ngq) epilogue(acc[gq], gq, unit, sub);
    }
  };
  if (n_mine > 0) {
    int64_t i = 0;
    if (pre) {  // the two units multiplied before the thresholds existed
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        set_class(cwP[u], gw + u * W);
#pragma unroll
        for (int sub = 0; sub < GF::SUBS; ++sub) epilogue(accP[u][sub], 0, gw + u * W, sub);
      }
      i = 2;
    }
    for (; i + 2 <= n_mine; i += 2) {  // pairs: A0 then A1, the other buffer's loads always in flight
      RG_SLOAD(A1, fbase, gw + (i + 1) * W);
      cw1 = cls_word(gw + (i + 1) * W);
      process(A0, gw + i * W, cw0);
      if (i + 2 < n_mine) {
        RG_SLOAD(A0, fbase, gw + (i + 2) * W);
        cw0 = cls_word(gw + (i + 2) * W);
      }
      process(A1, gw + (i + 1) * W, cw1);
      if (refresh && (i & 6) == 6) {   // (wave 0 of a workgroup that moved on early) the others' maxima may have arrived
        bool any_missing = false;
        for (int q = 0; q < B; ++q) {
          bool mq;
          const unsigned uq = load_part(q, &mq);
          any_missing = any_missing || __any(mq);
          make_threshold(q, uq);
        }
        refresh = any_missing;
      }
    }
    if (i < n_mine) process(A0, gw + i * W, cw0);  // odd count: the last unit, nothing behind it
  }
#undef RG_SLOAD

  RG_SSTAMP(6);
  // ---- 4. exact scores of this wave's candidates: (key, query) pairs, 16 rows per memory round trip ------------------
  // entries [e_lo, e_hi) of this wave's buffer (at most 64 of them: the pair buffer holds 512 pairs)
  auto score_entries = [&](int e_lo, int e_hi) {
    int2* pairs = pair_all + wave * SMALL_PAIRBUF;
    float* tile = tile_all + wave * 16 * RESCORE_LD;
    const int i = e_lo + lane;
    uint2 e = make_uint2(0u, 0u);
    if (i < e_hi) e = wbuf[i];
    unsigned mk = e.y & 0xFFu;
    const int q = (int)((e.y >> 8) & 0xFFu);
    int incl = __popc(mk);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    const int total = __shfl(incl, 63);  // <= 512 = SMALL_PAIRBUF
    int at = incl - __popc(mk);
    while (mk) {
      const int r = __ffs(mk) - 1;
      mk &= mk - 1;
      pairs[at++] = make_int2((int)e.x + (r & 3) + 16 * (r >> 2), q);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int r0 = 0; r0 < total; r0 += 16) {
      int key = -1, pq = 0;
      if (lane < 16 && r0 + lane < total) {
        const int2 pr = pairs[r0 + lane];
        key = pr.x;
        pq = pr.y;
      }
      const float s = coop_scores_few<D>(reinterpret_cast<const float4*>(qn + pq * QLD), p.Kn, key, lane, tile);
      // theta is a proven lower bound of the query's final k-th best EXACT score: a key that scores below it is out
      // (the approximate pass let it through on its error bound; ~3 of 4 candidates end here)
      if (key >= 0 && s < theta_lds[pq]) key = -1;
      const unsigned long long have = __ballot(key >= 0);
      int base = 0;
      if (lane == 0) base = atomicAdd(misc, __popcll(have));
      base = __shfl(base, 0);
      if (key >= 0) {
        const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(have >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)have, 0u));
        if (slot < SMALL_WG_LIST) {
          wg_s[slot] = s;
          wg_k[slot] = key;
          wg_q[slot] = pq;
        } else {
          append_global(pq, key, s);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // (the pair buffer is rewritten by the next chunk)
  };
  // Only a workgroup's k best pairs of a query can be among the query's winners.  prune_list marks every pair of the LDS
  // list below the workgroup's own k-th best of its query (wg_q = -1) and -- that k-th best being the exact score of k
  // distinct keys -- raises the query's theta to it.  All eight waves call it (wave w: queries w, w + 8, ...), between
  // barriers; n = pairs in the list.
  auto prune_list = [&](int n) {
    for (int q = wave; q < B; q += 8) {
      unsigned long long held[SMALL_WG_LIST / 64];
      int cq = 0;
#pragma unroll
      for (int u = 0; u < SMALL_WG_LIST / 64; ++u) {
        const int i = lane + 64 * u;
        held[u] = (i < n && wg_q[i] == q) ? small_key64(wg_s[i], wg_k[i]) : 0ull;
        cq += __popcll(__ballot(held[u] != 0ull));
      }
      if (cq < k) continue;   // (wave-uniform)
      unsigned long long bound = ~0ull;   // the previous round's maximum: keys are distinct, each round finds the next one
      for (int r_ = 0; r_ < k; ++r_) {
        unsigned long long best = 0ull;
#pragma unroll
        for (int u = 0; u < SMALL_WG_LIST / 64; ++u) best = (held[u] < bound && held[u] > best) ? held[u] : best;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const unsigned long long o = ((unsigned long long)(unsigned)__shfl_xor((int)(best >> 32), off) << 32) |
                                       (unsigned)__shfl_xor((int)(unsigned)best, off);
          best = o > best ? o : best;
        }
        bound = best;
      }
#pragma unroll
      for (int u = 0; u < SMALL_WG_LIST / 64; ++u)
        if (held[u] != 0ull && held[u] < bound) wg_q[lane + 64 * u] = -1;   // below this workgroup's own k-th best
      if (lane == 0) raise_theta(q, select_unord((unsigned)(bound >> 32)));
    }
  };
  {
    // A wave whose pairs fit its eighth of the LDS list (every ordinary call: a few pairs per workgroup) scores them at
    // once, without waiting for anybody.  A wave with more -- a FLOOD: a workgroup that moved on without a bound passes its
    // whole share, a query next to thousands of near-duplicates -- announces its rounds of 8 entries (<= 64 pairs a wave,
    // 512 a round); behind the barrier every wave joins the rounds: the list is pruned to the workgroup's k best per query
    // and compacted, theta raised, then the flooding waves score their next 8 entries -- so nothing spills into the
    // queries' global lists and the flood thins out at once.
    int np = 0;
    for (int i = lane; i < wcnt; i += 64) np += __popc(wbuf[i].y & 0xFFu);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) np += __shfl_xor(np, off);
    const bool flooding = np > SMALL_WG_LIST / 8;   // (wave-uniform)
    if (RG_UNLIKELY(flooding)) {
      if (lane == 0) atomicMax(misc + 4, (wcnt + 7) / 8);
    } else {
      for (int i0 = 0; i0 < wcnt; i0 += 64) score_entries(i0, i0 + 64 < wcnt ? i0 + 64 : wcnt);
    }
    __syncthreads();
    const int nrounds = misc[4];   // (block-uniform; 0: nobody floods)
    if (RG_UNLIKELY(nrounds > 0))
    for (int rd = 0; rd < nrounds; ++rd) {
      const int n = misc[0] < SMALL_WG_LIST ? misc[0] : SMALL_WG_LIST;
      if (n >= SMALL_PRUNE_MIN) {   // (block-uniform)
        prune_list(n);
        __syncthreads();
        // compaction: every thread holds its two entries, then all are written back densely
        float cs_[2];
        int ck_[2], cq_[2];
        bool keep[2];
        int* wcount = misc + 8;   // [16] kept entries per (slot c, wave)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int i = tid + 512 * c;
          keep[c] = i < n && wg_q[i] >= 0;
          cs_[c] = keep[c] ? wg_s[i] : 0.f;
          ck_[c] = keep[c] ? wg_k[i] : 0;
          cq_[c] = keep[c] ? wg_q[i] : 0;
          const unsigned long long bal = __ballot(keep[c]);
          if (lane == 0) wcount[8 * c + wave] = __popcll(bal);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          int base = 0;
          for (int t = 0; t < 8 * c + wave; ++t) base += wcount[t];
          const unsigned long long bal = __ballot(keep[c]);
          const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          if (keep[c]) {
            wg_s[pos] = cs_[c];
            wg_k[pos] = ck_[c];
            wg_q[pos] = cq_[c];
          }
        }
        if (tid == 0) {
          int tot = 0;
          for (int t = 0; t < 16; ++t) tot += wcount[t];
          misc[0] = tot;
        }
        __syncthreads();
      }
      if (flooding && 8 * rd < wcnt) score_entries(8 * rd, 8 * rd + 8 < wcnt ? 8 * rd + 8 : wcnt);
      __syncthreads();
    }
  }
  __syncthreads();
  RG_SSTAMP(7);
  {  // one reservation per query for the workgroup's pairs
    const int n = misc[0] < SMALL_WG_LIST ? misc[0] : SMALL_WG_LIST;
    if (RG_UNLIKELY(n >= SMALL_PRUNE_MIN)) {  // (block-uniform) many pairs: only this workgroup's k best of a query can be among the winners
      prune_list(n);
      __syncthreads();
    }
    int r[SMALL_WG_LIST / 512];
#pragma unroll
    for (int c = 0; c < SMALL_WG_LIST / 512; ++c) {
      const int i = tid + 512 * c;
      r[c] = (i < n && wg_q[i] >= 0) ? atomicAdd(qcnt + wg_q[i], 1) : 0;
    }
    __syncthreads();
    if (tid < B && qcnt[tid] > 0) qbase[tid] = atomicAdd(cnt_g + 32 * tid, qcnt[tid]);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < SMALL_WG_LIST / 512; ++c) {
      const int i = tid + 512 * c;
      if (i < n && wg_q[i] >= 0) {
        const int q = wg_q[i];
        const int pos = qbase[q] + r[c];
        if (pos < p.list_cap) {  // (agent-scope stores: written through, visible to the last workgroup without a fence)
          __hip_atomic_store(p.list_s + (int64_t)q * SMALL_LIST_CAP + pos, wg_s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(p.list_k + (int64_t)q * SMALL_LIST_CAP + pos, wg_k[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }

  // ---- 5. the last workgroup selects ----------------------------------------------------------------------------------
  RG_SSTAMP(8);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's list stores and reservations are acknowledged
  __syncthreads();
  RG_SSTAMP(9);
  if (tid == 0) misc[2] = __hip_atomic_fetch_add(p.state + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1;
  __syncthreads();
  if (!misc[2]) return;
  RG_SSTAMP(10);
  // per-wave staging for the selection (the stream's buffers are free now): 512 canonical keys + winners
  unsigned long long* surv = reinterpret_cast<unsigned long long*>(pair_all + wave * SMALL_PAIRBUF);
  // a wave's queries: wave, wave + 8, ...; their counts and first 256 pairs are requested before the first is used
  int nq[4];
  float pf_s[4][4];
  int pf_k[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int q = wave + 8 * c;
    nq[c] = q < B ? __hip_atomic_load(cnt_g + 32 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      pf_s[c][u] = RG_NEG_INF;
      pf_k[c][u] = INT_MAX;
      if (q < B) {  // (slots beyond the count hold stale pairs: masked below)
        pf_s[c][u] = __hip_atomic_load(p.list_s + (int64_t)q * SMALL_LIST_CAP + lane + 64 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pf_k[c][u] = __hip_atomic_load(p.list_k + (int64_t)q * SMALL_LIST_CAP + lane + 64 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
#ifdef RG_SMALL_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RG_SSTAMP(13);
#endif
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int q = wave + 8 * c;
    if (q >= B) break;
#ifdef RG_SMALL_TIMING
    if (c == 1) { RG_SSTAMP(14); }
#endif
    float* os = p.out_s + (int64_t)q * k;
    int64_t* oi = p.out_i + (int64_t)q * k;
    if (sc_flag[q]) {  // zero query: every score +0, canonical order = row order
      if (lane < k) {
        os[lane] = 0.f;
        oi[lane] = p.idx_base + lane;
      }
      continue;
    }
    const int n = nq[c];
    if (n > p.list_cap) {
      if (lane == 0) atomicOr(qcnt + q, 1 << 30);  // (marks the query for the scan below; qcnt is free now)
      continue;
    }
    const float* ls = p.list_s + (int64_t)q * SMALL_LIST_CAP;
    const int* lk = p.list_k + (int64_t)q * SMALL_LIST_CAP;
    auto load_pair = [&](int i, float& s, int& id) {  // (other workgroups' stores: past this CU's caches)
      s = RG_NEG_INF;
      id = INT_MAX;
      if (i < n) {
        s = __hip_atomic_load(ls + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        id = __hip_atomic_load(lk + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };
    if (n <= 128) {
      float s2[2];
      int id2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bool have = lane + 64 * u < n;
        s2[u] = have ? pf_s[c][u] : RG_NEG_INF;
        id2[u] = have ? pf_k[c][u] : INT_MAX;
      }
      wave_select<2>(s2, id2, k, lane, p.idx_base, os, oi, kth_lds + q);
      continue;
    }
    if (n <= 256) {
      // (k rounds of a wave-wide maximum -- wave_select<4> -- measured slower than the bound + ranking by counting at
      // k = 10: 18 vs 11.6 us for the two queries of a wave)
      unsigned long long held[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) held[u] = lane + 64 * u < n ? small_key64(pf_s[c][u], pf_k[c][u]) : 0ull;
      if (small_select_held<4>(held, k, lane, p.idx_base, surv, os, oi, kth_lds + q)) continue;
    } else if (n <= 1024) {  // the list in registers: sixteen pairs per lane, all loads in flight at once
      unsigned long long held[16];
      float s16[16];
      int id16[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) load_pair(lane + 64 * u, s16[u], id16[u]);
#pragma unroll
      for (int u = 0; u < 16; ++u) held[u] = id16[u] == INT_MAX ? 0ull : small_key64(s16[u], id16[u]);
      if (small_select_held<16>(held, k, lane, p.idx_base, surv, os, oi, kth_lds + q)) continue;
    }
    // longer lists (and ties by the hundred: a bank of duplicates the collapsing did not see): chunks of 1024 pairs -- sixteen
    // per lane, all loads in flight -- against the running winners, which ride in a seventeenth slot between chunks (LDS)
    {
      float* ws_ = reinterpret_cast<float*>(surv);                 // [32] running scores
      int64_t* wi_ = reinterpret_cast<int64_t*>(surv + 32);        // [32] running local ids
      for (int i0 = 0; i0 < n; i0 += 1024) {
        float s17[17];
        int id17[17];
#pragma unroll
        for (int u = 0; u < 16; ++u) load_pair(i0 + lane + 64 * u, s17[u], id17[u]);
        s17[16] = RG_NEG_INF;
        id17[16] = INT_MAX;
        if (i0 > 0 && lane < k) {
          s17[16] = ws_[lane];
          id17[16] = wi_[lane] >= INT_MAX ? INT_MAX : (int)wi_[lane];
        }
        __builtin_amdgcn_wave_barrier();
        if (i0 + 1024 >= n) {
          wave_select<17>(s17, id17, k, lane, p.idx_base, os, oi, kth_lds + q);
        } else {
          wave_select<17>(s17, id17, k, lane, 0, ws_, wi_);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
    }
  }
  __syncthreads();
  RG_SSTAMP(11);
  // overflowed queries (a list beyond its cap: floods of several queries at once -- rare): which ones, by one ballot per wave over
  // the queries' counters (block-uniform).  They are LISTED with the prior's misses for the sliced exact scan of the fixup
  // launch behind this kernel (<= 2 ms a query; until round 5 this workgroup scanned the bank itself: 25 ms a query).
  const int n_over = __popc((unsigned)__ballot(lane < B && (qcnt[lane < 32 ? lane : 0] >> 30) != 0));
  // the call's statistics words and -- under a prior -- the proof: a query's answer is exact iff its k-th best pair scores at
  // least the prior (every key scoring at least that passed the filter: csrc/topk_filter.hip, filter_verify_prior_kernel);
  // the others are listed, in query order, for the fixup launch behind this one.  (Zero queries were answered in place;
  // a query whose list overflowed is listed whatever it scores.)
  __syncthreads();
  if (wave == 0) {  // lane q judges query q; the wave combines (misses in query order by ballot + prefix count)
    const int q = lane < 32 ? lane : 0;
    const bool live = lane < B && !sc_flag[q];
    const float kth = kth_lds[q];
    const bool scanned = (qcnt[q] >> 30) != 0;
    const bool miss = live && p.use_prior && !scanned && !(kth >= p.prior);
    const bool listed = miss || (live && scanned);   // (a list beyond its cap: also the fixup launch's)
    const unsigned long long mm = __ballot(listed);
    const int n_miss = __popcll(__ballot(miss)), n_listed = __popcll(mm);
    if (listed) p.miss_list[(int)__builtin_amdgcn_mbcnt_hi((unsigned)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mm, 0u))] = q;
    const bool counts = live && !listed && kth > RG_NEG_INF;
    int lo = counts ? f2ord(kth) : INT_MAX, hi = counts ? f2ord(kth) : INT_MIN;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
      lo = min(lo, __shfl_xor(lo, off));
      hi = max(hi, __shfl_xor(hi, off));
    }
    if (p.stats && lane < 32) {
      int v = 0;
      if (lane == 0) v = 0x52414753;
      else if (lane == 1) v = 1;
      else if (lane == 14) v = B;
      else if (lane == 16) v = p.use_prior;
      else if (lane == 17) v = n_miss;
      else if (lane == 18) v = lo;
      else if (lane == 19) v = hi;
      else if (lane == 20) v = n_over + n_miss;
      p.stats[lane] = v;
    }
    if (lane == 0) {
      if (p.miss_count) *p.miss_count = n_listed;
      *p.overflow = n_over + n_miss;
    }
  }
  if (p.fix_done && tid < B) p.fix_done[tid] = 0;
  // leave the state zeroed for the next call (every other workgroup has finished: the ticket said so)
  for (int i = tid; i < SMALL_STATE_INTS; i += 512) p.state[i] = 0;
  RG_SSTAMP(12);
#ifdef RG_SMALL_TIMING
  if (tid == 0)
    for (int i = 0; i < 16; ++i) g_small_t[1][i] = st_[i];
#endif
}

static int small_prefix_keys(int B, bool i8) {
  // keys of the bound pass's prefix (with G = 4 k parts the bound is worth the exact k-th best of ~0.85 of it): one query's
  // call streams 16 K keys of the bf16 copy before its pass over the int8 copy; more queries buy a sharper bound with a
  // longer prefix (their candidates cost B times as much).  The int8 bound is ~5x wider: twice the prefix.
  // (9 .. 16 queries: 24 576 x 2 keys = 1536 units = one per wave of 192 workgroups -- 16 queries 0.081 -> 0.076 ms against
  // 32 768, 9: 0.073 -> 0.070; 20 480 / 28 672: 0.077 - 0.078)
  int base = B <= 1 ? 8192 : (B <= 8 ? 16384 : (B <= 16 ? 24576 : 32768));
  if (const char* e = getenv("RAGRAPH_SMALL_PREFIX_BASE")) {  // (experiments; read per call)
    const int v = atoi(e);
    if (v >= 4096) base = v;
  }
  return i8 ? 2 * base : base;
}

template <int D, bool I8>
static int launch_small(const SmallParams& p, int grid, hipStream_t st) {
  using L = SmallLds<D, I8>;
  static DeviceOnce lds_once;
  if (hipError_t e = raise_dynamic_lds(lds_once, &topk_small_kernel<D, I8>, 160 * 1024); e != hipSuccess) {
    set_error("topk_cosine_small: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    return RAGRAPH_EDEVICE;
  }
  hipLaunchKernelGGL((topk_small_kernel<D, I8>), dim3((unsigned)grid), dim3(512), L::bytes, st, p);
  RG_CHECK_LAUNCH("topk_cosine_small");
#ifdef RG_SMALL_TIMING
  {
    (void)hipDeviceSynchronize();
    unsigned long long t[2][16];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_small_t), sizeof(t));
    static const char* names[13] = {"", "prepare", "bound", "operands", "wait", "thresholds", "stream", "rescoring", "store", "fence", "ticket", "select", "scan+reset"};
    for (int b = 0; b < 2; ++b) {
      fprintf(stderr, "[small timing, %s workgroup of %d, 10 ns ticks]", b ? "last" : "first", grid);
      for (int i = 1; i < 13; ++i)
        if (t[b][i] && t[b][i - 1]) fprintf(stderr, " %s %lld", names[i], (long long)(t[b][i] - t[b][i - 1]));
      fprintf(stderr, " | entered %lld after workgroup 0, total %lld\n", (long long)(t[b][0] - t[0][0]), (long long)(t[b][b ? 12 : 9] - t[b][0]));
    }
    fprintf(stderr, "[small timing] last workgroup, wave 0: selection loads %lld, first query %lld ticks\n", (long long)(t[1][13] - t[1][10]),
            t[1][14] ? (long long)(t[1][14] - t[1][13]) : (long long)(t[1][11] - t[1][13]));
  }
#endif
  return RAGRAPH_OK;
}

}  // namespace ragraph

using namespace ragraph;

extern "C" int ragraph_topk_cosine_small_ok(int64_t B, int64_t N, int D, int k) {
  return B >= 1 && B <= SMALL_MAX_B && (D == 64 || D == 128 || D == 256) && k >= 1 && k <= 32 && N >= 65536 && N < ((int64_t)1 << 31);
}
extern "C" size_t ragraph_topk_cosine_small_state_bytes(void) { return (size_t)SMALL_STATE_INTS * sizeof(int); }
// workspace: the queries' pair lists | normalised queries [B,D] | miss count + list | tickets and partial winners of the fixup
// launch (sized for its slicing: up to 32 listed queries x 16 slices x 32) | the call's 32 statistics words (the LAST 128 bytes)
struct SmallWs {
  size_t lists, qn, miss, done, part_s, part_i, stats, total;
};
static SmallWs small_ws(int64_t B, int D) {
  SmallWs w{};
  size_t at = 0;
  auto take = [&](size_t n) { const size_t a = at; at += align_up(n, 256); return a; };
  w.lists = take((size_t)B * SMALL_LIST_CAP * (sizeof(float) + sizeof(int)));
  w.qn = take((size_t)B * D * sizeof(float));
  w.miss = take((size_t)(1 + SMALL_MAX_B) * sizeof(int));
  w.done = take((size_t)1024 * sizeof(int));
  w.part_s = take((size_t)SMALL_MAX_B * 16 * 32 * sizeof(float));
  w.part_i = take((size_t)SMALL_MAX_B * 16 * 32 * sizeof(int64_t));
  w.stats = at;      // (exactly the last 128 bytes: callers find the words there)
  at += 128;
  w.total = at;
  return w;
}
extern "C" size_t ragraph_topk_cosine_small_workspace_bytes(int64_t B, int D, int k) {
  if (B < 1 || B > SMALL_MAX_B) return 0;
  return small_ws(B, D).total;
}

static bool small_uses_i8(int D) {
  static const bool i8_env = [] { const char* e = getenv("RAGRAPH_FILTER_I8_DIRECT"); return !e || atoi(e) != 0; }();  // A/B
  return (D == 128 || D == 256) && i8_env && filter_thread_i8_cap() != 0;
}
// keys of the bf16 prefix a call's bound pass reads; *i8 = 1 when its filter pass streams the int8 copy (under this thread's cap)
extern "C" int64_t ragraph_topk_cosine_small_prefix_keys(int64_t B, int64_t N, int D, int* i8) {
  if (!ragraph_topk_cosine_small_ok(B, N, D, 1)) return 0;
  const bool q8 = small_uses_i8(D);
  if (i8) *i8 = q8 ? 1 : 0;
  int64_t prefix = small_prefix_keys((int)B, q8);
  if (prefix > N / 4) prefix = N / 4;
  const int unit_b = D == 256 ? 32 : (D == 128 ? 64 : 128);
  return prefix / unit_b * unit_b;
}

extern "C" int ragraph_topk_cosine_small_f32(const float* Q, int64_t B, const float* Kn, const uint16_t* Kb, int64_t N, int D, int k,
                                             int64_t idx_base, float* out_scores, int64_t* out_idx, int* overflow, int* state,
                                             void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(Q && Kn && Kb && out_scores && out_idx && overflow && state && ws, RAGRAPH_EINVAL, "topk_cosine_small: null pointer");
  RG_REQUIRE(ragraph_topk_cosine_small_ok(B, N, D, k), RAGRAPH_EUNSUPPORTED,
             "topk_cosine_small: B=%lld N=%lld D=%d k=%d outside 1..32 queries, N >= 65536, D in {64,128,256}, k <= 32", (long long)B,
             (long long)N, D, k);
  RG_REQUIRE(aligned16(Q) && aligned16(Kn) && aligned16(Kb), RAGRAPH_EINVAL, "topk_cosine_small: Q / Kn / Kb must be 16-byte aligned");
  RG_REQUIRE(ws_bytes >= ragraph_topk_cosine_small_workspace_bytes(B, D, k), RAGRAPH_EWORKSPACE, "topk_cosine_small: workspace too small");
  const char* wt = getenv("RAGRAPH_SMALL_WAIT_TICKS");  // (read per call: the tests force the move-on-early path with 0)
  // (2 ms: a cold start -- code pages, TLB -- can delay single workgroups by tens of microseconds, and a workgroup that
  // moves on without the others' maxima passes every key of its share; the limit only has to keep a call from hanging
  // when another process holds the CUs our later workgroups need)
  const unsigned wait_ticks = wt ? (unsigned)atoll(wt) : 200000u;
  const bool i8 = small_uses_i8(D);
  const int64_t npad = (N + FILTER_PAD_KEYS - 1) / FILTER_PAD_KEYS * FILTER_PAD_KEYS;
  SmallParams p{};
  p.Q = Q;
  p.Kn = Kn;
  p.Kb = Kb;
  p.max_kerr2 = reinterpret_cast<const unsigned*>(Kb + npad * D);
  const FilterI8View v8 = filter_i8_view(Kb, N, D);
  p.Kb8 = v8.K8;
  p.tail8 = v8.tail8;
  p.cls8 = v8.cls;
  p.N = N;
  p.idx_base = idx_base;
  p.B = (int)B;
  p.k = k;
  const int unit_b = D == 256 ? 32 : (D == 128 ? 64 : 128);            // keys per 16-KiB unit of the bf16 copy
  const int unit_f = i8 ? 2 * unit_b : unit_b;                          // ... of the copy the filter pass streams
  int64_t prefix = small_prefix_keys((int)B, i8);
  if (prefix > N / 4) prefix = N / 4;
  p.bound_units = prefix / unit_b;
  // (parts = the workgroups the units are dealt over, min(units, 64 .. grid): at least k of them)
  RG_REQUIRE(p.bound_units >= 2 * k, RAGRAPH_EUNSUPPORTED, "topk_cosine_small: the bank is too short for a bound pass of k = %d parts", k);
  const float prior = filter_thread_prior();
  const bool spec = prior == prior && prior > -2.f && prior < 2.f;
  const SmallWs wsl = small_ws(B, D);
  char* wb = static_cast<char*>(ws);
  p.use_prior = spec ? 1 : 0;
  p.prior = spec ? prior : 0.f;
  p.qn_out = reinterpret_cast<float*>(wb + wsl.qn);   // (the normalised rows, for the fixup launch's scans)
  p.miss_count = reinterpret_cast<int*>(wb + wsl.miss);
  p.miss_list = p.miss_count + 1;
  p.fix_done = reinterpret_cast<int*>(wb + wsl.done);
  p.stats = reinterpret_cast<int*>(wb + wsl.stats);
  if (spec) p.bound_units = 0;   // (no bound pass: no units, no parts, nothing published or awaited)
  p.parts = 0;   // (set below, once the grid is known)
  p.nunits = cdiv(N, (int64_t)unit_f);   // (the copies are padded to whole units: 256 keys)
  p.wait_ticks = wait_ticks;
  p.state = state;
  p.list_s = reinterpret_cast<float*>(ws);   // (SmallWs::lists = 0)
  p.list_k = reinterpret_cast<int*>(p.list_s + (size_t)B * SMALL_LIST_CAP);
  p.list_cap = SMALL_LIST_CAP;
  if (const char* e = getenv("RAGRAPH_SMALL_LIST_CAP")) {   // (test hook, read per call: lists that overflow on an ordinary bank)
    const int v = atoi(e);
    if (v >= 1 && v < SMALL_LIST_CAP) p.list_cap = v;
  }
  p.out_s = out_scores;
  p.out_i = out_idx;
  p.overflow = overflow;
  const int cus = device_cus_multiple_of_8();
  int64_t grid = cdiv(p.nunits, (int64_t)8);
  if (grid > cus) grid = cus;
  {
    int64_t gb = p.bound_units < 64 ? p.bound_units : (cdiv(p.bound_units, (int64_t)8) < 64 ? (int64_t)64 : cdiv(p.bound_units, (int64_t)8));
    p.parts = (int)(gb < grid ? gb : grid);
  }
  if (spec) p.parts = 0;
  RG_REQUIRE(spec || p.parts >= k, RAGRAPH_EUNSUPPORTED, "topk_cosine_small: %d workgroups cannot make a bound of k = %d parts", p.parts, k);
  hipStream_t st = as_stream(stream);
  int rc;
  if (D == 256) rc = i8 ? launch_small<256, true>(p, (int)grid, st) : launch_small<256, false>(p, (int)grid, st);
  else if (D == 128) rc = i8 ? launch_small<128, true>(p, (int)grid, st) : launch_small<128, false>(p, (int)grid, st);
  else rc = launch_small<64, false>(p, (int)grid, st);
  if (rc != RAGRAPH_OK) return rc;
  // the queries the prior was too high for and those whose lists overflowed (none, as a rule: the launch returns at once):
  // exact scans in key slices
  return launch_overflow_fixup(D, p.qn_out, Kn, N, k, idx_base, p.miss_count, p.miss_list, out_scores, out_idx, p.fix_done,
                               reinterpret_cast<float*>(wb + wsl.part_s), reinterpret_cast<int64_t*>(wb + wsl.part_i), B, stream);
}
