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
#include <stdlib.h>

#include "common.h"
#include "segment_plan.h"

namespace ragraph {

// Wave-cooperative insert into a SORTED k-list (k <= 64) that lives in LDS as [k][QS]: lane p holds entry p, one
// ballot finds the insert position, one shuffle shifts the tail.  All 64 lanes must call it with wave-uniform
// (q, s, idx).  Returns the new k-th best score.
__device__ __forceinline__ float list_insert_coop(float* ls, int* li, int q, int k, int QS, float s, int idx,
                                                  int lane) {
  const bool mine = lane < k;
  float es = mine ? ls[lane * QS + q] : 0.f;
  int ei = mine ? li[lane * QS + q] : 0;
  // entries that stay ahead of the candidate: a prefix of the sorted list
  const unsigned long long ahead = __ballot(mine && cand_better(es, ei, s, idx));
  const int pos = __popcll(ahead);
  const float us = __shfl_up(es, 1);
  const int ui = __shfl_up(ei, 1);
  if (pos < k) {  // wave-uniform
    if (lane == pos) {
      es = s;
      ei = idx;
    } else if (lane > pos) {
      es = us;
      ei = ui;
    }
    if (mine && lane >= pos) {
      ls[lane * QS + q] = es;
      li[lane * QS + q] = ei;
    }
  }
  return __shfl(es, k - 1);
}

// Lane-private insert into the sorted k-list at `ls/li` (this lane's query): shift from the tail.  Used in the DENSE
// regime (first tiles of a stream, when most lanes hold candidates): 32 queries insert in parallel, ~1k cycles per
// round of up to 64 candidates, where the cooperative insert would take them one at a time.
__device__ __forceinline__ void list_insert_lane(float* ls, int* li, int k, int stride, float s, int idx) {
  if (!cand_better(s, idx, ls[(k - 1) * stride], li[(k - 1) * stride])) return;
  int p = k - 1;
  while (p > 0) {
    const float ps = ls[(p - 1) * stride];
    const int pi = li[(p - 1) * stride];
    if (!cand_better(s, idx, ps, pi)) break;
    ls[p * stride] = ps;
    li[p * stride] = pi;
    --p;
  }
  ls[p * stride] = s;
  li[p * stride] = idx;
}

struct TopkParams {
  const float* Qn;   // [B,D] normalised queries (big kernel) / RAW queries (small-batch kernel)
  const float* Kn;   // [N,D] normalised keys
  const float* Kp;   // [N,D] the same rows packed [even k | odd k] (ragraph_pack_keys_f32) for the LDS-DMA ring, or NULL
  int64_t B, N;
  int k;
  int nsplit;
  int ngroups;             // small/mid-batch kernel: groups of 16 queries (workgroup b serves group b % ngroups)
  int64_t qtiles;          // query tiles of 256 (big kernel)
  int xcd_map;             // 1: XCD-aware block mapping (enough query tiles to give every XCD its own)
  int wgs_per_group;       // big kernel: persistent workgroups per group (one group per XCD when xcd_map, else one group)
  int lb_min;              // big kernel: shortest share of the leftover tiles a workgroup takes (stages)
  int warm_stages;         // big kernel: cost of one list warm-up in stages (planning only)
  int depth[2];            // big kernel: lockstep steps of the plan for groups with as many tiles as group 0 / one fewer
  int64_t nstages_total;   // big kernel: stages in one full pass over the bank
  float* part_s;           // [B][nsplit][k]
  int* part_i;
  const float* thr_init;   // small-batch kernel: per-query lower bound of the k-th best score at [q*k + k-1], or NULL
  int ablate;              // DIAGNOSTIC ONLY (env RAGRAPH_TOPK_ABLATE, results invalid when non-zero): bit0 skip the
                           // top-k epilogue, bit1 skip global loads + LDS writes, bit2 skip the stage barrier (both
                           // 2-slot variant); DMA ring: bit4 skip the DMA, bit5 skip the FULL / FREE counters
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
  // RING = 2: double buffer + one workgroup barrier per stage.  RING = 3: three slots handed over through FULL / FREE
  // counters in LDS, no barrier: a wave may run up to one stage ahead of or behind the others, so a wave that is busy
  // inserting candidates does not stall the other seven (the barrier made every stage as slow as its slowest wave).
  // RING = 4 (D = 256, packed bank): four slots filled by LDS-DMA (global_load_lds_dwordx4), same counters; the fourth
  // slot pays for the later FULL signal (a DMA is signalled one stage after it was issued) so the one-stage slack stays.
  static size_t lds_bytes(int k, int ring) {
    return sizeof(float) * ((size_t)ring * STAGE_FLOATS) + (size_t)k * QT * 8 + (ring >= 3 ? 32 : 0);
  }
};

__device__ __forceinline__ void ring_wait(unsigned* ctr, unsigned target) {  // wave-uniform spin on an LDS counter
  while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void ring_signal(unsigned* ctr, int lane) {  // after this wave's LDS accesses are complete
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

typedef __attribute__((address_space(3))) void lds_void;        // operand types of __builtin_amdgcn_global_load_lds
typedef __attribute__((address_space(1))) const void gbl_void;

#ifdef RG_TOPK_TIMING  // diagnostic build only (tools/dev): per-wave cycle totals of the DMA ring's phases
__device__ unsigned long long g_topk_timing[8];
#define RG_T(var_) const unsigned long long var_ = __builtin_amdgcn_s_memtime()
#else
#define RG_T(var_)
#endif

template <int D, int RING>
__global__ void __launch_bounds__(512, 2) topk_stream_kernel(TopkParams p) {
  using C = TopkCfg<D>;
  // ONE __shared__ object; everything below is an offset from it so every access stays a ds_* instruction.
  extern __shared__ float4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  constexpr int OFF_LS = RING * C::STAGE_FLOATS;  // [QT][k] scores, then [QT][k] indices: one SORTED list per query
  float* ls = smem + OFF_LS;
  int* li = reinterpret_cast<int*>(smem + OFF_LS + p.k * C::QT);
  unsigned* full = reinterpret_cast<unsigned*>(smem + OFF_LS + 2 * p.k * C::QT);  // [RING] wave-writes per slot
  unsigned* freec = full + 4;                                                       // [RING] wave-reads-done per slot

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // known-uniform: row / list bases stay in SGPRs
  const int j = lane & 31, h = lane >> 5;
  const int k = p.k;

  // ---- persistent workgroups, one per CU (LDS-limited), each walking a short list of SEGMENTS ------------------
  // A segment = (query tile, contiguous stage range of the bank) with its own sorted lists and one partial-list slot.
  // Workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the blocks that share an XCD and its 4 MiB L2), so
  // with xcd_map the 32 workgroups of XCD x form a group that owns query tiles x, x+8, ...; otherwise all workgroups
  // form one group owning every tile.  segment_plan.h hands every workgroup of a group the same number of stages, in
  // as few segments as that allows, and lines the segments up so that workgroups running side by side read the same
  // stages (a stage then comes from HBM once per group; the other 31 of 32 fetches hit that L2).  Placement only
  // affects speed: any plan that covers every (tile, stage) once gives the same result.
  const int x = p.xcd_map ? (int)(blockIdx.x & 7) : 0;
  const int64_t nq = p.xcd_map ? ((p.qtiles - x + 7) >> 3) : p.qtiles;  // query tiles owned by this group
  const int64_t NS = p.nstages_total;
  const int64_t nq0 = p.xcd_map ? ((p.qtiles + 7) >> 3) : p.qtiles;     // group 0's count; the others have nq0 or nq0-1
  SegmentWalker walker(nq, NS, p.wgs_per_group, p.lb_min, p.warm_stages, p.depth[nq != nq0],
                       p.xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
  Segment seg;
  while (walker.next(seg)) {
  const int64_t ql = seg.tile, st0 = seg.st0, st1 = seg.st1;
  const int split = seg.slot;  // this segment's slot among the tile's partial lists
  const int64_t qtile = p.xcd_map ? x + 8 * ql : ql;
  const int64_t q0 = qtile * C::QT;
  const int64_t n_begin = st0 * C::STAGE_KEYS;
  const int64_t n_end = min(p.N, st1 * C::STAGE_KEYS);
  const int nstages = (int)(st1 - st0);

  // ---- B operand: this lane's query (row q0 + wave*32 + j), k-slots h, h+2, h+4, ... ------------------------
  float breg[D / 2];
  {
    int64_t q = q0 + wave * 32 + j;
    if (q > p.B - 1) q = p.B - 1;  // clamp: results of padded queries are never written
    const float4* qp = reinterpret_cast<const float4*>(p.Qn + q * D);
    // in batches of 16 loads: hipcc otherwise issues all D/4 loads at once and spills long-lived lane constants around
    // that 256-VGPR peak, with the reloads landing inside the stage loop
    constexpr int QB = 16;
#pragma unroll
    for (int c0 = 0; c0 < D / 4; c0 += QB) {
#pragma unroll
      for (int c = c0; c < c0 + QB; ++c) {
        const float4 v = qp[c];
        breg[2 * c] = h ? v.y : v.x;
        breg[2 * c + 1] = h ? v.w : v.z;
      }
      // pin each batch's selects behind its loads: 16 float4 of temporaries instead of 64
#pragma unroll
      for (int c = c0; c < c0 + QB; ++c) asm volatile("" : "+v"(breg[2 * c]), "+v"(breg[2 * c + 1]));
      asm volatile("" ::: "memory");
    }
  }

  for (int i = tid; i < k * C::QT; i += C::THREADS) {
    ls[i] = RG_NEG_INF;
    li[i] = RG_IDX_NONE;
  }

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
  // LDS image: row = [even k (D/2 floats) | odd k (D/2 floats)] + 16-B pad.  Lane (j,h) of the MFMA then reads the
  // float4 {k = 2m+h : m = 4c..4c+3} of key j with ONE ds_read_b128 and no select; the de-interleave costs two 8-byte
  // stores per staged float4 instead of one 16-byte store (store traffic is 1/8 of the read traffic).
#define RG_ST2(dst_, a_, b_)                                                                      \
  do { (dst_)[0] = (a_); (dst_)[1] = (b_); } while (0) /* two dwords -> ds_write2_b32, no register shuffling */
#define RG_STAGE_WRITE(buf_)                                                                      \
  do {                                                                                            \
    float* d_ = smem + (buf_) * C::STAGE_FLOATS + srow * C::ROW + (scol >> 1);                    \
    RG_ST2(d_, pre0.x, pre0.z);                                                                   \
    RG_ST2(d_ + D / 2, pre0.y, pre0.w);                                                           \
    RG_ST2(d_ + SROWS * C::ROW, pre1.x, pre1.z);                                                  \
    RG_ST2(d_ + SROWS * C::ROW + D / 2, pre1.y, pre1.w);                                          \
    RG_ST2(d_ + 2 * SROWS * C::ROW, pre2.x, pre2.z);                                              \
    RG_ST2(d_ + 2 * SROWS * C::ROW + D / 2, pre2.y, pre2.w);                                      \
    RG_ST2(d_ + 3 * SROWS * C::ROW, pre3.x, pre3.z);                                              \
    RG_ST2(d_ + 3 * SROWS * C::ROW + D / 2, pre3.y, pre3.w);                                      \
  } while (0)

  float thr = RG_NEG_INF;
#ifdef RG_TOPK_TIMING
  unsigned long long tw_epi = 0;
#endif

  // one stage of MFMAs + top-k epilogue for this wave's 32 queries
  auto compute_stage = [&](int s, int cur) {
#pragma unroll 1
    for (int t = 0; t < C::TILES; ++t) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* arow = smem + cur + (t * 32 + j) * C::ROW + h * (D / 2);
#pragma unroll
      for (int c = 0; c < D / 8; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(arow + 4 * c);  // key[j][2m+h], m = 4c..4c+3
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, breg[4 * c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, breg[4 * c + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, breg[4 * c + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, breg[4 * c + 3], acc, 0, 0, 0);
      }

      // ---- epilogue: acc[r] = score(key row (r&3) + 8*(r>>2) + 4*h of the tile, query j) ------------------
      RG_T(te0);
      float m = acc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
      if (p.ablate & 1) {
        asm volatile("" ::"v"(m));
        m = RG_NEG_INF;
      }
      if (__any(m >= thr)) {
        // rare path (~k ln(n/k) times per query over the stream): the wave inserts its candidates one at a time with
        // all 64 lanes cooperating on each insert (sorted list: one ballot for the position, one shuffle for the shift).
        const int key_base = (int)(n_begin + (int64_t)s * C::STAGE_KEYS + t * 32) + 4 * h;
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int idx = key_base + (r & 3) + 8 * (r >> 2);
          if (acc[r] >= thr && idx < (int)n_end) mask |= 1u << r;
        }
        unsigned long long pend = __ballot(mask != 0);
        if (__popcll(pend) > 8) {
          // DENSE regime (start of a stream): every lane inserts into its own query's list; the two half-wave lanes of
          // a query take turns.  Afterwards both re-read the list's k-th score as their threshold.
          float* qls = ls + (wave * 32 + j) * k;
          int* qli = li + (wave * 32 + j) * k;
          while (pend) {
            const int r0 = __ffs(mask) - 1;
            float my_sc = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) my_sc = (r0 == r) ? acc[r] : my_sc;
            const int idx = key_base + (r0 & 3) + 8 * (r0 >> 2);
#pragma unroll 1
            for (int hh = 0; hh < 2; ++hh) {
              if (mask != 0 && h == hh) list_insert_lane(qls, qli, k, 1, my_sc, idx);
            }
            mask &= mask - 1;
            thr = qls[k - 1];
            if (mask) {
#pragma unroll
              for (int r = 0; r < 16; ++r)
                if ((mask >> r & 1u) && acc[r] < thr) mask &= ~(1u << r);
            }
            pend = __ballot(mask != 0);
          }
        }
        while (pend) {
          const int src = __ffsll((long long)pend) - 1;  // wave-uniform: lowest lane with a candidate
          const int r0 = __ffs(mask) - 1;                // lowest pending register of THIS lane (used on lane src)
          float my_sc = acc[0];
#pragma unroll
          for (int r = 1; r < 16; ++r) my_sc = (r0 == r) ? acc[r] : my_sc;
          const float sc = __shfl(my_sc, src);
          const int idx = __shfl(key_base + (r0 & 3) + 8 * (r0 >> 2), src);
          const int qs = src & 31;
          const float nthr = list_insert_coop(ls + (wave * 32 + qs) * k, li + (wave * 32 + qs) * k, 0, k, 1, sc, idx,
                                              lane);
          if (j == qs) thr = nthr;  // both half-wave lanes of that query
          if (lane == src) mask &= mask - 1;
          // drop remaining candidates the tightened threshold already excludes (only lanes of query qs can change)
          if (j == qs && mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if ((mask >> r & 1u) && acc[r] < thr) mask &= ~(1u << r);
          }
          pend = __ballot(mask != 0);
        }
      }
#ifdef RG_TOPK_TIMING
      tw_epi += __builtin_amdgcn_s_memtime() - te0;
#endif
    }
  };

  if constexpr (RING == 2) {
    RG_STAGE_LOAD(0);
    RG_STAGE_WRITE(0);
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
      const bool more = (s + 1 < nstages);
      RG_STAGE_LOAD(more ? s + 1 : s);  // in flight under this stage's MFMAs
      __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks the loads below the MFMA block (to save VGPRs)
      compute_stage(s, (s & 1) * C::STAGE_FLOATS);
      if (more && !(p.ablate & 2)) RG_STAGE_WRITE((s + 1) & 1);
      if (!(p.ablate & 4)) __syncthreads();
    }
  } else if constexpr (RING == 4) {
    // LDS-DMA ring over the PACKED bank: a stage row is 1 KiB contiguous in HBM and in LDS, so ONE
    // global_load_lds_dwordx4 per row (wave-uniform LDS base, lane l -> bytes 16 l) fills it with no VGPRs and no
    // ds_write.  Wave w owns rows 4w..4w+3 of every stage.  Stage s+3 is issued at the end of iteration s into the slot
    // stage s-1 lived in, retired (s_waitcnt vmcnt(0)) and signalled FULL at the end of iteration s+1, read in s+3.
    static_assert(D == 256, "one wave-instruction must cover exactly one padded LDS row");
    const unsigned lds_base = (unsigned)(size_t)(lds_void*)smem;  // LDS byte address of the dynamic segment
    const unsigned lane16 = 16u * lane;  // the lane's byte offset inside a 1 KiB row, in HBM and in LDS
    auto dma_stage = [&](int s_, int slot_) {
      const int64_t key0_ = n_begin + (int64_t)s_ * C::STAGE_KEYS + wave * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t r_ = min(key0_ + i, p.N - 1);  // tail: duplicate the last key, masked by index later
        const float* g_ = p.Kp + r_ * D;              // wave-uniform row base (SGPR pair) + lane16
        // Issued as asm, not __builtin_amdgcn_global_load_lds: hipcc makes every later ds_read wait vmcnt(0) for a DMA
        // it knows about (possible alias), which would expose the DMA latency at the top of each stage.  The waits for
        // these loads are the explicit s_waitcnt vmcnt(0) statements below.
        const unsigned dst_ = lds_base + 4u * (unsigned)(slot_ * C::STAGE_FLOATS + (wave * 4 + i) * C::ROW);
        unsigned keep_;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
            : "=&s"(keep_)
            : "v"(lane16), "s"(dst_), "s"(g_)
            : "memory");
      }
    };
    if (tid < 8) full[tid] = 0;
    const int pro = nstages < 3 ? nstages : 3;
    for (int s = 0; s < pro; ++s) dma_stage(s, s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < pro) full[tid] = C::WAVES;
    __syncthreads();
    int pending = -1;
    const bool no_dma = p.ablate & 16, no_flags = p.ablate & 32;  // timing-only diagnostics (results invalid)
#ifdef RG_TOPK_TIMING
    unsigned long long tw_full = 0, tw_comp = 0, tw_sig = 0, tw_free = 0, tw_dma = 0;
#endif
    for (int s = 0; s < nstages; ++s) {
      const int slot = s & 3, gen = s >> 2;
      RG_T(t0);
      if (!no_flags) ring_wait(full + slot, (unsigned)(C::WAVES * (gen + 1)));  // all 8 waves' rows of stage s have landed
      RG_T(t1);
      compute_stage(s, slot * C::STAGE_FLOATS);
      RG_T(t2);
      if (!no_flags) ring_signal(freec + slot, lane);                // this wave is done reading stage s
      if (pending >= 0) {                                            // the DMA issued one iteration ago
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!no_flags) ring_signal(full + pending, lane);
        pending = -1;
      }
      RG_T(t3);
#ifdef RG_TOPK_TIMING
      unsigned long long t4 = t3, t5 = t3;
#endif
      if (s + 3 < nstages) {
        const int ws = (s + 3) & 3;                                  // the slot stage s-1 lived in
        if (!no_flags) ring_wait(freec + ws, (unsigned)(C::WAVES * ((s + 3) >> 2)));  // all 8 waves are done reading stage s-1
#ifdef RG_TOPK_TIMING
        t4 = __builtin_amdgcn_s_memtime();
#endif
        if (!no_dma) dma_stage(s + 3, ws);
        pending = ws;
#ifdef RG_TOPK_TIMING
        t5 = __builtin_amdgcn_s_memtime();
#endif
      }
#ifdef RG_TOPK_TIMING
      tw_full += t1 - t0; tw_comp += t2 - t1; tw_sig += t3 - t2; tw_free += t4 - t3; tw_dma += t5 - t4;
#endif
    }
#ifdef RG_TOPK_TIMING
    if (lane == 0) {
      atomicAdd(&g_topk_timing[0], tw_full); atomicAdd(&g_topk_timing[1], tw_comp); atomicAdd(&g_topk_timing[2], tw_sig);
      atomicAdd(&g_topk_timing[3], tw_free); atomicAdd(&g_topk_timing[4], tw_dma); atomicAdd(&g_topk_timing[5], (unsigned long long)nstages);
      atomicAdd(&g_topk_timing[6], tw_epi);
    }
#endif
    __syncthreads();
  } else {
    // slots 0 and 1 are filled up front; afterwards stage s+2 is loaded during stage s and written behind it
    if (tid < 8) full[tid] = 0;
    RG_STAGE_LOAD(0);
    RG_STAGE_WRITE(0);
    if (nstages > 1) {
      RG_STAGE_LOAD(1);
      RG_STAGE_WRITE(1);
    }
    __syncthreads();
    if (tid == 0) {
      full[0] = C::WAVES;
      full[1] = C::WAVES;
    }
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
      const int slot = s % 3, gen = s / 3;
      const bool ahead = (s + 2 < nstages);
      RG_STAGE_LOAD(ahead ? s + 2 : s);  // two stages ahead, in flight under this stage's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      ring_wait(full + slot, (unsigned)(C::WAVES * (gen + 1)));      // all 8 waves have written stage s
      compute_stage(s, slot * C::STAGE_FLOATS);
      ring_signal(freec + slot, lane);                               // this wave is done reading stage s
      if (ahead) {
        const int ws = (s + 2) % 3;                                  // the slot stage s-1 lived in
        ring_wait(freec + ws, (unsigned)(C::WAVES * ((s + 2) / 3))); // all 8 waves are done reading stage s-1
        RG_STAGE_WRITE(ws);
        ring_signal(full + ws, lane);
      }
    }
    __syncthreads();
  }

#undef RG_STAGE_LOAD
#undef RG_STAGE_WRITE
#undef RG_ST2

  // ---- write this segment's sorted candidates; empty list entries stay (-inf, IDX_NONE) ---------------------
  for (int i = tid; i < k * C::QT; i += C::THREADS) {
    const int q = i / k, pos = i % k;
    const int64_t qg = q0 + q;
    if (qg < p.B) {
      const int64_t o = (qg * p.nsplit + split) * k + pos;
      p.part_s[o] = ls[i];
      p.part_i[o] = li[i];
    }
  }
  if (seg.last) {  // the tile's last segment: blank the slots this tile does not use
    const int unused = (p.nsplit - 1 - split) * k;
    for (int i = tid; i < unused * C::QT; i += C::THREADS) {
      const int q = i / unused, pos = i % unused;
      const int64_t qg = q0 + q;
      if (qg < p.B) {
        const int64_t o = (qg * p.nsplit + split + 1) * k + pos;
        p.part_s[o] = RG_NEG_INF;
        p.part_i[o] = RG_IDX_NONE;
      }
    }
  }
  __syncthreads();  // the lists and flags are re-initialised by the next segment
  }  // segments
}

// ------------------------------------------------------------------------------------------------------------------
// Small-batch variant (B <= 16): graph classification sends ONE query per forward (RAGraph_graph/RAGraph.py:50), so
// the pass over the bank is pure HBM streaming (arithmetic intensity B/2 flop/B).  Same numerics as the big kernel
// (natural-order fmaf chain) on v_mfma_f32_16x16x4_f32: keys are the A operand (16 rows), the <= 16 queries sit in
// D/4 VGPRs as the B operand, a lane ends up with 4 scores of ONE query (col = lane & 15, rows 4*(lane>>4)+r).
//   * every WAVE streams its own 16 KiB tiles (tile t of wave w: t = w, w + #waves, ...): 16 global_load_dwordx4 per
//     lane, each wave-instruction one full 1 KiB row at D=256; the next tile's loads are issued right after the
//     current tile is written to the wave-private LDS image, so each wave keeps 16 KiB in flight and a CU 128 KiB
//     (Little: ~25 KiB/CU needed at 10 B/clk/CU and ~2.5k cycles loaded latency);
//   * LDS is only the transpose (row-major rows in, MFMA k-slots out): row stride D+2 floats makes the ds_read_b32 of
//     lane (j, s) at [j][4m+s] hit 32 distinct banks per half-wave; no workgroup barrier in the loop;
//   * MFMA needs 64 x 32 cycles per 16 KiB per SIMD = 4x the HBM rate, so the kernel stays HBM-bound up to B = 16.
// ------------------------------------------------------------------------------------------------------------------
template <int D>
struct SmallCfg {
  static constexpr int WAVES = 8;
  static constexpr int SUB = 256 / D;              // 16-key MFMA sub-tiles per 16 KiB tile
  static constexpr int TILE_KEYS = 16 * SUB;
  static constexpr int ROW = D + 2;                // floats
  static constexpr int TILE_FLOATS = TILE_KEYS * ROW;
  static constexpr int RPI = 256 / D;              // rows covered by one wave-wide float4 load instruction
  static size_t lds_bytes(int k) { return (size_t)WAVES * (sizeof(float) * TILE_FLOATS + (size_t)k * 16 * 8); }
};

template <int D>
__global__ void __launch_bounds__(512, 2) topk_smallb_kernel(TopkParams p) {
  using C = SmallCfg<D>;
  extern __shared__ float4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, sl = lane >> 4;
  const int k = p.k;
  const int per_wave = C::TILE_FLOATS + 2 * k * 16;  // tile | ls[k][16] | li[k][16]  (one sorted list per query)
  float* tile = smem + wave * per_wave;
  float* ls = tile + C::TILE_FLOATS;
  int* li = reinterpret_cast<int*>(ls + k * 16);

  // B operand: query j, k-slots sl, sl+4, ...  (queries >= B are clamped; their lists are never offered to).
  // The queries arrive RAW: each wave normalises them itself with the same reduction tree and the same correctly
  // rounded sqrt / divide as normalize_rows_kernel (bit-identical), which saves a launch on this latency-bound path.
  // Workgroup b serves query group b % G (16 queries) and is the (b / G)-th of the gridDim.x / G workgroups that sweep
  // the bank for that group.  With G > 1 every group streams the whole bank; the groups run side by side, so the
  // re-reads are L2 / Infinity-Cache hits and HBM still sees the bank about once.
  const int G = p.ngroups;
  const int grp = blockIdx.x % G;
  const int wgi = blockIdx.x / G, nwg = gridDim.x / G;
  const int64_t qbase = (int64_t)grp * 16;
  float breg[D / 4];
  {
    float my_d = 1.f;
    const int nq = (int)min((int64_t)16, p.B - qbase);  // queries of this group
    for (int jj = 0; jj < nq; ++jj) {
      float pp = 0.f;
      for (int c = lane; c < D / 4; c += 64) {
        const float4 v = reinterpret_cast<const float4*>(p.Qn + (qbase + jj) * D)[c];
        pp = fmaf(v.x, v.x, pp);
        pp = fmaf(v.y, v.y, pp);
        pp = fmaf(v.z, v.z, pp);
        pp = fmaf(v.w, v.w, pp);
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) pp = __fadd_rn(pp, __shfl_xor(pp, off));
      const float d = fmaxf(sqrtf(pp), 1e-12f);
      if (jj == j || (j >= nq && jj == nq - 1)) my_d = d;
    }
    const int64_t q = (qbase + j < p.B) ? qbase + j : p.B - 1;
    const float4* qp = reinterpret_cast<const float4*>(p.Qn + q * D);
#pragma unroll
    for (int m = 0; m < D / 4; ++m) {
      const float4 v = qp[m];
      const float x = sl == 0 ? v.x : sl == 1 ? v.y : sl == 2 ? v.z : v.w;
      breg[m] = x / my_d;
    }
  }
  for (int i = lane; i < k * 16; i += 64) {
    ls[i] = RG_NEG_INF;
    li[i] = RG_IDX_NONE;
  }

  const int64_t ntiles = (p.N + C::TILE_KEYS - 1) / C::TILE_KEYS;
  const int64_t gw = (int64_t)wgi * C::WAVES + wave, nw = (int64_t)nwg * C::WAVES;
  const bool live = qbase + j < p.B;
  // A floor from the sampled pre-pass (k-th best over a prefix of the bank) is a valid lower bound of the final k-th
  // best: keys below it can never be selected, so the lists only ever see the few keys that can.
  const float thr_floor = (live && p.thr_init) ? p.thr_init[(qbase + j) * k + k - 1] : RG_NEG_INF;
  float thr = live ? thr_floor : __builtin_huge_valf();

  float4 pre[16];
  const int lrow = lane / (D / 4), lcol = 4 * (lane % (D / 4));
  auto load_tile = [&](int64_t t) {
    const int64_t key0 = t * C::TILE_KEYS + lrow;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int64_t r = key0 + (int64_t)i * C::RPI;
      if (r > p.N - 1) r = p.N - 1;
      pre[i] = *reinterpret_cast<const float4*>(p.Kn + r * D + lcol);
    }
  };

  int64_t t = gw;
  if (t < ntiles) load_tile(t);
  for (; t < ntiles; t += nw) {
    // transpose image: row-major rows in (two 8-byte stores: odd rows are only 8-byte aligned at stride D+2)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float* d = tile + (lrow + i * C::RPI) * C::ROW + lcol;
      *reinterpret_cast<float2*>(d) = make_float2(pre[i].x, pre[i].y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(pre[i].z, pre[i].w);
    }
    const int64_t tn = t + nw;
    load_tile(tn < ntiles ? tn : t);  // next tile in flight under this tile's MFMAs (clamped reload on the last one)
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (int sub = 0; sub < C::SUB; ++sub) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* a = tile + (sub * 16 + j) * C::ROW + sl;
#pragma unroll
      for (int m = 0; m < D / 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * m], breg[m], acc, 0, 0, 0);

      const float mx = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
      if (__any(mx >= thr)) {
        // rare path: the wave inserts the candidates one at a time, all 64 lanes cooperating on each insert
        const int key_base = (int)(t * C::TILE_KEYS) + sub * 16 + 4 * sl;
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (acc[r] >= thr && key_base + r < (int)p.N) mask |= 1u << r;
        unsigned long long pend = __ballot(mask != 0);
        if (__popcll(pend) > 8) {
          // DENSE regime (no floor yet, or many queries): lane-private inserts, the 4 lanes of a query take turns
          while (pend) {
            const int r0 = __ffs(mask) - 1;
            const float my_sc = r0 == 1 ? acc[1] : r0 == 2 ? acc[2] : r0 == 3 ? acc[3] : acc[0];
#pragma unroll 1
            for (int ss = 0; ss < 4; ++ss) {
              if (mask != 0 && sl == ss) list_insert_lane(ls + j, li + j, k, 16, my_sc, key_base + r0);
            }
            mask &= mask - 1;
            if (live) thr = fmaxf(thr_floor, ls[(k - 1) * 16 + j]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if ((mask >> r & 1u) && acc[r] < thr) mask &= ~(1u << r);
            pend = __ballot(mask != 0);
          }
        }
        while (pend) {
          const int src = __ffsll((long long)pend) - 1;          // wave-uniform: lowest lane with a candidate
          const int r0 = __ffs(mask) - 1;                        // (meaningful on lane src)
          const float my_sc = r0 == 1 ? acc[1] : r0 == 2 ? acc[2] : r0 == 3 ? acc[3] : acc[0];
          const float sc = __shfl(my_sc, src);
          const int idx = __shfl(key_base + r0, src);
          const int q = src & 15;
          const float nthr = list_insert_coop(ls, li, q, k, 16, sc, idx, lane);
          if (live && j == q) thr = fmaxf(thr_floor, nthr);
          if (lane == src) mask &= mask - 1;
          // drop this lane's remaining candidates that the tightened threshold already excludes
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((mask >> r & 1u) && acc[r] < thr) mask &= ~(1u << r);
          pend = __ballot(mask != 0);
        }
      }
    }
  }

  // ---- workgroup merge: 8 wave lists per query -> one sorted partial per (query, workgroup) -------------------
  __syncthreads();
  for (int q = wave; q < 16 && qbase + q < p.B; q += C::WAVES) {
    float prev_s = __builtin_huge_valf();
    int prev_i = -1;
    for (int r = 0; r < k; ++r) {
      float best_s = RG_NEG_INF;
      int best_i = RG_IDX_NONE;
      for (int c = lane; c < C::WAVES * k; c += 64) {
        const float* wl = smem + (c / k) * per_wave + C::TILE_FLOATS;
        const float s = wl[(c % k) * 16 + q];
        const int i = reinterpret_cast<const int*>(wl + k * 16)[(c % k) * 16 + q];
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
        const int64_t o = ((qbase + q) * nwg + wgi) * k + r;
        p.part_s[o] = best_s;
        p.part_i[o] = best_i;
      }
      prev_s = best_s;
      prev_i = best_i;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Select the canonical top-k among M candidates per query (partials of the splits, or the per-GPU lists).
// One wave per query; round r picks the best candidate strictly worse than round r-1's winner, so no marking is
// needed and unsorted input is fine.  Candidate c of query b, list g: element (g*gs + b*bs + c), c in [0, k).
// ------------------------------------------------------------------------------------------------------------------
template <typename IdxT, int CPL>  // CPL = candidates per lane held in registers (M <= 64 * CPL)
__global__ void __launch_bounds__(256) topk_select_kernel(const float* __restrict__ cs, const IdxT* __restrict__ ci,
                                                          int G, int64_t B, int k, int64_t gs, int64_t bs,
                                                          int64_t idx_base, float* __restrict__ out_s,
                                                          int64_t* __restrict__ out_i) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int M = G * k;
  // one coalesced sweep: every candidate is loaded exactly once, all loads in flight together
  float s[CPL];
  int64_t id[CPL];
#pragma unroll
  for (int u = 0; u < CPL; ++u) {
    const int c = lane + 64 * u;
    s[u] = RG_NEG_INF;
    id[u] = INT64_MAX;
    if (c < M) {
      const int64_t o = (int64_t)(c / k) * gs + b * bs + (c % k);
      s[u] = cs[o];
      id[u] = (int64_t)ci[o];
    }
  }
  float prev_s = __builtin_huge_valf();
  int64_t prev_i = -1;  // everything is worse than (+inf, -1)
  for (int r = 0; r < k; ++r) {
    float best_s = RG_NEG_INF;
    int64_t best_i = INT64_MAX;
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
      const bool after_prev = (s[u] < prev_s) || (s[u] == prev_s && id[u] > prev_i);
      const bool beats = (s[u] > best_s) || (s[u] == best_s && id[u] < best_i);
      if (after_prev && beats) {
        best_s = s[u];
        best_i = id[u];
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

// k-way merge of G SORTED lists per query (the small-batch kernel's per-workgroup partials; the per-GPU lists of the
// sharded bank).  One wave per query: the lists are staged in LDS once, each lane keeps the heads of <= 4 lists in
// registers, and a round costs one wave argmax plus one LDS read for the winner -- O(k log 64) instead of O(k * G*k/64).
template <typename IdxT>
__global__ void __launch_bounds__(256) topk_merge_sorted_kernel(const float* __restrict__ cs, const IdxT* __restrict__ ci,
                                                               int G, int64_t B, int k, int64_t gs, int64_t bs,
                                                               int64_t idx_base, float* __restrict__ out_s,
                                                               int64_t* __restrict__ out_i) {
  extern __shared__ float4 smem4[];
  float* Ls = reinterpret_cast<float*>(smem4);        // [G*k]
  IdxT* Li = reinterpret_cast<IdxT*>(Ls + G * k);     // [G*k]
  const int lane = threadIdx.x & 63;
  const int64_t b = blockIdx.x;
  const int M = G * k;
  for (int c = threadIdx.x; c < M; c += 256) {  // all 4 waves stage; wave 0 merges
    const int64_t o = (int64_t)(c / k) * gs + b * bs + (c % k);
    Ls[c] = cs[o];
    Li[c] = ci[o];
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  float hs[4];
  int64_t hi[4];
  int hp[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int g = lane + 64 * u;
    hp[u] = 0;
    hs[u] = (g < G) ? Ls[g * k] : RG_NEG_INF;
    hi[u] = (g < G) ? (int64_t)Li[g * k] : INT64_MAX;
  }
  for (int r = 0; r < k; ++r) {
    float best_s = hs[0];
    int64_t best_i = hi[0];
    int best_u = 0;
#pragma unroll
    for (int u = 1; u < 4; ++u)
      if ((hs[u] > best_s) || (hs[u] == best_s && hi[u] < best_i)) {
        best_s = hs[u];
        best_i = hi[u];
        best_u = u;
      }
    float ws = best_s;
    int64_t wi = best_i;
    int wl = lane;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float os = __shfl_xor(ws, off);
      const int64_t oi = __shfl_xor(wi, off);
      const int ol = __shfl_xor(wl, off);
      if ((os > ws) || (os == ws && oi < wi)) {
        ws = os;
        wi = oi;
        wl = ol;
      }
    }
    if (lane == 0) {
      out_s[b * k + r] = ws;
      out_i[b * k + r] = (wi == INT64_MAX) ? wi : wi + idx_base;
    }
    if (lane == wl) {  // advance the winning list (indices are unique, so exactly one lane / one head matches)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u == best_u) {
          const int g = lane + 64 * u;
          const int np = ++hp[u];
          hs[u] = (np < k) ? Ls[g * k + np] : RG_NEG_INF;
          hi[u] = (np < k) ? (int64_t)Li[g * k + np] : INT64_MAX;
        }
    }
  }
}

template <typename IdxT>
static int launch_merge_sorted(const float* cs, const IdxT* ci, int G, int64_t B, int k, int64_t gs, int64_t bs,
                               int64_t idx_base, float* out_s, int64_t* out_i, hipStream_t st) {
  if (G > 256) {
    set_error("topk merge: %d lists per query exceed 256", G);
    return RAGRAPH_EUNSUPPORTED;
  }
  const size_t lds = (size_t)G * k * (sizeof(float) + sizeof(IdxT));
  hipLaunchKernelGGL(topk_merge_sorted_kernel<IdxT>, dim3((unsigned)B), dim3(256), lds, st, cs, ci, G, B, k, gs, bs,
                     idx_base, out_s, out_i);
  RG_CHECK_LAUNCH("topk_merge_sorted");
  return RAGRAPH_OK;
}

template <typename IdxT>
static int launch_select(const float* cs, const IdxT* ci, int G, int64_t B, int k, int64_t gs, int64_t bs,
                         int64_t idx_base, float* out_s, int64_t* out_i, hipStream_t st) {
  const int M = G * k;
  const int wpb = 4;
  dim3 grid((unsigned)cdiv(B, wpb)), block(wpb * 64);
#define RG_SEL(CPL_)                                                                                          \
  hipLaunchKernelGGL((topk_select_kernel<IdxT, CPL_>), grid, block, 0, st, cs, ci, G, B, k, gs, bs, idx_base, \
                     out_s, out_i)
  if (M <= 64) RG_SEL(1);
  else if (M <= 128) RG_SEL(2);
  else if (M <= 256) RG_SEL(4);
  else if (M <= 512) RG_SEL(8);
  else if (M <= 1024) RG_SEL(16);
  else if (M <= 2048) RG_SEL(32);
  else if (M <= 4096) RG_SEL(64);
  else {
    set_error("topk select: %d candidates per query exceed 4096", M);
    return RAGRAPH_EUNSUPPORTED;
  }
#undef RG_SEL
  RG_CHECK_LAUNCH("topk_select");
  return RAGRAPH_OK;
}

// ---- work planning -----------------------------------------------------------------------------------------------
// Tile kernel: one persistent workgroup per CU walks segments (see the kernel); the plan fixes the group shape and the
// number of partial-list slots per query.  Streaming kernel: nsplit = workgroups per query group.
struct TopkPlan {
  int nsplit;          // partial lists per query
  int xcd_map, wgs_per_group, lb_min, warm_stages, depth[2];
  int64_t nstages_total;
  size_t qn_bytes, part_s_bytes, part_i_bytes;
};

static int device_cus() { return device_cus_multiple_of_8(); }  // per device (common.h)

// Up to this many queries take the wave-streaming kernel (groups of 16 queries, 16x16x4 MFMA): it fills the chip at any
// batch size, whereas the tile kernel needs >= a few query tiles of 256 to do so (measured crossover).
static const int SMALLB_MAX = 128;  // measured at N = 1M, D = 256: streaming 0.49 / 0.78 / 1.36 ms at B = 32 / 64 / 128; tile kernel 1.5 ms flat up to B = 256
// its 8 wave-private 16 KiB tiles + lists must fit the 160 KiB LDS (k <= 31 at D = 256, 30 at D = 128, 28 at D = 64)
static bool use_streaming(int64_t B, int k, int D) {
  static const int64_t bmax = [] {  // RAGRAPH_TOPK_STREAM_MAX: diagnostic override of the crossover (read once)
    const char* e = getenv("RAGRAPH_TOPK_STREAM_MAX");
    return e ? (int64_t)atoi(e) : (int64_t)SMALLB_MAX;
  }();
  const size_t tile_bytes = sizeof(float) * 16 * (256 / D) * (D + 2);  // SmallCfg<D>::TILE_FLOATS
  return B <= bmax && 8 * (tile_bytes + 128 * (size_t)k) <= 160 * 1024;
}

static TopkPlan plan_topk(int64_t B, int64_t N, int D, int k) {
  if (use_streaming(B, k, D)) {  // streaming kernel: nsplit = workgroups PER GROUP (one sorted partial list each)
    const int tile_keys = 16 * (256 / D);
    const int64_t ntiles = cdiv(N, tile_keys);
    const int64_t G = cdiv(B, 16);
    int64_t wgs = cdiv(ntiles, 8 * 2);  // >= 2 tiles per wave so the prefetch has something to overlap
    const int64_t cap = G <= 2 ? 256 : (512 / G > 0 ? 512 / G : 1);  // ~256-512 workgroups in total
    if (wgs > cap) wgs = cap;
    if (wgs * k > 4096) wgs = 4096 / k;  // the select kernel holds <= 4096 candidates per query
    if (wgs < 1) wgs = 1;
    TopkPlan ps;
    ps.nsplit = (int)wgs;
    ps.xcd_map = ps.wgs_per_group = ps.lb_min = ps.warm_stages = ps.depth[0] = ps.depth[1] = 0;
    ps.nstages_total = 0;
    ps.qn_bytes = align_up((size_t)B * D * sizeof(float), 256);
    ps.part_s_bytes = align_up((size_t)B * wgs * k * sizeof(float), 256);
    ps.part_i_bytes = align_up((size_t)B * wgs * k * sizeof(int), 256);
    return ps;
  }
  const int QT = 256, CUS = device_cus();
  const int stage_keys = 32 * (256 / D);
  const int64_t qtiles = cdiv(B, QT);
  const int64_t NS = cdiv(N, stage_keys);
  TopkPlan pl;
  pl.xcd_map = qtiles >= 64 ? 1 : 0;  // enough query tiles to give every XCD its own
  const int G = pl.xcd_map ? 8 : 1;
  pl.wgs_per_group = CUS / G;
  pl.nstages_total = NS;
  // a piece of the leftover line is at least 4 stages (its prefetch pipeline) and long enough that no tile is cut into
  // more lists than the select kernel holds (4096 candidates per query)
  const int64_t cap = 4096 / k;
  int64_t lb_min = cdiv(NS, cap - 1);
  if (lb_min < 4) lb_min = 4;
  pl.lb_min = (int)lb_min;
  // measured on MI355X (k = 10, D = 256): a list warm-up costs about as much as streaming 420 k keys
  pl.warm_stages = (int)cdiv(420 * (int64_t)k, stage_keys);
  int64_t slots = 1;
  const int64_t nq0 = pl.xcd_map ? (qtiles + 7) / 8 : qtiles;
  for (int v = 0; v < 2; ++v) {  // groups own nq0 or (some of them, with the XCD mapping) nq0 - 1 tiles
    const int64_t nq = nq0 - v;
    pl.depth[v] = 0;
    if (nq < 1 || (v == 1 && (!pl.xcd_map || qtiles % 8 == 0))) continue;
    const SegmentWalker::Choice ch = SegmentWalker::choose_depth(nq, NS, pl.wgs_per_group, pl.lb_min, pl.warm_stages);
    static const int depth_cap = [] {  // RAGRAPH_TOPK_DEPTH: diagnostic cap on the lockstep steps (read once)
      const char* e = getenv("RAGRAPH_TOPK_DEPTH");
      return e ? atoi(e) : 1 << 30;
    }();
    pl.depth[v] = ch.depth < depth_cap ? ch.depth : depth_cap;
    int64_t sl = ch.slots;
    if (pl.depth[v] != ch.depth) {  // recount the slots of the capped plan
      SegmentWalker w(nq, NS, pl.wgs_per_group, pl.lb_min, pl.warm_stages, pl.depth[v], -1);
      Segment sg;
      while (w.next(sg)) {
      }
      sl = w.used;
    }
    if (sl > slots) slots = sl;
  }
  pl.nsplit = (int)slots;
  pl.qn_bytes = align_up((size_t)B * D * sizeof(float), 256);
  pl.part_s_bytes = align_up((size_t)B * slots * k * sizeof(float), 256);
  pl.part_i_bytes = align_up((size_t)B * slots * k * sizeof(int), 256);
  return pl;
}

template <int D, int RING>
static int launch_topk_ring(const TopkParams& p, int64_t qtiles, hipStream_t st) {
  using C = TopkCfg<D>;
  const size_t lds = C::lds_bytes(p.k, RING);
  static DeviceOnce lds_once;  // per device (common.h)
  if (hipError_t e = raise_dynamic_lds(lds_once, &topk_stream_kernel<D, RING>, 160 * 1024); e != hipSuccess) {
    set_error("topk_cosine: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    return RAGRAPH_EDEVICE;
  }
  const int64_t grid = (int64_t)p.wgs_per_group * (p.xcd_map ? 8 : 1);  // one persistent workgroup per CU
#ifdef RG_TOPK_TIMING
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_topk_timing), zero, sizeof(zero));
#endif
  hipLaunchKernelGGL((topk_stream_kernel<D, RING>), dim3((unsigned)grid), dim3(C::THREADS), lds, st, p);
  RG_CHECK_LAUNCH("topk_cosine");
#ifdef RG_TOPK_TIMING
  (void)hipDeviceSynchronize();
  unsigned long long t[8];
  (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_topk_timing), sizeof(t));
  if (t[5]) {
    const double n = (double)t[5];  // wave-stages
    fprintf(stderr, "[topk timing] RING=%d wave-stages=%.0f cycles/stage: wait_full %.1f compute %.1f (of which epilogue %.1f) "
            "signal+vmcnt %.1f wait_free %.1f dma_issue %.1f total %.1f\n", RING, n, t[0] / n, t[1] / n, t[6] / n, t[2] / n,
            t[3] / n, t[4] / n, (t[0] + t[1] + t[2] + t[3] + t[4]) / n);
  }
#endif
  return RAGRAPH_OK;
}

template <int D>
static int launch_topk(const TopkParams& p, int64_t qtiles, hipStream_t st) {
  // the barrier-free 3-slot ring needs 3 x 33 KB of stages next to the 2 KB * k of lists in the 160 KB LDS
  static const int ring_env = [] {  // diagnostic override (read once): RAGRAPH_TOPK_RING = 2, 3 or 4
    const char* e = getenv("RAGRAPH_TOPK_RING");
    return e ? atoi(e) : 4;
  }();
  const bool fits3 = TopkCfg<D>::lds_bytes(p.k, 3) <= 160 * 1024;
  const bool want3 = ring_env >= 3;
  if constexpr (D == 256) {
    const bool fits4 = TopkCfg<D>::lds_bytes(p.k, 4) <= 160 * 1024;
    if (p.Kp && fits4 && ring_env != 2 && ring_env != 3 && !(p.ablate & 6)) return launch_topk_ring<D, 4>(p, qtiles, st);
  }
  if (fits3 && want3 && !(p.ablate & 6)) return launch_topk_ring<D, 3>(p, qtiles, st);
  return launch_topk_ring<D, 2>(p, qtiles, st);
}

template <int D>
static int launch_smallb(const TopkParams& p, hipStream_t st) {
  using C = SmallCfg<D>;
  const size_t lds = C::lds_bytes(p.k);
  static DeviceOnce lds_once;  // per device (common.h)
  if (hipError_t e = raise_dynamic_lds(lds_once, &topk_smallb_kernel<D>, 160 * 1024); e != hipSuccess) {
    set_error("topk_cosine(small batch): cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    return RAGRAPH_EDEVICE;
  }
  hipLaunchKernelGGL(topk_smallb_kernel<D>, dim3((unsigned)(p.nsplit * p.ngroups)), dim3(512), lds, st, p);
  RG_CHECK_LAUNCH("topk_cosine(small batch)");
  return RAGRAPH_OK;
}

}  // namespace ragraph

using namespace ragraph;

// 32 < k <= 64 (e.g. RAGraph_edge 'vanilla' phase on amazon, retrieve_num = 50, modules/RAGraph.py:40): the per-query
// lists no longer fit next to the stages in LDS, so the scores of a slab of queries are materialised with the dense
// kernel (the same fmaf chains, hence the same bits) and selected by topk_rows -- the reference's own slab idiom
// (modules/RAGraph.py:298-311), kept on the device.
// Small score matrices are cheaper to WRITE than to avoid: the dense kernel fills the whole chip whatever B is, and
// topk_rows has no list warm-up per workgroup, while the fused kernels pay launch-sized fixed costs (query operands,
// ring, list inserts) that only a long key stream amortises.  Measured (us, slab vs fused kernels, k = 10): 64 x 20000
// x 256: 55 vs 209; 128 x 50000: 106 vs 341; 1500 x 10000: 170 vs 257; 2000 x 16000: 292 vs 369; 4000 x 4000 x 128:
// 121 vs 136; not 2708 x 10000 x 128 (186 vs 168) nor one to a few queries against a long bank (the dense kernel pads
// to 64 rows; the streaming kernel runs at HBM speed).  RAGRAPH_TOPK_SLAB=0 turns the rule off (A/B).
static bool use_slab(int64_t B, int64_t N, int D) {
  const char* e = getenv("RAGRAPH_TOPK_SLAB");  // (read per call: the tests switch it to reach both families of kernels)
  if (e && atoi(e) == 0) return false;
  if (B <= 128 && N <= 8192 && B * N <= ((int64_t)1 << 20)) return true;  // a handful of queries x a tiny bank
  // (D = 64: the scores cost more to write than to compute -- 4096 x 16384 x 64: 228 vs 379 us -- up to 2^26 of them)
  const int64_t lim = D == 256 ? (int64_t)1 << 25 : D == 64 ? (int64_t)1 << 26 : (int64_t)1 << 24;
  return B >= 8 && N <= 131072 && B * N <= lim;
}

static int64_t slab_rows_for(int64_t B, int64_t N) {
  int64_t rows = ((int64_t)1 << 30) / (4 * N);  // ~1 GiB of scores per slab
  if (rows < 64) rows = 64;
  return rows < B ? rows : B;
}

// Any other row width (the reference accepts every emb_size: SimilarityFunctions.py:6-16) takes the slab path: row norms,
// the dense kernel and the row top-k are written for any D -- the same fmaf chains, hence the same bits as the oracle.
static bool fused_width(int D) { return D == 64 || D == 128 || D == 256; }

// The dense kernel takes up to 65535 blocks of 64 columns per launch: longer banks are scored in key chunks whose
// per-chunk lists are merged in canonical order.
constexpr int64_t SLAB_KEY_CHUNK = (int64_t)1 << 22;

static int64_t slab_chunks(int64_t N) { return cdiv(N, SLAB_KEY_CHUNK); }

static size_t slab_workspace_bytes(int64_t B, int64_t N, int D, int k) {
  const int64_t rows = slab_rows_for(B, N), G = slab_chunks(N), nc = cdiv(N, G);
  size_t bytes = align_up((size_t)B * D * sizeof(float), 256) + align_up((size_t)rows * nc * sizeof(float), 256);
  if (G > 1) bytes += align_up((size_t)G * rows * k * sizeof(float), 256) + align_up((size_t)G * rows * k * sizeof(int64_t), 256);
  return bytes;
}

extern "C" size_t ragraph_topk_cosine_workspace_bytes(int64_t B, int64_t N, int D, int k) {
  if (B < 1 || N < 1 || k < 1 || D < 1) return 0;
  if (!fused_width(D) || k > 32 || use_slab(B, N, D)) return slab_workspace_bytes(B, N, D, k);
  TopkPlan pl = plan_topk(B, N, D, k);
  return pl.qn_bytes + pl.part_s_bytes + pl.part_i_bytes;
}

extern "C" int ragraph_topk_rows_f32(const float* S, int64_t B, int64_t N, int64_t ld, int k, float* out_scores,
                                     int64_t* out_idx, void* stream);
extern "C" int ragraph_linear_f32(const float* X, int64_t M, int K, const float* W, int64_t N, const float* bias,
                                  int act, float alpha, float* Y, void* stream);

namespace ragraph {
__global__ void __launch_bounds__(256) add_idx_base_kernel(int64_t* idx, int64_t n, int64_t base) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) idx[i] += base;
}
}  // namespace ragraph

namespace ragraph {
// Kp[n] = [Kn[n][0], Kn[n][2], ... | Kn[n][1], Kn[n][3], ...]: the tile kernel's LDS row image, made once per bank
// update so the stream can be moved HBM -> LDS by DMA.  One float4 of output per thread.
__global__ void __launch_bounds__(256) pack_keys_kernel(const float* __restrict__ Kn, int64_t n4, int D,
                                                        float* __restrict__ Kp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // output float4 index
  if (i >= n4) return;
  const int per_row = D / 4, half = D / 8;
  const int64_t row = i / per_row;
  const int c = (int)(i % per_row);
  const int odd = c >= half;
  const int m0 = 4 * (c - (odd ? half : 0));  // first pair index of this chunk
  const float* src = Kn + row * D + 2 * m0 + odd;
  float4 v;
  v.x = src[0];
  v.y = src[2];
  v.z = src[4];
  v.w = src[6];
  reinterpret_cast<float4*>(Kp)[i] = v;
}
}  // namespace ragraph

extern "C" int ragraph_pack_keys_f32(const float* Kn, int64_t N, int D, float* Kp, void* stream) {
  RG_REQUIRE(Kn && Kp, RAGRAPH_EINVAL, "pack_keys: null pointer");
  RG_REQUIRE(N >= 1, RAGRAPH_EINVAL, "pack_keys: N=%lld must be >= 1", (long long)N);
  RG_REQUIRE(D == 64 || D == 128 || D == 256, RAGRAPH_EUNSUPPORTED, "pack_keys: D=%d not in {64,128,256}", D);
  RG_REQUIRE(aligned16(Kn) && aligned16(Kp) && Kn != Kp, RAGRAPH_EINVAL, "pack_keys: Kn, Kp must be distinct, 16-B aligned");
  const int64_t n4 = N * (D / 4);
  hipLaunchKernelGGL(pack_keys_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, as_stream(stream), Kn, n4, D, Kp);
  RG_CHECK_LAUNCH("pack_keys");
  return RAGRAPH_OK;
}

extern "C" int ragraph_topk_cosine_f32(const float* Q, int64_t B, const float* Kn, int64_t N, int D, int k,
                                       int64_t idx_base, float* out_scores, int64_t* out_idx, void* ws,
                                       size_t ws_bytes, void* stream) {
  return ragraph_topk_cosine_bank_f32(Q, B, Kn, nullptr, N, D, k, idx_base, out_scores, out_idx, ws, ws_bytes, stream);
}

extern "C" int ragraph_topk_cosine_bank_f32(const float* Q, int64_t B, const float* Kn, const float* Kp, int64_t N,
                                            int D, int k, int64_t idx_base, float* out_scores, int64_t* out_idx,
                                            void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(Q && Kn && out_scores && out_idx && ws, RAGRAPH_EINVAL, "topk_cosine: null pointer");
  RG_REQUIRE(!Kp || aligned16(Kp), RAGRAPH_EINVAL, "topk_cosine: Kp must be 16-B aligned");
  RG_REQUIRE(B >= 1 && N >= 1, RAGRAPH_EINVAL, "topk_cosine: B=%lld N=%lld must be >= 1", (long long)B, (long long)N);
  RG_REQUIRE(k >= 1 && k <= N, RAGRAPH_EINVAL, "topk_cosine: k=%d out of range for N=%lld (torch.topk raises too)", k,
             (long long)N);
  RG_REQUIRE(D >= 1, RAGRAPH_EINVAL, "topk_cosine: D=%d must be >= 1", D);
  RG_REQUIRE(k <= RAGRAPH_TOPK_MAX, RAGRAPH_EUNSUPPORTED, "topk_cosine: k=%d > %d not supported by the fused kernel", k,
             RAGRAPH_TOPK_MAX);
  RG_REQUIRE(N < (int64_t)INT_MAX - 1024, RAGRAPH_EUNSUPPORTED, "topk_cosine: shard rows must fit int32");
  RG_REQUIRE(aligned16(Q) && aligned16(Kn) && aligned16(ws), RAGRAPH_EINVAL, "topk_cosine: Q, Kn, ws must be 16-B aligned");
  // A handful of queries against a bank of a few thousand keys (graph classification: 16 graphs x the training set's
  // 1113): every key is still a candidate for every list, and the streaming kernel's cooperative inserts -- one
  // (query, key) at a time -- were 45 us of a 150 us forward.  The slab path (dense kernel + topk_rows) has no lists.
  // Row widths other than 64 / 128 / 256 always take it (any D >= 1).
  if (!fused_width(D) || k > 32 || use_slab(B, N, D)) {  // materialised slabs, see slab_rows_for()
    const size_t qn_bytes = align_up((size_t)B * D * sizeof(float), 256);
    const int64_t rows = slab_rows_for(B, N), G = slab_chunks(N), nc = cdiv(N, G);
    const size_t s_bytes = align_up((size_t)rows * nc * sizeof(float), 256);
    const size_t ps_bytes = G > 1 ? align_up((size_t)G * rows * k * sizeof(float), 256) : 0;
    RG_REQUIRE(ws_bytes >= slab_workspace_bytes(B, N, D, k), RAGRAPH_EWORKSPACE, "topk_cosine: workspace too small");
    RG_REQUIRE(G * k <= 4096, RAGRAPH_EUNSUPPORTED, "topk_cosine: %lld key chunks x k=%d exceed the merge", (long long)G, k);
    hipStream_t st = as_stream(stream);
    float* Qn = reinterpret_cast<float*>(ws);
    float* S = reinterpret_cast<float*>(static_cast<char*>(ws) + qn_bytes);
    float* part_s = reinterpret_cast<float*>(static_cast<char*>(ws) + qn_bytes + s_bytes);
    int64_t* part_i = reinterpret_cast<int64_t*>(static_cast<char*>(ws) + qn_bytes + s_bytes + ps_bytes);
    int rc = ragraph_normalize_rows_f32(Q, B, D, Qn, stream);
    for (int64_t b0 = 0; rc == RAGRAPH_OK && b0 < B; b0 += rows) {
      const int64_t nb = (B - b0 < rows) ? B - b0 : rows;
      if (G == 1) {
        rc = ragraph_linear_f32(Qn + b0 * D, nb, D, Kn, N, nullptr, RAGRAPH_ACT_NONE, 0.f, S, stream);
        if (rc == RAGRAPH_OK) rc = ragraph_topk_rows_f32(S, nb, N, N, k, out_scores + b0 * k, out_idx + b0 * k, stream);
        continue;
      }
      for (int64_t g = 0; rc == RAGRAPH_OK && g < G; ++g) {  // key chunks: lists [G][nb][k], indices made bank-relative
        const int64_t n0 = g * nc, nn = (N - n0 < nc) ? N - n0 : nc;  // (balanced chunks: each far longer than k)
        rc = ragraph_linear_f32(Qn + b0 * D, nb, D, Kn + n0 * D, nn, nullptr, RAGRAPH_ACT_NONE, 0.f, S, stream);
        if (rc != RAGRAPH_OK) break;
        rc = ragraph_topk_rows_f32(S, nb, nn, nn, k, part_s + g * nb * k, part_i + g * nb * k, stream);
        if (rc == RAGRAPH_OK && n0 != 0) {
          hipLaunchKernelGGL(add_idx_base_kernel, dim3((unsigned)cdiv(nb * k, 256)), dim3(256), 0, st, part_i + g * nb * k,
                             nb * k, n0);
          RG_CHECK_LAUNCH("topk_cosine(chunk base)");
        }
      }
      if (rc == RAGRAPH_OK)
        rc = launch_select<int64_t>(part_s, part_i, (int)G, nb, k, nb * k, (int64_t)k, (int64_t)0, out_scores + b0 * k,
                                    out_idx + b0 * k, st);
    }
    if (rc == RAGRAPH_OK && idx_base != 0) {
      hipLaunchKernelGGL(add_idx_base_kernel, dim3((unsigned)cdiv(B * k, 256)), dim3(256), 0, as_stream(stream), out_idx,
                         B * k, idx_base);
      RG_CHECK_LAUNCH("topk_cosine(add base)");
    }
    return rc;
  }
  TopkPlan pl = plan_topk(B, N, D, k);
  RG_REQUIRE(ws_bytes >= pl.qn_bytes + pl.part_s_bytes + pl.part_i_bytes, RAGRAPH_EWORKSPACE,
             "topk_cosine: workspace %zu < %zu", ws_bytes, pl.qn_bytes + pl.part_s_bytes + pl.part_i_bytes);
  hipStream_t st = as_stream(stream);
  char* w = static_cast<char*>(ws);
  float* Qn = reinterpret_cast<float*>(w);
  float* part_s = reinterpret_cast<float*>(w + pl.qn_bytes);
  int* part_i = reinterpret_cast<int*>(w + pl.qn_bytes + pl.part_s_bytes);

  int rc = RAGRAPH_OK;
  const bool streaming = use_streaming(B, k, D);
  if (!streaming) {  // the streaming kernel normalises its queries itself
    rc = ragraph_normalize_rows_f32(Q, B, D, Qn, stream);
    if (rc != RAGRAPH_OK) return rc;
  }

  TopkParams p;
  p.Qn = streaming ? Q : Qn;
  p.Kn = Kn;
  p.Kp = Kp;
  p.B = B;
  p.N = N;
  p.k = k;
  p.nsplit = pl.nsplit;
  p.ngroups = (int)cdiv(B, 16);
  p.qtiles = cdiv(B, 256);
  p.xcd_map = pl.xcd_map;
  p.wgs_per_group = pl.wgs_per_group;
  p.lb_min = pl.lb_min;
  p.warm_stages = pl.warm_stages;
  p.depth[0] = pl.depth[0];
  p.depth[1] = pl.depth[1];
  p.nstages_total = pl.nstages_total;
  p.part_s = part_s;
  p.part_i = part_i;
  static const int ablate_env = [] {  // timing-only diagnostic switches (read once), see TopkParams::ablate
    const char* e = getenv("RAGRAPH_TOPK_ABLATE");
    return e ? atoi(e) : 0;
  }();
  p.ablate = ablate_env;
  const int64_t qtiles = cdiv(B, 256);
  p.thr_init = nullptr;
  if (streaming) {
    // >= 4 queries: each of the ~2000 waves sees only N/2000 keys, so its lists would stay in their warm-up for the
    // whole stream.  A pre-pass over a 4096-key prefix (1/256 of a 1M bank) gives every query the k-th best score of
    // that prefix -- a valid lower bound of the final k-th best -- and the main pass then only ever inserts the
    // ~k*N/4096 keys per query that beat it.  For 1-3 queries the two extra launches cost more than they save.
    const int64_t prefix = 4096;
    if (B >= 4 && N >= 16 * prefix && k <= prefix) {
      TopkParams pp = p;
      pp.N = prefix;
      pp.nsplit = (int)cdiv(cdiv(prefix, 16 * (256 / D)), 8);  // one tile per wave ...
      if (pp.nsplit > pl.nsplit) pp.nsplit = pl.nsplit;         // ... within the main pass's partial-list workspace
      rc = D == 256 ? launch_smallb<256>(pp, st) : D == 128 ? launch_smallb<128>(pp, st) : launch_smallb<64>(pp, st);
      if (rc != RAGRAPH_OK) return rc;
      rc = launch_merge_sorted<int>(part_s, part_i, pp.nsplit, B, k, (int64_t)k, (int64_t)pp.nsplit * k, 0, out_scores,
                                    out_idx, st);
      if (rc != RAGRAPH_OK) return rc;
      p.thr_init = out_scores;  // read by the main pass, overwritten by the final merge after it (stream order)
    }
    rc = D == 256 ? launch_smallb<256>(p, st) : D == 128 ? launch_smallb<128>(p, st) : launch_smallb<64>(p, st);
  } else if (D == 256) {
    rc = launch_topk<256>(p, qtiles, st);
  } else if (D == 128) {
    rc = launch_topk<128>(p, qtiles, st);
  } else {
    rc = launch_topk<64>(p, qtiles, st);
  }
  if (rc != RAGRAPH_OK) return rc;

  // partial layout [B][nsplit][k]: list g of query b at b*(nsplit*k) + g*k
  if (streaming)  // per-workgroup partials of the streaming kernel are sorted
    return launch_merge_sorted<int>(part_s, part_i, pl.nsplit, B, k, (int64_t)k, (int64_t)pl.nsplit * k, idx_base,
                                    out_scores, out_idx, st);
  return launch_select<int>(part_s, part_i, pl.nsplit, B, k, (int64_t)k, (int64_t)pl.nsplit * k, idx_base, out_scores,
                            out_idx, st);
}

extern "C" int ragraph_topk_merge_f32(const float* scores, const int64_t* idx, int G, int64_t B, int k,
                                      float* out_scores, int64_t* out_idx, void* stream) {
  RG_REQUIRE(scores && idx && out_scores && out_idx, RAGRAPH_EINVAL, "topk_merge: null pointer");
  RG_REQUIRE(G >= 1 && B >= 1 && k >= 1, RAGRAPH_EINVAL, "topk_merge: G,B,k must be >= 1");
  RG_REQUIRE((int64_t)G * k <= 4096, RAGRAPH_EUNSUPPORTED, "topk_merge: G*k=%lld > 4096", (long long)G * k);
  return launch_select<int64_t>(scores, idx, G, B, k, (int64_t)B * k, (int64_t)k, (int64_t)0, out_scores, out_idx,
                                as_stream(stream));
}
