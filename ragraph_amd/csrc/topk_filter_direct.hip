// bf16 filter of the exact cosine top-k for UP TO 256 QUERIES (SimilarityFunctions.py:6-16 + ToyGraphBase.py:66-67 at the
// reference's real batch sizes: graph classification sends 1 query per forward, RAGraph_node a few hundred).
//
// With so few queries the score matrix is cheap (2 * 256 * N * D flops = 52 us of bf16 MFMA at N = 1M, D = 256) and the call
// is bound by ONE pass over the bf16 bank copy (2 N D bytes): the kernel is built around that stream, not around the
// matrix cores.
//   * The bank copy is stored in MFMA fragment order (filter_common.h: v_mfma_f32_16x16x32_bf16, 16 keys x 32 elements
//     per 1-KiB block): a wave fetches an A operand with one coalesced 1-KiB global_load_dwordx4 straight into the
//     registers the MFMA reads.  No LDS staging, no ring, no
//     barrier, no hand-over between waves: every wave streams its own 16-KiB units (512 key-elements x 16 blocks), the
//     next unit's sixteen loads in flight while the current one is multiplied -- 8 waves x 16 KiB = 128 KiB in flight per
//     CU.  Units are dealt round-robin over all waves of the grid, so neighbouring waves read neighbouring 16 KiB.
//   * The queries are the B operands, converted once per call to bf16 in fragment order too (1 KiB per k-step and group
//     of 16; filter_prep_kernel).  Up to 32 of them live in registers (D/4 VGPRs); 33..256 are copied into LDS: every
//     unit is multiplied with each group in turn, one conflict-free ds_read_b128 per k-step feeding the MFMAs of BOTH
//     16-key halves of every sub-tile of the unit (32 SUBS cycles of MFMA per KiB read).  The key stream is thus read
//     ONCE for all groups.
//   * Epilogue as in the ring kernel: one wave-uniform test of the accumulators' maxima per 32 keys x 32 queries; passing
//     keys leave through a wave-private LDS buffer and reach the per-query candidate lists in flushes.
//   * BOUND mode: no thresholds, no candidates -- per query the best approximate score of each of `ngroups` consecutive
//     parts of the key range (atomicMax on order-preserving ints), from which the first lower bound of the k-th best
//     score is made (filter_threshold).
// The key loads are plain (nontemporal) loads: hipcc counts its own vmcnt waits -- block i of the current unit is needed
// when the unit's younger loads and the next unit's sixteen are still outstanding -- which it gets right as long as no
// asm statement claims the registers in between (an earlier version with asm loads and tied asm waits made the register
// allocator copy registers whose loads had not landed).  The B-operand reads from LDS are asm, run ahead of the MFMAs
// and are counted with lgkmcnt by hand.
#include "filter_common.h"
#include <type_traits>

namespace ragraph {

template <int D_>
struct DirectCfg {
  static constexpr int D = D_;
  static constexpr int WAVES = 8, THREADS = 512;
  static constexpr int KSTEPS = D / 16;              // 1-KiB blocks per 32-key sub-tile (2 halves x KS32): 16 / 8 / 4
  static constexpr int KS32 = D / 32;                // MFMA k-steps (32 elements) per sub-tile: 8 / 4 / 2
  static constexpr int UNIT_BLOCKS = 16;             // blocks per unit = 16 KiB in flight per wave and buffer
  static constexpr int SUBS = UNIT_BLOCKS / KSTEPS;  // sub-tiles per unit: 1 / 2 / 4
  static constexpr int UNIT_KEYS = 32 * SUBS;        // 32 / 64 / 128
  // entries (8 B) of a wave's candidate buffer; one group pass over a unit pushes at most 64 SUBS of them, two groups are
  // processed per flush check (D = 32: the int8 copy of a D = 64 bank, 8 sub-tiles per unit)
  static constexpr int CAND_BUF = SUBS == 1 ? 256 : (SUBS <= 4 ? 512 : 128 * SUBS + 64);
  static constexpr int GROUP_BYTES = KS32 * 1024;    // one group of 16 queries as bf16 B operands
#ifndef RG_DIRECT_LA
#define RG_DIRECT_LA 4
#endif
  static constexpr int LA = KS32 < RG_DIRECT_LA ? KS32 : RG_DIRECT_LA;  // B-operand reads in flight ahead of their MFMAs
  static constexpr size_t lds_bytes(int groups_in_lds) {
    return (size_t)groups_in_lds * GROUP_BYTES + (size_t)WAVES * CAND_BUF * 8 + 512 * sizeof(float) + WAVES * sizeof(int);
  }
};

struct DirectParams {
  const uint16_t* Qb;
  const uint16_t* Kb;
  int64_t B;
  int64_t unit0, nunits;  // units [unit0, unit0 + nunits) of the bank copy
  int64_t key_end;        // keys >= key_end never pass (padding, or the next level's)
  FilterThr thr;
  int* count;             // [B][FILTER_COUNT_STRIDE]
  int* cand;
  int cap, nsub, subcap;  // nsub sub-lists of subcap = cap / nsub slots per query
  int scored;             // (int8) lists of {key, I} pairs (int2 slots): filter_common.h DirectArgs
  int* gmax;              // BOUND
  int ngroups;            // BOUND: parts of the range
  int qgroups;            // groups of 16 queries (2..16, even: the query image is padded to whole 32s)
};

#ifdef RG_DIRECT_TIMING  // diagnostic build: wall-clock stamps (10 ns ticks) of the first and the last workgroup's thread 0
__device__ unsigned long long g_direct_t[2][2][8];
#define RG_DSTAMP(i_)                                                                       \
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                \
  g_direct_t[BOUND][blockIdx.x != 0][i_] = wall_clock64()
#else
#define RG_DSTAMP(i_)
#endif

typedef int i32x4 __attribute__((ext_vector_type(4)));

// QREG: <= 32 queries, B operands in registers; else `qgroups` groups in LDS.
// I8: the pass runs on the int8 copy (filter_common.h): the unit / sub-tile geometry of a bf16 bank of D / 2 elements (a key is
// D bytes: the 0.512 GB stream of a 1M x 256 bank becomes 0.256 GB), v_mfma_i32_16x16x64_i8, integer thresholds; p.Qb is the
// queries' int8 image (filter_prep_kernel), same block structure as the bf16 one at half the k-steps.
template <int D, bool QREG, bool BOUND, bool I8 = false>
__global__ void __launch_bounds__(512, 2) topk_filter_direct_kernel(DirectParams p) {
  using C = DirectCfg<I8 ? D / 2 : D>;
  static_assert(!(I8 && BOUND), "the bound pass runs on the bf16 copy");
  using acc_t = typename std::conditional<I8, i32x4, f32x4>::type;
  RG_DSTAMP(0);
  extern __shared__ float4 dsmem4[];
  char* smem = reinterpret_cast<char*>(dsmem4);
  const int ngl = QREG ? 0 : p.qgroups;
  char* qlds = smem;                                                             // [ngl][KS32][64] x 16 B
  uint2* wbuf_all = reinterpret_cast<uint2*>(smem + (size_t)ngl * C::GROUP_BYTES);  // [WAVES][CAND_BUF]
  float* thr_lds = reinterpret_cast<float*>(wbuf_all + C::WAVES * C::CAND_BUF);     // [512]: [256] thresholds (int8: for keys of
                                                                                    // NORMAL granules), [256] int8: of HEAVY ones

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;

  // ---- thresholds (one per query of the tile) and, beyond 32 queries, the B operands in LDS -------------------------
  if (tid < (I8 ? 512 : 256)) {
    if constexpr (I8) {  // (the integer thresholds of the two classes of granules, kept as bits in the float array)
      int t = INT_MAX;
      if ((tid & 255) < p.B) t = filter_threshold_i8(p.thr, tid & 255, tid >> 8);
      thr_lds[tid] = __int_as_float(t);
    } else {
      float t = __builtin_huge_valf();  // padded queries never pass
      if (!BOUND && tid < p.B) t = filter_threshold(p.thr, tid);
      thr_lds[tid] = t;
    }
  }
  // the B operands: filter_prep_kernel left them in HBM as bf16 in fragment order, so a workgroup copies its image
  // into LDS linearly (33..256 queries) or a wave takes its 64 VGPRs straight from it (<= 32 queries)
  if constexpr (!QREG) {
    const int pieces = ngl * C::KS32 * 64;
    for (int i = tid; i < pieces; i += C::THREADS)
      *reinterpret_cast<f32x4*>(qlds + (size_t)i * 16) = reinterpret_cast<const f32x4*>(p.Qb)[i];
  }
  bf16x8 bq[QREG ? C::KSTEPS : 1];
  if constexpr (QREG) {
#pragma unroll
    for (int t = 0; t < C::KSTEPS; ++t) bq[t] = reinterpret_cast<const bf16x8*>(p.Qb)[t * 64 + lane];
  }
  __syncthreads();
  // everything the prologue loaded is retired before the first counted wait (see the header comment)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  RG_DSTAMP(1);

  // ---- this wave's units: gw, gw + W, gw + 2 W, ... (filter: neighbouring waves read neighbouring 16 KiB), or -- the
  // bound pass -- one contiguous run, so that a wave stays inside one part of the range and keeps that part's running
  // maxima to itself (an atomic per unit would sit in the wave's vmcnt queue in front of every prefetched block)
  const int64_t W = (int64_t)gridDim.x * C::WAVES;
  const int64_t gw = (int64_t)blockIdx.x * C::WAVES + wave;
  const int64_t chunk = (p.nunits + W - 1) / W;
  const int64_t ubase = BOUND ? gw * chunk : gw;
  const int64_t ustride = BOUND ? 1 : W;
  const int64_t n_mine = BOUND ? (ubase < p.nunits ? (p.nunits - ubase < chunk ? p.nunits - ubase : chunk) : 0)
                               : (gw < p.nunits ? (p.nunits - gw + W - 1) / W : 0);
  uint2* wbuf = wbuf_all + wave * C::CAND_BUF;
  // bound pass: the wave's running maxima of the current part, one per query of the tile, in its (otherwise unused)
  // candidate buffer
  float* gm_lds = reinterpret_cast<float*>(wbuf);
  int cur_part = -1;  // wave-uniform
  if constexpr (BOUND) {
#pragma unroll
    for (int e = 0; e < 4; ++e) gm_lds[lane + 64 * e] = RG_NEG_INF;
  }
  auto flush_max = [&]() {
    if (cur_part >= 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int qq = lane + 64 * e;
        const float v = gm_lds[qq];
        if (qq < p.B && v > RG_NEG_INF) atomicMax(p.gmax + (int64_t)qq * p.ngroups + cur_part, f2ord(v));
        gm_lds[qq] = RG_NEG_INF;
      }
    }
  };
  int wcnt = 0;  // wave-uniform
  const unsigned lane16 = (unsigned)lane * 16u;

  const int sub = (int)(gw & (p.nsub - 1));  // this wave's sub-list of every query
  auto flush = [&]() {
    for (int i0 = 0; i0 < wcnt; i0 += 64) {
      const int i = i0 + lane;
      if (i < wcnt) {
        const uint2 e = wbuf[i];
        // (int8 entries: mask | query << 8 | (ceil(I / 256) << 1 | class) << 16 -- the lane's largest sum as an upper bound in
        // units of 256, and the class of the unit's granule)
        const int64_t q = I8 ? ((e.y >> 8) & 0xFFu) : (e.y >> 16);
        unsigned mk = I8 ? (e.y & 0xFFu) : (e.y & 0xFFFFu);
        int slot = atomicAdd(p.count + q * FILTER_COUNT_STRIDE + sub, __popc(mk));
        // retired here on every path: a returning atomic hipcc still considers pending where the flush rejoins the
        // unit loop would put its vmcnt(0) -- which also drains the prefetched unit -- in front of every group pass
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(slot) : : "memory");
        while (mk) {
          const int r = __ffs(mk) - 1;
          mk &= mk - 1;
          if (slot < p.subcap) {
            const int64_t at = q * p.cap + sub * p.subcap + slot;
            const int key = (int)e.x + (r & 3) + 16 * (r >> 2);
            // (scored lists carry (I << 1) | class; the entry's upper half is (ceil(I / 256) << 1) | class)
            if (I8 && p.scored) reinterpret_cast<int2*>(p.cand)[at] = make_int2(key, (((int)e.y >> 17) * 512) | (int)((e.y >> 16) & 1u));
            else p.cand[at] = key;
          }
          ++slot;
        }
      }
    }
    wcnt = 0;
  };

  f32x4 A0[16], A1[16];
  // sixteen 1-KiB blocks of unit `u`, this lane's 16 bytes of each: plain loads -- hipcc counts its own vmcnt waits
  // (block i of the current unit is needed when 15 - i younger loads of the unit and the next unit's 16 are outstanding)
#define RG_DLOAD(buf_, u_)                                                                                         \
  {                                                                                                                \
    const char* ub_ = reinterpret_cast<const char*>(p.Kb) + (uint64_t)(u_) * (C::UNIT_BLOCKS * 1024) + lane16;     \
    _Pragma("unroll") for (int b_ = 0; b_ < 16; ++b_)                                                              \
      buf_[b_] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ub_ + b_ * 1024));                      \
  }
#define RG_DWAIT(cnt_, reg_)

  // epilogue of a 32-key sub-tile against one group of 16 queries: a[h][r] = approximate score of key 16 h + 4 g + r of the
  // sub-tile for query 16 gq + j
  [[maybe_unused]] int cls_unit = 0;  // (int8) class of the granule the unit in process lies in (wave-uniform)
  auto epilogue = [&](const acc_t (&a)[2], int gq, int64_t unit, int sub, int part) {
#if defined(RG_DIRECT_ABL) && (RG_DIRECT_ABL & 1)   // timing build: no epilogue (results invalid); the scores stay live
    if (a[0][0] + a[1][3] == 123456) wbuf[0] = make_uint2(1u, 2u);
    return;
#endif
    if constexpr (I8) {  // integer sums against the integer threshold
      int m = a[0][0];
#pragma unroll
      for (int r = 1; r < 4; ++r) m = max(m, a[0][r]);
#pragma unroll
      for (int r = 0; r < 4; ++r) m = max(m, a[1][r]);
      const int th = __float_as_int(thr_lds[256 * cls_unit + 16 * gq + j]);
      if (__any(m >= th)) {
        // (the pass bits by subtract + v_alignbit, as the ring kernel's pass_mask: th clamped beyond any |I| < 2^23)
        unsigned mk = 0;
        const unsigned tm1 = (unsigned)(max(-(1 << 24), min(1 << 24, th)) - 1);
#pragma unroll
        for (int b = 7; b >= 0; --b) mk = __builtin_amdgcn_alignbit(mk, tm1 - (unsigned)a[b >> 2][b & 3], 31);
        const int64_t key_base = (unit * C::SUBS + sub) * 32 + 4 * g;
        if (key_base + 32 > p.key_end) {
          unsigned vm = 0;
#pragma unroll
          for (int r = 0; r < 8; ++r) vm |= (key_base + (r & 3) + 16 * (r >> 2) < p.key_end) ? (1u << r) : 0u;
          mk &= vm;
        }
        const unsigned long long bal = __ballot(mk != 0);
        if (bal) {
          const int pos = wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          if (mk) wbuf[pos] = make_uint2((unsigned)key_base, mk | ((unsigned)(16 * gq + j) << 8) |
                                                             ((unsigned)((((m + 255) >> 8) << 1) | cls_unit) << 16));
          wcnt += __popcll(bal);
        }
      }
      return;
    }
    float m = (float)a[0][0];  // (a chain, not a tree: hipcc folds it into v_max3_f32)
#pragma unroll
    for (int r = 1; r < 4; ++r) m = fmaxf(m, (float)a[0][r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, (float)a[1][r]);
    if constexpr (BOUND) {
      if (part != cur_part) {  // wave-uniform: the run crossed into the next part
        flush_max();
        cur_part = part;
      }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      if (g == 0) gm_lds[16 * gq + j] = fmaxf(gm_lds[16 * gq + j], m);
    } else {
      const float th = thr_lds[16 * gq + j];
      if (__any(m >= th)) {
        unsigned mk = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) mk |= ((float)a[h][r] >= th) ? (1u << (4 * h + r)) : 0u;
        const int64_t key_base = (unit * C::SUBS + sub) * 32 + 4 * g;  // the lane's keys: + r + 16 h  (mask bit 4 h + r)
        if (key_base + 32 > p.key_end) {
          unsigned vm = 0;
#pragma unroll
          for (int r = 0; r < 8; ++r) vm |= (key_base + (r & 3) + 16 * (r >> 2) < p.key_end) ? (1u << r) : 0u;
          mk &= vm;
        }
        if (p.thr.ablate == 2) mk = 0;  // (timing only: masks computed, candidates dropped)
        const unsigned long long bal = __ballot(mk != 0);
        if (bal) {
          const int pos = wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          if (mk) wbuf[pos] = make_uint2((unsigned)key_base, ((unsigned)(16 * gq + j) << 16) | mk);
          wcnt += __popcll(bal);
        }
      }
    }
  };

  // One unit against every query group.  The unit's sixteen blocks are k-step major (filter_common.h): block
  // (sub * KS32 + t) * 2 + h = k-step t of half h of sub-tile sub.  Up to 32 queries: the B operands are registers.
  // Beyond: they come from LDS, one 1-KiB fragment per (group, k-step) that feeds the 2 SUBS MFMAs of every half of the
  // unit -- 32 SUBS cycles of MFMA, less than an LDS round trip at D = 256 -- so the reads run LA steps ahead of the
  // MFMAs, across group boundaries (the groups are contiguous in LDS: read number r is simply KiB r of the image; the
  // last LA reads run past it into the candidate buffers and are never used), as asm loads with counted lgkmcnt waits
  // (hipcc's own schedule keeps one read ahead and idles the matrix pipe).
  const unsigned qaddr0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)qlds + lane16;
  auto process = [&](f32x4 (&A)[16], int64_t unit, [[maybe_unused]] unsigned cw, auto next_tag) {
    if constexpr (I8) {  // the unit's granule = unit / 2; its class word was requested with the unit's blocks
      // (settled here, in front of the counted LDS reads below)
      cls_unit = __builtin_amdgcn_readfirstlane((int)((cw >> ((unit >> 1) & 31)) & 1u));
    }
    int part0 = 0;
    if constexpr (BOUND) part0 = (int)(((unit - p.unit0) * C::SUBS) * p.ngroups / (p.nunits * C::SUBS));
    // (sub-tiles of one unit lie in one part or in two neighbouring ones: the division is per unit, not per sub-tile)
    auto part_of = [&](int sub) {
      if constexpr (!BOUND) return 0;
      if (C::SUBS == 1 || sub == 0) return part0;
      return (int)(((unit - p.unit0) * C::SUBS + sub) * p.ngroups / (p.nunits * C::SUBS));
    };
    if constexpr (QREG) {
      if constexpr (!BOUND) {
        if (wcnt > C::CAND_BUF - 128 * C::SUBS) flush();  // two groups x SUBS sub-tiles x <= 64 entries
      }
#pragma unroll
      for (int sub = 0; sub < C::SUBS; ++sub) {
        acc_t acc[2][2];  // [group][half]
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) acc[gq][0] = acc[gq][1] = acc_t{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < C::KS32; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
              if constexpr (I8)
                acc[gq][h] = __builtin_bit_cast(acc_t, __builtin_amdgcn_mfma_i32_16x16x64_i8(
                    __builtin_bit_cast(i32x4, A[(sub * C::KS32 + t) * 2 + h]), __builtin_bit_cast(i32x4, bq[gq * C::KS32 + t]),
                    __builtin_bit_cast(i32x4, acc[gq][h]), 0, 0, 0));
              else
                acc[gq][h] = __builtin_bit_cast(acc_t, __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, A[(sub * C::KS32 + t) * 2 + h]), bq[gq * C::KS32 + t],
                    __builtin_bit_cast(f32x4, acc[gq][h]), 0, 0, 0));
            }
          }
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) epilogue(acc[gq], gq, unit, sub, part_of(sub));
      }
    } else {
      f32x4 fr[C::LA];
#if defined(RG_DIRECT_ABL) && (RG_DIRECT_ABL & 2)   // timing build: no LDS fragment reads (results invalid)
#pragma unroll
      for (int t = 0; t < C::LA; ++t) fr[t] = f32x4{1.f, 2.f, 3.f, 4.f};
#define RG_BREAD(slot_, addr_, blk_) asm volatile("" : "+v"(fr[(slot_) % C::LA]))
#define RG_BWAIT(slot_) asm volatile("" : "+v"(fr[(slot_) % C::LA]))
#else
#define RG_BREAD(slot_, addr_, blk_) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(slot_) % C::LA]) : "v"(addr_), "n"((blk_) * 1024))
#define RG_BWAIT(slot_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[(slot_) % C::LA]) : "n"(C::LA - 1))
#endif
#pragma unroll
      for (int t = 0; t < C::LA; ++t) RG_BREAD(t, qaddr0, t);
      for (int gq = 0; gq < ngl; ++gq) {
        if constexpr (!BOUND) {
          if (wcnt > C::CAND_BUF - 64 * C::SUBS) flush();
        }
        const unsigned cur = qaddr0 + (unsigned)gq * C::GROUP_BYTES;
        acc_t acc[C::SUBS][2];
#pragma unroll
        for (int sub = 0; sub < C::SUBS; ++sub) acc[sub][0] = acc[sub][1] = acc_t{0, 0, 0, 0};
#define RG_GSTEP(t_)                                                                                            \
  if constexpr ((t_) < C::KS32) {                                                                               \
    RG_BWAIT(t_);                                                                                               \
    {                                                                                                           \
      _Pragma("unroll") for (int sub = 0; sub < C::SUBS; ++sub)                                                 \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                         \
          if constexpr (I8)                                                                                     \
            acc[sub][h] = __builtin_bit_cast(acc_t, __builtin_amdgcn_mfma_i32_16x16x64_i8(                      \
                __builtin_bit_cast(i32x4, A[(sub * C::KS32 + (t_)) * 2 + h]),                                   \
                __builtin_bit_cast(i32x4, fr[(t_) % C::LA]), __builtin_bit_cast(i32x4, acc[sub][h]), 0, 0, 0)); \
          else                                                                                                  \
            acc[sub][h] = __builtin_bit_cast(acc_t, __builtin_amdgcn_mfma_f32_16x16x32_bf16(                    \
                __builtin_bit_cast(bf16x8, A[(sub * C::KS32 + (t_)) * 2 + h]),                                  \
                __builtin_bit_cast(bf16x8, fr[(t_) % C::LA]), __builtin_bit_cast(f32x4, acc[sub][h]), 0, 0, 0)); \
        }                                                                                                       \
    }                                                                                                           \
    RG_BREAD(t_, cur, (t_) + C::LA);                                                                            \
  }
        RG_GSTEP(0) RG_GSTEP(1) RG_GSTEP(2) RG_GSTEP(3) RG_GSTEP(4) RG_GSTEP(5) RG_GSTEP(6) RG_GSTEP(7)
#undef RG_GSTEP
#pragma unroll
        for (int sub = 0; sub < C::SUBS; ++sub) epilogue(acc[sub], gq, unit, sub, part_of(sub));
      }
      // the LA reads issued past the last step land in registers nobody uses: retire them
#pragma unroll
      for (int t = 0; t < C::LA; ++t) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[t]));
#undef RG_BWAIT
#undef RG_BREAD
    }
  };

  // (int8) the class word of unit u's granule: a scalar load, requested a whole unit ahead like the blocks
  auto cls_word = [&](int64_t u) -> unsigned {
    if constexpr (I8) return p.thr.cls8[__builtin_amdgcn_readfirstlane((int)(u >> 6))];
    else return 0u;
  };
  if (n_mine > 0) {
    const int64_t u0 = p.unit0 + ubase;
    RG_DLOAD(A0, u0);
    unsigned w0 = cls_word(u0), w1 = 0u;
    int64_t i = 0;
    for (; i + 2 <= n_mine; i += 2) {  // pairs: A0 then A1, the other buffer's loads always in flight
      RG_DLOAD(A1, u0 + (i + 1) * ustride);
      w1 = cls_word(u0 + (i + 1) * ustride);
      process(A0, u0 + i * ustride, w0, std::true_type{});
      if (i + 2 < n_mine) {
        RG_DLOAD(A0, u0 + (i + 2) * ustride);
        w0 = cls_word(u0 + (i + 2) * ustride);
        process(A1, u0 + (i + 1) * ustride, w1, std::true_type{});
      } else {
        process(A1, u0 + (i + 1) * ustride, w1, std::false_type{});
      }
    }
    if (i < n_mine) process(A0, u0 + i * ustride, w0, std::false_type{});  // odd count: the last unit, nothing behind it
  }
  RG_DSTAMP(2);
  if constexpr (BOUND) {
    // The waves of a workgroup run through neighbouring units, so most of them end in the same part: their maxima are
    // combined in LDS and leave as ONE atomic per query and part -- a few hundred waves updating the same k words of a
    // query serialise in the L2 otherwise (8 - 17 us of a 25 us launch).
    int* wave_part = reinterpret_cast<int*>(thr_lds + 256);  // [WAVES], behind the thresholds
    if (lane == 0) wave_part[wave] = cur_part;
    __syncthreads();
    if (tid < 256 && tid < p.B) {
      const float* all = reinterpret_cast<const float*>(wbuf_all);
      float m = RG_NEG_INF;
      int part = -1;
      for (int w = 0; w < C::WAVES; ++w) {
        const int pw = wave_part[w];
        if (pw < 0) continue;
        if (pw != part) {
          if (part >= 0 && m > RG_NEG_INF) atomicMax(p.gmax + (int64_t)tid * p.ngroups + part, f2ord(m));
          part = pw;
          m = RG_NEG_INF;
        }
        m = fmaxf(m, all[(size_t)w * C::CAND_BUF * 2 + tid]);
      }
      if (part >= 0 && m > RG_NEG_INF) atomicMax(p.gmax + (int64_t)tid * p.ngroups + part, f2ord(m));
    }
  } else {
    flush();
  }
  RG_DSTAMP(3);
#undef RG_DWAIT
#undef RG_DLOAD
}

template <int D, bool QREG, bool BOUND, bool I8 = false>
static int launch_direct(const DirectParams& p, int grid, size_t lds, hipStream_t st) {
  static DeviceOnce lds_once;  // per device (common.h)
  if (hipError_t e = raise_dynamic_lds(lds_once, &topk_filter_direct_kernel<D, QREG, BOUND, I8>, 160 * 1024); e != hipSuccess) {
    set_error("topk_cosine_filtered(direct): cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    return RAGRAPH_EDEVICE;
  }
  hipLaunchKernelGGL((topk_filter_direct_kernel<D, QREG, BOUND, I8>), dim3((unsigned)grid), dim3(512), lds, st, p);
  RG_CHECK_LAUNCH("topk_cosine_filtered(direct filter)");
#ifdef RG_DIRECT_TIMING
  {
    (void)hipDeviceSynchronize();
    unsigned long long t[2][2][8];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_direct_t), sizeof(t));
    for (int b = 0; b < 2; ++b)
      fprintf(stderr, "[direct timing, %s pass, %s block of %d, 10 ns ticks] prologue %lld units %lld tail %lld (started %lld after block 0)\n",
              BOUND ? "bound" : "filter", b ? "last" : "first", grid, (long long)(t[BOUND][b][1] - t[BOUND][b][0]),
              (long long)(t[BOUND][b][2] - t[BOUND][b][1]), (long long)(t[BOUND][b][3] - t[BOUND][b][2]),
              (long long)(t[BOUND][b][0] - t[BOUND][0][0]));
  }
#endif
  return RAGRAPH_OK;
}

// The same launch on the int8 copy (a.i8: a.Kb = the int8 copy, a.Qb = the queries' int8 image): the geometry of D / 2.
template <int D>
static int launch_filter_direct_i8(const DirectArgs& a, hipStream_t st) {
  using C = DirectCfg<D / 2>;
  RG_REQUIRE(a.B >= 1 && a.B <= 256 && a.bound_groups == 0, RAGRAPH_EINVAL, "filter(direct, int8): bad batch / mode");
  RG_REQUIRE(a.key0 % C::UNIT_KEYS == 0 && a.key1 > a.key0, RAGRAPH_EINVAL, "filter(direct, int8): bad key range");
  DirectParams p{};
  p.Qb = a.Qb;
  p.Kb = a.Kb;
  p.B = a.B;
  p.unit0 = a.key0 / C::UNIT_KEYS;
  p.nunits = cdiv(a.key1 - a.key0, (int64_t)C::UNIT_KEYS);
  p.key_end = a.key1;
  p.thr = a.thr;
  p.count = a.count;
  p.cand = a.cand;
  p.cap = a.cap;
  p.nsub = a.nsub < 1 ? 1 : a.nsub;
  RG_REQUIRE((p.nsub & (p.nsub - 1)) == 0 && p.nsub <= FILTER_COUNT_STRIDE, RAGRAPH_EINVAL, "filter(direct): nsub=%d", p.nsub);
  p.subcap = a.cap / p.nsub;
  p.scored = a.scored;
  p.qgroups = 2 * (int)cdiv(a.B, 32);
  const int cus = device_cus_multiple_of_8();
  int64_t grid = cdiv(p.nunits, (int64_t)C::WAVES);
  if (grid > cus) grid = cus;
  const bool qreg = a.B <= 32;
  const size_t lds = C::lds_bytes(qreg ? 0 : p.qgroups);
  return qreg ? launch_direct<D, true, false, true>(p, (int)grid, lds, st) : launch_direct<D, false, false, true>(p, (int)grid, lds, st);
}

template <int D>
int launch_filter_direct(const DirectArgs& a, hipStream_t st) {
  using C = DirectCfg<D>;
  if (a.i8) return launch_filter_direct_i8<D>(a, st);   // (D = 64: the geometry of a 32-element bf16 row, one MFMA per half)
  RG_REQUIRE(a.B >= 1 && a.B <= 256, RAGRAPH_EINVAL, "filter(direct): B=%lld not in [1,256]", (long long)a.B);
  RG_REQUIRE(a.key0 % C::UNIT_KEYS == 0 && a.key1 > a.key0, RAGRAPH_EINVAL, "filter(direct): bad key range");
  DirectParams p{};
  p.Qb = a.Qb;
  p.Kb = a.Kb;
  p.B = a.B;
  p.unit0 = a.key0 / C::UNIT_KEYS;
  p.nunits = cdiv(a.key1 - a.key0, (int64_t)C::UNIT_KEYS);  // (the bank copy is padded to whole units)
  p.key_end = a.key1;
  p.thr = a.thr;
  p.count = a.count;
  p.cand = a.cand;
  p.cap = a.cap;
  p.nsub = a.nsub < 1 ? 1 : a.nsub;
  RG_REQUIRE((p.nsub & (p.nsub - 1)) == 0 && p.nsub <= FILTER_COUNT_STRIDE, RAGRAPH_EINVAL, "filter(direct): nsub=%d", p.nsub);
  p.subcap = a.cap / p.nsub;
  p.gmax = a.gmax_out;
  p.ngroups = a.bound_groups;
  p.qgroups = 2 * (int)cdiv(a.B, 32);
  const bool bound = a.bound_groups > 0;
  if (bound)
    RG_REQUIRE(p.nunits * C::SUBS >= a.bound_groups, RAGRAPH_EINVAL, "filter(direct): bound range shorter than its parts");
  const int cus = device_cus_multiple_of_8();
  int64_t grid = cdiv(p.nunits, (int64_t)C::WAVES);
  if (grid > cus) grid = cus;
  const bool qreg = a.B <= 32;
  const size_t lds = C::lds_bytes(qreg ? 0 : p.qgroups);
  if (qreg) return bound ? launch_direct<D, true, true>(p, (int)grid, lds, st) : launch_direct<D, true, false>(p, (int)grid, lds, st);
  return bound ? launch_direct<D, false, true>(p, (int)grid, lds, st) : launch_direct<D, false, false>(p, (int)grid, lds, st);
}

template int launch_filter_direct<64>(const DirectArgs&, hipStream_t);
template int launch_filter_direct<128>(const DirectArgs&, hipStream_t);
template int launch_filter_direct<256>(const DirectArgs&, hipStream_t);

}  // namespace ragraph
