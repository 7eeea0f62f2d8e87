// bf16 filter of the exact cosine top-k for UP TO 256 QUERIES (SimilarityFunctions.py:6-16 + ToyGraphBase.py:66-67 at the
// reference's real batch sizes: graph classification sends 1 query per forward, RAGraph_node a few hundred).
//
// With so few queries the score matrix is cheap (2 * 256 * N * D flops = 52 us of bf16 MFMA at N = 1M, D = 256) and the call
// is bound by ONE pass over the bf16 bank copy (2 N D bytes): the kernel is built around that stream, not around the
// matrix cores.
//   * The bank copy is stored in MFMA fragment order (filter_common.h): a wave fetches the A operand of a k-step with one
//     coalesced 1-KiB global_load_dwordx4 straight into the registers the MFMA reads.  No LDS staging, no ring, no
//     barrier, no hand-over between waves: every wave streams its own 16-KiB units (512 key-elements x 16 blocks), the
//     next unit's sixteen loads in flight while the current one is multiplied -- 8 waves x 16 KiB = 128 KiB in flight per
//     CU.  Units are dealt round-robin over all waves of the grid, so neighbouring waves read neighbouring 16 KiB.
//   * The queries are the B operands, converted once per call to bf16 in fragment order too (1 KiB per k-step and group
//     of 32; filter_prep_kernel).  Up to 32 of them live in registers (D/4 VGPRs); 33..256 are copied into LDS: every
//     unit is multiplied with each group in turn, one conflict-free ds_read_b128 per MFMA (128 B/clk/CU, half the LDS
//     rate).  The key stream is thus read ONCE for all groups.
//   * Epilogue as in the ring kernel: one wave-uniform test of the accumulators' maxima per 32 keys x 32 queries; passing
//     keys leave through a wave-private LDS buffer and reach the per-query candidate lists in flushes.
//   * BOUND mode: no thresholds, no candidates -- per query the best approximate score of each of `ngroups` consecutive
//     parts of the key range (atomicMax on order-preserving ints), from which the first lower bound of the k-th best
//     score is made (filter_threshold).
// The loads are inline asm (hipcc would otherwise serialise them behind its own waitcnt bookkeeping) and are waited for
// with counted s_waitcnt vmcnt: block i of the current unit is needed when (15 - i) younger loads of the unit and, if a
// next unit exists, its 16 loads may still be outstanding.  Any other vector-memory operation the compiler issues in
// between is younger than the awaited load or older than all of them, so it can only make a wait stricter.
#include "filter_common.h"
#include <type_traits>

namespace ragraph {

template <int D_>
struct DirectCfg {
  static constexpr int D = D_;
  static constexpr int WAVES = 8, THREADS = 512;
  static constexpr int KSTEPS = D / 16;              // MFMA k-steps (1-KiB blocks) per 32-key sub-tile: 16 / 8 / 4
  static constexpr int UNIT_BLOCKS = 16;             // blocks per unit = 16 KiB in flight per wave and buffer
  static constexpr int SUBS = UNIT_BLOCKS / KSTEPS;  // sub-tiles per unit: 1 / 2 / 4
  static constexpr int UNIT_KEYS = 32 * SUBS;        // 32 / 64 / 128
  static constexpr int CAND_BUF = SUBS == 1 ? 256 : 512;  // entries (8 B) of a wave's candidate buffer; one group pass
                                                         // over a unit pushes at most 64 SUBS of them
  static constexpr int GROUP_BYTES = KSTEPS * 1024;  // one group of 32 queries as bf16 B operands
  static constexpr size_t lds_bytes(int groups_in_lds) {
    return (size_t)groups_in_lds * GROUP_BYTES + (size_t)WAVES * CAND_BUF * 8 + 256 * sizeof(float) + WAVES * sizeof(int);
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
  int* gmax;              // BOUND
  int ngroups;            // BOUND: parts of the range
  int qgroups;            // groups of 32 queries (1..8)
};

// QREG: <= 32 queries, B operands in registers; else `qgroups` groups in LDS.
template <int D, bool QREG, bool BOUND>
__global__ void __launch_bounds__(512, 2) topk_filter_direct_kernel(DirectParams p) {
  using C = DirectCfg<D>;
  extern __shared__ float4 dsmem4[];
  char* smem = reinterpret_cast<char*>(dsmem4);
  const int ngl = QREG ? 0 : p.qgroups;
  char* qlds = smem;                                                             // [ngl][KSTEPS][64] x 16 B
  uint2* wbuf_all = reinterpret_cast<uint2*>(smem + (size_t)ngl * C::GROUP_BYTES);  // [WAVES][CAND_BUF]
  float* thr_lds = reinterpret_cast<float*>(wbuf_all + C::WAVES * C::CAND_BUF);     // [256]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, g = lane >> 5;

  // ---- thresholds (one per query of the tile) and, beyond 32 queries, the B operands in LDS -------------------------
  if (tid < 256) {
    float t = __builtin_huge_valf();  // padded queries never pass
    if (!BOUND && tid < p.B) t = filter_threshold(p.thr, tid);
    thr_lds[tid] = t;
  }
  // the B operands: filter_prep_kernel left them in HBM as bf16 in fragment order, so a workgroup copies its image
  // into LDS linearly (33..256 queries) or a wave takes its 64 VGPRs straight from it (<= 32 queries)
  if constexpr (!QREG) {
    const int pieces = ngl * C::KSTEPS * 64;
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
        const int64_t q = e.y >> 16;
        unsigned mk = e.y & 0xFFFFu;
        int slot = atomicAdd(p.count + q * FILTER_COUNT_STRIDE + sub, __popc(mk));
        // retired here on every path: a returning atomic hipcc still considers pending where the flush rejoins the
        // unit loop would put its vmcnt(0) -- which also drains the prefetched unit -- in front of every group pass
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(slot) : : "memory");
        while (mk) {
          const int r = __ffs(mk) - 1;
          mk &= mk - 1;
          if (slot < p.subcap) p.cand[q * p.cap + sub * p.subcap + slot] = (int)e.x + (r & 3) + 8 * (r >> 2);
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

  // epilogue of a 32-key sub-tile against one query group: acc[r] = approximate score of key row (r & 3) + 8 (r >> 2) + 4 g
  auto epilogue = [&](const f32x16& acc, int gq, int64_t unit, int sub, int part) {
    float m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
    if constexpr (BOUND) {
      if (part != cur_part) {  // wave-uniform: the run crossed into the next part
        flush_max();
        cur_part = part;
      }
      m = fmaxf(m, __shfl_xor(m, 32));
      if (g == 0) gm_lds[32 * gq + j] = fmaxf(gm_lds[32 * gq + j], m);
    } else {
      const float th = thr_lds[32 * gq + j];
      if (__any(m >= th)) {
        unsigned mk = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) mk |= (acc[r] >= th) ? (1u << r) : 0u;
        const int64_t key_base = (unit * C::SUBS + sub) * 32 + 4 * g;  // the lane's keys: + (r & 3) + 8 (r >> 2)
        if (key_base + 28 >= p.key_end) {
          unsigned vm = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) vm |= (key_base + (r & 3) + 8 * (r >> 2) < p.key_end) ? (1u << r) : 0u;
          mk &= vm;
        }
        if (p.thr.ablate == 2) mk = 0;  // (timing only: masks computed, candidates dropped)
        const unsigned long long bal = __ballot(mk != 0);
        if (bal) {
          const int pos = wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          if (mk) wbuf[pos] = make_uint2((unsigned)key_base, ((unsigned)(32 * gq + j) << 16) | mk);
          wcnt += __popcll(bal);
        }
      }
    }
  };

  // One unit against every query group.  The first group's pass carries the counted vmcnt waits: block i is needed
  // when the unit's 15 - i younger loads and, if a next unit is in flight, its 16 may still be outstanding; the other
  // groups find the unit in registers.  Beyond 32 queries the B operands come from LDS, and a k-step is only 32 cycles
  // of MFMA -- less than an LDS round trip -- so their reads run FOUR steps ahead of the MFMAs, across group boundaries,
  // as asm loads with counted lgkmcnt waits (hipcc's own schedule keeps one read ahead and idles the matrix pipe).
  const unsigned qaddr0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)qlds + lane16;
  auto process = [&](f32x4 (&A)[16], int64_t unit, auto next_tag) {
    [[maybe_unused]] constexpr int BEHIND = decltype(next_tag)::value ? 16 : 0;
    int part0 = 0;
    if constexpr (BOUND) part0 = (int)(((unit - p.unit0) * C::SUBS) * p.ngroups / (p.nunits * C::SUBS));
    // (sub-tiles of one unit lie in one part or in two neighbouring ones: the division is per unit, not per sub-tile)
    auto part_of = [&](int sub) {
      if constexpr (!BOUND) return 0;
      if (C::SUBS == 1 || sub == 0) return part0;
      return (int)(((unit - p.unit0) * C::SUBS + sub) * p.ngroups / (p.nunits * C::SUBS));
    };
    f32x16 acc;
    if constexpr (QREG) {
#define RG_STEP(i_)                                                                                             \
  {                                                                                                             \
    if constexpr ((i_) % C::KSTEPS == 0) {                                                                      \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.f;                                              \
    }                                                                                                           \
    RG_DWAIT(BEHIND + 15 - (i_), A[i_]);                                                                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[i_]), bq[(i_) % C::KSTEPS], acc, 0, 0, 0); \
    if constexpr ((i_) % C::KSTEPS == C::KSTEPS - 1) epilogue(acc, 0, unit, (i_) / C::KSTEPS, part_of((i_) / C::KSTEPS)); \
  }
      if constexpr (!BOUND) {
        if (wcnt > C::CAND_BUF - 64 * C::SUBS) flush();
      }
      RG_STEP(0) RG_STEP(1) RG_STEP(2) RG_STEP(3) RG_STEP(4) RG_STEP(5) RG_STEP(6) RG_STEP(7)
      RG_STEP(8) RG_STEP(9) RG_STEP(10) RG_STEP(11) RG_STEP(12) RG_STEP(13) RG_STEP(14) RG_STEP(15)
#undef RG_STEP
    } else {
      f32x4 fr[4];
      // step i of a group pass reads B block i % KSTEPS of the group; the read of step i + 4 is issued behind MFMA i --
      // for the last four steps that is the NEXT group's block (the last group reads group 0's again: unused)
#define RG_BREAD(slot_, addr_, blk_) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(slot_)&3]) : "v"(addr_), "n"(((blk_) % C::KSTEPS) * 1024))
#define RG_BWAIT(cnt_, slot_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[(slot_)&3]) : "n"(cnt_))
#define RG_GSTEP(i_, FIRST_)                                                                                    \
  {                                                                                                             \
    if constexpr ((i_) % C::KSTEPS == 0) {                                                                      \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.f;                                              \
    }                                                                                                           \
    RG_BWAIT(3, i_);                                                                                            \
    if constexpr (FIRST_) RG_DWAIT(BEHIND + 15 - (i_), A[i_]);                                                  \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[i_]),                            \
                                                  __builtin_bit_cast(bf16x8, fr[(i_)&3]), acc, 0, 0, 0);        \
    if constexpr ((i_) + 4 < 16) RG_BREAD(i_, cur, (i_) + 4);                                                   \
    else RG_BREAD(i_, nxt, (i_) + 4 - 16);                                                                      \
    if constexpr ((i_) % C::KSTEPS == C::KSTEPS - 1) epilogue(acc, gq, unit, (i_) / C::KSTEPS, part_of((i_) / C::KSTEPS)); \
  }
#define RG_GPASS(FIRST_)                                                                                        \
  RG_GSTEP(0, FIRST_) RG_GSTEP(1, FIRST_) RG_GSTEP(2, FIRST_) RG_GSTEP(3, FIRST_) RG_GSTEP(4, FIRST_)           \
  RG_GSTEP(5, FIRST_) RG_GSTEP(6, FIRST_) RG_GSTEP(7, FIRST_) RG_GSTEP(8, FIRST_) RG_GSTEP(9, FIRST_)           \
  RG_GSTEP(10, FIRST_) RG_GSTEP(11, FIRST_) RG_GSTEP(12, FIRST_) RG_GSTEP(13, FIRST_) RG_GSTEP(14, FIRST_)      \
  RG_GSTEP(15, FIRST_)
      unsigned cur = qaddr0, nxt = ngl > 1 ? qaddr0 + C::GROUP_BYTES : qaddr0;
      RG_BREAD(0, cur, 0);
      RG_BREAD(1, cur, 1);
      RG_BREAD(2, cur, 2);
      RG_BREAD(3, cur, 3);
      {
        const int gq = 0;
        if constexpr (!BOUND) {
          if (wcnt > C::CAND_BUF - 64 * C::SUBS) flush();
        }
        RG_GPASS(true)
      }
      for (int gq = 1; gq < ngl; ++gq) {
        cur = nxt;
        nxt = gq + 1 < ngl ? cur + C::GROUP_BYTES : qaddr0;
        if constexpr (!BOUND) {
          if (wcnt > C::CAND_BUF - 64 * C::SUBS) flush();
        }
        RG_GPASS(false)
      }
      // the four reads issued past the last step land in registers nobody uses: retire them
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]));
#undef RG_GPASS
#undef RG_GSTEP
#undef RG_BWAIT
#undef RG_BREAD
    }
  };

  if (n_mine > 0) {
    const int64_t u0 = p.unit0 + ubase;
    RG_DLOAD(A0, u0);
    int64_t i = 0;
    for (; i + 2 <= n_mine; i += 2) {  // pairs: A0 then A1, the other buffer's loads always in flight
      RG_DLOAD(A1, u0 + (i + 1) * ustride);
      process(A0, u0 + i * ustride, std::true_type{});
      if (i + 2 < n_mine) {
        RG_DLOAD(A0, u0 + (i + 2) * ustride);
        process(A1, u0 + (i + 1) * ustride, std::true_type{});
      } else {
        process(A1, u0 + (i + 1) * ustride, std::false_type{});
      }
    }
    if (i < n_mine) process(A0, u0 + i * ustride, std::false_type{});  // odd count: the last unit, nothing behind it
  }
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
#undef RG_DWAIT
#undef RG_DLOAD
}

template <int D, bool QREG, bool BOUND>
static int launch_direct(const DirectParams& p, int grid, size_t lds, hipStream_t st) {
  static DeviceOnce lds_once;  // per device (common.h)
  if (hipError_t e = raise_dynamic_lds(lds_once, &topk_filter_direct_kernel<D, QREG, BOUND>, 160 * 1024); e != hipSuccess) {
    set_error("topk_cosine_filtered(direct): cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    return RAGRAPH_EDEVICE;
  }
  hipLaunchKernelGGL((topk_filter_direct_kernel<D, QREG, BOUND>), dim3((unsigned)grid), dim3(512), lds, st, p);
  RG_CHECK_LAUNCH("topk_cosine_filtered(direct filter)");
  return RAGRAPH_OK;
}

template <int D>
int launch_filter_direct(const DirectArgs& a, hipStream_t st) {
  using C = DirectCfg<D>;
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
  p.qgroups = (int)cdiv(a.B, 32);
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
