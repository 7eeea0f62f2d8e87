// The ring kernel of the filtered top-k (topk_filter_kernel: LDS-DMA key ring, bf16 / int8 MFMA, bound pass, candidate and scored lists, pipelined epilogue).
// Part of csrc/topk_filter.hip (textually included there, inside its namespace / after its helpers): split out in round 6 so
// that the ring, the candidate path and the launch plumbing can be read -- and changed -- apart.  No include guard on purpose:
// these are not stand-alone headers.

#ifdef RG_RING_STAMPS  // diagnostic build only: wall-clock stamps (10 ns ticks) through the first segment of workgroup 0 and
                       // of the last workgroup: entry, operands loaded, thresholds ready, ring primed, stages done, flushed
__device__ unsigned long long g_ring_t[2][2][8];
__device__ unsigned long long g_ring_span[2][2];   // [BOUND][earliest entry, latest exit] over all workgroups
__device__ unsigned long long g_ring_max[2][8];    // [BOUND][phase]: the longest phase over all workgroups' first segments
#define RG_RSTAMP(i_)                                                                                     \
  if (threadIdx.x == 0 && first_seg) {                                                                    \
    const unsigned long long now_ = wall_clock64();                                                       \
    if (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) g_ring_t[BOUND][blockIdx.x != 0][i_] = now_;      \
    if ((i_) == 0) atomicMin(&g_ring_span[BOUND][0], now_);                                               \
    else atomicMax(&g_ring_max[BOUND][i_], now_ - rs_prev);                                               \
    rs_prev = now_;                                                                                       \
  }
#else
#define RG_RSTAMP(i_)
#endif
#ifdef RG_TOPK_TIMING  // diagnostic build only: per-wave cycle totals of the ring's phases
__device__ unsigned long long g_filter_timing[8];
#define RG_FT(var_) const unsigned long long var_ = __builtin_amdgcn_s_memtime()
#else
#define RG_FT(var_)
#endif

// QW = queries per wave: 64 (four groups of 16 sharing every A fragment; query tile = 512), 32 (tile = 256) or 128
// (D = 64, long streams: tile = 1024).
// BOUND: no thresholds, no candidates -- the launch only records, per query, the best approximate score of each of
// p.ngroups consecutive parts of its key range (filter_prepare_kernel turns them into the first lower bound).
// I8: the level runs on the int8 copy (filter_common.h): the ring geometry of a bf16 bank of D / 2 elements (a key is D
// bytes), v_mfma_i32_16x16x64_i8, integer thresholds; the queries are quantised from the normalised fp32 rows here.
// SCORED (int8 levels of large calls): a list entry is {key, I} -- the integer sum that admitted the key (for a lane with two
// passing keys of one query: the larger of the two for both, an upper bound) -- in an int2 list of p.cap entries; the
// rescoring (topk_rescore_scored_kernel) then scores the most promising entries first and never fetches the rows of
// those whose I cannot reach the exact k-th best found that way.
// PIPE (int8 levels at D = 256): the epilogue of sub-tile u runs INSIDE the MFMA stream of sub-tile u + 1 -- two sets of
// accumulators, the maxima of one query group after each of the next sub-tile's first steps, the candidate path behind them --
// so a wave's vector work sits beside its OWN matrix work instead of waiting for the SIMD partner to be in the other phase.
#ifndef RG_RING_FOLD   // (-DRG_RING_FOLD=0: the D = 64 int8 levels without the folded thresholds -- A/B builds)
#define RG_RING_FOLD 1
#endif
template <int D, int QW, bool BOUND = false, bool I8 = false, bool SCORED = false, bool PIPE = false>
__global__ void __launch_bounds__(512, 2) topk_filter_kernel(FilterParams p) {
  using C = FilterCfg<I8 ? D / 2 : D>;
  static_assert(!PIPE || (I8 && !BOUND && C::KSTEPS >= QW / 16 + 2), "PIPE: int8 filter levels, one step per query group + 2");
  static_assert(!(I8 && BOUND), "the bound pass runs on the bf16 copy");
  static_assert(I8 || !SCORED, "scored lists carry the int8 levels' integer sums");
  static_assert(QW == 32 || QW == 64 || QW == 96 || QW == 128, "two, four, six or eight query groups of 16 per wave");
  constexpr int QT = C::WAVES * QW;
  constexpr int NG = QW / 16;  // query groups per wave: each A fragment (16 keys x 32 elements) feeds NG MFMAs
  // FOLD (int8 levels without the pipelined epilogue): a sub-tile's accumulators start at -T instead of 0 -- the MFMA adds
  // its integer sums to them exactly -- so "does any score reach its query's threshold" is ONE sign test over the maxima of all
  // groups instead of a compare per group: a third fewer vector instructions on the path every sub-tile takes.
  // The start values are the MFMAs' C operands straight from registers (a quad of -T per group, rebuilt when the stage's
  // class changes): four more registers per group and no instruction -- so only where it pays: D = 64, whose sub-tiles are
  // two MFMAs per group against the same epilogue (4096 x 4M x 64: 0.935 -> 0.913 ms, 65 536: 13.99 -> 13.86; at D = 128 the
  // same change measured +- 0; D = 256 with six groups has no registers to spare: the quads spilled, and start values moved
  // into the accumulators by v_mov cost four times what the fold saves).  tools/gpu_fold_ab.sh
  constexpr bool FOLD = RG_RING_FOLD && I8 && !PIPE && !BOUND && D == 64;
  extern __shared__ float4 fsmem4[];
  char* smem = reinterpret_cast<char*>(fsmem4);
  unsigned* full = reinterpret_cast<unsigned*>(smem + C::SLOTS * C::STAGE_BYTES);  // [SLOTS] then freec [SLOTS]
  unsigned* freec = full + C::SLOTS;
  const unsigned lds_base = (unsigned)(size_t)(lds_void_f*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;

  // DMA: instruction i of wave w copies the stage's (DMAS w + i)-th 1-KiB block -- one k-step of half a 32-key sub-tile in
  // fragment order -- to the same offset of the ring slot; lane l moves bytes [16 l, 16 l + 16) of it.  The LDS image IS
  // the HBM image, and a k-step's A operand is one ds_read_b128 at 16 l: consecutive lanes, consecutive pieces, no
  // bank conflicts and no swizzle.
  unsigned voff[C::DMAS];
#pragma unroll
  for (int i = 0; i < C::DMAS; ++i) voff[i] = (unsigned)(i * 1024 + lane * 16);
  auto dma_stage = [&](int64_t stage_abs, int slot) {  // stage_abs: stage index over the whole bank
    // (wave-uniform by construction; the readfirstlanes keep it in SGPRs whatever hipcc's divergence analysis makes of
    // the loop around it)
    const uint64_t goff = (uint64_t)stage_abs * C::STAGE_BYTES + (uint64_t)(C::DMAS * wave * 1024);
    const char* gbase = reinterpret_cast<const char*>(p.Kb) +
                        (((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(goff >> 32)) << 32) |
                         (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)goff));  // (returns int: no sign extension)
#pragma unroll
    for (int i = 0; i < C::DMAS; ++i) {
      const unsigned dst = lds_base + (unsigned)(slot * C::STAGE_BYTES + (C::DMAS * wave + i) * 1024);
      unsigned keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(voff[i]), "s"(dst), "s"(gbase)
          : "memory");
    }
  };
  // A fragment of step n of a stage (sub-tile n / KSTEPS, k-step n % KSTEPS): block n of the slot, this lane's 16 bytes
  const unsigned apos = lds_base + (unsigned)lane * 16u;

  const int x = p.xcd_map ? (int)(blockIdx.x & 7) : 0;
  const int64_t nq = p.xcd_map ? ((p.qtiles - x + 7) >> 3) : p.qtiles;
  const int64_t nq0 = p.xcd_map ? ((p.qtiles + 7) >> 3) : p.qtiles;
  SegmentWalker walker(nq, p.nstages_total, p.wgs_per_group, p.lb_min, 0, p.depth[nq != nq0],
                       p.xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
  Segment seg;
  [[maybe_unused]] bool first_seg = true;
#ifdef RG_RING_STAMPS
  unsigned long long rs_prev = 0;
#endif
  RG_RSTAMP(0);
  while (walker.next(seg)) {
    const int64_t qtile = p.xcd_map ? x + 8 * seg.tile : seg.tile;
    const int64_t q_lo = qtile * QT + wave * QW + j;  // group gq's query: q_lo + 16 gq
    const int64_t st0 = seg.st0;
    const int nstages = (int)(seg.st1 - seg.st0);

    // ---- B operands: group gq's query q_lo + 16 gq, k-step t = elements 32 t + 8 g .. + 7, converted to bf16 (RNE) ----
    // (int8 levels: elements 64 t + 16 g .. + 15, quantised with the query's scale exactly as filter_prep_kernel did)
    bf16x8 bq[I8 ? 1 : NG][I8 ? 1 : C::KS32];
    i32x4 bqi[I8 ? NG : 1][I8 ? C::KS32 : 1];
    constexpr int TB = C::KS32 < 4 ? C::KS32 : 4;
    if constexpr (I8) {
      if (p.Qb) {  // prepared int8 image (filter_prep_kernel, up to FILTER_QB_MAX_B queries): block (group * KS32 + t), this
                   // lane's 16 bytes -- the very bytes the quantisation below would produce; groups beyond the batch: zeros
        const int64_t qg0 = (qtile * QT + wave * QW) >> 4;
        const int64_t ngroups16 = ((p.B + 31) / 32 * 32) >> 4;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
          const bool have = qg0 + gq < ngroups16;  // (wave-uniform)
          const i32x4* src = reinterpret_cast<const i32x4*>(p.Qb) + ((qg0 + gq) * C::KS32) * 64 + lane;
#pragma unroll
          for (int t = 0; t < C::KS32; ++t) {
            i32x4 z = {0, 0, 0, 0};
            bqi[gq][t] = have ? src[t * 64] : z;
          }
        }
      } else
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        const int64_t qq = q_lo + 16 * gq;
        const int64_t qr = qq < p.B ? qq : p.B - 1;
        const float sq = p.thr.qscale[qr];
        const float inv_sq = sq > 0.f ? 1.f / sq : 0.f;
        const float* r0 = p.Qn + qr * D + 16 * g;
#pragma unroll
        for (int t = 0; t < C::KS32; ++t) {
          float4 u[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) u[c] = *reinterpret_cast<const float4*>(r0 + 64 * t + 4 * c);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            unsigned w = 0u;
            if (sq > 0.f)
              w = ((unsigned)quantize_i8(u[c].x, inv_sq) & 0xFFu) | (((unsigned)quantize_i8(u[c].y, inv_sq) & 0xFFu) << 8) |
                  (((unsigned)quantize_i8(u[c].z, inv_sq) & 0xFFu) << 16) | (((unsigned)quantize_i8(u[c].w, inv_sq) & 0xFFu) << 24);
            bqi[gq][t][c] = (int)w;
          }
          asm volatile("" : "+v"(bqi[gq][t]));
        }
        asm volatile("" ::: "memory");
      }
    } else if (p.Qb) {  // prepared image: block (group * KS32 + t), this lane's 16 bytes; groups beyond the padded batch: zeros
      const int64_t qg0 = (qtile * QT + wave * QW) >> 4;
      const int64_t ngroups16 = ((p.B + 31) / 32 * 32) >> 4;
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        const bool have = qg0 + gq < ngroups16;  // (wave-uniform)
        const bf16x8* src = reinterpret_cast<const bf16x8*>(p.Qb) + ((qg0 + gq) * C::KS32) * 64 + lane;
#pragma unroll
        for (int t = 0; t < C::KS32; ++t) {
          bf16x8 z;
#pragma unroll
          for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
          bq[gq][t] = have ? src[t * 64] : z;
        }
      }
    } else
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
      const int64_t qq = q_lo + 16 * gq;
      const float* r0 = p.Qn + (qq < p.B ? qq : p.B - 1) * D + 8 * g;
#pragma unroll
      for (int t0 = 0; t0 < C::KS32; t0 += TB) {  // batches of (up to) 4 steps = 8 float4 in flight
#pragma unroll
        for (int t = t0; t < t0 + TB; ++t) {
          const float4 u0 = *reinterpret_cast<const float4*>(r0 + 32 * t), u1 = *reinterpret_cast<const float4*>(r0 + 32 * t + 4);
          bq[gq][t][0] = (__bf16)u0.x; bq[gq][t][1] = (__bf16)u0.y; bq[gq][t][2] = (__bf16)u0.z; bq[gq][t][3] = (__bf16)u0.w;
          bq[gq][t][4] = (__bf16)u1.x; bq[gq][t][5] = (__bf16)u1.y; bq[gq][t][6] = (__bf16)u1.z; bq[gq][t][7] = (__bf16)u1.w;
        }
#pragma unroll
        for (int t = t0; t < t0 + TB; ++t) asm volatile("" : "+v"(bq[gq][t]));
        asm volatile("" ::: "memory");
      }
    }
    RG_RSTAMP(1);
    // padded queries never pass: +inf threshold
    float thr[NG];
    int thr8[NG], thr8h[NG];  // (int8 levels) the integer thresholds: keys of NORMAL / of HEAVY granules (filter_common.h)
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
      thr[gq] = (!BOUND && !I8 && q_lo + 16 * gq < p.B) ? filter_threshold(p.thr, q_lo + 16 * gq) : __builtin_huge_valf();
      thr8[gq] = (I8 && q_lo + 16 * gq < p.B) ? filter_threshold_i8(p.thr, q_lo + 16 * gq, 0) : INT_MAX;
      thr8h[gq] = (I8 && q_lo + 16 * gq < p.B) ? filter_threshold_i8(p.thr, q_lo + 16 * gq, 1) : INT_MAX;
    }
    // bound pass: running maxima of the current group (group g = stages [ceil(g n / G), ceil((g+1) n / G)) of the range)
    float gm[NG];
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) gm[gq] = RG_NEG_INF;
    int grp = 0;
    int64_t grp_end = 0;  // first stage (range-relative) of the next group
    auto group_of = [&](int64_t t) { return (int)(t * p.ngroups / p.nstages_total); };  // largest g with ceil(g n / G) <= t
    auto flush_max = [&]() {
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        // the four lanes j + 16 g of a query hold the maxima of its keys 4 g .. 4 g + 3 (mod 16): one atomic per query,
        // not four on one address in one instruction
        float v = gm[gq];
        v = fmaxf(v, __shfl_xor(v, 16));
        v = fmaxf(v, __shfl_xor(v, 32));
        if (g == 0 && q_lo + 16 * gq < p.B) atomicMax(p.gmax + (q_lo + 16 * gq) * p.ngroups + grp, f2ord(v));
        gm[gq] = RG_NEG_INF;
      }
    };
    if constexpr (BOUND) {
      grp = group_of(st0);
      grp_end = ((int64_t)(grp + 1) * p.nstages_total + p.ngroups - 1) / p.ngroups;
    }
    // Candidates: a sub-tile that holds any (one wave-uniform test of the accumulators' maxima) turns each lane's 8 scores
    // per query group (keys 4 g + r of both halves against query j of the group) into a pass MASK without a branch, and the lanes with a non-zero mask push one 8-byte entry
    // {(query within the wave) << 26 | offset of the lane's key group from key_org, mask} into a wave-private LDS buffer
    // (position by ballot + mbcnt).  No atomics and no memory wait inside the MFMA stream, ~100 VALU instructions that
    // fit under the other wave's MFMAs.  A full buffer, and the end of the segment, flush the entries to the queries'
    // lists in global memory: one returning atomic per entry (64 entries per round trip), then the keys of its mask.
    // key_org moves up (after a flush) every 2^16 stages so that offsets stay inside 26 bits.
    uint2* wbuf = reinterpret_cast<uint2*>(smem + C::SLOTS * C::STAGE_BYTES + 64) + wave * C::CAND_BUF;
    const int64_t q_wave = qtile * QT + wave * QW;
    const bool wave_live = q_wave < p.B;  // (wave-uniform)
    int wcnt = 0;  // wave-uniform
    int key_org = (int)((p.stage_base + st0) * C::STAGE_KEYS);
#ifdef RG_TOPK_TIMING
    unsigned long long tfl = 0, nfl = 0;
#endif
    auto flush = [&]() {
#ifdef RG_TOPK_TIMING
      const unsigned long long tf0 = __builtin_amdgcn_s_memtime();
      nfl += wcnt > 0 ? 1 : 0;
#endif
      for (int i0 = 0; i0 < wcnt; i0 += 64) {
        const int i = i0 + lane;
        if (i < wcnt) {
          const uint2 e = wbuf[i];
          const int64_t q = q_wave + (e.x >> 25);
          const int key0 = key_org + (int)(e.x & 0x1FFFFFFu);
          unsigned mk = SCORED ? (e.y & 0xFFu) : e.y;
          int slot = atomicAdd(p.count + q * p.cstride, __popc(mk));
          // retired here on every path: a return hipcc still considers pending where the flush rejoins the stage loop
          // would put its vmcnt(0) -- which also drains the DMA ring -- in front of every sub-tile
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(slot) : : "memory");
          while (mk) {
            const int r = __ffs(mk) - 1;
            mk &= mk - 1;
            if constexpr (SCORED) {
              if (slot < p.cap) reinterpret_cast<int2*>(p.cand)[q * p.cap + slot] = make_int2(key0 + (r & 3) + 16 * (r >> 2), (int)e.y >> 8);
            } else {
              if (slot < p.cap) p.cand[q * p.cap + slot] = key0 + (r & 3) + 16 * (r >> 2);
            }
            ++slot;
          }
        }
      }
      wcnt = 0;
#ifdef RG_TOPK_TIMING
      tfl += __builtin_amdgcn_s_memtime() - tf0;
#endif
    };
    // hipcc does not know about the asm DMA loads, and any vmcnt(0) it emits inside the stage loop (for a global load it
    // still considers pending at the loop's back edge) would drain them every sub-tile: retire the thresholds here and
    // hand them to the loop as plain register values
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    RG_RSTAMP(2);
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) asm volatile("" : "+v"(thr[gq]), "+v"(thr8[gq]), "+v"(thr8h[gq]));
    // thr_i: the threshold IN FORCE (int8: of the class of the stage being multiplied -- set at the top of a stage from
    // thr_n / thr_h when the class changes); thr_p (PIPE, whose epilogue of a stage's last sub-tile runs inside the next
    // stage): the previous stage's.  The two classes' integers are on different grids: a sub-tile is only ever tested
    // against the thresholds of its own granule's class.
    // (registers: the other class's thresholds are kept as thr_x = normal XOR heavy -- a class change toggles thr_i with it)
    int thr_i[NG];
    [[maybe_unused]] int thr_x[NG], thr_p[NG];
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {  // (int8: INT_MIN / INT_MAX -- everything / nothing passes -- clamped beyond any |I| < 2^23)
      thr_i[gq] = I8 ? max(-(1 << 24), min(1 << 24, thr8[gq])) : (thr[gq] >= 0.f ? __float_as_int(thr[gq]) : INT_MIN);
      thr_x[gq] = thr_i[gq] ^ max(-(1 << 24), min(1 << 24, thr8h[gq]));
      thr_p[gq] = thr_i[gq];
    }
    [[maybe_unused]] i32x4 ntq[FOLD ? NG : 1];   // FOLD: {-T, -T, -T, -T} per group, the accumulators' start values
    if constexpr (FOLD) {
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) ntq[gq] = i32x4{-thr_i[gq], -thr_i[gq], -thr_i[gq], -thr_i[gq]};
    }
    [[maybe_unused]] unsigned cls_word = 0u;   // class bits of the 32 stages around the current one (SGPR)
    [[maybe_unused]] int cls_cur = 0, cls_prev = 0, cls_state = 0, cls_inforce = 0;   // cls_inforce: the class thr_i holds

    // ---- ring prologue ------------------------------------------------------------------------------------------
    const int pro = nstages < C::SLOTS - 1 ? nstages : C::SLOTS - 1;
    if (tid < 2 * C::SLOTS) full[tid] = 0;
    for (int s = 0; s < pro; ++s) dma_stage(p.stage_base + st0 + s, s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < pro) full[tid] = C::WAVES;
    __syncthreads();
    RG_RSTAMP(3);
    // Partners on a SIMD (waves w and w + 4) run the same program -- a sub-tile's MFMAs, then its epilogue's VALU work -- and
    // the SIMD arbitrates between them by priority, then AGE: at equal priority the older wave takes every issue slot it can
    // use, runs a stage ahead of its partner, and then sleeps at the ring (a slot is reused when EVERY wave has left it) while
    // the partner runs alone, its epilogues beside nobody's MFMAs: the matrix pipe was busy 65 % of the last level's cycles
    // (SQ_VALU_MFMA_BUSY_CYCLES; wait for a free slot: 16 % of a wave's time, -DRG_TOPK_TIMING).  So the priority follows the
    // partner's progress -- sub-tiles done, one word per wave behind the ring flags, written per sub-tile and read once per
    // stage: a wave more than `lead` sub-tiles ahead yields (priority 0), one that is behind takes over (2), else 1.  Last
    // level of the bench 13.4 -> 12.8 ms, wait for a free slot 1467 -> 516 ticks per stage (profiles/r4_ring_priority.txt;
    // a start offset between the halves, static priority for the second half, priorities alternating per stage, priority
    // per phase -- MFMAs high / epilogue low and the reverse -- and shares of the DMA deferred instead of waited for: all
    // within noise or slower).
    int* prog = reinterpret_cast<int*>(freec + C::SLOTS);  // [WAVES]
    const int lead = BOUND ? 0 : p.partner_lead;   // (the bound pass's epilogue is eight maxima: nothing to arbitrate for)
    if (lead) {
      if (lane == 0) prog[wave] = 0;
      __builtin_amdgcn_s_setprio(1);
    }
    int partner_prog = 0;
    const unsigned prog_partner_addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(prog + (wave ^ (C::WAVES / 2)));

    int pending = -1;
    // PIPE: accumulators of the sub-tile in flight and of the one whose epilogue is running (sets alternate per sub-tile and
    // live across stages); the "previous sub-tile" of a segment's first one is a set no threshold admits
    using accp_t = typename std::conditional<I8, i32x4, f32x4>::type;
    accp_t accp[2][2][NG];   // (unused without PIPE)
    int pmi[NG];
    bool phit = false;
    if constexpr (PIPE) {
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        accp[1][0][gq] = accp[1][1][gq] = accp_t{INT_MIN, INT_MIN, INT_MIN, INT_MIN};
        pmi[gq] = INT_MIN;
      }
    }
#ifdef RG_TOPK_TIMING
    unsigned long long tw[6] = {0, 0, 0, 0, 0, 0};
#endif
    for (int s = 0; s < nstages; ++s) {
      const int slot = s & (C::SLOTS - 1), gen = s / C::SLOTS;
      if (!PIPE && (s & 0x7FFF) == 0 && s > 0) {  // keep the entries' key offsets inside 25 bits
        flush();
        key_org += 0x8000 * C::STAGE_KEYS;
      }
      if constexpr (BOUND) {
        if (st0 + s >= grp_end) {  // (groups hold at least one stage each: at most one boundary per stage)
          flush_max();
          ++grp;
          grp_end = ((int64_t)(grp + 1) * p.nstages_total + p.ngroups - 1) / p.ngroups;
        }
      }
      if constexpr (I8) {  // the stage's class: one scalar word per 32 stages, a select per group only when the class changes
        const int64_t stage_abs = p.stage_base + st0 + s;
        const int sa = __builtin_amdgcn_readfirstlane((int)(stage_abs & 31));
        if (s == 0 || sa == 0) cls_word = p.thr.cls8[__builtin_amdgcn_readfirstlane((int)(stage_abs >> 5))];
        cls_prev = s > 0 ? cls_cur : 0;
        cls_cur = (int)((cls_word >> sa) & 1u);
        const int state = cls_cur | (cls_prev << 1);
        if (state != cls_state) {  // (wave-uniform; never taken on a bank without heavy granules)
          cls_state = state;
          const bool toggle = cls_cur != cls_inforce, other = cls_prev != cls_cur;
          cls_inforce = cls_cur;
#pragma unroll
          for (int gq = 0; gq < NG; ++gq) {
            if (toggle) thr_i[gq] ^= thr_x[gq];
            if constexpr (PIPE) thr_p[gq] = other ? thr_i[gq] ^ thr_x[gq] : thr_i[gq];
            if constexpr (FOLD) ntq[gq] = i32x4{-thr_i[gq], -thr_i[gq], -thr_i[gq], -thr_i[gq]};
          }
        }
      }
      RG_FT(t0);
      fring_wait(full + slot, (unsigned)(C::WAVES * (gen + 1)));
      RG_FT(t1);
      // epilogue of sub-tile u: a[h][gq][r] = approximate score of key 16 h + 4 g + r of the sub-tile for query j of group gq
      using acc_t = typename std::conditional<I8, i32x4, f32x4>::type;
      auto pass_mask = [&](const acc_t (&a)[2][NG], int gq, [[maybe_unused]] int th_i8) {  // float scores against thr, integer sums against th_i8
        unsigned mk = 0;
        if constexpr (I8) {
          // bit = the sign of (thr - 1) - I, in unsigned arithmetic (|I| < 2^23 and thr_i is clamped to +-2^24: no wrap),
          // shifted into the mask by v_alignbit ({mask, e} >> 31 = mask << 1 | sign(e)): two plain VALU instructions per
          // score where compare + select + or through VCC is three plus a wait state, on a path that half of the last
          // level's sub-tiles take (and every sub-tile of the first)
          if constexpr (FOLD) {  // (the accumulators hold I - T: the bit is the sign of ~(I - T))
#pragma unroll
            for (int b = 7; b >= 0; --b) mk = __builtin_amdgcn_alignbit(mk, ~(unsigned)a[b >> 2][gq][b & 3], 31);
            return mk;
          }
          const unsigned tm1 = (unsigned)(th_i8 - 1);
#pragma unroll
          for (int b = 7; b >= 0; --b) mk = __builtin_amdgcn_alignbit(mk, tm1 - (unsigned)a[b >> 2][gq][b & 3], 31);
          return mk;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = a[h][gq][r] >= thr[gq];
            mk |= ok ? (1u << (4 * h + r)) : 0u;
          }
        return mk;
      };
      // the groups' entries of one sub-tile: ballots first, ONE buffer check per (up to) four groups -- a flush is ~60
      // instructions and every copy of it sits in the stage loop's instruction stream
      // (SCORED: the lane's largest sum and the class of the keys' granule ride in the entry's upper 24 bits as (I << 1) | class
      // -- |I| <= 127^2 * 256 < 2^22)
      auto push_groups = [&](const unsigned (&km)[NG], const int (&mi)[NG], unsigned off, [[maybe_unused]] int cls_of) {
        constexpr int GB = NG < 4 ? NG : (NG % 4 == 0 ? 4 : 3);  // groups per check: at most 64 GB = 256 entries < CAND_BUF
#pragma unroll
        for (int g0 = 0; g0 < NG; g0 += GB) {
          unsigned long long bm[GB];
          int tot = 0;
#pragma unroll
          for (int i = 0; i < GB; ++i) {
            bm[i] = __ballot(km[g0 + i] != 0);
            tot += __popcll(bm[i]);
          }
          if (tot == 0) continue;
          if (wcnt + tot > C::CAND_BUF) flush();
#pragma unroll
          for (int i = 0; i < GB; ++i) {
            if (bm[i]) {
              const int pos = wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm[i] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm[i], 0u));
              if (km[g0 + i])
                wbuf[pos] = make_uint2(((unsigned)(j + 16 * (g0 + i)) << 25) | off,
                                       SCORED ? (km[g0 + i] | ((unsigned)((mi[g0 + i] << 1) | cls_of) << 8)) : km[g0 + i]);
              wcnt += __popcll(bm[i]);
            }
          }
        }
      };
      auto as_bits = [](auto x) {  // a score as the signed integer the hit test compares
        if constexpr (I8) return (int)x;
        else return __float_as_int(x);
      };
      auto epilogue = [&](int u, const acc_t (&a)[2][NG]) {
        // Filter levels test "does any of the lane's 8 scores reach the threshold" on the scores' BIT PATTERNS as signed
        // integers: for a threshold >= +0 that is the float comparison (negative scores are negative integers, the MFMA
        // never produces -0 or NaN from finite operands), v_max3_i32 needs none of the canonicalising v_max x, x that
        // fmaxf puts in front of accumulator values, and a false positive would only send the sub-tile through the exact
        // float masks below.  (thr_i = INT_MIN for a negative threshold: always the exact path.)
        float m[NG];
        int mi[NG];
        bool hit = false;
        if constexpr (BOUND) {
#pragma unroll
          for (int gq = 0; gq < NG; ++gq) {  // (a chain, not a tree: hipcc folds it into v_max3_f32 -- 4 instructions, not 7)
            m[gq] = (float)a[0][gq][0];
#pragma unroll
            for (int r = 1; r < 4; ++r) m[gq] = fmaxf(m[gq], (float)a[0][gq][r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) m[gq] = fmaxf(m[gq], (float)a[1][gq][r]);
            gm[gq] = fmaxf(gm[gq], m[gq]);
          }
        } else if constexpr (FOLD) {  // the groups' maxima of I - T, their maximum, one sign test
#ifdef RG_FOLD_ONE_CHAIN   // (A/B build: one chain over all 8 NG values -- the groups' own maxima are then made again on a hit)
          int mall = as_bits(a[0][0][0]);
#pragma unroll
          for (int gq = 0; gq < NG; ++gq)
#pragma unroll
            for (int e = (gq == 0 ? 1 : 0); e < 8; ++e) mall = max(mall, as_bits(a[e >> 2][gq][e & 3]));
          hit = mall >= 0;
#else
          // (at D = 64 a sub-tile is 32 keys x QW queries for only two MFMAs per group: with pass rates of a few 1e-4 most
          // sub-tiles have a hit somewhere, and a single chain over everything meant the groups' maxima were computed twice)
#pragma unroll
          for (int gq = 0; gq < NG; ++gq) {
            mi[gq] = as_bits(a[0][gq][0]);
#pragma unroll
            for (int r = 1; r < 4; ++r) mi[gq] = max(mi[gq], as_bits(a[0][gq][r]));
#pragma unroll
            for (int r = 0; r < 4; ++r) mi[gq] = max(mi[gq], as_bits(a[1][gq][r]));
          }
          int mall = mi[0];
#pragma unroll
          for (int gq = 1; gq < NG; ++gq) mall = max(mall, mi[gq]);
          hit = mall >= 0;
#endif
        } else {
#pragma unroll
          for (int gq = 0; gq < NG; ++gq) {
            mi[gq] = as_bits(a[0][gq][0]);
#pragma unroll
            for (int r = 1; r < 4; ++r) mi[gq] = max(mi[gq], as_bits(a[0][gq][r]));
#pragma unroll
            for (int r = 0; r < 4; ++r) mi[gq] = max(mi[gq], as_bits(a[1][gq][r]));
            hit = hit || (mi[gq] >= thr_i[gq]);
          }
        }
        if constexpr (BOUND) {
        } else if (__any(hit)) {
          const int stage_key0 = (int)((p.stage_base + st0 + s) * C::STAGE_KEYS);
          const int key_base = stage_key0 + 32 * u + 4 * g;  // the lane's keys: key_base + r + 16 h  (mask bit 4 h + r)
          // (a group without a passing lane skips its compares: at the later levels a sub-tile that has a candidate at
          // all usually has it in one group only)
          unsigned km[NG];
          if constexpr (FOLD) {  // scored entries carry I itself: + T
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) {
#ifdef RG_FOLD_ONE_CHAIN
              mi[gq] = as_bits(a[0][gq][0]);
#pragma unroll
              for (int r = 1; r < 4; ++r) mi[gq] = max(mi[gq], as_bits(a[0][gq][r]));
#pragma unroll
              for (int r = 0; r < 4; ++r) mi[gq] = max(mi[gq], as_bits(a[1][gq][r]));
#endif
              km[gq] = 0;
              if (__any(mi[gq] >= 0)) km[gq] = pass_mask(a, gq, 0);
              mi[gq] += thr_i[gq];
            }
          } else
#pragma unroll
          for (int gq = 0; gq < NG; ++gq) {
            km[gq] = 0;
            if (__any(mi[gq] >= thr_i[gq])) km[gq] = pass_mask(a, gq, thr_i[gq]);
          }
          if (stage_key0 + C::STAGE_KEYS > (int)p.N) {  // the range's last stage: keys >= N (padding, or the next level's)
            unsigned vm = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) vm |= (key_base + (r & 3) + 16 * (r >> 2) < (int)p.N) ? (1u << r) : 0u;
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) km[gq] &= vm;
          }
          push_groups(km, mi, (unsigned)(key_base - key_org), cls_cur);
        }
      };
      // PIPE: the same epilogue in pieces -- one group's maxima per step, then the candidate path -- over the OTHER set
      // (th: the thresholds of the sub-tile's own stage -- thr_p for the previous stage's last sub-tile, else thr_i)
      auto epi_fast = [&](const acc_t (&a)[2][NG], int gq, const int (&th)[NG]) {
        int m = as_bits(a[0][gq][0]);
#pragma unroll
        for (int r = 1; r < 4; ++r) m = max(m, as_bits(a[0][gq][r]));
#pragma unroll
        for (int r = 0; r < 4; ++r) m = max(m, as_bits(a[1][gq][r]));
        pmi[gq] = m;
        phit = phit || (m >= th[gq]);
      };
      auto epi_slow = [&](const acc_t (&a)[2][NG], int stage_key0, int u, int cls_of, const int (&th)[NG]) {
        if (__any(phit)) {
          const int key_base = stage_key0 + 32 * u + 4 * g;
          unsigned km[NG];
#pragma unroll
          for (int gq = 0; gq < NG; ++gq) {
            km[gq] = 0;
            if (__any(pmi[gq] >= th[gq])) km[gq] = pass_mask(a, gq, th[gq]);
          }
          if (stage_key0 + C::STAGE_KEYS > (int)p.N) {
            unsigned vm = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) vm |= (key_base + (r & 3) + 16 * (r >> 2) < (int)p.N) ? (1u << r) : 0u;
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) km[gq] &= vm;
          }
          push_groups(km, pmi, (unsigned)(key_base - key_org), cls_of);
        }
        phit = false;
      };
      // ---- SUBS sub-tiles of 32 keys x QW queries, KSTEPS fragments each (k-step major: both 16-key halves of a step);
      // one A fragment feeds all NG query groups.
      // A step is only 64 cycles of MFMA, less than an LDS round trip, so the fragment reads run FOUR steps ahead of
      // their MFMAs (hipcc's own schedule keeps one ahead and the matrix pipe idles half the time).  They are asm
      // loads, invisible to hipcc's waitcnt bookkeeping: RG_FWAIT counts them (LDS returns in order; anything else
      // outstanding only makes the wait stricter) and names the fragment so its MFMAs stay behind the wait.
      const unsigned addr = apos + (unsigned)(slot * C::STAGE_BYTES);
      acc_t acc[2][NG];   // (PIPE: accp instead)
      f32x4 fr[4];
      const int stage_key0_now = (int)((p.stage_base + st0 + s) * C::STAGE_KEYS);
#define RG_FREAD(n_)                                                                                       \
  asm volatile("ds_read_b128 %0, %1 offset:%2"                                                             \
               : "=v"(fr[(n_)&3])                                                                           \
               : "v"(addr), "i"((n_) * 1024))
#define RG_FWAIT(c_, n_) asm volatile("s_waitcnt lgkmcnt(" #c_ ")" : "+v"(fr[(n_)&3]))
#define RG_FSTEP(n_)                                                                                       \
  {                                                                                                        \
    constexpr int u_ = (n_) / C::KSTEPS, r_ = (n_) % C::KSTEPS, set_ = u_ & 1;                             \
    if constexpr (r_ == 0) {                                                                               \
      if constexpr (PIPE) {                                                                                \
        _Pragma("unroll") for (int gq = 0; gq < NG; ++gq) accp[set_][0][gq] = accp[set_][1][gq] = acc_t{0, 0, 0, 0}; \
      } else if constexpr (FOLD) {                                                                         \
        _Pragma("unroll") for (int gq = 0; gq < NG; ++gq)                                                  \
          acc[0][gq] = acc[1][gq] = __builtin_bit_cast(acc_t, ntq[gq]);                                    \
      } else {                                                                                             \
        _Pragma("unroll") for (int gq = 0; gq < NG; ++gq) acc[0][gq] = acc[1][gq] = acc_t{0, 0, 0, 0};     \
      }                                                                                                    \
    }                                                                                                      \
    if constexpr ((n_) + 3 < C::NSTEP) RG_FWAIT(3, n_);                                                     \
    else if constexpr ((n_) + 2 < C::NSTEP) RG_FWAIT(2, n_);                                                \
    else if constexpr ((n_) + 1 < C::NSTEP) RG_FWAIT(1, n_);                                                \
    else RG_FWAIT(0, n_);                                                                                   \
    if constexpr (I8 && PIPE) {                                                                            \
      const i32x4 a_ = __builtin_bit_cast(i32x4, fr[(n_)&3]);                                               \
      _Pragma("unroll") for (int gq = 0; gq < NG; ++gq)                                                     \
        accp[set_][(n_) & 1][gq] = __builtin_bit_cast(acc_t, __builtin_amdgcn_mfma_i32_16x16x64_i8(         \
            a_, bqi[gq][((n_) >> 1) % C::KS32], __builtin_bit_cast(i32x4, accp[set_][(n_) & 1][gq]), 0, 0, 0)); \
    } else if constexpr (I8) {                                                                             \
      const i32x4 a_ = __builtin_bit_cast(i32x4, fr[(n_)&3]);                                               \
      _Pragma("unroll") for (int gq = 0; gq < NG; ++gq)                                                     \
        acc[(n_) & 1][gq] = __builtin_bit_cast(acc_t, __builtin_amdgcn_mfma_i32_16x16x64_i8(                \
            a_, bqi[gq][((n_) >> 1) % C::KS32], __builtin_bit_cast(i32x4, acc[(n_) & 1][gq]), 0, 0, 0));    \
    } else {                                                                                               \
      const bf16x8 a_ = __builtin_bit_cast(bf16x8, fr[(n_)&3]);                                             \
      _Pragma("unroll") for (int gq = 0; gq < NG; ++gq)                                                     \
        acc[(n_) & 1][gq] = __builtin_bit_cast(acc_t, __builtin_amdgcn_mfma_f32_16x16x32_bf16(              \
            a_, bq[gq][((n_) >> 1) % C::KS32], __builtin_bit_cast(f32x4, acc[(n_) & 1][gq]), 0, 0, 0));     \
    }                                                                                                      \
    if constexpr ((n_) + 4 < C::NSTEP) RG_FREAD((n_) + 4);                                                  \
    if constexpr (PIPE) {                                                                                  \
      /* the previous sub-tile (the other set): group r - 1 behind step r, the candidate path behind step NG + 1 */ \
      if constexpr (r_ >= 1 && r_ <= NG) epi_fast(accp[set_ ^ 1], r_ - 1, RG_PTHR(u_));                    \
      if constexpr (r_ == NG + 1) {                                                                        \
        epi_slow(accp[set_ ^ 1], u_ == 0 ? stage_key0_now - C::STAGE_KEYS : stage_key0_now, u_ == 0 ? C::SUBS - 1 : u_ - 1, \
                 u_ == 0 ? cls_prev : cls_cur, RG_PTHR(u_));                                                \
        if constexpr (u_ == 0) {                                                                           \
          if ((s & 0x7FFF) == 0 && s > 0) { /* keep the entries' key offsets inside 25 bits */             \
            flush();                                                                                       \
            key_org += 0x8000 * C::STAGE_KEYS;                                                             \
          }                                                                                                \
        }                                                                                                  \
      }                                                                                                    \
      if constexpr (r_ == C::KSTEPS - 1) {                                                                 \
        if (lead && lane == 0)                                                                             \
          __hip_atomic_store(prog + wave, s * C::SUBS + u_ + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
      }                                                                                                    \
    } else if constexpr (r_ == C::KSTEPS - 1) {                                                            \
      epilogue(u_, acc);                                                                                   \
      if (lead && lane == 0)                                                                               \
        __hip_atomic_store(prog + wave, s * C::SUBS + u_ + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    }                                                                                                      \
  }
#define RG_PTHR(u_) ((u_) == 0 ? thr_p : thr_i)
#define RG_FSTEP8(n_) RG_FSTEP(n_) RG_FSTEP((n_) + 1) RG_FSTEP((n_) + 2) RG_FSTEP((n_) + 3) \
    RG_FSTEP((n_) + 4) RG_FSTEP((n_) + 5) RG_FSTEP((n_) + 6) RG_FSTEP((n_) + 7)
      // (a wave whose 64 queries all lie beyond the batch -- the tail of a ragged last tile: 300 queries fill 4.7 of a
      // tile's 8 waves -- only takes part in the ring: no fragment reads, no MFMAs, the SIMD to its partner)
      if (wave_live) {
        RG_FREAD(0);
        RG_FREAD(1);
        RG_FREAD(2);
        RG_FREAD(3);
        RG_FSTEP8(0) RG_FSTEP8(8) RG_FSTEP8(16)
        // (the partner's progress: one more read in the in-order LDS queue -- the counted waits only get stricter -- landed
        // by the stage's last wait)
        if (lead) asm volatile("ds_read_b32 %0, %1" : "=v"(partner_prog) : "v"(prog_partner_addr));
        RG_FSTEP8(24)
        if constexpr (PIPE) {  // the segment's last sub-tile: no next sub-tile for its epilogue to ride in
          if (s == nstages - 1) {
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) epi_fast(accp[(C::SUBS - 1) & 1], gq, thr_i);
            epi_slow(accp[(C::SUBS - 1) & 1], stage_key0_now, C::SUBS - 1, cls_cur, thr_i);
          }
        }
        if (lead) {
          asm volatile("" : "+v"(partner_prog));
          const int d = (s + 1) * C::SUBS - __builtin_amdgcn_readfirstlane(partner_prog);
          if (d >= lead + 1) __builtin_amdgcn_s_setprio(0);
          else if (d <= 1 - lead) __builtin_amdgcn_s_setprio(2);
          else __builtin_amdgcn_s_setprio(1);
        }
      }
#undef RG_FSTEP8
#undef RG_PTHR
#undef RG_FSTEP
#undef RG_FWAIT
#undef RG_FREAD
      RG_FT(t2);
      fring_signal(freec + slot, lane);
      if (pending >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fring_signal(full + pending, lane);
        pending = -1;
      }
      RG_FT(t3);
#ifdef RG_TOPK_TIMING
      unsigned long long t4 = t3, t5 = t3;
#endif
      if (s + C::SLOTS - 1 < nstages) {
        const int ws = (s + C::SLOTS - 1) & (C::SLOTS - 1);  // the slot stage s-1 lived in
        const unsigned need = (unsigned)(C::WAVES * ((s + C::SLOTS - 1) / C::SLOTS));
        fring_wait(freec + ws, need);
#ifdef RG_TOPK_TIMING
        t4 = __builtin_amdgcn_s_memtime();
#endif
        dma_stage(p.stage_base + st0 + s + C::SLOTS - 1, ws);
        pending = ws;
#ifdef RG_TOPK_TIMING
        t5 = __builtin_amdgcn_s_memtime();
#endif
      }
#ifdef RG_TOPK_TIMING
      tw[0] += t1 - t0; tw[1] += t2 - t1; tw[2] += t3 - t2; tw[3] += t4 - t3; tw[4] += t5 - t4; tw[5] += 1;
#endif
    }
#ifdef RG_TOPK_TIMING
    if (lane == 0) {
      for (int i = 0; i < 6; ++i) atomicAdd(&g_filter_timing[i], tw[i]);
      atomicAdd(&g_filter_timing[6], tfl);   // (flushes inside the stage loop: part of `compute`)
      atomicAdd(&g_filter_timing[7], nfl);
    }
#endif
    RG_RSTAMP(4);
    if (lead) __builtin_amdgcn_s_setprio(0);
    flush();
    if constexpr (BOUND) flush_max();
    RG_RSTAMP(5);
    first_seg = false;
    __syncthreads();  // flags are re-initialised by the next segment
  }
#ifdef RG_RING_STAMPS
  if (threadIdx.x == 0) atomicMax(&g_ring_span[BOUND][1], wall_clock64());
#endif
}
