// Exact rescoring of the filter levels' candidates (one wave / one workgroup per query, cooperative row staging, scored lists, sliced wide form).
// Part of csrc/topk_filter.hip (textually included there, inside its namespace / after its helpers): split out in round 6 so
// that the ring, the candidate path and the launch plumbing can be read -- and changed -- apart.  No include guard on purpose:
// these are not stand-alone headers.

// flag[b] == 2 (filter_prep_kernel: a ZERO query).  Intermediate levels leave it alone (nothing passed for it; its running
// result is never used); the final level writes its answer -- every score +0, the canonical order is the index order --
// and the query is neither counted in *overflow nor listed for the scan.  Returns true when the wave is done with the query.
__device__ __forceinline__ bool zero_query_level(unsigned char fl, int final_level, int k, int64_t idx_base, int lane,
                                                 float* __restrict__ out_s, int64_t* __restrict__ out_i) {
  if (fl != 2) return false;
  if (final_level && lane < k) {
    out_s[lane] = 0.f;
    out_i[lane] = idx_base + lane;
  }
  return true;
}

// One wave per query.  prev_* (the previous level's exact top-k, local indices) may alias out_*.  A query whose list
// overflowed (now or at an earlier level: flag) is appended to overflow_idx by the final level.  Most queries hold far
// fewer candidates than the capacity: the slot count is a wave-uniform choice among 1, 2, 4, 8 and CPL.
template <int D, int CPL>
__global__ void __launch_bounds__(256) topk_rescore_kernel(const float* __restrict__ Qn, const float* __restrict__ Kn,
                                                           int* __restrict__ count, const int* __restrict__ cand,
                                                           int64_t B, int cap, int cs, int k, int64_t idx_base,
                                                           const float* prev_s, const int64_t* prev_i, int final_level,
                                                           float* out_s, int64_t* out_i, int* __restrict__ overflow,
                                                           int* __restrict__ overflow_list,
                                                           unsigned char* __restrict__ flag, int* __restrict__ cstat) {
  __shared__ float4 qs[4][D / 4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  if (b >= B) return;  // whole wave
  if (lane < D / 4) qs[w][lane] = reinterpret_cast<const float4*>(Qn + b * D)[lane];  // the query row
  __builtin_amdgcn_wave_barrier();
  int n = count[b * cs];
  const unsigned char fl = flag[b];
  bool over = fl != 0;
  if (lane == 0 && n >= 0) count[b * cs] = 0;  // the next level starts from an empty list (ordered behind the read through n)
  if (lane == 0) note_candidates(cstat, b, n);
  if (zero_query_level(fl, final_level, k, idx_base, lane, out_s + b * k, out_i + b * k)) return;
  if (n > cap) {  // slots reserved beyond the capacity: candidates were dropped
    over = true;
    n = cap;
  }
  if (lane == 0) {
    if (final_level) {
      if (over) {
        const int pos = atomicAdd(overflow, 1);
        overflow_list[pos] = (int)b;
        flag[b] = 1;   // (listed: a speculative call's verify launch must not list it again)
      }
    } else if (over) {
      flag[b] = 1;
    }
  }
  const int64_t base = final_level ? idx_base : 0;
  const float* ps = prev_s ? prev_s + b * k : nullptr;
  const int64_t* pi = prev_i ? prev_i + b * k : nullptr;
  const int* cb = cand + b * cap;
#define RG_RESCORE(NS_) rescore_query<D, NS_>(qs[w], Kn, cb, n, lane, k, base, ps, pi, out_s + b * k, out_i + b * k)
  if (n <= 64) RG_RESCORE(1);
  else if (n <= 128) RG_RESCORE(2);
  else if (n <= 256) RG_RESCORE(4);
  else if (n <= 512) RG_RESCORE(8);
  else RG_RESCORE(CPL);
#undef RG_RESCORE
}

// Large batches: as topk_rescore_kernel, rows staged through LDS (coop_scores); two waves per workgroup.
// FEWTILE: the variant for levels that leave a query a handful of candidates (the later levels over a sharded bank,
// whose bounds were sharpened across the shards): a 16-row tile instead of 64, so that four times as many waves fit a
// CU -- such a level is a chain of memory latencies per query, and occupancy is what hides them; the rare longer list
// takes the lane-private row reads.
template <int D, int CPL, bool FEWTILE = false>
__global__ void __launch_bounds__(128, FEWTILE ? 4 : 1) topk_rescore_coop_kernel(const float* __restrict__ Qn, const float* __restrict__ Kn,
                                                                int* __restrict__ count,
                                                                const int* __restrict__ cand, int64_t B, int cap, int cs, int k,
                                                                int64_t idx_base, const float* prev_s,
                                                                const int64_t* prev_i, int final_level, float* out_s,
                                                                int64_t* out_i, int* __restrict__ overflow,
                                                                int* __restrict__ overflow_list,
                                                                unsigned char* __restrict__ flag, int64_t scan_n,
                                                                int* __restrict__ cstat) {
  __shared__ float4 qs[2][D / 4];
  __shared__ __attribute__((aligned(16))) float tile[2][(FEWTILE ? 16 : 64) * RESCORE_LD];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 2 + w;
  if (b >= B) return;  // whole wave
  // (the count, the flag and the query row are independent loads: issued together, one latency)
  int n = count[b * cs];
  const unsigned char fl = flag[b];
  bool over = fl != 0;
  float4 qv4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < D / 4) qv4 = reinterpret_cast<const float4*>(Qn + b * D)[lane];
  if (lane < D / 4) qs[w][lane] = qv4;
  __builtin_amdgcn_wave_barrier();
  if (lane == 0 && n >= 0) count[b * cs] = 0;  // the next level starts from an empty list (ordered behind the read through n)
  if (lane == 0) note_candidates(cstat, b, n);
  if (zero_query_level(fl, final_level, k, idx_base, lane, out_s + b * k, out_i + b * k)) return;
  if (n > cap) {
    over = true;
    n = cap;
  }
  if (final_level && over && scan_n > 0) {  // (wave-uniform) calls of up to 16384 queries: the query's wave scans the bank
    if (lane == 0) atomicAdd(overflow, 1);  // itself, and the call needs no fallback launch
    exact_scan_wave<D>(qs[w], Kn, scan_n, k, idx_base, lane, out_s + b * k, out_i + b * k);
    return;
  }
  if (lane == 0) {
    if (final_level) {
      if (over) {
        const int pos = atomicAdd(overflow, 1);
        overflow_list[pos] = (int)b;
        flag[b] = 1;   // (listed: a speculative call's verify launch must not list it again)
      }
    } else if (over) {
      flag[b] = 1;
    }
  }
  const int64_t base = final_level ? idx_base : 0;
  const float* ps = prev_s ? prev_s + b * k : nullptr;
  const int64_t* pi = prev_i ? prev_i + b * k : nullptr;
  const int* cb = cand + b * cap;
#define RG_RESCORE(NS_, COOP_) \
  rescore_query<D, NS_, COOP_>(qs[w], Kn, cb, n, lane, k, base, ps, pi, out_s + b * k, out_i + b * k, tile[w])
  if (n <= 16)
    rescore_query<D, 1, true, true>(qs[w], Kn, cb, n, lane, k, base, ps, pi, out_s + b * k, out_i + b * k, tile[w]);
  else if (n <= 64) RG_RESCORE(1, !FEWTILE);
  else if (n <= 128) RG_RESCORE(2, !FEWTILE);
  else if (n <= 256) RG_RESCORE(4, !FEWTILE);
  else if (n <= 512) RG_RESCORE(8, false);  // long lists are rare: the plain form keeps the kernel out of scratch
  else RG_RESCORE(CPL, false);
#undef RG_RESCORE
}

// SCORED lists (the int8 levels of large calls: entries {key, I}, topk_filter_kernel<..., SCORED>): one wave per query, in
// two rounds.  Round 1 scores the SCORED_R1 entries with the largest I exactly; the k-th best of those and the previous
// level's winners is a lower bound theta_e of the query's final k-th best score -- k distinct keys reach it -- and a key
// can only enter the top-k if its exact score s >= theta_e, so its I >= (theta_e - eps) / (s_q s_k) (the level's own bound,
// filter_threshold_i8_at; a lane's shared I is an upper bound, which only keeps an entry in).  Round 2 scores the entries
// that pass THAT threshold; the rest are never fetched.  The int8 bound admits ~3x the candidates of the bf16 one
// because its eps is ~5x wider -- but theta_e sits ~0.3 sigma above the threshold the level ran with (that came from a
// quarter of the keys), and two thirds of the admitted keys fall below it: ~120 -> ~35 row gathers per query on the
// bench's last level.  Which entries round 1 takes changes the work, never the result: every entry that could belong
// to the top-k is scored with the same fmaf chain, and the selection is the canonical one.
constexpr int SCORED_R1 = 16;
// The selections here are by COUNTING over the few pairs in play (a pair's rank = the number of better pairs, each
// broadcast once with v_readlane), not wave_select's rounds over every slot: the kernel runs one wave per query and
// ~3000 VALU instructions of selection per query were as long as its row gathers.
__device__ __forceinline__ bool pair_gt(unsigned ah, unsigned al, unsigned bh, unsigned bl) {
  return ah > bh || (ah == bh && al > bl);
}
// Returns false (nothing written) when more than 64 round-2 entries beat round 1's k-th pair: the caller then scores
// the whole list the plain way (a level whose first bound was useless; rare).
template <int D, int NS, int ROWS = 64>
__device__ __forceinline__ bool rescore_scored_query(const float4* __restrict__ qrow, const float* __restrict__ Kn,
                                                     const int2* __restrict__ cb, int n, int lane, int k, int64_t base,
                                                     const float* prev_s, const int64_t* prev_i, float* out_s, int64_t* out_i,
                                                     float* sm, int* surv, int* stage, const FilterThr& thr, int64_t b) {
  int key[NS], iv[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int c = lane + 64 * u;
    int2 e = make_int2(-1, INT_MIN);
    if (c < n) e = cb[c];
    key[u] = e.x;
    iv[u] = e.y;
  }
  // round 1: every lane's best entry; the SCORED_R1 lanes with the largest of those (ties: lower lane) -- n > 24, so the
  // first 25 lanes hold an entry each and round 1 is full
  int bi = iv[0], bkey = key[0];
#pragma unroll
  for (int u = 1; u < NS; ++u) {
    const bool t = iv[u] > bi;
    bi = t ? iv[u] : bi;
    bkey = t ? key[u] : bkey;
  }
  const int nl = n < 64 ? n : 64;
  int rank = 0;
  for (int o = 0; o < nl; ++o) {
    const int io = __builtin_amdgcn_readlane(bi, o);
    rank += (io > bi || (io == bi && o < lane)) ? 1 : 0;
  }
  const bool lane_r1 = lane < nl && rank < SCORED_R1;
  if (lane_r1) stage[rank] = bkey;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int r1key = lane < SCORED_R1 ? stage[lane] : -1;
  __builtin_amdgcn_wave_barrier();
  const float e1 = coop_scores_few<D>(qrow, Kn, r1key, lane, sm);
  // W: round 1 (lanes 0..15) and the previous winners (lanes 16..16+k-1) as canonical 64-bit keys (0 = no pair)
  unsigned wh = 0u, wl = 0u;
  if (lane < SCORED_R1) {
    wh = select_ord(e1);
    wl = ~(unsigned)r1key;
  } else if (prev_s && lane < SCORED_R1 + k) {
    const int64_t pv = prev_i[lane - SCORED_R1];
    if (pv < INT_MAX) {
      wh = select_ord(prev_s[lane - SCORED_R1]);
      wl = ~(unsigned)(int)pv;
    }
  }
  const int nwl = prev_s ? SCORED_R1 + k : SCORED_R1;
  const bool w_valid = (wh | wl) != 0u;
  const int n_w = __popcll(__ballot(w_valid));
  int rank_w = 0;
  for (int o = 0; o < nwl; ++o) {
    const unsigned oh = (unsigned)__builtin_amdgcn_readlane((int)wh, o), ol = (unsigned)__builtin_amdgcn_readlane((int)wl, o);
    rank_w += pair_gt(oh, ol, wh, wl) ? 1 : 0;
  }
  // theta_e = the k-th best of W, a lower bound of the final k-th best (-inf: fewer than k pairs, everything is scored)
  unsigned kh = 0u, kl = 0u;
  if (n_w >= k) {
    const unsigned long long at = __ballot(w_valid && rank_w == k - 1);
    const int src = __ffsll((long long)at) - 1;
    kh = (unsigned)__builtin_amdgcn_readlane((int)wh, src);
    kl = (unsigned)__builtin_amdgcn_readlane((int)wl, src);
  }
  const float theta_e = n_w >= k ? select_unord(kh) : RG_NEG_INF;
  // (an entry's integer is (I << 1) | class of its key's granule: each class has its own bound -- compared on the doubled scale)
  const int t_e0 = filter_threshold_i8_at(thr, b, theta_e, 0), t_e1 = filter_threshold_i8_at(thr, b, theta_e, 1);
  auto twice = [](int t) { return t <= -(1 << 24) ? INT_MIN : (t >= (1 << 24) ? INT_MAX : 2 * t); };
  const int t2_e0 = twice(t_e0), t2_e1 = twice(t_e1);
  // round 2, four slots (256 entries) at a time: the entries outside round 1 whose I reaches t_e are compacted into the
  // wave's list and scored in batches of 64; those that beat W's k-th pair (a handful) are kept, one per lane of the
  // "beaters" row
  unsigned bh = 0u, bl = 0u;  // lane p: beater p
  int nb = 0;
#pragma unroll
  for (int u0 = 0; u0 < NS; u0 += 4) {
    int ns = 0;
#pragma unroll
    for (int u = u0; u < (u0 + 4 < NS ? u0 + 4 : NS); ++u) {
      const bool in_r1 = lane_r1 && key[u] == bkey;
      const bool keep = key[u] >= 0 && !in_r1 && iv[u] >= ((iv[u] & 1) ? t2_e1 : t2_e0);
      const unsigned long long bm = __ballot(keep);
      const int pos = ns + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
      if (keep) surv[pos] = key[u];
      ns += __popcll(bm);
    }
    if (ns == 0) continue;  // (wave-uniform)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int c0 = 0; c0 < ns; c0 += ROWS) {  // (ROWS = 32: the half tile of the large calls' kernel)
      const int c = c0 + lane;
      const int kk = (lane < ROWS && c < ns) ? surv[c] : -1;
      const float acc = ns - c0 <= 16 ? coop_scores_few<D>(qrow, Kn, kk, lane, sm) : coop_scores<D, ROWS>(qrow, Kn, kk, lane, sm);
      const unsigned sh = select_ord(acc), sl = ~(unsigned)kk;
      const bool beats = kk >= 0 && pair_gt(sh, sl, kh, kl);
      const unsigned long long bm = __ballot(beats);
      const int cnt = __popcll(bm);
      if (nb + cnt > 64) return false;  // (wave-uniform)
      const int pos = nb + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
      if (beats) {
        stage[pos] = (int)sh;
        stage[64 + pos] = (int)sl;
      }
      nb += cnt;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (the next four slots reuse the list)
  }
  if (nb > 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < nb) {
      bh = (unsigned)stage[lane];
      bl = (unsigned)stage[64 + lane];
    }
  }
  // final ranks: a pair of W gains the beaters better than it; a beater counts the better pairs of both rows
  int rank_b = 0;
  for (int o = 0; o < nb; ++o) {
    const unsigned oh = (unsigned)__builtin_amdgcn_readlane((int)bh, o), ol = (unsigned)__builtin_amdgcn_readlane((int)bl, o);
    rank_w += pair_gt(oh, ol, wh, wl) ? 1 : 0;
    rank_b += pair_gt(oh, ol, bh, bl) ? 1 : 0;
  }
  if (nb > 0)
    for (int o = 0; o < nwl; ++o) {
      const unsigned oh = (unsigned)__builtin_amdgcn_readlane((int)wh, o), ol = (unsigned)__builtin_amdgcn_readlane((int)wl, o);
      rank_b += pair_gt(oh, ol, bh, bl) ? 1 : 0;
    }
  if (w_valid && rank_w < k) {
    out_s[rank_w] = select_unord(wh);
    out_i[rank_w] = (int64_t)(int)~wl + base;
  }
  if (lane < nb && rank_b < k) {
    out_s[rank_b] = select_unord(bh);
    out_i[rank_b] = (int64_t)(int)~bl + base;
  }
  if (lane < k && lane >= n_w + nb) {
    out_s[lane] = RG_NEG_INF;
    out_i[lane] = INT64_MAX;
  }
  return true;
}

// SMALL (calls of 8192 queries and more, whose lists average ~120 entries and whose second round ~25 rows): half tiles and
// at most four entry slots per lane -- 15 KB less LDS per workgroup and ~50 fewer registers, three waves per SIMD instead
// of two for a kernel that lives on hiding row-gather latency; the few longer lists take lane-private row reads.
template <int D, bool SMALL = false>
__global__ void __launch_bounds__(128) topk_rescore_scored_kernel(const float* __restrict__ Qn, const float* __restrict__ Kn,
                                                                  int* __restrict__ count, const int2* __restrict__ cand,
                                                                  int64_t B, int cap, int cs, int k, int64_t idx_base,
                                                                  const float* prev_s, const int64_t* prev_i, int final_level,
                                                                  float* out_s, int64_t* out_i, int* __restrict__ overflow,
                                                                  int* __restrict__ overflow_list,
                                                                  unsigned char* __restrict__ flag, int64_t scan_n, FilterThr thr,
                                                                  int* __restrict__ cstat) {
  constexpr int ROWS = SMALL ? 32 : 64;
  __shared__ float4 qs[2][D / 4];
  __shared__ __attribute__((aligned(16))) float tile[2][ROWS * RESCORE_LD];
  __shared__ int surv[2][256];
  __shared__ int stage[2][128];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 2 + w;
  if (b >= B) return;  // whole wave
  // (the count, the flag and the query row are independent loads: issued together, one latency)
  int n = count[b * cs];
  const unsigned char fl = flag[b];
  bool over = fl != 0;
  float4 qv4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < D / 4) qv4 = reinterpret_cast<const float4*>(Qn + b * D)[lane];
  if (lane < D / 4) qs[w][lane] = qv4;
  __builtin_amdgcn_wave_barrier();
  if (lane == 0 && n >= 0) count[b * cs] = 0;  // the next level starts from an empty list
  if (lane == 0) note_candidates(cstat, b, n);
  if (zero_query_level(fl, final_level, k, idx_base, lane, out_s + b * k, out_i + b * k)) return;
  if (n > cap) {
    over = true;
    n = cap;
  }
  if (final_level && over && scan_n > 0) {
    if (lane == 0) atomicAdd(overflow, 1);
    exact_scan_wave<D>(qs[w], Kn, scan_n, k, idx_base, lane, out_s + b * k, out_i + b * k);
    return;
  }
  if (lane == 0) {
    if (final_level) {
      if (over) {
        const int pos = atomicAdd(overflow, 1);
        overflow_list[pos] = (int)b;
        flag[b] = 1;   // (listed: a speculative call's verify launch must not list it again)
      }
    } else if (over) {
      flag[b] = 1;
    }
  }
  const int64_t base = final_level ? idx_base : 0;
  const float* ps = prev_s ? prev_s + b * k : nullptr;
  const int64_t* pi = prev_i ? prev_i + b * k : nullptr;
  const int2* cb = cand + b * cap;
  const int* ck = reinterpret_cast<const int*>(cb);
#define RG_SCORED(NS_) \
  rescore_scored_query<D, NS_, ROWS>(qs[w], Kn, cb, n, lane, k, base, ps, pi, out_s + b * k, out_i + b * k, tile[w], surv[w], \
                                     stage[w], thr, b)
  bool done = false;
  if (n > 24) {  // (round 1 alone would take most of a shorter list)
    if (n <= 64) done = RG_SCORED(1);
    else if (n <= 128) done = RG_SCORED(2);
    else if (n <= 256) done = RG_SCORED(4);
    else if constexpr (!SMALL) {
      if (n <= 512) done = RG_SCORED(8);  // (the single level of a few hundred queries admits ~400 each and prunes 90 %)
      else if (n <= 1024) done = RG_SCORED(16);
    }
  }
#undef RG_SCORED
  if (done) return;
  // every entry the plain way: short lists, lists beyond 1024 entries, more than 64 entries beating round 1's k-th pair
#define RG_PLAIN(NS_, COOP_, FEW_) \
  rescore_query<D, NS_, COOP_, FEW_, 2>(qs[w], Kn, ck, n, lane, k, base, ps, pi, out_s + b * k, out_i + b * k, tile[w])
  if (n <= 16) RG_PLAIN(1, true, true);
  else if (n <= 64) RG_PLAIN(1, !SMALL, false);   // (the half tile holds no 64-row batch: lane-private row reads)
  else if (n <= 128) RG_PLAIN(2, !SMALL, false);
  else if (n <= 256) RG_PLAIN(4, !SMALL, false);
  else if (n <= 512) RG_PLAIN(8, false, false);   // long plain lists are rare: lane-private row reads
  else if (n <= 1024) RG_PLAIN(16, false, false);
  else RG_PLAIN(32, false, false);
#undef RG_PLAIN
}

// The exact fallback for a query whose candidate list overflowed (thousands of keys within eps of the k-th best:
// near-duplicate banks, zero queries), ON THE DEVICE: one workgroup scans the whole bank for it with the fp32 chain
// (coop_scores: 64 rows per step through an LDS tile), four waves a quarter of the keys each with a register-resident
// sorted list (lane p = entry p), merged at the end.  No host read-back, so the call stays asynchronous and HIP-graph
// capturable; a bank that sends many queries here is slow (one full fp32 scan per query and workgroup), which KeyIndex
// notices from the count after the fact and stops filtering that bank.  All 256 threads of the workgroup must call.
template <int D>
__device__ __forceinline__ void exact_scan_query(const float4* qs /* LDS: the query row */, const float* __restrict__ Kn,
                                                 int64_t N, int k, int64_t idx_base, float (*tile)[64 * RESCORE_LD],
                                                 float (*ps)[32], int64_t (*pi)[32], float* __restrict__ out_s,
                                                 int64_t* __restrict__ out_i) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (zero_query_answer<D>(qs, w == 0 ? k : 0, idx_base, lane, out_s, out_i)) {  // (no scan; every wave takes the same branch,
    __syncthreads();                                                             // wave 0 writes)
    return;
  }
  float es = RG_NEG_INF;  // lane p < k: entry p of this wave's sorted list
  int ei = INT_MAX;
  float kth_s = RG_NEG_INF;
  int kth_i = INT_MAX;
  for (int64_t base = (int64_t)w * 64; base < N; base += 256) {
    const int key = base + lane < N ? (int)(base + lane) : -1;
    const float sc = coop_scores<D>(qs, Kn, key, lane, tile[w]);
    unsigned long long pend = __ballot(key >= 0 && cand_better(sc, key, kth_s, kth_i));
    while (pend) {
      const int src = __ffsll((long long)pend) - 1;
      pend &= pend - 1;
      const float s = __shfl(sc, src);
      const int id = __shfl(key, src);
      const unsigned long long ahead = __ballot(lane < k && cand_better(es, ei, s, id));
      const int pos = __popcll(ahead);
      const float us = __shfl_up(es, 1);
      const int ui = __shfl_up(ei, 1);
      if (pos < k) {
        if (lane == pos) {
          es = s;
          ei = id;
        } else if (lane > pos && lane < k) {
          es = us;
          ei = ui;
        }
      }
      kth_s = __shfl(es, k - 1);
      kth_i = __shfl(ei, k - 1);
      pend &= __ballot(key >= 0 && cand_better(sc, key, kth_s, kth_i));
    }
  }
  if (lane < 32) {
    ps[w][lane] = lane < k ? es : RG_NEG_INF;
    pi[w][lane] = (lane < k && ei != INT_MAX) ? (int64_t)ei : INT64_MAX;
  }
  __syncthreads();
  if (w == 0) {  // 4 k <= 128 partial winners: two per lane
    float s2[2];
    int id2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u;
      const bool have = e < 4 * k;
      s2[u] = have ? ps[e / k][e % k] : RG_NEG_INF;
      const int64_t pv = have ? pi[e / k][e % k] : INT64_MAX;
      id2[u] = pv >= INT_MAX ? INT_MAX : (int)pv;
    }
    wave_select<2>(s2, id2, k, lane, idx_base, out_s, out_i);
  }
  __syncthreads();
}

// Small batches: one WORKGROUP per query.  The narrow kernel's wave walks its lane's candidates one after the other,
// each a latency-bound chain of row loads, and a few hundred waves do not hide that; here four waves take a quarter of
// the list each, leave their top-k in LDS, and wave 0 merges the four (and the previous level's winners, which ride
// with wave 0's quarter).
// SLICED (a handful of queries: gridDim.y = S workgroups per query): a workgroup rescans only sub-list blockIdx.y of
// the query (the direct kernel filled S of them) and leaves its k winners (local ids) in part_s / part_i [B][S][k]; the
// query's last workgroup to finish (a ticket in the query's counter line) merges them -- no second launch.
// One workgroup walking ~800 candidates of a lone query is ~30 us of dependent row gathers; eight of them take ~8.
#ifdef RG_WIDE_TIMING
__device__ unsigned long long g_wide_t[16];
#define RG_WSTAMP(i_) if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_wide_t[i_] = wall_clock64()
#else
#define RG_WSTAMP(i_)
#endif
template <int D, bool SLICED, bool COOP>
__global__ void __launch_bounds__(256) topk_rescore_wide_kernel(const float* __restrict__ Qn, const float* __restrict__ Kn,
                                                                int* __restrict__ count,
                                                                const int* __restrict__ cand, int64_t B, int64_t N, int cap,
                                                                int cs, int k, int64_t idx_base, const float* prev_s,
                                                                const int64_t* prev_i, int final_level, float* out_s,
                                                                int64_t* out_i, int* __restrict__ overflow,
                                                                int* __restrict__ overflow_list,
                                                                unsigned char* __restrict__ flag,
                                                                float* __restrict__ part_s, int* __restrict__ part_i,
                                                                int* __restrict__ cstat) {
  __shared__ float4 qs[D / 4];
  __shared__ float ps[4][32];
  __shared__ int64_t pi[4][32];
  // COOP (up to 256 queries): rows fetched cooperatively through a per-wave LDS tile (coop_scores) -- a lane walking its
  // own 1-KiB row is a chain of ~8 memory latencies; SLICED also scans the bank through it when a list overflowed
  __shared__ __attribute__((aligned(16))) float tile[(COOP || SLICED) ? 4 : 1][(COOP || SLICED) ? 64 * RESCORE_LD : 4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = blockIdx.x;
  RG_WSTAMP(0);
  if (threadIdx.x < D / 4) qs[threadIdx.x] = reinterpret_cast<const float4*>(Qn + b * D)[threadIdx.x];
  __syncthreads();
  RG_WSTAMP(1);
  // SLICED: sub-list blockIdx.y of the query (the direct kernel filled gridDim.y of them, cap / gridDim.y slots each);
  // the merge launch looks after overflow and empties the counters
  const int subcap = SLICED ? cap / (int)gridDim.y : cap;
  int n = count[b * cs + (SLICED ? (int)blockIdx.y : 0)];
  if (threadIdx.x == 0 && cstat && (b & 63) == 0) {  // (sliced: every sub-list's workgroup adds its part, the first one counts the query)
    atomicAdd(cstat, n);
    if (!SLICED || blockIdx.y == 0) atomicAdd(cstat + 3, 1);
  }
  bool over = false;
  if (n > subcap) {
    over = true;
    n = subcap;
  }
  if constexpr (!SLICED) {
    over = over || flag[b] != 0;
    if (final_level && over) {  // (block-uniform) listed for the fixup launch behind this one, which scans it in key slices
      __syncthreads();          // every wave has read the counter  (round 4 scanned up to 64 queries right here: ONE
                                // workgroup reading the whole bank, 5 - 7 ms per query where the sliced launch takes 0.5 - 1.7)
      // a ZERO query (flagged by the prepare launch) is answered here -- scores +0, rows in order -- and is neither listed
      // nor counted in *overflow: no kernel of the call counts zero queries (the one-wave kernels' zero_query_level alike)
      const bool zero = zero_query_answer<D>(qs, w == 0 ? k : 0, idx_base, lane, out_s + b * k, out_i + b * k);
      if (threadIdx.x == 0) {
        count[b * cs] = 0;
        if (!zero) {
          overflow_list[atomicAdd(overflow, 1)] = (int)b;
          flag[b] = 1;   // (listed: a speculative call's verify launch must not list it again)
        }
      }
      return;
    }
    if (threadIdx.x == 0 && over) flag[b] = 1;
  }
  const int lo0 = SLICED ? (int)blockIdx.y * subcap : 0;
  const int per = (n + 3) / 4;  // <= 512
  const int lo = w * per;
  const int nw = lo >= n ? 0 : (n - lo < per ? n - lo : per);
  const int* cb = cand + b * cap + lo0 + lo;
  const float* pps = (!SLICED && prev_s && w == 0) ? prev_s + b * k : nullptr;
  const int64_t* ppi = (!SLICED && prev_i && w == 0) ? prev_i + b * k : nullptr;
#define RG_RESCORE(NS_) rescore_query<D, NS_, COOP>(qs, Kn, cb, nw, lane, k, 0, pps, ppi, ps[w], pi[w], tile[w])
  if (COOP && nw <= 16) rescore_query<D, 1, true, true>(qs, Kn, cb, nw, lane, k, 0, pps, ppi, ps[w], pi[w], tile[w]);
  else if (nw <= 64) RG_RESCORE(1);
  else if (nw <= 128) RG_RESCORE(2);
  else if (nw <= 256) RG_RESCORE(4);
  else RG_RESCORE(8);
#undef RG_RESCORE
  RG_WSTAMP(2);
  __syncthreads();
  RG_WSTAMP(3);
  if constexpr (!SLICED) {
    if (threadIdx.x == 0) count[b * cs] = 0;  // every wave has read it: the next level starts from an empty list
  }
  if (w == 0) {  // 4 k <= 128 partial winners: two per lane
    float s[2];
    int id[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u;
      const bool have = e < 4 * k;
      s[u] = have ? ps[e / k][e % k] : RG_NEG_INF;
      const int64_t pv = have ? pi[e / k][e % k] : INT64_MAX;
      id[u] = pv >= INT_MAX ? INT_MAX : (int)pv;
    }
    if constexpr (SLICED) {
      // (wave_select writes 64-bit ids: staged through LDS, stored as the 32-bit local ids the merge expects)
      wave_select<2>(s, id, k, lane, 0, ps[0], pi[0]);
      __builtin_amdgcn_wave_barrier();
      const int64_t slot = (b * gridDim.y + blockIdx.y) * k;
      if (lane < k) {
        part_s[slot + lane] = ps[0][lane];
        part_i[slot + lane] = pi[0][lane] >= INT_MAX ? INT_MAX : (int)pi[0][lane];
      }
    } else {
      wave_select<2>(s, id, k, lane, final_level ? idx_base : 0, out_s + b * k, out_i + b * k);
    }
  }
  if constexpr (SLICED) {
    // ---- the query's LAST workgroup to get here merges the S slices' winners (S k <= 256: four per lane) with the
    // previous level's, does the level's bookkeeping (overflow flag / count, empty lists for the next level) and writes
    // the running result; a query that overflowed a sub-list gets the exact scan right here on the final level, so
    // these calls need no merge launch and no separate fallback launch.
    __shared__ int last_sh;
    const int S = (int)gridDim.y;
    int* cnt = count + b * cs;  // [0, S): the sub-lists' counters; [FILTER_TICKET_SLOT]: workgroups done
    RG_WSTAMP(4);
    if (w == 0) {
      __threadfence();  // this slice's winners are visible device-wide before its ticket
      RG_WSTAMP(5);
      if (lane == 0) last_sh = atomicAdd(cnt + FILTER_TICKET_SLOT, 1) == S - 1;
    }
    __syncthreads();
    RG_WSTAMP(6);
    if (!last_sh) return;
    __threadfence();
    RG_WSTAMP(7);
    int nmax = 0;
    for (int s_ = 0; s_ < S; ++s_) nmax = max(nmax, cnt[s_]);
    const bool over_q = flag[b] != 0 || nmax > subcap;
    __syncthreads();  // every thread has read the counters
    if (threadIdx.x < S) cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) cnt[FILTER_TICKET_SLOT] = 0;
    if (final_level && over_q) {  // (block-uniform) the fixup launch behind this one scans the bank for it, in key slices
      const bool zero = zero_query_answer<D>(qs, w == 0 ? k : 0, idx_base, lane, out_s + b * k, out_i + b * k);   // (see above)
      if (threadIdx.x == 0 && !zero) {
        overflow_list[atomicAdd(overflow, 1)] = (int)b;
        flag[b] = 1;
      }
      return;
    }
    if (w != 0) return;
    if (lane == 0 && over_q) flag[b] = 1;
    auto part_entry = [&](int e, float& sv, int& iv) {  // entries [0, S k): the slices' winners; [S k, S k + k): the previous level's
      sv = RG_NEG_INF;
      iv = INT_MAX;
      if (e < S * k) {  // (the other workgroups' stores: read past this CU's and XCD's caches)
        sv = __hip_atomic_load(part_s + b * S * k + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        iv = __hip_atomic_load(part_i + b * S * k + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (prev_s && e < S * k + k) {
        sv = prev_s[b * k + e - S * k];
        const int64_t pv = prev_i[b * k + e - S * k];
        iv = pv >= INT_MAX ? INT_MAX : (int)pv;
      }
    };
    RG_WSTAMP(8);
    if (S * k + k <= 128) {  // (wave-uniform) two slots per lane: the rank-by-counting selection
      float s[2];
      int id[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) part_entry(lane + 64 * u, s[u], id[u]);
      wave_select<2>(s, id, k, lane, final_level ? idx_base : 0, out_s + b * k, out_i + b * k);
    } else {
      float s[5];
      int id[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) part_entry(lane + 64 * u, s[u], id[u]);
      wave_select<5>(s, id, k, lane, final_level ? idx_base : 0, out_s + b * k, out_i + b * k);
    }
    RG_WSTAMP(9);
  }
}
