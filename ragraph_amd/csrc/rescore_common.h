// Device helpers shared by the rescoring kernels of the bf16-filtered exact top-k (topk_filter.hip) and the single-launch
// small-bank kernel (topk_fused.hip): canonical selection among a wave's (score, id) pairs and the exact fp32 rescoring
// of a candidate list (the k = 0..D-1 fmaf chain from +0: the chain the f32 MFMA and the oracle compute).
#pragma once
#include "filter_common.h"
#include <climits>

namespace ragraph {

// ---- canonical top-k of a wave's 64 NSL (score, id) pairs (score descending, id ascending; k <= 32) -------------------
// A pair travels as one 64-bit key whose unsigned order is the canonical order: the score's bits made monotone in the
// upper word, ~id in the lower; an empty slot (-inf, INT_MAX) and a pair already taken are key 0, which decodes to
// (-inf, INT64_MAX).  Round r: every lane's best remaining key, the wave's maximum of those -- four DPP steps inside
// each row of 16 lanes (quad_perm / row_half_mirror / row_mirror: no LDS traffic) and the four row maxima through
// v_readlane, i.e. a wave-uniform value -- which lane r keeps and every lane strikes from its slots.  (The __shfl_xor
// butterfly this replaces was twelve dependent ds_bpermute per round: ~0.8 us a round, 8 us a call at k = 10, three
// such calls in a row in the rescoring of a small batch.)
__device__ __forceinline__ unsigned select_ord(float f) {
  unsigned u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;  // -0 == +0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float select_unord(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

template <int CTRL>
__device__ __forceinline__ void select_dpp_max(unsigned& hi, unsigned& lo) {
  const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xF, 0xF, false);
  const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xF, 0xF, false);
  const bool take = ohi > hi || (ohi == hi && olo > lo);
  hi = take ? ohi : hi;
  lo = take ? olo : lo;
}

template <int NSL>
__device__ __forceinline__ void wave_select(const float (&s)[NSL], const int (&id)[NSL], int k, int lane, int64_t base,
                                            float* out_s, int64_t* out_i, float* kth_out = nullptr) {
  // kth_out (optional, e.g. an LDS word): receives the k-th winner's score, -inf when fewer than k pairs exist
  unsigned khi[NSL], klo[NSL];
#pragma unroll
  for (int u = 0; u < NSL; ++u) {
    const bool empty = id[u] == INT_MAX;
    khi[u] = empty ? 0u : select_ord(s[u]);
    klo[u] = empty ? 0u : ~(unsigned)id[u];
  }
  if constexpr (NSL <= 2) {
    // Up to 128 pairs: RANK BY COUNTING instead of k rounds.  Every key is broadcast once (v_readlane, a wave-uniform
    // value) and each lane counts the keys greater than its own: 64 NSL x NSL 64-bit compares, ~1 us at NSL = 2 whatever
    // k is, where the rounds take ~0.35 us each.  A pair with rank r < k is winner r.  Distinct non-empty pairs have
    // distinct keys (a key index enters a query's lists once per call), so ranks are unique; empty slots (key 0) fill
    // the places behind the last real pair.
    unsigned long long mine[NSL];
    int rank[NSL];
    int n_real = 0;
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      mine[u] = ((unsigned long long)khi[u] << 32) | klo[u];
      rank[u] = 0;
      n_real += __popcll(__ballot(mine[u] != 0ull));
    }
#pragma unroll
    for (int v = 0; v < NSL; ++v) {
      // (only up to the slot's last pair: a list of 40 candidates leaves most of its second slot empty)
      const unsigned long long nz = __ballot(mine[v] != 0ull);
      const int top = nz ? 64 - __clzll((long long)nz) : 0;
      for (int o = 0; o < top; ++o) {
        const unsigned long long src = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)khi[v], o) << 32) |
                                       (unsigned)__builtin_amdgcn_readlane((int)klo[v], o);
#pragma unroll
        for (int u = 0; u < NSL; ++u) rank[u] += src > mine[u] ? 1 : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      if (mine[u] != 0ull && rank[u] < k) {
        out_s[rank[u]] = select_unord(khi[u]);
        out_i[rank[u]] = (int64_t)(int)~klo[u] + base;
      }
    }
    if (lane < k && lane >= n_real) {
      out_s[lane] = RG_NEG_INF;
      out_i[lane] = INT64_MAX;
    }
    if (kth_out) {
#pragma unroll
      for (int u = 0; u < NSL; ++u)
        if (mine[u] != 0ull && rank[u] == k - 1) *kth_out = select_unord(khi[u]);
      if (lane == 0 && n_real < k) *kth_out = RG_NEG_INF;
    }
    return;
  }
  unsigned my_hi = 0u, my_lo = 0u;  // lane r: the r-th winner
  for (int r = 0; r < k; ++r) {
    unsigned bh = khi[0], bl = klo[0];
#pragma unroll
    for (int u = 1; u < NSL; ++u) {
      const bool take = khi[u] > bh || (khi[u] == bh && klo[u] > bl);
      bh = take ? khi[u] : bh;
      bl = take ? klo[u] : bl;
    }
    select_dpp_max<0xB1>(bh, bl);   // quad_perm [1,0,3,2]: lane ^ 1
    select_dpp_max<0x4E>(bh, bl);   // quad_perm [2,3,0,1]: lane ^ 2
    select_dpp_max<0x141>(bh, bl);  // row_half_mirror: the other quad of the 8
    select_dpp_max<0x140>(bh, bl);  // row_mirror: the other half of the 16
    unsigned mh = (unsigned)__builtin_amdgcn_readlane((int)bh, 0), ml = (unsigned)__builtin_amdgcn_readlane((int)bl, 0);
#pragma unroll
    for (int row = 1; row < 4; ++row) {
      const unsigned h = (unsigned)__builtin_amdgcn_readlane((int)bh, 16 * row);
      const unsigned l = (unsigned)__builtin_amdgcn_readlane((int)bl, 16 * row);
      const bool take = h > mh || (h == mh && l > ml);
      mh = take ? h : mh;
      ml = take ? l : ml;
    }
    if (lane == r) {
      my_hi = mh;
      my_lo = ml;
    }
    if ((mh | ml) == 0u) break;  // nothing left: the remaining winners stay empty
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      const bool hit = khi[u] == mh && klo[u] == ml;
      khi[u] = hit ? 0u : khi[u];
      klo[u] = hit ? 0u : klo[u];
    }
  }
  if (lane < k) {
    const bool empty = (my_hi | my_lo) == 0u;
    out_s[lane] = empty ? RG_NEG_INF : select_unord(my_hi);
    out_i[lane] = empty ? INT64_MAX : (int64_t)(int)~my_lo + base;
    if (kth_out && lane == k - 1) *kth_out = empty ? RG_NEG_INF : select_unord(my_hi);
  }
}

constexpr int RESCORE_LD = 68;  // floats per staged row: 16-B aligned, and 16 lanes x ds_read_b128 hit 64 distinct banks

// Exact scores of 64 candidates (lane l: key `key`, -1 = none) with the rows fetched COOPERATIVELY: a load instruction
// covers four rows x 256 B (16 lanes per row, coalesced) instead of one 16-B piece of 64 different rows, which is what
// the texture path's line rate pays for; the 64 x 64-float block goes through the wave's LDS tile and every lane then
// runs its own candidate's fmaf chain over it in natural order -- the same chain, so the same bits.
// ROWS = 32: candidates in lanes 0..31 only, a tile of 32 rows (half the LDS and half the load registers: the scored
// rescoring of large calls, whose second round is ~25 rows, runs three waves per SIMD with it instead of two).
template <int D, int ROWS = 64>
__device__ __forceinline__ float coop_scores(const float4* __restrict__ qrow, const float* __restrict__ Kn, int key,
                                             int lane, float* sm) {
  static_assert(ROWS == 64 || ROWS == 32, "whole or half tiles");
  constexpr int NT = ROWS / 4;  // load instructions per 64-float block: four rows x 256 B each
  float acc = 0.f;
  const int rr = lane >> 4, cc = lane & 15;
  constexpr int NDC = D / 64;
  // the next 64-float block's loads are in flight while this one is staged and consumed: a wave's chain is one
  // memory latency per list, not one per block (the kernel runs eight waves per CU and lives on latency hiding)
  int krow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) krow[t] = __shfl(key, 4 * t + rr);
  float4 v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (krow[t] >= 0) v[t] = *reinterpret_cast<const float4*>(Kn + (int64_t)krow[t] * D + cc * 4);
  }
#pragma unroll 1
  for (int dc = 0; dc < NDC; ++dc) {
    __builtin_amdgcn_wave_barrier();  // (single wave: LDS executes its requests in order; only the compiler must not reorder)
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<float4*>(sm + (4 * t + rr) * RESCORE_LD + cc * 4) = v[t];
    if (dc + 1 < NDC) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (krow[t] >= 0) v[t] = *reinterpret_cast<const float4*>(Kn + (int64_t)krow[t] * D + cc * 4 + (dc + 1) * 64);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (ROWS == 64 || lane < ROWS) {
#pragma unroll
      for (int e4 = 0; e4 < 16; ++e4) {
        const float4 kv = *reinterpret_cast<const float4*>(sm + lane * RESCORE_LD + e4 * 4);
        const float4 qv = qrow[dc * 16 + e4];
        acc = fmaf(qv.x, kv.x, acc);
        acc = fmaf(qv.y, kv.y, acc);
        acc = fmaf(qv.z, kv.z, acc);
        acc = fmaf(qv.w, kv.w, acc);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  return acc;
}

// The same for at most 16 candidates (lanes 0..15; sharded banks and late levels leave a query a handful): only rows
// 0..15 exist, so ALL of a row's 64-float blocks are fetched at once -- one memory latency per query instead of one per
// block -- and staged block by block; the chains are the same.
template <int D>
__device__ __forceinline__ float coop_scores_few(const float4* __restrict__ qrow, const float* __restrict__ Kn, int key,
                                                 int lane, float* sm) {
  const int rr = lane >> 4, cc = lane & 15;
  constexpr int NDC = D / 64;
  int krow[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) krow[t] = __shfl(key, 4 * t + rr);
  float4 v[NDC][4];
#pragma unroll
  for (int dc = 0; dc < NDC; ++dc)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      v[dc][t] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (krow[t] >= 0) v[dc][t] = *reinterpret_cast<const float4*>(Kn + (int64_t)krow[t] * D + cc * 4 + dc * 64);
    }
  float acc = 0.f;
#pragma unroll
  for (int dc = 0; dc < NDC; ++dc) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(sm + (4 * t + rr) * RESCORE_LD + cc * 4) = v[dc][t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 16) {
#pragma unroll
      for (int e4 = 0; e4 < 16; ++e4) {
        const float4 kv = *reinterpret_cast<const float4*>(sm + lane * RESCORE_LD + e4 * 4);
        const float4 qv = qrow[dc * 16 + e4];
        acc = fmaf(qv.x, kv.x, acc);
        acc = fmaf(qv.y, kv.y, acc);
        acc = fmaf(qv.z, kv.z, acc);
        acc = fmaf(qv.w, kv.w, acc);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  return acc;
}

// Rescoring of one query by one wave with NS candidate slots per lane (n <= 64 * NS reserved slots): exact scores (one
// lane per candidate, the k = 0..D-1 fmaf chain from +0), merge with the previous level's winners, canonical top-k.
// COOP: rows staged through the LDS tile `sm` (coop_scores); else every lane reads its own row.
// CS: ints per list entry (2: the scored lists' {key, I} pairs, of which only the key is read here).
template <int D, int NS, bool COOP = false, bool FEW = false, int CS = 1>
__device__ __forceinline__ void rescore_query(const float4* __restrict__ qrow, const float* __restrict__ Kn,
                                              const int* __restrict__ cand, int n, int lane, int k, int64_t base,
                                              const float* prev_s, const int64_t* prev_i, float* out_s, int64_t* out_i,
                                              float* sm = nullptr) {
  const int first_keys = cand[lane * CS];  // (no dependence on n: the list has >= 64 slots; issued next to the count's load)
  float s[NS + 1];
  int id[NS + 1];
  // the previous level's winners ride along as already-scored candidates (lane l holds entry l; k <= 32)
  s[NS] = RG_NEG_INF;
  id[NS] = INT_MAX;
  if (prev_s && lane < k) {
    s[NS] = prev_s[lane];
    const int64_t pv = prev_i[lane];
    id[NS] = pv >= INT_MAX ? INT_MAX : (int)pv;
  }
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int c = lane + 64 * u;
    s[u] = RG_NEG_INF;
    id[u] = INT_MAX;
    int key = -1;
    if (u == 0) key = c < n ? first_keys : -1;
    else if (c < n) key = cand[c * CS];
    if constexpr (COOP) {
      if (64 * u < n) {  // wave-uniform
        const float acc = FEW ? coop_scores_few<D>(qrow, Kn, key, lane, sm) : coop_scores<D>(qrow, Kn, key, lane, sm);
        if (key >= 0) {
          s[u] = acc;
          id[u] = key;
        }
      }
    } else if (key >= 0) {
      const float4* kr = reinterpret_cast<const float4*>(Kn + (int64_t)key * D);
      float acc = 0.f;
#pragma unroll 8
      for (int d4 = 0; d4 < D / 4; ++d4) {
        const float4 kv = kr[d4], qv = qrow[d4];
        acc = fmaf(qv.x, kv.x, acc);
        acc = fmaf(qv.y, kv.y, acc);
        acc = fmaf(qv.z, kv.z, acc);
        acc = fmaf(qv.w, kv.w, acc);
      }
      s[u] = acc;
      id[u] = key;
    }
  }
  wave_select<NS + 1>(s, id, k, lane, base, out_s, out_i);
}

// The exact fallback for one query by ONE wave (a candidate list overflowed: near-duplicate banks, zero queries): every
// lane scores its own key per step with the fp32 chain, the wave keeps the sorted list (lane p = entry p).  Slow -- a
// full scan of the bank by 64 lanes -- and rare; the four-wave, LDS-staged form is exact_scan_query (topk_filter.hip).
// A ZERO query (the usual reason for an overflow on an ordinary bank: every score is +0, so every key passes any bound)
// needs no scan: all chains end at +0 and the canonical order is the index order.  Wave-uniform result.
template <int D>
__device__ __forceinline__ bool zero_query_answer(const float4* qrow, int k, int64_t idx_base, int lane,
                                                  float* __restrict__ out_s, int64_t* __restrict__ out_i) {
  bool nz = false;
  for (int d4 = lane; d4 < D / 4; d4 += 64) {
    const float4 v = qrow[d4];
    nz = nz || ((__float_as_uint(v.x) | __float_as_uint(v.y) | __float_as_uint(v.z) | __float_as_uint(v.w)) & 0x7FFFFFFFu) != 0u;
  }
  if (__any(nz)) return false;
  if (lane < k) {
    out_s[lane] = 0.f;
    out_i[lane] = idx_base + lane;
  }
  return true;
}

template <int D>
__device__ __forceinline__ void exact_scan_wave(const float4* qrow, const float* __restrict__ Kn, int64_t N, int k,
                                                int64_t idx_base, int lane, float* __restrict__ out_s,
                                                int64_t* __restrict__ out_i) {
  if (zero_query_answer<D>(qrow, k, idx_base, lane, out_s, out_i)) return;
  float es = RG_NEG_INF;
  int ei = INT_MAX;
  float kth_s = RG_NEG_INF;
  int kth_i = INT_MAX;
  for (int64_t base = 0; base < N; base += 64) {
    const int key = base + lane < N ? (int)(base + lane) : -1;
    float sc = 0.f;
    if (key >= 0) {
      const float4* kr = reinterpret_cast<const float4*>(Kn + (int64_t)key * D);
#pragma unroll 8
      for (int d4 = 0; d4 < D / 4; ++d4) {
        const float4 kv = kr[d4], qv = qrow[d4];
        sc = fmaf(qv.x, kv.x, sc);
        sc = fmaf(qv.y, kv.y, sc);
        sc = fmaf(qv.z, kv.z, sc);
        sc = fmaf(qv.w, kv.w, sc);
      }
    }
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
  if (lane < k) {
    out_s[lane] = es;
    out_i[lane] = ei == INT_MAX ? INT64_MAX : (int64_t)ei + idx_base;
  }
}

}  // namespace ragraph
