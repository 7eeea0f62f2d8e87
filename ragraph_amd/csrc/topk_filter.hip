// Exact cosine top-k through a bf16 MFMA filter  (SimilarityFunctions.py:6-16 + ToyGraphBase.py:66-67, any batch size
// against banks of >= 64 k keys; D = 64 is the edge flavour's RAGraph_edge/modules/RAGraph.py:298-324).
//
// The fp32 tile kernel (topk_cosine.hip) spends 2·B·N·D fp32 MFMA flops; the bf16 matrix cores are 16x faster.  This
// path returns the SAME bits with most of the work on them:
//   1. a first lower bound theta[q] of the final k-th best exact score of q.  Banks of >= 8192 keys: the BOUND pass --
//      this file's kernel over a prefix of the bank, recording per query the best approximate score of each of G = 4 k
//      parts (a handful of queries: k); the parts' best keys are distinct, so k keys score at least (the k-th largest of
//      those maxima) - eps(q).  Smaller banks: the k-th score of an exact top-k over the first n0 keys (the fp32 tile
//      kernel, or up to 16384 queries a score slab + topk_rows).
//   2. filter (this file, bf16 MFMA): approximate scores s~ = bf16(q)·bf16(key), fp32 accumulate, over the next, larger
//      part of the bank.  With q^ = q + dq, k^ = k + dk: |s~ - s| <= |dq||k| + |q||dk| + |dq||dk| + accumulation error
//      (Cauchy-Schwarz) <= eps(q), computed from the query's actual |dq| and the bank's largest |dk| (<= 2^-7 in the worst
//      case, ~0.003 typically) -- a bound for EVERY pair.  A key of the exact top-k has s >= theta[q], hence
//      s~ >= theta[q] - eps(q): every key that passes goes to the query's candidate list; nothing else can be in the result.
//   3. rescoring: the exact score of every candidate as the fp32 fmaf chain in natural k order from +0 (one lane per
//      candidate) -- the chain the f32 MFMA and the oracle compute, so the same bits -- merged with the previous level's
//      winners, canonical top-k (score descending, index ascending).  That is the exact top-k of everything seen so far
//      and a tighter theta for the next level.  The schedule (n0, one to three levels) depends on the batch size:
//      filter_schedule() below, readable through ragraph_topk_cosine_filtered_plan.
// A query whose candidate list overflows its capacity (adversarial banks: thousands of keys within eps of the k-th
// best) is recomputed by an exact fp32 scan of the bank ON THE DEVICE (exact_scan_query: inside the sliced rescoring
// launch for <= 64 queries; otherwise topk_overflow_fixup_kernel, one launch behind the last level that cuts the scans of
// a few queries into key slices); *overflow only counts those rows.  The call never synchronises and reads nothing back.
// Levels may run on an int8 copy of the bank instead (v_mfma_i32_16x16x64_i8, integer thresholds: filter_common.h), whose
// candidate lists can carry the integer score that admitted each key (SCORED: topk_rescore_scored_kernel).
//
// Two filter kernels share one bank layout (filter_common.h: MFMA fragment order of v_mfma_f32_16x16x32_bf16):
// topk_filter_direct_kernel (topk_filter_direct.hip) for up to 256 queries, and this file's RING kernel above that:
// workgroup = 8 waves x 64 queries = 512 queries (x 32 = 256, x 128 = 1024 at D = 64 on long streams); a wave keeps its
// queries as QW/16 groups of B operands (D/8 VGPRs each) and streams the bf16 bank (2 D bytes per key) through a 4-slot
// LDS ring of 32 KiB stages (64 / 128 / 256 keys at D = 256 / 128 / 64) filled by LDS-DMA, one 1-KiB block per
// global_load_lds_dwordx4 -- the LDS image is the HBM image, conflict-free for the lanes' ds_read_b128 -- handed over by
// FULL/FREE counters.  One fragment read feeds QW/16 MFMAs.  Candidates leave the MFMA stream through wave-private LDS
// buffers (branch-free pass masks, ballot + mbcnt positions) and reach the per-query lists in global memory in flushes.
// Work plan: segment_plan.h with zero warm-up cost (there are no lists): every workgroup gets the same number of stages.
#include "filter_common.h"
#include "rescore_common.h"
#include <new>
#include "segment_plan.h"
#include <cmath>
#include <vector>
#include <type_traits>

namespace ragraph {

typedef __attribute__((address_space(3))) void lds_void_f;
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int D_>
struct FilterCfg {
  static constexpr int D = D_;
  static constexpr int WAVES = 8, THREADS = 512;
  static constexpr int ROW_BYTES = D * 2;                 // one bf16 key
  static constexpr int CR = D / 8;                        // 16-B chunks per row
  static constexpr int KSTEPS = D / 16;                   // 1-KiB blocks (A fragments) per 32-key sub-tile: 2 halves x KS32
  static constexpr int KS32 = D / 32;                     // MFMA k-steps (32 elements) per sub-tile: 8 / 4 / 2
  static constexpr int STAGE_BYTES = FILTER_STAGE_BYTES;
  static constexpr int STAGE_KEYS = STAGE_BYTES / ROW_BYTES;  // 64 / 128 / 256
  static constexpr int SUBS = STAGE_KEYS / 32;            // 32-key MFMA sub-tiles per stage: 2 / 4 / 8
  static constexpr int NSTEP = SUBS * KSTEPS;             // = 32 fragment steps (x QW/16 query groups) per stage for every D
  static constexpr int SLOTS = 4;                         // 128 KiB of ring; the hand-over protocol needs >= 3 slots
  static constexpr int DMAS = STAGE_BYTES / 1024 / WAVES; // 1 KiB DMA instructions per wave and stage = 4
  static constexpr int RPI = 1024 / ROW_BYTES;            // key rows per DMA instruction: 2 / 4 / 8
  static constexpr int CAND_BUF = 480;                    // entries of a wave's candidate buffer (8 B each)
  static constexpr size_t LDS_BYTES = (size_t)SLOTS * STAGE_BYTES + 64 + (size_t)WAVES * CAND_BUF * 8;
};
static_assert(FilterCfg<256>::NSTEP == 32 && FilterCfg<128>::NSTEP == 32 && FilterCfg<64>::NSTEP == 32, "32 steps per stage");

struct FilterParams {
  const float* Qn;        // [B,D] normalised queries (fp32)
  const uint16_t* Kb;     // bf16 keys in MFMA fragment order (filter_common.h), rows >= N zero
  FilterThr thr;          // how a query's pass threshold theta[q] - eps(q) is obtained (filter_common.h)
  int* count;             // [B][cstride] candidate slots reserved so far (filter_count_stride)
  int cstride;
  const uint16_t* Qb;     // queries as bf16 B operands in fragment order (padded to 32s), or NULL: convert from Qn
  int* cand;              // [B,cap] candidate key indices (local to this shard)
  int64_t B, N;           // N = end of the key range (keys >= N never pass)
  int cap;
  int64_t stage_base;     // first stage of the key range this launch filters
  int64_t qtiles, nstages_total;  // nstages_total = stages in the range
  int xcd_map, wgs_per_group, lb_min, depth[2];
  int partner_lead;       // > 0: a wave's priority follows its SIMD partner's progress (see the stage loop); 0: off
  // bound pass (BOUND kernels): per query, the maxima of `ngroups` consecutive stage ranges of the launch's key range
  int* gmax;              // [B, ngroups] as order-preserving ints (f2ord), pre-filled with f2ord(-inf)
  int ngroups;
};

__device__ __forceinline__ void fring_wait(unsigned* ctr, unsigned target) {
  while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}
#ifndef RG_RING_RELAXED
#define RG_RING_RELAXED 1
#endif
// What a signal publishes is ordered by hand: `freec` (this wave is done READING the slot) follows the wave's own ds_reads in
// the LDS queue, which executes a wave's operations in order; `full` (this wave's share of the stage has LANDED) follows an
// explicit s_waitcnt vmcnt(0).  A release fence here would wait for every outstanding store of a candidate flush as well.
// The COMPILER must keep that order too: the asm ds_read / s_waitcnt statements in front of a signal carry no memory
// clobber, so an empty asm with one pins the relaxed atomic behind them (no instruction, no wait).
__device__ __forceinline__ void fring_signal(unsigned* ctr, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, RG_RING_RELAXED ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// fp32 -> bf16 (round to nearest even) of the bank in MFMA fragment order, rows [N, Npad) zero so the stream never needs
// a tail clamp, and
// max_k |dk|^2 of the bank (see FILTER_EPS_SLACK) by an integer max: non-negative floats order like their bit patterns.
template <int D>
__global__ void __launch_bounds__(256) keys_to_bf16_kernel(const float* __restrict__ Kn, int64_t N, int64_t Npad,
                                                           uint16_t* __restrict__ Kb, unsigned* __restrict__ max_err2) {
  constexpr int TPR = D / 8;  // threads per row (one thread = 8 elements): 8 / 16 / 32, inside one half-wave
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = i / TPR;
  bf16x8 o;
  float e2 = 0.f;
  if (i < Npad * TPR && row < N) {
    const float4 a = reinterpret_cast<const float4*>(Kn)[2 * i], b = reinterpret_cast<const float4*>(Kn)[2 * i + 1];
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o[e] = (__bf16)x[e];
      const float d = x[e] - (float)o[e];
      e2 = fmaf(d, d, e2);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)0.f;
  }
  if (i < Npad * TPR) {
    // fragment order (filter_common.h): 16-B piece c = 4 t + g of key row 32 u + 16 h + j goes to block 2 t + h of
    // sub-tile u, lane j + 16 g
    const int c = (int)(i % TPR);
    const int64_t dst = filter_block_offset(row >> 5, D, c >> 2, (int)(row >> 4) & 1) + (((c & 3) * 16 + (int)(row & 15)) << 4);
    *reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(Kb) + dst) = o;
  }
#pragma unroll
  for (int off = TPR / 2; off >= 1; off >>= 1) e2 += __shfl_xor(e2, off);
  // (a plain read first: after the first few rows almost none beats the running maximum, so almost none pays for the
  // atomic on this one address)
  if ((threadIdx.x & (TPR - 1)) == 0 &&
      __float_as_uint(e2) > __hip_atomic_load(max_err2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(max_err2, __float_as_uint(e2));
}

// The int8 copy (filter_common.h, "TWO SCALES"): the granules' largest |k_i| (and the bank's), the cut between the two classes,
// then quantise + lay out + each class's largest |dk|^2 + the class bits.  The tail row and the class words are zeroed by the
// caller before the first kernel.
template <int D>
__global__ void __launch_bounds__(256) i8_granule_absmax_kernel(const float* __restrict__ Kn, int64_t N, float* __restrict__ gmax,
                                                                unsigned* __restrict__ tail8) {
  constexpr int GK = filter_i8_granule_keys(D);
  const int64_t row0 = (int64_t)blockIdx.x * GK;
  const int64_t rows = N - row0 < GK ? N - row0 : GK;   // (<= 0: a granule of padding)
  const int64_t n4 = rows > 0 ? rows * (D / 4) : 0;
  const float4* src = reinterpret_cast<const float4*>(Kn + row0 * D);
  unsigned m = 0u;
  for (int64_t i = threadIdx.x; i < n4; i += 256) {
    const float4 v = src[i];
    m = max(max(m, __float_as_uint(fabsf(v.x))), max(__float_as_uint(fabsf(v.y)), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w)))));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
  __shared__ unsigned wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    gmax[blockIdx.x] = __uint_as_float(m);
    if (m > __hip_atomic_load(tail8 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(tail8 + 2, m);
  }
}

// The cut (one workgroup).  A histogram of the granules' maxima over the upper 15 bits of their (non-negative) float
// patterns -- bins 1.6 % wide -- then, for every occupied bin's upper edge x as the cut, the modelled candidates
//   (granules <= x) f(x) + (granules > x) f(x_max),   f(x) = exp(lambda |dk|(x)),  |dk|(x) = (x / 127) sqrt(D / 12)
// (uniform rounding errors of a row quantised on the grid x / 127; lambda = 60: the bench bank's candidates triple when eps
// grows from the bf16 bound's 0.004 to the single scale's 0.0215).  Whatever comes out is only a matter of speed: every
// class's error is MEASURED by the quantising kernel and the bounds use the measurements.  lambda <= 0: cut = the maximum.
constexpr int I8_CUT_BINS = 1 << 13;   // (unit rows: |k_i| <= 1 = bin 8128; anything larger shares the last bin, which is never a cut)
__global__ void __launch_bounds__(1024) i8_cut_kernel(const float* __restrict__ gmax, int64_t granules, int D, float lambda,
                                                      unsigned* __restrict__ tail8) {
  __shared__ int hist[I8_CUT_BINS];
  __shared__ int tsum[1024];
  __shared__ float tcost[1024];
  __shared__ int tbin[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < I8_CUT_BINS; i += 1024) hist[i] = 0;
  __syncthreads();
  for (int64_t i = tid; i < granules; i += 1024) atomicAdd(hist + min((int)(__float_as_uint(gmax[i]) >> 17), I8_CUT_BINS - 1), 1);
  __syncthreads();
  // thread t owns bins 16 t .. 16 t + 15: their sum, an exclusive prefix over the threads, then the cost of every occupied bin's
  // upper edge as the cut; the cheapest (ties: the lowest) wins
  constexpr int PER = I8_CUT_BINS / 1024;
  int mine = 0;
#pragma unroll
  for (int e = 0; e < PER; ++e) mine += hist[tid * PER + e];
  tsum[tid] = mine;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // (Hillis-Steele inclusive scan)
    const int v = tid >= off ? tsum[tid - off] : 0;
    __syncthreads();
    tsum[tid] += v;
    __syncthreads();
  }
  const float xmax = __uint_as_float(tail8[2]);
  const int top = min((int)(__float_as_uint(xmax) >> 17), I8_CUT_BINS - 1);
  const float c = lambda * sqrtf((float)D / 12.f) / 127.f;
  const float fmax_ = expf(c * xmax);
  float best = __builtin_huge_valf();
  int best_bin = -1;
  if (lambda > 0.f && xmax > 0.f) {
    int64_t below = tsum[tid] - mine;
    for (int e = tid * PER; e < tid * PER + PER && e < top; ++e) {
      if (hist[e] == 0) continue;
      below += hist[e];
      const float edge = __uint_as_float((unsigned)(e + 1) << 17);   // every maximum of bins <= e lies below it
      const float cost = (float)below * expf(c * edge) + (float)(granules - below) * fmax_;
      if (cost < best) {
        best = cost;
        best_bin = e;
      }
    }
  }
  tcost[tid] = best;
  tbin[tid] = best_bin;
  __syncthreads();
  for (int off = 512; off >= 1; off >>= 1) {
    if (tid < off && (tcost[tid + off] < tcost[tid] || (tcost[tid + off] == tcost[tid] && tbin[tid + off] >= 0 &&
                                                         (tbin[tid] < 0 || tbin[tid + off] < tbin[tid])))) {
      tcost[tid] = tcost[tid + off];
      tbin[tid] = tbin[tid + off];
    }
    __syncthreads();
  }
  if (tid != 0) return;
  float cut = xmax;   // (one class: no occupied bin below the top one, lambda <= 0, a zero bank -- or no cut beats it)
  if (tbin[0] >= 0 && tcost[0] < (float)granules * fmax_) cut = __uint_as_float((unsigned)(tbin[0] + 1) << 17);
  tail8[5] = __float_as_uint(cut);
  tail8[1] = __float_as_uint(cut / 127.f);
  tail8[4] = __float_as_uint(xmax / 127.f);
  tail8[7] = granules > INT_MAX ? (unsigned)INT_MAX : (unsigned)granules;
}

template <int D>
__global__ void __launch_bounds__(256) keys_to_i8_kernel(const float* __restrict__ Kn, int64_t N, int64_t Npad,
                                                         signed char* __restrict__ Kb8, unsigned* __restrict__ tail8,
                                                         const float* __restrict__ gmax, unsigned* __restrict__ cls) {
  constexpr int TPR = D / 16;  // threads per row (one thread = 16 elements = one lane's piece of a block): 4 / 8 / 16
  constexpr int GK = filter_i8_granule_keys(D);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = i / TPR;
  const int64_t gr = row / GK;
  const bool live = i < Npad * TPR;
  const bool heavy = live && gmax[gr] > __uint_as_float(tail8[5]);
  const float sk = __uint_as_float(tail8[heavy ? 4 : 1]);
  const float inv_sk = sk > 0.f ? 1.f / sk : 0.f;
  if (heavy && row == gr * GK && i == row * TPR) {   // the granule's first thread: its class bit
    atomicOr(cls + (gr >> 5), 1u << (gr & 31));
    atomicAdd(tail8 + 6, 1u);
  }
  unsigned w[4] = {0u, 0u, 0u, 0u};
  float e2 = 0.f;
  if (live && row < N && sk > 0.f) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 a = reinterpret_cast<const float4*>(Kn)[4 * i + c];
      const float x[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int qi = quantize_i8(x[e], inv_sk);
        w[c] |= ((unsigned)qi & 0xFFu) << (8 * e);
        const float d = fmaf(sk, (float)qi, -x[e]);
        e2 = fmaf(d, d, e2);
      }
    }
  }  // (an all-zero bank: the copy is zero, the error is the row itself: 0)
  if (live) {
    const int c = (int)(i % TPR);  // piece c = 4 t + g of the row
    const int64_t dst = filter_i8_block_offset(row >> 5, D, c >> 2, (int)(row >> 4) & 1) + (((c & 3) * 16 + (int)(row & 15)) << 4);
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Kb8) + dst) = make_uint4(w[0], w[1], w[2], w[3]);
  }
#pragma unroll
  for (int off = TPR / 2; off >= 1; off >>= 1) e2 += __shfl_xor(e2, off);
  e2 *= 1.000001f;  // (the fmaf's rounding of each difference)
  unsigned* slot = tail8 + (heavy ? 3 : 0);
  if ((threadIdx.x & (TPR - 1)) == 0 && __float_as_uint(e2) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(slot, __float_as_uint(e2));
}

// Everything a call needs before its first filter launch, in ONE launch (one wave per query): the normalised query row
// (the norm tree, sqrt and divisions of normalize_rows_kernel, so the same bits), |dq| of its bf16 rounding, an empty
// candidate list, a clear overflow flag, and -- before a bound pass -- the group maxima at -inf.
// Candidate statistics of a call (FILTER_STATS_INTS ints at the very end of the caller's workspace; zeroed and labelled by
// the prepare launch): every 64th query adds its candidate count of the level to cstat[0] and 1 to cstat[3] -- a sampled
// mean the owner of the bank reads back asynchronously (ragraph_amd/kernels_index.py: a bank whose int8 levels pass hundreds
// of candidates per query WITHOUT overflowing is slower on int8 than on bf16, and nothing else would notice).
// Layout: [0] magic, [1] levels, [2 + l] sampled candidates of level l, [5 + l] sampled queries, [8 + l] level l ran on
// int8, [11 + l] keys of level l, [14] queries of the call, [15] zero queries among them, [16] 1: the call filtered with a
// SPECULATIVE first bound (ragraph_topk_cosine_filtered_set_prior), [17] queries whose speculation failed (answered by the
// exact scan), [18] / [19] the smallest / largest final exact k-th best score of the call's queries as order-preserving
// ints (f2ord; what the owner of the bank builds the next call's prior from), [20] the call's final *overflow (so that ONE
// copy of these words tells the owner everything), [21..32) reserved.
constexpr int FILTER_STATS_INTS = 32;
constexpr int FILTER_STATS_MAGIC = 0x52414753;
__device__ __forceinline__ void note_candidates(int* cstat, int64_t b, int n) {
  if (cstat && (b & 63) == 0) {
    atomicAdd(cstat, n);
    atomicAdd(cstat + 3, 1);
  }
}
struct FilterStatsInit {
  int nlev, i8[3], keys[3];
};

constexpr int FILTER_FIX_MAX_Q = 1024;  // overflowed queries whose scan topk_overflow_fixup_kernel may cut into slices
constexpr int FILTER_FIX_SLICES = 16;   // at most (16 x 32 partial winners: eight per lane of the merging wave)

template <int D>
__global__ void __launch_bounds__(256) filter_prep_kernel(const float* __restrict__ Q, int64_t B, float* __restrict__ Qn,
                                                          float* __restrict__ eq, int* __restrict__ count,
                                                          unsigned char* __restrict__ flag, int* __restrict__ overflow,
                                                          int* __restrict__ gmax, int ngroups,
                                                          uint16_t* __restrict__ Qb, int cstride,
                                                          float* __restrict__ eq8, float* __restrict__ qscale,
                                                          signed char* __restrict__ Qb8, int* __restrict__ fix_done,
                                                          int* __restrict__ stats, FilterStatsInit si,
                                                          float* __restrict__ theta_init, float prior) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q == 0 && lane == 0) *overflow = 0;
  if (q == 0 && lane < FILTER_STATS_INTS && stats) {
    int v = 0;
    if (lane == 0) v = FILTER_STATS_MAGIC;
    else if (lane == 1) v = si.nlev;
    else if (lane >= 8 && lane < 11) v = si.i8[lane - 8];
    else if (lane >= 11 && lane < 14) v = si.keys[lane - 11];
    else if (lane == 14) v = B > INT_MAX ? INT_MAX : (int)B;
    else if (lane == 16) v = theta_init ? 1 : 0;
    else if (lane == 18) v = INT_MAX;   // (minimum of the k-th best scores: nothing recorded yet)
    else if (lane == 19) v = INT_MIN;
    stats[lane] = v;
  }
  if (theta_init && q < B && lane == 0) theta_init[q] = prior;   // a speculative first bound: the same for every query
  if (q < FILTER_FIX_MAX_Q && lane == 0) fix_done[q] = 0;  // tickets of topk_overflow_fixup_kernel
  if (q >= (Qb ? (B + 31) / 32 * 32 : B)) return;
  constexpr int NCH = D / 4;  // float4 chunks per row: 16 / 32 / 64 -- at most one per lane
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < NCH && q < B) v = reinterpret_cast<const float4*>(Q + q * D)[lane];
  float p = 0.f;
  p = fmaf(v.x, v.x, p);
  p = fmaf(v.y, v.y, p);
  p = fmaf(v.z, v.z, p);
  p = fmaf(v.w, v.w, p);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) p = __fadd_rn(p, __shfl_xor(p, off));
  const float d = fmaxf(sqrtf(p), 1e-12f);
  v.x = v.x / d; v.y = v.y / d; v.z = v.z / d; v.w = v.w / d;
  if (lane < NCH && q < B) reinterpret_cast<float4*>(Qn + q * D)[lane] = v;
  if (Qb && lane < NCH) {
    // the direct kernel's B operands (<= 256 queries): bf16 in fragment order (filter_common.h, DirectArgs::Qb); this
    // lane's elements 4 l .. 4 l + 3 are half of one 16-byte piece.  (The launch covers the padding queries of the last
    // group of 32 too: they get zero rows.)
    const int e0 = 4 * lane, t = e0 >> 5, gg = (e0 >> 3) & 3;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    char* base = reinterpret_cast<char*>(Qb) + ((q >> 4) * (D / 32) + t) * 1024 + (gg * 16 + (int)(q & 15)) * 16 + (e0 & 7) * 2;
    *reinterpret_cast<bf16x4*>(base) = o;
  }
  float e2 = 0.f;
  {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dd = x[e] - (float)(__bf16)x[e];
      e2 = fmaf(dd, dd, e2);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) e2 += __shfl_xor(e2, off);
  float e8 = 0.f, sq = 0.f;
  unsigned am = max(max(__float_as_uint(fabsf(v.x)), __float_as_uint(fabsf(v.y))), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w))));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) am = max(am, (unsigned)__shfl_xor((int)am, off));
  if (eq8) {  // (kernel-uniform) the query's int8 scale and rounding error (filter_common.h; the ring kernel re-quantises
              // the row with the SAME expression, so this is the error of the operands it multiplies)
    sq = __uint_as_float(am) / 127.f;
    unsigned w8 = 0u;
    if (sq > 0.f) {
      const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int qi = quantize_i8(x[e], 1.f / sq);
        w8 |= ((unsigned)qi & 0xFFu) << (8 * e);
        const float dd = fmaf(sq, (float)qi, -x[e]);
        e8 = fmaf(dd, dd, e8);
      }
    }
    if (Qb8 && lane < NCH) {
      // the direct kernel's int8 B operands (<= 256 queries), fragment order: this lane's elements 4 l .. 4 l + 3 are dword
      // l % 4 of the 16-byte piece l / 4 = 4 t + g of its row (padding queries of the last group of 32: zero rows)
      const int c = lane >> 2;
      char* base = reinterpret_cast<char*>(Qb8) + ((q >> 4) * (D / 64) + (c >> 2)) * 1024 + ((c & 3) * 16 + (int)(q & 15)) * 16 + (lane & 3) * 4;
      *reinterpret_cast<unsigned*>(base) = w8;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) e8 += __shfl_xor(e8, off);
  }
  if (q >= B) return;
  if (lane == 0) {
    eq[q] = sqrtf(e2) * 1.0000002f;  // (any summation order of the squares stays below this)
    // a ZERO query scores +0 against every key: within any bound of its k-th best, i.e. its lists can only overflow -- it
    // is flagged as overflowed from the start (nothing passes the filter for a flagged query: FilterThr::flag) and the
    // final level's scan path answers it without scanning (zero_query_answer)
    flag[q] = am == 0u ? 2 : 0;  // (2: a zero query -- the one-wave rescoring kernels answer it at the final level, uncounted)
    if (am == 0u && stats) atomicAdd(stats + 15, 1);  // (zero queries of the call: some kernels count them as overflowed, the owner of
                                                      // the bank subtracts them before it judges the bank)
    if (eq8) {
      eq8[q] = sqrtf(e8) * 1.000001f;
      qscale[q] = sq;
    }
  }
  if (lane < cstride) count[q * cstride + lane] = 0;
  if (gmax)
    for (int gi = lane; gi < ngroups; gi += 64) gmax[q * ngroups + gi] = f2ord(RG_NEG_INF);
}

// Sharded banks (ragraph_topk_cosine_filtered_sharded_f32): the bound a level filters with is kept in theta[B] so that
// the caller can sharpen it across the shards between the phases.  After the bound pass: theta = min over the parts of
// the part's best approximate score, minus eps; after an exact level 0 or a rescoring level: theta = max(theta, the
// shard's k-th exact score so far) (-inf while the shard has fewer than k candidates).
__global__ void __launch_bounds__(256) filter_theta_kernel(FilterThr t, int64_t B, int init, float* __restrict__ theta) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= B) return;
  float v;
  if (t.gmax) {
    int m = t.gmax[q * t.ngroups];
    for (int g = 1; g < t.ngroups; ++g) m = min(m, t.gmax[q * t.ngroups + g]);
    v = __fsub_rn(ord2f(m), filter_eps(t, q));
  } else {
    v = t.prev_scores[q * t.k + t.k - 1];
  }
  theta[q] = init ? v : fmaxf(theta[q], v);
}

// After the bound pass: the G >= k part maxima of a query, each minus eps(q), are lower bounds of the exact scores of G
// DISTINCT keys (one per part).  The k-th largest of them is therefore a lower bound of the final k-th best score:
// theta.  With G = 4 k parts it is worth the exact k-th best of ~ 0.85 of the prefix (two of the sample's best k keys
// share a part k^2 / 2G ~ 1.2 times on average); with G = k parts (round 1: the minimum of k maxima) only of
// prefix / (ln k + 1).  Sharded banks: the k largest, descending, also go to scores[B,k] and travel through the same
// exchange as a level's exact scores (the k-th largest of the union of all shards' values bounds the global k-th best).
__global__ void __launch_bounds__(256) filter_bound_scores_kernel(FilterThr t, int64_t B, float* __restrict__ scores,
                                                                  float* __restrict__ theta) {
  // one wave per query: lane l holds parts l and l + 64 (G <= 128) and ranks them by counting (ties by part index)
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= B) return;
  const float eps = filter_eps(t, q);
  const int G = t.ngroups;
  float v[2];
  int rank[2] = {0, 0};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int g = lane + 64 * u;
    v[u] = g < G ? __fsub_rn(ord2f(t.gmax[q * G + g]), eps) : RG_NEG_INF;
  }
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    if (64 * w >= G) break;  // (wave-uniform)
    const int on = G - 64 * w < 64 ? G - 64 * w : 64;
    for (int o = 0; o < on; ++o) {
      const float x = __shfl(v[w], o);
      const int xi = o + 64 * w;
#pragma unroll
      for (int u = 0; u < 2; ++u) rank[u] += (x > v[u] || (x == v[u] && xi < lane + 64 * u)) ? 1 : 0;
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (lane + 64 * u < G && rank[u] < t.k) {
      if (scores) scores[q * t.k + rank[u]] = v[u];
      if (rank[u] == t.k - 1) theta[q] = v[u];
    }
  }
}

// Sharded banks, after a level: theta[b] = max(theta[b], k-th largest of the union of every shard's best m exact scores
// of query b) -- the k-th largest of a SUBSET of all scores is a lower bound of the k-th largest of all.  `gathered` is the
// all_gather's [G, B, m] layout as it stands; G m <= 64: one wave per query, lane l holds one score and ranks it by
// counting (ties broken by lane, so duplicates count as many times as they occur).
__global__ void __launch_bounds__(256) theta_sharpen_kernel(const float* __restrict__ gathered, int G, int64_t B, int m, int k,
                                                            float* __restrict__ theta) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int n = G * m;
  float v = RG_NEG_INF;
  if (lane < n) v = gathered[((int64_t)(lane / m) * B + b) * m + (lane % m)];
  int rank = 0;
  for (int o = 0; o < n; ++o) {
    const float u = __shfl(v, o);
    rank += (u > v || (u == v && o < lane)) ? 1 : 0;
  }
  if (lane < n && rank == k - 1) theta[b] = fmaxf(theta[b], v);
}

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
        } else if constexpr (FOLD) {  // one chain of maxima over every group's I - T, one sign test
          int mall = as_bits(a[0][0][0]);
#pragma unroll
          for (int gq = 0; gq < NG; ++gq)
#pragma unroll
            for (int e = (gq == 0 ? 1 : 0); e < 8; ++e) mall = max(mall, as_bits(a[e >> 2][gq][e & 3]));
          hit = mall >= 0;
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
          if constexpr (FOLD) {  // the groups' own maxima, only now; scored entries carry I itself: + T
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) {
              mi[gq] = as_bits(a[0][gq][0]);
#pragma unroll
              for (int r = 1; r < 4; ++r) mi[gq] = max(mi[gq], as_bits(a[0][gq][r]));
#pragma unroll
              for (int r = 0; r < 4; ++r) mi[gq] = max(mi[gq], as_bits(a[1][gq][r]));
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

// A call that filtered with a SPECULATIVE first bound (ragraph_topk_cosine_filtered_set_prior: theta = the prior for every
// query, no bound pass) is exact for a query iff its final k-th best candidate scores at least the prior: a level filtered
// with theta_l = max(prior, the running exact k-th best) <= the final k-th best, so every key that scores at least the
// final k-th best passed its level.  A query whose k-th best is below the prior (or that found fewer than k candidates) is
// listed for the exact scan of the fixup launch behind this one, like a query whose list overflowed.  The same pass
// records the smallest / largest final k-th best of the call (stats[18] / [19]): the next call's prior comes from them.
// (Zero queries -- flag 2 -- are answered without a scan and not judged; queries already listed by the final level
// -- flag 1 / an overflowed list -- carry -inf or stale rows: listed twice would be scanned twice, so they are skipped by
// their flag.)
__global__ void __launch_bounds__(256) filter_verify_prior_kernel(const float* __restrict__ out_s, int64_t B, int k, float prior,
                                                                  int speculative, const unsigned char* __restrict__ flag,
                                                                  int* __restrict__ overflow, int* __restrict__ overflow_list,
                                                                  int* __restrict__ stats) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int lo = INT_MAX, hi = INT_MIN, failed = 0;
  if (q < B && flag[q] != 2) {
    const float kth = out_s[q * k + k - 1];
    if (speculative && flag[q] == 0 && !(kth >= prior)) {
      overflow_list[atomicAdd(overflow, 1)] = (int)q;
      failed = 1;
    } else if (kth > RG_NEG_INF) {
      lo = hi = f2ord(kth);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off));
    hi = max(hi, __shfl_xor(hi, off));
    failed += __shfl_xor(failed, off);
  }
  // one set of atomics per WORKGROUP (a returning atomic on one address costs ~11 ns chip-wide: 1600 waves of a 100 000-query
  // call at three each were 38 us of a launch that reads 400 KB)
  __shared__ int red[3][4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][w] = lo;
    red[1][w] = hi;
    red[2][w] = failed;
  }
  __syncthreads();
  if (threadIdx.x == 0 && stats) {
    lo = min(min(red[0][0], red[0][1]), min(red[0][2], red[0][3]));
    hi = max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3]));
    failed = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    if (lo != INT_MAX) atomicMin(stats + 18, lo);
    if (hi != INT_MIN) atomicMax(stats + 19, hi);
    if (failed) atomicAdd(stats + 17, failed);
  }
}

// Sharded banks under a speculative first bound: the verdict of the rows' OWNER, behind the merge of the shards' lists.  One
// workgroup walks the R merged rows: a row is proven iff its k-th best reaches the prior (an all-zero query -- every score +0 --
// is answered by index order and needs no proof); out[0] = rows that missed, out[1] = -(smallest proven k-th best), out[2] =
// the largest, out[3] = this shard's candidates per query over its levels (the call's statistics words; -1 without them),
// out[4] = its overflowed lists: five numbers that ONE all_reduce MAX turns into the group's (ragraph_amd/sharded.py).
__global__ void __launch_bounds__(256) verify_merged_prior_kernel(const float* __restrict__ s, int64_t R, int k, float prior,
                                                                  int speculative, const int* __restrict__ words,
                                                                  const int* __restrict__ overflow, float* __restrict__ out) {
  __shared__ float red[3][4];
  float miss = 0.f, neg_lo = RG_NEG_INF, hi = RG_NEG_INF;
  for (int64_t q = threadIdx.x; q < R; q += 256) {
    const float top = s[q * k], kth = s[q * k + k - 1];
    const bool zero = top == 0.f && kth == 0.f;
    const bool ok = zero || !speculative || kth >= prior;
    if (!ok) miss += 1.f;
    else if (!zero && kth > RG_NEG_INF) {
      neg_lo = fmaxf(neg_lo, -kth);
      hi = fmaxf(hi, kth);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    miss += __shfl_xor(miss, off);
    neg_lo = fmaxf(neg_lo, __shfl_xor(neg_lo, off));
    hi = fmaxf(hi, __shfl_xor(hi, off));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][w] = miss;
    red[1][w] = neg_lo;
    red[2][w] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    out[1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    out[2] = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
    float cand = -1.f;
    if (words && words[0] == FILTER_STATS_MAGIC) {
      cand = 0.f;
      for (int l = 0; l < 3; ++l)
        if (words[5 + l] > 0) cand += (float)words[2 + l] / (float)words[5 + l];
    }
    out[3] = cand;
    out[4] = overflow ? (float)*overflow : 0.f;
  }
}

// Large batches (the one-wave-per-query rescoring kernels): the final level has listed the overflowed queries, and this
// launch -- a fixed grid that finds an empty list on ordinary banks and returns -- runs exact_scan_query for each.
// (Below 2048 queries the workgroup-per-query rescoring kernels call it themselves and this launch is not made.)
// The exact scan of the queries the final level could not serve, in ONE launch behind it (calls of 65 queries and more;
// smaller ones scan inside their rescoring launch).  Few overflowed queries -- the usual case when there are any: a
// tight cluster next to a handful of queries -- would leave the chip idle behind one workgroup per query (25 ms per
// scan of 1M x 256 keys; 94 ms when the query's own rescoring wave did it), so a query's scan is cut into up to
// FILTER_FIX_SLICES key slices (as many as keep ~256 workgroups busy), each workgroup leaves its slice's k winners in
// part_s / part_i, and the query's last slice to finish (a ticket) merges them: 1.6 ms for one query.  Many overflowed
// queries take one workgroup each as before.  A ZERO query is answered without a scan.
template <int D>
__global__ void __launch_bounds__(256) topk_overflow_fixup_kernel(const float* __restrict__ Qn, const float* __restrict__ Kn,
                                                                  int64_t N, int k, int64_t idx_base,
                                                                  const int* __restrict__ overflow,
                                                                  const int* __restrict__ overflow_list,
                                                                  int64_t* __restrict__ overflow_idx_out,
                                                                  float* __restrict__ out_s, int64_t* __restrict__ out_i,
                                                                  int* __restrict__ done, float* __restrict__ part_s,
                                                                  int64_t* __restrict__ part_i, int64_t B,
                                                                  const unsigned char* __restrict__ flag,
                                                                  int* __restrict__ stats) {
  __shared__ float4 qs[D / 4];
  __shared__ __attribute__((aligned(16))) float tile[4][64 * RESCORE_LD];
  __shared__ float ps[4][32];
  __shared__ int64_t pi[4][32];
  __shared__ int ticket_s;
  const int n_over = *overflow;
  if (stats && blockIdx.x == 0 && threadIdx.x == 0) stats[20] = n_over;   // (final: every launch that counts runs before this one)
  if (stats && stats[16] == 0) {
    // the smallest / largest final k-th best score of the call's queries (stats[18] / [19]; a speculative call's verify
    // launch has recorded them already): one value per thread, wave-reduced, two atomics per wave that saw any.
    int lo = INT_MAX, hi = INT_MIN;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < B; q += (int64_t)gridDim.x * 256) {
      // (zero queries -- flag 2 -- have no k-th best; rows flagged 1 are listed for the scans below and hold what torch.empty
      // left or a stale candidate row that the scans rewrite during this very launch: neither may reach the history words)
      const float kth = flag[q] != 0 ? RG_NEG_INF : out_s[q * k + k - 1];
      if (kth > RG_NEG_INF) {
        lo = min(lo, f2ord(kth));
        hi = max(hi, f2ord(kth));
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      lo = min(lo, __shfl_xor(lo, off));
      hi = max(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
      if (lo != INT_MAX) atomicMin(stats + 18, lo);
      if (hi != INT_MIN) atomicMax(stats + 19, hi);
    }
  }
  if (n_over <= 0) return;
  int SL = 1;
  if (n_over <= FILTER_FIX_MAX_Q)
    while (SL < FILTER_FIX_SLICES && 2 * SL * n_over <= (int)gridDim.x) SL *= 2;
  const int64_t chunk = ((N + SL - 1) / SL + 63) / 64 * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t item = blockIdx.x; item < (int64_t)n_over * SL; item += gridDim.x) {
    const int o = (int)(item / SL), sl = (int)(item % SL);
    const int64_t b = overflow_list[o];
    if (overflow_idx_out && threadIdx.x == 0 && sl == 0) overflow_idx_out[o] = b;
    __syncthreads();  // (the previous item's readers of qs)
    if (threadIdx.x < D / 4) qs[threadIdx.x] = reinterpret_cast<const float4*>(Qn + b * D)[threadIdx.x];
    __syncthreads();
    if (SL == 1) {
      exact_scan_query<D>(qs, Kn, N, k, idx_base, tile, ps, pi, out_s + b * k, out_i + b * k);
      continue;
    }
    const int64_t lo = sl * chunk, hi = lo + chunk < N ? lo + chunk : N;
    float* my_s = part_s + ((int64_t)o * SL + sl) * 32;
    int64_t* my_i = part_i + ((int64_t)o * SL + sl) * 32;
    if (lo < hi) {
      exact_scan_query<D>(qs, Kn + lo * D, hi - lo, k, lo, tile, ps, pi, my_s, my_i);   // (indices local to the bank)
    } else if (threadIdx.x < k) {
      my_s[threadIdx.x] = RG_NEG_INF;
      my_i[threadIdx.x] = INT64_MAX;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) ticket_s = atomicAdd(done + o, 1);
    __syncthreads();
    if (ticket_s != SL - 1) continue;  // (workgroup-uniform)
    __threadfence();
    if (w == 0) {  // the query's last slice: SL k <= 512 partial winners, eight slots per lane
      float s8[8];
      int id8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = lane + 64 * u;
        s8[u] = RG_NEG_INF;
        id8[u] = INT_MAX;
        if (e < SL * k) {
          const int64_t at = ((int64_t)o * SL + e / k) * 32 + e % k;
          const int64_t pv = __builtin_nontemporal_load(part_i + at);
          if (pv < INT_MAX) {
            s8[u] = __builtin_nontemporal_load(part_s + at);
            id8[u] = (int)pv;
          }
        }
      }
      wave_select<8>(s8, id8, k, lane, idx_base, out_s + b * k, out_i + b * k);
    }
  }
}

static bool rescore_coop() {  // RAGRAPH_RESCORE_COOP=0: every lane reads its own row (A/B)
  static const bool on = [] {
    const char* e = getenv("RAGRAPH_RESCORE_COOP");
    return !(e && atoi(e) == 0);
  }();
  return on;
}

// Scored candidate lists for the int8 levels (topk_rescore_scored_kernel): calls whose rescoring is bound by the row
// gathers, i.e. the ones that take the one-wave-per-query kernels.  RAGRAPH_FILTER_SCORED=0/1: A/B.
// D = 256 only: measured with / without (ms per call, profiles/r3_scored_ab.txt) 2048 x 1M x 256: 0.874 / 0.836, 16384:
// 4.50 / 4.21, 100 000: 24.4 / 22.8; but 50 000 x 2M x 128: 12.71 / 12.63 and 65 536 x 4M x 64: 15.73 / 15.93 -- shorter rows
// are cheaper to fetch and their scores spread wider against the same eps (fewer extra candidates to prune), so the
// second round only adds latency.
// k <= 16: round 1 is 16 rows and lists beyond 256 entries take the plain path -- 50 000 x 1M x 256 at k = 16: 14.5 / 13.7 ms,
// k = 20: 15.6 / 15.8, k = 32: 18.7 / 20.3.
static bool filter_scored_lists(int64_t B, int D, int k) {
  const char* env_s = getenv("RAGRAPH_FILTER_SCORED");  // (read per call: the tests switch it)
  const int env = env_s ? atoi(env_s) : -1;
  // every call of the ring kernel (> 256 queries) ...: a scored list needs so few rows that ONE wave per query beats the
  // four-wave workgroups of the wide kernels even at a few hundred queries, whose single level admits ~380 candidates per
  // query and prunes 90 % of them (257 x 1M x 256: 0.214 -> 0.189 ms, 512: 0.267 -> 0.228, 1024: 0.436 -> 0.377, 1536: 0.580 ->
  // 0.490; RAGRAPH_FILTER_SCORED_MIN_B: A/B)
  // ... and the direct kernel's calls of 65 - 256 queries (entries carry ceil(I / 256)): 128 x 1M: 0.111 -> 0.106 ms, 256:
  // 0.148 -> 0.136.  Up to 64 queries the direct kernel keeps several sub-lists per query and several workgroups rescore
  // each: one wave per query measured slower there (one query 0.075 -> 0.080 ms).
  static const int64_t min_b = [] { const char* e = getenv("RAGRAPH_FILTER_SCORED_MIN_B"); return e ? (int64_t)atoll(e) : (int64_t)65; }();
  if (B < (min_b > 65 ? min_b : 65) || !rescore_coop()) return false;
  if (env >= 0) return env != 0;
  return D == 256 && k <= 16;
}

static bool filter_wide_waves(int64_t B) {  // RAGRAPH_FILTER_QW128=0/1: A/B; default from 1024 queries (one full tile)
  static const int env = [] {
    const char* e = getenv("RAGRAPH_FILTER_QW128");
    return e ? atoi(e) : -1;
  }();
  return env < 0 ? B >= 1024 : env != 0;
}

static int filter_device_cus() { return device_cus_multiple_of_8(); }  // per device (common.h)

}  // namespace ragraph

using namespace ragraph;

// Optional timing of the filter kernel alone (bench.py's roofline): events recorded around its launches on the caller's
// stream, into a CALLER-OWNED object attached to the calling thread (no process-global state: the attachment is
// thread-local, like the int8 cap).
struct ragraph_filter_profile {
  hipEvent_t ev[2 * 4];  // three filter levels + the bound pass (slot 3)
  int have, bound;
  int i8[4];
  int64_t keys[4];
};
static thread_local ragraph_filter_profile* t_prof = nullptr;
extern "C" ragraph_filter_profile* ragraph_filter_profile_create(void) {
  ragraph_filter_profile* p = new (std::nothrow) ragraph_filter_profile();
  if (!p) {
    set_error("profile: out of memory");
    return nullptr;
  }
  for (int i = 0; i < 2 * 4; ++i)
    if (hipEventCreate(&p->ev[i]) != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(p->ev[j]);
      delete p;
      set_error("profile: cannot create events");
      return nullptr;
    }
  return p;
}
extern "C" void ragraph_filter_profile_destroy(ragraph_filter_profile* p) {
  if (!p) return;
  if (t_prof == p) t_prof = nullptr;
  for (int i = 0; i < 2 * 4; ++i) (void)hipEventDestroy(p->ev[i]);
  delete p;
}
extern "C" ragraph_filter_profile* ragraph_filter_profile_attach(ragraph_filter_profile* p) {
  ragraph_filter_profile* old = t_prof;
  t_prof = p;
  if (p) p->have = p->bound = 0;
  return old;
}
// Milliseconds the filter kernel ran in the most recent call recorded into `p` (its launches summed; synchronises with
// them), or a negative number if none was timed.
extern "C" float ragraph_filter_profile_last_ms(ragraph_filter_profile* p) {
  if (!p || !p->have) return -1.f;
  float total = 0.f;
  if (p->bound) {
    float ms = 0.f;
    if (hipEventSynchronize(p->ev[7]) != hipSuccess || hipEventElapsedTime(&ms, p->ev[6], p->ev[7]) != hipSuccess) return -1.f;
    total += ms;
  }
  for (int l = 0; l < p->have; ++l) {
    float ms = 0.f;
    if (hipEventSynchronize(p->ev[2 * l + 1]) != hipSuccess || hipEventElapsedTime(&ms, p->ev[2 * l], p->ev[2 * l + 1]) != hipSuccess)
      return -1.f;
    total += ms;
  }
  return total;
}

// Per launch of the most recent call: slot 0..2 = the filter levels, slot 3 = the bound pass.  ms_host[s] (negative: no such
// launch), i8_host[s] = 1 if the level ran on the int8 copy, keys_host[s] = keys it covered.  Host arrays of 4 entries.
extern "C" int ragraph_filter_profile_levels(ragraph_filter_profile* p, float* ms_host, int* i8_host, int64_t* keys_host) {
  RG_REQUIRE(p && ms_host && i8_host && keys_host, RAGRAPH_EINVAL, "profile: null pointer");
  for (int s_ = 0; s_ < 4; ++s_) {
    ms_host[s_] = -1.f;
    i8_host[s_] = p->i8[s_];
    keys_host[s_] = p->keys[s_];
    const bool have = s_ == 3 ? p->bound != 0 : s_ < p->have;
    if (!have) continue;
    float ms = 0.f;
    if (hipEventSynchronize(p->ev[2 * s_ + 1]) == hipSuccess && hipEventElapsedTime(&ms, p->ev[2 * s_], p->ev[2 * s_ + 1]) == hipSuccess)
      ms_host[s_] = ms;
  }
  return RAGRAPH_OK;
}

// Schedule of a call: exact fp32 top-k over the first n0 keys (its k-th score is the first bound), then bf16 filter +
// exact rescoring over [0, e1), [e1, e2), ... [.., N).  A level's k-th exact score is the next level's bound, so a level
// lets through ~1.3 k (its end / the previous end) keys per query.
//   * Large batches (the bench's 100 k queries): n0 = N/256, ends N/32, N/4, N -- ~100, ~100 and ~40 candidates per
//     query; level 0 is the fp32 tile kernel.  The matrix work dominates, three levels keep the rescoring at ~8 %.
//   * Small and medium batches (B <= 16384): a level costs ~60 us whatever it filters (launches, ring prologue, the
//     rescoring kernel's latency) while candidates are cheap, so fewer, steeper levels win; and level 0 is a slab --
//     the dense kernel (same fmaf chains as everything else) writes the B x n0 scores, topk_rows selects -- which
//     spreads over the whole chip where the tile kernel would run one query tile on a few CUs.  n0 and the number of
//     levels minimise   slab(B, n0) + L (60 us + B * 1.3 k r * 0.4 ns),  r = (N / n0)^(1/L),  under 1.3 k r <= cap / 2.
constexpr int FILTER_MAX_LEVELS = 3;
static int64_t filter_round_up(int64_t n) { return (n + FILTER_PAD_KEYS - 1) / FILTER_PAD_KEYS * FILTER_PAD_KEYS; }

struct FilterSchedule {
  int64_t bound_keys;               // > 0: no exact level 0 -- the first bound comes from a bf16 pass over keys [0, bound_keys)
  int64_t n0;                       // level 0: exact top-k over keys [0, n0)   (bound_keys == 0)
  int slab0;                        // level 0 by dense kernel + topk_rows (needs B * n0 floats of workspace)
  int nlev;                         // filter levels
  int64_t ends[FILTER_MAX_LEVELS];  // their ends (multiples of 256 except the last = N)
  int i8_levels;                    // the last i8_levels levels run on the int8 copy (filter_common.h)
};

constexpr int64_t FILTER_SLAB_MAX_B = 16384;
// up to this many queries the prepare launch also leaves the queries as bf16 (and int8) B operands in fragment order: the direct
// kernel's image (<= 256), and the ring kernel's operand load -- 32 independent 16-byte loads per lane instead of eight
// dependent batches of fp32 loads + conversions (13 us per segment at D = 256), which short launches cannot amortise
// (every filtered call: KeyIndex cuts batches at 262 144 queries.  Large batches have long segments on ONE GPU -- the images
// save ~0.7 % of the bench step -- but the short launches of a key-sharded rank do not: the bound launch of one rank of 8
// spent a quarter of its 0.27 ms converting operands.  The images are 3 D bytes per query: 77 MB at 100 000 queries.)
constexpr int64_t FILTER_QB_MAX_B = 262144;
constexpr int64_t FILTER_SLAB_MAX_SCORES = (int64_t)1 << 26;  // 256 MiB of scores

// Banks of >= 8192 keys (KeyIndex sends >= 16384) take their first bound from the BOUND pass instead of an
// exact level 0: the filter kernel itself runs over the first bound_keys keys and records, per query, the best approximate
// score of each of k consecutive parts; the smallest of the k maxima, minus eps, bounds the final k-th best from below
// (filter_prepare_kernel).  As a bound it is worth the exact k-th best of ~bound_keys / (ln k + 1) keys, and it costs a
// bf16 pass with no lists, no inserts and no fp32 matrix work: 0.7 ms instead of the tile kernel's 3.2 ms for the
// bench's 100 k queries, 40 us instead of the slab's 110 us for 256.  RAGRAPH_FILTER_EXACT_LEVEL0=1 keeps the exact
// level 0 (A/B).
static bool filter_bound_pass_enabled() {
  static const bool on = [] {
    const char* e = getenv("RAGRAPH_FILTER_EXACT_LEVEL0");
    return !(e && atoi(e) != 0);
  }();
  return on;
}

// Parts of the bound pass's prefix: 4 k (at most 128), as many as the prefix has stages (a part is at least one ring stage;
// sub-tiles of the direct kernel are finer), never fewer than k.
static int filter_bound_parts(int k, int64_t bound_keys, int D, int64_t B = 1 << 20, int n_shards = 1) {
  if (B <= 64) return k;  // a handful of queries: the minimum of k part maxima, taken inside the filter launch's prologue
                          // (filter_threshold) -- the extra selection launch would cost more than the shorter prefix saves
  int64_t g = 4 * (int64_t)k;
  if (g > 128) g = 128;
  if (n_shards > 1) g = (g + n_shards - 1) / n_shards;  // (pooled through the exchange: 4 k parts over all shards)
  const int64_t avail = bound_keys == INT64_MAX ? g : bound_keys / (FILTER_STAGE_BYTES / (2 * D));
  if (g > avail) g = avail;
  return (int)(g < k ? k : g);
}
// prefix keys per key of exact sample the bound is worth (see filter_bound_scores_kernel)
static double filter_bound_eff(int k, int parts) {
  if (parts >= 4 * k || parts >= 128) return 1.2;
  if (parts >= 2 * k) return 1.5;
  return log((double)k) + 1.0;
}

// D = 64: a stage of the int8 copy holds 512 keys (32 KB / 64 B) and a level starts at a whole stage, so the inner level ends
// are multiples of 512 -- a level that started at an odd multiple of 256 would begin with the previous level's last 256
// keys again, and a key listed twice breaks the selection (distinct pairs are what its ranks count).
static void filter_align_ends(FilterSchedule& sc, int D) {
  if (D != 64) return;
  for (int l = 0; l + 1 < sc.nlev; ++l) {
    const int64_t e = sc.ends[l] / 512 * 512;
    if (e >= 512 && (l == 0 || e > sc.ends[l - 1])) sc.ends[l] = e;
  }
  if (sc.bound_keys > sc.ends[0]) sc.bound_keys = sc.ends[0] / FILTER_PAD_KEYS * FILTER_PAD_KEYS;
}

// n_shards > 1 (row-sharded bank, N = the largest shard): the shards pool their first samples through the exchange, so
// the sample is planned for the WHOLE bank and every shard scans its share of the prefix.
static int rescore_slices(int64_t B, int k);
// Sharded banks of up to this many shards keep the SCORED lists on their int8 levels (and the schedule that goes with them):
// a shard's own round-1 bound comes from 1 / G of the keys while the level's threshold was pooled over all shards' earlier
// levels -- at G = 2 the shard's bound is still the sharper one (half of the bank against a quarter), from G = 4 it is not and
// the second round only adds latency (profiles/r3_emul.txt).  RAGRAPH_FILTER_SCORED_SHARDS: A/B.
static int filter_scored_shards() {
  static const int v = [] { const char* e = getenv("RAGRAPH_FILTER_SCORED_SHARDS"); return e ? atoi(e) : 2; }();
  return v;
}
static FilterSchedule filter_schedule(int64_t B, int64_t N, int D, int k, int n_shards = 1) {
  FilterSchedule sc{};
  const int cap = 2048;
  // scored lists (one bank, >= 2048 queries): an int8 level's rescoring fetches about a third of its candidates' rows, which
  // makes int8 pay on EVERY level (the bench step, 2 / 3 int8 levels: 24.3 / 23.85 ms; without the scores 26.9 / 27.7)
  const bool scored = (n_shards == 1 || n_shards <= filter_scored_shards()) && B >= 2048 && filter_scored_lists(B, D, k);  // (below 2048 queries the plain lists' plans
                                                                                   // stay: a smaller first sample measured slower)
  // (the model's price of an int8 candidate under scored lists, relative to the plain lists'; fitted: 0.6 moves 8192+ queries
  // x 1M keys from two levels to three, all int8 -- 8192: 2.37 -> 2.33 ms, 16384: 4.30 -> 4.13 -- while 0.45 also shrank the
  // first sample of 2048 - 8192 queries, which measured 2 - 4 % slower; RAGRAPH_FILTER_SCORED_CAND: A/B)
  static const double scored_cand = [] { const char* e = getenv("RAGRAPH_FILTER_SCORED_CAND"); return e ? atof(e) : 0.6; }();
  const bool bound = N >= 8192 && filter_bound_pass_enabled();
  // int8 levels (filter_common.h): D = 128 / 256, the ring kernel's batch sizes, banks long enough to be matrix-bound (an
  // int8 level quantises its queries from the fp32 rows per segment where the bf16 levels of up to 16384 queries load a
  // prepared image -- Cora-sized 2708 x 10 000 x 128: 0.087 -> 0.100 ms)
  static const bool i8_d64 = [] { const char* e = getenv("RAGRAPH_FILTER_I8_D64"); return !e || atoi(e) != 0; }();  // A/B
  // (D = 64, the edge flavour: one MFMA per 16-key half and query group, so the epilogue weighs more -- 65 536 x 4M x 64:
  // 22.5 -> 15.5 ms with eight groups per wave; eps is the same 0.02 but the scores' spread is 1/8: fewer extra candidates)
  // (with the prepared int8 operand image and the scored lists, D = 256 also pays on banks of 32 768+ keys from 2048 queries:
  // 4096 x 40 000: 0.189 -> 0.160 ms, 2100 x 60 000: 0.180 -> 0.150, 16 384 x 50 000: 0.64 -> 0.49; not at D = 128 -- 8192 x
  // 50 000: 0.237 -> 0.244 -- nor on shorter banks -- 8192 x 20 000 x 256: 0.221 -> 0.238)
  const bool i8_ok = (D == 128 || D == 256 || (D == 64 && i8_d64)) && B > 256 &&
                     (N * n_shards >= 65536 || (D == 256 && B >= 2048 && N * n_shards >= 32768));
  const bool mid_i8 = i8_ok && N * n_shards < 65536;
  static const bool i8_direct_env = [] { const char* e = getenv("RAGRAPH_FILTER_I8_DIRECT"); return !e || atoi(e) != 0; }();  // A/B
  // (D = 64, round 5: the edge flavour's calls of up to 256 queries -- half the stream, the scores' spread 1/8 against the
  // same eps; RAGRAPH_FILTER_I8_DIRECT_D64=0: A/B)
  static const bool i8_direct_d64 = [] { const char* e = getenv("RAGRAPH_FILTER_I8_DIRECT_D64"); return !e || atoi(e) != 0; }();
  const bool i8_direct = (D == 128 || D == 256 || (D == 64 && i8_direct_d64)) && B <= 256 && N * n_shards >= 65536 && i8_direct_env;
  // bound_keys / eff_div ~ the exact sample the bound is worth: planned for 4 k parts, corrected below if the prefix is
  // too short for that many
  const double eff_div = filter_bound_eff(k, B <= 64 ? k : 4 * k);
  auto prefix_for = [&](int64_t n0) {  // prefix whose bound is worth the exact k-th best of n0 keys
    int64_t nA = filter_round_up((int64_t)((double)n0 * eff_div));
    const int parts = filter_bound_parts(k, nA, D, B);
    if (parts < 4 * k && parts < 128) nA = filter_round_up((int64_t)((double)n0 * filter_bound_eff(k, parts)));
    return nA;
  };
  if (B > FILTER_SLAB_MAX_B || N < 4 * 4096) {
    // (the first sample: with 4 k parts the bound pass is cheap enough for N / 64 -- fewer candidates at the first level,
    // whose sub-tiles otherwise nearly all take the candidate path: 100k x 1M: 38.0 -> 36.5 ms against N / 256;
    // RAGRAPH_FILTER_N0DIV: A/B)
    static const int64_t n0div = [] { const char* e = getenv("RAGRAPH_FILTER_N0DIV"); return e ? (int64_t)atoll(e) : (int64_t)64; }();
    int64_t n0 = N * n_shards / n0div;  // (over all shards)
    if (n0 < 4096) n0 = 4096;
    int64_t nA = bound ? prefix_for(n0) : 0;
    if (n_shards > 1) {  // this shard's share
      n0 /= n_shards;
      nA = filter_round_up(nA / n_shards);
      const int64_t min_keys = filter_round_up((int64_t)k * (FILTER_STAGE_BYTES / (2 * D)));
      if (nA < min_keys) nA = min_keys;
    }
    if (n0 > N) n0 = N;
    if (n0 < k) n0 = k < N ? k : N;
    if (bound) {
      if (nA > N / 4) nA = N / 4 / FILTER_PAD_KEYS * FILTER_PAD_KEYS;
      sc.bound_keys = nA;
    }
    sc.n0 = n0;
    sc.slab0 = 0;  // (measured at 100 k queries: slabs of 16384 cost 3.6 ms -- dense kernel 104 TFLOP/s, topk_rows bound
                   // by its list inserts -- against the tile kernel's 3.2 ms)
    sc.nlev = 0;
    int64_t prev = n0;
    // (RAGRAPH_FILTER_FRACS="a,b": the first ends as fractions N/a, N/b of the bank -- schedule experiments)
    int64_t fracs[2] = {32, 4};
    int nfr = 2;
    if (const char* fe = getenv("RAGRAPH_FILTER_FRACS")) {
      long long a = 0, b = 0;
      nfr = sscanf(fe, "%lld,%lld", &a, &b);
      if (nfr < 1 || a < 2) nfr = 0;
      fracs[0] = a;
      fracs[1] = b;
      if (nfr == 2 && b < 2) nfr = 1;
    }
    for (int fi = 0; fi < nfr; ++fi) {
      const int64_t frac = fracs[fi];
      int64_t e = filter_round_up(N / frac);
      if (e < 4 * prev) e = filter_round_up(4 * prev);  // a level is at least 4x what came before
      if (e * 2 >= N) break;                            // too close to the end: the last level takes the rest
      sc.ends[sc.nlev++] = e;
      prev = e;
    }
    sc.ends[sc.nlev++] = N;
    // the k keys behind the bound must lie inside the first level (it has to find at least k candidates)
    if (sc.bound_keys > sc.ends[0]) sc.bound_keys = sc.ends[0] / FILTER_PAD_KEYS * FILTER_PAD_KEYS;
    if (sc.bound_keys / (FILTER_STAGE_BYTES / (2 * D)) < k) sc.bound_keys = 0;  // every part needs a stage of its own
    sc.i8_levels = i8_ok && B >= 1024 ? (scored ? 3 : 2) : 0;  // (banks below 4 x 4096 keys come here with any batch)
    filter_align_ends(sc, D);
    return sc;
  }
  double best = 1e30;
  int64_t best_n0 = 4096, best_nA = 0;
  int best_L = FILTER_MAX_LEVELS, best_i8 = 0;
  // RAGRAPH_FILTER_FORCE_N0 / _L: schedule experiments (n0 = the exact sample the first bound is worth, L levels)
  static const int64_t force_n0 = [] { const char* e = getenv("RAGRAPH_FILTER_FORCE_N0"); return e ? (int64_t)atoll(e) : (int64_t)0; }();
  static const int force_L = [] { const char* e = getenv("RAGRAPH_FILTER_FORCE_L"); return e ? atoi(e) : 0; }();
  const int stage_keys = FILTER_STAGE_BYTES / (2 * D);
  const double tiles = (double)((B + 511) / 512);
  for (int64_t n0 = 4096; n0 * 4 <= N; n0 *= 2) {
    if (force_n0 > 0 && n0 != force_n0) continue;
    double first;  // cost of the first bound, us
    int64_t nA = 0;
    if (bound) {
      nA = prefix_for(n0);
      if (nA * 4 > N) break;
      if (B <= 256)  // direct kernel: the prefix streams at ~5 TB/s (8.7 / 22 / 40 us for 54 k / 216 k / 216 k keys x 1 / 16 / 256 queries)
        first = 6.0 + (double)nA * 2.0 * D / 5.0e6 * (1.0 + (double)B / 320.0);
      else
        first = 35.0 + (double)nA * 2.0 * D / 3.0e6 + tiles * (double)(nA / stage_keys) * 3.1 / 256.0;
    } else {
      if (B * n0 > FILTER_SLAB_MAX_SCORES) break;
      first = 30.0 + (double)B * (double)n0 * (2.0 * D / 1.0e8 + 4.0 / 3.0e6);
    }
    for (int L = 1; L <= FILTER_MAX_LEVELS; ++L) {
      if (force_L > 0 && L != force_L) continue;
      const double r = pow((double)N / (double)n0, 1.0 / L);
      if (1.3 * k * r > cap / 2 && !(force_n0 > 0 && force_L > 0)) continue;
      // a level: launches + the rescoring kernels' latency floor, plus ~0.4 - 0.5 ns per candidate (1 KB row gather each)
      if (B <= 256) {
        // Direct kernel.  On the int8 copy its pass streams half the bytes and does half the matrix work (one query: 78 ->
        // 40 us of kernel; 256: 124 -> ~75) while ~3x the candidates come back: the same model with 3.9 k r candidates per
        // level and that saving decides between the two -- and moves n0 up when int8 wins.
        // (int8 wins at every batch size from 1 to 256 on the 1M x 256 bank -- 0.106 -> 0.074, 0.124 -> 0.091, 0.139 -> 0.109,
        // 0.192 -> 0.153 ms -- so where it is eligible the model only chooses ITS schedule; the two constants are not
        // comparable across the dtypes)
        for (int q8 = i8_direct ? 1 : 0; q8 <= (i8_direct ? 1 : 0); ++q8) {
          const double cands = 1.3 * k * r * (q8 ? 3.0 : 1.0);
          // (a handful of queries keep S sub-lists of `cap` slots each: filter_cap)
          if (cands > cap * (q8 ? rescore_slices(B, k) : 1) / 2 && !(force_n0 > 0 && force_L > 0)) continue;
          // (the sliced / wide rescoring of a small call is a latency chain: measured 1.2 - 3.7 ns per candidate on the int8
          // schedules -- forced n0 at 1 / 16 / 64 queries, profiles/r3_i8_ab.txt -- where round 2 fitted 0.5 to its bf16 ones)
          const double cost = first + L * (25.0 + (double)B * cands * (q8 ? 2.0e-3 : 0.5e-3)) + (L - 1) * (q8 ? 30.0 : 15.0)  /* (a second pass start-up; 256 queries, int8: one level 0.156, two 0.162 ms) */
                              - (q8 ? 32.0 + 0.08 * (double)B : 0.0);
          if (cost < best) {
            best = cost;
            best_n0 = n0;
            best_nA = nA;
            best_L = L;
            best_i8 = q8 ? L : 0;
          }
        }
        continue;
      }
      // Ring kernel: + the matrix work of each level -- 2 B keys D at ~1.25 PFLOP/s on the bf16 copy, ~2.4 Pop/s on the
      // int8 copy, whose ~5x wider bound passes ~3x the candidates -- for 0, 1 or 2 trailing int8
      // levels.  (Without int8 the matrix term is the same for every (n0, L): the choice among those is round 2's.)
      // Measured against forced schedules at 512 / 1024 / 2048 queries x 1M keys (profiles/r3_i8_ab.txt).
      // (a candidate costs ~0.4 ns while a level's rescoring is a latency chain -- up to ~1000 queries -- and ~0.18 ns once it
      // is bound by the row gathers: 100 000 queries x ~130 candidates x 1 KiB in 1.8 ms)
      const double per_cand = 0.18e-3 + 0.22e-3 * (B <= 1024 ? 1.0 : 1024.0 / (double)B);
      // (int8 candidates per bf16 candidate.  3.0 until the copy got its two scales and the calls their speculative bounds; 2.0
      // fits what tools/i8_rule_grid.py measures now -- 105 shapes of 300 .. 16 384 queries x 70 k .. 1 M keys x D = 64 / 128 /
      // 256, KeyIndex in its steady state, geomean 0.971 of the old rule's time; the banks of 150 k - 500 k keys that moved to
      // int8 0.77 - 0.9 (1100 x 300 k x 256: 0.188 -> 0.144 ms); 1.5 loses up to 1.6 x on 70 k-key banks.  profiles/r5_i8_rule_grid.txt)
      static const double i8_candf = [] { const char* e = getenv("RAGRAPH_FILTER_I8_CANDF"); return e ? atof(e) : 2.0; }();  // A/B
      // (mid_i8 -- D = 256 banks of 32 768 .. 65 535 keys: the constants below were fitted on million-key banks and overprice
      // these shapes' candidates; what measured faster there is the bf16 plan with every level moved to int8: see below)
      for (int i8 = 0; i8 <= (i8_ok && !mid_i8 ? (L < 2 || scored ? L : 2) : 0); ++i8) {
        double cost = first, e_prev = 0.0, e = (double)n0;
        bool fits = true;
        for (int l = 0; l < L; ++l) {
          e = l + 1 == L ? (double)N : e * r;
          const bool q8 = l >= L - i8;
          const double cands = 1.3 * k * r * (q8 ? i8_candf : 1.0);
          if (cands > cap / 2 && !(force_n0 > 0 && force_L > 0)) fits = false;
          cost += 60.0 + (e - e_prev) * (double)B * 2.0 * D / (q8 ? 2.4e9 : 1.25e9) +
                  (double)B * cands * per_cand * (q8 && scored ? scored_cand : 1.0);
          e_prev = e;
        }
        if (fits && cost < best) {
          best = cost;
          best_n0 = n0;
          best_nA = nA;
          best_L = L;
          best_i8 = i8;
        }
      }
    }
  }
  sc.bound_keys = bound ? (best_nA ? best_nA : prefix_for(4096)) : 0;
  if (bound && B > 128 && B <= 256 && n_shards == 1) {
    // the direct kernel deals the prefix's 16-KiB units over all waves of the chip in contiguous runs: 2.4 units per wave take
    // as long as 3 -- a prefix of whole rounds (8 waves x CUs units) costs what it reads: 256 queries x 1M 0.1406 -> 0.1381 ms,
    // 192: 0.1198 -> 0.1180 (up to 128 queries, whose pass is cheaper per key, the shorter prefix loses more than it saves:
    // 64 queries 0.110 -> 0.116).  RAGRAPH_FILTER_BOUND_ROUNDS=0: A/B
    static const int align_env = [] { const char* e = getenv("RAGRAPH_FILTER_BOUND_ROUNDS"); return e ? atoi(e) : 1; }();
    if (align_env) {
      const int64_t round_keys = (int64_t)8 * filter_device_cus() * (16384 / (2 * D));
      int64_t r = (sc.bound_keys + round_keys / 2) / round_keys;
      if (r < 1) r = 1;
      if (r * round_keys * 4 <= N && r * round_keys >= (int64_t)k * 4 * (FILTER_STAGE_BYTES / (2 * D))) sc.bound_keys = r * round_keys;
    }
  }
  sc.n0 = best_n0;
  sc.i8_levels = mid_i8 ? best_L : best_i8;
  sc.slab0 = 1;
  sc.nlev = 0;
  const double r = pow((double)N / (double)best_n0, 1.0 / best_L);
  double e = (double)best_n0;
  for (int l = 0; l + 1 < best_L; ++l) {
    e *= r;
    const int64_t ei = filter_round_up((int64_t)e);
    if (ei * 2 >= N) break;
    sc.ends[sc.nlev++] = ei;
  }
  sc.ends[sc.nlev++] = N;
  if (sc.bound_keys > sc.ends[0]) sc.bound_keys = sc.ends[0] / FILTER_PAD_KEYS * FILTER_PAD_KEYS;
  if (sc.bound_keys / stage_keys < k) sc.bound_keys = 0;  // every part needs a stage of its own: else the exact slab
  if (sc.bound_keys == 0 && B * sc.n0 > FILTER_SLAB_MAX_SCORES) sc.slab0 = 0;
  filter_align_ends(sc, D);
  return sc;
}

// Which levels run on the int8 copy: the LAST level of a large batch (D = 128 / 256).  Its threshold is the highest of the
// call, so the ~4x wider eps costs ~100 extra candidates per query (1 KiB row gathers: ~2.5 ms at the bench shape) where
// the matrix work of three quarters of the bank halves (25.6 -> ~13 ms).  Earlier levels and smaller batches stay on
// bf16: a level of a few thousand queries is not matrix-bound enough to pay for the extra rescoring.
// RAGRAPH_FILTER_I8 = n forces the last n levels (0: none) -- A/B runs and the tests of the int8 path on small shapes.
// A caller that knows its bank (ragraph_amd/kernels_index.py: the copy's measured error, or a call that overflowed) caps
// the int8 levels of ITS thread's following calls: -1 = the rule below, 0 = none.  Thread-local: no shared state.
static thread_local int t_max_i8_levels = -1;
// A SPECULATIVE first bound for the calling thread's following filtered calls (NaN = none, the default): the owner of a bank
// that has answered many queries knows where their k-th best scores lie (the statistics words of every call), and a call
// that starts from theta = prior for every query needs no bound pass -- filter_verify_prior_kernel proves each query's
// answer afterwards and sends the (rare) misses to the exact scan, so the result is exact whatever the prior is.
static thread_local float t_prior = __builtin_nanf("");
extern "C" float ragraph_topk_cosine_filtered_set_prior(float theta_prior) {
  const float old = t_prior;
  t_prior = theta_prior;
  return old;
}
extern "C" int ragraph_topk_cosine_filtered_max_i8_levels(int n) {
  const int old = t_max_i8_levels;
  t_max_i8_levels = n < 0 ? -1 : n;
  return old;
}
int ragraph::filter_thread_i8_cap() { return t_max_i8_levels; }  // (topk_small.hip: the single-launch call honours the same cap)
float ragraph::filter_thread_prior() { return t_prior; }
int ragraph::launch_overflow_fixup(int D, const float* Qn, const float* Kn, int64_t N, int k, int64_t idx_base, const int* count,
                                   const int* list, float* out_s, int64_t* out_i, int* done, float* part_s, int64_t* part_i,
                                   int64_t B, void* stream) {
  hipStream_t st = as_stream(stream);
#define RG_FIX(D_)                                                                                                          \
  hipLaunchKernelGGL(topk_overflow_fixup_kernel<D_>, dim3(256), dim3(256), 0, st, Qn, Kn, N, k, idx_base, count, list,       \
                     (int64_t*)nullptr, out_s, out_i, done, part_s, part_i, B, (const unsigned char*)nullptr, (int*)nullptr)
  if (D == 256) RG_FIX(256);
  else if (D == 128) RG_FIX(128);
  else RG_FIX(64);
#undef RG_FIX
  RG_CHECK_LAUNCH("overflow fixup");
  return RAGRAPH_OK;
}

static int filter_i8_levels(const FilterSchedule& sc, int64_t B, int D, int64_t N) {
  const char* env = getenv("RAGRAPH_FILTER_I8");  // (read per call: the tests switch it)
  const int force = env ? atoi(env) : -1;
  if (D != 64 && D != 128 && D != 256) return 0;
  if (B <= 256) {  // the direct kernel's int8 form: every level or none, as the schedule planned
    if (t_max_i8_levels == 0 || sc.i8_levels == 0) return 0;
    return sc.nlev;
  }
  if (force >= 0) return force < sc.nlev ? force : sc.nlev;
  if (t_max_i8_levels == 0) return 0;
  // The schedule plans them (filter_schedule: sc.i8_levels -- the level STRUCTURE never depends on the per-thread cap, so
  // the shards of a bank keep the same phases whatever each thinks of its rows).  Measured on the 1M x 256 bank (ms per
  // call, 0 / 1 / 2 int8 levels on round 2's schedules; profiles/r3_i8_ab.txt): 1024 queries 0.538 / 0.519 / 0.505; 2048:
  // 0.98 / 0.80 / 0.83; 4096: 1.75 / 1.34 / 1.28; 16384: 6.09 / 4.55 / 4.19; 100 000 (the bench step): 38.3 / 28.3 / 26.9
  // (three: 27.7); with the schedule chosen for int8 (two levels, the second on int8): 512: 0.314 -> 0.276, 1024: 0.509 -> 0.426.
  int n = sc.i8_levels < sc.nlev ? sc.i8_levels : sc.nlev;
  if (t_max_i8_levels > 0 && n > t_max_i8_levels) n = t_max_i8_levels;
  return n;
}

// workspace of level 0: the tile kernel's, or the score slab
static size_t filter_level0_ws(const FilterSchedule& sc, int64_t B, int D, int k) {
  if (sc.bound_keys > 0) return 0;
  return sc.slab0 ? align_up((size_t)(B < FILTER_SLAB_MAX_B ? B : FILTER_SLAB_MAX_B) * (size_t)sc.n0 * sizeof(float), 256)
                  : ragraph_topk_cosine_workspace_bytes(B, sc.n0, D, k);
}

static bool filter_dim_ok(int D) { return D == 64 || D == 128 || D == 256; }

extern "C" int ragraph_keys_to_bf16(const float* Kn, int64_t N, int D, uint16_t* Kb, void* stream) {
  RG_REQUIRE(Kn && Kb, RAGRAPH_EINVAL, "keys_to_bf16: null pointer");
  RG_REQUIRE(N >= 1, RAGRAPH_EINVAL, "keys_to_bf16: N=%lld must be >= 1", (long long)N);
  RG_REQUIRE(filter_dim_ok(D), RAGRAPH_EUNSUPPORTED, "keys_to_bf16: D=%d not in {64,128,256}", D);
  RG_REQUIRE(aligned16(Kn) && aligned16(Kb), RAGRAPH_EINVAL, "keys_to_bf16: pointers must be 16-B aligned");
  const int64_t npad = filter_round_up(N);
  unsigned* tail = reinterpret_cast<unsigned*>(Kb + npad * D);  // the extra row: max_k |dk|^2 as float bits
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(tail, 0, (size_t)D * sizeof(uint16_t), st) != hipSuccess) {
    set_error("keys_to_bf16: memset failed");
    return RAGRAPH_EDEVICE;
  }
  const dim3 grid((unsigned)cdiv(npad * (D / 8), 256));
  if (D == 256) hipLaunchKernelGGL(keys_to_bf16_kernel<256>, grid, dim3(256), 0, st, Kn, N, npad, Kb, tail);
  else if (D == 128) hipLaunchKernelGGL(keys_to_bf16_kernel<128>, grid, dim3(256), 0, st, Kn, N, npad, Kb, tail);
  else hipLaunchKernelGGL(keys_to_bf16_kernel<64>, grid, dim3(256), 0, st, Kn, N, npad, Kb, tail);
  RG_CHECK_LAUNCH("keys_to_bf16");
  // the int8 copy behind it (filter_common.h): the granules' largest |k_i| -> the cut and the two scales, then quantise + lay out
  const FilterI8View v8 = filter_i8_view(Kb, N, D);
  signed char* Kb8 = const_cast<signed char*>(v8.K8);
  unsigned* tail8 = const_cast<unsigned*>(v8.tail8);
  if (hipMemsetAsync(tail8, 0, (size_t)D * sizeof(uint16_t) + filter_i8_table_bytes(N, D), st) != hipSuccess) {
    set_error("keys_to_bf16: memset failed");
    return RAGRAPH_EDEVICE;
  }
  float* gmax = const_cast<float*>(v8.gmax);
  unsigned* cls = const_cast<unsigned*>(v8.cls);
  const dim3 gridg((unsigned)v8.granules);
  if (D == 256) hipLaunchKernelGGL(i8_granule_absmax_kernel<256>, gridg, dim3(256), 0, st, Kn, N, gmax, tail8);
  else if (D == 128) hipLaunchKernelGGL(i8_granule_absmax_kernel<128>, gridg, dim3(256), 0, st, Kn, N, gmax, tail8);
  else hipLaunchKernelGGL(i8_granule_absmax_kernel<64>, gridg, dim3(256), 0, st, Kn, N, gmax, tail8);
  static const float lambda = [] {   // (experiments: RAGRAPH_I8_ONE_SCALE=1 -- the single scale of rounds 3 / 4; RAGRAPH_I8_CUT_LAMBDA)
    const char* e = getenv("RAGRAPH_I8_ONE_SCALE");
    if (e && e[0] == '1') return 0.f;
    const char* l = getenv("RAGRAPH_I8_CUT_LAMBDA");
    return l ? (float)atof(l) : 60.f;
  }();
  hipLaunchKernelGGL(i8_cut_kernel, dim3(1), dim3(1024), 0, st, gmax, v8.granules, D, lambda, tail8);
  const dim3 grid8((unsigned)cdiv(npad * (D / 16), 256));
  if (D == 256) hipLaunchKernelGGL(keys_to_i8_kernel<256>, grid8, dim3(256), 0, st, Kn, N, npad, Kb8, tail8, gmax, cls);
  else if (D == 128) hipLaunchKernelGGL(keys_to_i8_kernel<128>, grid8, dim3(256), 0, st, Kn, N, npad, Kb8, tail8, gmax, cls);
  else hipLaunchKernelGGL(keys_to_i8_kernel<64>, grid8, dim3(256), 0, st, Kn, N, npad, Kb8, tail8, gmax, cls);
  RG_CHECK_LAUNCH("keys_to_bf16(int8 copy)");
  return RAGRAPH_OK;
}

// rows of D uint16: the bf16 copy padded to whole ring stages + one row that carries the bank's largest rounding error,
// then the int8 copy (half as many rows) + one row with its largest error and its scale
constexpr int64_t FILTER_COPY_SLACK_ROWS = 256;  // rows of 2 D bytes: >= 16 KB at any supported D
extern "C" int64_t ragraph_keys_bf16_rows(int64_t N) {
  if (N < 1) return 0;
  const int64_t npad = filter_round_up(N);
  // + FILTER_COPY_SLACK_ROWS: at D = 64 a stage of the int8 copy is 512 keys, so the last stage of a bank padded to an odd
  // multiple of 256 keys reads 16 KB past the copy's rows (keys >= N never pass): the buffer must own those bytes
  // + the int8 copy's granule table (filter_i8_table_bytes: < 5 bytes per 32 KiB of int8 rows, i.e. per 64 rows of 2 D bytes at
  // most -- whatever D)
  return npad + 1 + npad / 2 + 1 + (npad / 4096 + 8) + FILTER_COPY_SLACK_ROWS;
}

extern "C" int ragraph_topk_cosine_filtered_cap(int k) { return 2048; }
static int rescore_slices(int64_t B, int k);
// slots of a query's candidate region: one list, or (<= 64 queries) one full-size list per rescoring slice
static int filter_cap(int64_t B, int k) { return ragraph_topk_cosine_filtered_cap(k) * (B <= 64 ? rescore_slices(B, k) : 1); }

static size_t filter_ws_carve(char* w, int64_t B, int D, int k, int cap, struct FilterWs* out);

constexpr size_t FILTER_STATS_BYTES = 256;  // (the statistics block plus the slack that aligns it)
extern "C" size_t ragraph_topk_cosine_filtered_stats_offset(size_t ws_bytes) {
  return ws_bytes < FILTER_STATS_INTS * sizeof(int) ? 0 : (ws_bytes - FILTER_STATS_INTS * sizeof(int)) & ~(size_t)15;
}

static size_t filter_workspace_bytes(int64_t B, int64_t N, int D, int k, int n_shards) {
  if (B < 1 || N < 1 || k < 1 || n_shards < 1 || !filter_dim_ok(D)) return 0;
  const int cap = filter_cap(B, k);
  // run_filtered may turn the planned bound pass into an exact level 0 (a shard shorter than twice the prefix, a shard's
  // share of a pooled sample, the schedule switches): size for whichever of the two needs more, and run_filtered checks
  // the schedule it really runs against ws_bytes before carving
  FilterSchedule sc = filter_schedule(B, N, D, k, n_shards);
  size_t level0 = filter_level0_ws(sc, B, D, k);
  if (sc.bound_keys > 0) {
    sc.bound_keys = 0;
    const size_t exact0 = filter_level0_ws(sc, B, D, k);
    if (exact0 > level0) level0 = exact0;
  }
  return level0 + filter_ws_carve(nullptr, B, D, k, cap, nullptr) + FILTER_STATS_BYTES;
}

extern "C" size_t ragraph_topk_cosine_filtered_workspace_bytes(int64_t B, int64_t N, int D, int k) {
  return filter_workspace_bytes(B, N, D, k, 1);
}
// The sharded entry plans for (plan_N, n_shards): its first sample can be another one than the single bank's of plan_N rows.
extern "C" size_t ragraph_topk_cosine_filtered_sharded_workspace_bytes(int64_t B, int64_t plan_N, int D, int k, int n_shards) {
  if (!filter_dim_ok(D)) return 0;  // (ragraph_topk_cosine_f32 alone takes other widths)
  const size_t a = filter_workspace_bytes(B, plan_N, D, k, n_shards), b = filter_workspace_bytes(B, plan_N, D, k, 1);
  const size_t c = ragraph_topk_cosine_workspace_bytes(B, plan_N, D, k);   // (a short shard's exact top-k)
  const size_t ab = a > b ? a : b;  // (exchange = NULL runs the single-bank schedule)
  return ab > c ? ab : c;
}

extern "C" int ragraph_topk_cosine_filtered_i8_levels(int64_t B, int64_t N, int D, int k) {
  if (B < 1 || N < 1 || k < 1 || k > 32 || k > N || !filter_dim_ok(D)) return 0;
  const FilterSchedule sc = filter_schedule(B, N, D, k);
  return filter_i8_levels(sc, B, D, N);
}

extern "C" int ragraph_topk_cosine_filtered_plan(int64_t B, int64_t N, int D, int k, int64_t plan[7]) {
  RG_REQUIRE(plan, RAGRAPH_EINVAL, "topk_cosine_filtered_plan: null pointer");
  RG_REQUIRE(filter_dim_ok(D), RAGRAPH_EUNSUPPORTED, "topk_cosine_filtered_plan: D=%d not in {64,128,256}", D);
  RG_REQUIRE(B >= 1 && N >= 1 && k >= 1 && k <= 32 && k <= N, RAGRAPH_EINVAL, "topk_cosine_filtered_plan: bad B/N/k");
  const FilterSchedule sc = filter_schedule(B, N, D, k);
  plan[0] = sc.n0;
  plan[1] = sc.bound_keys > 0 ? 2 : sc.slab0;
  plan[2] = sc.nlev;
  for (int l = 0; l < FILTER_MAX_LEVELS; ++l) plan[3 + l] = l < sc.nlev ? sc.ends[l] : 0;
  plan[6] = sc.bound_keys;
  return sc.nlev;
}

// Everything one call keeps in its workspace behind level 0's scratch.
struct FilterWs {
  float* Qn;            // [B,D] normalised queries
  uint16_t* Qb;         // (B <= FILTER_QB_MAX_B) the same as bf16 B operands in fragment order, padded to whole groups of 32
  float* eq;            // [B] |dq|
  int* count;           // [B][filter_count_stride(B)] candidate slots reserved in the current level (per sub-list)
  unsigned char* flag;  // [B] the list overflowed at an earlier level
  int* cand;            // [B,cap] candidate keys
  int* gmax;            // [B, FILTER_BOUND_PARTS_MAX] part maxima of the bound pass
  float* theta;         // [B] the first bound
  int* overflow_list;   // [B] queries the final level sends to the exact fallback
  int* fix_done;        // [FILTER_FIX_MAX_Q] tickets, [.. x FILTER_FIX_SLICES x 32] partial winners of the sliced fallback scans
  float* fix_s;
  int64_t* fix_i;
  float* part_s;        // (B <= 64) sliced rescoring: [B][8][k] partial winners
  int* part_i;
  float* eq8;           // [B] |dq| of the int8 rounding, [B] the query's int8 scale (int8 levels)
  float* qscale;
  signed char* Qb8;     // (B <= FILTER_QB_MAX_B) the queries as int8 B operands in fragment order, padded to whole groups of 32
};

static size_t filter_ws_carve(char* w, int64_t B, int D, int k, int cap, FilterWs* out) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* ptr = w ? w + off : nullptr;
    off += align_up(bytes, 256);
    return ptr;
  };
  FilterWs f;
  f.Qn = reinterpret_cast<float*>(take((size_t)B * D * sizeof(float)));
  f.Qb = B <= FILTER_QB_MAX_B ? reinterpret_cast<uint16_t*>(take((size_t)((B + 31) / 32 * 32) * D * sizeof(uint16_t))) : nullptr;
  f.eq = reinterpret_cast<float*>(take((size_t)B * sizeof(float)));
  f.count = reinterpret_cast<int*>(take((size_t)B * filter_count_stride(B) * sizeof(int)));
  f.flag = reinterpret_cast<unsigned char*>(take((size_t)B));
  // (a call that may keep scored lists -- {key, I} -- gets 8 bytes per slot; sharded calls of the same shape do not use them)
  f.cand = reinterpret_cast<int*>(take((size_t)B * cap * (filter_scored_lists(B, D, k) ? sizeof(int2) : sizeof(int))));
  f.gmax = reinterpret_cast<int*>(take((size_t)B * filter_bound_parts(k, INT64_MAX, 256) * sizeof(int)));
  f.theta = reinterpret_cast<float*>(take((size_t)B * sizeof(float)));
  f.overflow_list = reinterpret_cast<int*>(take((size_t)B * sizeof(int)));
  f.fix_done = reinterpret_cast<int*>(take((size_t)FILTER_FIX_MAX_Q * sizeof(int)));
  f.fix_s = reinterpret_cast<float*>(take((size_t)FILTER_FIX_MAX_Q * FILTER_FIX_SLICES * 32 * sizeof(float)));
  f.fix_i = reinterpret_cast<int64_t*>(take((size_t)FILTER_FIX_MAX_Q * FILTER_FIX_SLICES * 32 * sizeof(int64_t)));
  f.part_s = B <= 64 ? reinterpret_cast<float*>(take((size_t)B * 8 * k * sizeof(float))) : nullptr;
  f.part_i = B <= 64 ? reinterpret_cast<int*>(take((size_t)B * 8 * k * sizeof(int))) : nullptr;
  f.eq8 = reinterpret_cast<float*>(take((size_t)B * sizeof(float)));
  f.qscale = reinterpret_cast<float*>(take((size_t)B * sizeof(float)));
  f.Qb8 = B <= FILTER_QB_MAX_B ? reinterpret_cast<signed char*>(take((size_t)((B + 31) / 32 * 32) * D)) : nullptr;
  if (out) *out = f;
  return off;
}

// A handful of queries: S workgroups rescore a query (S k <= 256 partial winners for the merge launch), and the direct
// kernel keeps S sub-lists per query, one per rescoring workgroup (filter_common.h: FILTER_COUNT_STRIDE).
static int rescore_slices(int64_t B, int k) {
  static const int slice_env = [] {  // RAGRAPH_RESCORE_SLICES: A/B (0 or 1 = never slice; a power of two <= 8)
    const char* e = getenv("RAGRAPH_RESCORE_SLICES");
    return e ? atoi(e) : -1;
  }();
  int S = B <= 16 ? 8 : (B <= 32 ? 4 : (B <= 64 ? 2 : 1));
  if (slice_env >= 0) S = slice_env >= 8 ? 8 : (slice_env >= 4 ? 4 : (slice_env >= 2 ? 2 : 1));
  if (B > 64) S = 1;  // (part_s / part_i exist up to 64 queries)
  while (S > 1 && S * k > 256) S >>= 1;
  return S;
}

// Ring-kernel launch shared by the filter levels and the bound pass (B > 256: the direct kernel takes smaller batches).
template <int D, int QW, bool BOUND, bool I8 = false, bool SCORED = false, bool PIPE = false>
static int launch_ring(FilterParams p, int64_t B, int prof_slot, hipStream_t st) {
  using C = FilterCfg<I8 ? D / 2 : D>;
  p.qtiles = cdiv(B, (int64_t)C::WAVES * QW);
  const int CUS = filter_device_cus();
  p.xcd_map = p.qtiles >= 64 ? 1 : 0;
  p.wgs_per_group = CUS / (p.xcd_map ? 8 : 1);
  {
    const char* e = getenv("RAGRAPH_FILTER_PARTNER_LEAD");  // (read per call: A/B; 0 = equal priorities, the hardware's age order)
    p.partner_lead = e ? atoi(e) : 1;
  }
  // shortest piece of a key stream a workgroup takes: 8 stages when there is work for everybody, fewer on short launches
  // (a bound pass of 18 stages x 6 query tiles gave 14 workgroups 8 stages each and 242 nothing: 19 us of stage loop where
  // 108 workgroups need 2.5; tools/check_segment_plan.cpp covers lb_min = 1)
  {
    const int64_t per_wg = p.qtiles * p.nstages_total / CUS;
    p.lb_min = per_wg >= 16 ? 8 : (per_wg >= 8 ? 4 : (per_wg >= 3 ? 2 : 1));
  }
  const int64_t nq0 = p.xcd_map ? (p.qtiles + 7) / 8 : p.qtiles;
  for (int v = 0; v < 2; ++v) {
    const int64_t nq = nq0 - v;
    p.depth[v] = 0;
    if (nq < 1 || (v == 1 && (!p.xcd_map || p.qtiles % 8 == 0))) continue;
    p.depth[v] = SegmentWalker::choose_depth(nq, p.nstages_total, p.wgs_per_group, p.lb_min, 0).depth;
    // Short launches cut the remainder at once (depth 0): a workgroup WALKS the plan's lockstep steps to its segment, every
    // step a few 64-bit divisions, and with a few tiles over 256 workgroups the last ones walk ~40 of them -- 22.8 us of
    // a 39-us launch (2708 queries x 10 000 keys: -DRG_RING_STAMPS); the re-reads the steps save are nothing at this size.
    static const int depth_env = [] { const char* e = getenv("RAGRAPH_FILTER_DEPTH"); return e ? atoi(e) : -1; }();  // A/B
    if (p.qtiles * p.nstages_total / CUS < 16) p.depth[v] = 0;
    if (depth_env >= 0) p.depth[v] = depth_env < p.depth[v] ? depth_env : p.depth[v];
  }
  static DeviceOnce lds_once;  // per template instance and device (common.h)
  if (hipError_t e = raise_dynamic_lds(lds_once, &topk_filter_kernel<D, QW, BOUND, I8, SCORED, PIPE>, (int)C::LDS_BYTES); e != hipSuccess) {
    set_error("topk_cosine_filtered: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    return RAGRAPH_EDEVICE;
  }
  if (t_prof) (void)hipEventRecord(t_prof->ev[2 * prof_slot], st);
  hipLaunchKernelGGL((topk_filter_kernel<D, QW, BOUND, I8, SCORED, PIPE>), dim3((unsigned)CUS), dim3(C::THREADS), C::LDS_BYTES, st, p);
  if (t_prof) (void)hipEventRecord(t_prof->ev[2 * prof_slot + 1], st);
  RG_CHECK_LAUNCH("topk_cosine_filtered(filter)");
#ifdef RG_RING_STAMPS
  {
    (void)hipDeviceSynchronize();
    unsigned long long t[2][2][8];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ring_t), sizeof(t));
    for (int b = 0; b < 2; ++b)
      fprintf(stderr, "[ring stamps, %s%s launch, %s workgroup, 10 ns ticks] operands %lld thresholds %lld ring primed %lld stages %lld "
              "flush %lld (entered %lld after workgroup 0)\n", BOUND ? "bound" : "filter", I8 ? " int8" : "", b ? "last" : "first",
              (long long)(t[BOUND][b][1] - t[BOUND][b][0]), (long long)(t[BOUND][b][2] - t[BOUND][b][1]),
              (long long)(t[BOUND][b][3] - t[BOUND][b][2]), (long long)(t[BOUND][b][4] - t[BOUND][b][3]),
              (long long)(t[BOUND][b][5] - t[BOUND][b][4]), (long long)(t[BOUND][b][0] - t[BOUND][0][0]));
    unsigned long long span[2][2], mx[2][8];
    (void)hipMemcpyFromSymbol(span, HIP_SYMBOL(g_ring_span), sizeof(span));
    (void)hipMemcpyFromSymbol(mx, HIP_SYMBOL(g_ring_max), sizeof(mx));
    fprintf(stderr, "[ring stamps, all workgroups] first entry to last exit %lld; longest first-segment phases: operands %lld thresholds %lld "
            "ring primed %lld stages %lld flush %lld\n", (long long)(span[BOUND][1] - span[BOUND][0]), (long long)mx[BOUND][1],
            (long long)mx[BOUND][2], (long long)mx[BOUND][3], (long long)mx[BOUND][4], (long long)mx[BOUND][5]);
    unsigned long long init_span[2][2] = {{~0ull, 0ull}, {~0ull, 0ull}}, zero[2][8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ring_span), init_span, sizeof(init_span));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ring_max), zero, sizeof(zero));
  }
#endif
#ifdef RG_TOPK_TIMING
  {
    (void)hipDeviceSynchronize();
    unsigned long long t[8];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_filter_timing), sizeof(t));
    const double n = (double)t[5];
    if (n > 0)
      fprintf(stderr, "[filter timing] slot %d D=%d wave-stages=%.0f ticks/stage: wait_full %.1f compute %.1f signal+vmcnt "
              "%.1f wait_free %.1f dma_issue %.1f total %.1f; flushes in the loop: %.0f, %.1f ticks each = %.1f per stage\n",
              prof_slot, D, n, t[0] / n, t[1] / n, t[2] / n, t[3] / n, t[4] / n, (t[0] + t[1] + t[2] + t[3] + t[4]) / n, (double)t[7],
              t[7] ? (double)t[6] / (double)t[7] : 0.0, t[6] / n);
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_filter_timing), zero, sizeof(zero));
  }
#endif
  return RAGRAPH_OK;
}

// One bf16 pass over keys [key0, key1) of the bank copy: a filter level (bound_groups = 0: candidates of every query
// whose threshold `thr` describes) or the bound pass (bound_groups = k group maxima into gmax).  Up to 256 queries take
// the direct kernel (topk_filter_direct.hip: the stream, not the matrix work, is what such a call costs), more the ring
// kernel with two -- at D = 64 and long streams four -- query groups per wave.
template <int D>
static int run_bf16_pass(const FilterWs& f, const uint16_t* Kb, int64_t B, int64_t key0, int64_t key1, const FilterThr& thr,
                         int cap, int bound_groups, int prof_slot, hipStream_t st, const signed char* Kb8 = nullptr,
                         bool scored = false) {
  using C = FilterCfg<D>;
  {
    if (Kb8 && B > 256) {  // an int8 level (filter_i8_levels): the ring kernel over the int8 copy, stages of twice as many keys
      using C8 = FilterCfg<D / 2>;
      FilterParams p{};
      p.Qn = f.Qn;
      p.Kb = reinterpret_cast<const uint16_t*>(Kb8);
      p.thr = thr;
      p.count = f.count;
      p.cstride = filter_count_stride(B);
      static const bool i8_image = [] { const char* e = getenv("RAGRAPH_FILTER_I8_IMAGE"); return !e || atoi(e) != 0; }();  // A/B
      // (the int8 image, or NULL beyond FILTER_QB_MAX_B queries: quantised per segment)
      p.Qb = i8_image ? reinterpret_cast<const uint16_t*>(f.Qb8) : nullptr;
      p.cand = f.cand;
      p.B = B;
      p.N = key1;
      p.cap = cap;  // (scored lists: {key, I} pairs, the same number of slots -- filter_ws_carve gives them 8 bytes each)
      RG_REQUIRE(key0 % C8::STAGE_KEYS == 0, RAGRAPH_EINVAL, "topk_cosine_filtered: an int8 level must start at a whole stage "
                 "(key %lld, %d keys per stage)", (long long)key0, (int)C8::STAGE_KEYS);
      p.stage_base = key0 / C8::STAGE_KEYS;
      p.nstages_total = cdiv(key1 - key0, C8::STAGE_KEYS);
      // int8 operands are 16 bytes per 64 elements: SIX query groups per wave (tile = 768 queries) fit the registers four
      // bf16 groups take (224 VGPRs, no scratch), and an A fragment then feeds six MFMAs, a stage 3072 cycles of them
      // between two ring hand-overs: the bench's last level 14.45 -> 13.8 ms (A/B on one box, profiles/r3_i8_ab.txt).
      // Eight groups (tile = 1024) spill (256 VGPRs + 80 B of scratch): 14.2 ms.  Long streams only, and only where the
      // larger tile does not add padding queries (the last tile of 4096 queries would be a third full).
      const char* e = getenv("RAGRAPH_FILTER_I8_QW");
      const int qw_env = e ? atoi(e) : 0;
      const int64_t pad64 = cdiv(B, (int64_t)512) * 512 - B;
      auto fits = [&](int64_t tile) {  // a long stream per workgroup, and at most 2 % more padding queries than tiles of 512
        return cdiv(B, tile) * p.nstages_total >= 32 * (int64_t)filter_device_cus() && (cdiv(B, tile) * tile - B - pad64) * 50 <= B;
      };
      // (D = 64: eight groups are 32 registers of operands -- no spill -- and 65 536 x 4M x 64 runs 15.5 ms against 16.3 with
      // six and 17.0 with four)
      const int qw = qw_env ? qw_env : (D == 64 && fits(1024) ? 128 : (fits(768) ? 96 : 64));
      const bool long128 = qw == 128 && cdiv(B, (int64_t)1024) * p.nstages_total >= 32 * (int64_t)filter_device_cus();
      const bool long96 = qw == 96 && cdiv(B, (int64_t)768) * p.nstages_total >= 32 * (int64_t)filter_device_cus();
      if constexpr (D == 256) {
        // four groups per wave: the epilogue of a sub-tile inside the next one's MFMAs (PIPE: 222 VGPRs) -- 512 x 1M 0.213 ->
        // 0.2005 ms, 1024 0.324 -> 0.308, 4096 1.053 -> 1.026, 16 384 3.53 -> 3.49; on the long streams that take six groups
        // it only draws level (13.03 vs 13.03 - 13.13 ms: six groups in one set of accumulators stay).  RAGRAPH_FILTER_PIPE=0/2: A/B
        const char* pe = getenv("RAGRAPH_FILTER_PIPE");  // (read per call)
        const int pv = pe ? atoi(pe) : 1;
        if ((pv == 1 && !long128 && !long96) || pv == 2)
          return scored ? launch_ring<D, 64, false, true, true, true>(p, B, prof_slot, st)
                        : launch_ring<D, 64, false, true, false, true>(p, B, prof_slot, st);
      }
      if (scored) {
        if (long128) return launch_ring<D, 128, false, true, true>(p, B, prof_slot, st);
        if (long96) return launch_ring<D, 96, false, true, true>(p, B, prof_slot, st);
        return launch_ring<D, 64, false, true, true>(p, B, prof_slot, st);
      }
      if (long128) return launch_ring<D, 128, false, true>(p, B, prof_slot, st);
      if (long96) return launch_ring<D, 96, false, true>(p, B, prof_slot, st);
      return launch_ring<D, 64, false, true>(p, B, prof_slot, st);
    }
  }
  if (B <= 256) {
    DirectArgs a{};
    a.Qb = Kb8 ? reinterpret_cast<const uint16_t*>(f.Qb8) : f.Qb;
    a.Kb = Kb8 ? reinterpret_cast<const uint16_t*>(Kb8) : Kb;
    a.i8 = Kb8 ? 1 : 0;
    a.scored = Kb8 && scored ? 1 : 0;
    a.B = B;
    a.key0 = key0;
    a.key1 = key1;
    a.thr = thr;
    a.count = f.count;
    a.cand = f.cand;
    a.cap = cap;
    a.nsub = rescore_slices(B, thr.k);
    a.gmax_out = f.gmax;
    a.bound_groups = bound_groups;
    if (t_prof) (void)hipEventRecord(t_prof->ev[2 * prof_slot], st);
    const int rc = launch_filter_direct<D>(a, st);
    if (t_prof) (void)hipEventRecord(t_prof->ev[2 * prof_slot + 1], st);
    return rc;
  }
  FilterParams p{};
  p.Qn = f.Qn;
  p.Kb = Kb;
  p.thr = thr;
  p.count = f.count;
  p.cstride = filter_count_stride(B);
  p.Qb = f.Qb;
  p.cand = f.cand;
  p.gmax = bound_groups > 0 ? f.gmax : nullptr;
  p.ngroups = bound_groups;
  p.B = B;
  p.N = key1;
  p.cap = cap;
  p.stage_base = key0 / C::STAGE_KEYS;  // key0 is a multiple of 256
  p.nstages_total = cdiv(key1 - key0, C::STAGE_KEYS);
  if (bound_groups > 0) return launch_ring<D, 64, true>(p, B, prof_slot, st);
  if constexpr (D == 64) {  // short rows leave registers for four query groups per wave: half the LDS reads and ring
                            // hand-overs per MFMA (the edge flavour's D)
    // (only where a workgroup keeps its 1024 queries for a long stream: on a short bank the larger tiles mean fewer,
    // shorter segments, each paying the operand loads again -- 8192 x 40000 x 64: 0.58 vs 0.46 ms)
    if (filter_wide_waves(B) && cdiv(B, (int64_t)1024) * p.nstages_total >= 32 * (int64_t)filter_device_cus())
      return launch_ring<D, 128, false>(p, B, prof_slot, st);
  }
  return launch_ring<D, 64, false>(p, B, prof_slot, st);
}

// Exact rescoring of a level's candidates (+ merge with the running result when `merge`) and canonical selection.
template <int D>
static int run_rescore(const FilterWs& f, const float* Kn, int64_t N, int64_t B, int cap, int k, int64_t idx_base, int merge,
                       int final_level, float* out_scores, int64_t* out_idx, int* overflow, int* fallback_done, bool few,
                       hipStream_t st, const FilterThr* scored_thr = nullptr, int* cstat = nullptr) {
  const float* ps = merge ? out_scores : nullptr;
  const int64_t* pi = merge ? out_idx : nullptr;
  static const int64_t wide_max_b = [] {  // RAGRAPH_RESCORE_WIDE_BELOW: A/B of the crossover
    const char* e = getenv("RAGRAPH_RESCORE_WIDE_BELOW");
    return e ? (int64_t)atoll(e) : (int64_t)2048;  // measured: 512 queries 0.39 (wide) vs 0.44 ms, 1024-2048 equal, 4095: 1.98 vs 1.89
  }();
  static const int64_t wide_coop_max_b = [] {  // RAGRAPH_RESCORE_WIDE_COOP_MAX: A/B
    const char* e = getenv("RAGRAPH_RESCORE_WIDE_COOP_MAX");
    return e ? (int64_t)atoll(e) : (int64_t)256;
  }();
  // a handful of queries: S workgroups per query, each with its sub-list
  const int S = B <= 256 ? rescore_slices(B, k) : 1;
  const int cs = filter_count_stride(B);
  // mid-sized calls: an overflowed query is scanned by its own rescoring wave (no fallback launch); large batches keep
  // the dedicated launch, whose four-wave workgroups scan a bank faster when MANY queries overflow
  // the one-wave-per-query kernels leave overflowed queries to topk_overflow_fixup_kernel (their own wave scanning a
  // million keys took 94 ms; RAGRAPH_RESCORE_SCAN_IN_WAVE=1: the old behaviour, A/B)
  static const bool scan_in_wave = [] { const char* e = getenv("RAGRAPH_RESCORE_SCAN_IN_WAVE"); return e && atoi(e) != 0; }();
  const int64_t scan_n = scan_in_wave && B <= FILTER_SLAB_MAX_B ? N : 0;
  if (scored_thr) {  // (filter_scored_lists: a large call's int8 level)
    *fallback_done = scan_n > 0;
    static const bool small_env = [] { const char* e = getenv("RAGRAPH_RESCORE_SCORED_SMALL"); return !e || atoi(e) != 0; }();  // A/B
    if (B >= 8192 && small_env)
      hipLaunchKernelGGL((topk_rescore_scored_kernel<D, true>), dim3((unsigned)cdiv(B, 2)), dim3(128), 0, st, f.Qn, Kn, f.count,
                         reinterpret_cast<const int2*>(f.cand), B, cap, cs, k, idx_base, ps, pi, final_level, out_scores, out_idx,
                         overflow, f.overflow_list, f.flag, scan_n, *scored_thr, cstat);
    else
      hipLaunchKernelGGL((topk_rescore_scored_kernel<D, false>), dim3((unsigned)cdiv(B, 2)), dim3(128), 0, st, f.Qn, Kn, f.count,
                         reinterpret_cast<const int2*>(f.cand), B, cap, cs, k, idx_base, ps, pi, final_level, out_scores, out_idx,
                         overflow, f.overflow_list, f.flag, scan_n, *scored_thr, cstat);
  } else if (B < wide_max_b && S > 1) {
    hipLaunchKernelGGL((topk_rescore_wide_kernel<D, true, true>), dim3((unsigned)B, (unsigned)S), dim3(256), 0, st, f.Qn, Kn, f.count,
                       f.cand, B, N, cap, cs, k, idx_base, ps, pi, final_level, out_scores, out_idx, overflow, f.overflow_list,
                       f.flag, f.part_s, f.part_i, cstat);
    *fallback_done = 0;   // (every call lists its overflowed queries for the sliced fixup launch)
#ifdef RG_WIDE_TIMING
    {
      (void)hipDeviceSynchronize();
      unsigned long long t[16];
      (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_wide_t), sizeof(t));
      fprintf(stderr, "[wide timing, block 0, 10 ns ticks]");
      for (int i = 1; i < 10; ++i) fprintf(stderr, " %d:%lld", i, (long long)(t[i] - t[0]));
      fprintf(stderr, "\n");
    }
#endif
  } else if (B <= wide_coop_max_b) {  // one workgroup per CU: its 70 KB of tiles cost no occupancy
    *fallback_done = 0;
    hipLaunchKernelGGL((topk_rescore_wide_kernel<D, false, true>), dim3((unsigned)B), dim3(256), 0, st, f.Qn, Kn, f.count, f.cand,
                       B, N, cap, cs, k, idx_base, ps, pi, final_level, out_scores, out_idx, overflow, f.overflow_list, f.flag,
                       (float*)nullptr, (int*)nullptr, cstat);
  } else if (B < wide_max_b) {  // too few queries to fill the chip with one wave each
    *fallback_done = 0;
    hipLaunchKernelGGL((topk_rescore_wide_kernel<D, false, false>), dim3((unsigned)B), dim3(256), 0, st, f.Qn, Kn, f.count, f.cand,
                       B, N, cap, cs, k, idx_base, ps, pi, final_level, out_scores, out_idx, overflow, f.overflow_list, f.flag,
                       (float*)nullptr, (int*)nullptr, cstat);
  } else if (rescore_coop() && few) {
    *fallback_done = scan_n > 0;
    hipLaunchKernelGGL((topk_rescore_coop_kernel<D, 32, true>), dim3((unsigned)cdiv(B, 2)), dim3(128), 0, st, f.Qn, Kn, f.count,
                       f.cand, B, cap, cs, k, idx_base, ps, pi, final_level, out_scores, out_idx, overflow, f.overflow_list, f.flag,
                       scan_n, cstat);
  } else if (rescore_coop()) {
    *fallback_done = scan_n > 0;
    hipLaunchKernelGGL((topk_rescore_coop_kernel<D, 32>), dim3((unsigned)cdiv(B, 2)), dim3(128), 0, st, f.Qn, Kn, f.count,
                       f.cand, B, cap, cs, k, idx_base, ps, pi, final_level, out_scores, out_idx, overflow, f.overflow_list, f.flag,
                       scan_n, cstat);
  } else
    hipLaunchKernelGGL((topk_rescore_kernel<D, 32>), dim3((unsigned)cdiv(B, 4)), dim3(256), 0, st, f.Qn, Kn, f.count, f.cand,
                       B, cap, cs, k, idx_base, ps, pi, final_level, out_scores, out_idx, overflow, f.overflow_list, f.flag, cstat);
  RG_CHECK_LAUNCH("topk_cosine_filtered(rescore)");
  return RAGRAPH_OK;
}

template <int D>
static int run_filtered(const float* Q, int64_t B, const float* Kn, const float* Kp, const uint16_t* Kb, int64_t N, int k,
                        int64_t idx_base, float* out_scores, int64_t* out_idx, int* overflow, int64_t* overflow_idx,
                        void* ws, size_t ws_bytes, void* stream, int64_t plan_N, float* theta, ragraph_exchange_fn exchange,
                        void* ctx, int n_shards) {
  hipStream_t st = as_stream(stream);
  const int cap = filter_cap(B, k);
  FilterSchedule sc = filter_schedule(B, plan_N, D, k, exchange ? n_shards : 1);  // (sharded banks: the same schedule on every shard)
  // A speculative first bound (this thread's prior).  Sharded banks: whether the call speculates must be the SAME decision on
  // every rank -- it removes the bound pass AND its exchange (phase 0) --, so it is taken from what every rank shares: the
  // prior (the caller derives it from pooled statistics and sets it on every rank alike) and the PLAN's bound pass (plan_N),
  // before any adjustment to this shard's own length.  The proof is the caller's too: a query is exact iff the k-th best of
  // the MERGED lists reaches the prior (ragraph_amd/sharded.py verifies at the rows' owner and re-runs without the prior).
  const float prior = t_prior;
  const bool prior_ok = prior == prior && prior > -2.f && prior < 2.f;
  const bool spec_x = exchange && prior_ok && sc.bound_keys > 0;
  if (spec_x && sc.nlev == 2 && B <= 4096) {  // (as below for one bank: the prior IS the bound a first level would give)
    sc.nlev = 1;
    sc.ends[0] = plan_N;
    if (sc.i8_levels > 1) sc.i8_levels = 1;
  }
  // Three levels under the prior, many shards: the first level (1 / 32 of the shard) exists to sharpen the bound pass's
  // bound, and under the group's prior it passes about ONE candidate per query and shard (measured, 8 shards of the 1M bank) --
  // a filter launch on the bf16 copy, a rescoring launch over every query and an exchange for nothing.  From
  // RAGRAPH_FILTER_SPEC_SHARDS_TWO_LEVELS shards (default 2: every sharded bank; 0 = never) the call runs levels [0, N / 4) and
  // [N / 4, N): emulated rank of 2 / 4 / 8, ms per step: 11.14 -> 10.83, 6.29 -> 5.92, 3.73 -> 3.46 (profiles/r6_multi_one_gpu.txt).
  static const int spec_two = [] { const char* e = getenv("RAGRAPH_FILTER_SPEC_SHARDS_TWO_LEVELS"); return e ? atoi(e) : 2; }();
  if (spec_x && sc.nlev == 3 && spec_two > 0 && n_shards >= spec_two) {
    sc.ends[0] = sc.ends[1];
    sc.ends[1] = sc.ends[2];
    sc.nlev = 2;
    if (sc.i8_levels > 2) sc.i8_levels = 2;
  }
  // A shard SHORTER than the largest one (shards of a bank whose exact duplicates were collapsed per shard hold different
  // numbers of unique rows): the same phases -- the exchanges must line up across the ranks -- over proportionally fewer
  // keys; a shard too short for that structure takes part as an EXACT participant: its fp32 top-k once, offered at every
  // exchange (exact scores of k distinct keys are valid lower bounds at every phase).
  bool exact_participant = false;
  if (exchange && plan_N - N > 1024) {
    int64_t prev = 0;
    for (int l = 0; l + 1 < sc.nlev; ++l) {
      int64_t e = (int64_t)((double)sc.ends[l] * (double)N / (double)plan_N) / 512 * 512;   // (512: whole int8 stages at any width)
      if (e < prev + 512 || e + 512 > N) exact_participant = true;
      sc.ends[l] = e;
      prev = e;
    }
    sc.bound_keys = (int64_t)((double)sc.bound_keys * (double)N / (double)plan_N) / FILTER_PAD_KEYS * FILTER_PAD_KEYS;
    if (sc.nlev > 1 && sc.bound_keys > sc.ends[0]) sc.bound_keys = sc.ends[0];
    if (sc.bound_keys / (FILTER_STAGE_BYTES / (2 * D)) < (int64_t)filter_bound_parts(k, sc.bound_keys, D, B, n_shards)) sc.bound_keys = 0;
    sc.n0 = sc.n0 < N ? sc.n0 : N;
    if (N < 16384 || N * 8 < plan_N) exact_participant = true;
  }
  if (exact_participant) {
    RG_REQUIRE(ws_bytes >= ragraph_topk_cosine_workspace_bytes(B, N, D, k), RAGRAPH_EWORKSPACE,
               "topk_cosine_filtered: workspace too small for a short shard's exact top-k");
    int rc0 = ragraph_topk_cosine_bank_f32(Q, B, Kn, D == 256 ? Kp : nullptr, N, D, k, idx_base, out_scores, out_idx, ws, ws_bytes, stream);
    if (rc0 != RAGRAPH_OK) return rc0;
    if (hipMemsetAsync(overflow, 0, sizeof(int), st) != hipSuccess) {
      set_error("topk_cosine_filtered: memset failed");
      return RAGRAPH_EDEVICE;
    }
    FilterThr t0{};
    t0.prev_scores = out_scores;
    t0.k = k;
    for (int ph = spec_x ? 1 : 0; ph < sc.nlev; ++ph) {   // phase 0 (not under a speculative bound) + one exchange behind every level but the last
      hipLaunchKernelGGL(filter_theta_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, st, t0, B, ph == (spec_x ? 1 : 0) ? 1 : 0, theta);
      RG_CHECK_LAUNCH("topk_cosine_filtered(theta)");
      exchange(ctx, ph);
    }
    return RAGRAPH_OK;
  }
  sc.ends[sc.nlev - 1] = N;
  if (sc.bound_keys > N / 2) sc.bound_keys = 0, sc.n0 = sc.n0 < N ? sc.n0 : N;
  if (exchange && n_shards > 1 && sc.bound_keys > 0 && B <= FILTER_SLAB_MAX_B && plan_N >= 4 * 4096) {  // (the cost-model branch)
    // G shards pool their samples through the exchange (the k-th largest of the union of every shard's best group
    // maxima): each scans 1/G of the prefix one bank would -- at least one stage per part
    const int64_t min_keys = filter_round_up((int64_t)k * (FILTER_STAGE_BYTES / (2 * D)));
    int64_t bk = filter_round_up(sc.bound_keys / n_shards);
    sc.bound_keys = bk < min_keys ? (min_keys < sc.bound_keys ? min_keys : sc.bound_keys) : bk;
  }
  static const int ablate = [] {  // RAGRAPH_FILTER_ABLATE=1: no key passes the filter (timing only, results invalid)
    const char* e = getenv("RAGRAPH_FILTER_ABLATE");
    return e ? atoi(e) : 0;
  }();
  if (t_prof) t_prof->have = t_prof->bound = 0;

  char* w = static_cast<char*>(ws);
  const size_t sample_ws = filter_level0_ws(sc, B, D, k);
  FilterWs f;
  const size_t used = sample_ws + filter_ws_carve(w + sample_ws, B, D, k, cap, &f) + FILTER_STATS_BYTES;
  RG_REQUIRE(used <= ws_bytes, RAGRAPH_EWORKSPACE, "topk_cosine_filtered: the schedule of this call needs %zu bytes of workspace, "
             "%zu given", used, ws_bytes);
  // the call's candidate statistics: the last FILTER_STATS_INTS ints of the workspace AS PASSED (ragraph_topk_cosine_filtered_stats_offset)
  int* stats = reinterpret_cast<int*>(w + ragraph_topk_cosine_filtered_stats_offset(ws_bytes));
  const unsigned* max_kerr2 = reinterpret_cast<const unsigned*>(Kb + filter_round_up(N) * D);
  // the int8 copy lies behind the bf16 copy and its tail row (ragraph_keys_to_bf16)
  const FilterI8View v8 = filter_i8_view(Kb, N, D);
  const signed char* Kb8 = v8.K8;
  const unsigned* tail8 = v8.tail8;
  sc.i8_levels = filter_i8_levels(sc, B, D, plan_N);
  // a speculative first bound (this thread's prior; single banks whose schedule has a bound pass to save): no bound pass,
  // theta = prior for every query, every level filters with max(prior, the running k-th best), and the verify launch
  // behind the last level sends the queries the prior was too high for to the exact scan
  const bool spec = exchange ? spec_x : (sc.bound_keys > 0 && prior_ok && N == plan_N);
  const bool bound = sc.bound_keys > 0 && !spec;
  // A first level exists to give the second a tighter bound than the bound pass could; the prior already is one.  Measured
  // with it (1M x 256, ms per call, two levels / one): 2048 queries 0.544 / 0.512, 4096: 0.920 / 0.898 -- but 16 384: 3.18 /
  // 3.85, and the three levels of 100 000 queries stay (20.5 ms per step against 22.8 with two): up to 4096 queries one level.
  if (spec && !exchange && sc.nlev == 2 && B <= 4096) {
    sc.nlev = 1;
    sc.ends[0] = N;
    if (sc.i8_levels > 1) sc.i8_levels = 1;
  }

  FilterStatsInit stats_init{};
  stats_init.nlev = sc.nlev;
  for (int l = 0; l < sc.nlev && l < 3; ++l) {
    stats_init.i8[l] = l >= sc.nlev - sc.i8_levels;
    const int64_t lk = sc.ends[l] - (l ? sc.ends[l - 1] : 0);
    stats_init.keys[l] = lk > INT_MAX ? INT_MAX : (int)lk;
  }
  // one launch: normalised queries, their bf16 rounding errors, empty lists, clear flags (+ group maxima at -inf)
  hipLaunchKernelGGL(filter_prep_kernel<D>, dim3((unsigned)cdiv(B <= FILTER_QB_MAX_B ? (B + 31) / 32 * 32 : B, 4)), dim3(256), 0, st, Q, B,
                     f.Qn, f.eq, f.count, f.flag, overflow, bound ? f.gmax : nullptr, bound ? filter_bound_parts(k, sc.bound_keys, D, B, exchange ? n_shards : 1) : k,
                     // (the bf16 operand image: only launches on the bf16 copy read it -- a call whose levels all run on the int8
                     // copy and that has no bound pass, e.g. every call under a prior, saves writing 2 D bytes per query)
                     B <= FILTER_QB_MAX_B && (B <= 256 || bound || sc.nlev > sc.i8_levels) ? f.Qb : nullptr,
                     filter_count_stride(B), sc.i8_levels > 0 ? f.eq8 : nullptr, f.qscale, B <= FILTER_QB_MAX_B ? f.Qb8 : nullptr, f.fix_done,
                     stats, stats_init, spec ? (exchange ? theta : f.theta) : nullptr, prior);
  RG_CHECK_LAUNCH("topk_cosine_filtered(prepare)");

  FilterThr thr{};
  thr.eq = f.eq;
  thr.max_kerr2 = max_kerr2;
  thr.eq8 = f.eq8;
  thr.qscale = f.qscale;
  thr.tail8 = tail8;
  thr.cls8 = v8.cls;
  thr.flag = f.flag;
  thr.k = k;
  const int parts = bound ? filter_bound_parts(k, sc.bound_keys, D, B, exchange ? n_shards : 1) : k;
  thr.ngroups = parts;
  thr.ablate = ablate;
  int rc = RAGRAPH_OK, fallback_done = 0;
  // the first bound: group maxima of a bf16 pass over a prefix, or an exact level 0 over the first n0 keys (out_scores /
  // out_idx hold every level's running result, local indices)
  if (spec) {
    // (nothing to compute: the prepare launch has written theta)
  } else if (bound) {
    rc = run_bf16_pass<D>(f, Kb, B, 0, sc.bound_keys, thr, cap, parts, 3, st);
    if (t_prof) {
      t_prof->bound = 1;
      t_prof->i8[3] = 0;
      t_prof->keys[3] = sc.bound_keys;
    }
  } else if (sc.slab0) {
    float* S = reinterpret_cast<float*>(w);  // one slab of scores, reused: written and read back while it is in cache
    for (int64_t b0 = 0; b0 < B && rc == RAGRAPH_OK; b0 += FILTER_SLAB_MAX_B) {
      const int64_t nb = B - b0 < FILTER_SLAB_MAX_B ? B - b0 : FILTER_SLAB_MAX_B;
      rc = ragraph_linear_f32(f.Qn + b0 * D, nb, D, Kn, sc.n0, nullptr, RAGRAPH_ACT_NONE, 0.f, S, stream);
      if (rc == RAGRAPH_OK)
        rc = ragraph_topk_rows_f32(S, nb, sc.n0, sc.n0, k, out_scores + b0 * k, out_idx + b0 * k, stream);
    }
  } else {
    rc = ragraph_topk_cosine_bank_f32(Q, B, Kn, D == 256 ? Kp : nullptr, sc.n0, D, k, 0, out_scores, out_idx, ws, sample_ws,
                                      stream);
  }
  if (rc != RAGRAPH_OK) return rc;
  if (exchange && spec) {  // theta = the prior on every shard (the prepare launch wrote it): nothing to pool, no phase 0
    thr.theta = theta;
  } else if (exchange) {  // the first bound leaves through theta / out_scores, and comes back as a bound on the k-th best of ALL shards
    thr.gmax = bound ? f.gmax : nullptr;
    thr.prev_scores = out_scores;
    if (bound)  // k lower bounds of distinct keys' exact scores, descending, where a level leaves its exact top-k
      hipLaunchKernelGGL(filter_bound_scores_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, st, thr, B, out_scores, theta);
    else
      hipLaunchKernelGGL(filter_theta_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, st, thr, B, 1, theta);
    RG_CHECK_LAUNCH("topk_cosine_filtered(theta)");
    exchange(ctx, 0);
    thr.theta = theta;
  } else if (bound && parts > k) {  // theta = the k-th largest of the part maxima, minus eps
    thr.gmax = f.gmax;
    hipLaunchKernelGGL(filter_bound_scores_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, st, thr, B, (float*)nullptr, f.theta);
    RG_CHECK_LAUNCH("topk_cosine_filtered(theta)");
  }
  int64_t key0 = 0;
  for (int l = 0; l < sc.nlev; ++l) {  // the first level re-reads [0, n0): its keys pass the bound and need no merge
    thr.gmax = (l == 0 && bound && !exchange && parts == k) ? f.gmax : nullptr;  // (k parts: the minimum, inline)
    if (!exchange) thr.theta = (spec || (l == 0 && bound && parts > k)) ? f.theta : nullptr;
    thr.prev_scores = out_scores;
    const bool i8_level = l >= sc.nlev - sc.i8_levels;
    // (sharded banks keep the plain lists: a level's threshold already is the k-th best over ALL shards -- sharper than
    // anything round 1 can find among this shard's keys, so nothing is pruned and only the second round's latency and the
    // 16-row tiles' occupancy are lost: emulated rank of 2 / 4 / 8 GPUs 13.49 -> 13.26 / 7.61 -> 8.00 / 4.76 -> 5.24 ms per
    // step, profiles/r3_emul.txt.  RAGRAPH_FILTER_SCORED_SHARDS = largest shard count that takes them: A/B.)
    const bool scored = i8_level && (!exchange || n_shards <= filter_scored_shards()) && filter_scored_lists(B, D, k);
    rc = run_bf16_pass<D>(f, Kb, B, key0, sc.ends[l], thr, cap, 0, l, st, i8_level ? Kb8 : nullptr, scored);
    if (t_prof) {
      t_prof->i8[l] = l >= sc.nlev - sc.i8_levels;
      t_prof->keys[l] = sc.ends[l] - key0;
    }
    if (rc != RAGRAPH_OK) return rc;
    static const bool dbg_counts = [] { const char* e = getenv("RAGRAPH_FILTER_DEBUG_COUNTS"); return e && atoi(e) != 0; }();
    if (dbg_counts) {  // diagnostic (synchronises): the level's candidate counts per query, before the rescoring resets them
      const int cs_ = filter_count_stride(B);
      std::vector<int> h((size_t)B * cs_);
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(h.data(), f.count, h.size() * sizeof(int), hipMemcpyDeviceToHost);
      const int nsub_ = B <= 256 ? rescore_slices(B, k) : 1;
      long long tot = 0, mx = 0, over_half = 0, over_cap = 0;
      for (int64_t b = 0; b < B; ++b) {
        long long c = 0;
        for (int u = 0; u < nsub_; ++u) c += h[(size_t)b * cs_ + u];
        tot += c;
        mx = c > mx ? c : mx;
        over_half += c > cap / 2;
        over_cap += c > cap;
      }
      fprintf(stderr, "[filter counts] level %d (%s%s, keys %lld..%lld): mean %.1f max %lld per query; %lld of %lld queries above %d, %lld above %d\n",
              l, i8_level ? "int8" : "bf16", scored ? ", scored" : "", (long long)key0, (long long)sc.ends[l], (double)tot / (double)B, mx,
              over_half, (long long)B, cap / 2, over_cap, cap);
    }
    if (t_prof) t_prof->have = l + 1;
    rc = run_rescore<D>(f, Kn, N, B, cap, k, idx_base, l > 0, l == sc.nlev - 1, out_scores, out_idx, overflow,
                        &fallback_done, exchange != nullptr && l > 0, st, scored ? &thr : nullptr, l < 3 ? stats + 2 + l : nullptr);
    if (rc != RAGRAPH_OK) return rc;
    key0 = sc.ends[l];
    if (spec && !exchange && l + 1 < sc.nlev) {  // the next level filters with max(prior, the exact k-th best so far)
      FilterThr t2 = thr;
      t2.gmax = nullptr;
      t2.theta = nullptr;
      hipLaunchKernelGGL(filter_theta_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, st, t2, B, 0, f.theta);
      RG_CHECK_LAUNCH("topk_cosine_filtered(theta)");
    }
    if (exchange && l + 1 < sc.nlev) {  // this shard's k-th exact score so far sharpens theta; then the other shards'
      FilterThr t2 = thr;
      t2.gmax = nullptr;
      t2.theta = nullptr;
      hipLaunchKernelGGL(filter_theta_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, st, t2, B, 0, theta);
      RG_CHECK_LAUNCH("topk_cosine_filtered(theta)");
      exchange(ctx, 1 + l);
    }
  }
  // overflowed queries (none on ordinary banks): exact fp32 scan on the device -- no host read-back (the sliced
  // rescoring of a handful of queries has done it inside its merge launch)
  if (spec && !exchange) {   // (a shard's lists prove nothing alone: the owner of a row verifies the MERGED k-th best)
    hipLaunchKernelGGL(filter_verify_prior_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, st, out_scores, B, k, prior, 1, f.flag,
                       overflow, f.overflow_list, stats);
    RG_CHECK_LAUNCH("topk_cosine_filtered(verify)");
    fallback_done = 0;
  }
  if (fallback_done) return RAGRAPH_OK;
  hipLaunchKernelGGL(topk_overflow_fixup_kernel<D>, dim3(256), dim3(256), 0, st, f.Qn, Kn, N, k, idx_base,
                     overflow, f.overflow_list, overflow_idx, out_scores, out_idx, f.fix_done, f.fix_s, f.fix_i, B, f.flag,
                     exchange ? nullptr : stats);
  RG_CHECK_LAUNCH("topk_cosine_filtered(overflow fallback)");
  return RAGRAPH_OK;
}

static int filtered_entry(const float* Q, int64_t B, const float* Kn, const float* Kp, const uint16_t* Kb, int64_t N, int D,
                          int k, int64_t idx_base, float* out_scores, int64_t* out_idx, int* overflow, int64_t* overflow_idx,
                          void* ws, size_t ws_bytes, void* stream, int64_t plan_N, float* theta, ragraph_exchange_fn exchange,
                          void* ctx, int n_shards) {
  RG_REQUIRE(Q && Kn && Kb && out_scores && out_idx && overflow && ws, RAGRAPH_EINVAL, "topk_cosine_filtered: null pointer");
  RG_REQUIRE(n_shards >= 1, RAGRAPH_EINVAL, "topk_cosine_filtered: n_shards=%d", n_shards);
  RG_REQUIRE(filter_dim_ok(D), RAGRAPH_EUNSUPPORTED, "topk_cosine_filtered: D=%d not in {64,128,256}", D);
  RG_REQUIRE(B >= 1 && N >= 1 && k >= 1 && k <= 32 && k <= N, RAGRAPH_EINVAL, "topk_cosine_filtered: bad B/N/k");
  RG_REQUIRE(N < (int64_t)INT_MAX - 1024, RAGRAPH_EUNSUPPORTED, "topk_cosine_filtered: shard rows must fit int32");
  RG_REQUIRE(plan_N >= N && (exchange || plan_N - N <= 1024), RAGRAPH_EINVAL, "topk_cosine_filtered: plan_N must be the largest shard's size");
  RG_REQUIRE(!exchange || theta, RAGRAPH_EINVAL, "topk_cosine_filtered: an exchange needs the theta buffer");
  RG_REQUIRE(aligned16(Q) && aligned16(Kn) && aligned16(Kb) && aligned16(ws), RAGRAPH_EINVAL,
             "topk_cosine_filtered: pointers must be 16-B aligned");
  const size_t need = ragraph_topk_cosine_filtered_workspace_bytes(B, plan_N, D, k);
  RG_REQUIRE(ws_bytes >= need, RAGRAPH_EWORKSPACE, "topk_cosine_filtered: workspace %zu < %zu", ws_bytes, need);
  if (D == 256)
    return run_filtered<256>(Q, B, Kn, Kp, Kb, N, k, idx_base, out_scores, out_idx, overflow, overflow_idx, ws, ws_bytes, stream,
                             plan_N, theta, exchange, ctx, n_shards);
  if (D == 128)
    return run_filtered<128>(Q, B, Kn, Kp, Kb, N, k, idx_base, out_scores, out_idx, overflow, overflow_idx, ws, ws_bytes, stream,
                             plan_N, theta, exchange, ctx, n_shards);
  return run_filtered<64>(Q, B, Kn, Kp, Kb, N, k, idx_base, out_scores, out_idx, overflow, overflow_idx, ws, ws_bytes, stream,
                          plan_N, theta, exchange, ctx, n_shards);
}

extern "C" int ragraph_topk_cosine_filtered_f32(const float* Q, int64_t B, const float* Kn, const float* Kp,
                                                const uint16_t* Kb, int64_t N, int D, int k, int64_t idx_base,
                                                float* out_scores, int64_t* out_idx, int* overflow,
                                                int64_t* overflow_idx, void* ws, size_t ws_bytes, void* stream) {
  return filtered_entry(Q, B, Kn, Kp, Kb, N, D, k, idx_base, out_scores, out_idx, overflow, overflow_idx, ws, ws_bytes, stream,
                        N, nullptr, nullptr, nullptr, 1);
}

extern "C" int ragraph_topk_cosine_filtered_sharded_f32(const float* Q, int64_t B, const float* Kn, const float* Kp,
                                                        const uint16_t* Kb, int64_t N, int D, int k, int64_t idx_base,
                                                        float* out_scores, int64_t* out_idx, int* overflow,
                                                        int64_t* overflow_idx, void* ws, size_t ws_bytes, void* stream,
                                                        int64_t plan_N, float* theta, ragraph_exchange_fn exchange,
                                                        void* ctx, int n_shards) {
  return filtered_entry(Q, B, Kn, Kp, Kb, N, D, k, idx_base, out_scores, out_idx, overflow, overflow_idx, ws, ws_bytes, stream,
                        plan_N, theta, exchange, ctx, n_shards);
}

extern "C" int ragraph_topk_cosine_filtered_sharded_speculates(int64_t B, int64_t plan_N, int D, int k, int n_shards) {
  if (!filter_dim_ok(D) || B < 1 || plan_N < 1 || k < 1 || k > 32 || n_shards < 1) return 0;
  return filter_schedule(B, plan_N, D, k, n_shards).bound_keys > 0 ? 1 : 0;
}

extern "C" int ragraph_verify_merged_prior_f32(const float* merged_scores, int64_t R, int k, float prior, int speculative,
                                               const int* stats_words, const int* overflow, float* out5, void* stream) {
  RG_REQUIRE(out5 && (merged_scores || R == 0), RAGRAPH_EINVAL, "verify_merged_prior: null pointer");
  RG_REQUIRE(R >= 0 && k >= 1, RAGRAPH_EINVAL, "verify_merged_prior: bad R/k");
  hipLaunchKernelGGL(verify_merged_prior_kernel, dim3(1), dim3(256), 0, as_stream(stream), merged_scores, R, k, prior, speculative,
                     stats_words, overflow, out5);
  RG_CHECK_LAUNCH("verify_merged_prior");
  return RAGRAPH_OK;
}

extern "C" int ragraph_theta_sharpen_f32(const float* gathered, int G, int64_t B, int m, int k, float* theta, void* stream) {
  RG_REQUIRE(gathered && theta, RAGRAPH_EINVAL, "theta_sharpen: null pointer");
  RG_REQUIRE(G >= 1 && m >= 1 && k >= 1 && G * m <= 64 && G * m >= k, RAGRAPH_EINVAL,
             "theta_sharpen: need k <= G*m <= 64 (G=%d m=%d k=%d)", G, m, k);
  if (B <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(theta_sharpen_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, as_stream(stream), gathered, G, B, m, k,
                     theta);
  RG_CHECK_LAUNCH("theta_sharpen");
  return RAGRAPH_OK;
}
