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

#include "filter_copies.h"

#include "filter_prepare.h"

#include "filter_ring.h"

#include "filter_rescore.h"

#include "filter_verify_fixup.h"

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

#include "filter_schedule.h"

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
  static const int64_t prep_wide_b = [] { const char* e = getenv("RAGRAPH_FILTER_PREP_WIDE_B"); return e ? (int64_t)atoll(e) : (int64_t)8192; }();   // A/B
#define RG_PREP(R_)                                                                                                                     \
  hipLaunchKernelGGL((filter_prep_kernel<D, R_>), dim3((unsigned)cdiv(B <= FILTER_QB_MAX_B ? (B + 31) / 32 * 32 : B, 4 * (R_))), dim3(256), 0, st, Q, B, \
                     f.Qn, f.eq, f.count, f.flag, overflow, bound ? f.gmax : nullptr, bound ? filter_bound_parts(k, sc.bound_keys, D, B, exchange ? n_shards : 1) : k, \
                     /* (the bf16 operand image: only launches on the bf16 copy read it -- a call whose levels all run on the int8 */ \
                     /* copy and that has no bound pass, e.g. every call under a prior, saves writing 2 D bytes per query) */           \
                     B <= FILTER_QB_MAX_B && (B <= 256 || bound || sc.nlev > sc.i8_levels) ? f.Qb : nullptr,                           \
                     filter_count_stride(B), sc.i8_levels > 0 ? f.eq8 : nullptr, f.qscale, B <= FILTER_QB_MAX_B ? f.Qb8 : nullptr, f.fix_done, \
                     stats, stats_init, spec ? (exchange ? theta : f.theta) : nullptr, prior)
  if (B >= prep_wide_b) RG_PREP(4);   // (100 000 x 256: 120 us with one row per wave, 74 with two or four, 162 with eight)
  else RG_PREP(1);
#undef RG_PREP
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
  if (G * m <= 32 && B >= 4096)
    hipLaunchKernelGGL(theta_sharpen2_kernel, dim3((unsigned)cdiv(B, 8)), dim3(256), 0, as_stream(stream), gathered, G, B, m, k,
                       theta);
  else
    hipLaunchKernelGGL(theta_sharpen_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, as_stream(stream), gathered, G, B, m, k,
                       theta);
  RG_CHECK_LAUNCH("theta_sharpen");
  return RAGRAPH_OK;
}
