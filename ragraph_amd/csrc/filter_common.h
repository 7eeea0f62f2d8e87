// Shared pieces of the bf16-filtered exact top-k (topk_filter.hip: schedule, ring kernel for many queries, rescoring;
// topk_filter_direct.hip: the register-fed kernel for up to 256 queries).
#pragma once
#include "common.h"
#include <climits>

namespace ragraph {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Error bound of the bf16 scores.  The rounded rows are q + dq and k + dk, so
//   |(q+dq).(k+dk) - q.k| <= |dq||k| + |q||dk| + |dq||dk|      (Cauchy-Schwarz, Euclidean norms)
// bf16 keeps 8 significant bits, i.e. |dq| <= 2^-8 |q| at worst, but the actual |dq| of a query and the largest |dk| of
// the bank are known exactly: ragraph_keys_to_bf16 leaves max_k |dk| behind the bank copy and the query side is
// computed per call.  eps(q) = (|dq| + max|dk| + |dq| max|dk|)(1 + 2^-10) + 2^-12: the factor covers the fp32 rounding
// of the norms and the 1e-7 by which a normalised row's norm can exceed 1, the constant the fp32 accumulation of the
// (exact) bf16 products -- in ANY order: the bank copy is stored in MFMA fragment order and the products of a score are
// summed in whatever order the matrix core takes them -- and the rounding of the exact chain itself (< 2^-14 each).
// Typically eps ~ 0.003, a third of the worst case 2^-7.
constexpr float FILTER_EPS_SLACK = 0.000244140625f;  // 2^-12

constexpr int FILTER_STAGE_BYTES = 32 * 1024;  // one ring slot of the ring kernel
constexpr int FILTER_PAD_KEYS = 256;           // bank rows are padded to whole stages of any D (256 keys at D = 64)

// Layout of the bf16 bank copy (ragraph_keys_to_bf16): MFMA FRAGMENT ORDER of v_mfma_f32_16x16x32_bf16 (keys = the A
// operand's 16 rows, 32 elements per k-step).  Keys are taken in sub-tiles of 32 = two halves of 16; a sub-tile is
// D/32 k-steps x 2 halves = D/16 blocks of 1 KiB; block 2 t + h of sub-tile u (byte ((u * D/16) + 2 t + h) * 1024) is the
// A operand of k-step t for half h: lane l = j + 16 g (j < 16, g < 4) owns 16 bytes at 16 l -- elements 32 t + 8 g .. + 7
// of key 32 u + 16 h + j.  A wave therefore fetches an operand with ONE fully coalesced 1-KiB load
// (global_load_dwordx4, or global_load_lds_dwordx4 into a ring slot whose image needs no swizzle: consecutive lanes read
// consecutive 16-B pieces), and no kernel transposes anything.  2 D bytes per key, as a row-major copy has.  The blocks
// of a sub-tile are k-step major (both halves of step 0, then of step 1, ...): the order the kernels consume them in.
// Why this shape and not 32x32x16 (round 1): on random operands the chip holds a higher clock under it -- 1.71 vs 1.61
// PFLOP/s sustained in the filter's own inner loop (tools/microbench/mfma_bf16_shape_bench.hip) -- at the same LDS
// traffic per flop (one 1-KiB fragment read feeds 64 MFMA cycles either way).
// The result of one MFMA: lane j + 16 g holds scores of query column j against keys 4 g + r (r < 4) of the half.
__host__ __device__ constexpr int64_t filter_block_offset(int64_t subtile, int D, int t, int h) {
  return (subtile * (D / 16) + 2 * t + h) * 1024;
}

// The INT8 copy behind the bf16 copy (ragraph_keys_to_bf16 makes both).  v_mfma_i32_16x16x64_i8 issues at the cycles of the
// bf16 MFMA with twice the K: in the filter's own inner loop it sustains 3.29 - 3.53 Pop/s where bf16 sustains 1.68 PFLOP/s
// (tools/microbench/mfma_i8_bench.hip) -- twice the (query, key) pairs per second -- and a key is 1 byte per element.
//   * keys: one scale per CLASS of granules (two classes: "TWO SCALES" below; s_k stands for the class's scale and max |dk|
//     for the class's largest error), s_k = max |k_i| / 127 over the class (a per-key scale would have to be applied to every
//     score before the threshold test; the bound below only depends on the LARGEST key error anyway), ki = rint(k / s_k);
//   * queries: a scale per query, s_q = max |q_i| / 127, qi = rint(q / s_q) (a query's scale folds into its threshold);
//   * the approximate score is s~ = s_q s_k I with I = sum qi ki an EXACT integer (|I| <= 127^2 D < 2^23), and with
//     q^ = s_q qi = q + dq, k^ = s_k ki = k + dk:  |s~ - s| <= |q^||dk| + |dq||k| <= |dq| + max|dk| + |dq| max|dk| (+ the
//     2e-7 by which a normalised row's norm can exceed 1): filter_eps' formula with the int8 errors, ~0.02 on Gaussian
//     rows where bf16 gives 0.004.  A key of the exact top-k has I >= (theta - eps) / (s_q s_k): the test runs on the
//     integers, against T = floor of that quotient - 2 (the quotient's fp32 rounding is < 1 unit at |I| < 2^23).
//   A wider eps lets ~3x as many keys through (150 instead of 50 per query at the last level of the bench shape), so int8
//   serves the LATE levels of large batches, where the threshold is high and the matrix work is what costs.
// Layout: the geometry of the bf16 copy at half the row bytes -- sub-tiles of 32 keys = D/64 k-steps x 2 halves = D/32
// blocks of 1 KiB; block 2 t + h of sub-tile u: lane j + 16 g owns the 16 bytes holding elements 64 t + 16 g .. + 15 of
// key 32 u + 16 h + j (A operand of v_mfma_i32_16x16x64_i8; any fixed assignment of the 64 elements to the four lane
// groups is right as long as the query operand uses the same one: the integer sum does not depend on the order).
// TWO SCALES (round 5).  One scale for the whole bank means one heavy-tailed row (a one-hot key: |k_i| = 1) puts every
// ordinary row on a grid three times too coarse, and even a Gaussian bank pays for its single largest entry.  So the copy
// is cut into GRANULES of one ring stage (32 KiB of int8 rows: 128 / 256 / 512 keys at D = 256 / 128 / 64) and every
// granule belongs to one of two classes: NORMAL (its largest |k_i| <= cut: quantised with s_N = cut / 127) or HEAVY (the
// rest: s_H = max |k_i| / 127, the old scale).  The cut minimises a model of the candidates the two bounds admit
// (i8_cut_kernel).  Each class has its own largest |dk| and hence its own eps and integer threshold per query; a kernel
// picks the threshold by the class BIT of the granule it is in (one bit per granule behind the tail row) -- a select per
// stage / unit, nothing per score.  Scored candidate lists carry the class in bit 0 of their integer ((I << 1) | class).
// Tail of the int8 copy (one row): [0] max |dk|^2 over the NORMAL granules' keys (float bits), [1] s_N (float), [2] max
// |k_i| of the bank (float bits, >= 0), [3] max |dk|^2 over the HEAVY granules' keys, [4] s_H, [5] the cut (float bits),
// [6] heavy granules, [7] granules.  Behind the tail row: the granules' largest |k_i| (floats), then the class bits.
__host__ __device__ constexpr int64_t filter_i8_block_offset(int64_t subtile, int D, int t, int h) {
  return (subtile * (D / 32) + 2 * t + h) * 1024;
}
__host__ __device__ constexpr int filter_i8_granule_keys(int D) { return FILTER_STAGE_BYTES / D; }
__host__ __device__ constexpr int64_t filter_pad_keys(int64_t n) { return (n + FILTER_PAD_KEYS - 1) / FILTER_PAD_KEYS * FILTER_PAD_KEYS; }
// The pieces of a bank copy (ragraph_keys_to_bf16's buffer) that belong to the int8 image
struct FilterI8View {
  const signed char* K8;   // the image: filter_pad_keys(N) rows of D bytes, fragment order
  const unsigned* tail8;   // the tail row (above)
  const float* gmax;       // [granules] largest |k_i| of each granule
  const unsigned* cls;     // [ceil(granules / 32)] class bits: bit (g & 31) of word g >> 5 set = granule g is HEAVY
  int64_t granules;
};
inline int64_t filter_i8_granules(int64_t N, int D) {
  const int gk = filter_i8_granule_keys(D);
  return (filter_pad_keys(N) + gk - 1) / gk;
}
inline size_t filter_i8_table_bytes(int64_t N, int D) {  // gmax floats (16-B aligned) + class words + one spare word
  const int64_t g = filter_i8_granules(N, D);
  return (size_t)((g * 4 + 15) / 16 * 16) + (size_t)((g + 31) / 32 * 4) + 16;
}
inline FilterI8View filter_i8_view(const uint16_t* Kb, int64_t N, int D) {
  const int64_t npad = filter_pad_keys(N);
  FilterI8View v;
  v.K8 = reinterpret_cast<const signed char*>(Kb + (npad + 1) * D);
  v.tail8 = reinterpret_cast<const unsigned*>(v.K8 + npad * D);
  v.granules = filter_i8_granules(N, D);
  v.gmax = reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.tail8) + (size_t)D * 2);
  v.cls = reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(v.gmax) + (v.granules * 4 + 15) / 16 * 16);
  return v;
}

// rint(x * inv_scale) clamped to [-127, 127], inv_scale = 1 / scale computed ONCE per row with one fp32 division: a multiply
// per element (the ring kernel quantises 256 elements per lane and segment: divisions were 5 us of every segment).  Whatever
// integer comes out, the bound uses the error of THAT integer (|s q - x| is measured, not assumed), and the prepare launch and
// the kernels call this one function with the same inv_scale, so they produce the same operands.
__device__ __forceinline__ int quantize_i8(float x, float inv_scale) {
  return (int)fminf(fmaxf(__builtin_rintf(__fmul_rn(x, inv_scale)), -127.f), 127.f);
}

// This thread's cap on int8 levels (ragraph_topk_cosine_filtered_max_i8_levels; -1 = the library's rule, 0 = none).
int filter_thread_i8_cap();
// This thread's speculative first bound (ragraph_topk_cosine_filtered_set_prior; NaN = none).
float filter_thread_prior();
// topk_overflow_fixup_kernel (topk_filter.hip) for another caller's list of queries: exact fp32 scans of the bank, cut into
// key slices, for the *count queries listed (a launch that returns at once when the list is empty).  done: >= 1024 zeroed
// ints; part_s / part_i: 1024 x 16 x 32 floats / int64s.
int launch_overflow_fixup(int D, const float* Qn, const float* Kn, int64_t N, int k, int64_t idx_base, const int* count,
                          const int* list, float* out_s, int64_t* out_i, int* done, float* part_s, int64_t* part_i, int64_t B,
                          void* stream);

// float <-> int with the same order (for atomicMax on scores of either sign)
__device__ __forceinline__ int f2ord(float f) {
  const int b = __float_as_int(f);
  return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord2f(int o) { return __int_as_float(o >= 0 ? o : o ^ 0x7FFFFFFF); }

// How a filter launch obtains a query's pass threshold thr[q] = theta[q] - eps(q), where theta[q] is a proven lower
// bound of the query's final k-th best exact score: the k-th exact score of everything rescored so far (prev_scores),
// or -- gmax != NULL, right after the bound pass of a handful of queries -- the smallest of the k part maxima of approximate
// scores minus eps(q) (each part's best key has an exact score >= its approximate one - eps, so k distinct keys score at
// least that); larger batches record 4 k parts and pass theta = the k-th largest of them (filter_bound_scores_kernel).
struct FilterThr {
  const float* theta;         // [B] the bound itself, handed in (sharded banks: sharpened across the shards), or NULL
  const float* prev_scores;   // [B,k] running exact top-k (descending), or NULL
  const int* gmax;            // [B,ngroups] order-preserving ints of the bound pass, or NULL
  const float* eq;            // [B] |dq| of the query's bf16 rounding
  const unsigned* max_kerr2;  // bank: max_k |dk|^2 as float bits
  int k, ngroups;
  int ablate;                 // timing only: nothing passes
  // int8 levels: [B] |dq| of the query's int8 rounding, [B] its scale s_q, the int8 copy's tail (max |dk|^2, s_k)
  const float* eq8;
  const float* qscale;
  const unsigned* tail8;
  const unsigned* cls8;       // the int8 copy's class bits (FilterI8View::cls)
  const unsigned char* flag;  // [B] non-zero: the query's final answer comes from the exact scan anyway (a list of an
                              // earlier level overflowed, or it is a ZERO query -- every score +0, every key within any
                              // bound: filter_prep_kernel marks it) -- nothing passes the filter for it
};

__device__ __forceinline__ float filter_eps(const FilterThr& t, int64_t q) {
  const float ek = sqrtf(__uint_as_float(*t.max_kerr2));
  const float e = t.eq[q];
  return fmaf(fmaf(e, ek, e + ek), 1.0009765625f, FILTER_EPS_SLACK);  // rounding direction is inside the factor
}

// theta[q]: the proven lower bound of the query's final k-th best exact score a level filters with
__device__ __forceinline__ float filter_theta(const FilterThr& t, int64_t q) {
  if (t.theta) return t.theta[q];
  if (t.gmax) {
    int m = t.gmax[q * t.ngroups];
    for (int g = 1; g < t.ngroups; ++g) m = min(m, t.gmax[q * t.ngroups + g]);
    return __fsub_rn(ord2f(m), filter_eps(t, q));   // (the bound pass ran on the bf16 copy)
  }
  return t.prev_scores[q * t.k + t.k - 1];
}

__device__ __forceinline__ float filter_threshold(const FilterThr& t, int64_t q) {
  const unsigned char fl = t.flag ? t.flag[q] : 0;  // (loaded next to the bound's operands: one latency, not two)
  const float thr = __fsub_rn(filter_theta(t, q), filter_eps(t, q));
  return (t.ablate == 1 || fl) ? __builtin_huge_valf() : thr;
}

// The int8 levels' integer threshold: a key can only belong to the exact top-k if I = sum qi ki >= the result (see the
// layout comment above).  INT_MIN: everything passes (a zero BANK, whose scale is 0; zero queries are flagged and pass nothing).
// (_at: for a given lower bound theta of the query's exact k-th best score)
// (heavy: for keys of a HEAVY granule -- that class's error and scale, tail words 3 / 4)
__device__ __forceinline__ int filter_threshold_i8_of(float theta, float e, float ek, float sc) {
  const float eps = fmaf(fmaf(e, ek, e + ek), 1.0009765625f, FILTER_EPS_SLACK);
  if (!(sc > 0.f)) return INT_MIN;
  const float x = __fsub_rn(theta, eps) / sc;
  if (!(x > -8.4e6f)) return INT_MIN;   // (also NaN)
  if (x > 8.4e6f) return INT_MAX;       // beyond any |I| <= 127^2 * 256: nothing can pass
  return (int)floorf(x) - 2;
}
__device__ __forceinline__ int filter_threshold_i8_at(const FilterThr& t, int64_t q, float theta, int heavy) {
  return filter_threshold_i8_of(theta, t.eq8[q], sqrtf(__uint_as_float(t.tail8[heavy ? 3 : 0])),
                                t.qscale[q] * __uint_as_float(t.tail8[heavy ? 4 : 1]));
}
__device__ __forceinline__ int filter_threshold_i8(const FilterThr& t, int64_t q, int heavy) {
  const unsigned char fl = t.flag ? t.flag[q] : 0;
  const int thr = filter_threshold_i8_at(t, q, filter_theta(t, q), heavy);
  return (t.ablate == 1 || fl) ? INT_MAX : thr;
}
// the class of the granule a key lies in, from the class words (any thread; a cached 4-byte read)
__device__ __forceinline__ int filter_i8_class(const unsigned* cls, int64_t granule) {
  return (int)((cls[granule >> 5] >> (granule & 31)) & 1u);
}

// Candidate counters of a call of fewer than 2048 queries.  A returning atomicAdd costs ~11 ns per operation on ONE address
// and ~1.8 ns on one 128-byte line chip-wide (tools/microbench/atomic_contention_bench.hip): 16 queries x 800
// candidates on sixteen neighbouring ints were 25 us of serialised atomics behind a 75 us stream.  So every query's
// counters sit on their own line (FILTER_COUNT_STRIDE ints apart), and a query of a very small batch -- whose list the
// sliced rescoring cuts into S parts anyway -- gets S sub-lists with a counter each, the waves spread over them.
constexpr int FILTER_COUNT_STRIDE = 32;
constexpr int FILTER_TICKET_SLOT = 16;  // int of a query's line that counts its finished rescoring workgroups
__host__ __device__ constexpr int filter_count_stride(int64_t B) { return B < 2048 ? FILTER_COUNT_STRIDE : 1; }

// One filter launch of the direct kernel (topk_filter_direct.hip): up to 256 queries against keys [key0, key1) of the
// bf16 copy (key0 a multiple of 32).  bound_groups > 0: the BOUND pass -- no thresholds, no candidates, the launch
// records per query the best approximate score of each of bound_groups consecutive parts of the range into gmax_out.
struct DirectArgs {
  const uint16_t* Qb;     // the normalised queries as bf16 B operands in fragment order (filter_prep_kernel):
                          // block (qg * D/32 + t) = k-step t of queries 16 qg .. 16 qg + 15, lane j + 16 g of it
                          // = elements 32 t + 8 g .. + 7 of query 16 qg + j; padded to whole groups of 32 queries, zeros
  const uint16_t* Kb;     // fragment-order bf16 keys
  int64_t B, key0, key1;
  FilterThr thr;          // (filter launches)
  int* count;             // [B][FILTER_COUNT_STRIDE] candidate slots reserved so far, per sub-list
  int* cand;              // [B,cap] candidate key indices (local to this shard): nsub sub-lists of cap / nsub slots
  int cap;
  int nsub;               // sub-lists per query (a power of two <= FILTER_COUNT_STRIDE; wave w appends to w % nsub)
  int* gmax_out;          // (bound pass) [B,bound_groups]
  int bound_groups;
  int scored;             // (i8) the lists hold {key, I} pairs in int2 slots, I = an upper bound of the integer sum that
                          // admitted the key (the lane's largest, rounded up to a multiple of 256): topk_filter.hip,
                          // topk_rescore_scored_kernel
  int i8;                 // the pass runs on the int8 copy: Kb = that copy, Qb = the queries' int8 image (Qb8 of the prepare
                          // launch: block (qg * D/64 + t), lane j + 16 g = elements 64 t + 16 g .. + 15 of query 16 qg + j)
};
template <int D>
int launch_filter_direct(const DirectArgs& a, hipStream_t st);

}  // namespace ragraph
