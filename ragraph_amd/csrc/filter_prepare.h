// Per-call preparation and bounds of the filtered top-k: statistics words, filter_prep_kernel, theta / bound-score kernels, theta_sharpen.
// Part of csrc/topk_filter.hip (textually included there, inside its namespace / after its helpers): split out in round 6 so
// that the ring, the candidate path and the launch plumbing can be read -- and changed -- apart.  No include guard on purpose:
// these are not stand-alone headers.

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

// R rows per wave: one wave per query row is 100 000 waves of a microsecond of work at c2 -- the launch is bound by how fast waves
// start, not by its 200 MB; with R = 4 a wave has its four row loads in flight at once and a quarter of the waves exist
// (calls of FILTER_PREP_WIDE_B queries and more; small calls keep one row per wave: they want every CU at once).
template <int D, int R = 1>
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
  constexpr int NCH = D / 4;  // float4 chunks per row: 16 / 32 / 64 -- at most one per lane
  const int64_t q_first = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  float4 vpre[R];
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    vpre[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < NCH && q_first + rr < B) vpre[rr] = reinterpret_cast<const float4*>(Q + (q_first + rr) * D)[lane];
  }
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
  const int64_t q = q_first + rr;
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
  if (q >= (Qb ? (B + 31) / 32 * 32 : B)) continue;
  float4 v = vpre[rr];
  float p = 0.f;
  p = fmaf(v.x, v.x, p);
  p = fmaf(v.y, v.y, p);
  p = fmaf(v.z, v.z, p);
  p = fmaf(v.w, v.w, p);
  p = wave_sum_fixed_tree(p);   // (the bits of six __shfl_xor steps: common.h)
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
  e2 = wave_sum_any_order(e2);
  float e8 = 0.f, sq = 0.f;
  unsigned am = max(max(__float_as_uint(fabsf(v.x)), __float_as_uint(fabsf(v.y))), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w))));
  am = wave_max_u32(am);
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
    e8 = wave_sum_any_order(e8);
  }
  if (q >= B) continue;
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
  }   // rows of this wave
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
      const float x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[w]), o));   // (o: wave-uniform)
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
  for (int o = 0; o < n; ++o) {   // (o is wave-uniform: a v_readlane, not an LDS round trip per step)
    const float u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), o));
    rank += (u > v || (u == v && o < lane)) ? 1 : 0;
  }
  if (lane < n && rank == k - 1) theta[b] = fmaxf(theta[b], v);
}

// The same with TWO queries per wave (G m <= 32: every shard count up to 8 at k = 10): query 2 w in lanes 0-31, 2 w + 1 in lanes
// 32-63, one 32-wide shuffle per step serves both -- the launch is bound by the vector instructions of the counting loop
// (100 000 queries x 32 steps: 42 us), so half the waves are half the time.
__global__ void __launch_bounds__(256) theta_sharpen2_kernel(const float* __restrict__ gathered, int G, int64_t B, int m, int k,
                                                             float* __restrict__ theta) {
  const int lane = threadIdx.x & 63, l32 = lane & 31;
  const int64_t b = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  const int n = G * m;
  const bool live = b < B && l32 < n;
  float v = RG_NEG_INF;
  if (live) v = gathered[((int64_t)(l32 / m) * B + b) * m + (l32 % m)];
  int rank = 0;
  for (int o = 0; o < n; ++o) {
    const float u = __shfl(v, o, 32);
    rank += (u > v || (u == v && o < l32)) ? 1 : 0;
  }
  if (live && rank == k - 1) theta[b] = fmaxf(theta[b], v);
}
