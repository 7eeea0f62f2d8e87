// Exact cosine top-k in ONE launch for SMALL banks  (SimilarityFunctions.py:6-16 + ToyGraphBase.py:66-67 at the reference's
// own operating point: BASELINE config 1's 10 000 x 128 toy bank against the 2708 nodes of a Cora-sized forward, a few
// hundred to a few thousand queries against a bank whose bf16 copy is a few MB).
//
// The filtered path of topk_filter.hip is six dependent launches (prepare, bound pass, bound selection, filter level,
// rescoring, overflow check): 5 - 25 us each whatever the size, ~90 us for 7 GFLOP of scores.  Its phases only need to
// agree on a query's bound, and when ONE workgroup sees the whole bank for its queries nothing has to cross workgroups:
//   workgroup = 32 queries (two groups of 16 = the B operands of v_mfma_f32_16x16x32_bf16, in registers) x 8 waves, the
//   waves taking 16-KiB units of the bank copy (fragment order, filter_common.h) round-robin from the L2;
//   phase 0  normalise the 32 queries (normalize_rows' tree: same bits), |dq| of their bf16 rounding, B operands;
//   phase A  bound: best approximate score of each of G = 4 k parts of a prefix (N/8 of the bank; N/4 for k > 8) in LDS,
//            theta[q] = k-th largest of the G part maxima - eps(q)           (the proof of topk_filter.hip, step 1);
//   phase B  filter: every key with s~ >= theta - eps goes to the query's candidate list in LDS (step 2);
//   phase C  exact fp32 scores of all the tile's candidates (the k = 0..D-1 fmaf chain from +0), 64 per batch, batches
//            dealt over the waves, every lane's row in flight at once; canonical selection (rescore_common.h), a wave
//            per query; a list that overflows (near-duplicate banks) is answered by an exact scan of the bank.
// A workgroup reads from L2 at ~80 GB/s (measured: 2.4 MiB in 31 us), so the bank is SPLIT over S workgroups per tile
// (S tiles >= the chip): each runs phases 0 and A for the tile (the same bound: the prefix is small), filters and rescoring
// its units, and leaves its exact top-k of every query in a small workspace; the tile's last workgroup to finish (a
// ticket) merges the S lists.  Still one launch, nothing crosses tiles.
// Same bits as every other path (tests/test_gpu_kernels.py: against the oracle and the fp32 kernels).
#include "rescore_common.h"
#include <type_traits>

namespace ragraph {

template <int D_>
struct FusedCfg {
  static constexpr int D = D_;
  static constexpr int WAVES = 8, THREADS = 512, QT = 32;
  static constexpr int KSTEPS = D / 16;              // 1-KiB blocks per 32-key sub-tile
  static constexpr int KS32 = D / 32;                // MFMA k-steps per sub-tile
  static constexpr int UNIT_BLOCKS = 16;
  static constexpr int SUBS = UNIT_BLOCKS / KSTEPS;  // sub-tiles per 16-KiB unit: 1 / 2 / 4
  static constexpr int UNIT_KEYS = 32 * SUBS;
  static constexpr int CAP = 256;                    // candidate slots per query and workgroup (expected: ~100 / splits)
  static constexpr int GMAX = 64;                    // parts of the bound's prefix (4 k, k <= 16)
  static constexpr int MAXB = QT * (CAP / 64);       // 64-candidate batches of a tile
  static constexpr size_t QN_BYTES = (size_t)QT * D * 4, QB_BYTES = (size_t)QT * D * 2;
  // qn | qb | partmax [QT][GMAX] | cand [QT][CAP] | sc [QT][CAP] | cnt, thr, epsq [QT] | batch table [MAXB] | staging
  static constexpr size_t LDS_BYTES = QN_BYTES + QB_BYTES + (size_t)QT * GMAX * 4 + 2 * (size_t)QT * CAP * 4 + 3 * QT * 4 +
                                      MAXB * 4 + (size_t)WAVES * 32 * 12 + 16;
};

struct FusedParams {
  const float* Q;        // [B,D] raw queries
  const float* Kn;       // [N,D] normalised keys (fp32: the exact rescoring)
  const uint16_t* Kb;    // bf16 copy in fragment order, padded, + the max |dk|^2 word (ragraph_keys_to_bf16)
  int64_t B, N;
  int k, G, psubs;       // parts and 32-key sub-tiles of the bound's prefix (32 psubs <= N)
  int64_t nunits;        // 16-KiB units that hold keys < N
  int64_t idx_base;
  float* out_s;
  int64_t* out_i;
  // key splits: S workgroups per query tile, workgroup s taking units s, s + S, ...; each leaves its exact top-k of every
  // query (local ids) in part_*, the tile's last workgroup to finish (ticket) merges the S lists
  int S;
  int* tickets;          // [tiles] zero before the first call; every call leaves them zero
  float* part_s;         // [tiles][S][QT][k]
  int* part_i;
  int* part_over;        // [tiles][S][QT] this split's candidate list of the query overflowed
};

#ifdef RG_FUSED_TIMING  // diagnostic build: wall-clock stamps (10 ns ticks) of workgroup 0 / wave 0 and of the last block
__device__ unsigned long long g_fused_t[2][16];
#define RG_FSTAMP(i_)                                                                                     \
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) g_fused_t[blockIdx.x != 0][i_] = wall_clock64()
#else
#define RG_FSTAMP(i_)
#endif

template <int D>
__global__ void __launch_bounds__(512, 2) topk_fused_kernel(FusedParams p) {
  using C = FusedCfg<D>;
  RG_FSTAMP(0);
  extern __shared__ float4 fused_smem4[];
  char* smem = reinterpret_cast<char*>(fused_smem4);
  float* qn = reinterpret_cast<float*>(smem);                                    // [QT][D]
  char* qb = smem + C::QN_BYTES;                                                 // [QT/16][KS32][64] x 16 B
  int* partmax = reinterpret_cast<int*>(qb + C::QB_BYTES);                       // [QT][GMAX]
  int* cand = partmax + C::QT * C::GMAX;                                         // [QT][CAP]
  float* sc = reinterpret_cast<float*>(cand + C::QT * C::CAP);                   // [QT][CAP] exact scores
  int* cnt = reinterpret_cast<int*>(sc + C::QT * C::CAP);                        // [QT]
  float* thr = reinterpret_cast<float*>(cnt + C::QT);                            // [QT]
  float* epsq = thr + C::QT;                                                     // [QT]
  int* btab = reinterpret_cast<int*>(epsq + C::QT);                              // [MAXB] batch = ql | first slot << 8
  float* stage_s = reinterpret_cast<float*>(btab + C::MAXB);                     // [WAVES][32]
  int64_t* stage_i = reinterpret_cast<int64_t*>(stage_s + C::WAVES * 32);        // [WAVES][32]
  int* misc = reinterpret_cast<int*>(stage_i + C::WAVES * 32);                   // [0] batches, [1] last workgroup of the tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int S = p.S;
  const int64_t tile = blockIdx.x / S;
  const int split = (int)(blockIdx.x % S);
  const int64_t q0 = tile * C::QT;
  const unsigned* max_kerr2 = reinterpret_cast<const unsigned*>(p.Kb + ((p.N + FILTER_PAD_KEYS - 1) / FILTER_PAD_KEYS) * FILTER_PAD_KEYS * D);

  // ---- phase 0: this wave's four queries (filter_prep_kernel's arithmetic, into LDS) ---------------------------------------
  {
    constexpr int NCH = D / 4;
    const float ek = sqrtf(__uint_as_float(*max_kerr2));
    float4 vq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // the four rows' loads in flight together
      const int64_t q = q0 + 4 * wave + i;
      vq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane < NCH && q < p.B) vq[i] = reinterpret_cast<const float4*>(p.Q + q * D)[lane];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ql = 4 * wave + i;
      float4 v = vq[i];
      float pn = 0.f;
      pn = fmaf(v.x, v.x, pn);
      pn = fmaf(v.y, v.y, pn);
      pn = fmaf(v.z, v.z, pn);
      pn = fmaf(v.w, v.w, pn);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) pn = __fadd_rn(pn, __shfl_xor(pn, off));
      const float d = fmaxf(sqrtf(pn), 1e-12f);
      v.x = v.x / d; v.y = v.y / d; v.z = v.z / d; v.w = v.w / d;
      float e2 = 0.f;
      if (lane < NCH) {
        reinterpret_cast<float4*>(qn + ql * D)[lane] = v;
        const int e0 = 4 * lane, t = e0 >> 5, gg = (e0 >> 3) & 3;
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(qb + ((ql >> 4) * C::KS32 + t) * 1024 + (gg * 16 + (ql & 15)) * 16 + (e0 & 7) * 2) = o;
        const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dd = x[e] - (float)(__bf16)x[e];
          e2 = fmaf(dd, dd, e2);
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) e2 += __shfl_xor(e2, off);
      const float e = sqrtf(e2) * 1.0000002f;
      if (lane == 0) {
        epsq[ql] = fmaf(fmaf(e, ek, e + ek), 1.0009765625f, FILTER_EPS_SLACK);   // filter_eps
        cnt[ql] = 0;
      }
      partmax[ql * C::GMAX + lane] = f2ord(RG_NEG_INF);
    }
  }
  __syncthreads();
  RG_FSTAMP(1);
  bf16x8 bq[2 * C::KS32];  // [group][k-step]
#pragma unroll
  for (int t = 0; t < 2 * C::KS32; ++t) bq[t] = reinterpret_cast<const bf16x8*>(qb)[t * 64 + lane];

  f32x4 A0[16], A1[16];
#define RG_ULOAD(buf_, u_)                                                                                         \
  {                                                                                                                \
    const char* ub_ = reinterpret_cast<const char*>(p.Kb) + (uint64_t)(u_) * (C::UNIT_BLOCKS * 1024) + lane * 16;  \
    _Pragma("unroll") for (int b_ = 0; b_ < 16; ++b_) buf_[b_] = *reinterpret_cast<const f32x4*>(ub_ + b_ * 1024); \
  }
  float my_thr[2] = {0.f, 0.f};  // (phase B) the thresholds of this lane's two query columns
  // one unit against the two query groups; BOUND_: part maxima of the prefix, else candidates
  auto process = [&](f32x4 (&A)[16], int64_t unit, auto bound_tag) {
    constexpr bool BOUND_ = decltype(bound_tag)::value;
#pragma unroll
    for (int sub = 0; sub < C::SUBS; ++sub) {
      const int64_t st = unit * C::SUBS + sub;  // sub-tile index over the bank
      if (BOUND_ && st >= p.psubs) break;       // (wave-uniform)
      f32x4 acc[2][2];
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) acc[gq][0] = acc[gq][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < C::KS32; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bf16x8 a_ = __builtin_bit_cast(bf16x8, A[(sub * C::KS32 + t) * 2 + h]);
#pragma unroll
          for (int gq = 0; gq < 2; ++gq)
            acc[gq][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, bq[gq * C::KS32 + t], acc[gq][h], 0, 0, 0);
        }
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        float m = acc[gq][0][0];
#pragma unroll
        for (int r = 1; r < 4; ++r) m = fmaxf(m, acc[gq][0][r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[gq][1][r]);
        if constexpr (BOUND_) {
          m = fmaxf(m, __shfl_xor(m, 16));
          m = fmaxf(m, __shfl_xor(m, 32));
          const int part = (int)(st * p.G / p.psubs);
          if (g == 0) atomicMax(partmax + (16 * gq + j) * C::GMAX + part, f2ord(m));
        } else {
          const float th = my_thr[gq];
          if (__any(m >= th)) {
            unsigned mk = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int r = 0; r < 4; ++r) mk |= (acc[gq][h][r] >= th) ? (1u << (4 * h + r)) : 0u;
            const int64_t key_base = st * 32 + 4 * g;  // the lane's keys: + r + 16 h  (mask bit 4 h + r)
            const int ql = 16 * gq + j;
            while (mk) {
              const int r = __ffs(mk) - 1;
              mk &= mk - 1;
              const int64_t key = key_base + (r & 3) + 16 * (r >> 2);
              if (key < p.N) {
                const int slot = atomicAdd(cnt + ql, 1);
                if (slot < C::CAP) cand[ql * C::CAP + slot] = (int)key;
              }
            }
          }
        }
      }
    }
  };

  // ---- phase A: the bound from the prefix's part maxima (every split of a tile computes the same one) --------------------
  {
    const int64_t punits = (p.psubs + C::SUBS - 1) / C::SUBS;
    const int64_t n_mine = wave < punits ? (punits - wave + C::WAVES - 1) / C::WAVES : 0;
    if (n_mine > 0) {
      RG_ULOAD(A0, wave);
      int64_t i = 0;
      for (; i + 2 <= n_mine; i += 2) {
        RG_ULOAD(A1, wave + (i + 1) * C::WAVES);
        process(A0, wave + i * C::WAVES, std::true_type{});
        if (i + 2 < n_mine) RG_ULOAD(A0, wave + (i + 2) * C::WAVES);
        process(A1, wave + (i + 1) * C::WAVES, std::true_type{});
      }
      if (i < n_mine) process(A0, wave + i * C::WAVES, std::true_type{});
    }
  }
  // this wave's first filter unit travels while the bound is selected
  const int64_t ustride = (int64_t)S * C::WAVES;
  const int64_t ufirst = split + (int64_t)S * wave;
  const int64_t n_mine = ufirst < p.nunits ? (p.nunits - ufirst + ustride - 1) / ustride : 0;
  if (n_mine > 0) RG_ULOAD(A0, ufirst);
  __syncthreads();
  RG_FSTAMP(2);
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {  // theta = k-th largest of the G part maxima - eps: lane l holds part l, ranked by counting
    const int ql = 4 * wave + i;
    const float eps = epsq[ql];
    const float v = lane < p.G ? __fsub_rn(ord2f(partmax[ql * C::GMAX + lane]), eps) : RG_NEG_INF;
    int rank = 0;
    for (int o = 0; o < p.G; ++o) {  // (o is wave-uniform: v_readlane, not a ds_bpermute round trip per part)
      const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), o));
      rank += (x > v || (x == v && o < lane)) ? 1 : 0;
    }
    // (the padding queries of the last tile -- zero rows, every score 0 -- must not pass anything)
    if (lane < p.G && rank == p.k - 1) thr[ql] = q0 + ql < p.B ? __fsub_rn(v, eps) : __builtin_huge_valf();
  }
  __syncthreads();
  my_thr[0] = thr[j];
  my_thr[1] = thr[16 + j];
  RG_FSTAMP(3);

  // ---- phase B: the filter over this split's units of the bank, double-buffered ------------------------------------------------
#if defined(RG_FUSED_ABL) && (RG_FUSED_ABL & 2)  // timing build: no filter pass (results invalid)
  if (false)
#endif
  if (n_mine > 0) {
    int64_t i = 0;
    for (; i + 2 <= n_mine; i += 2) {
      RG_ULOAD(A1, ufirst + (i + 1) * ustride);
      process(A0, ufirst + i * ustride, std::false_type{});
      if (i + 2 < n_mine) RG_ULOAD(A0, ufirst + (i + 2) * ustride);
      process(A1, ufirst + (i + 1) * ustride, std::false_type{});
    }
    if (i < n_mine) process(A0, ufirst + i * ustride, std::false_type{});
  }
#undef RG_ULOAD
  RG_FSTAMP(4);
  __syncthreads();
  RG_FSTAMP(5);
#if defined(RG_FUSED_ABL) && (RG_FUSED_ABL & 1)  // timing build: no rescoring (results invalid)
  if (cnt[0] == 123456789) p.out_s[0] = 1.f;
  return;
#endif

  // ---- phase C: exact fp32 scores of all candidates of the tile, 64 per batch, the batches dealt over the waves -------------
  if (wave == 0) {  // batch table: query ql has ceil(min(cnt, CAP) / 64) batches
    const int n = lane < C::QT ? min(cnt[lane], C::CAP) : 0;
    const int nb = (n + 63) >> 6;
    int off = nb;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int o = __shfl_up(off, d);
      if (lane >= d) off += o;
    }
    for (int b = 0; b < nb; ++b) btab[off - nb + b] = lane | (b << 8);
    if (lane == C::QT - 1) misc[0] = off;
  }
  __syncthreads();
  RG_FSTAMP(6);
  {
    const int nbatch = misc[0];
    for (int bi = wave; bi < nbatch; bi += C::WAVES) {
      const int e = btab[bi];
      const int ql = e & 255, c = 64 * (e >> 8) + lane;
      const int n = min(cnt[ql], C::CAP);
      if (c < n) {
        const int key = cand[ql * C::CAP + c];
        const float4* kr = reinterpret_cast<const float4*>(p.Kn + (int64_t)key * D);
        const float4* qrow = reinterpret_cast<const float4*>(qn + ql * D);
        float acc = 0.f;
#pragma unroll
        for (int d0 = 0; d0 < D / 4; d0 += 32) {  // 32 float4 (512 B of the row) in flight per lane
          float4 kv[32];
#pragma unroll
          for (int d4 = 0; d4 < 32; ++d4)
            if (d0 + d4 < D / 4) kv[d4] = kr[d0 + d4];
#pragma unroll
          for (int d4 = 0; d4 < 32; ++d4)
            if (d0 + d4 < D / 4) {
              const float4 qv = qrow[d0 + d4];
              acc = fmaf(qv.x, kv[d4].x, acc);
              acc = fmaf(qv.y, kv[d4].y, acc);
              acc = fmaf(qv.z, kv[d4].z, acc);
              acc = fmaf(qv.w, kv[d4].w, acc);
            }
        }
        sc[ql * C::CAP + c] = acc;
      }
    }
  }
  RG_FSTAMP(7);
  __syncthreads();
  RG_FSTAMP(8);

  // ---- selection: a wave per query; S = 1 writes the result, else this split's top-k (local ids) for the merge ---------------
  float* st_s = stage_s + wave * 32;
  int64_t* st_i = stage_i + wave * 32;
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    const int ql = 4 * wave + i;
    const int64_t q = q0 + ql;
    if (q >= p.B) break;  // (wave-uniform)
    const int n = cnt[ql];
    const bool over = n > C::CAP;
    float* os = p.out_s + q * p.k;
    int64_t* oi = p.out_i + q * p.k;
    if (S == 1 && over) {
      exact_scan_wave<D>(reinterpret_cast<const float4*>(qn + ql * D), p.Kn, p.N, p.k, p.idx_base, lane, os, oi);
      continue;
    }
    float s4[4];
    int id4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = lane + 64 * u;
      const bool have = c < n && c < C::CAP;
      s4[u] = have ? sc[ql * C::CAP + c] : RG_NEG_INF;
      id4[u] = have ? cand[ql * C::CAP + c] : INT_MAX;
    }
    // (k <= 16 rounds of four DPP steps: cheaper here than wave_select<2>'s 128 broadcasts)
    if (S == 1) {
      wave_select<4>(s4, id4, p.k, lane, p.idx_base, os, oi);
    } else {
      wave_select<4>(s4, id4, p.k, lane, 0, st_s, st_i);
      __builtin_amdgcn_wave_barrier();
      const int64_t slot = ((tile * S + split) * C::QT + ql) * p.k;
      if (lane < p.k) {
        p.part_s[slot + lane] = st_s[lane];
        p.part_i[slot + lane] = st_i[lane] >= INT_MAX ? INT_MAX : (int)st_i[lane];
      }
      if (lane == 0) p.part_over[(tile * S + split) * C::QT + ql] = over ? 1 : 0;
      __builtin_amdgcn_wave_barrier();
    }
  }
  RG_FSTAMP(9);
  if (S == 1) return;

  // ---- the tile's LAST workgroup merges the S splits' lists (the ticket pattern of topk_rescore_wide_kernel) -----------------
  __syncthreads();
  if (tid == 0) {
    __threadfence();  // this split's lists are visible device-wide before its ticket
    misc[1] = atomicAdd(p.tickets + tile, 1) == S - 1;
  }
  __syncthreads();
  RG_FSTAMP(10);
  if (!misc[1]) return;
  __threadfence();
  RG_FSTAMP(11);
  if (tid == 0) p.tickets[tile] = 0;  // (every split has arrived: the next call finds zero)
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    const int ql = 4 * wave + i;
    const int64_t q = q0 + ql;
    if (q >= p.B) break;
    float* os = p.out_s + q * p.k;
    int64_t* oi = p.out_i + q * p.k;
    int over = 0;
    if (lane < S)
      over = __hip_atomic_load(p.part_over + (tile * S + lane) * C::QT + ql, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__any(over != 0)) {  // some split dropped candidates of this query: the exact scan answers it
      exact_scan_wave<D>(reinterpret_cast<const float4*>(qn + ql * D), p.Kn, p.N, p.k, p.idx_base, lane, os, oi);
      continue;
    }
    float s1[1] = {RG_NEG_INF};
    int id1[1] = {INT_MAX};
    if (lane < S * p.k) {  // (the other workgroups' stores: read past this CU's and XCD's caches)
      const int64_t e = ((tile * S + lane / p.k) * C::QT + ql) * p.k + lane % p.k;
      s1[0] = __hip_atomic_load(p.part_s + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      id1[0] = __hip_atomic_load(p.part_i + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wave_select<1>(s1, id1, p.k, lane, p.idx_base, os, oi);
  }
  RG_FSTAMP(12);
}

}  // namespace ragraph

using namespace ragraph;

static bool fused_shape_ok(int64_t B, int64_t N, int D, int k) {
  return B >= 1 && (D == 64 || D == 128 || D == 256) && k >= 1 && k <= 16 && N >= 32 * 4 * (int64_t)k &&
         N < (int64_t)INT_MAX - 1024;
}

extern "C" int ragraph_topk_cosine_fused_ok(int64_t B, int64_t N, int D, int k) { return fused_shape_ok(B, N, D, k) ? 1 : 0; }

template <int D>
static int launch_fused(const FusedParams& p, hipStream_t st) {
  using C = FusedCfg<D>;
  static DeviceOnce lds_once;
  if (hipError_t e = raise_dynamic_lds(lds_once, &topk_fused_kernel<D>, (int)C::LDS_BYTES); e != hipSuccess) {
    set_error("topk_cosine_fused: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    return RAGRAPH_EDEVICE;
  }
  hipLaunchKernelGGL(topk_fused_kernel<D>, dim3((unsigned)(cdiv(p.B, (int64_t)C::QT) * p.S)), dim3(C::THREADS), C::LDS_BYTES, st, p);
  RG_CHECK_LAUNCH("topk_cosine_fused");
#ifdef RG_FUSED_TIMING
  {
    (void)hipDeviceSynchronize();
    unsigned long long t[2][16];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_fused_t), sizeof(t));
    for (int b = 0; b < 2; ++b) {
      fprintf(stderr, "[fused timing, %s block, 10 ns ticks since its start]", b ? "last" : "first");
      for (int i = 1; i < 13; ++i) fprintf(stderr, " %d:%lld", i, (long long)(t[b][i] - t[b][0]));
      fprintf(stderr, "   (last block started %lld after the first)\n", (long long)(t[1][0] - t[0][0]));
    }
  }
#endif
  return RAGRAPH_OK;
}

// Key splits per query tile: enough workgroups for the chip (every workgroup streams 1/S of the copy), S k <= 64 (the merge
// is one wave_select over a lane per entry), at least a few units per split.
static int fused_splits(int64_t B, int64_t N, int D, int k) {
  static const int force = [] { const char* e = getenv("RAGRAPH_FUSED_S"); return e ? atoi(e) : 0; }();  // A/B
  if (force >= 1 && force <= 8 && force * k <= 64) return force;
  const int64_t tiles = cdiv(B, (int64_t)32);
  const int cus = device_cus_multiple_of_8();
  int64_t S = cus / tiles;
  if (S > 8) S = 8;
  if (S > 64 / k) S = 64 / k;
  const int64_t units = cdiv(N * D * 2, (int64_t)16384);
  if (S > units / 16) S = units / 16;
  return S < 1 ? 1 : (int)S;
}

extern "C" size_t ragraph_topk_cosine_fused_workspace_bytes(int64_t B, int64_t N, int D, int k) {
  if (!fused_shape_ok(B, N, D, k)) return 0;
  const int64_t tiles = cdiv(B, (int64_t)32);
  const int S = fused_splits(B, N, D, k);
  if (S == 1) return 256;
  return align_up((size_t)tiles * S * 32 * k * 4, 256) * 2 + align_up((size_t)tiles * S * 32 * 4, 256);
}

extern "C" int ragraph_topk_cosine_fused_f32(const float* Q, int64_t B, const float* Kn, const uint16_t* Kb, int64_t N, int D,
                                             int k, int64_t idx_base, float* out_scores, int64_t* out_idx, int* tickets,
                                             void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(Q && Kn && Kb && out_scores && out_idx && tickets && ws, RAGRAPH_EINVAL, "topk_cosine_fused: null pointer");
  RG_REQUIRE(ws_bytes >= ragraph_topk_cosine_fused_workspace_bytes(B, N, D, k), RAGRAPH_EWORKSPACE,
             "topk_cosine_fused: workspace %zu < %zu", ws_bytes, ragraph_topk_cosine_fused_workspace_bytes(B, N, D, k));
  RG_REQUIRE(fused_shape_ok(B, N, D, k), RAGRAPH_EUNSUPPORTED,
             "topk_cosine_fused: needs D in {64,128,256}, k <= 16 and N >= 128 k (B=%lld N=%lld D=%d k=%d)", (long long)B,
             (long long)N, D, k);
  RG_REQUIRE(aligned16(Q) && aligned16(Kn) && aligned16(Kb), RAGRAPH_EINVAL, "topk_cosine_fused: pointers must be 16-B aligned");
  FusedParams p{};
  p.Q = Q;
  p.Kn = Kn;
  p.Kb = Kb;
  p.B = B;
  p.N = N;
  p.k = k;
  p.G = 4 * k;
  // the bound's prefix: N/8 of the bank (N/4 for k > 8: ~100 expected candidates per query either way), at least one
  // 32-key sub-tile per part, whole sub-tiles of REAL keys only (the copy's padding rows would score 0)
  int64_t psubs = N / (k > 8 ? 4 : 8) / 32;
  if (psubs < p.G) psubs = p.G;
  if (psubs > N / 32) psubs = N / 32;
  p.psubs = (int)psubs;
  p.idx_base = idx_base;
  p.out_s = out_scores;
  p.out_i = out_idx;
  p.S = fused_splits(B, N, D, k);
  p.tickets = tickets;
  {
    const int64_t tiles = cdiv(B, (int64_t)32);
    const size_t lists = align_up((size_t)tiles * p.S * 32 * k * 4, 256);
    p.part_s = reinterpret_cast<float*>(ws);
    p.part_i = reinterpret_cast<int*>(static_cast<char*>(ws) + lists);
    p.part_over = reinterpret_cast<int*>(static_cast<char*>(ws) + 2 * lists);
  }
  hipStream_t st = as_stream(stream);
  if (D == 256) {
    p.nunits = cdiv(N, (int64_t)FusedCfg<256>::UNIT_KEYS);
    return launch_fused<256>(p, st);
  }
  if (D == 128) {
    p.nunits = cdiv(N, (int64_t)FusedCfg<128>::UNIT_KEYS);
    return launch_fused<128>(p, st);
  }
  p.nunits = cdiv(N, (int64_t)FusedCfg<64>::UNIT_KEYS);
  return launch_fused<64>(p, st);
}
