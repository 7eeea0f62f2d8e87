// Y = act(X @ W^T + b) on the fp32 MFMA (v_mfma_f32_32x32x2_f32), generic M, N, K.
//
// Replaces self.fc(seq) of layers/gcn.py:32, TaskDecoder.py:15-16 (fc1 + LeakyReLU, fc2) and the materialised score
// matrix of SimilarityFunctions.py:14.  These are the genuinely dense, small contractions of the path (c2: 100k x 128
// x 256 = 6.5 GFLOP per forward, < 1 % of the retrieval work), so this kernel is correctness- and numerics-first:
// each output is one fmaf chain over k = 0..K-1 (natural order, from +0), bias added after, bit-identical to
// oracle/ragraph_oracle.c for every shape.
#include "common.h"

namespace ragraph {

// 256 threads = 4 waves as 2 (M) x 2 (N); block tile 64 x 64; K chunks of 32 staged through LDS (row stride 33 floats:
// the ds_read_b32 of lane (i, h) at [i][2kk + h] then hits 32 distinct banks per half-wave).
constexpr int LBM = 64, LBN = 64, LKC = 32, LLD = LKC + 1;

__global__ void __launch_bounds__(256) linear_kernel(const float* __restrict__ X, int64_t M, int K,
                                                     const float* __restrict__ W, int64_t N,
                                                     const float* __restrict__ bias, int act, float alpha,
                                                     float* __restrict__ Y) {
  __shared__ float As[2][LBM * LLD];
  __shared__ float Bs[2][LBN * LLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * LBM;
  const int64_t n0 = (int64_t)blockIdx.y * LBN;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // stage: 64 x 32 of X and of W per chunk, zero-padded outside (0*0 leaves a chain unchanged).  The next chunk's 16
  // values per thread are in flight under the current chunk's MFMAs and written to the other buffer behind them: one
  // barrier per chunk and no exposed load latency (the GCN encode of a Cora-sized graph, K = 1433, is 45 chunks on 86
  // workgroups -- with load, barrier, MFMA, barrier in sequence it was 223 us of a 475 us forward).
  constexpr int PER = (LBM * LKC) / 256;
  // TWO register sets: chunk c travels in set c & 1, loaded two chunks ahead of its MFMAs -- with one set (loaded one
  // chunk ahead) an iteration cost a memory latency, ~1.1 us for 0.5 us of MFMA: the Cora-sized encode (K = 1433: 45
  // chunks on 86 workgroups) 52.6 us
  float ra[2][PER], rb[2][PER];
  auto gload = [&](int k0, float (&pa)[PER], float (&pb)[PER]) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * 256;
      const int row = e / LKC, c = e % LKC;
      const int kk = k0 + c;
      const int64_t gm = m0 + row, gn = n0 + row;
      pa[u] = (gm < M && kk < K) ? X[gm * K + kk] : 0.f;
      pb[u] = (gn < N && kk < K) ? W[gn * K + kk] : 0.f;
    }
  };
  auto sstore = [&](int buf, const float (&pa)[PER], const float (&pb)[PER]) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * 256;
      const int row = e / LKC, c = e % LKC;
      As[buf][row * LLD + c] = pa[u];
      Bs[buf][row * LLD + c] = pb[u];
    }
  };
  auto compute = [&](int buf) {
    const float* a = As[buf] + (wr * 32 + j) * LLD + h;
    const float* b = Bs[buf] + (wc * 32 + j) * LLD + h;
#pragma unroll
    for (int kk = 0; kk < LKC / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk], b[2 * kk], acc, 0, 0, 0);
  };
  const int nch = (K + LKC - 1) / LKC;
  gload(0, ra[0], rb[0]);
  sstore(0, ra[0], rb[0]);
  if (nch > 1) gload(LKC, ra[1], rb[1]);
  if (nch > 2) gload(2 * LKC, ra[0], rb[0]);
  __syncthreads();
  // iteration ch: MFMAs of chunk ch (LDS buffer ch & 1); chunk ch + 1 (set (ch + 1) & 1, loaded two iterations ago) goes
  // to the other buffer -- whose last readers passed the barrier that ended chunk ch - 1 --, chunk ch + 3 into its set
  for (int ch = 0; ch < nch; ch += 2) {
    compute(0);
    if (ch + 1 < nch) sstore(1, ra[1], rb[1]);
    if (ch + 3 < nch) gload((ch + 3) * LKC, ra[1], rb[1]);
    __syncthreads();
    if (ch + 1 >= nch) break;
    compute(1);
    if (ch + 2 < nch) sstore(0, ra[0], rb[0]);
    if (ch + 4 < nch) gload((ch + 4) * LKC, ra[0], rb[0]);
    __syncthreads();
  }

  const int64_t n = n0 + wc * 32 + j;
  if (n < N) {
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < M) {
        float v = acc[r];
        if (bias) v = __fadd_rn(v, bv);
        Y[m * N + n] = apply_act(v, act, alpha);
      }
    }
  }
}

// ---- tall X, a handful of output columns (the class head: TaskDecoder's fc2, 256 -> C) --------------------------------------
// 100 000 x 256 -> 3 through the 64 x 64 MFMA tiles is 61 idle columns of 64 and 51 us for 100 MB of X; the work is reading X.
// A wave owns 64 rows, lane = row: X passes through a wave-private LDS tile (coalesced 16-byte loads in, the lane's row out --
// rows 33 floats apart, conflict-free), W sits in LDS and is read as a broadcast, and every output is the plain fmaf chain
// over k = 0 .. K - 1 from +0 -- what the MFMA kernels compute, so the same bits.  The next chunk's loads are in flight under
// the chains of the current one.
constexpr int NARROW_MAX_N = 8, NARROW_KC = 32, NARROW_LD = NARROW_KC + 1;
template <int NN>
__global__ void __launch_bounds__(256) linear_narrow_kernel(const float* __restrict__ X, int64_t M, int K,
                                                            const float* __restrict__ W, const float* __restrict__ bias, int act,
                                                            float alpha, float* __restrict__ Y) {
  extern __shared__ float4 lin_smem4[];
  float* Ws = reinterpret_cast<float*>(lin_smem4);               // [NN][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* tile = Ws + NN * K + wave * (64 * NARROW_LD);            // this wave's [64][NARROW_LD]
  for (int e = tid; e < NN * K; e += 256) Ws[e] = W[e];
  __syncthreads();
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 64;     // the wave's rows
  if (row0 >= M) return;
  // staging: load u of a chunk brings rows 8 u + lane / 8, floats 4 (lane % 8) .. + 3 of the chunk (8 x 128 B per instruction)
  const int sr = lane >> 3, sc = 4 * (lane & 7);
  float4 pre[8];
  auto gload = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t r = row0 + 8 * u + sr;
      pre[u] = *reinterpret_cast<const float4*>(X + (r < M ? r : M - 1) * K + k0 + sc);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float* d = tile + (8 * u + sr) * NARROW_LD + sc;
      d[0] = pre[u].x; d[1] = pre[u].y; d[2] = pre[u].z; d[3] = pre[u].w;
    }
  };
  float acc[NN];
#pragma unroll
  for (int c = 0; c < NN; ++c) acc[c] = 0.f;
  gload(0);
  for (int k0 = 0; k0 < K; k0 += NARROW_KC) {
    sstore();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (k0 + NARROW_KC < K) gload(k0 + NARROW_KC);
    const float* xr = tile + lane * NARROW_LD;
#pragma unroll 8
    for (int kk = 0; kk < NARROW_KC; ++kk) {
      const float x = xr[kk];
#pragma unroll
      for (int c = 0; c < NN; ++c) acc[c] = fmaf(x, Ws[c * K + k0 + kk], acc[c]);
    }
    __builtin_amdgcn_wave_barrier();   // (the tile is rewritten next round: every lane has read its row)
  }
  const int64_t m = row0 + lane;
  if (m < M) {
#pragma unroll
    for (int c = 0; c < NN; ++c) {
      float v = acc[c];
      if (bias) v = __fadd_rn(v, bias[c]);
      Y[m * NN + c] = apply_act(v, act, alpha);
    }
  }
}

// ---- few output tiles, long K: 32 x 32 block tile, one 16 x 16 MFMA tile per wave -----------------------------------
// A 64 x 64 block of linear_kernel is a chain of K/2 dependent v_mfma_f32_32x32x2_f32 per wave (64 cycles each): the GCN
// encode of a Cora-sized graph (2708 x 1433 -> 128: 86 blocks on 256 CUs, 717 dependent MFMAs) took 50 us with two
// thirds of the chip idle.  v_mfma_f32_16x16x4_f32 walks four k-steps in 32 cycles -- the same fmaf chain per output, so
// the same bits -- and a 16 x 16 tile per wave spreads the same outputs over four times the waves.
constexpr int SBM = 32, SBN = 32;

__global__ void __launch_bounds__(256) linear_small_kernel(const float* __restrict__ X, int64_t M, int K,
                                                           const float* __restrict__ W, int64_t N,
                                                           const float* __restrict__ bias, int act, float alpha,
                                                           float* __restrict__ Y) {
  __shared__ float As[2][SBM * LLD];
  __shared__ float Bs[2][SBN * LLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * SBM;
  const int64_t n0 = (int64_t)blockIdx.y * SBN;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // stage: 32 x 32 of X and of W per chunk (4 + 4 values per thread), two register sets as in linear_kernel
  constexpr int PER = (SBM * LKC) / 256;
  float ra[2][PER], rb[2][PER];
  auto gload = [&](int k0, float (&pa)[PER], float (&pb)[PER]) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * 256;
      const int row = e / LKC, c = e % LKC;
      const int kk = k0 + c;
      const int64_t gm = m0 + row, gn = n0 + row;
      pa[u] = (gm < M && kk < K) ? X[gm * K + kk] : 0.f;
      pb[u] = (gn < N && kk < K) ? W[gn * K + kk] : 0.f;
    }
  };
  auto sstore = [&](int buf, const float (&pa)[PER], const float (&pb)[PER]) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * 256;
      const int row = e / LKC, c = e % LKC;
      As[buf][row * LLD + c] = pa[u];
      Bs[buf][row * LLD + c] = pb[u];
    }
  };
  auto compute = [&](int buf) {  // lane (i, kq): A[i][4 s + kq], B[4 s + kq][j]: k-step s of the chunk
    const float* a = As[buf] + (wr * 16 + i16) * LLD + kq;
    const float* b = Bs[buf] + (wc * 16 + i16) * LLD + kq;
#pragma unroll
    for (int s4 = 0; s4 < LKC / 4; ++s4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * s4], b[4 * s4], acc, 0, 0, 0);
  };
  const int nch = (K + LKC - 1) / LKC;
  gload(0, ra[0], rb[0]);
  sstore(0, ra[0], rb[0]);
  if (nch > 1) gload(LKC, ra[1], rb[1]);
  if (nch > 2) gload(2 * LKC, ra[0], rb[0]);
  __syncthreads();
  for (int ch = 0; ch < nch; ch += 2) {
    compute(0);
    if (ch + 1 < nch) sstore(1, ra[1], rb[1]);
    if (ch + 3 < nch) gload((ch + 3) * LKC, ra[1], rb[1]);
    __syncthreads();
    if (ch + 1 >= nch) break;
    compute(1);
    if (ch + 2 < nch) sstore(0, ra[0], rb[0]);
    if (ch + 4 < nch) gload((ch + 4) * LKC, ra[0], rb[0]);
    __syncthreads();
  }
  // result: lane (j = lane % 16, kq): rows 4 kq + r of the wave's tile, column j
  const int64_t n = n0 + wc * 16 + i16;
  if (n < N) {
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = m0 + wr * 16 + 4 * kq + r;
      if (m < M) {
        float v = acc[r];
        if (bias) v = __fadd_rn(v, bv);
        Y[m * N + n] = apply_act(v, act, alpha);
      }
    }
  }
}

// ---- large shapes: 128 x 128 block tile, 4 waves x (64 x 64), K chunks of 32 double-buffered through LDS -----------
// Used when M >= 128, N >= 128, K % 4 == 0 and the rows are 16-B aligned (the GCN encode of c2: 100k x 128 -> 256).
// Same numerics as linear_kernel (every output is the k = 0..K-1 fmaf chain the MFMA computes, bias added after), so
// the two kernels are interchangeable bit for bit.  Per wave and K chunk: 4 A-row + 4 B-row ds_read_b128 feed 64 MFMAs
// (each fragment is used by two tiles); the LDS image is de-interleaved like the top-k kernel's (row = [even k | odd k]
// + 16 B pad: one conflict-free ds_read_b128 = four MFMA steps); the next chunk's 8 float4 per thread are in flight
// under the MFMAs and written behind them; one barrier per chunk.
constexpr int TBM = 128, TBN = 128, TKC = 32, TLD = TKC + 4;

__global__ void __launch_bounds__(256, 2) linear_tile_kernel(const float* __restrict__ X, int64_t M, int K,
                                                             const float* __restrict__ W, int64_t N,
                                                             const float* __restrict__ bias, int act, float alpha,
                                                             float* __restrict__ Y) {
  extern __shared__ float4 lin_smem4[];
  float* smem = reinterpret_cast<float*>(lin_smem4);  // A[2][128][TLD] then B[2][128][TLD]
  constexpr int BUF = TBM * TLD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * TBM;
  const int64_t n0 = (int64_t)blockIdx.y * TBN;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // staging: thread t owns float4 q = t % 8 (k = 4q..4q+3 of the chunk) of rows t / 8 + 32 u, u = 0..3, of A and of B
  const int srow = tid >> 3, sq = tid & 7;
  float4 pa[4], pb[4];
  auto stage_load = [&](int k0) {
    const int kk = k0 + 4 * sq;
    const bool kin = kk < K;  // K % 4 == 0: a float4 is inside or outside as a whole
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t gm = m0 + srow + 32 * u, gn = n0 + srow + 32 * u;
      pa[u] = (kin && gm < M) ? *reinterpret_cast<const float4*>(X + gm * K + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
      pb[u] = (kin && gn < N) ? *reinterpret_cast<const float4*>(W + gn * K + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage_write = [&](int buf) {
    float* A = smem + buf * BUF;
    float* B = smem + 2 * BUF + buf * BUF;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float* da = A + (srow + 32 * u) * TLD + 2 * sq;
      float* db = B + (srow + 32 * u) * TLD + 2 * sq;
      da[0] = pa[u].x; da[1] = pa[u].z; da[TKC / 2] = pa[u].y; da[TKC / 2 + 1] = pa[u].w;
      db[0] = pb[u].x; db[1] = pb[u].z; db[TKC / 2] = pb[u].y; db[TKC / 2 + 1] = pb[u].w;
    }
  };

  const int nchunks = (K + TKC - 1) / TKC;
  stage_load(0);
  stage_write(0);
  __syncthreads();
  for (int s = 0; s < nchunks; ++s) {
    const bool more = s + 1 < nchunks;
    if (more) stage_load((s + 1) * TKC);  // in flight under this chunk's MFMAs
    const float* A = smem + (s & 1) * BUF + (wr * 64 + j) * TLD + h * (TKC / 2);
    const float* B = smem + 2 * BUF + (s & 1) * BUF + (wc * 64 + j) * TLD + h * (TKC / 2);
#pragma unroll
    for (int c = 0; c < TKC / 8; ++c) {
      const float4 a0 = *reinterpret_cast<const float4*>(A + 4 * c);
      const float4 a1 = *reinterpret_cast<const float4*>(A + 32 * TLD + 4 * c);
      const float4 b0 = *reinterpret_cast<const float4*>(B + 4 * c);
      const float4 b1 = *reinterpret_cast<const float4*>(B + 32 * TLD + 4 * c);
#define RG_STEP(e_)                                                                        \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.e_, b0.e_, acc[0][0], 0, 0, 0);       \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.e_, b1.e_, acc[0][1], 0, 0, 0);       \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.e_, b0.e_, acc[1][0], 0, 0, 0);       \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.e_, b1.e_, acc[1][1], 0, 0, 0);
      RG_STEP(x) RG_STEP(y) RG_STEP(z) RG_STEP(w)
#undef RG_STEP
    }
    if (more) stage_write((s + 1) & 1);
    __syncthreads();
  }

#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int64_t n = n0 + wc * 64 + b * 32 + j;
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M) {
          float v = acc[a][b][r];
          if (bias) v = __fadd_rn(v, bv);
          Y[m * N + n] = apply_act(v, act, alpha);
        }
      }
  }
}

// ---- tall X, N <= 256-wide blocks: stream the rows of X past W held in registers --------------------------------
// The GCN encode of c2 is 100k x 128 -> 256: X is read once, Y written once (153 MB) against 6.5 GFLOP, i.e. it should
// run at the speed of its memory traffic.  Same structure as the fused top-k tile kernel (topk_cosine.hip) with the
// roles keys -> rows of X (streamed in 32 KiB stages through LDS, de-interleaved [even k | odd k] rows, one
// conflict-free ds_read_b128 per four MFMAs) and queries -> rows of W (wave w keeps the 32 output columns
// n0 + 32 w .. + 31 as its B operand in K/2 VGPRs for the whole stream); the epilogue stores the 32 x 32 accumulator
// tile (+ bias, activation) instead of selecting from it.  Double buffered with one barrier per stage; the next
// stage's four float4 per thread are in flight under the MFMAs.  Same fmaf chains as the other two kernels.
template <int KD>
struct LinStreamCfg {
  static constexpr int WAVES = 8, THREADS = 512;
  static constexpr int TILES = 256 / KD;            // 32-row MFMA tiles per stage
  static constexpr int STAGE_ROWS = 32 * TILES;     // 32 KiB of X per stage
  static constexpr int ROW = KD + 4;                // padded LDS row (floats)
  static constexpr int STAGE_FLOATS = STAGE_ROWS * ROW;
  static constexpr size_t LDS_BYTES = sizeof(float) * 2 * STAGE_FLOATS;
};

template <int KD>
__global__ void __launch_bounds__(512, 2) linear_stream_kernel(const float* __restrict__ X, int64_t M,
                                                               const float* __restrict__ W, int64_t N,
                                                               const float* __restrict__ bias, int act, float alpha,
                                                               float* __restrict__ Y, int64_t stages_per_wg) {
  using C = LinStreamCfg<KD>;
  extern __shared__ float4 lin_smem4[];
  float* smem = reinterpret_cast<float*>(lin_smem4);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int64_t n = (int64_t)blockIdx.y * 256 + wave * 32 + j;  // this lane's output column
  const int64_t total_stages = (M + C::STAGE_ROWS - 1) / C::STAGE_ROWS;
  const int64_t st0 = (int64_t)blockIdx.x * stages_per_wg;
  const int64_t st1 = min(total_stages, st0 + stages_per_wg);
  if (st0 >= st1) return;
  const int nstages = (int)(st1 - st0);

  // B operand: row n of W (clamped; columns >= N are computed and never stored), k-slots h, h+2, ...
  float breg[KD / 2];
  {
    const float4* wp = reinterpret_cast<const float4*>(W + (n < N ? n : N - 1) * KD);
#pragma unroll
    for (int c0 = 0; c0 < KD / 4; c0 += 16) {
#pragma unroll
      for (int c = c0; c < c0 + 16 && c < KD / 4; ++c) {
        const float4 v = wp[c];
        breg[2 * c] = h ? v.y : v.x;
        breg[2 * c + 1] = h ? v.w : v.z;
      }
#pragma unroll
      for (int c = c0; c < c0 + 16 && c < KD / 4; ++c) asm volatile("" : "+v"(breg[2 * c]), "+v"(breg[2 * c + 1]));
      asm volatile("" ::: "memory");
    }
  }
  const float bv = (bias && n < N) ? bias[n] : 0.f;

  // staging: thread t owns float4 chunks t, t + 512, ... of the stage (row = chunk / (KD/4)); rows >= M are clamped
  float4 pre0, pre1, pre2, pre3;
  const int srow = tid / (KD / 4), scol = 4 * (tid % (KD / 4));
  constexpr int SROWS = C::THREADS / (KD / 4);
  auto stage_load = [&](int64_t s_) {
    const int64_t r_ = (st0 + s_) * C::STAGE_ROWS + srow, last_ = M - 1;
    pre0 = *reinterpret_cast<const float4*>(X + min(r_, last_) * KD + scol);
    pre1 = *reinterpret_cast<const float4*>(X + min(r_ + SROWS, last_) * KD + scol);
    pre2 = *reinterpret_cast<const float4*>(X + min(r_ + 2 * SROWS, last_) * KD + scol);
    pre3 = *reinterpret_cast<const float4*>(X + min(r_ + 3 * SROWS, last_) * KD + scol);
  };
  auto stage_write = [&](int buf) {
    float* d = smem + buf * C::STAGE_FLOATS + srow * C::ROW + (scol >> 1);
#define RG_W2(o_, v_)                                   \
  d[(o_)] = (v_).x; d[(o_) + 1] = (v_).z;               \
  d[(o_) + KD / 2] = (v_).y; d[(o_) + KD / 2 + 1] = (v_).w;
    RG_W2(0, pre0) RG_W2(SROWS * C::ROW, pre1) RG_W2(2 * SROWS * C::ROW, pre2) RG_W2(3 * SROWS * C::ROW, pre3)
#undef RG_W2
  };

  stage_load(0);
  stage_write(0);
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    const bool more = s + 1 < nstages;
    stage_load(more ? s + 1 : s);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads above the MFMA block
#pragma unroll 1
    for (int t = 0; t < C::TILES; ++t) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* arow = smem + (s & 1) * C::STAGE_FLOATS + (t * 32 + j) * C::ROW + h * (KD / 2);
#pragma unroll
      for (int c = 0; c < KD / 8; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(arow + 4 * c);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, breg[4 * c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, breg[4 * c + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, breg[4 * c + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, breg[4 * c + 3], acc, 0, 0, 0);
      }
      // acc[r] = Y[row (r&3) + 8 (r>>2) + 4 h of the tile][n]: for a fixed r the 32 lanes of a half-wave store 128 B
      const int64_t m_base = (st0 + s) * C::STAGE_ROWS + t * 32 + 4 * h;
      if (n < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = m_base + (r & 3) + 8 * (r >> 2);
          if (m < M) {
            float v = acc[r];
            if (bias) v = __fadd_rn(v, bv);
            Y[m * N + n] = apply_act(v, act, alpha);
          }
        }
      }
    }
    if (more) stage_write((s + 1) & 1);
    __syncthreads();
  }
}

// ---- the aggregate-first GCN layer in ONE launch (round 6): Y = act((A X) W^T + b) ---------------------------------------
// layers/gcn.py:36-40 evaluated as (A_hat X) W^T (inference of a layer at most half as wide in as out) was
// two launches -- the narrow aggregation writes [n, KD], the dense layer reads it back -- of 67 + 90 us at c2 (100 000 x 128 ->
// 256), one bound by the CUs' gather path, the other by the matrix pipe.  Here a workgroup is 16 waves of two kinds: waves 8-15
// make the stages -- a group of KD / 4 lanes walks an output row's edges (the fmaf chain of spmm_csr_kernel: edge order, from
// +0; rows in pairs, eight 16-byte gathers per row in flight per lane) and leaves the aggregated row in the LDS stage buffer in
// the dense kernel's [even k | odd k] layout -- while waves 0-7 run linear_stream_kernel's MFMA chains on the stage before: the
// bits of the two-launch form, the aggregated table never in HBM, gathers and MFMAs on the same CU at the same time.  The
// gather waves keep two prefetches ahead of themselves (the row pointers of the stage after next, the first KD / 4 (col, val)
// pairs of the next stage's rows: one load covers a row of up to KD / 4 edges), so a stage waits for ONE dependent latency,
// the gathers themselves.  Rows longer than ROW_BLOCK edges are summed in blocks as everywhere (csrc/sparse.hip).
constexpr int AGG_ROW_BLOCK = 4096;   // = sparse.hip's ROW_BLOCK (the oracle's ORACLE_ROW_BLOCK)
template <int KD>
struct AggLinCfg {
  static constexpr int THREADS = 1024, MFMA_THREADS = 512;
  static constexpr int LPR = KD / 4;                       // lanes per aggregated row (a float4 each)
  static constexpr int GROUPS = (THREADS - MFMA_THREADS) / LPR;
  static constexpr int R = 2;                              // rows per lane group per stage (one pair)
  static constexpr int STAGE_ROWS = R * GROUPS;            // 32 (KD = 128) / 64 (KD = 64)
  static constexpr int TILES = STAGE_ROWS / 32;
  static constexpr int ROW = KD + 4;                       // padded LDS row (floats)
  static constexpr int STAGE_FLOATS = STAGE_ROWS * ROW;
#ifndef RG_AGGLIN_NB
#define RG_AGGLIN_NB 4
#endif
  static constexpr int NB = RG_AGGLIN_NB;                  // ring of stage buffers: the gather half runs up to 3 stages ahead
  static constexpr size_t LDS_BYTES = sizeof(float) * NB * STAGE_FLOATS + sizeof(int) * 2 * NB;
};

// the two halves meet through counters in LDS, not barriers: ready[b] counts the gather waves that have finished a stage in
// buffer b, freed[b] the MFMA waves that are done reading one (both only grow: stage s of buffer b = s % NB is complete at
// ready[b] = 8 (s / NB + 1) and may be overwritten at freed[b] = 8 (s / NB + 1)).  A stage's gathers take one to three
// dependent round trips depending on its longest row, its MFMAs a fixed time: in lockstep every stage cost the slower of
// the two (136 us at c2), with the ring the sums overlap.
__device__ __forceinline__ void agglin_wait(int* f, int target) {
#ifdef RG_AGGLIN_FREE_RUN   // (diagnostic build: the halves never wait for each other -- wrong results, the cost of sharing a CU)
  return;
#endif
  int spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 26)) __builtin_trap();   // (seconds: a lost count is a bug, not a wait)
  }
}
__device__ __forceinline__ void agglin_signal(int* f, int lane) {
  if (lane == 0) __hip_atomic_fetch_add(f, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int KD>
__global__ void __launch_bounds__(1024, 1) spmm_linear_stream_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                     const float* __restrict__ val, const float* __restrict__ X,
                                                                     int64_t M, const float* __restrict__ W, int64_t N,
                                                                     const float* __restrict__ bias, int act, float alpha,
                                                                     float* __restrict__ Y) {
  using C = AggLinCfg<KD>;
  extern __shared__ float4 lin_smem4[];
  float* smem = reinterpret_cast<float*>(lin_smem4);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t total_stages = (M + C::STAGE_ROWS - 1) / C::STAGE_ROWS;
  const int64_t st0 = (int64_t)blockIdx.x * total_stages / gridDim.x;          // stages dealt evenly: counts differ by one at most
  const int64_t st1 = (int64_t)(blockIdx.x + 1) * total_stages / gridDim.x;
  if (st0 >= st1) return;
  const int nstages = (int)(st1 - st0);
  int* ready = reinterpret_cast<int*>(smem + C::NB * C::STAGE_FLOATS);
  int* freed = ready + C::NB;
  if (tid < 2 * C::NB) ready[tid] = 0;
  __syncthreads();

  if (wave < 8) {
    // ---- the dense half: linear_stream_kernel's chains on the stages the other half has marked ready ----
    const int j = lane & 31, h = lane >> 5;
    const int64_t n = (int64_t)blockIdx.y * 256 + wave * 32 + j;  // this lane's output column
    float breg[KD / 2];   // B operand: row n of W, k-slots h, h + 2, ... (as linear_stream_kernel)
    {
      const float4* wp = reinterpret_cast<const float4*>(W + (n < N ? n : N - 1) * KD);
#pragma unroll
      for (int c0 = 0; c0 < KD / 4; c0 += 16) {
#pragma unroll
        for (int c = c0; c < c0 + 16 && c < KD / 4; ++c) {
          const float4 v = wp[c];
          breg[2 * c] = h ? v.y : v.x;
          breg[2 * c + 1] = h ? v.w : v.z;
        }
#pragma unroll
        for (int c = c0; c < c0 + 16 && c < KD / 4; ++c) asm volatile("" : "+v"(breg[2 * c]), "+v"(breg[2 * c + 1]));
        asm volatile("" ::: "memory");
      }
    }
    const float bv = (bias && n < N) ? bias[n] : 0.f;
    for (int s = 0; s < nstages; ++s) {
      const int buf = s % C::NB;
      agglin_wait(ready + buf, 8 * (s / C::NB + 1));
#ifndef RG_AGGLIN_NO_MFMA   // (diagnostic build: the gather half alone)
#pragma unroll 1
      for (int t = 0; t < C::TILES; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* arow = smem + buf * C::STAGE_FLOATS + (t * 32 + j) * C::ROW + h * (KD / 2);
#pragma unroll
        for (int c = 0; c < KD / 8; ++c) {
          const float4 v = *reinterpret_cast<const float4*>(arow + 4 * c);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, breg[4 * c], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, breg[4 * c + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, breg[4 * c + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, breg[4 * c + 3], acc, 0, 0, 0);
        }
        if (t == C::TILES - 1) agglin_signal(freed + buf, lane);   // (every read of the buffer has returned: the MFMAs took them)
        const int64_t m_base = (st0 + s) * C::STAGE_ROWS + t * 32 + 4 * h;
        if (n < N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t m = m_base + (r & 3) + 8 * (r >> 2);
            if (m < M) {
              float v = acc[r];
              if (bias) v = __fadd_rn(v, bv);
              Y[m * N + n] = apply_act(v, act, alpha);
            }
          }
        }
      }
#else
      agglin_signal(freed + buf, lane);
#endif
    }
    return;
  }

  // ---- the gather half ----
  // (the gather waves' stages are chains of dependent round trips, the MFMA waves have slack: whoever of the two can issue,
  // these go first -- 126 -> 121 us at c2; raising the MFMA waves instead changed nothing)
  __builtin_amdgcn_s_setprio(3);
  constexpr int LPR = C::LPR;
  const int gt = tid - C::MFMA_THREADS;
  const int grp = gt / LPR, lr = gt % LPR, gbase = (lane / LPR) * LPR;   // gbase: first lane of my group in the wave
  const float4* X4 = reinterpret_cast<const float4*>(X);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fma4 = [](float v, const float4& x, float4& a) {
    a.x = fmaf(v, x.x, a.x); a.y = fmaf(v, x.y, a.y); a.z = fmaf(v, x.z, a.z); a.w = fmaf(v, x.w, a.w);
  };
  // a block of one row by itself (hub rows): the (col, val) pairs of a chunk loaded by the group's first lanes
  auto chain_one = [&](int64_t e0, int cnt) {
    float4 acc = zero4;
    for (int base = 0; base < cnt; base += 8) {
      int my_c = 0;
      float my_v = 0.f;
      if (lr < 8 && base + lr < cnt) {
        my_c = col[e0 + base + lr];
        my_v = val[e0 + base + lr];
      }
      float4 x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = __shfl(my_c, gbase + k);
        x[k] = base + k < cnt ? X4[(int64_t)c * LPR + lr] : zero4;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = __shfl(my_v, gbase + k);
        if (base + k < cnt) fma4(v, x[k], acc);
      }
    }
    return acc;
  };
  // per-row state: cur = the stage being made (row pointers + the first LPR pairs in registers, lane lr holding edge lr),
  // nxt = the stage after it (row pointers only: its pairs are fetched while cur is gathered)
  int64_t cur_e0[2], nxt_e0[2];
  int cur_deg[2], nxt_deg[2];   // (saturated: a hub row reads its exact length again)
  int cur_c[2];
  float cur_v[2];
  auto load_rowptr = [&](int stage, int64_t* e0, int* deg) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t r = (st0 + stage) * C::STAGE_ROWS + grp + C::GROUPS * i;
      e0[i] = 0;
      deg[i] = 0;
      if (stage < nstages && r < M) {
        e0[i] = rowptr[r];
        const int64_t d64 = rowptr[r + 1] - e0[i];
        deg[i] = d64 > AGG_ROW_BLOCK ? AGG_ROW_BLOCK + 1 : (int)d64;
      }
    }
  };
  auto load_pairs = [&](const int64_t* e0, const int* deg, int* c, float* v) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      c[i] = 0;
      v[i] = 0.f;
      if (lr < deg[i]) {
        c[i] = col[e0[i] + lr];
        v[i] = val[e0[i] + lr];
      }
    }
  };
  auto make_stage = [&](int stage) {   // rows of `stage` = A x into LDS buffer stage & 1; cur <- the stage after
    int nc[2];
    float nv[2];
    int64_t nn_e0[2];
    int nn_deg[2];
    load_pairs(nxt_e0, nxt_deg, nc, nv);            // (in flight under the gathers below)
    load_rowptr(stage + 2, nn_e0, nn_deg);
    float4 a = zero4, b = zero4;
    if (cur_deg[0] > AGG_ROW_BLOCK || cur_deg[1] > AGG_ROW_BLOCK) {
      // a hub row in the pair: blocks of AGG_ROW_BLOCK edges, each its own chain, the block sums added in order
#pragma unroll 1
      for (int i = 0; i < 2; ++i) {
        const int64_t r = (st0 + stage) * C::STAGE_ROWS + grp + C::GROUPS * i;
        const int64_t e0 = i ? cur_e0[1] : cur_e0[0];
        const int64_t deg = (i ? cur_deg[1] : cur_deg[0]) > AGG_ROW_BLOCK ? rowptr[r + 1] - e0 : (int64_t)(i ? cur_deg[1] : cur_deg[0]);
        float4 tot = zero4;
#pragma unroll 1
        for (int64_t b0 = 0; b0 < deg; b0 += AGG_ROW_BLOCK) {
          const float4 p = chain_one(e0 + b0, (int)(deg - b0 < AGG_ROW_BLOCK ? deg - b0 : AGG_ROW_BLOCK));
          if (b0) {
            tot.x = __fadd_rn(tot.x, p.x); tot.y = __fadd_rn(tot.y, p.y); tot.z = __fadd_rn(tot.z, p.z); tot.w = __fadd_rn(tot.w, p.w);
          } else {
            tot = p;
          }
        }
        if (i) b = tot; else a = tot;
      }
    } else {
      const int dA = cur_deg[0], dB = cur_deg[1];
      int cA = cur_c[0], cB = cur_c[1];
      float vA = cur_v[0], vB = cur_v[1];
#ifdef RG_AGGLIN_NO_GATHER   // (diagnostic build: the dense half alone)
      const int dmax = 0;
#else
      const int dmax = dA > dB ? dA : dB;
#endif
      for (int base = 0; base < dmax; base += 8) {
        const int off = base % LPR;
        if (base && off == 0) {   // past the pairs in registers: the next LPR of them
          cA = 0; vA = 0.f; cB = 0; vB = 0.f;
          if (base + lr < dA) { cA = col[cur_e0[0] + base + lr]; vA = val[cur_e0[0] + base + lr]; }
          if (base + lr < dB) { cB = col[cur_e0[1] + base + lr]; vB = val[cur_e0[1] + base + lr]; }
        }
        float4 xa[8], xb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int c = __shfl(cA, gbase + off + k);   // (0 past the row's end: row 0 is loaded and not used -- a load under
          xa[k] = X4[(int64_t)c * LPR + lr];            //  a condition is a branch and a wait of its own per load)
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int c = __shfl(cB, gbase + off + k);
          xb[k] = X4[(int64_t)c * LPR + lr];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float v = __shfl(vA, gbase + off + k);
          if (base + k < dA) fma4(v, xa[k], a);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float v = __shfl(vB, gbase + off + k);
          if (base + k < dB) fma4(v, xb[k], b);
        }
      }
    }
    const int buf = stage % C::NB;
    if (stage >= C::NB) agglin_wait(freed + buf, 8 * (stage / C::NB));   // the MFMA half is done with the stage NB before
    float* d = smem + buf * C::STAGE_FLOATS + grp * C::ROW + ((4 * lr) >> 1);
    d[0] = a.x; d[1] = a.z;
    d[KD / 2] = a.y; d[KD / 2 + 1] = a.w;
    d += C::GROUPS * C::ROW;
    d[0] = b.x; d[1] = b.z;
    d[KD / 2] = b.y; d[KD / 2 + 1] = b.w;
    agglin_signal(ready + buf, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      cur_e0[i] = nxt_e0[i]; cur_deg[i] = nxt_deg[i]; cur_c[i] = nc[i]; cur_v[i] = nv[i];
      nxt_e0[i] = nn_e0[i]; nxt_deg[i] = nn_deg[i];
    }
  };

  load_rowptr(0, cur_e0, cur_deg);
  load_rowptr(1, nxt_e0, nxt_deg);
  load_pairs(cur_e0, cur_deg, cur_c, cur_v);
#pragma unroll 1
  for (int s = 0; s < nstages; ++s) make_stage(s);
}

template <int KD>
static int launch_spmm_linear(const int64_t* rowptr, const int32_t* col, const float* val, const float* X, int64_t M, const float* W,
                              int64_t N, const float* bias, int act, float alpha, float* Y, hipStream_t st) {
  using C = AggLinCfg<KD>;
  static DeviceOnce lds_once;
  if (hipError_t e = raise_dynamic_lds(lds_once, &spmm_linear_stream_kernel<KD>, (int)C::LDS_BYTES); e != hipSuccess) {
    set_error("spmm_linear: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    return RAGRAPH_EDEVICE;
  }
  const int64_t total_stages = cdiv(M, C::STAGE_ROWS);
  const int64_t col_blocks = cdiv(N, 256);
  int64_t wgs = device_cus_multiple_of_8() / col_blocks;   // one workgroup per CU (every column block aggregates its rows again)
  if (wgs < 1) wgs = 1;
  if (wgs > total_stages) wgs = total_stages;
  hipLaunchKernelGGL(spmm_linear_stream_kernel<KD>, dim3((unsigned)wgs, (unsigned)col_blocks), dim3(C::THREADS), C::LDS_BYTES, st,
                     rowptr, col, val, X, M, W, N, bias, act, alpha, Y);
  RG_CHECK_LAUNCH("spmm_linear");
  return RAGRAPH_OK;
}

template <int KD>
static int launch_linear_stream(const float* X, int64_t M, const float* W, int64_t N, const float* bias, int act,
                                float alpha, float* Y, hipStream_t st) {
  using C = LinStreamCfg<KD>;
  static DeviceOnce lds_once;  // per device (common.h)
  if (hipError_t e = raise_dynamic_lds(lds_once, &linear_stream_kernel<KD>, (int)C::LDS_BYTES); e != hipSuccess) {
    set_error("linear: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
    return RAGRAPH_EDEVICE;
  }
  const int64_t total_stages = cdiv(M, C::STAGE_ROWS);
  const int64_t col_blocks = cdiv(N, 256);
  // ~2 workgroups per CU in flight (LDS 66 KB, <= 128 VGPRs); at least 4 stages each so the pipeline fills
  static const int64_t wgs_env = [] {  // RAGRAPH_LINEAR_WGS: A/B of the workgroup count
    const char* e = getenv("RAGRAPH_LINEAR_WGS");
    return e ? (int64_t)atoll(e) : (int64_t)0;
  }();
  // two workgroups per CU hide each other's latencies on long streams (1M x 128 -> 256: 639 vs 693 us); with a few
  // stages per CU one workgroup each keeps the CUs evenly loaded (100k x 256 -> 256: 148 vs 168 us; 391 workgroups on
  // 256 CUs left half of them with twice the work)
  const int64_t wgs_target = wgs_env > 0 ? wgs_env : (total_stages >= 16 * 256 ? 512 : 256);
  int64_t wgs = wgs_target / col_blocks;
  if (wgs < 1) wgs = 1;
  int64_t per = cdiv(total_stages, wgs);
  // (a short stream -- a rank's slice of the rows: 12 500 x 128 -> 256 is 196 stages -- used to keep "at least 4 stages per
  // workgroup so the pipeline fills": 49 workgroups on 256 CUs, 54 us where the whole 100 000-row stream takes 96.  Fewer
  // stages per workgroup and all CUs busy; RAGRAPH_LINEAR_MIN_STAGES: A/B)
  static const int64_t min_stages = [] {
    const char* e = getenv("RAGRAPH_LINEAR_MIN_STAGES");
    return e ? (int64_t)atoll(e) : (int64_t)1;
  }();
  if (per < min_stages) per = min_stages;
  wgs = cdiv(total_stages, per);
  hipLaunchKernelGGL(linear_stream_kernel<KD>, dim3((unsigned)wgs, (unsigned)col_blocks), dim3(C::THREADS), C::LDS_BYTES,
                     st, X, M, W, N, bias, act, alpha, Y, per);
  RG_CHECK_LAUNCH("linear(stream)");
  return RAGRAPH_OK;
}

// ---- C = A^T B for TALL operands: A [n, M], B [n, N] row-major, C [M, N] -- the weight gradient of a dense layer,
// gW = gY^T X (autograd._Linear.backward; finetune-rag.py:81-84 trains the decoder on every node of the batch: n = 100 000,
// M = N = 256 at c2).  Through ragraph_linear_f32 that product needed both operands TRANSPOSED (two 100-MB copies) and ran as
// 16 blocks with a 100 000-long chain each: 1.8 ms a product.  Here the contraction index is the ROW: a v_mfma_f32_32x32x2
// step consumes two rows, its A operand is 32 consecutive floats of each of them (lane (j, h) = A[r + h][i0 + j]: one
// coalesced dword per lane straight from the row-major operand, no transpose, no LDS), a wave holds a 64 x 64 tile of C, a
// workgroup 128 x 128, and the rows are cut into `splits` ranges over gridDim.y whose partial tiles are summed in range
// order by a second launch (deterministic: every C[i][j] is a fixed tree -- fmaf chains over the rows of a range from +0,
// the ranges added in order).
__global__ void __launch_bounds__(256) linear_tn_kernel(const float* __restrict__ A, const float* __restrict__ B, int64_t n,
                                                        int M, int N, int64_t chunk, int tiles_n, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int i0 = tm * 128 + (wave >> 1) * 64, j0 = tn * 128 + (wave & 1) * 64;
  const int64_t r0 = (int64_t)blockIdx.y * chunk, r1 = r0 + chunk < n ? r0 + chunk : n;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const bool am[2] = {i0 + j < M, i0 + 32 + j < M};
  const bool bm[2] = {j0 + j < N, j0 + 32 + j < N};
  const float* ap = A + i0 + j;
  const float* bp = B + j0 + j;
  constexpr int U = 4;   // row pairs in flight
  for (int64_t r = r0; r < r1; r += 2 * U) {
    float av[U][2], bv[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = r + 2 * u + h;
      const bool ok = row < r1;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        av[u][c] = (ok && am[c]) ? ap[row * M + 32 * c] : 0.f;
        bv[u][c] = (ok && bm[c]) ? bp[row * N + 32 * c] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
  }
  float* out = part + (int64_t)blockIdx.y * M * N;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = j0 + 32 * b + j;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M) out[(int64_t)row * N + col] = acc[a][b][r];
      }
    }
}

__global__ void __launch_bounds__(256) linear_tn_reduce_kernel(const float* __restrict__ part, int splits, int64_t mn,
                                                               float* __restrict__ C) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= mn) return;
  float s = part[e];
  for (int p = 1; p < splits; ++p) s = __fadd_rn(s, part[(int64_t)p * mn + e]);
  C[e] = s;
}

static int linear_tn_splits(int64_t n, int M, int N) {
  const int64_t tiles = cdiv(M, 128) * cdiv(N, 128);
  int64_t s = cdiv(1024, tiles);
  const int64_t max_s = cdiv(n, 256);   // (at least 256 rows a range)
  if (s > max_s) s = max_s;
  return (int)(s < 1 ? 1 : s);
}

}  // namespace ragraph

using namespace ragraph;

extern "C" size_t ragraph_linear_tn_workspace_bytes(int64_t n, int M, int N) {
  if (n < 1 || M < 1 || N < 1) return 0;
  return (size_t)linear_tn_splits(n, M, N) * (size_t)M * (size_t)N * sizeof(float);
}

extern "C" int ragraph_linear_tn_f32(const float* A, const float* B, int64_t n, int M, int N, float* C, void* ws, size_t ws_bytes,
                                     void* stream) {
  RG_REQUIRE(A && B && C && ws, RAGRAPH_EINVAL, "linear_tn: null pointer");
  RG_REQUIRE(n >= 1 && M >= 1 && N >= 1, RAGRAPH_EINVAL, "linear_tn: n, M, N must be >= 1");
  const int splits = linear_tn_splits(n, M, N);
  RG_REQUIRE(ws_bytes >= ragraph_linear_tn_workspace_bytes(n, M, N), RAGRAPH_EWORKSPACE, "linear_tn: workspace too small");
  int64_t chunk = cdiv(n, (int64_t)splits);
  chunk = (chunk + 7) / 8 * 8;   // (whole groups of four row pairs)
  const int tiles_n = (int)cdiv(N, 128);
  dim3 grid((unsigned)(cdiv(M, 128) * tiles_n), (unsigned)cdiv(n, chunk));
  hipLaunchKernelGGL(linear_tn_kernel, grid, dim3(256), 0, as_stream(stream), A, B, n, M, N, chunk, tiles_n, static_cast<float*>(ws));
  RG_CHECK_LAUNCH("linear_tn");
  const int64_t mn = (int64_t)M * N;
  hipLaunchKernelGGL(linear_tn_reduce_kernel, dim3((unsigned)cdiv(mn, 256)), dim3(256), 0, as_stream(stream), static_cast<const float*>(ws),
                     (int)grid.y, mn, C);
  RG_CHECK_LAUNCH("linear_tn(reduce)");
  return RAGRAPH_OK;
}

extern "C" int ragraph_linear_f32(const float* X, int64_t M, int K, const float* W, int64_t N, const float* bias,
                                  int act, float alpha, float* Y, void* stream) {
  RG_REQUIRE(X && W && Y, RAGRAPH_EINVAL, "linear: null pointer");
  RG_REQUIRE(M >= 1 && N >= 1 && K >= 1, RAGRAPH_EINVAL, "linear: M,N,K must be >= 1");
  RG_REQUIRE(act >= RAGRAPH_ACT_NONE && act <= RAGRAPH_ACT_ELU, RAGRAPH_EINVAL, "linear: bad act %d", act);
  RG_REQUIRE(cdiv(N, LBN) <= 65535, RAGRAPH_EUNSUPPORTED, "linear: N=%lld too large for one launch", (long long)N);
  static const bool tile_ok = [] {  // RAGRAPH_LINEAR_TILE=0: diagnostic switch back to the small-tile kernel (read once)
    const char* e = getenv("RAGRAPH_LINEAR_TILE");
    return !(e && atoi(e) == 0);
  }();
  // tall X, a handful of output columns: the row-streaming kernel (RAGRAPH_LINEAR_NARROW=0: the MFMA tiles, A/B)
  static const bool narrow_ok = [] { const char* e = getenv("RAGRAPH_LINEAR_NARROW"); return !(e && atoi(e) == 0); }();
  if (narrow_ok && N <= NARROW_MAX_N && M >= 4096 && K % NARROW_KC == 0 && K <= 2048 && aligned16(X)) {
    const size_t lds = sizeof(float) * ((size_t)N * K + 4 * 64 * NARROW_LD);
    dim3 grid((unsigned)cdiv(M, 256));
    hipStream_t st = as_stream(stream);
#define RG_NARROW(NN_)                                                                                                    \
  case NN_: {                                                                                                               \
    static DeviceOnce once;                                                                                                 \
    if (hipError_t e = raise_dynamic_lds(once, &linear_narrow_kernel<NN_>, (int)(sizeof(float) * ((size_t)NN_ * 2048 + 4 * 64 * NARROW_LD))); \
        e != hipSuccess) {                                                                                                  \
      set_error("linear: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));                                        \
      return RAGRAPH_EDEVICE;                                                                                               \
    }                                                                                                                       \
    hipLaunchKernelGGL(linear_narrow_kernel<NN_>, grid, dim3(256), lds, st, X, M, K, W, bias, act, alpha, Y);               \
    break;                                                                                                                  \
  }
    switch ((int)N) {
      RG_NARROW(1) RG_NARROW(2) RG_NARROW(3) RG_NARROW(4) RG_NARROW(5) RG_NARROW(6) RG_NARROW(7) RG_NARROW(8)
    }
#undef RG_NARROW
    RG_CHECK_LAUNCH("linear(narrow)");
    return RAGRAPH_OK;
  }
  // tall X against blocks of 256 columns; the last block's idle waves must stay under 25 % of all of them
  if (tile_ok && M >= 4096 && (K == 64 || K == 128 || K == 256) && aligned16(X) && aligned16(W) &&
      4 * N >= 3 * 256 * cdiv(N, 256) && cdiv(N, 256) <= 65535) {
    hipStream_t st = as_stream(stream);
    return K == 256 ? launch_linear_stream<256>(X, M, W, N, bias, act, alpha, Y, st)
           : K == 128 ? launch_linear_stream<128>(X, M, W, N, bias, act, alpha, Y, st)
                      : launch_linear_stream<64>(X, M, W, N, bias, act, alpha, Y, st);
  }
  // (up to a few waves of 128 x 128 tiles the 64 x 64 kernel -- four times the workgroups, next chunk in flight under
  // the MFMAs -- is as fast or faster: 2708 x 128 -> 128: 30 -> 10 us; 600 x 256 -> 20000: 124 -> 103; 300 x 256 ->
  // 5000: 39 -> 20; equal from ~1000 tiles, and the tile kernel wins on large problems: 4096 x 256 -> 125000: 112 TF)
  if (tile_ok && M >= TBM && N >= TBN && K % 4 == 0 && aligned16(X) && aligned16(W) &&
      cdiv(M, TBM) * cdiv(N, TBN) >= 2048) {
    const size_t lds = sizeof(float) * 4 * TBM * TLD;
    static DeviceOnce lds_once;  // per device (common.h)
    if (hipError_t e = raise_dynamic_lds(lds_once, &linear_tile_kernel, (int)lds); e != hipSuccess) {
      set_error("linear: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
      return RAGRAPH_EDEVICE;
    }
    dim3 grid((unsigned)cdiv(M, TBM), (unsigned)cdiv(N, TBN));
    hipLaunchKernelGGL(linear_tile_kernel, grid, dim3(256), lds, as_stream(stream), X, M, K, W, N, bias, act, alpha, Y);
    RG_CHECK_LAUNCH("linear(tile)");
    return RAGRAPH_OK;
  }
  // few 64 x 64 blocks and a long K: the chain of dependent MFMAs per wave is what the launch costs -- quarter-size tiles
  if (cdiv(M, LBM) * cdiv(N, LBN) <= 192 && K >= 256 && cdiv(N, SBN) <= 65535) {
    dim3 grid((unsigned)cdiv(M, SBM), (unsigned)cdiv(N, SBN));
    hipLaunchKernelGGL(linear_small_kernel, grid, dim3(256), 0, as_stream(stream), X, M, K, W, N, bias, act, alpha, Y);
    RG_CHECK_LAUNCH("linear(small)");
    return RAGRAPH_OK;
  }
  dim3 grid((unsigned)cdiv(M, LBM), (unsigned)cdiv(N, LBN));
  hipLaunchKernelGGL(linear_kernel, grid, dim3(256), 0, as_stream(stream), X, M, K, W, N, bias, act, alpha, Y);
  RG_CHECK_LAUNCH("linear");
  return RAGRAPH_OK;
}

// a4, inference association (A_hat X) W^T in one launch  -- layers/gcn.py:36-40 (a layer at most half as
// wide in as out).  rowptr [M + 1] (a slice of a larger graph's row pointers gives those rows), col / val the graph's, X [*, K]
// the table gathered from, W [N, K], Y [M, N] = act((A X) W^T + bias).  K in {64, 128}; the bits of ragraph_spmm_csr_f32
// followed by ragraph_linear_f32.
extern "C" int ragraph_spmm_linear_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t M, const float* X, int K,
                                       const float* W, int64_t N, const float* bias, int act, float alpha, float* Y, void* stream) {
  RG_REQUIRE(rowptr && col && val && X && W && Y, RAGRAPH_EINVAL, "spmm_linear: null pointer");
  RG_REQUIRE(M >= 1 && N >= 1, RAGRAPH_EINVAL, "spmm_linear: M, N must be >= 1");
  RG_REQUIRE(K == 64 || K == 128, RAGRAPH_EUNSUPPORTED, "spmm_linear: K=%d not in {64, 128}", K);
  RG_REQUIRE(aligned16(X) && aligned16(W) && X != Y, RAGRAPH_EINVAL, "spmm_linear: X, W must be 16-B aligned, Y distinct from X");
  RG_REQUIRE(act >= RAGRAPH_ACT_NONE && act <= RAGRAPH_ACT_ELU, RAGRAPH_EINVAL, "spmm_linear: bad act %d", act);
  RG_REQUIRE(cdiv(N, 256) <= 65535, RAGRAPH_EUNSUPPORTED, "spmm_linear: N=%lld too large for one launch", (long long)N);
  hipStream_t st = as_stream(stream);
  return K == 128 ? launch_spmm_linear<128>(rowptr, col, val, X, M, W, N, bias, act, alpha, Y, st)
                  : launch_spmm_linear<64>(rowptr, col, val, X, M, W, N, bias, act, alpha, Y, st);
}
