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
  __shared__ float As[LBM * LLD];
  __shared__ float Bs[LBN * LLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * LBM;
  const int64_t n0 = (int64_t)blockIdx.y * LBN;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int k0 = 0; k0 < K; k0 += LKC) {
    // stage: 64 x 32 of X and of W, zero-padded outside (0*0 leaves a chain unchanged)
#pragma unroll
    for (int u = 0; u < (LBM * LKC) / 256; ++u) {
      const int e = tid + u * 256;
      const int row = e / LKC, c = e % LKC;
      const int kk = k0 + c;
      const int64_t gm = m0 + row, gn = n0 + row;
      As[row * LLD + c] = (gm < M && kk < K) ? X[gm * K + kk] : 0.f;
      Bs[row * LLD + c] = (gn < N && kk < K) ? W[gn * K + kk] : 0.f;
    }
    __syncthreads();
    const float* a = As + (wr * 32 + j) * LLD + h;
    const float* b = Bs + (wc * 32 + j) * LLD + h;
#pragma unroll
    for (int kk = 0; kk < LKC / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk], b[2 * kk], acc, 0, 0, 0);
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

}  // namespace ragraph

using namespace ragraph;

extern "C" int ragraph_linear_f32(const float* X, int64_t M, int K, const float* W, int64_t N, const float* bias,
                                  int act, float alpha, float* Y, void* stream) {
  RG_REQUIRE(X && W && Y, RAGRAPH_EINVAL, "linear: null pointer");
  RG_REQUIRE(M >= 1 && N >= 1 && K >= 1, RAGRAPH_EINVAL, "linear: M,N,K must be >= 1");
  RG_REQUIRE(act >= RAGRAPH_ACT_NONE && act <= RAGRAPH_ACT_ELU, RAGRAPH_EINVAL, "linear: bad act %d", act);
  RG_REQUIRE(cdiv(N, LBN) <= 65535, RAGRAPH_EUNSUPPORTED, "linear: N=%lld too large for one launch", (long long)N);
  dim3 grid((unsigned)cdiv(M, LBM), (unsigned)cdiv(N, LBN));
  hipLaunchKernelGGL(linear_kernel, grid, dim3(256), 0, as_stream(stream), X, M, K, W, N, bias, act, alpha, Y);
  RG_CHECK_LAUNCH("linear");
  return RAGRAPH_OK;
}
