// K5 + K6 in ONE launch: prompt fusion, the two-layer task decoder and the label mix of an inference forward --
//   hidden = query * (1 - w) + rag * w                     RAGraph_node/RAGraph.py:53   (ragraph_axpby_f32)
//   logits = fc2(LeakyReLU(fc1(hidden)))                   ragraph_utils/TaskDecoder.py:14-17 (ragraph_linear_f32 x 2)
//   out    = softmax(logits) * (1 - l) + rag_label * l     RAGraph_node/RAGraph.py:55-57 (ragraph_softmax_mix_f32)
// The four separate launches are what a Cora-sized or PROTEINS-sized forward spends its time on (launch-bound: ~8 us of
// work); a workgroup here takes TR rows through all four steps with the intermediates in LDS.  Every value is computed
// by the same operations in the same order as the separate entries (k-ascending fmaf chains from +0, bias added after,
// two multiplies and an add for the mixes), so the result has the same bits.  Large batches (the MFMA tile kernel of
// ragraph_linear_f32 is the faster fc1 there) keep the separate launches: the host side chooses (ragraph_amd/RAGraph.py).
#include "common.h"

namespace ragraph {

constexpr int DECODE_TR = 4;  // rows per workgroup

__global__ void __launch_bounds__(256) fuse_decode_kernel(const float* __restrict__ query, const float* __restrict__ rag,
                                                          int64_t n, int D, float wq, float wr,
                                                          const float* __restrict__ W1, const float* __restrict__ b1, int H,
                                                          float slope, const float* __restrict__ W2,
                                                          const float* __restrict__ b2, int C,
                                                          const float* __restrict__ rag_label, float lambda,
                                                          float* __restrict__ out) {
  extern __shared__ float4 decode_smem4[];
  float* hid = reinterpret_cast<float*>(decode_smem4);  // [TR][Dp]
  const int Dp = (D + 3) & ~3, Hp = (H + 3) & ~3;
  float* h1 = hid + DECODE_TR * Dp;                      // [TR][Hp]
  float* lg = h1 + DECODE_TR * Hp;                       // [TR][C]
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * DECODE_TR;
  const int rows = n - row0 < DECODE_TR ? (int)(n - row0) : DECODE_TR;

  // ---- hidden = query * wq + rag * wr (absent rows: zeros, never stored) ----
  for (int i = tid; i < DECODE_TR * Dp; i += 256) {
    const int r = i / Dp, d = i - r * Dp;
    float v = 0.f;
    if (r < rows && d < D) {
      const int64_t g = (row0 + r) * D + d;
      v = __fadd_rn(__fmul_rn(query[g], wq), __fmul_rn(rag[g], wr));
    }
    hid[i] = v;
  }
  __syncthreads();

  // ---- fc1 + LeakyReLU: thread -> output column j, TR chains at once (one W1 element feeds TR fmaf) ----
  const bool vec1 = (D & 3) == 0;
  for (int j = tid; j < H; j += 256) {
    float acc[DECODE_TR];
#pragma unroll
    for (int r = 0; r < DECODE_TR; ++r) acc[r] = 0.f;
    const float* w = W1 + (int64_t)j * D;
    if (vec1) {
      for (int k = 0; k < D; k += 4) {
        const float4 wv = *reinterpret_cast<const float4*>(w + k);
#pragma unroll
        for (int r = 0; r < DECODE_TR; ++r) {
          const float4 hv = *reinterpret_cast<const float4*>(hid + r * Dp + k);
          acc[r] = fmaf(hv.x, wv.x, acc[r]);
          acc[r] = fmaf(hv.y, wv.y, acc[r]);
          acc[r] = fmaf(hv.z, wv.z, acc[r]);
          acc[r] = fmaf(hv.w, wv.w, acc[r]);
        }
      }
    } else {
      for (int k = 0; k < D; ++k) {
        const float wv = w[k];
#pragma unroll
        for (int r = 0; r < DECODE_TR; ++r) acc[r] = fmaf(hid[r * Dp + k], wv, acc[r]);
      }
    }
    const float bv = b1 ? b1[j] : 0.f;
#pragma unroll
    for (int r = 0; r < DECODE_TR; ++r) {
      float v = acc[r];
      if (b1) v = __fadd_rn(v, bv);
      h1[r * Hp + j] = apply_act(v, RAGRAPH_ACT_LEAKY, slope);
    }
  }
  __syncthreads();

  // ---- fc2: one (row, class) chain per thread ----
  for (int o = tid; o < DECODE_TR * C; o += 256) {
    const int r = o / C, c = o - r * C;
    const float* w = W2 + (int64_t)c * H;
    float acc = 0.f;
    for (int k = 0; k < H; ++k) acc = fmaf(h1[r * Hp + k], w[k], acc);
    if (b2) acc = __fadd_rn(acc, b2[c]);
    lg[o] = acc;
  }
  __syncthreads();

  // ---- softmax * (1 - lambda) + rag_label * lambda: one thread per row (softmax_mix_kernel's order) ----
  if (tid < rows) {
    const float* x = lg + tid * C;
    const int64_t b = row0 + tid;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = __fadd_rn(s, expf(x[c] - m));
    const float one_m = 1.f - lambda;
    for (int c = 0; c < C; ++c) {
      float p = expf(x[c] - m) / s;
      if (rag_label) p = __fadd_rn(__fmul_rn(p, one_m), __fmul_rn(rag_label[b * C + c], lambda));
      out[b * C + c] = p;
    }
  }
}

}  // namespace ragraph

using namespace ragraph;

extern "C" int ragraph_fuse_decode_f32(const float* query, const float* rag, int64_t n, int D, float wq, float wr,
                                       const float* W1, const float* b1, int H, float slope, const float* W2,
                                       const float* b2, int C, const float* rag_label, float lambda, float* out,
                                       void* stream) {
  RG_REQUIRE(query && rag && W1 && W2 && out, RAGRAPH_EINVAL, "fuse_decode: null pointer");
  RG_REQUIRE(D >= 1 && H >= 1 && C >= 1, RAGRAPH_EINVAL, "fuse_decode: bad D/H/C");
  const size_t lds = (size_t)DECODE_TR * (((D + 3) & ~3) + ((H + 3) & ~3) + C) * sizeof(float);
  RG_REQUIRE(lds <= 64 * 1024, RAGRAPH_EUNSUPPORTED, "fuse_decode: D=%d H=%d C=%d exceed the LDS tile", D, H, C);
  RG_REQUIRE((D & 3) != 0 || (aligned16(W1)), RAGRAPH_EINVAL, "fuse_decode: W1 must be 16-B aligned");
  if (n <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(fuse_decode_kernel, dim3((unsigned)cdiv(n, (int64_t)DECODE_TR)), dim3(256), lds, as_stream(stream), query,
                     rag, n, D, wq, wr, W1, b1, H, slope, W2, b2, C, rag_label, lambda, out);
  RG_CHECK_LAUNCH("fuse_decode");
  return RAGRAPH_OK;
}
