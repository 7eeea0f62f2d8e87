// CSR neighbour aggregation and per-segment reductions (HBM / gather-bound; no MFMA: integer-indexed row gathers).
//
//   spmm_csr           PReLU(A_hat (X W^T) + b)            layers/gcn.py:36-40
//                      relu(A_tilde x)  (k hops)           ragraph_utils/Propagation.py:19-25
//                      scatter_sum(emb[src]*norm, dst)     RAGraph_edge/modules/RAGraph.py:232-240, modules/utils.py:17-32
//   csr_row_normalize  adj / adj.sum(1)                    Propagation.py:15-16
//   segment_softmax    scatter_softmax(times, dst)         RAGraph_edge/modules/RAGraph.py:261
//   segment_reduce     mean(dim=0) / per-graph sum readout RAGraph_graph/RAGraph.py:50,63; downprompt.py:98-112
//
// The reference multiplies a DENSE n x n adjacency (layers/gcn.py:36) -- 99.99 % zeros at n = 1e5.  Here a row's
// neighbours are walked once: LPR = D/4 lanes hold one output row as float4 (one wave = one 1 KiB row at D = 256, so
// every neighbour gather is a single coalesced 1 KiB wave-instruction), edges are consumed four at a time so four
// row gathers are in flight per wave, and bias / activation / residual are fused into the store.  No atomics: the
// sum is one fmaf chain in CSR order, deterministic and bit-identical to the oracle.
#include "common.h"

namespace ragraph {

// A row's edges are consumed in chunks of CH = 16: the first lanes of the row's lane group load the chunk's (col, val)
// pairs with ONE coalesced load each, the group broadcasts them, and all gathers of the chunk are issued back to back
// before the first fmaf -- three dependent memory latencies per row (rowptr, edge list, X rows) instead of one pair per
// four edges.  The sum is still the sequential fmaf chain in CSR order.
template <int LPR>
__global__ void __launch_bounds__(256) spmm_csr_kernel(const int64_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col,
                                                       const float* __restrict__ val, int64_t n,
                                                       const float* __restrict__ X, int D,
                                                       const float* __restrict__ bias, int act, float alpha, float beta,
                                                       const float* __restrict__ Yin, float* __restrict__ Y) {
  constexpr int RPB = 256 / LPR;  // rows per block
  constexpr int CH = 16;          // edges per chunk (<= LPR)
  const int lr = threadIdx.x % LPR;
  const int gbase = (threadIdx.x & 63) - lr;  // first lane of this row's group inside the wave
  int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
  const bool live = row < n;  // dead groups run along with zero edges: the shuffles need every lane
  if (!live) row = n - 1;
  const int64_t e0 = rowptr[row];
  const int deg = live ? (int)(rowptr[row + 1] - e0) : 0;
  const int D4 = D >> 2;
  const float4* X4 = reinterpret_cast<const float4*>(X);

  for (int c4 = lr; c4 < ((D4 + LPR - 1) / LPR) * LPR; c4 += LPR) {  // uniform trip count: shuffles inside
    const bool colok = c4 < D4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < deg; base += CH) {
      int my_c = 0;
      float my_v = 0.f;
      if (lr < CH && base + lr < deg) {
        my_c = col[e0 + base + lr];
        my_v = val[e0 + base + lr];
      }
      const int cnt = deg - base < CH ? deg - base : CH;
      float4 x[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = __shfl(my_c, gbase + i);
        x[i] = (i < cnt && colok) ? X4[(int64_t)c * D4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const float v = __shfl(my_v, gbase + i);
        if (i < cnt) {
          acc.x = fmaf(v, x[i].x, acc.x); acc.y = fmaf(v, x[i].y, acc.y);
          acc.z = fmaf(v, x[i].z, acc.z); acc.w = fmaf(v, x[i].w, acc.w);
        }
      }
    }
    if (!live || !colok) continue;
    if (bias) {
      const float4 b = reinterpret_cast<const float4*>(bias)[c4];
      acc.x = __fadd_rn(acc.x, b.x); acc.y = __fadd_rn(acc.y, b.y); acc.z = __fadd_rn(acc.z, b.z); acc.w = __fadd_rn(acc.w, b.w);
    }
    acc.x = apply_act(acc.x, act, alpha); acc.y = apply_act(acc.y, act, alpha);
    acc.z = apply_act(acc.z, act, alpha); acc.w = apply_act(acc.w, act, alpha);
    if (Yin) {
      const float4 y = reinterpret_cast<const float4*>(Yin)[row * D4 + c4];
      acc.x = fmaf(beta, y.x, acc.x); acc.y = fmaf(beta, y.y, acc.y); acc.z = fmaf(beta, y.z, acc.z); acc.w = fmaf(beta, y.w, acc.w);
    }
    reinterpret_cast<float4*>(Y)[row * D4 + c4] = acc;
  }
}

// One thread per row: rows are short (mean degree ~10) and the sum must be sequential to match the oracle.
__global__ void __launch_bounds__(256) csr_row_normalize_kernel(const int64_t* __restrict__ rowptr,
                                                                const float* __restrict__ val, int64_t n,
                                                                float* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  float s = 0.f;
  for (int64_t e = e0; e < e1; ++e) s = __fadd_rn(s, val[e]);
  for (int64_t e = e0; e < e1; ++e) out[e] = val[e] / s;
}

__global__ void __launch_bounds__(256) segment_softmax_kernel(const int64_t* __restrict__ rowptr,
                                                              const float* __restrict__ x, int64_t n,
                                                              float* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  if (e0 == e1) return;
  float m = x[e0];
  for (int64_t e = e0 + 1; e < e1; ++e) m = fmaxf(m, x[e]);
  float s = 0.f;
  for (int64_t e = e0; e < e1; ++e) {
    const float ex = expf(x[e] - m);
    out[e] = ex;
    s = __fadd_rn(s, ex);
  }
  for (int64_t e = e0; e < e1; ++e) out[e] = out[e] / s;
}

// One workgroup per segment; thread t owns float4 chunks t, t+256, ...; rows are summed sequentially (oracle order).
__global__ void __launch_bounds__(256) segment_reduce_kernel(const float* __restrict__ X, int D,
                                                             const int64_t* __restrict__ seg_ptr,
                                                             const float* __restrict__ w, int mean_mode,
                                                             float* __restrict__ out) {
  const int64_t g = blockIdx.x;
  const int64_t r0 = seg_ptr[g], r1 = seg_ptr[g + 1];
  const int D4 = D >> 2;
  const float4* X4 = reinterpret_cast<const float4*>(X);
  for (int c4 = threadIdx.x; c4 < D4; c4 += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 wv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (w) wv = reinterpret_cast<const float4*>(w)[c4];
    for (int64_t r = r0; r < r1; ++r) {
      float4 x = X4[r * D4 + c4];
      if (w) { x.x = __fmul_rn(wv.x, x.x); x.y = __fmul_rn(wv.y, x.y); x.z = __fmul_rn(wv.z, x.z); x.w = __fmul_rn(wv.w, x.w); }
      acc.x = __fadd_rn(acc.x, x.x); acc.y = __fadd_rn(acc.y, x.y); acc.z = __fadd_rn(acc.z, x.z); acc.w = __fadd_rn(acc.w, x.w);
    }
    if (mean_mode) {
      const float len = (float)(r1 - r0);
      acc.x = acc.x / len; acc.y = acc.y / len; acc.z = acc.z / len; acc.w = acc.w / len;
    }
    reinterpret_cast<float4*>(out)[g * D4 + c4] = acc;
  }
}

}  // namespace ragraph

using namespace ragraph;

extern "C" int ragraph_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n,
                                    const float* X, int D, const float* bias, int act, float alpha, float beta,
                                    const float* Y_in, float* Y, void* stream) {
  RG_REQUIRE(rowptr && X && Y, RAGRAPH_EINVAL, "spmm_csr: null pointer");
  RG_REQUIRE(n >= 0 && D >= 4 && (D & 3) == 0, RAGRAPH_EINVAL, "spmm_csr: D=%d must be a positive multiple of 4", D);
  RG_REQUIRE(aligned16(X) && aligned16(Y) && (!bias || aligned16(bias)) && (!Y_in || aligned16(Y_in)), RAGRAPH_EINVAL,
             "spmm_csr: X, Y, bias, Y_in must be 16-B aligned");
  RG_REQUIRE(X != Y, RAGRAPH_EINVAL, "spmm_csr: Y must not alias X");
  RG_REQUIRE(act >= RAGRAPH_ACT_NONE && act <= RAGRAPH_ACT_ELU, RAGRAPH_EINVAL, "spmm_csr: bad act %d", act);
  if (n == 0) return RAGRAPH_OK;
  hipStream_t st = as_stream(stream);
  const int D4 = D >> 2;
  if (D4 <= 16) {
    hipLaunchKernelGGL(spmm_csr_kernel<16>, dim3((unsigned)cdiv(n, 16)), dim3(256), 0, st, rowptr, col, val, n, X, D,
                       bias, act, alpha, beta, Y_in, Y);
  } else if (D4 <= 32) {
    hipLaunchKernelGGL(spmm_csr_kernel<32>, dim3((unsigned)cdiv(n, 8)), dim3(256), 0, st, rowptr, col, val, n, X, D,
                       bias, act, alpha, beta, Y_in, Y);
  } else {
    hipLaunchKernelGGL(spmm_csr_kernel<64>, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, st, rowptr, col, val, n, X, D,
                       bias, act, alpha, beta, Y_in, Y);
  }
  RG_CHECK_LAUNCH("spmm_csr");
  return RAGRAPH_OK;
}

extern "C" int ragraph_csr_row_normalize_f32(const int64_t* rowptr, const float* val, int64_t n, float* val_out,
                                             void* stream) {
  RG_REQUIRE(rowptr && val && val_out, RAGRAPH_EINVAL, "csr_row_normalize: null pointer");
  if (n <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(csr_row_normalize_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), rowptr,
                     val, n, val_out);
  RG_CHECK_LAUNCH("csr_row_normalize");
  return RAGRAPH_OK;
}

extern "C" int ragraph_segment_softmax_f32(const int64_t* rowptr, const float* x, int64_t n, float* out,
                                           void* stream) {
  RG_REQUIRE(rowptr && x && out, RAGRAPH_EINVAL, "segment_softmax: null pointer");
  if (n <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(segment_softmax_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), rowptr, x,
                     n, out);
  RG_CHECK_LAUNCH("segment_softmax");
  return RAGRAPH_OK;
}

extern "C" int ragraph_segment_reduce_f32(const float* X, int D, const int64_t* seg_ptr, int64_t G, const float* w,
                                          int mean_mode, float* out, void* stream) {
  RG_REQUIRE(X && seg_ptr && out, RAGRAPH_EINVAL, "segment_reduce: null pointer");
  RG_REQUIRE(D >= 4 && (D & 3) == 0, RAGRAPH_EINVAL, "segment_reduce: D=%d must be a positive multiple of 4", D);
  RG_REQUIRE(aligned16(X) && aligned16(out) && (!w || aligned16(w)), RAGRAPH_EINVAL, "segment_reduce: 16-B alignment");
  if (G <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(segment_reduce_kernel, dim3((unsigned)G), dim3(256), 0, as_stream(stream), X, D, seg_ptr, w,
                     mean_mode, out);
  RG_CHECK_LAUNCH("segment_reduce");
  return RAGRAPH_OK;
}
