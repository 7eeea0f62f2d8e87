// Toy-bank construction on the device: the deterministic arithmetic of
//   InverseSampling.compute_sample_prob / pagerank_algorithm / degree_centrality_algorithm
//       RAGraph_node/ragraph_utils/InverseSampling.py:6-56 (dense); RAGraph_edge/modules/ragraph_utils/InverseSampling.py:6-62
//       (sparse, same recurrence with the dangling mass added explicitly)
//   PositionAwareEncoder.floyd_warshall / encode_position_aware_code   RAGraph_node/ragraph_utils/PositionAwareEncoder.py:6-48
// batched over resource graphs, with no host round trip inside: the reference runs them per 40-node graph in Python
// (a dense mat-vec and a host-side convergence test per power iteration; a Python double loop per position code).
//
// pagerank: the graphs of a batch are the segments [graph_ptr[g], graph_ptr[g+1]) of one block-diagonal CSR (a single
// big graph is one segment).  One power iteration = two launches: `step` (every node pulls its in-neighbours' mass: one
// fmaf chain in CSR order) and `check` (one workgroup per graph: L1 change in a fixed order, the reference's
// break-BEFORE-assign -- a converged graph keeps the previous iterate, as pagerank_algorithm returns it --, the
// dangling mass for the next step).  The host enqueues max_iter iterations back to back; converged graphs turn their
// launches into no-ops through a device flag, nothing is read back in between.
#include "common.h"

namespace ragraph {

constexpr int PR_THREADS = 256;

// deterministic block sum (fixed tree) of one value per thread
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = __fadd_rn(v, __shfl_xor(v, off));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  const float t = __fadd_rn(__fadd_rn(sh[0], sh[1]), __fadd_rn(sh[2], sh[3]));
  __syncthreads();
  return t;
}

// p = 1/N_g, done = 0, dangling[g] = (sum over the graph's zero-out-degree nodes of 1/N_g) / N_g
__global__ void __launch_bounds__(PR_THREADS) pagerank_init_kernel(const float* __restrict__ out_deg,
                                                                   const int64_t* __restrict__ graph_ptr,
                                                                   float* __restrict__ p, float* __restrict__ dangling,
                                                                   int* __restrict__ done, int* __restrict__ iters) {
  __shared__ float sh[4];
  const int g = blockIdx.x;
  const int64_t lo = graph_ptr[g], hi = graph_ptr[g + 1];
  const float n = (float)(hi - lo);
  const float p0 = 1.0f / n;
  float dang = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += PR_THREADS) {
    p[i] = p0;
    if (out_deg[i] == 0.f) dang = __fadd_rn(dang, p0);
  }
  dang = block_sum_256(dang, sh);
  if (threadIdx.x == 0) {
    dangling[g] = dang / n;
    done[g] = hi > lo ? 0 : 1;
    iters[g] = 0;
  }
}

// new_p[j] = (1 - d)/N + d * (sum_i adj[i][j] / out_deg[i] * p[i] + dangling)   for every node j of an unconverged graph
__global__ void __launch_bounds__(PR_THREADS) pagerank_step_kernel(const int64_t* __restrict__ rowptrT,
                                                                   const int32_t* __restrict__ colT,
                                                                   const float* __restrict__ valT,
                                                                   const float* __restrict__ out_deg,
                                                                   const int64_t* __restrict__ graph_ptr,
                                                                   const int32_t* __restrict__ graph_of, int64_t n, float d,
                                                                   const float* __restrict__ p, const float* __restrict__ dangling,
                                                                   const int* __restrict__ done, float* __restrict__ new_p) {
  const int64_t j = (int64_t)blockIdx.x * PR_THREADS + threadIdx.x;
  if (j >= n) return;
  const int g = graph_of[j];
  if (done[g]) return;
  const float ng = (float)(graph_ptr[g + 1] - graph_ptr[g]);
  float s = 0.f;
  for (int64_t e = rowptrT[j]; e < rowptrT[j + 1]; ++e) {
    const int i = colT[e];
    s = fmaf(valT[e] / out_deg[i], p[i], s);  // (a dangling node has no out-edge, so out_deg[i] != 0 here)
  }
  new_p[j] = __fadd_rn((1.0f - d) / ng, __fmul_rn(d, __fadd_rn(s, dangling[g])));
}

// one workgroup per graph: ||new_p - p||_1 < eps ? keep p and mark done : p = new_p; dangling mass of the new iterate
__global__ void __launch_bounds__(PR_THREADS) pagerank_check_kernel(const float* __restrict__ out_deg,
                                                                    const int64_t* __restrict__ graph_ptr, float eps,
                                                                    float* __restrict__ p, const float* __restrict__ new_p,
                                                                    float* __restrict__ dangling, int* __restrict__ done,
                                                                    int* __restrict__ iters) {
  __shared__ float sh[4];
  const int g = blockIdx.x;
  if (done[g]) return;
  const int64_t lo = graph_ptr[g], hi = graph_ptr[g + 1];
  float diff = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += PR_THREADS) diff = __fadd_rn(diff, fabsf(__fsub_rn(new_p[i], p[i])));
  diff = block_sum_256(diff, sh);
  if (diff < eps) {  // InverseSampling.py:41-43: break before `p = new_p`
    if (threadIdx.x == 0) done[g] = 1;
    return;
  }
  const float n = (float)(hi - lo);
  float dang = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += PR_THREADS) {
    const float v = new_p[i];
    p[i] = v;
    if (out_deg[i] == 0.f) dang = __fadd_rn(dang, v);
  }
  dang = block_sum_256(dang, sh);
  if (threadIdx.x == 0) {
    dangling[g] = dang / n;
    iters[g] += 1;
  }
}

// InverseSampling.compute_sample_prob (:6-19): importance = alpha * pagerank + (1 - alpha) * degree / (N - 1);
// prob = (1 / (importance + eps)) / sum over the graph.  One workgroup per graph, fixed-order sum.
__global__ void __launch_bounds__(PR_THREADS) sample_prob_kernel(const float* __restrict__ pagerank,
                                                                 const float* __restrict__ col_sum,
                                                                 const int64_t* __restrict__ graph_ptr, float alpha, float eps,
                                                                 float* __restrict__ prob) {
  __shared__ float sh[4];
  const int g = blockIdx.x;
  const int64_t lo = graph_ptr[g], hi = graph_ptr[g + 1];
  const float nm1 = (float)(hi - lo - 1);
  float tot = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += PR_THREADS) {
    const float imp = __fadd_rn(__fmul_rn(alpha, pagerank[i]), __fmul_rn(1.0f - alpha, col_sum[i] / nm1));
    const float inv = 1.0f / __fadd_rn(imp, eps);
    prob[i] = inv;
    tot = __fadd_rn(tot, inv);
  }
  tot = block_sum_256(tot, sh);
  for (int64_t i = lo + threadIdx.x; i < hi; i += PR_THREADS) prob[i] = prob[i] / tot;
}

// row sums of a CSR matrix (sequential fp32 adds in CSR order): out-degrees; on the transposed CSR: column sums
__global__ void __launch_bounds__(256) csr_row_sums_kernel(const int64_t* __restrict__ rowptr, const float* __restrict__ val,
                                                           int64_t n, float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  float s = 0.f;
  for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) s = __fadd_rn(s, val[e]);
  out[r] = s;
}

// Batched Floyd-Warshall + position codes for G small graphs of n <= 64 nodes each (the sampled toy graphs: n = 10): one
// workgroup per graph, the distance matrix in LDS.  dist init as PositionAwareEncoder.py:38-41 (0 -> inf, diagonal 0),
// then n min-plus steps (row k and column k are fixed points of step k, so the in-place update is race-free between
// barriers); code[u][a] = 1 / (dist[u][anchor_a] + 1) if that distance < dis_q else 0 (:14-22).
__global__ void __launch_bounds__(256) fw_position_batch_kernel(const float* __restrict__ adj, int n,
                                                                const int64_t* __restrict__ anchors, int A, float dis_q,
                                                                float* __restrict__ dist_out, float* __restrict__ code_out) {
  __shared__ float d[64 * 64];
  const int g = blockIdx.x;
  const float* a = adj + (int64_t)g * n * n;
  for (int e = threadIdx.x; e < n * n; e += 256) {
    const int i = e / n, j = e % n;
    const float v = a[e];
    d[e] = (i == j) ? 0.f : (v == 0.f ? __builtin_huge_valf() : v);
  }
  __syncthreads();
  for (int kk = 0; kk < n; ++kk) {
    for (int e = threadIdx.x; e < n * n; e += 256) {
      const int i = e / n, j = e % n;
      const float via = __fadd_rn(d[i * n + kk], d[kk * n + j]);
      if (via < d[e]) d[e] = via;
    }
    __syncthreads();
  }
  if (dist_out)
    for (int e = threadIdx.x; e < n * n; e += 256) dist_out[(int64_t)g * n * n + e] = d[e];
  for (int e = threadIdx.x; e < n * A; e += 256) {
    const int u = e / A, ai = e % A;
    const float dd = d[u * n + (int)anchors[(int64_t)g * A + ai]];
    code_out[(int64_t)g * n * A + e] = (dd < dis_q) ? 1.f / (dd + 1.f) : 0.f;
  }
}

}  // namespace ragraph

using namespace ragraph;

extern "C" size_t ragraph_pagerank_workspace_bytes(int64_t n, int64_t G) {
  return align_up((size_t)n * sizeof(float), 256) + align_up((size_t)G * sizeof(float), 256) +
         align_up((size_t)G * sizeof(int), 256);
}

extern "C" int ragraph_pagerank_f32(const int64_t* rowptrT, const int32_t* colT, const float* valT, const float* out_deg,
                                    const int64_t* graph_ptr, const int32_t* graph_of, int64_t G, int64_t n, float d, float eps,
                                    int max_iter, float* p, int* iters, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(rowptrT && colT && valT && out_deg && graph_ptr && graph_of && p && iters && ws, RAGRAPH_EINVAL,
             "pagerank: null pointer");
  RG_REQUIRE(G >= 1 && n >= 1 && max_iter >= 1, RAGRAPH_EINVAL, "pagerank: bad G/n/max_iter");
  RG_REQUIRE(ws_bytes >= ragraph_pagerank_workspace_bytes(n, G), RAGRAPH_EWORKSPACE, "pagerank: workspace too small");
  hipStream_t st = as_stream(stream);
  char* w = static_cast<char*>(ws);
  float* new_p = reinterpret_cast<float*>(w);
  float* dangling = reinterpret_cast<float*>(w + align_up((size_t)n * sizeof(float), 256));
  int* done = reinterpret_cast<int*>(reinterpret_cast<char*>(dangling) + align_up((size_t)G * sizeof(float), 256));
  hipLaunchKernelGGL(pagerank_init_kernel, dim3((unsigned)G), dim3(PR_THREADS), 0, st, out_deg, graph_ptr, p, dangling, done,
                     iters);
  for (int it = 0; it < max_iter; ++it) {
    hipLaunchKernelGGL(pagerank_step_kernel, dim3((unsigned)cdiv(n, PR_THREADS)), dim3(PR_THREADS), 0, st, rowptrT, colT, valT,
                       out_deg, graph_ptr, graph_of, n, d, p, dangling, done, new_p);
    hipLaunchKernelGGL(pagerank_check_kernel, dim3((unsigned)G), dim3(PR_THREADS), 0, st, out_deg, graph_ptr, eps, p, new_p,
                       dangling, done, iters);
  }
  RG_CHECK_LAUNCH("pagerank");
  return RAGRAPH_OK;
}

extern "C" int ragraph_sample_prob_f32(const float* pagerank, const float* col_sum, const int64_t* graph_ptr, int64_t G,
                                       float alpha, float eps, float* prob, void* stream) {
  RG_REQUIRE(pagerank && col_sum && graph_ptr && prob, RAGRAPH_EINVAL, "sample_prob: null pointer");
  if (G <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(sample_prob_kernel, dim3((unsigned)G), dim3(PR_THREADS), 0, as_stream(stream), pagerank, col_sum,
                     graph_ptr, alpha, eps, prob);
  RG_CHECK_LAUNCH("sample_prob");
  return RAGRAPH_OK;
}

extern "C" int ragraph_csr_row_sums_f32(const int64_t* rowptr, const float* val, int64_t n, float* out, void* stream) {
  RG_REQUIRE(rowptr && val && out, RAGRAPH_EINVAL, "csr_row_sums: null pointer");
  if (n <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(csr_row_sums_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), rowptr, val, n, out);
  RG_CHECK_LAUNCH("csr_row_sums");
  return RAGRAPH_OK;
}

extern "C" int ragraph_position_codes_batch_f32(const float* adj, int64_t G, int n, const int64_t* anchors, int A, float dis_q,
                                                float* dist_out, float* codes, void* stream) {
  RG_REQUIRE(adj && anchors && codes, RAGRAPH_EINVAL, "position_codes_batch: null pointer");
  RG_REQUIRE(n >= 1 && n <= 64 && A >= 1, RAGRAPH_EUNSUPPORTED, "position_codes_batch: n=%d not in [1,64]", n);
  if (G <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(fw_position_batch_kernel, dim3((unsigned)G), dim3(256), 0, as_stream(stream), adj, n, anchors, A, dis_q,
                     dist_out, codes);
  RG_CHECK_LAUNCH("position_codes_batch");
  return RAGRAPH_OK;
}
