// Row-wise helpers around the two big kernels: L2 normalisation, winner gathers, fusion arithmetic, class softmax.
// All are HBM-bound elementwise / gather work: coalesced float4 (16 B per lane) accesses, one wave per row.
#include <stdarg.h>

#include "common.h"

namespace ragraph {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- a1: F.normalize(x, p=2, dim=-1, eps=1e-12) (SimilarityFunctions.py:8,11) ---------------------------------------
// One wave per row.  Norm tree (the oracle restates it): lane l accumulates, with fmaf, the squares of the elements of
// its float4 chunks c = l, l+64, ... (element order inside a chunk 0..3), then a 6-step xor butterfly 32,16,...,1.
template <bool VEC4>
__global__ void __launch_bounds__(256) normalize_rows_kernel(const float* __restrict__ X, int64_t n, int D,
                                                             float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* x = X + row * D;
  float* o = out + row * D;
  const int nchunk = (D + 3) >> 2;
  float p = 0.f;
  if (VEC4) {
    for (int c = lane; c < nchunk; c += 64) {
      const float4 v = reinterpret_cast<const float4*>(x)[c];
      p = fmaf(v.x, v.x, p);
      p = fmaf(v.y, v.y, p);
      p = fmaf(v.z, v.z, p);
      p = fmaf(v.w, v.w, p);
    }
  } else {
    for (int c = lane; c < nchunk; c += 64) {
      for (int e = 4 * c; e < 4 * c + 4 && e < D; ++e) p = fmaf(x[e], x[e], p);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) p = __fadd_rn(p, __shfl_xor(p, off));
  const float d = fmaxf(sqrtf(p), 1e-12f);
  if (VEC4) {
    for (int c = lane; c < nchunk; c += 64) {
      float4 v = reinterpret_cast<const float4*>(x)[c];
      v.x = v.x / d; v.y = v.y / d; v.z = v.z / d; v.w = v.w / d;
      reinterpret_cast<float4*>(o)[c] = v;
    }
  } else {
    for (int e = lane; e < D; e += 64) o[e] = x[e] / d;
  }
}

// ---- a2: V[idx] (ToyGraphBase.py:70-71) -----------------------------------------------------------------------------
template <bool VEC4>
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ V, int64_t N, int D,
                                                          const int64_t* __restrict__ idx, int64_t M, int64_t base,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int64_t r = idx[m] - base;
  const bool own = (r >= 0 && r < N);
  if (VEC4) {
    const int D4 = D >> 2;
    for (int c = lane; c < D4; c += 64) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (own) v = reinterpret_cast<const float4*>(V + r * D)[c];
      reinterpret_cast<float4*>(out + m * D)[c] = v;
    }
  } else {
    for (int e = lane; e < D; e += 64) out[m * D + e] = own ? V[r * D + e] : 0.f;
  }
}

// ---- a8: sum_k V[idx], mean_k L[idx] (RAGraph.py:48-49); one wave per query, winners added in rank order -------------
// Gather-bound (B*k random 1 KiB rows): 16 B per lane so one wave-instruction fetches a whole row at D = 256.  The wave
// reads its k indices FIRST (one coalesced load, lane u = winner u; blocks of 64 for larger k) and hands them out through
// v_readlane: a row's address is then a scalar base + the lane's offset and the loads of 8 winners go out back to back.
// (Round 6: with each index loaded right before its row the compiler's s_waitcnt vmcnt(0) for the index also waited for the
// row before it -- ONE gather in flight per wave where the source said eight; 218 -> see DESIGN.md section 4.7.)  Winners outside
// [base, base + N) -- another shard's -- read row 0 and add +0, which leaves every partial sum unchanged.
template <bool VEC4>
__global__ void __launch_bounds__(256) gather_reduce_kernel(const float* __restrict__ V, int D,
                                                            const float* __restrict__ L, int C, int64_t N,
                                                            const int64_t* __restrict__ idx, int64_t B, int k,
                                                            int64_t base, float v_scale, float* __restrict__ sumV,
                                                            float* __restrict__ meanL, const float* __restrict__ mixA,
                                                            float wa, float wb) {
  // mixA: the row written is mixA[b] * wa + (v_scale * sum) * wb -- ragraph_axpby_f32 behind the reduction (two multiplies
  // and an add, uncontracted), without the sum in memory (ragraph_gather_reduce_mix_f32)
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int64_t* ib = idx + b * k;
  auto row_of = [&](int64_t mine, int u) {   // winner u of the block whose rows the lanes hold (u wave-uniform); -1: not mine
    return ((int64_t)__builtin_amdgcn_readlane((int)(mine >> 32), u) << 32) | (unsigned)__builtin_amdgcn_readlane((int)mine, u);
  };
  auto load_block = [&](int jb) {            // lane u: row of winner jb + u inside this shard, or -1
    int64_t r = -1;
    if (jb + lane < k) {
      r = ib[jb + lane] - base;
      if (r < 0 || r >= N) r = -1;
    }
    return r;
  };
  if (VEC4) {
    const int D4 = D >> 2;
    for (int c = lane; c < ((D4 + 63) / 64) * 64; c += 64) {   // (uniform trip count: readlanes inside)
      const bool colok = c < D4;
      const int cc = colok ? c : 0;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int jb = 0; jb < k && N > 0; jb += 64) {   // (an empty shard has no row 0 to read)
        const int64_t mine = load_block(jb);
        const int nb = k - jb < 64 ? k - jb : 64;
        for (int j0 = 0; j0 < nb; j0 += 8) {
          float4 v[8];
          bool ok[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int64_t r = j0 + u < nb ? row_of(mine, j0 + u) : -1;
            ok[u] = r >= 0;
            v[u] = reinterpret_cast<const float4*>(V + (ok[u] ? r : 0) * D)[cc];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (j0 + u < nb) {
              const float4 x = ok[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
              acc.x = __fadd_rn(acc.x, x.x); acc.y = __fadd_rn(acc.y, x.y);
              acc.z = __fadd_rn(acc.z, x.z); acc.w = __fadd_rn(acc.w, x.w);
            }
          }
        }
      }
      if (v_scale != 1.f) {
        acc.x = __fmul_rn(acc.x, v_scale); acc.y = __fmul_rn(acc.y, v_scale);
        acc.z = __fmul_rn(acc.z, v_scale); acc.w = __fmul_rn(acc.w, v_scale);
      }
      if (mixA && colok) {
        const float4 a = reinterpret_cast<const float4*>(mixA + b * D)[c];
        acc.x = __fadd_rn(__fmul_rn(a.x, wa), __fmul_rn(acc.x, wb)); acc.y = __fadd_rn(__fmul_rn(a.y, wa), __fmul_rn(acc.y, wb));
        acc.z = __fadd_rn(__fmul_rn(a.z, wa), __fmul_rn(acc.z, wb)); acc.w = __fadd_rn(__fmul_rn(a.w, wa), __fmul_rn(acc.w, wb));
      }
      if (colok) reinterpret_cast<float4*>(sumV + b * D)[c] = acc;
    }
  } else {
    for (int e = lane; e < D; e += 64) {
      float acc = 0.f;
      for (int jx = 0; jx < k; ++jx) {
        const int64_t r = ib[jx] - base;
        if (r >= 0 && r < N) acc = __fadd_rn(acc, V[r * D + e]);
      }
      acc = (v_scale == 1.f) ? acc : __fmul_rn(acc, v_scale);
      sumV[b * D + e] = mixA ? __fadd_rn(__fmul_rn(mixA[b * D + e], wa), __fmul_rn(acc, wb)) : acc;
    }
  }
  if (L && meanL) {
    for (int c = lane; c < ((C + 63) / 64) * 64; c += 64) {
      const bool colok = c < C;
      float acc = 0.f;
      for (int jb = 0; jb < k; jb += 64) {
        const int64_t mine = load_block(jb);
        const int nb = k - jb < 64 ? k - jb : 64;
        for (int jx = 0; jx < nb; ++jx) {
          const int64_t r = row_of(mine, jx);
          if (r >= 0 && colok) acc = __fadd_rn(acc, L[r * C + c]);
        }
      }
      if (colok) meanL[b * C + c] = acc / (float)k;
    }
  }
}

// ---- a8: hidden = a*wa + b*wb, uncontracted (RAGraph.py:53) ---------------------------------------------------------
__global__ void __launch_bounds__(256) axpby_kernel(const float* __restrict__ a, float wa, const float* __restrict__ b,
                                                    float wb, int64_t n, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    out[i] = __fadd_rn(__fmul_rn(a[i], wa), __fmul_rn(b[i], wb));
}

// the same with the weights read from device memory (a trainable [1, 2] mixing weight, RAGraph_node/downprompt.py:100-114: no
// host read-back): out = a * w[ia] + b * w[ib]; an index < 0 stands for a zero weight
__global__ void __launch_bounds__(256) axpby_dev_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ w, int ia, int ib, int64_t n,
                                                        float* __restrict__ out) {
  const float wa = ia >= 0 ? w[ia] : 0.f, wb = ib >= 0 ? w[ib] : 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    out[i] = __fadd_rn(__fmul_rn(a[i], wa), __fmul_rn(b[i], wb));
}

// ---- a13: emb_gate  x * sigmoid(z)  (RAGraph_edge/modules/RAGraph.py:168; z = x @ W + b from the linear kernel) --------
__global__ void __launch_bounds__(256) sigmoid_gate_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                                           int64_t n, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    out[i] = __fmul_rn(x[i], 1.f / (1.f + expf(-z[i])));
}

// ---- backward of the fused epilogues (fine-tuning: RAGraph_node_fewshot/RAGraph.py:69 trains the decode layer through
// its SpMM; RAGraph_edge/modules/RAGraph.py:335-355 trains embeddings / gate through the propagation) -------------------
// gz = gy * act'(z) expressed through the OUTPUT y (sign(y) = sign(z) for the leaky family; ELU: y + alpha for y < 0);
// t (optional, PReLU): the slope's gradient terms gy * z for z < 0, z = y / alpha.
__global__ void __launch_bounds__(256) act_grad_kernel(const float* __restrict__ y, const float* __restrict__ gy, int64_t n,
                                                       int act, float alpha, float* __restrict__ gz, float* __restrict__ t) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float yv = y[i], g = gy[i];
    float d = 1.f;
    if (act == RAGRAPH_ACT_RELU) d = yv > 0.f ? 1.f : 0.f;
    else if (act == RAGRAPH_ACT_PRELU || act == RAGRAPH_ACT_LEAKY) d = yv >= 0.f ? 1.f : alpha;
    else if (act == RAGRAPH_ACT_ELU) d = yv > 0.f ? 1.f : yv + alpha;
    gz[i] = g * d;
    if (t) t[i] = (yv < 0.f && alpha != 0.f) ? g * (yv / alpha) : 0.f;
  }
}

// emb_gate backward: out = x * s, s = sigmoid(z):  gx = g * s,  gz = g * x * s * (1 - s)
__global__ void __launch_bounds__(256) sigmoid_gate_grad_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                                                const float* __restrict__ g, int64_t n, float* __restrict__ gx,
                                                                float* __restrict__ gz) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float sg = 1.f / (1.f + expf(-z[i]));
    gx[i] = g[i] * sg;
    gz[i] = g[i] * x[i] * sg * (1.f - sg);
  }
}

// softmax backward (RAGraph.py:55-57): out = p * (g - sum_c g p), g = go * scale; one thread per row
__global__ void __launch_bounds__(256) softmax_grad_kernel(const float* __restrict__ p, const float* __restrict__ go, int64_t B,
                                                           int C, float scale, float* __restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float dot = 0.f;
  for (int c = 0; c < C; ++c) dot = fmaf(go[b * C + c] * scale, p[b * C + c], dot);
  for (int c = 0; c < C; ++c) out[b * C + c] = p[b * C + c] * (go[b * C + c] * scale - dot);
}

// downstreamprompt.forward: out[r,:] = act(x[r,:] * w[:]) -- act none in the graph flavour (RAGraph_graph/downprompt.py:
// 164-168), ELU in the node flavour (RAGraph_node/downprompt.py:118-130)
__global__ void __launch_bounds__(256) mul_cols_kernel(const float* __restrict__ x, const float* __restrict__ w, int64_t n, int D,
                                                       int act, float alpha, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n * D; i += stride)
    out[i] = apply_act(__fmul_rn(x[i], w[i % D]), act, alpha);
}

// elementwise product (the prompt weight's gradient = column sums of g * x: mul + segment_reduce)
__global__ void __launch_bounds__(256) mul_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                  float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = __fmul_rn(a[i], b[i]);
}

// ---- a12: (t - t_min) / (t_max - t_min) on int64 time steps (RAGraph_edge/modules/RAGraph.py:254-257) ----------------
__global__ void __launch_bounds__(256) time_rescale_kernel(const int64_t* __restrict__ t, int64_t n, float tmin,
                                                           float tmax, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  const float den = tmax - tmin;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = ((float)t[i] - tmin) / den;
}

// ---- a8: softmax(logits)*(1-lambda) + rag_label*lambda (RAGraph.py:55-57); one thread per row, C is a class count ----
__global__ void __launch_bounds__(256) softmax_mix_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ rag, int64_t B, int C,
                                                          float lambda, int log_mode, float* __restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float* x = logits + b * C;
  float m = x[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s = __fadd_rn(s, expf(x[c] - m));
  const float one_m = 1.f - lambda;
  const float ls = logf(s);
  for (int c = 0; c < C; ++c) {
    float p;
    if (log_mode)
      p = (x[c] - m) - ls;
    else
      p = expf(x[c] - m) / s;
    if (rag) p = __fadd_rn(__fmul_rn(p, one_m), __fmul_rn(rag[b * C + c], lambda));
    out[b * C + c] = p;
  }
}

// ---- a10: cosine to class prototypes (+softmax / log_softmax), downprompt.py:41-56; one wave per embedding ----------
__global__ void __launch_bounds__(256) proto_cosine_kernel(const float* __restrict__ emb, int64_t G, int D,
                                                           const float* __restrict__ proto, int C, int mode,
                                                           float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= G) return;
  const float* x = emb + g * D;
  float xx = 0.f;
  for (int e = lane; e < D; e += 64) xx = fmaf(x[e], x[e], xx);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) xx = __fadd_rn(xx, __shfl_xor(xx, off));
  float myv = 0.f;  // lane c keeps the cosine to class c
  for (int c = 0; c < C; ++c) {
    const float* pc = proto + (int64_t)c * D;
    float xy = 0.f, yy = 0.f;
    for (int e = lane; e < D; e += 64) {
      xy = fmaf(x[e], pc[e], xy);
      yy = fmaf(pc[e], pc[e], yy);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      xy = __fadd_rn(xy, __shfl_xor(xy, off));
      yy = __fadd_rn(yy, __shfl_xor(yy, off));
    }
    // torch.cosine_similarity(x, y, dim=0, eps=1e-8): x.y / (max(|x|, eps) * max(|y|, eps))
    const float cs = xy / (fmaxf(sqrtf(xx), 1e-8f) * fmaxf(sqrtf(yy), 1e-8f));
    if (lane == c) myv = cs;
  }
  if (mode != 0) {
    float v = (lane < C) ? myv : RG_NEG_INF;
    float m = v;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float ex = (lane < C) ? expf(v - m) : 0.f;
    float s = ex;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s = __fadd_rn(s, __shfl_xor(s, off));
    myv = (mode == 1) ? ex / s : (v - m) - logf(s);
  }
  if (lane < C) out[g * C + lane] = myv;
}

// Backward of proto_cosine with respect to the EMBEDDINGS (the prototypes are constants of a forward,
// RAGraph_node/downprompt.py:24,41-46): out = f(cos), f = identity / softmax / log_softmax;
//   gcos_c = go_c                          (mode 0)
//          = p_c (go_c - sum_j go_j p_j)   (mode 1, p = out)
//          = go_c - exp(out_c) sum_j go_j  (mode 2)
//   gemb[e] = sum_c gcos_c (p_c[e] / (|x| |p_c|) - cos_c x[e] / |x|^2).   One wave per embedding.
__global__ void __launch_bounds__(256) proto_cosine_grad_kernel(const float* __restrict__ emb, int64_t G, int D,
                                                                const float* __restrict__ proto, int C, int mode,
                                                                const float* __restrict__ out,
                                                                const float* __restrict__ gout,
                                                                float* __restrict__ gemb) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= G) return;
  const float* x = emb + g * D;
  float xx = 0.f;
  for (int e = lane; e < D; e += 64) xx = fmaf(x[e], x[e], xx);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) xx = __fadd_rn(xx, __shfl_xor(xx, off));
  const float nx = fmaxf(sqrtf(xx), 1e-8f);
  const float o = (lane < C) ? out[g * C + lane] : 0.f;
  const float go = (lane < C) ? gout[g * C + lane] : 0.f;
  float gc = go;
  if (mode == 1) {
    float dot = go * o;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off);
    gc = o * (go - dot);
  } else if (mode == 2) {
    float sum = go;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    gc = go - expf(o) * sum;
  }
  float sx = 0.f;  // sum_c gcos_c cos_c / |x|^2
  for (int e0 = 0; e0 < D; e0 += 64) {
    const int e = e0 + lane;
    if (e < D) gemb[g * D + e] = 0.f;
  }
  for (int c = 0; c < C; ++c) {
    const float* pc = proto + (int64_t)c * D;
    float xy = 0.f, yy = 0.f;
    for (int e = lane; e < D; e += 64) {
      xy = fmaf(x[e], pc[e], xy);
      yy = fmaf(pc[e], pc[e], yy);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      xy = __fadd_rn(xy, __shfl_xor(xy, off));
      yy = __fadd_rn(yy, __shfl_xor(yy, off));
    }
    const float np = fmaxf(sqrtf(yy), 1e-8f);
    const float cs = xy / (nx * np);
    const float gcc = __shfl(gc, c);
    sx += gcc * cs / (nx * nx);
    const float a = gcc / (nx * np);
    for (int e = lane; e < D; e += 64) gemb[g * D + e] += a * pc[e];
  }
  for (int e = lane; e < D; e += 64) gemb[g * D + e] -= sx * x[e];
}

// Backward of proto_cosine with respect to the PROTOTYPES (a training step of the node flavour rebuilds them from the very
// embeddings they are compared with and keeps them in the graph: RAGraph_node/downprompt.py:24-46):
//   gproto_c[e] = sum_g gcos_gc (x_g[e] / (|x_g| |p_c|) - cos_gc p_c[e] / |p_c|^2),  gcos as in proto_cosine_grad_kernel.
// Two launches, every sum in a fixed order: a workgroup takes 256 consecutive embeddings, its four waves 64 each, one after the
// other, into lane-private LDS accumulators [wave][C][D]; the waves' sums are added in wave order into partial[block][C][D],
// and the second launch adds the blocks in order.
__global__ void __launch_bounds__(256) proto_cosine_grad_proto_kernel(const float* __restrict__ emb, int64_t G, int D,
                                                                      const float* __restrict__ proto, int C, int mode,
                                                                      const float* __restrict__ out,
                                                                      const float* __restrict__ gout,
                                                                      float* __restrict__ partial) {
  extern __shared__ float pacc[];  // [4][C][D]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* acc = pacc + (size_t)w * C * D;
  for (int i = lane; i < C * D; i += 64) acc[i] = 0.f;
  float np_mine = 1.f;  // lane c: |p_c|
  for (int c = 0; c < C; ++c) {
    const float* pc = proto + (int64_t)c * D;
    float yy = 0.f;
    for (int e = lane; e < D; e += 64) yy = fmaf(pc[e], pc[e], yy);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) yy = __fadd_rn(yy, __shfl_xor(yy, off));
    if (lane == c) np_mine = fmaxf(sqrtf(yy), 1e-8f);
  }
  const int64_t g0 = (int64_t)blockIdx.x * 256 + w * 64;
  for (int64_t g = g0; g < g0 + 64 && g < G; ++g) {
    const float* x = emb + g * D;
    float xx = 0.f;
    for (int e = lane; e < D; e += 64) xx = fmaf(x[e], x[e], xx);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) xx = __fadd_rn(xx, __shfl_xor(xx, off));
    const float nx = fmaxf(sqrtf(xx), 1e-8f);
    const float o = (lane < C) ? out[g * C + lane] : 0.f;
    const float go = (lane < C) ? gout[g * C + lane] : 0.f;
    float gc = go;
    if (mode == 1) {
      float dot = go * o;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off);
      gc = o * (go - dot);
    } else if (mode == 2) {
      float sum = go;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
      gc = go - expf(o) * sum;
    }
    for (int c = 0; c < C; ++c) {
      const float* pc = proto + (int64_t)c * D;
      float xy = 0.f;
      for (int e = lane; e < D; e += 64) xy = fmaf(x[e], pc[e], xy);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) xy = __fadd_rn(xy, __shfl_xor(xy, off));
      const float np = __shfl(np_mine, c);
      const float cs = xy / (nx * np);
      const float gcc = __shfl(gc, c);
      const float a = gcc / (nx * np), b = gcc * cs / (np * np);
      for (int e = lane; e < D; e += 64) acc[c * D + e] += a * x[e] - b * pc[e];
    }
  }
  __syncthreads();
  float* dst = partial + (int64_t)blockIdx.x * C * D;
  for (int i = threadIdx.x; i < C * D; i += 256) {
    float v = pacc[i];
    for (int ww = 1; ww < 4; ++ww) v = __fadd_rn(v, pacc[(size_t)ww * C * D + i]);
    dst[i] = v;
  }
}

__global__ void __launch_bounds__(256) proto_cosine_grad_proto_sum_kernel(const float* __restrict__ partial, int64_t blocks,
                                                                          int CD, float* __restrict__ gproto) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= CD) return;
  float v = 0.f;
  for (int64_t b = 0; b < blocks; ++b) v = __fadd_rn(v, partial[b * CD + i]);
  gproto[i] = v;
}

}  // namespace ragraph

using namespace ragraph;

extern "C" int ragraph_abi_version(void) { return RAGRAPH_ABI_VERSION; }
extern "C" const char* ragraph_last_error(void) { return g_err; }

extern "C" int ragraph_device_check(void) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    set_error("no HIP device visible: %s", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return RAGRAPH_EDEVICE;
  }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    set_error("cannot query the current HIP device");
    return RAGRAPH_EDEVICE;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; libragraph_hip.so carries gfx950 (MI355X) code only", dev, prop.gcnArchName);
    return RAGRAPH_EDEVICE;
  }
  return RAGRAPH_OK;
}

extern "C" int ragraph_normalize_rows_f32(const float* X, int64_t n, int D, float* out, void* stream) {
  RG_REQUIRE(X && out, RAGRAPH_EINVAL, "normalize_rows: null pointer");
  RG_REQUIRE(n >= 0 && D >= 1, RAGRAPH_EINVAL, "normalize_rows: n=%lld D=%d", (long long)n, D);
  if (n == 0) return RAGRAPH_OK;
  const bool vec = (D % 4 == 0) && aligned16(X) && aligned16(out);
  dim3 grid((unsigned)cdiv(n, 4));
  if (vec)
    hipLaunchKernelGGL(normalize_rows_kernel<true>, grid, dim3(256), 0, as_stream(stream), X, n, D, out);
  else
    hipLaunchKernelGGL(normalize_rows_kernel<false>, grid, dim3(256), 0, as_stream(stream), X, n, D, out);
  RG_CHECK_LAUNCH("normalize_rows");
  return RAGRAPH_OK;
}

extern "C" int ragraph_gather_rows_f32(const float* V, int64_t N, int D, const int64_t* idx, int64_t M,
                                       int64_t idx_base, float* out, void* stream) {
  RG_REQUIRE(V && idx && out, RAGRAPH_EINVAL, "gather_rows: null pointer");
  RG_REQUIRE(N >= 0 && D >= 1 && M >= 0, RAGRAPH_EINVAL, "gather_rows: bad shape");
  if (M == 0) return RAGRAPH_OK;
  const bool vec = (D % 4 == 0) && aligned16(V) && aligned16(out);
  dim3 grid((unsigned)cdiv(M, 4));
  if (vec)
    hipLaunchKernelGGL(gather_rows_kernel<true>, grid, dim3(256), 0, as_stream(stream), V, N, D, idx, M, idx_base, out);
  else
    hipLaunchKernelGGL(gather_rows_kernel<false>, grid, dim3(256), 0, as_stream(stream), V, N, D, idx, M, idx_base, out);
  RG_CHECK_LAUNCH("gather_rows");
  return RAGRAPH_OK;
}

static int launch_gather_reduce(const float* V, int D, const float* L, int C, int64_t N, const int64_t* idx, int64_t B, int k,
                                int64_t idx_base, float v_scale, float* sum_V, float* mean_L, const float* A, float wa, float wb,
                                void* stream) {
  RG_REQUIRE(V && idx && sum_V, RAGRAPH_EINVAL, "gather_reduce: null pointer");
  RG_REQUIRE(D >= 1 && k >= 1 && B >= 0 && N >= 0, RAGRAPH_EINVAL, "gather_reduce: bad shape");
  RG_REQUIRE((L == nullptr) == (mean_L == nullptr), RAGRAPH_EINVAL, "gather_reduce: L and mean_L go together");
  RG_REQUIRE(!L || C >= 1, RAGRAPH_EINVAL, "gather_reduce: C=%d", C);
  if (B == 0) return RAGRAPH_OK;
  const bool vec = (D % 4 == 0) && aligned16(V) && aligned16(sum_V) && (!A || aligned16(A));
  if (vec)
    hipLaunchKernelGGL(gather_reduce_kernel<true>, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, as_stream(stream), V, D, L,
                       C, N, idx, B, k, idx_base, v_scale, sum_V, mean_L, A, wa, wb);
  else
    hipLaunchKernelGGL(gather_reduce_kernel<false>, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, as_stream(stream), V, D, L,
                       C, N, idx, B, k, idx_base, v_scale, sum_V, mean_L, A, wa, wb);
  RG_CHECK_LAUNCH("gather_reduce");
  return RAGRAPH_OK;
}

extern "C" int ragraph_gather_reduce_f32(const float* V, int D, const float* L, int C, int64_t N, const int64_t* idx,
                                         int64_t B, int k, int64_t idx_base, float v_scale, float* sum_V,
                                         float* mean_L, void* stream) {
  return launch_gather_reduce(V, D, L, C, N, idx, B, k, idx_base, v_scale, sum_V, mean_L, nullptr, 0.f, 0.f, stream);
}

// a8, the reduction and the prompt fusion in one launch  -- RAGraph.py:48-49 + :53: out = A * wa + (v_scale * sum_k V[idx]) * wb
extern "C" int ragraph_gather_reduce_mix_f32(const float* V, int D, const float* L, int C, int64_t N, const int64_t* idx,
                                             int64_t B, int k, int64_t idx_base, float v_scale, const float* A, float wa,
                                             float wb, float* out, float* mean_L, void* stream) {
  RG_REQUIRE(A, RAGRAPH_EINVAL, "gather_reduce_mix: null pointer");
  RG_REQUIRE(A != out, RAGRAPH_EINVAL, "gather_reduce_mix: out must not alias A");
  return launch_gather_reduce(V, D, L, C, N, idx, B, k, idx_base, v_scale, out, mean_L, A, wa, wb, stream);
}

extern "C" int ragraph_axpby_f32(const float* a, float wa, const float* b, float wb, int64_t n, float* out,
                                 void* stream) {
  RG_REQUIRE(a && b && out, RAGRAPH_EINVAL, "axpby: null pointer");
  if (n <= 0) return RAGRAPH_OK;
  int64_t blocks = cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a, wa, b, wb, n, out);
  RG_CHECK_LAUNCH("axpby");
  return RAGRAPH_OK;
}

extern "C" int ragraph_softmax_mix_f32(const float* logits, const float* rag_label, int64_t B, int C, float lambda,
                                       int log_mode, float* out, void* stream) {
  RG_REQUIRE(logits && out, RAGRAPH_EINVAL, "softmax_mix: null pointer");
  RG_REQUIRE(C >= 1 && C <= 1024, RAGRAPH_EUNSUPPORTED, "softmax_mix: C=%d not in [1,1024]", C);
  if (B <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(softmax_mix_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, as_stream(stream), logits,
                     rag_label, B, C, lambda, log_mode, out);
  RG_CHECK_LAUNCH("softmax_mix");
  return RAGRAPH_OK;
}

extern "C" int ragraph_proto_cosine_f32(const float* emb, int64_t G, int D, const float* proto, int C, int mode,
                                        float* out, void* stream) {
  RG_REQUIRE(emb && proto && out, RAGRAPH_EINVAL, "proto_cosine: null pointer");
  RG_REQUIRE(C >= 1 && C <= 64, RAGRAPH_EUNSUPPORTED, "proto_cosine: C=%d not in [1,64]", C);
  RG_REQUIRE(D >= 1 && mode >= 0 && mode <= 2, RAGRAPH_EINVAL, "proto_cosine: bad D/mode");
  if (G <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(proto_cosine_kernel, dim3((unsigned)cdiv(G, 4)), dim3(256), 0, as_stream(stream), emb, G, D, proto,
                     C, mode, out);
  RG_CHECK_LAUNCH("proto_cosine");
  return RAGRAPH_OK;
}

extern "C" int ragraph_proto_cosine_grad_f32(const float* emb, int64_t G, int D, const float* proto, int C, int mode,
                                             const float* out, const float* gout, float* gemb, void* stream) {
  RG_REQUIRE(emb && proto && out && gout && gemb, RAGRAPH_EINVAL, "proto_cosine_grad: null pointer");
  RG_REQUIRE(C >= 1 && C <= 64, RAGRAPH_EUNSUPPORTED, "proto_cosine_grad: C=%d not in [1,64]", C);
  RG_REQUIRE(D >= 1 && mode >= 0 && mode <= 2, RAGRAPH_EINVAL, "proto_cosine_grad: bad D/mode");
  if (G <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(proto_cosine_grad_kernel, dim3((unsigned)cdiv(G, 4)), dim3(256), 0, as_stream(stream), emb, G, D,
                     proto, C, mode, out, gout, gemb);
  RG_CHECK_LAUNCH("proto_cosine_grad");
  return RAGRAPH_OK;
}

extern "C" size_t ragraph_proto_cosine_grad_proto_workspace_bytes(int64_t G, int C, int D) {
  return (size_t)(G > 0 ? cdiv(G, 256) : 1) * (size_t)C * (size_t)D * sizeof(float);
}

extern "C" int ragraph_proto_cosine_grad_proto_f32(const float* emb, int64_t G, int D, const float* proto, int C, int mode,
                                                   const float* out, const float* gout, float* gproto, void* workspace,
                                                   size_t workspace_bytes, void* stream) {
  RG_REQUIRE(emb && proto && out && gout && gproto && workspace, RAGRAPH_EINVAL, "proto_cosine_grad_proto: null pointer");
  RG_REQUIRE(C >= 1 && C <= 64, RAGRAPH_EUNSUPPORTED, "proto_cosine_grad_proto: C=%d not in [1,64]", C);
  RG_REQUIRE(D >= 1 && mode >= 0 && mode <= 2, RAGRAPH_EINVAL, "proto_cosine_grad_proto: bad D/mode");
  RG_REQUIRE((int64_t)C * D <= 8192, RAGRAPH_EUNSUPPORTED, "proto_cosine_grad_proto: C * D = %lld > 8192 (LDS accumulators)",
             (long long)C * D);
  RG_REQUIRE(workspace_bytes >= ragraph_proto_cosine_grad_proto_workspace_bytes(G, C, D), RAGRAPH_EINVAL,
             "proto_cosine_grad_proto: workspace of %zu bytes is too small", workspace_bytes);
  hipStream_t st = as_stream(stream);
  const int64_t blocks = G > 0 ? cdiv(G, 256) : 0;
  float* partial = reinterpret_cast<float*>(workspace);
  const size_t lds = (size_t)4 * C * D * sizeof(float);
  if (blocks > 0) {
    static DeviceOnce lds_once;
    if (hipError_t e = raise_dynamic_lds(lds_once, &proto_cosine_grad_proto_kernel, 160 * 1024); e != hipSuccess) {
      set_error("proto_cosine_grad_proto: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
      return RAGRAPH_EDEVICE;
    }
    hipLaunchKernelGGL(proto_cosine_grad_proto_kernel, dim3((unsigned)blocks), dim3(256), lds, st, emb, G, D, proto, C, mode,
                       out, gout, partial);
    RG_CHECK_LAUNCH("proto_cosine_grad_proto");
  }
  hipLaunchKernelGGL(proto_cosine_grad_proto_sum_kernel, dim3((unsigned)cdiv((int64_t)C * D, 256)), dim3(256), 0, st, partial,
                     blocks, C * D, gproto);
  RG_CHECK_LAUNCH("proto_cosine_grad_proto(sum)");
  return RAGRAPH_OK;
}

extern "C" int ragraph_axpby_dev_f32(const float* a, const float* b, const float* w, int ia, int ib, int64_t n, float* out,
                                     void* stream) {
  RG_REQUIRE(a && b && w && out, RAGRAPH_EINVAL, "axpby_dev: null pointer");
  RG_REQUIRE(ia < 64 && ib < 64, RAGRAPH_EINVAL, "axpby_dev: weight index out of range");
  if (n <= 0) return RAGRAPH_OK;
  int64_t blocks = cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(axpby_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a, b, w, ia, ib, n, out);
  RG_CHECK_LAUNCH("axpby_dev");
  return RAGRAPH_OK;
}

extern "C" int ragraph_sigmoid_gate_f32(const float* x, const float* z, int64_t n, float* out, void* stream) {
  RG_REQUIRE(x && z && out, RAGRAPH_EINVAL, "sigmoid_gate: null pointer");
  if (n <= 0) return RAGRAPH_OK;
  int64_t blocks = cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sigmoid_gate_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, z, n, out);
  RG_CHECK_LAUNCH("sigmoid_gate");
  return RAGRAPH_OK;
}

extern "C" int ragraph_time_rescale_f32(const int64_t* t, int64_t n, float t_min, float t_max, float* out,
                                        void* stream) {
  RG_REQUIRE(t && out, RAGRAPH_EINVAL, "time_rescale: null pointer");
  if (n <= 0) return RAGRAPH_OK;
  int64_t blocks = cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(time_rescale_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), t, n, t_min, t_max, out);
  RG_CHECK_LAUNCH("time_rescale");
  return RAGRAPH_OK;
}

static unsigned ew_blocks(int64_t n) {
  int64_t b = cdiv(n, 256);
  return (unsigned)(b > 2048 ? 2048 : b);
}

extern "C" int ragraph_act_grad_f32(const float* y, const float* gy, int64_t n, int act, float alpha, float* gz,
                                    float* alpha_terms, void* stream) {
  RG_REQUIRE(y && gy && gz, RAGRAPH_EINVAL, "act_grad: null pointer");
  if (n <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(act_grad_kernel, dim3(ew_blocks(n)), dim3(256), 0, as_stream(stream), y, gy, n, act, alpha, gz,
                     alpha_terms);
  RG_CHECK_LAUNCH("act_grad");
  return RAGRAPH_OK;
}

extern "C" int ragraph_sigmoid_gate_grad_f32(const float* x, const float* z, const float* g, int64_t n, float* gx, float* gz,
                                             void* stream) {
  RG_REQUIRE(x && z && g && gx && gz, RAGRAPH_EINVAL, "sigmoid_gate_grad: null pointer");
  if (n <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(sigmoid_gate_grad_kernel, dim3(ew_blocks(n)), dim3(256), 0, as_stream(stream), x, z, g, n, gx, gz);
  RG_CHECK_LAUNCH("sigmoid_gate_grad");
  return RAGRAPH_OK;
}

extern "C" int ragraph_softmax_grad_f32(const float* p, const float* go, int64_t B, int C, float scale, float* out,
                                        void* stream) {
  RG_REQUIRE(p && go && out, RAGRAPH_EINVAL, "softmax_grad: null pointer");
  if (B <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(softmax_grad_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, as_stream(stream), p, go, B, C, scale, out);
  RG_CHECK_LAUNCH("softmax_grad");
  return RAGRAPH_OK;
}

extern "C" int ragraph_mul_cols_act_f32(const float* x, const float* w, int64_t n, int D, int act, float alpha, float* out,
                                        void* stream) {
  RG_REQUIRE(x && w && out && D >= 1, RAGRAPH_EINVAL, "mul_cols: bad argument");
  RG_REQUIRE(act >= RAGRAPH_ACT_NONE && act <= RAGRAPH_ACT_ELU, RAGRAPH_EINVAL, "mul_cols: unknown activation %d", act);
  if (n <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(mul_cols_kernel, dim3(ew_blocks(n * D)), dim3(256), 0, as_stream(stream), x, w, n, D, act, alpha, out);
  RG_CHECK_LAUNCH("mul_cols");
  return RAGRAPH_OK;
}

extern "C" int ragraph_mul_cols_f32(const float* x, const float* w, int64_t n, int D, float* out, void* stream) {
  return ragraph_mul_cols_act_f32(x, w, n, D, RAGRAPH_ACT_NONE, 0.f, out, stream);
}

extern "C" int ragraph_mul_f32(const float* a, const float* b, int64_t n, float* out, void* stream) {
  RG_REQUIRE(a && b && out, RAGRAPH_EINVAL, "mul: null pointer");
  if (n <= 0) return RAGRAPH_OK;
  hipLaunchKernelGGL(mul_kernel, dim3(ew_blocks(n)), dim3(256), 0, as_stream(stream), a, b, n, out);
  RG_CHECK_LAUNCH("mul");
  return RAGRAPH_OK;
}
