// Column-sliced SpMM hop: does making an XCD's share of X fit (or nearly fit) its 4-MiB L2 beat one wave per 1-KiB row?
//
// The product kernel (csrc/sparse.hip, spmm_csr_kernel<64>) gathers whole 1-KiB rows of X [n, 256]: on c2's random graph
// (n = 100 000, degree ~11) every XCD touches all 102 MB of X, its L2 holds 4 % of it, and the hop moves ~0.98 GB over
// the fabric for 0.214 GB of algorithmic bytes (profiles/r2_spmm_locality.txt, FETCH_SIZE doubled as the guide says).
// Here X is cut into P = D / W column panels of W floats; workgroup b runs on XCD b % 8, and the workgroups of XCD x
// take panels x, x + 8, ... one after the other (P < 8: 8 / P XCDs share a panel, each a contiguous part of the rows),
// so the rows an XCD gathers from at any time are n * W * 4 bytes: 12.8 MB at W = 32, 6.4 MB at W = 16, 3.2 MB at W = 8.
// A row's fmaf chain is unchanged (same edges, same order, per element), so the result has the bits of the row kernel.
// Layouts: "panel" = [P][n][W] (a panel is contiguous: what a fused multi-hop propagate would keep between hops),
// "rowmajor" = the natural [n][D] with the panel as a column offset (lines of one panel lie 1 KiB apart).
//
//   hipcc --offload-arch=gfx950 -O3 -I include -o /tmp/spmm_panel tools/microbench/spmm_panel_bench.hip \
//         -L ragraph_amd/csrc -lragraph_hip -Wl,-rpath,$PWD/ragraph_amd/csrc
//   /tmp/spmm_panel [variant] [reps]     variant: all | base | p8 | p16 | p32 | p64 | r16 | r32 | r64
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "ragraph_hip.h"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

// W floats per panel row; LP = W / 4 lanes (float4 each) own one output row of the panel; a wave holds 64 / LP rows.
// Edges in chunks of CH = 8: lane lr of a row's group loads edges lr, lr + LP, ... of the chunk, the group broadcasts
// them, all gathers of the chunk are issued before the first fmaf.
template <int W>
__global__ void __launch_bounds__(256) spmm_panel_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const float* __restrict__ val, int64_t n, const float* __restrict__ X,
                                                         int64_t x_panel_stride, int64_t x_row_stride, float* __restrict__ Y,
                                                         int64_t y_panel_stride, int64_t y_row_stride, int P, int relu) {
  constexpr int LP = W / 4;
  constexpr int ROWS_W = 64 / LP;      // rows per wave
  constexpr int ROWS_B = 4 * ROWS_W;   // rows per workgroup
  constexpr int CH = 8;
  constexpr int PER = CH / LP > 0 ? CH / LP : 1;   // chunk edges a lane loads
  const int lane = threadIdx.x & 63;
  const int lr = lane % LP;
  const int gbase = lane - lr;
  const unsigned xcd = blockIdx.x % 8, i = blockIdx.x / 8;   // i-th workgroup of this XCD
  const int64_t row_blocks = (n + ROWS_B - 1) / ROWS_B;
  int panel;
  int64_t rb;
  if (P >= 8) {  // XCD x: panels x, x + 8, ... in turn, all row blocks of one before the next
    panel = (int)xcd + 8 * (int)(i / row_blocks);
    rb = i % row_blocks;
  } else {       // 8 / P XCDs share a panel, each a contiguous part of the row blocks
    const int share = 8 / P;
    panel = (int)(xcd % P);
    const int64_t per = (row_blocks + share - 1) / share;
    rb = (int64_t)(xcd / P) * per + i;
    if (i >= per) return;
  }
  if (panel >= P || rb >= row_blocks) return;
  int64_t row = rb * ROWS_B + (threadIdx.x >> 6) * ROWS_W + lane / LP;
  const bool live = row < n;
  if (!live) row = n - 1;
  const int64_t e0 = rowptr[row];
  const int deg = live ? (int)(rowptr[row + 1] - e0) : 0;
  // the longest row of the wave sets the trip count (shuffles need every lane)
  int maxdeg = deg;
#pragma unroll
  for (int off = 32; off >= LP; off >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, off));
  const float4* Xp = reinterpret_cast<const float4*>(X + (int64_t)panel * x_panel_stride) + lr;
  const int64_t xrs4 = x_row_stride / 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int base = 0; base < maxdeg; base += CH) {
    int my_c[PER];
    float my_v[PER];
#pragma unroll
    for (int s = 0; s < PER; ++s) {
      const int e = base + lr + s * LP;
      const bool ok = (LP <= CH || lr < CH) && e < deg;
      my_c[s] = ok ? col[e0 + e] : 0;
      my_v[s] = ok ? val[e0 + e] : 0.f;
    }
    float4 x[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = __shfl(my_c[k / LP < PER ? k / LP : 0], gbase + (k % LP));
      x[k] = (base + k < deg) ? Xp[(int64_t)c * xrs4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const float v = __shfl(my_v[k / LP < PER ? k / LP : 0], gbase + (k % LP));
      if (base + k < deg) {
        acc.x = fmaf(v, x[k].x, acc.x); acc.y = fmaf(v, x[k].y, acc.y);
        acc.z = fmaf(v, x[k].z, acc.z); acc.w = fmaf(v, x[k].w, acc.w);
      }
    }
  }
  if (!live) return;
  if (relu) {
    acc.x = acc.x > 0.f ? acc.x : 0.f; acc.y = acc.y > 0.f ? acc.y : 0.f;
    acc.z = acc.z > 0.f ? acc.z : 0.f; acc.w = acc.w > 0.f ? acc.w : 0.f;
  }
  reinterpret_cast<float4*>(Y + (int64_t)panel * y_panel_stride + row * y_row_stride)[lr] = acc;
}

// [n][D] <-> [P][n][W]
__global__ void to_panel(const float* __restrict__ X, int64_t n, int D, int W, float* __restrict__ Xp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * D) return;
  const int64_t r = i / D;
  const int c = (int)(i % D);
  Xp[(int64_t)(c / W) * n * W + r * W + c % W] = X[i];
}
__global__ void from_panel(const float* __restrict__ Xp, int64_t n, int D, int W, float* __restrict__ X) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * D) return;
  const int64_t r = i / D;
  const int c = (int)(i % D);
  X[i] = Xp[(int64_t)(c / W) * n * W + r * W + c % W];
}

template <int W>
static void launch_panel(const int64_t* rp, const int32_t* col, const float* val, int64_t n, int D, const float* X, float* Y,
                         bool panel_layout) {
  const int P = D / W;
  constexpr int ROWS_B = 4 * (64 / (W / 4));
  const int64_t row_blocks = (n + ROWS_B - 1) / ROWS_B;
  int64_t per_xcd = P >= 8 ? row_blocks * (P / 8) : (row_blocks + 8 / P - 1) / (8 / P);
  const int64_t xps = panel_layout ? n * W : W, xrs = panel_layout ? W : D;
  hipLaunchKernelGGL(spmm_panel_kernel<W>, dim3((unsigned)(per_xcd * 8)), dim3(256), 0, 0, rp, col, val, n, X, xps, xrs, Y, xps,
                     xrs, P, 1);
}

int main(int argc, char** argv) {
  const std::string which = argc > 1 ? argv[1] : "all";
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int64_t n = argc > 3 ? atoll(argv[3]) : 100000;
  const int D = 256, deg = 10;
  // c2's graph: Erdos-Renyi (mean degree 10, symmetric) + ring + self loops, ascending columns per row
  std::mt19937_64 rng(8);
  std::vector<std::vector<int32_t>> adj(n);
  const int64_t m = n * deg / 2;
  for (int64_t e = 0; e < m; ++e) {
    const int64_t a = rng() % n, b = rng() % n;
    if (a == b) continue;
    adj[a].push_back((int32_t)b);
    adj[b].push_back((int32_t)a);
  }
  for (int64_t r = 0; r < n; ++r) {
    adj[r].push_back((int32_t)r);
    adj[r].push_back((int32_t)((r + 1) % n));
    adj[r].push_back((int32_t)((r + n - 1) % n));
    std::sort(adj[r].begin(), adj[r].end());
    adj[r].erase(std::unique(adj[r].begin(), adj[r].end()), adj[r].end());
  }
  std::vector<int64_t> rowptr(n + 1, 0);
  for (int64_t r = 0; r < n; ++r) rowptr[r + 1] = rowptr[r] + (int64_t)adj[r].size();
  const int64_t nnz = rowptr[n];
  std::vector<int32_t> col(nnz);
  std::vector<float> val(nnz), X((size_t)n * D);
  std::uniform_real_distribution<float> uf(-1.f, 1.f);
  std::mt19937 r32(9);
  for (int64_t r = 0; r < n; ++r)
    for (size_t j = 0; j < adj[r].size(); ++j) {
      col[rowptr[r] + j] = adj[r][j];
      val[rowptr[r] + j] = 1.f / (float)adj[r].size();
    }
  for (auto& x : X) x = uf(r32);
  printf("n=%lld nnz=%lld D=%d: algorithmic %.1f MB per hop (X once + Y once + CSR)\n", (long long)n, (long long)nnz, D,
         (nnz * 8.0 + 8.0 * (n + 1) + 8.0 * n * D) / 1e6);

  int64_t* d_rp;
  int32_t* d_col;
  float *d_val, *d_X, *d_Xp, *d_Y, *d_Yp, *d_Yr;
  CK(hipMalloc(&d_rp, (n + 1) * 8));
  CK(hipMalloc(&d_col, nnz * 4));
  CK(hipMalloc(&d_val, nnz * 4));
  const size_t xb = (size_t)n * D * 4;
  CK(hipMalloc(&d_X, xb)); CK(hipMalloc(&d_Xp, xb)); CK(hipMalloc(&d_Y, xb)); CK(hipMalloc(&d_Yp, xb)); CK(hipMalloc(&d_Yr, xb));
  CK(hipMemcpy(d_rp, rowptr.data(), (n + 1) * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val, val.data(), nnz * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_X, X.data(), xb, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-10s %8.1f us per hop\n", name, ms * 1000.f / reps);
    return ms * 1000.f / reps;
  };
  // baseline: the product kernel through the C ABI
  std::vector<float> ref((size_t)n * D), got((size_t)n * D);
  auto base = [&]() {
    if (ragraph_spmm_csr_f32(d_rp, d_col, d_val, n, d_X, D, nullptr, RAGRAPH_ACT_RELU, 0.f, 0.f, nullptr, d_Y, nullptr) != 0) {
      fprintf(stderr, "spmm_csr failed: %s\n", ragraph_last_error());
      exit(1);
    }
  };
  base();
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ref.data(), d_Y, xb, hipMemcpyDeviceToHost));
  if (which == "all" || which == "base") time_it("base", base);
  auto check = [&](const char* name, const float* dY) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), dY, xb, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += memcmp(&got[i], &ref[i], 4) != 0;
    printf("%-10s %s (%zu of %zu values differ from the row kernel)\n", name, bad ? "MISMATCH" : "bit-exact", bad, got.size());
  };
  const unsigned eb = (unsigned)((n * D + 255) / 256);
#define PANEL(Wv, tag)                                                                                         \
  if (which == "all" || which == tag) {                                                                        \
    hipLaunchKernelGGL(to_panel, dim3(eb), dim3(256), 0, 0, d_X, n, D, Wv, d_Xp);                              \
    time_it(tag, [&]() { launch_panel<Wv>(d_rp, d_col, d_val, n, D, d_Xp, d_Yp, true); });                     \
    hipLaunchKernelGGL(from_panel, dim3(eb), dim3(256), 0, 0, d_Yp, n, D, Wv, d_Yr);                           \
    check(tag, d_Yr);                                                                                          \
  }
#define ROWM(Wv, tag)                                                                                          \
  if (which == "all" || which == tag) {                                                                        \
    CK(hipMemset(d_Yr, 0, xb));                                                                                \
    time_it(tag, [&]() { launch_panel<Wv>(d_rp, d_col, d_val, n, D, d_X, d_Yr, false); });                     \
    check(tag, d_Yr);                                                                                          \
  }
  PANEL(8, "p8") PANEL(16, "p16") PANEL(32, "p32") PANEL(64, "p64")
  ROWM(16, "r16") ROWM(32, "r32") ROWM(64, "r64")
  return 0;
}
