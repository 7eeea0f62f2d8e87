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

// Rows longer than ROW_BLOCK edges (the hubs of a power-law graph) are summed in blocks: each block of ROW_BLOCK
// consecutive edges is its own sequential chain from +0 and the block sums are added in block order -- the order
// oracle/ragraph_oracle.c uses (ORACLE_ROW_BLOCK).  One chain over the 3 M edges of c5's most popular item kept a
// single lane group busy for 31 ms per layer while the chip idled; with a workspace the blocks of long rows are
// separate tasks for the whole chip (long_rows_kernel lists them, spmm_long_blocks_kernel sums them,
// spmm_long_finish_kernel adds a row's block sums in order and applies the epilogue).  Without a workspace the row's
// lane group walks its blocks itself: same bits, no parallelism.
constexpr int ROW_BLOCK = 4096;

// A row's edges are consumed in chunks of CH = 16: the first lanes of the row's lane group load the chunk's (col, val)
// pairs with ONE coalesced load each, the group broadcasts them, and all gathers of the chunk are issued back to back
// before the first fmaf -- three dependent memory latencies per row (rowptr, edge list, X rows) instead of one pair per
// four edges.  The sum is the sequential fmaf chain over edges [e0, e0 + cnt_edges) in CSR order, from +0.
template <int LPR>
__device__ __forceinline__ float4 row_chain(const int32_t* __restrict__ col, const float* __restrict__ val, int64_t e0,
                                            int cnt_edges, const float4* __restrict__ X4, int D4, int c4, bool colok,
                                            int lr, int gbase) {
  constexpr int CH = 16;  // edges per chunk (<= LPR)
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int base = 0; base < cnt_edges; base += CH) {
    int my_c = 0;
    float my_v = 0.f;
    if (lr < CH && base + lr < cnt_edges) {
      my_c = col[e0 + base + lr];
      my_v = val[e0 + base + lr];
    }
    const int cnt = cnt_edges - base < CH ? cnt_edges - base : CH;
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
  return acc;
}

__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(__fadd_rn(a.x, b.x), __fadd_rn(a.y, b.y), __fadd_rn(a.z, b.z), __fadd_rn(a.w, b.w));
}

// panel_n > 0: Y is PANEL-major -- [D / 32][panel_n][32] floats (a row's eight-float4 pieces of one 128-byte line go to the
// panel of their column block): the layout the column-panel hops below read.
__device__ __forceinline__ void spmm_epilogue_store(float4 acc, const float* __restrict__ bias, int act, float alpha,
                                                    float beta, const float* __restrict__ Yin, float* __restrict__ Y,
                                                    int64_t row, int D4, int c4, int64_t panel_n = 0) {
  if (bias) acc = add4(acc, reinterpret_cast<const float4*>(bias)[c4]);
  acc.x = apply_act(acc.x, act, alpha); acc.y = apply_act(acc.y, act, alpha);
  acc.z = apply_act(acc.z, act, alpha); acc.w = apply_act(acc.w, act, alpha);
  if (Yin) {
    const float4 y = reinterpret_cast<const float4*>(Yin)[row * D4 + c4];
    acc.x = fmaf(beta, y.x, acc.x); acc.y = fmaf(beta, y.y, acc.y); acc.z = fmaf(beta, y.z, acc.z); acc.w = fmaf(beta, y.w, acc.w);
  }
  if (panel_n > 0) reinterpret_cast<float4*>(Y)[((int64_t)(c4 >> 3) * panel_n + row) * 8 + (c4 & 7)] = acc;
  else reinterpret_cast<float4*>(Y)[row * D4 + c4] = acc;
}

// skip_long != 0: rows longer than ROW_BLOCK are left to the long-row kernels.
template <int LPR>
__global__ void __launch_bounds__(256) spmm_csr_kernel(const int64_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col,
                                                       const float* __restrict__ val, int64_t n,
                                                       const float* __restrict__ X, int D,
                                                       const float* __restrict__ bias, int act, float alpha, float beta,
                                                       const float* __restrict__ Yin, float* __restrict__ Y,
                                                       int skip_long, unsigned RUN, int64_t panel_n = 0) {
  constexpr int RPB = 256 / LPR;  // rows per block
  const int lr = threadIdx.x % LPR;
  const int gbase = (threadIdx.x & 63) - lr;  // first lane of this row's group inside the wave
  // Workgroup b runs on XCD b % 8, each with its own L2.  Inside every group of 8 x 32 consecutive workgroups an XCD gets
  // a CONTIGUOUS run of 32 (128 rows at D = 256) instead of every eighth, so that rows which share neighbours (self
  // loops, ring / community structure, any locality in the node numbering) share an L2; the groups keep the XCDs'
  // loads interleaved (one contiguous eighth per XCD was 60 % slower on c5's bipartite graph: users and items differ).
  // RUN = 32 by default (RAGRAPH_SPMM_XCD_RUN).  Measured on c2's shape (tools/spmm_locality.py, profiles/
  // r2_spmm_locality.txt, restated in DESIGN.md section 4.3 with FETCH_SIZE doubled as gfx950 requires): on c2's
  // Erdos-Renyi graph an XCD's L2 hits 15 - 20 % of the row gathers and the hop moves 0.98 GB over the fabric in 133 us --
  // 7.3 TB/s, the chip's ceiling for random row gathers from a 100 MB table; a community-ordered graph with RUN = 256
  // misses 0.35 GB and takes 100 us.  Column panels of one 128-byte line per row and XCD (profiles/r3_spmm_panel.txt)
  // raise the hit rate to 34 % and take 12 % off; narrower panels move more requests than they save.  Two restructurings
  // that attack the per-row dependent chain instead (rows of a run as one edge stream; row pointers and edge pairs
  // fetched one and two tiles ahead) were 10 - 50 % slower: latency is not what bounds it.
  const unsigned grp = blockIdx.x / (8 * RUN), in = blockIdx.x % (8 * RUN);
  const unsigned blk = (grp + 1) * (8 * RUN) <= gridDim.x ? grp * (8 * RUN) + (in % 8) * RUN + in / 8 : blockIdx.x;
  int64_t row = (int64_t)blk * RPB + threadIdx.x / LPR;
  bool live = row < n;  // dead groups run along with zero edges: the shuffles need every lane
  if (!live) row = n - 1;
  const int64_t e0 = rowptr[row];
  int64_t deg = live ? rowptr[row + 1] - e0 : 0;
  if (skip_long && deg > ROW_BLOCK) {
    live = false;
    deg = 0;
  }
  const int D4 = D >> 2;
  const float4* X4 = reinterpret_cast<const float4*>(X);

  for (int c4 = lr; c4 < ((D4 + LPR - 1) / LPR) * LPR; c4 += LPR) {  // uniform trip count: shuffles inside
    const bool colok = c4 < D4;
    float4 acc;
    if (deg <= ROW_BLOCK) {
      acc = row_chain<LPR>(col, val, e0, (int)deg, X4, D4, c4, colok, lr, gbase);
    } else {  // no workspace: this group walks the row's blocks itself
      acc = row_chain<LPR>(col, val, e0, ROW_BLOCK, X4, D4, c4, colok, lr, gbase);
      for (int64_t b0 = ROW_BLOCK; b0 < deg; b0 += ROW_BLOCK) {
        const int cnt = deg - b0 < ROW_BLOCK ? (int)(deg - b0) : ROW_BLOCK;
        acc = add4(acc, row_chain<LPR>(col, val, e0 + b0, cnt, X4, D4, c4, colok, lr, gbase));
      }
    }
    if (!live || !colok) continue;
    spmm_epilogue_store(acc, bias, act, alpha, beta, Yin, Y, row, D4, c4, panel_n);
  }
}

// ---- column-panel hops (round 4; measured as a microbenchmark in round 3: profiles/r3_spmm_panel.txt) -----------------
// On a graph without locality every XCD gathers from all of X (102 MB at c2) and its 4-MiB L2 hits 15 % of the row
// gathers.  Between the hops of a k-hop propagation the features can stay PANEL-major, [D / 32][n][32]: a panel row is one
// 128-byte line, workgroup b runs on XCD b % 8 and the workgroups of XCD x work through panels x, x + 8, ... (D = 256:
// exactly one panel per XCD), so the rows an XCD gathers from are n x 128 bytes (12.8 MB at c2: a third of them fit its L2;
// hit rate 34 %, traffic -11 %, time -12 %).  LP = 8 lanes hold one output row of the panel as float4, a wave 8 rows; edges
// in chunks of 8.  A row's chain is the row kernel's -- same edges, same order, per element, blocks of ROW_BLOCK edges each
// from +0 and added in order -- so every layout gives the same bits.
__global__ void __launch_bounds__(256) spmm_panel_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const float* __restrict__ val, int64_t n, const float* __restrict__ Xp,
                                                         int64_t x_panel_stride, int x_row4, float* __restrict__ Y,
                                                         int64_t y_panel_stride, int64_t y_row_stride, int P, int act,
                                                         float alpha) {
  constexpr int LP = 8, ROWS_W = 8, ROWS_B = 32, CH = 8;
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
  const int64_t deg = live ? rowptr[row + 1] - e0 : 0;
  int64_t maxdeg = deg;  // the longest row of the wave sets the trip counts (shuffles need every lane)
#pragma unroll
  for (int off = 32; off >= LP; off >>= 1) {
    const int64_t o = ((int64_t)__shfl_xor((int)(maxdeg >> 32), off) << 32) | (unsigned)__shfl_xor((int)maxdeg, off);
    maxdeg = o > maxdeg ? o : maxdeg;
  }
  // X panel-major: panel stride = 32 x (rows of the table), a row 8 float4; X row-major: the XCD's 128-byte slice of every
  // 4 D-byte row -- panel stride 32 floats, a row D / 4 float4 (the same lines per XCD either way)
  const float4* X4 = reinterpret_cast<const float4*>(Xp + (int64_t)panel * x_panel_stride) + lr;
  float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t b0 = 0; b0 < maxdeg || b0 == 0; b0 += ROW_BLOCK) {
    const int cnt_w = maxdeg - b0 < ROW_BLOCK ? (int)(maxdeg - b0) : ROW_BLOCK;          // wave-uniform
    const int cnt = deg - b0 < ROW_BLOCK ? (int)(deg - b0 > 0 ? deg - b0 : 0) : ROW_BLOCK;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < cnt_w; base += CH) {
      const int e = base + lr;
      const bool ok = e < cnt;
      const int my_c = ok ? col[e0 + b0 + e] : 0;
      const float my_v = ok ? val[e0 + b0 + e] : 0.f;
      float4 x[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int c = __shfl(my_c, gbase + k);
        x[k] = (base + k < cnt) ? X4[(int64_t)c * x_row4] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const float v = __shfl(my_v, gbase + k);
        if (base + k < cnt) {
          acc.x = fmaf(v, x[k].x, acc.x); acc.y = fmaf(v, x[k].y, acc.y);
          acc.z = fmaf(v, x[k].z, acc.z); acc.w = fmaf(v, x[k].w, acc.w);
        }
      }
    }
    if (b0 == 0) total = acc;
    else if (b0 < deg) total = add4(total, acc);
  }
  if (!live) return;
  total.x = apply_act(total.x, act, alpha); total.y = apply_act(total.y, act, alpha);
  total.z = apply_act(total.z, act, alpha); total.w = apply_act(total.w, act, alpha);
  reinterpret_cast<float4*>(Y + (int64_t)panel * y_panel_stride + row * y_row_stride)[lr] = total;
}

// ---- graph-tiled hops (round 6) ------------------------------------------------------------------------------------------
// On a graph without locality an XCD gathers from a 12.8 MB panel (c2) through a 4 MiB L2: two thirds of its row gathers miss
// (profiles/r5_gnn_plain_pmc_traffic.txt).  Here the GRAPH is tiled (ragraph_amd/graph.py: CSRGraph.tile_plan, once per
// graph): the source rows are cut into S blocks of a few MB of 128-byte panel lines, the destination rows into chunks of
// RC = 128 RG rows whose 32-column accumulators live in ONE workgroup's LDS (RG x 16 KiB), and the edges are stored once more
// in (chunk, 8-lane group, source block, row, column) order.  A workgroup (one per CU; 32 per XCD) takes a chunk's edges in
// that order: the groups of an XCD walk their runs at the same pace, so at any moment they gather from about the same source
// block and a line fetched for one edge is hit in L2 by the others that point at it.  A row's chain is unchanged: its edges
// are consumed in ascending column order (the CSR order of a graph with sorted columns, which the plan requires), block
// after block, the running sum parked in LDS between the row's visits -- the same fmaf sequence from +0, the same bits as
// spmm_csr_kernel / spmm_panel_kernel.  (Rows longer than ROW_BLOCK edges and graphs with unsorted columns have no plan.)
// Eight lanes own RG consecutive rows of the chunk (float4 each: one 128-byte panel line per row); no barrier anywhere -- a
// group only ever touches its own rows' sums.  The eight groups of a wave read their batch's edge words from ONE place --
// col3 / val3 / row3 [wave-batch][64]: lane 8 g + i holds edge i of group g's batch -- three coalesced loads per 64 edges (a
// group's run on its own lines costs the L1 twenty-four line requests per batch, streamed from HBM: 177 us instead of 125 in
// tools/microbench/spmm_tiled_bench.hip); a group whose run is shorter than its wave's longest is padded with NULL edges
// (row 0xFFFF: gathered from line 0, never accumulated).  What this kernel costs and why it is NOT the default: DESIGN.md 4.3.
struct TilePlan {
  const int* wp;                 // [C * 16 + 1] first wave-batch of (chunk, wave)
  const int* col3;               // [(wave-batches + 1) * 64] source row of every slot
  const float* val3;
  const unsigned short* row3;    // destination row inside the chunk, 0xFFFF = no edge
  int RG, C;
};

__global__ void __launch_bounds__(1024) spmm_tiled_kernel(TilePlan t, int64_t n, const float* __restrict__ Xp,
                                                          int64_t x_panel_stride, int x_row4, float* __restrict__ Y,
                                                          int64_t y_panel_stride, int64_t y_row_stride, int P, int act,
                                                          float alpha) {
  extern __shared__ float4 tile_acc[];   // [RC][8]
  constexpr int NG = 128;
  const int lr = threadIdx.x & 7, gi = threadIdx.x >> 3, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned xcd = blockIdx.x % 8, w = blockIdx.x / 8, wpx = gridDim.x / 8;
  const int share = P >= 8 ? 1 : 8 / P;             // XCDs that share a panel: each takes every share-th chunk
  const int sub = P >= 8 ? 0 : (int)(xcd / P);
  const int RC = t.RG * NG;
  float4* mine = tile_acc + (gi * t.RG) * 8 + lr;   // my 16 bytes of my group's first row
#define RG_SWZ(v_, k_) __builtin_amdgcn_ds_swizzle((v_), ((k_) << 5) | 0x18)   /* lane k of every group of 8 */
  for (int panel = P >= 8 ? (int)xcd : (int)(xcd % P); panel < P; panel += 8) {
    const float4* X4 = reinterpret_cast<const float4*>(Xp + (int64_t)panel * x_panel_stride) + lr;
    for (int c = sub + share * (int)w; c < t.C; c += share * (int)wpx) {
      for (int j = 0; j < t.RG; ++j) mine[j * 8] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int b0 = t.wp[c * 16 + wv], b1 = t.wp[c * 16 + wv + 1];
      int cur = -1;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int b = b0; b < b1; ++b) {
        const int64_t o = (int64_t)b * 64 + lane;
        const int my_c = t.col3[o];
        const float my_v = t.val3[o];
        const int my_r = t.row3[o];
        float4 x[8];
#define RG_LINE(k_) x[k_] = X4[(int64_t)RG_SWZ(my_c, k_) * x_row4]
        RG_LINE(0); RG_LINE(1); RG_LINE(2); RG_LINE(3); RG_LINE(4); RG_LINE(5); RG_LINE(6); RG_LINE(7);
#undef RG_LINE
        // one edge into the running sum of its row (a row is visited once per source block; its sum waits in LDS in between)
#define RG_EDGE(k_)                                                                              \
  do {                                                                                           \
    const int r_ = RG_SWZ(my_r, k_);                                                             \
    const float v_ = __int_as_float(RG_SWZ(__float_as_int(my_v), k_));                           \
    if (r_ != 0xFFFF) {                                                                          \
      if (r_ != cur) {                                                                           \
        if (cur >= 0) tile_acc[cur * 8 + lr] = acc;                                              \
        acc = tile_acc[r_ * 8 + lr];                                                             \
        cur = r_;                                                                                \
      }                                                                                          \
      acc.x = fmaf(v_, x[k_].x, acc.x); acc.y = fmaf(v_, x[k_].y, acc.y);                        \
      acc.z = fmaf(v_, x[k_].z, acc.z); acc.w = fmaf(v_, x[k_].w, acc.w);                        \
    }                                                                                            \
  } while (0)
        RG_EDGE(0); RG_EDGE(1); RG_EDGE(2); RG_EDGE(3); RG_EDGE(4); RG_EDGE(5); RG_EDGE(6); RG_EDGE(7);
#undef RG_EDGE
      }
      if (cur >= 0) tile_acc[cur * 8 + lr] = acc;
      const int64_t row0 = (int64_t)c * RC + (int64_t)gi * t.RG;
      for (int j = 0; j < t.RG; ++j) {
        if (row0 + j >= n) break;
        float4 v = mine[j * 8];
        v.x = apply_act(v.x, act, alpha); v.y = apply_act(v.y, act, alpha);
        v.z = apply_act(v.z, act, alpha); v.w = apply_act(v.w, act, alpha);
        reinterpret_cast<float4*>(Y + (int64_t)panel * y_panel_stride + (row0 + j) * y_row_stride)[lr] = v;
      }
    }
  }
#undef RG_SWZ
}

// ---- long rows, with a workspace ------------------------------------------------------------------------------------
struct LongRows {
  int* ctr;           // [0] number of long rows, [1] number of block tasks
  int64_t* rows;      // [max_rows]  row index of long row i
  int* pos;           // [max_rows]  first task / partial slot of long row i (its blocks follow in order)
  int64_t* task_row;  // [max_tasks]
  int* task_blk;      // [max_tasks]
  float* partial;     // [max_tasks, D] block sums (spmm) / [max_tasks] (softmax)
  int64_t max_rows, max_tasks;
};

// One thread per row: a long row reserves its task slots (a contiguous range, so its blocks stay in order; which range
// is decided by atomics and does not matter) and writes its tasks.
__global__ void __launch_bounds__(256) long_rows_kernel(const int64_t* __restrict__ rowptr, int64_t n, LongRows w) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  const int64_t deg = rowptr[row + 1] - rowptr[row];
  if (deg <= ROW_BLOCK) return;
  const int nb = (int)((deg + ROW_BLOCK - 1) / ROW_BLOCK);
  const int li = atomicAdd(w.ctr, 1);
  const int p = atomicAdd(w.ctr + 1, nb);
  w.rows[li] = row;
  w.pos[li] = p;
  for (int b = 0; b < nb; ++b) {
    w.task_row[p + b] = row;
    w.task_blk[p + b] = b;
  }
}

// One lane group per (long row, block): the block's chain -> partial[task].
template <int LPR>
__global__ void __launch_bounds__(256) spmm_long_blocks_kernel(const int64_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ col,
                                                               const float* __restrict__ val,
                                                               const float* __restrict__ X, int D, LongRows w) {
  constexpr int RPB = 256 / LPR;
  const int lr = threadIdx.x % LPR;
  const int gbase = (threadIdx.x & 63) - lr;
  const int64_t t = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
  const int ntask = w.ctr[1];
  const bool live = t < ntask;
  const int64_t tt = live ? t : 0;
  int64_t e0 = 0;
  int cnt = 0;
  if (ntask > 0) {
    const int64_t row = w.task_row[tt];
    const int64_t r0 = rowptr[row], r1 = rowptr[row + 1];
    e0 = r0 + (int64_t)w.task_blk[tt] * ROW_BLOCK;
    cnt = live ? (int)(r1 - e0 < ROW_BLOCK ? r1 - e0 : ROW_BLOCK) : 0;
  }
  const int D4 = D >> 2;
  const float4* X4 = reinterpret_cast<const float4*>(X);
  for (int c4 = lr; c4 < ((D4 + LPR - 1) / LPR) * LPR; c4 += LPR) {
    const bool colok = c4 < D4;
    const float4 acc = row_chain<LPR>(col, val, e0, cnt, X4, D4, c4, colok, lr, gbase);
    if (live && colok) reinterpret_cast<float4*>(w.partial)[t * D4 + c4] = acc;
  }
}

// One lane group per long row: its block sums in block order, then the epilogue.
template <int LPR>
__global__ void __launch_bounds__(256) spmm_long_finish_kernel(const int64_t* __restrict__ rowptr, int D,
                                                               const float* __restrict__ bias, int act, float alpha,
                                                               float beta, const float* __restrict__ Yin,
                                                               float* __restrict__ Y, LongRows w) {
  constexpr int RPB = 256 / LPR;
  const int lr = threadIdx.x % LPR;
  const int64_t li = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
  if (li >= w.ctr[0]) return;
  const int64_t row = w.rows[li];
  const int64_t deg = rowptr[row + 1] - rowptr[row];
  const int nb = (int)((deg + ROW_BLOCK - 1) / ROW_BLOCK);
  const int64_t p = w.pos[li];
  const int D4 = D >> 2;
  const float4* P4 = reinterpret_cast<const float4*>(w.partial);
  for (int c4 = lr; c4 < D4; c4 += LPR) {
    float4 acc = P4[p * D4 + c4];
    for (int b = 1; b < nb; ++b) acc = add4(acc, P4[(p + b) * D4 + c4]);
    spmm_epilogue_store(acc, bias, act, alpha, beta, Yin, Y, row, D4, c4);
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

// One thread per segment; a segment longer than ROW_BLOCK sums its exponentials in blocks (see ROW_BLOCK), itself when
// there is no workspace, else it is left to segment_softmax_long_kernel.
__global__ void __launch_bounds__(256) segment_softmax_kernel(const int64_t* __restrict__ rowptr,
                                                              const float* __restrict__ x, int64_t n,
                                                              float* __restrict__ out, int skip_long) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  if (e0 == e1) return;
  if (skip_long && e1 - e0 > ROW_BLOCK) return;
  float m = x[e0];
  for (int64_t e = e0 + 1; e < e1; ++e) m = fmaxf(m, x[e]);
  float s = 0.f;
  for (int64_t b0 = e0; b0 < e1; b0 += ROW_BLOCK) {
    const int64_t b1 = b0 + ROW_BLOCK < e1 ? b0 + ROW_BLOCK : e1;
    float sb = 0.f;
    for (int64_t e = b0; e < b1; ++e) {
      const float ex = expf(x[e] - m);
      out[e] = ex;
      sb = __fadd_rn(sb, ex);
    }
    s = (b0 == e0) ? sb : __fadd_rn(s, sb);
  }
  for (int64_t e = e0; e < e1; ++e) out[e] = out[e] / s;
}

// One workgroup per long segment: maximum by a tree (order-free), one thread per block for the block sums (sequential,
// as the contract says), thread 0 adds them in block order, everybody normalises.
__global__ void __launch_bounds__(256) segment_softmax_long_kernel(const int64_t* __restrict__ rowptr,
                                                                   const float* __restrict__ x,
                                                                   float* __restrict__ out, LongRows w) {
  __shared__ float red[256];
  const int64_t li = blockIdx.x;
  if (li >= w.ctr[0]) return;
  const int tid = threadIdx.x;
  const int64_t row = w.rows[li];
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  const int nb = (int)((e1 - e0 + ROW_BLOCK - 1) / ROW_BLOCK);
  float m = RG_NEG_INF;
  for (int64_t e = e0 + tid; e < e1; e += 256) m = fmaxf(m, x[e]);
  red[tid] = m;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) red[tid] = fmaxf(red[tid], red[tid + off]);
    __syncthreads();
  }
  m = red[0];
  __syncthreads();
  float* psum = w.partial + w.pos[li];
  for (int b = tid; b < nb; b += 256) {
    const int64_t b0 = e0 + (int64_t)b * ROW_BLOCK;
    const int64_t b1 = b0 + ROW_BLOCK < e1 ? b0 + ROW_BLOCK : e1;
    float sb = 0.f;
    for (int64_t e = b0; e < b1; ++e) {
      const float ex = expf(x[e] - m);
      out[e] = ex;
      sb = __fadd_rn(sb, ex);
    }
    psum[b] = sb;
  }
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    float s = psum[0];
    for (int b = 1; b < nb; ++b) s = __fadd_rn(s, psum[b]);
    red[0] = s;
  }
  __syncthreads();
  const float s = red[0];
  for (int64_t e = e0 + tid; e < e1; e += 256) out[e] = out[e] / s;
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
    // (eight rows' loads in flight; the additions stay in row order)
#pragma unroll 8
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

// The same for any D / alignment: thread t owns columns t, t+256, ...; identical additions in identical order.
__global__ void __launch_bounds__(256) segment_reduce_scalar_kernel(const float* __restrict__ X, int D,
                                                                    const int64_t* __restrict__ seg_ptr,
                                                                    const float* __restrict__ w, int mean_mode,
                                                                    float* __restrict__ out) {
  const int64_t g = blockIdx.x;
  const int64_t r0 = seg_ptr[g], r1 = seg_ptr[g + 1];
  for (int c = threadIdx.x; c < D; c += 256) {
    float acc = 0.f;
    const float wv = w ? w[c] : 1.f;
#pragma unroll 8
    for (int64_t r = r0; r < r1; ++r) {
      float x = X[r * D + c];
      if (w) x = __fmul_rn(wv, x);
      acc = __fadd_rn(acc, x);
    }
    if (mean_mode) acc = acc / (float)(r1 - r0);
    out[g * D + c] = acc;
  }
}

}  // namespace ragraph

using namespace ragraph;

// Workspace of the long-row path: counters, long-row list, block tasks, block sums.
static size_t long_rows_layout(int64_t nnz, int D, LongRows* w, char* base) {
  const int64_t max_rows = nnz / ROW_BLOCK + 1, max_tasks = 2 * (nnz / ROW_BLOCK) + 2;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  char* c = take(2 * sizeof(int));
  char* r = take((size_t)max_rows * sizeof(int64_t));
  char* ps = take((size_t)max_rows * sizeof(int));
  char* tr = take((size_t)max_tasks * sizeof(int64_t));
  char* tb = take((size_t)max_tasks * sizeof(int));
  char* pa = take((size_t)max_tasks * (size_t)(D > 0 ? D : 1) * sizeof(float));
  if (w) {
    w->ctr = reinterpret_cast<int*>(c);
    w->rows = reinterpret_cast<int64_t*>(r);
    w->pos = reinterpret_cast<int*>(ps);
    w->task_row = reinterpret_cast<int64_t*>(tr);
    w->task_blk = reinterpret_cast<int*>(tb);
    w->partial = reinterpret_cast<float*>(pa);
    w->max_rows = max_rows;
    w->max_tasks = max_tasks;
  }
  return off;
}

extern "C" size_t ragraph_sparse_workspace_bytes(int64_t nnz, int D) {
  if (nnz < 0 || D < 0) return 0;
  return long_rows_layout(nnz, D, nullptr, nullptr);
}

static int find_long_rows(const int64_t* rowptr, int64_t n, const LongRows& w, hipStream_t st) {
  if (hipMemsetAsync(w.ctr, 0, 2 * sizeof(int), st) != hipSuccess) {
    set_error("sparse: memset failed");
    return RAGRAPH_EDEVICE;
  }
  hipLaunchKernelGGL(long_rows_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, rowptr, n, w);
  RG_CHECK_LAUNCH("sparse(long rows)");
  return RAGRAPH_OK;
}

extern "C" int ragraph_spmm_csr_ws_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n,
                                       const float* X, int D, const float* bias, int act, float alpha, float beta,
                                       const float* Y_in, float* Y, int64_t nnz, void* ws, size_t ws_bytes,
                                       void* stream) {
  RG_REQUIRE(rowptr && X && Y, RAGRAPH_EINVAL, "spmm_csr: null pointer");
  RG_REQUIRE(n >= 0 && D >= 4 && (D & 3) == 0, RAGRAPH_EINVAL, "spmm_csr: D=%d must be a positive multiple of 4", D);
  RG_REQUIRE(aligned16(X) && aligned16(Y) && (!bias || aligned16(bias)) && (!Y_in || aligned16(Y_in)), RAGRAPH_EINVAL,
             "spmm_csr: X, Y, bias, Y_in must be 16-B aligned");
  RG_REQUIRE(X != Y, RAGRAPH_EINVAL, "spmm_csr: Y must not alias X");
  RG_REQUIRE(act >= RAGRAPH_ACT_NONE && act <= RAGRAPH_ACT_ELU, RAGRAPH_EINVAL, "spmm_csr: bad act %d", act);
  if (n == 0) return RAGRAPH_OK;
  hipStream_t st = as_stream(stream);
  LongRows w{};
  const bool par = ws != nullptr && nnz > ROW_BLOCK;  // (no row can be long otherwise)
  if (par) {
    RG_REQUIRE(aligned16(ws) && ws_bytes >= long_rows_layout(nnz, D, nullptr, nullptr), RAGRAPH_EWORKSPACE,
               "spmm_csr: workspace %zu < %zu", ws_bytes, long_rows_layout(nnz, D, nullptr, nullptr));
    long_rows_layout(nnz, D, &w, static_cast<char*>(ws));
    const int rc = find_long_rows(rowptr, n, w, st);
    if (rc != RAGRAPH_OK) return rc;
  }
  const int skip = par ? 1 : 0;
  const int D4 = D >> 2;
  static const unsigned xcd_run = [] {  // workgroups of one XCD's contiguous run (see spmm_csr_kernel)
    const char* e = getenv("RAGRAPH_SPMM_XCD_RUN");
    const int v = e ? atoi(e) : 32;
    return (unsigned)(v < 1 ? 1 : v);
  }();
#define RG_SPMM(LPR_)                                                                                                  \
  do {                                                                                                                 \
    constexpr int RPB_ = 256 / (LPR_);                                                                                 \
    hipLaunchKernelGGL(spmm_csr_kernel<LPR_>, dim3((unsigned)cdiv(n, RPB_)), dim3(256), 0, st, rowptr, col, val, n, X, \
                       D, bias, act, alpha, beta, Y_in, Y, skip, xcd_run);                                             \
    if (par) {                                                                                                         \
      hipLaunchKernelGGL(spmm_long_blocks_kernel<LPR_>, dim3((unsigned)cdiv(w.max_tasks, RPB_)), dim3(256), 0, st,     \
                         rowptr, col, val, X, D, w);                                                                   \
      hipLaunchKernelGGL(spmm_long_finish_kernel<LPR_>, dim3((unsigned)cdiv(w.max_rows, RPB_)), dim3(256), 0, st,      \
                         rowptr, D, bias, act, alpha, beta, Y_in, Y, w);                                               \
    }                                                                                                                  \
  } while (0)
  if (D4 <= 16) RG_SPMM(16);
  else if (D4 <= 32) RG_SPMM(32);
  else RG_SPMM(64);
#undef RG_SPMM
  RG_CHECK_LAUNCH("spmm_csr");
  return RAGRAPH_OK;
}

extern "C" int ragraph_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n,
                                    const float* X, int D, const float* bias, int act, float alpha, float beta,
                                    const float* Y_in, float* Y, void* stream) {
  return ragraph_spmm_csr_ws_f32(rowptr, col, val, n, X, D, bias, act, alpha, beta, Y_in, Y, 0, nullptr, 0, stream);
}

extern "C" int ragraph_spmm_csr_panels_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n, const float* X,
                                           int64_t x_rows, int x_panels, int D, int act, float alpha, float* Y, int y_panels,
                                           void* stream) {
  RG_REQUIRE(rowptr && X && Y, RAGRAPH_EINVAL, "spmm_csr_panels: null pointer");
  RG_REQUIRE(x_rows >= 1, RAGRAPH_EINVAL, "spmm_csr_panels: x_rows=%lld", (long long)x_rows);
  RG_REQUIRE(n >= 1 && (D == 64 || D == 128 || D == 256 || (D % 256 == 0 && D <= 2048)), RAGRAPH_EUNSUPPORTED,
             "spmm_csr_panels: D=%d (panels of 32 columns: 2, 4 or a multiple of 8 of them)", D);
  RG_REQUIRE(aligned16(X) && aligned16(Y) && X != Y, RAGRAPH_EINVAL, "spmm_csr_panels: X, Y must be 16-B aligned and distinct");
  RG_REQUIRE(act >= RAGRAPH_ACT_NONE && act <= RAGRAPH_ACT_ELU, RAGRAPH_EINVAL, "spmm_csr_panels: bad act %d", act);
  hipStream_t st = as_stream(stream);
  const int P = D / 32;
  // row-major in: the XCD-sliced kernel reading its 128-byte slice of every row (RAGRAPH_SPMM_ROW_SLICES=0: the row kernel)
  static const bool row_slices = [] { const char* e = getenv("RAGRAPH_SPMM_ROW_SLICES"); return !e || atoi(e) != 0; }();
  if (!x_panels && !row_slices) {  // the row kernel, storing row-major or panel-major
    static const unsigned xcd_run = [] {
      const char* e = getenv("RAGRAPH_SPMM_XCD_RUN");
      const int v = e ? atoi(e) : 32;
      return (unsigned)(v < 1 ? 1 : v);
    }();
    const int D4 = D >> 2;
    const int64_t pn = y_panels ? n : 0;
#define RG_SPMM_P(LPR_)                                                                                                     \
  hipLaunchKernelGGL(spmm_csr_kernel<LPR_>, dim3((unsigned)cdiv(n, 256 / (LPR_))), dim3(256), 0, st, rowptr, col, val, n, X, D, \
                     (const float*)nullptr, act, alpha, 0.f, (const float*)nullptr, Y, 0, xcd_run, pn)
    if (D4 <= 16) RG_SPMM_P(16);
    else if (D4 <= 32) RG_SPMM_P(32);
    else RG_SPMM_P(64);
#undef RG_SPMM_P
    RG_CHECK_LAUNCH("spmm_csr_panels");
    return RAGRAPH_OK;
  }
  const int64_t row_blocks = cdiv(n, 32);
  const int64_t per_xcd = P >= 8 ? row_blocks * (P / 8) : cdiv(row_blocks, (int64_t)(8 / P));
  RG_REQUIRE(per_xcd * 8 < ((int64_t)1 << 31), RAGRAPH_EUNSUPPORTED, "spmm_csr_panels: too many rows");
  hipLaunchKernelGGL(spmm_panel_kernel, dim3((unsigned)(per_xcd * 8)), dim3(256), 0, st, rowptr, col, val, n, X,
                     x_panels ? x_rows * 32 : (int64_t)32, x_panels ? 8 : D / 4, Y, y_panels ? n * 32 : (int64_t)32,
                     y_panels ? (int64_t)32 : (int64_t)D, P, act, alpha);
  RG_CHECK_LAUNCH("spmm_csr_panels");
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

extern "C" int ragraph_segment_softmax_ws_f32(const int64_t* rowptr, const float* x, int64_t n, int64_t nnz, float* out,
                                              void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(rowptr && x && out, RAGRAPH_EINVAL, "segment_softmax: null pointer");
  if (n <= 0) return RAGRAPH_OK;
  hipStream_t st = as_stream(stream);
  LongRows w{};
  const bool par = ws != nullptr && nnz > ROW_BLOCK;
  if (par) {
    RG_REQUIRE(aligned16(ws) && ws_bytes >= long_rows_layout(nnz, 0, nullptr, nullptr), RAGRAPH_EWORKSPACE,
               "segment_softmax: workspace %zu < %zu", ws_bytes, long_rows_layout(nnz, 0, nullptr, nullptr));
    long_rows_layout(nnz, 0, &w, static_cast<char*>(ws));
    const int rc = find_long_rows(rowptr, n, w, st);
    if (rc != RAGRAPH_OK) return rc;
  }
  hipLaunchKernelGGL(segment_softmax_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, rowptr, x, n, out,
                     par ? 1 : 0);
  if (par)
    hipLaunchKernelGGL(segment_softmax_long_kernel, dim3((unsigned)w.max_rows), dim3(256), 0, st, rowptr, x, out, w);
  RG_CHECK_LAUNCH("segment_softmax");
  return RAGRAPH_OK;
}

extern "C" int ragraph_segment_softmax_f32(const int64_t* rowptr, const float* x, int64_t n, float* out,
                                           void* stream) {
  return ragraph_segment_softmax_ws_f32(rowptr, x, n, 0, out, nullptr, 0, stream);
}

extern "C" int ragraph_segment_reduce_f32(const float* X, int D, const int64_t* seg_ptr, int64_t G, const float* w,
                                          int mean_mode, float* out, void* stream) {
  RG_REQUIRE(X && seg_ptr && out, RAGRAPH_EINVAL, "segment_reduce: null pointer");
  RG_REQUIRE(D >= 1, RAGRAPH_EINVAL, "segment_reduce: D=%d must be positive", D);
  if (G <= 0) return RAGRAPH_OK;
  if ((D & 3) == 0 && aligned16(X) && aligned16(out) && (!w || aligned16(w)))
    hipLaunchKernelGGL(segment_reduce_kernel, dim3((unsigned)G), dim3(256), 0, as_stream(stream), X, D, seg_ptr, w,
                       mean_mode, out);
  else
    hipLaunchKernelGGL(segment_reduce_scalar_kernel, dim3((unsigned)G), dim3(256), 0, as_stream(stream), X, D, seg_ptr, w,
                       mean_mode, out);
  RG_CHECK_LAUNCH("segment_reduce");
  return RAGRAPH_OK;
}

// a7 over a TILED graph (spmm_tiled_kernel): wp / col3 / val3 / row3 = the plan of ragraph_amd/graph.py (CSRGraph.tile_plan):
// C chunks of 128 RG destination rows, their edges in wave-batches of 64 slots.  Layouts as ragraph_spmm_csr_panels_f32.  Same
// bits as the other SpMM entries for a graph whose columns ascend inside every row.
extern "C" int ragraph_spmm_csr_tiled_f32(const int* wp, const int* col3, const float* val3, const unsigned short* row3, int RG,
                                          int C, int64_t n, const float* X, int64_t x_rows, int x_panels, int D, int act,
                                          float alpha, float* Y, int y_panels, void* stream) {
  RG_REQUIRE(wp && col3 && val3 && row3 && X && Y, RAGRAPH_EINVAL, "spmm_csr_tiled: null pointer");
  RG_REQUIRE(n >= 1 && x_rows >= 1 && (D == 64 || D == 128 || D == 256 || (D % 256 == 0 && D <= 2048)), RAGRAPH_EUNSUPPORTED,
             "spmm_csr_tiled: D=%d (panels of 32 columns: 2, 4 or a multiple of 8 of them)", D);
  RG_REQUIRE(RG >= 1 && RG <= 9 && C >= 1 && (int64_t)C * RG * 128 >= n, RAGRAPH_EINVAL,
             "spmm_csr_tiled: bad plan (RG=%d C=%d n=%lld)", RG, C, (long long)n);
  RG_REQUIRE(aligned16(X) && aligned16(Y) && X != Y, RAGRAPH_EINVAL, "spmm_csr_tiled: X, Y must be 16-B aligned and distinct");
  RG_REQUIRE(act >= RAGRAPH_ACT_NONE && act <= RAGRAPH_ACT_ELU, RAGRAPH_EINVAL, "spmm_csr_tiled: bad act %d", act);
  static DeviceOnce once;
  const int lds = RG * 128 * 128;   // RG x 16 KiB
  if (raise_dynamic_lds(once, spmm_tiled_kernel, 9 * 128 * 128) != hipSuccess) {
    set_error("spmm_csr_tiled: cannot raise the dynamic LDS limit");
    return RAGRAPH_EDEVICE;
  }
  TilePlan t{wp, col3, val3, row3, RG, C};
  const int P = D / 32;
  hipLaunchKernelGGL(spmm_tiled_kernel, dim3((unsigned)device_cus_multiple_of_8()), dim3(1024), (size_t)lds, as_stream(stream), t, n, X,
                     x_panels ? x_rows * 32 : (int64_t)32, x_panels ? 8 : D / 4, Y, y_panels ? n * 32 : (int64_t)32,
                     y_panels ? (int64_t)32 : (int64_t)D, P, act, alpha);
  RG_CHECK_LAUNCH("spmm_csr_tiled");
  return RAGRAPH_OK;
}
