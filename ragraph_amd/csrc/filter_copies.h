// The bank copies of the filtered top-k: bf16 and two-scale int8 images in MFMA fragment order (ragraph_keys_to_bf16).
// Part of csrc/topk_filter.hip (textually included there, inside its namespace / after its helpers): split out in round 6 so
// that the ring, the candidate path and the launch plumbing can be read -- and changed -- apart.  No include guard on purpose:
// these are not stand-alone headers.

// fp32 -> bf16 (round to nearest even) of the bank in MFMA fragment order, rows [N, Npad) zero so the stream never needs
// a tail clamp, and
// max_k |dk|^2 of the bank (see FILTER_EPS_SLACK) by an integer max: non-negative floats order like their bit patterns.
template <int D>
__global__ void __launch_bounds__(256) keys_to_bf16_kernel(const float* __restrict__ Kn, int64_t N, int64_t Npad,
                                                           uint16_t* __restrict__ Kb, unsigned* __restrict__ max_err2) {
  constexpr int TPR = D / 8;  // threads per row (one thread = 8 elements): 8 / 16 / 32, inside one half-wave
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = i / TPR;
  bf16x8 o;
  float e2 = 0.f;
  if (i < Npad * TPR && row < N) {
    const float4 a = reinterpret_cast<const float4*>(Kn)[2 * i], b = reinterpret_cast<const float4*>(Kn)[2 * i + 1];
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o[e] = (__bf16)x[e];
      const float d = x[e] - (float)o[e];
      e2 = fmaf(d, d, e2);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)0.f;
  }
  if (i < Npad * TPR) {
    // fragment order (filter_common.h): 16-B piece c = 4 t + g of key row 32 u + 16 h + j goes to block 2 t + h of
    // sub-tile u, lane j + 16 g
    const int c = (int)(i % TPR);
    const int64_t dst = filter_block_offset(row >> 5, D, c >> 2, (int)(row >> 4) & 1) + (((c & 3) * 16 + (int)(row & 15)) << 4);
    *reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(Kb) + dst) = o;
  }
#pragma unroll
  for (int off = TPR / 2; off >= 1; off >>= 1) e2 += __shfl_xor(e2, off);
  // (a plain read first: after the first few rows almost none beats the running maximum, so almost none pays for the
  // atomic on this one address)
  if ((threadIdx.x & (TPR - 1)) == 0 &&
      __float_as_uint(e2) > __hip_atomic_load(max_err2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(max_err2, __float_as_uint(e2));
}

// The int8 copy (filter_common.h, "TWO SCALES"): the granules' largest |k_i| (and the bank's), the cut between the two classes,
// then quantise + lay out + each class's largest |dk|^2 + the class bits.  The tail row and the class words are zeroed by the
// caller before the first kernel.
template <int D>
__global__ void __launch_bounds__(256) i8_granule_absmax_kernel(const float* __restrict__ Kn, int64_t N, float* __restrict__ gmax,
                                                                unsigned* __restrict__ tail8) {
  constexpr int GK = filter_i8_granule_keys(D);
  const int64_t row0 = (int64_t)blockIdx.x * GK;
  const int64_t rows = N - row0 < GK ? N - row0 : GK;   // (<= 0: a granule of padding)
  const int64_t n4 = rows > 0 ? rows * (D / 4) : 0;
  const float4* src = reinterpret_cast<const float4*>(Kn + row0 * D);
  unsigned m = 0u;
  for (int64_t i = threadIdx.x; i < n4; i += 256) {
    const float4 v = src[i];
    m = max(max(m, __float_as_uint(fabsf(v.x))), max(__float_as_uint(fabsf(v.y)), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w)))));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
  __shared__ unsigned wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    gmax[blockIdx.x] = __uint_as_float(m);
    if (m > __hip_atomic_load(tail8 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(tail8 + 2, m);
  }
}

// The cut (one workgroup).  A histogram of the granules' maxima over the upper 15 bits of their (non-negative) float
// patterns -- bins 1.6 % wide -- then, for every occupied bin's upper edge x as the cut, the modelled candidates
//   (granules <= x) f(x) + (granules > x) f(x_max),   f(x) = exp(lambda |dk|(x)),  |dk|(x) = (x / 127) sqrt(D / 12)
// (uniform rounding errors of a row quantised on the grid x / 127; lambda = 60: the bench bank's candidates triple when eps
// grows from the bf16 bound's 0.004 to the single scale's 0.0215).  Whatever comes out is only a matter of speed: every
// class's error is MEASURED by the quantising kernel and the bounds use the measurements.  lambda <= 0: cut = the maximum.
constexpr int I8_CUT_BINS = 1 << 13;   // (unit rows: |k_i| <= 1 = bin 8128; anything larger shares the last bin, which is never a cut)
__global__ void __launch_bounds__(1024) i8_cut_kernel(const float* __restrict__ gmax, int64_t granules, int D, float lambda,
                                                      unsigned* __restrict__ tail8) {
  __shared__ int hist[I8_CUT_BINS];
  __shared__ int tsum[1024];
  __shared__ float tcost[1024];
  __shared__ int tbin[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < I8_CUT_BINS; i += 1024) hist[i] = 0;
  __syncthreads();
  for (int64_t i = tid; i < granules; i += 1024) atomicAdd(hist + min((int)(__float_as_uint(gmax[i]) >> 17), I8_CUT_BINS - 1), 1);
  __syncthreads();
  // thread t owns bins 16 t .. 16 t + 15: their sum, an exclusive prefix over the threads, then the cost of every occupied bin's
  // upper edge as the cut; the cheapest (ties: the lowest) wins
  constexpr int PER = I8_CUT_BINS / 1024;
  int mine = 0;
#pragma unroll
  for (int e = 0; e < PER; ++e) mine += hist[tid * PER + e];
  tsum[tid] = mine;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // (Hillis-Steele inclusive scan)
    const int v = tid >= off ? tsum[tid - off] : 0;
    __syncthreads();
    tsum[tid] += v;
    __syncthreads();
  }
  const float xmax = __uint_as_float(tail8[2]);
  const int top = min((int)(__float_as_uint(xmax) >> 17), I8_CUT_BINS - 1);
  const float c = lambda * sqrtf((float)D / 12.f) / 127.f;
  const float fmax_ = expf(c * xmax);
  float best = __builtin_huge_valf();
  int best_bin = -1;
  if (lambda > 0.f && xmax > 0.f) {
    int64_t below = tsum[tid] - mine;
    for (int e = tid * PER; e < tid * PER + PER && e < top; ++e) {
      if (hist[e] == 0) continue;
      below += hist[e];
      const float edge = __uint_as_float((unsigned)(e + 1) << 17);   // every maximum of bins <= e lies below it
      const float cost = (float)below * expf(c * edge) + (float)(granules - below) * fmax_;
      if (cost < best) {
        best = cost;
        best_bin = e;
      }
    }
  }
  tcost[tid] = best;
  tbin[tid] = best_bin;
  __syncthreads();
  for (int off = 512; off >= 1; off >>= 1) {
    if (tid < off && (tcost[tid + off] < tcost[tid] || (tcost[tid + off] == tcost[tid] && tbin[tid + off] >= 0 &&
                                                         (tbin[tid] < 0 || tbin[tid + off] < tbin[tid])))) {
      tcost[tid] = tcost[tid + off];
      tbin[tid] = tbin[tid + off];
    }
    __syncthreads();
  }
  if (tid != 0) return;
  float cut = xmax;   // (one class: no occupied bin below the top one, lambda <= 0, a zero bank -- or no cut beats it)
  if (tbin[0] >= 0 && tcost[0] < (float)granules * fmax_) cut = __uint_as_float((unsigned)(tbin[0] + 1) << 17);
  tail8[5] = __float_as_uint(cut);
  tail8[1] = __float_as_uint(cut / 127.f);
  tail8[4] = __float_as_uint(xmax / 127.f);
  tail8[7] = granules > INT_MAX ? (unsigned)INT_MAX : (unsigned)granules;
}

template <int D>
__global__ void __launch_bounds__(256) keys_to_i8_kernel(const float* __restrict__ Kn, int64_t N, int64_t Npad,
                                                         signed char* __restrict__ Kb8, unsigned* __restrict__ tail8,
                                                         const float* __restrict__ gmax, unsigned* __restrict__ cls) {
  constexpr int TPR = D / 16;  // threads per row (one thread = 16 elements = one lane's piece of a block): 4 / 8 / 16
  constexpr int GK = filter_i8_granule_keys(D);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = i / TPR;
  const int64_t gr = row / GK;
  const bool live = i < Npad * TPR;
  const bool heavy = live && gmax[gr] > __uint_as_float(tail8[5]);
  const float sk = __uint_as_float(tail8[heavy ? 4 : 1]);
  const float inv_sk = sk > 0.f ? 1.f / sk : 0.f;
  if (heavy && row == gr * GK && i == row * TPR) {   // the granule's first thread: its class bit
    atomicOr(cls + (gr >> 5), 1u << (gr & 31));
    atomicAdd(tail8 + 6, 1u);
  }
  unsigned w[4] = {0u, 0u, 0u, 0u};
  float e2 = 0.f;
  if (live && row < N && sk > 0.f) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 a = reinterpret_cast<const float4*>(Kn)[4 * i + c];
      const float x[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int qi = quantize_i8(x[e], inv_sk);
        w[c] |= ((unsigned)qi & 0xFFu) << (8 * e);
        const float d = fmaf(sk, (float)qi, -x[e]);
        e2 = fmaf(d, d, e2);
      }
    }
  }  // (an all-zero bank: the copy is zero, the error is the row itself: 0)
  if (live) {
    const int c = (int)(i % TPR);  // piece c = 4 t + g of the row
    const int64_t dst = filter_i8_block_offset(row >> 5, D, c >> 2, (int)(row >> 4) & 1) + (((c & 3) * 16 + (int)(row & 15)) << 4);
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Kb8) + dst) = make_uint4(w[0], w[1], w[2], w[3]);
  }
#pragma unroll
  for (int off = TPR / 2; off >= 1; off >>= 1) e2 += __shfl_xor(e2, off);
  e2 *= 1.000001f;  // (the fmaf's rounding of each difference)
  unsigned* slot = tail8 + (heavy ? 3 : 0);
  if ((threadIdx.x & (TPR - 1)) == 0 && __float_as_uint(e2) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(slot, __float_as_uint(e2));
}
