// Behind the last level: the verdict on a speculative first bound (single bank; merged rows of a sharded bank) and the sliced exact scans of the listed queries.
// Part of csrc/topk_filter.hip (textually included there, inside its namespace / after its helpers): split out in round 6 so
// that the ring, the candidate path and the launch plumbing can be read -- and changed -- apart.  No include guard on purpose:
// these are not stand-alone headers.

// A call that filtered with a SPECULATIVE first bound (ragraph_topk_cosine_filtered_set_prior: theta = the prior for every
// query, no bound pass) is exact for a query iff its final k-th best candidate scores at least the prior: a level filtered
// with theta_l = max(prior, the running exact k-th best) <= the final k-th best, so every key that scores at least the
// final k-th best passed its level.  A query whose k-th best is below the prior (or that found fewer than k candidates) is
// listed for the exact scan of the fixup launch behind this one, like a query whose list overflowed.  The same pass
// records the smallest / largest final k-th best of the call (stats[18] / [19]): the next call's prior comes from them.
// (Zero queries -- flag 2 -- are answered without a scan and not judged; queries already listed by the final level
// -- flag 1 / an overflowed list -- carry -inf or stale rows: listed twice would be scanned twice, so they are skipped by
// their flag.)
__global__ void __launch_bounds__(256) filter_verify_prior_kernel(const float* __restrict__ out_s, int64_t B, int k, float prior,
                                                                  int speculative, const unsigned char* __restrict__ flag,
                                                                  int* __restrict__ overflow, int* __restrict__ overflow_list,
                                                                  int* __restrict__ stats) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int lo = INT_MAX, hi = INT_MIN, failed = 0;
  if (q < B && flag[q] != 2) {
    const float kth = out_s[q * k + k - 1];
    if (speculative && flag[q] == 0 && !(kth >= prior)) {
      overflow_list[atomicAdd(overflow, 1)] = (int)q;
      failed = 1;
    } else if (kth > RG_NEG_INF) {
      lo = hi = f2ord(kth);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off));
    hi = max(hi, __shfl_xor(hi, off));
    failed += __shfl_xor(failed, off);
  }
  // one set of atomics per WORKGROUP (a returning atomic on one address costs ~11 ns chip-wide: 1600 waves of a 100 000-query
  // call at three each were 38 us of a launch that reads 400 KB)
  __shared__ int red[3][4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][w] = lo;
    red[1][w] = hi;
    red[2][w] = failed;
  }
  __syncthreads();
  if (threadIdx.x == 0 && stats) {
    lo = min(min(red[0][0], red[0][1]), min(red[0][2], red[0][3]));
    hi = max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3]));
    failed = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    if (lo != INT_MAX) atomicMin(stats + 18, lo);
    if (hi != INT_MIN) atomicMax(stats + 19, hi);
    if (failed) atomicAdd(stats + 17, failed);
  }
}

// Sharded banks under a speculative first bound: the verdict of the rows' OWNER, behind the merge of the shards' lists.  One
// workgroup walks the R merged rows: a row is proven iff its k-th best reaches the prior (an all-zero query -- every score +0 --
// is answered by index order and needs no proof); out[0] = rows that missed, out[1] = -(smallest proven k-th best), out[2] =
// the largest, out[3] = this shard's candidates per query over its levels (the call's statistics words; -1 without them),
// out[4] = its overflowed lists: five numbers that ONE all_reduce MAX turns into the group's (ragraph_amd/sharded.py).
__global__ void __launch_bounds__(256) verify_merged_prior_kernel(const float* __restrict__ s, int64_t R, int k, float prior,
                                                                  int speculative, const int* __restrict__ words,
                                                                  const int* __restrict__ overflow, float* __restrict__ out) {
  __shared__ float red[3][4];
  float miss = 0.f, neg_lo = RG_NEG_INF, hi = RG_NEG_INF;
  for (int64_t q = threadIdx.x; q < R; q += 256) {
    const float top = s[q * k], kth = s[q * k + k - 1];
    const bool zero = top == 0.f && kth == 0.f;
    const bool ok = zero || !speculative || kth >= prior;
    if (!ok) miss += 1.f;
    else if (!zero && kth > RG_NEG_INF) {
      neg_lo = fmaxf(neg_lo, -kth);
      hi = fmaxf(hi, kth);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    miss += __shfl_xor(miss, off);
    neg_lo = fmaxf(neg_lo, __shfl_xor(neg_lo, off));
    hi = fmaxf(hi, __shfl_xor(hi, off));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][w] = miss;
    red[1][w] = neg_lo;
    red[2][w] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    out[1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    out[2] = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
    float cand = -1.f;
    if (words && words[0] == FILTER_STATS_MAGIC) {
      cand = 0.f;
      for (int l = 0; l < 3; ++l)
        if (words[5 + l] > 0) cand += (float)words[2 + l] / (float)words[5 + l];
    }
    out[3] = cand;
    out[4] = overflow ? (float)*overflow : 0.f;
  }
}

// Large batches (the one-wave-per-query rescoring kernels): the final level has listed the overflowed queries, and this
// launch -- a fixed grid that finds an empty list on ordinary banks and returns -- runs exact_scan_query for each.
// (Below 2048 queries the workgroup-per-query rescoring kernels call it themselves and this launch is not made.)
// The exact scan of the queries the final level could not serve, in ONE launch behind it (calls of 65 queries and more;
// smaller ones scan inside their rescoring launch).  Few overflowed queries -- the usual case when there are any: a
// tight cluster next to a handful of queries -- would leave the chip idle behind one workgroup per query (25 ms per
// scan of 1M x 256 keys; 94 ms when the query's own rescoring wave did it), so a query's scan is cut into up to
// FILTER_FIX_SLICES key slices (as many as keep ~256 workgroups busy), each workgroup leaves its slice's k winners in
// part_s / part_i, and the query's last slice to finish (a ticket) merges them: 1.6 ms for one query.  Many overflowed
// queries take one workgroup each as before.  A ZERO query is answered without a scan.
template <int D>
__global__ void __launch_bounds__(256) topk_overflow_fixup_kernel(const float* __restrict__ Qn, const float* __restrict__ Kn,
                                                                  int64_t N, int k, int64_t idx_base,
                                                                  const int* __restrict__ overflow,
                                                                  const int* __restrict__ overflow_list,
                                                                  int64_t* __restrict__ overflow_idx_out,
                                                                  float* __restrict__ out_s, int64_t* __restrict__ out_i,
                                                                  int* __restrict__ done, float* __restrict__ part_s,
                                                                  int64_t* __restrict__ part_i, int64_t B,
                                                                  const unsigned char* __restrict__ flag,
                                                                  int* __restrict__ stats) {
  __shared__ float4 qs[D / 4];
  __shared__ __attribute__((aligned(16))) float tile[4][64 * RESCORE_LD];
  __shared__ float ps[4][32];
  __shared__ int64_t pi[4][32];
  __shared__ int ticket_s;
  const int n_over = *overflow;
  if (stats && blockIdx.x == 0 && threadIdx.x == 0) stats[20] = n_over;   // (final: every launch that counts runs before this one)
  if (stats && stats[16] == 0) {
    // the smallest / largest final k-th best score of the call's queries (stats[18] / [19]; a speculative call's verify
    // launch has recorded them already): one value per thread, wave-reduced, two atomics per wave that saw any.
    int lo = INT_MAX, hi = INT_MIN;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < B; q += (int64_t)gridDim.x * 256) {
      // (zero queries -- flag 2 -- have no k-th best; rows flagged 1 are listed for the scans below and hold what torch.empty
      // left or a stale candidate row that the scans rewrite during this very launch: neither may reach the history words)
      const float kth = flag[q] != 0 ? RG_NEG_INF : out_s[q * k + k - 1];
      if (kth > RG_NEG_INF) {
        lo = min(lo, f2ord(kth));
        hi = max(hi, f2ord(kth));
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      lo = min(lo, __shfl_xor(lo, off));
      hi = max(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
      if (lo != INT_MAX) atomicMin(stats + 18, lo);
      if (hi != INT_MIN) atomicMax(stats + 19, hi);
    }
  }
  if (n_over <= 0) return;
  int SL = 1;
  if (n_over <= FILTER_FIX_MAX_Q)
    while (SL < FILTER_FIX_SLICES && 2 * SL * n_over <= (int)gridDim.x) SL *= 2;
  const int64_t chunk = ((N + SL - 1) / SL + 63) / 64 * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t item = blockIdx.x; item < (int64_t)n_over * SL; item += gridDim.x) {
    const int o = (int)(item / SL), sl = (int)(item % SL);
    const int64_t b = overflow_list[o];
    if (overflow_idx_out && threadIdx.x == 0 && sl == 0) overflow_idx_out[o] = b;
    __syncthreads();  // (the previous item's readers of qs)
    if (threadIdx.x < D / 4) qs[threadIdx.x] = reinterpret_cast<const float4*>(Qn + b * D)[threadIdx.x];
    __syncthreads();
    if (SL == 1) {
      exact_scan_query<D>(qs, Kn, N, k, idx_base, tile, ps, pi, out_s + b * k, out_i + b * k);
      continue;
    }
    const int64_t lo = sl * chunk, hi = lo + chunk < N ? lo + chunk : N;
    float* my_s = part_s + ((int64_t)o * SL + sl) * 32;
    int64_t* my_i = part_i + ((int64_t)o * SL + sl) * 32;
    if (lo < hi) {
      exact_scan_query<D>(qs, Kn + lo * D, hi - lo, k, lo, tile, ps, pi, my_s, my_i);   // (indices local to the bank)
    } else if (threadIdx.x < k) {
      my_s[threadIdx.x] = RG_NEG_INF;
      my_i[threadIdx.x] = INT64_MAX;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) ticket_s = atomicAdd(done + o, 1);
    __syncthreads();
    if (ticket_s != SL - 1) continue;  // (workgroup-uniform)
    __threadfence();
    if (w == 0) {  // the query's last slice: SL k <= 512 partial winners, eight slots per lane
      float s8[8];
      int id8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = lane + 64 * u;
        s8[u] = RG_NEG_INF;
        id8[u] = INT_MAX;
        if (e < SL * k) {
          const int64_t at = ((int64_t)o * SL + e / k) * 32 + e % k;
          const int64_t pv = __builtin_nontemporal_load(part_i + at);
          if (pv < INT_MAX) {
            s8[u] = __builtin_nontemporal_load(part_s + at);
            id8[u] = (int)pv;
          }
        }
      }
      wave_select<8>(s8, id8, k, lane, idx_base, out_s + b * k, out_i + b * k);
    }
  }
}
