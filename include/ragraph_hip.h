/*
 * ragraph_hip.h -- C ABI of libragraph_hip.so: the MI355X (gfx950) retrieve-and-propagate hot path of RAGraph.
 *
 * The reference (Artessay/RAGraph) has no FFI: its hot path is in-process torch calls.  This header is the
 * boundary a maintainer binds instead of those torch op chains (ctypes stub: INTEGRATION.md).  Each entry point
 * cites the reference lines (relative to the reference repo root) whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host; row-major, contiguous, fp32;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous on it;
 *   - no allocation, no synchronisation and no global state inside a call (graph-capturable, thread-safe per
 *     stream); scratch comes from the caller via (ws, ws_bytes) sized by the matching *_workspace_bytes().  Per-thread
 *     settings (the int8 cap, an attached ragraph_filter_profile) are thread-local, their objects caller-owned;
 *   - return 0 on success, a negative RAGRAPH_E* code otherwise; ragraph_last_error() gives the thread's last
 *     message.  There is NO CPU fallback: without a gfx950 device every compute entry returns RAGRAPH_EDEVICE.
 *
 * Numerics contract (what "parity" means; restated on the CPU in oracle/ragraph_oracle.c)
 *   - every dot product (cosine scores, GEMM, SpMM) is ONE fp32 fmaf chain in natural index order starting from
 *     +0 -- exactly what v_mfma_f32_32x32x2_f32 / _16x16x4_f32 compute -- so scores are bit-identical to the oracle
 *     regardless of batch size, tiling, split count or GPU count (one exception: SpMM rows and softmax segments of more
 *     than 4096 entries are summed in blocks of 4096, see "Hub rows" at ragraph_spmm_csr_ws_f32);
 *   - top-k order is canonical: score descending, then index ascending (torch.topk leaves ties unspecified);
 *   - row L2 norms use the fixed reduction tree documented at ragraph_normalize_rows_f32.
 */
#ifndef RAGRAPH_HIP_H
#define RAGRAPH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAGRAPH_ABI_VERSION 1

#define RAGRAPH_OK 0
#define RAGRAPH_EINVAL (-1)    /* bad argument (shape, alignment, null pointer)            */
#define RAGRAPH_EUNSUPPORTED (-2) /* valid request outside what the kernels cover (see each entry) */
#define RAGRAPH_EWORKSPACE (-3) /* ws_bytes smaller than *_workspace_bytes()                */
#define RAGRAPH_EDEVICE (-4)   /* HIP runtime / launch failure, or no gfx950 device        */

/* activation selector of the fused epilogues */
#define RAGRAPH_ACT_NONE 0
#define RAGRAPH_ACT_RELU 1   /* Propagation.py:25  F.relu                          */
#define RAGRAPH_ACT_PRELU 2  /* layers/gcn.py:9,40 nn.PReLU(), one shared alpha    */
#define RAGRAPH_ACT_LEAKY 3  /* TaskDecoder.py:7   nn.LeakyReLU(), slope = alpha   */
#define RAGRAPH_ACT_ELU 4    /* RAGraph_node/downprompt.py:14 nn.ELU(), alpha = 1  */

int ragraph_abi_version(void);
const char* ragraph_last_error(void);
/* 0 if a gfx950 device is usable by this process, RAGRAPH_EDEVICE otherwise (never touches the GPU at load time). */
int ragraph_device_check(void);

/* ------------------------------------------------------------------------------------------------------------
 * a1  F.normalize(x, p=2, dim=-1, eps=1e-12)   -- RAGraph_node/ragraph_utils/SimilarityFunctions.py:8,11
 *     out[r,:] = X[r,:] / max(||X[r,:]||_2, 1e-12).  In-place (out == X) allowed.  Any D >= 1.
 *     Norm tree: "lane" l of 64 takes the float4 chunks c = l, l+64, ... of the row and accumulates the squares of
 *     elements 4c..4c+3 (ascending) with fmaf; the 64 partials are combined by the butterfly p[l] += p[l^32]; ^16; ^8;
 *     ^4; ^2; ^1.  sqrtf and the division are correctly rounded.
 */
int ragraph_normalize_rows_f32(const float* X, int64_t n, int D, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a1+a2  fused cosine scores + top-k  -- SimilarityFunctions.py:6-16 + ToyGraphBase.py:66-67
 *        (torch.topk(normalize(Q) @ normalize(K).T, k, largest=True, sorted=True)); edge slab loop
 *        RAGraph_edge/modules/RAGraph.py:298-311.  The B x N score matrix is never written.
 *   Q   [B,D] raw (un-normalised) queries; normalised inside (a1) into the workspace.
 *   Kn  [N,D] key bank ALREADY row-normalised by ragraph_normalize_rows_f32 (done once per bank version; the
 *       reference re-normalises its stored keys on every call, SimilarityFunctions.py:11).
 *   k   1 <= k <= min(N, RAGRAPH_TOPK_MAX) (larger k: ragraph_topk_select_rows_f32 over score slabs).  B,N >= 1.
 *       Any D >= 1, as the reference (SimilarityFunctions.py:6-16 takes every emb_size): the fused kernels are written
 *       for D in {64,128,256}; every other width takes materialised score slabs (~1 GiB of the workspace: dense kernel +
 *       row top-k, banks beyond 2^22 rows in key chunks merged in canonical order) -- the same fmaf chains, the same
 *       bits.  (A bank narrower than 256 columns is better zero-padded ONCE to the next fused width by its owner, as
 *       ragraph_amd.KeyIndex does: zero columns change no bit of a norm or a score.)  B <= 128 takes the wave-streaming
 *       kernel (groups of 16 queries on 16x16x4 MFMA; HBM-bound up to 16 queries), larger B the MFMA-bound tile
 *       kernel (256 queries per workgroup on 32x32x2); same numerics, so B never changes a bit of the result.
 *       32 < k <= 64 materialises ~1 GiB slabs of scores in the workspace (dense kernel + row top-k), same bits.
 *   idx_base  added to every returned index (this shard's first global row).
 *   out_scores [B,k] fp32 descending; out_idx [B,k] int64 (torch indexing dtype).
 *   Unsupported (returns RAGRAPH_EUNSUPPORTED): k > RAGRAPH_TOPK_MAX, shards of >= 2^31 rows.  NaN scores are never selected
 *   (torch.topk would rank NaN first) -- inputs are finite by contract.
 */
#define RAGRAPH_TOPK_MAX 64
size_t ragraph_topk_cosine_workspace_bytes(int64_t B, int64_t N, int D, int k);
int ragraph_topk_cosine_f32(const float* Q, int64_t B, const float* Kn, int64_t N, int D, int k, int64_t idx_base,
                            float* out_scores, int64_t* out_idx, void* ws, size_t ws_bytes, void* stream);

/* Bank-resident form of a1.  The reference keeps resource_keys on the device between calls (ToyGraphBase.py:40-48
 * set_resources / add_resources); the stored bank here is Kn plus, optionally, a PACKED copy of the same rows
 * (row n = [Kn[n][0], Kn[n][2], ... | Kn[n][1], Kn[n][3], ...]) made once per bank update by ragraph_pack_keys_f32.
 * With Kp != NULL, D = 256 and k <= 14 the tile kernel (B > 128) moves the key stream HBM -> LDS by DMA
 * (global_load_lds_dwordx4) instead of through registers.  Results are bit-identical with and without Kp;
 * ragraph_topk_cosine_f32 is this call with Kp = NULL.  Kp holds exactly the values of Kn (caller's contract).
 */
int ragraph_pack_keys_f32(const float* Kn, int64_t N, int D, float* Kp, void* stream);
int ragraph_topk_cosine_bank_f32(const float* Q, int64_t B, const float* Kn, const float* Kp, int64_t N, int D, int k,
                                 int64_t idx_base, float* out_scores, int64_t* out_idx, void* ws, size_t ws_bytes,
                                 void* stream);

/* a1, large batches: the same result through a bf16 MFMA filter (ragraph_amd/csrc/topk_filter.hip).
 *   Exact by construction: (1) a lower bound of every query's final k-th best score -- the k-th exact score over a sample
 *   of the bank, or, for banks of >= 8192 keys, min over k parts of a prefix of the best approximate score in the part,
 *   minus eps(q) (k distinct keys score at least that); (2) a bf16 MFMA pass (16x the fp32 matrix rate) over the next, larger part of
 *   the bank keeps every key whose approximate score is within eps(q) of that bound, where eps(q) = |dq| + max|dk| +
 *   |dq| max|dk| (+ rounding slack) is computed from the actual bf16 rounding errors dq of the query and dk of the bank
 *   rows (<= 2^-7, typically 0.003): by Cauchy-Schwarz no pair's dot product moves by more; (3) the survivors (~100 per query) are
 *   rescored with the natural-order fp32 fmaf chain and selected in canonical order, which gives the exact top-k of
 *   everything seen so far and a tighter bound for the next level (large batches: [0,N/32), [N/32,N/4), [N/4,N); small
 *   ones: fewer, steeper levels -- ragraph_topk_cosine_filtered_plan).  The result has the same bits as
 *   ragraph_topk_cosine_f32.  D in {64,128,256}, k <= 32.
 *   Kb   bf16 copy of Kn made by ragraph_keys_to_bf16 (ragraph_keys_bf16_rows(N) rows x D, uint16 storage: the bf16 rows
 *        padded to a multiple of 256 keys, a tail row, the int8 rows, their tail row, the int8 copy's granule table, and
 *        256 rows of slack that the last stage of an int8 level may read).  The int8 rows come in two classes of granules
 *        of 32 KiB (128 / 256 / 512 keys at D = 256 / 128 / 64), each class on its own scale with its own measured error:
 *        their tail row (32-bit words) = [0] max |dk|^2 and [1] scale of the NORMAL granules, [2] the bank's largest |k_i|,
 *        [3] / [4] the HEAVY granules' max |dk|^2 / scale, [5] the cut between the classes (largest |k_i| a normal granule
 *        may hold), [6] heavy granules, [7] granules; behind it the granules' largest |k_i| (floats, padded to 16 bytes)
 *        and one class bit per granule (set = heavy).
 *   Kp   optional packed fp32 copy (ragraph_pack_keys_f32) for the fp32 level, or NULL.
 *   overflow  device int, set by the call: number of queries whose candidate list exceeded its capacity
 *        (ragraph_topk_cosine_filtered_cap(k) keys per list; a batch of <= 64 queries keeps up to 8 such lists per query,
 *        one per rescoring workgroup) at some level; only possible on banks with thousands of keys within EPS
 *        of a query's k-th best (near-duplicate banks; all-zero queries, which are answered without a scan, may or may
 *        not be counted).  The call itself recomputes those rows with an
 *        exact fp32 scan of the bank ON THE DEVICE (its last launch; an empty list on ordinary banks), so every row of
 *        the result is exact and nothing is read back: the call never synchronises and is HIP-graph capturable.  The
 *        count is diagnostic (a bank that sends many queries to the scan is better served by the fp32 kernels).
 *   overflow_idx  optional device int64[B]: the first *overflow entries receive the row numbers of those queries
 *        (batches of 2048 queries and more; smaller ones repair inside their rescoring launch and leave it untouched).
 */
int64_t ragraph_keys_bf16_rows(int64_t N);
int ragraph_keys_to_bf16(const float* Kn, int64_t N, int D, uint16_t* Kb, void* stream);
int ragraph_topk_cosine_filtered_cap(int k);
size_t ragraph_topk_cosine_filtered_workspace_bytes(int64_t B, int64_t N, int D, int k);
/* Candidate statistics of the most recent filtered call on a workspace: 32 ints at byte offset
 * ragraph_topk_cosine_filtered_stats_offset(ws_bytes) of the workspace AS PASSED to the call (its last bytes), written by the
 * call's own launches: [0] 0x52414753, [1] levels, [2+l] sum of the candidate counts of every 64th query at level l (l < 3),
 * [5+l] how many queries that sum covers, [8+l] 1 if level l ran on the int8 copy, [11+l] keys of level l, [14] queries of
 * the call, [15] all-zero queries among them (answered without a scan, never counted in *overflow), [16] 1 if the call
 * filtered with a speculative first bound (ragraph_topk_cosine_filtered_set_prior), [17] queries that bound was too high
 * for (answered by the exact scan, counted in *overflow), [18] / [19] the smallest / largest final exact k-th best score of
 * the call's queries as order-preserving ints (bits(x) for x >= 0, bits(x) ^ 0x7FFFFFFF below: csrc/filter_common.h f2ord; INT_MAX /
 * INT_MIN: none recorded), [20] the call's final *overflow (one copy of these words tells the owner of the bank
 * everything), [21..32) reserved.  sum / queries = candidates per query: what a level's rescoring costs.  The owner of a bank reads them back
 * asynchronously (ragraph_amd/kernels_index.py: a bank whose int8 levels pass hundreds of candidates per query without
 * overflowing is slower on int8 than on bf16 -- the overflow count alone would never show it). */
size_t ragraph_topk_cosine_filtered_stats_offset(size_t ws_bytes);
/* The schedule the call above will follow for this shape (host-side arithmetic only, no device work):
 * plan[1] = how the first lower bound of a query's k-th best score is made: 2 = bound pass (banks of >= 8192 keys: a
 * bf16 pass over keys [0, plan[6]) records the best approximate score of each of k parts; the smallest, minus eps, is
 * the bound -- worth the exact k-th best of plan[0] keys), 1 = exact top-k over keys [0, plan[0]) as a score slab (dense
 * kernel + row top-k), 0 = the same by the fp32 tile kernel; plan[2] = number of filter levels L (1..3),
 * plan[3..3+L) = their ends (increasing multiples of 256, the last = N; level l filters [end[l-1], end[l]), the first
 * starts at 0).  Returns L, or a negative error code. */
int ragraph_topk_cosine_filtered_plan(int64_t B, int64_t N, int D, int k, int64_t plan[7]);
/* How many of that schedule's LAST levels run on the int8 copy (v_mfma_i32_16x16x64_i8: twice the bf16 rate, a ~5x wider
 * error bound, so ~3x the candidates -- the late levels of batches of >= 1024 queries at D = 128 / 256). */
int ragraph_topk_cosine_filtered_i8_levels(int64_t B, int64_t N, int D, int k);
/* Cap on the int8 levels of the CALLING THREAD's following filtered calls (thread-local; -1 = the built-in rule, 0 = none);
 * returns the previous value.  For callers that know their bank: the int8 bound is only as tight as the largest entries
 * of the bank's ordinary granules allow (two scales: a few heavy-tailed rows cost their own granules only), so a bank of
 * heavy-tailed rows THROUGHOUT passes many more candidates on int8 than on bf16 -- still exact, but slower.
 * ragraph_amd/kernels_index.py sets it per bank from the copy's measured error row and from calls whose candidate lists
 * overflowed. */
int ragraph_topk_cosine_filtered_max_i8_levels(int n);
/* A SPECULATIVE first bound for the CALLING THREAD's following filtered calls (thread-local; NaN = none, the default);
 * returns the previous value.  A call on a single bank whose schedule starts with a bound pass (plan[1] == 2) then skips
 * that pass: every query starts from theta = theta_prior, each level filters with max(theta_prior, the exact k-th best so
 * far), and a verify launch behind the last level PROVES every answer -- a query is exact iff its k-th best candidate scores
 * at least theta_prior (every key scoring at least that passed its level) -- and sends the others to the exact scan of the
 * call's last launch (counted in *overflow and in statistics word [17]).  The result therefore has the bits of
 * ragraph_topk_cosine_f32 for ANY prior; a prior above a query's true k-th best costs that query an exact scan (~0.1 - 2
 * ms), one far below it costs candidates.  The owner of a bank derives it from the k-th best scores its earlier calls
 * reported (statistics words [18], [19]) and withdraws it when a call reports misses (ragraph_amd/kernels_index.py).
 * The sharded entry honours it too (below): there the PROOF is the caller's, over the merged lists.  The reference has no
 * counterpart (torch.topk over the full score matrix, ToyGraphBase.py:66-67): this only removes work. */
float ragraph_topk_cosine_filtered_set_prior(float theta_prior);
int ragraph_topk_cosine_filtered_f32(const float* Q, int64_t B, const float* Kn, const float* Kp, const uint16_t* Kb,
                                     int64_t N, int D, int k, int64_t idx_base, float* out_scores, int64_t* out_idx,
                                     int* overflow, int64_t* overflow_idx, void* ws, size_t ws_bytes, void* stream);

/* a1 over a ROW-SHARDED bank (one shard per GPU; the reference is single-GPU, so no reference line beyond the retrieval
 * itself -- SimilarityFunctions.py:6-16 + ToyGraphBase.py:66-67).  As ragraph_topk_cosine_filtered_f32 on this shard's
 * rows, with the per-query bound theta[B] (a proven lower bound of the query's final k-th best exact score over ALL
 * shards) exposed between the phases so that the caller can sharpen it across the shards -- what keeps a shard's
 * candidate lists at 1/G of a single GPU's instead of the same length:
 *   exchange(ctx, 0)      after the first bound: theta = this shard's bound, and out_scores[q, 0..k) = k lower bounds
 *                         (descending) of the exact scores of k DISTINCT keys of this shard -- after a bound pass the
 *                         parts' best approximate scores minus eps(q), after an exact level 0 its exact top-k.  The
 *                         caller may replace theta[q] by any valid lower bound of the global k-th best: all_reduce MAX of
 *                         theta, or (stronger) the k-th largest of the union of every shard's best m of those values
 *                         (all_gather + ragraph_theta_sharpen_f32);
 *   exchange(ctx, 1 + l)  after level l (not the last): out_scores holds this shard's running top-k (descending, -inf
 *                         where it has fewer than k candidates) and theta = max(theta, its k-th); same contract.
 * n_shards = G >= 1: the shards pool their first samples through exchange 0, so each scans 1/G of the prefix a single
 * bank would (pass 1 if the callback does not combine the shards' out_scores).
 * The callback runs on the calling host thread between launches; whatever it enqueues must be ordered on `stream` (a
 * torch.distributed collective on the current stream is).  plan_N = the LARGEST shard's row count: the schedule (hence
 * the number of callbacks) is computed from it, so that every rank makes the same calls; N <= plan_N <= N + 1024.
 * The result is this shard's exact top-k among its rows that can still be in the global top-k (fewer than k entries
 * are padded with -inf / INT64_MAX); ragraph_topk_merge_f32 over the shards' lists gives the global result, bit-identical
 * to one GPU.  exchange = NULL: exactly ragraph_topk_cosine_filtered_f32.
 * A SPECULATIVE first bound (ragraph_topk_cosine_filtered_set_prior on the calling thread, the SAME value on every rank --
 * a collective decision of the caller): when ragraph_topk_cosine_filtered_sharded_speculates(...) says so, the call skips its
 * bound pass AND exchange 0 (the phases run 1 .. L - 1), starts every query from theta = theta_prior and filters each level
 * with max(theta_prior, the pooled bound).  A shard's list then holds every key of the shard that scores at least
 * max(theta_prior, the global k-th best): the merged lists are the exact global top-k of every query whose MERGED k-th best
 * reaches theta_prior, which only the owner of the merged row can see -- it verifies, tells the other ranks (one tiny
 * all_reduce) and the call is repeated without the prior when any query missed (ragraph_amd/sharded.py). */
typedef void (*ragraph_exchange_fn)(void* ctx, int phase);
/* The owner's verdict behind the merge (one launch): merged_scores [R, k] = the merged global lists of the R rows this rank
 * finishes.  out5[0] = rows whose k-th best is below `prior` (speculative != 0; all-zero queries need no proof), out5[1] = -(the
 * smallest proven k-th best), out5[2] = the largest, out5[3] = this shard's candidates per query over its levels (from the
 * call's statistics words, or -1), out5[4] = its overflowed lists (*overflow, or 0).  all_reduce MAX over the ranks makes the
 * five numbers the group's; a non-zero out5[0] repeats the call without the prior. */
int ragraph_verify_merged_prior_f32(const float* merged_scores, int64_t R, int k, float prior, int speculative,
                                    const int* stats_words, const int* overflow, float* out5, void* stream);
/* 1 if a sharded call of this shape would skip its bound pass and exchange 0 under a valid speculative prior (the plan for
 * plan_N has a bound pass): the same answer on every rank -- it is computed from the shared plan only. */
int ragraph_topk_cosine_filtered_sharded_speculates(int64_t B, int64_t plan_N, int D, int k, int n_shards);
/* ws of the sharded entry: the schedule is planned for (plan_N, n_shards), whose first sample -- hence level-0 scratch --
 * can differ from the single bank's; at least ragraph_topk_cosine_filtered_workspace_bytes(B, plan_N, D, k). */
size_t ragraph_topk_cosine_filtered_sharded_workspace_bytes(int64_t B, int64_t plan_N, int D, int k, int n_shards);
int ragraph_topk_cosine_filtered_sharded_f32(const float* Q, int64_t B, const float* Kn, const float* Kp, const uint16_t* Kb,
                                             int64_t N, int D, int k, int64_t idx_base, float* out_scores, int64_t* out_idx,
                                             int* overflow, int64_t* overflow_idx, void* ws, size_t ws_bytes, void* stream,
                                             int64_t plan_N, float* theta, ragraph_exchange_fn exchange, void* ctx,
                                             int n_shards);

/* a1 + a2 in ONE launch for small banks  -- SimilarityFunctions.py:6-16 + ToyGraphBase.py:66-67 at the reference's own sizes
 * (BASELINE config 1: 2708 queries x a 10 000 x 128 toy bank, k = 5).  Same result, bit for bit, as
 * ragraph_topk_cosine_f32 / _filtered_f32: a workgroup takes 32 queries through all phases of the filtered path (query
 * normalisation, bf16 bound over a prefix, bf16 filter over the whole bank, exact fp32 rescoring, canonical selection,
 * exact scan of a query whose candidate list overflows); the bank is split over up to 8 workgroups per tile, whose
 * per-query lists the tile's last workgroup merges (a ticket), so nothing crosses tiles and there is no second launch.
 * Q [B,D] raw queries; Kn [N,D] unit rows; Kb = ragraph_keys_to_bf16(Kn).  D in {64,128,256}, k <= 16, N >= 128 k
 * (ragraph_topk_cosine_fused_ok: 1 if the shape is supported).  tickets: ceil(B/32) ints, ZERO before the first call; every
 * call leaves them zero (keep one buffer per stream).  ws: ragraph_topk_cosine_fused_workspace_bytes (the splits' lists).
 * Every tile streams the whole bf16 copy (2 N D bytes) from L2: meant for banks whose copy is a few MB
 * (ragraph_amd/kernels_index.py decides). */
/* a1 + a2 in ONE launch for up to 32 queries against a LARGE bank  -- SimilarityFunctions.py:6-16 + ToyGraphBase.py:66-67 at
 * the reference's smallest batches (RAGraph_graph/RAGraph.py:48-60 retrieves ONE query per forward).  Same result, bit for
 * bit, as ragraph_topk_cosine_f32 / _filtered_f32.  Every workgroup of the launch prepares the queries, takes its share of a
 * bf16 bound pass over a prefix of the bank (part maxima published by atomicMax), waits a BOUNDED time for the other
 * workgroups' shares (no grid barrier: the k-th largest of whatever maxima are published is a valid bound), streams its
 * share of the int8 copy (bf16 when ragraph_topk_cosine_filtered_max_i8_levels(0) is in force for the thread, or D = 64),
 * scores its own candidates exactly (fp32 chains) and appends the exact pairs to per-query lists; the last workgroup to
 * finish (a ticket) selects every query's canonical top-k, answers zero queries and LISTS the queries whose list of exact
 * pairs passed its capacity (near-duplicate banks) and, under a speculative first bound, those the bound was too high for; a
 * second launch behind it (every call; empty as a rule) answers the listed queries by exact scans cut into key slices.
 * ragraph_amd/csrc/topk_small.hip.
 *   Q [B,D] raw queries, 1 <= B <= 32; Kn [N,D] unit rows, N >= 65536; Kb = ragraph_keys_to_bf16(Kn); D in {64,128,256}; k <= 32
 *   (ragraph_topk_cosine_small_ok).  overflow: device int, set to the number of queries answered by the exact scan.
 *   state: ragraph_topk_cosine_small_state_bytes() bytes, ZERO before the first call; every call leaves them zero (one
 *   buffer per stream: calls on a stream are ordered).  ws: ragraph_topk_cosine_small_workspace_bytes (the pair lists).
 *   RAGRAPH_SMALL_WAIT_TICKS: the wait's limit in 10-ns ticks (default 200000 = 2 ms; 0: never wait -- test hook). */
int ragraph_topk_cosine_small_ok(int64_t B, int64_t N, int D, int k);
size_t ragraph_topk_cosine_small_state_bytes(void);
/* keys of the bf16 prefix the call's bound pass reads; *i8 (may be NULL) = 1 when its filter pass streams the int8 copy */
int64_t ragraph_topk_cosine_small_prefix_keys(int64_t B, int64_t N, int D, int* i8);
size_t ragraph_topk_cosine_small_workspace_bytes(int64_t B, int D, int k);
int ragraph_topk_cosine_small_f32(const float* Q, int64_t B, const float* Kn, const uint16_t* Kb, int64_t N, int D, int k,
                                  int64_t idx_base, float* out_scores, int64_t* out_idx, int* overflow, int* state, void* ws,
                                  size_t ws_bytes, void* stream);

int ragraph_topk_cosine_fused_ok(int64_t B, int64_t N, int D, int k);
size_t ragraph_topk_cosine_fused_workspace_bytes(int64_t B, int64_t N, int D, int k);
int ragraph_topk_cosine_fused_f32(const float* Q, int64_t B, const float* Kn, const uint16_t* Kb, int64_t N, int D, int k,
                                  int64_t idx_base, float* out_scores, int64_t* out_idx, int* tickets, void* ws,
                                  size_t ws_bytes, void* stream);

/* The per-level exchange of the sharded call: theta[b] = max(theta[b], k-th largest of the G*m scores gathered for query
 * b), gathered = the all_gather of every shard's best m exact scores, [G, B, m] as the collective leaves it; k <= G*m <= 64.
 * (The k-th largest of a subset of all scores bounds the k-th largest of all from below.) */
int ragraph_theta_sharpen_f32(const float* gathered, int G, int64_t B, int m, int k, float* theta, void* stream);

/* Measurement (bench.py's roofline; not part of the reference's interface): a CALLER-OWNED profile object.  While one is
 * attached to the calling thread (thread-local, like the int8 cap: no process-global state), every filtered call of that
 * thread records HIP events around its filter launches on the caller's stream into it -- such a call is not
 * graph-capturable.  create / destroy allocate and free the object and its events outside any compute call.
 *   ragraph_filter_profile_attach(p)   attach p (NULL: detach) to this thread; returns what was attached before
 *   ragraph_filter_profile_last_ms(p)  waits for the most recent recorded call's filter launches; their summed ms (< 0: none)
 *   ragraph_filter_profile_levels(p)   the same per launch: slots 0..2 = the filter levels, slot 3 = the bound pass; HOST
 *                                      arrays of four: ms_host (negative: no such launch), i8_host (1: the level ran on the
 *                                      int8 copy), keys_host (keys the launch covered). */
typedef struct ragraph_filter_profile ragraph_filter_profile;
ragraph_filter_profile* ragraph_filter_profile_create(void);
void ragraph_filter_profile_destroy(ragraph_filter_profile* p);
ragraph_filter_profile* ragraph_filter_profile_attach(ragraph_filter_profile* p);
float ragraph_filter_profile_last_ms(ragraph_filter_profile* p);
int ragraph_filter_profile_levels(ragraph_filter_profile* p, float* ms_host, int* i8_host, int64_t* keys_host);

/* Cross-shard / cross-split merge of sorted top-k lists (no counterpart in the reference: it is single-GPU).
 *   scores,idx [G,B,k] (list g of query b at ((g*B)+b)*k) -> out [B,k], canonical order; result independent of G.
 *   G*k <= 4096.
 */
int ragraph_topk_merge_f32(const float* scores, const int64_t* idx, int G, int64_t B, int k, float* out_scores,
                           int64_t* out_idx, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a3  banks of duplicates  -- the reference's own bank recipe makes most rows bit-identical:
 *     RAGraph_node/ragraph_utils/ToyGraphBase.py:91-119 appends 1 + num_augment_scale passes per resource graph, :98 draws
 *     a pass's rows WITH replacement, and Augmentation.py:9-20 multiplies the augmented passes' features by
 *     bernoulli(sample_prob * 0.01) (zero for nearly every node), so those passes store normalize(PReLU(bias)) over and
 *     over.  The search runs over one representative per group of identical rows; the winners are expanded afterwards.
 *
 * ragraph_dedup_rows_f32: groups the BIT-identical rows of Kn [N,D] (N < 2^31).
 *   stats      [2] int64: number of groups U, size of the largest group;
 *   uniq_row   [N] int64: entries 0..U-1 = the lowest row of every group, ascending (Kn[uniq_row[:U]] = the unique bank);
 *   group_ptr  [N+1] int32: group u's rows are members[group_ptr[u] .. group_ptr[u+1]) (entries beyond U equal N);
 *   members    [N] int32: every group's rows in ascending order.
 *   A 64-bit row hash orders the rows (stable radix sort); rows are merged only after a bit-by-bit comparison, so the
 *   grouping never depends on the hash being collision-free (a collision can split a group in two; the expansion below
 *   is correct for any partition into groups of identical rows).  Asynchronous; the caller reads `stats` when it needs U.
 *
 * ragraph_topk_expand_groups_f32: the canonical top-k of the bank from the canonical top-ku (ku = min(k, U)) of the unique
 *   rows: scores_u / idx_u [B,ku] (idx_u - idx_base_u = group number; entries outside [0,U) -- the -inf / INT64_MAX padding
 *   of a shard's list -- are empty groups) -> out_scores / out_idx [B,k] (bank rows + idx_base), score descending then row
 *   ascending; groups whose scores tie are merged by row; fewer than k rows in all: padded with -inf / INT64_MAX.
 *   Identical rows have identical scores (one fmaf chain over the same bits), so the result has the bits of the search
 *   over all N rows.  ku <= 64, k <= RAGRAPH_TOPK_MAX.
 */
size_t ragraph_dedup_rows_workspace_bytes(int64_t N);
int ragraph_dedup_rows_f32(const float* Kn, int64_t N, int D, int64_t* stats, int64_t* uniq_row, int32_t* group_ptr,
                           int32_t* members, void* ws, size_t ws_bytes, void* stream);
int ragraph_topk_expand_groups_f32(const float* scores_u, const int64_t* idx_u, int ku, int64_t idx_base_u,
                                   const int32_t* group_ptr, const int32_t* members, int64_t U, int64_t B, int k,
                                   int64_t idx_base, float* out_scores, int64_t* out_idx, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a2  value/label gather  -- ToyGraphBase.py:70-71  resource_values[topk_indices], resource_labels[topk_indices]
 *     out[m,:] = V[idx[m] - idx_base, :] for m in [0,M); rows with idx outside [idx_base, idx_base+N) are written
 *     as zeros (lets each shard gather only the winners it owns; a sum over shards completes it).  Any D >= 1.
 */
int ragraph_gather_rows_f32(const float* V, int64_t N, int D, const int64_t* idx, int64_t M, int64_t idx_base,
                            float* out, void* stream);

/* a8  sum_k V[idx] and mean_k L[idx]  -- RAGraph_node/RAGraph.py:48-49 (torch.sum / torch.mean over dim=1),
 *     edge: RAGraph_edge/modules/RAGraph.py:314,321 (mean of values: pass scale = 1/k through `v_scale`).
 *     sum_V[b,:] = v_scale * sum_{j<k} V[idx[b,j]]   (sequential j, fp32 adds);  mean_L[b,:] = (sum_j L[idx[b,j]]) / k.
 *     L / mean_L may be NULL (edge flavour has no labels).  Out-of-shard indices contribute zero.
 */
int ragraph_gather_reduce_f32(const float* V, int D, const float* L, int C, int64_t N, const int64_t* idx, int64_t B,
                              int k, int64_t idx_base, float v_scale, float* sum_V, float* mean_L, void* stream);

/* a8  the reduction and the prompt fusion in ONE launch  -- RAGraph_node/RAGraph.py:48-49 + :53:
 *       out[b,:] = A[b,:] * wa + (v_scale * sum_{j<k} V[idx[b,j]]) * wb;   mean_L as above.
 *     The bits of ragraph_gather_reduce_f32 followed by ragraph_axpby_f32(A, wa, sum_V, wb) (two multiplies and an add, not
 *     contracted), without sum_V in memory.  A, out [B,D], out must not alias A. */
int ragraph_gather_reduce_mix_f32(const float* V, int D, const float* L, int C, int64_t N, const int64_t* idx, int64_t B,
                                  int k, int64_t idx_base, float v_scale, const float* A, float wa, float wb, float* out,
                                  float* mean_L, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a4/a9/a1  dense  Y = act(X @ W^T + b)   -- layers/gcn.py:32 (self.fc, no bias), TaskDecoder.py:15-16
 *           (fc1+LeakyReLU, fc2), and the materialised score matrix of SimilarityFunctions.py:14 (X=Qn, W=Kn).
 *     X [M,K], W [N,K] (nn.Linear weight layout), bias [N] or NULL, Y [M,N].  Any M,K >= 1; 1 <= N <= 65535 * 64
 *     (= 4 194 240 output columns per launch, RAGRAPH_EUNSUPPORTED beyond: callers that materialise scores against a
 *     longer bank -- the k > 64 slab path -- cut the bank into column slabs).
 *     Each Y[m,n] is one fmaf chain over k = 0..K-1 from +0; then + bias (one fp32 add); then act.
 */
int ragraph_linear_f32(const float* X, int64_t M, int K, const float* W, int64_t N, const float* bias, int act,
                       float alpha, float* Y, void* stream);

/* f4 (fine-tuning backward)  C = A^T B for tall row-major operands  -- the weight gradient of a dense layer, gW = gY^T X
 *     (torch autograd of nn.Linear in RAGraph_node/finetune-rag.py:81-84, TaskDecoder.py:15-16; the edge flavour's gating
 *     weight, modules/RAGraph.py:168).  A [n,M], B [n,N], C [M,N]; any n, M, N >= 1.  The rows are cut into ranges whose
 *     partial products (fmaf chains over the range's rows in order, from +0) are added in range order: deterministic,
 *     and within fp32 rounding of torch's (tests: 1e-4 relative).  ws: ragraph_linear_tn_workspace_bytes(n, M, N). */
size_t ragraph_linear_tn_workspace_bytes(int64_t n, int M, int N);
int ragraph_linear_tn_f32(const float* A, const float* B, int64_t n, int M, int N, float* C, void* ws, size_t ws_bytes,
                          void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a4/a7/a11  CSR SpMM with fused epilogue
 *     Y[r,:] = act( sum_{e in row r} val[e] * X[col[e],:]  + bias ) + beta * Y_in[r,:]
 *   replaces torch.mm(adj, seq_fts) + bias + PReLU (layers/gcn.py:36-40), relu(adj_normalized @ x)
 *   (Propagation.py:22-25) and the gather-scale-scatter_add of RAGraph_edge/modules/RAGraph.py:232-240 (edges sorted
 *   by destination once; no atomics, so the sum order is the CSR order and the result is deterministic).
 *   rowptr [n+1] int64, col [nnz] int32, val [nnz] fp32, X [n_cols,D], Y [n,D]; D % 4 == 0; X,Y 16-byte aligned.
 *   Row sum = one fmaf chain in CSR order from +0 (rows of more than 4096 edges: see "Hub rows" below).  bias [D] or NULL; Y_in [n,D] or NULL (beta ignored then).
 *   Y must not alias X.
 */
int ragraph_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n, const float* X, int D,
                         const float* bias, int act, float alpha, float beta, const float* Y_in, float* Y,
                         void* stream);
/* Hub rows.  A row (segment) longer than 4096 entries is summed in blocks of 4096 consecutive entries: every block is its
 * own chain from +0 and the block sums are added in block order (deterministic; the oracle's order; it differs from one
 * long chain -- and from the reference's unordered scatter_add_ -- by fp32 rounding only).  The plain entry points walk
 * such a row's blocks with the row's own lanes; the _ws variants take a workspace of
 * ragraph_sparse_workspace_bytes(nnz, D) bytes (D = 0 for the softmax) and spread the blocks over the whole chip -- the
 * same bits, and 31 ms -> 2 ms per layer on a power-law graph whose most popular item holds 3 M of 44 M edges. */
size_t ragraph_sparse_workspace_bytes(int64_t nnz, int D);
int ragraph_spmm_csr_ws_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n, const float* X, int D,
                            const float* bias, int act, float alpha, float beta, const float* Y_in, float* Y, int64_t nnz,
                            void* ws, size_t ws_bytes, void* stream);

/* Integer bookkeeping primitives of ingestion (edge list -> CSR) and of the duplicate grouping: a stable LSD radix sort of
 * 64-bit keys by their low `bits` bits (8 bits per pass) with optional 4- or 8-byte values, and int32 prefix sums (totals
 * below 2^31).  Own kernels (csrc/sortscan.hip); exported for the tests.  Inputs are left as they are. */
size_t ragraph_radix_sort_workspace_bytes(int64_t n, int val_bytes);
int ragraph_radix_sort_u64(const uint64_t* keys_in, uint64_t* keys_out, const void* vals_in, void* vals_out, int val_bytes,
                           int64_t n, int bits, void* ws, size_t ws_bytes, void* stream);
size_t ragraph_scan_workspace_bytes(int64_t n);
int ragraph_scan_sum_i32(const int* in, int* out, int64_t n, int inclusive, void* ws, size_t ws_bytes, void* stream);

/* COO -> CSR, STABLE in the given edge order (sort_cols = 0: inside a row the edges keep their input order -- the order
 * scatter_add_ accumulates in, RAGraph_edge/modules/utils.py:17-32; sort_cols = 1: columns ascending inside a row).  The edge
 * flavour re-draws its edge set every training step (modules/RAGraph.py:337-343, modules/utils.py:40-53), a training SpMM
 * needs the transposed pattern and a row gather's backward its hit lists: all of them come through here (own radix sort; no
 * other library on any per-step path).  rowptr [n + 1]; perm [E]: the input position of CSR slot s; out_col [E] (optional,
 * int32): the columns in CSR order.  ws: ragraph_coo_to_csr_workspace_bytes(E, n).  E < 2^31, n < 2^31. */
size_t ragraph_coo_to_csr_workspace_bytes(int64_t E, int64_t n);
int ragraph_coo_to_csr_i64(const int64_t* row, const int64_t* col, int64_t E, int64_t n, int sort_cols, int64_t* rowptr,
                           int64_t* perm, int32_t* out_col, void* ws, size_t ws_bytes, void* stream);
/* rows[e] = the row of CSR slot e (the inverse of rowptr: torch.repeat_interleave(arange(n), counts) without its prefix sum). */
int ragraph_csr_row_ids_i64(const int64_t* rowptr, int64_t n, int64_t nnz, int64_t* rows, void* stream);

/* pos[j] = the position of the j-th non-zero byte of mask (ascending), *count = their number: the boolean indexing of the edge
 * flavour's per-step edge dropout (RAGraph_edge/modules/utils.py:40-53, edges[mask]) on the library's own prefix sums. */
size_t ragraph_mask_positions_workspace_bytes(int64_t E);
int ragraph_mask_positions_i64(const unsigned char* mask, int64_t E, int64_t* pos, int64_t* count, void* ws, size_t ws_bytes,
                               void* stream);

/* a7, several hops  -- Propagation.py:19-25 with the features PANEL-major between the hops: [D / 32][n][32] floats (a row's
 * 32-column blocks, one 128-byte line each).  The column-panel hop gives every XCD one panel (D = 256): the rows it gathers
 * from are n x 128 bytes instead of n x 1 KiB, a third of which fit its L2 on a graph without locality (c2: L2 hit rate 15 ->
 * 34 %, a hop 132 -> ~116 us).  x_panels / y_panels: 0 = row-major [n, D], 1 = panel-major.  Same chains as ragraph_spmm_csr_f32
 * (per element, CSR order, blocks of 4096 edges), hence the same bits in every layout; no bias / residual.
 * n = output rows (rowptr may be a slice of a larger graph's: a rank's rows), x_rows = rows of X (the panel stride of a
 * panel-major X).  D in {64, 128} or a multiple of 256. */
int ragraph_spmm_csr_panels_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n, const float* X,
                                int64_t x_rows, int x_panels, int D, int act, float alpha, float* Y, int y_panels, void* stream);

/* a4 in ONE launch, inference association  -- layers/gcn.py:36-40 evaluated as (A_hat X) W^T (a layer at most half as wide in as
 * out; same sum, another order: DESIGN.md section 2).  The dense layer's stream kernel makes its stages itself: the lanes that
 * would copy a row of the aggregated table into LDS walk that row's edges and leave the sum there, under the other stage's
 * MFMAs -- the aggregated [M, K] table never exists in HBM.  rowptr [M + 1] (a slice of a graph's row pointers gives those
 * rows), col / val the graph's, X [*, K] the table gathered from, W [N, K] (nn.Linear layout), Y [M, N] = act((A X) W^T + bias).
 * K in {64, 128}.  The bits of ragraph_spmm_csr_f32 followed by ragraph_linear_f32 (hub rows summed in blocks of 4096). */
int ragraph_spmm_linear_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t M, const float* X, int K,
                            const float* W, int64_t N, const float* bias, int act, float alpha, float* Y, void* stream);

/* a7 over a TILED graph (round 6; opt-in)  -- Propagation.py:19-25, layers/gcn.py:36.  The hop of ragraph_spmm_csr_panels_f32
 * with the GRAPH cut so that what an XCD gathers from at any moment is a few MB: source rows in blocks, destination rows in C
 * chunks of 128 RG rows whose 32-column sums stay in one workgroup's LDS, edges stored once more in (chunk, 8-lane group, source
 * block, row, column) order as wave-batches of 64 slots (plan made once per graph: ragraph_amd/graph.py CSRGraph.tile_plan):
 *   wp [16 C + 1] first wave-batch of (chunk, wave); col3 / val3 / row3 [(wave-batches + 1) 64]: slot 8 g + i of a wave-batch =
 *   edge i of group g's batch -- source row, value, destination row inside the chunk (0xFFFF: no edge).  RG <= 9.
 * A row's chain is unchanged -- ascending columns, one fmaf sequence from +0 -- so the result has the bits of the other SpMM
 * entries; required: columns ascending inside every row, no row longer than 4096 edges (else use those entries). */
int ragraph_spmm_csr_tiled_f32(const int* wp, const int* col3, const float* val3, const unsigned short* row3, int RG, int C,
                               int64_t n, const float* X, int64_t x_rows, int x_panels, int D, int act, float alpha, float* Y,
                               int y_panels, void* stream);

/* a7  adj / adj.sum(dim=1, keepdim=True)  -- Propagation.py:15-16.  val_out[e] = val[e] / rowsum(row(e)), rowsum =
 *     sequential fp32 adds in CSR order.  In-place allowed.  (A zero row sum gives inf/nan exactly as the reference.) */
int ragraph_csr_row_normalize_f32(const int64_t* rowptr, const float* val, int64_t n, float* val_out, void* stream);

/* a12  torch_scatter.scatter_softmax(x, dst)  -- RAGraph_edge/modules/RAGraph.py:261 (torch_scatter 2.1.2: per
 *      segment max, exp(x - max), divide by the segment sum).  Segments = CSR rows (edges sorted by destination). */
int ragraph_segment_softmax_f32(const int64_t* rowptr, const float* x, int64_t n, float* out, void* stream);
int ragraph_segment_softmax_ws_f32(const int64_t* rowptr, const float* x, int64_t n, int64_t nnz, float* out, void* ws,
                                   size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a8  fusion arithmetic  -- RAGraph_node/RAGraph.py:53  hidden = query*(1-w) + rag*w  (two multiplies, one add,
 *     NOT contracted to an fma, matching the eager reference).  a,b,out [n] elementwise; in-place allowed. */
int ragraph_axpby_f32(const float* a, float wa, const float* b, float wb, int64_t n, float* out, void* stream);

/* a8  K5 + K6 of an inference forward in ONE launch  -- RAGraph_node/RAGraph.py:53-57 + ragraph_utils/TaskDecoder.py:14-17:
 *       hidden = query * wq + rag * wr;  logits = fc2(LeakyReLU_slope(fc1(hidden)));
 *       out = softmax(logits) * (1 - lambda) + rag_label * lambda        (rag_label NULL: plain softmax)
 *     query, rag [n,D]; W1 [H,D], b1 [H] or NULL; W2 [C,H], b2 [C] or NULL; rag_label, out [n,C].  Same bits as
 *     ragraph_axpby_f32 -> ragraph_linear_f32 (LEAKY) -> ragraph_linear_f32 -> ragraph_softmax_mix_f32 (same chains, same
 *     order); meant for the launch-bound small forwards (Cora, PROTEINS batches) -- one workgroup per 4 rows, scalar
 *     fp32 -- while large batches are faster through the separate entries (MFMA tile kernel for fc1).
 *     4 (D + H + C) floats must fit 64 KB of LDS. */
int ragraph_fuse_decode_f32(const float* query, const float* rag, int64_t n, int D, float wq, float wr, const float* W1,
                            const float* b1, int H, float slope, const float* W2, const float* b2, int C,
                            const float* rag_label, float lambda, float* out, void* stream);

/* a8  torch.softmax(decode_label, dim=1) * (1-lambda) + rag_label * lambda  -- RAGraph_node/RAGraph.py:55-57.
 *     logits [B,C], rag_label [B,C] or NULL (then plain softmax; log_mode=1 gives log_softmax, downprompt.py:54).
 *     C <= 1024. */
int ragraph_softmax_mix_f32(const float* logits, const float* rag_label, int64_t B, int C, float lambda, int log_mode,
                            float* out, void* stream);

/* a8 (graph flavour)  torch.mean(x, dim=0)  -- RAGraph_graph/RAGraph.py:50,63; a10 per-graph sum readout
 *     split_and_batchify_graph_feats  -- RAGraph_graph/downprompt.py:98-112, with downstreamprompt's w * h
 *     (downprompt.py:154-168) fused into the pass.
 *     out[g,:] = scale_g * sum_{r in [seg_ptr[g], seg_ptr[g+1])} (w ? w[:] * X[r,:] : X[r,:]),  sequential r, fp32 adds.
 *     mean_mode=1: scale_g = 1/len_g (a division), else 1.  w [D] or NULL.  Any D >= 1 (float4 lanes when D % 4 == 0 and
 *     X, out, w are 16-byte aligned, scalar lanes otherwise: same additions in the same order). */
int ragraph_segment_reduce_f32(const float* X, int D, const int64_t* seg_ptr, int64_t G, const float* w, int mean_mode,
                               float* out, void* stream);

/* a10  cosine-to-prototype logits  -- RAGraph_graph/downprompt.py:41-56 (predict: cosine_similarity to each class
 *      mean, eps=1e-8, then log_softmax) and RAGraph_node/downprompt.py:41-46 (softmax).
 *      emb [G,D], proto [C,D] -> out [G,C];  mode 0 = raw cosine, 1 = softmax, 2 = log_softmax.  C <= 64. */
int ragraph_proto_cosine_f32(const float* emb, int64_t G, int D, const float* proto, int C, int mode, float* out,
                             void* stream);

/* a13  emb_gate: out = x * sigmoid(z)  -- RAGraph_edge/modules/RAGraph.py:168 (z = x @ gating_weight + gating_bias
 *      comes from ragraph_linear_f32).  Elementwise over n values; in-place allowed. */
int ragraph_sigmoid_gate_f32(const float* x, const float* z, int64_t n, float* out, void* stream);

/* a12  edge_times.float(), then (t - t_min) / (t_max - t_min)  -- RAGraph_edge/modules/RAGraph.py:254-257.
 *      t [n] int64 time steps; t_min / t_max are the caller's reductions (max_step overrides t_max, :255-256). */
int ragraph_time_rescale_f32(const int64_t* t, int64_t n, float t_min, float t_max, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * torch.topk(scores, k) over a MATERIALISED score matrix (canonical order, agrees with the fused kernel on equal
 * scores) -- few-shot retrieve after mixing structure and semantic similarities
 * (RAGraph_node_fewshot/ragraph_utils/ToyGraphBase.py:58-64) and the evaluation's top-20 of user x item ratings
 * (RAGraph_edge/utils/metrics.py:116).  S [B, ld] row-major with N <= ld valid columns; k <= min(N, 64).
 * out_scores [B,k], out_idx [B,k] int64. */
int ragraph_topk_rows_f32(const float* S, int64_t B, int64_t N, int64_t ld, int k, float* out_scores,
                          int64_t* out_idx, void* stream);

/* Large k: the canonical top-k SET of each row of a materialised score matrix -- the edge flavour's vanilla phase retrieves
 * with retrieve_num = 50 ... 100000 and consumes only the MEAN of the winners' values
 * (RAGraph_edge/modules/RAGraph.py:57,73 retrieve_num; :308-321 topk -> resource_values[idx] -> .mean(dim=1)).
 * S [B, ld] with N <= ld valid columns; any 1 <= k <= N.  out_kth [B] = the k-th largest score; out_idx [B,k] int64 = the
 * indices of the k winners under the canonical order (score descending, index ascending: of the scores equal to the
 * k-th, the lowest indices), written in ASCENDING index order (the winners' mean is then summed in a fixed order).
 * Exact radix select + ordered compaction, one workgroup per row, deterministic. */
int ragraph_topk_select_rows_f32(const float* S, int64_t B, int64_t N, int64_t ld, int64_t k, float* out_kth,
                                 int64_t* out_idx, void* stream);
/* The same with a workspace (ragraph_topk_select_rows_workspace_bytes(B, N)): rows of >= 65536 scores are cut into chunks
 * of 16384 and every pass of the selection is a launch over (chunk, row) -- integer histograms, a prefix over the chunks,
 * ordered compaction per chunk -- instead of one workgroup walking a 4 M-score row five times alone (64 x 4 M: 41 -> ~2 ms).
 * Same result, bit for bit.  B <= 65535 rows per call.  ws = NULL or short rows: ragraph_topk_select_rows_f32. */
size_t ragraph_topk_select_rows_workspace_bytes(int64_t B, int64_t N);
int ragraph_topk_select_rows_ws_f32(const float* S, int64_t B, int64_t N, int64_t ld, int64_t k, float* out_kth,
                                    int64_t* out_idx, void* ws, size_t ws_bytes, void* stream);

/* batch_pred[i, pos_list] = value  -- RAGraph_edge/utils/metrics.py:210-214 (_mask_history_pos, value = -1e8).
 * CSR (rowptr [B+1], col [nnz], both int64) lists the columns to overwrite in each row of S [B, ld]. */
int ragraph_scatter_fill_f32(float* S, int64_t B, int64_t N, int64_t ld, const int64_t* rowptr, const int64_t* col,
                             float value, void* stream);

/* All-pairs shortest paths -- ragraph_utils/PositionAwareEncoder.py:27-48: dist = adj with 0 -> inf, diagonal 0, then
 * n min-plus steps.  adj, dist dense [n,n] fp32 (dist may not alias adj).  n launches of an O(n^2) step: meant for the
 * query graphs of the few-shot flavour (n ~ 30-600), not for the 100k-node graph. */
int ragraph_floyd_warshall_f32(const float* adj, int n, float* dist, void* stream);

/* PositionAwareEncoder.py:6-24: out[u,a] = 1/(dist[u, anchors[a]] + 1) if that distance < dis_q else 0.
 * anchors [A] int64 (drawn by the caller: the reference uses torch.randint), out [n,A]. */
int ragraph_position_code_f32(const float* dist, int n, const int64_t* anchors, int A, float dis_q, float* out,
                              void* stream);

/* The same codes WITHOUT the all-pairs matrix: distances to the A anchors only, on the CSR of the (block-diagonal) query
 * batch -- what RAGraph_node_fewshot/ragraph_utils/ToyGraphBase.py:49-50 needs on every forward from
 * PositionAwareEncoder.py:6-24.  One workgroup per anchor relaxes d[u] = min(d[u], val[u,v] + d[v]) over the rows to its
 * fixpoint (the minimum over all walks u -> anchor of w1 + (w2 + (...)) in fp32; Floyd-Warshall associates the same sums
 * differently: equal to ~1 ulp).  val == 0 entries are no edges (the reference's `dist[adj == 0] = inf`), diagonal 0.
 * codes [n,A]; dist [n,A] or NULL receives the distances (inf = unreachable).  n <= 40000 (one anchor's vector in LDS). */
int ragraph_position_codes_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n,
                                   const int64_t* anchors, int A, float dis_q, float* codes, float* dist, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fine-tuning backward (section 7.3 hard part 4).  The matrix parts of a backward pass are the
 * forward entry points again (ragraph_linear_f32 on transposed operands; ragraph_spmm_csr_f32 on the TRANSPOSED CSR: the
 * few-shot flavour trains its decode layer through the SpMM, RAGraph_node_fewshot/RAGraph.py:69; the edge flavour trains
 * embeddings and gate through three propagation layers, RAGraph_edge/modules/RAGraph.py:280-283,335-355); these are the
 * element-wise derivatives of the fused epilogues.
 *   act_grad:  gz = gy * act'(z) written through the OUTPUT y (sign(y) = sign(z) for ReLU / PReLU / LeakyReLU); for PReLU
 *     alpha_terms (optional) receives gy * z on z < 0 (z = y / alpha): its sum is the slope's gradient (layers/gcn.py:9).
 *   sigmoid_gate_grad:  out = x * sigmoid(z) (modules/RAGraph.py:168): gx = g * s, gz = g * x * s * (1 - s).
 *   softmax_grad:  RAGraph_node/RAGraph.py:55-57: out = p * (g - sum_c g_c p_c), g = go * scale (scale = 1 - label_weight).
 *   mul_cols:  out[r,:] = x[r,:] * w  -- downstreamprompt.forward (RAGraph_graph/downprompt.py:164-168); its own backward.
 *   mul_cols_act:  out[r,:] = act(x[r,:] * w), act one of RAGRAPH_ACT_* -- the node flavour's downstreamprompt.forward,
 *     ELU(weight * h) (RAGraph_node/downprompt.py:118-130; alpha = 1); backward = act_grad through the output + mul_cols.
 *   mul:  out = a * b elementwise -- the prompt weight's gradient is the column sum of g * x (mul + segment_reduce). */
int ragraph_act_grad_f32(const float* y, const float* gy, int64_t n, int act, float alpha, float* gz, float* alpha_terms,
                         void* stream);
int ragraph_sigmoid_gate_grad_f32(const float* x, const float* z, const float* g, int64_t n, float* gx, float* gz, void* stream);
int ragraph_softmax_grad_f32(const float* p, const float* go, int64_t B, int C, float scale, float* out, void* stream);
int ragraph_mul_cols_f32(const float* x, const float* w, int64_t n, int D, float* out, void* stream);
int ragraph_mul_cols_act_f32(const float* x, const float* w, int64_t n, int D, int act, float alpha, float* out, void* stream);
int ragraph_mul_f32(const float* a, const float* b, int64_t n, float* out, void* stream);
/*   proto_cosine_grad:  gradient of ragraph_proto_cosine_f32 with respect to the embeddings (the prototypes are constants
 *     of a forward: RAGraph_node/downprompt.py:24,41-46, RAGraph_graph/downprompt.py:41-56).  out = the forward's output
 *     in the same mode, gout [G,C] -> gemb [G,D]. */
int ragraph_proto_cosine_grad_f32(const float* emb, int64_t G, int D, const float* proto, int C, int mode, const float* out,
                                  const float* gout, float* gemb, void* stream);
/*   proto_cosine_grad_proto:  the same gradient with respect to the PROTOTYPES -- a training step of the node flavour rebuilds
 *     them from the embeddings of that step and keeps them in the autograd graph (RAGraph_node/downprompt.py:24-25,59-78):
 *     gproto [C,D].  Sums in a fixed order (blocks of 256 embeddings, then the blocks): run-to-run identical.  C * D <= 8192.
 *     workspace: ragraph_proto_cosine_grad_proto_workspace_bytes(G, C, D). */
size_t ragraph_proto_cosine_grad_proto_workspace_bytes(int64_t G, int C, int D);
int ragraph_proto_cosine_grad_proto_f32(const float* emb, int64_t G, int D, const float* proto, int C, int mode, const float* out,
                                        const float* gout, float* gproto, void* workspace, size_t workspace_bytes, void* stream);
/*   axpby_dev:  out = a * w[ia] + b * w[ib] with the weights read from DEVICE memory (an index < 0: weight 0) -- the
 *     trainable [1, 2] mixing weight of weighted_feature (RAGraph_node/downprompt.py:100-114) without a host read-back;
 *     uncontracted like ragraph_axpby_f32. */
int ragraph_axpby_dev_f32(const float* a, const float* b, const float* w, int ia, int ib, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Toy-bank construction (the step before the hot path), batched over resource graphs.
 *
 * InverseSampling.pagerank_algorithm  -- RAGraph_node/ragraph_utils/InverseSampling.py:22-47 (dense);
 *   RAGraph_edge/modules/ragraph_utils/InverseSampling.py:22-60 (sparse): p = 1/N; repeat new_p = (1-d)/N + d (P^T p +
 *   dangling mass / N) until ||new_p - p||_1 < eps, returning the iterate BEFORE the converged step (break precedes the
 *   assignment, :41-43).  The batch is one block-diagonal matrix: graph g owns nodes [graph_ptr[g], graph_ptr[g+1]),
 *   graph_of[i] = g; a single big graph is G = 1.  (rowptrT, colT, valT) = CSR of the TRANSPOSED adjacency (row j lists
 *   the i with adj[i][j] != 0, valT = adj[i][j]); out_deg[i] = sum_j adj[i][j] (ragraph_csr_row_sums_f32 of the
 *   adjacency).  max_iter power iterations are enqueued back to back (converged graphs skip theirs on the device; d = 0.85
 *   and eps = 1e-6 converge within ~90); iters[g] = iterations graph g took, max_iter if it did not converge.
 *   No host synchronisation.  ws: ragraph_pagerank_workspace_bytes(n, G). */
size_t ragraph_pagerank_workspace_bytes(int64_t n, int64_t G);
int ragraph_pagerank_f32(const int64_t* rowptrT, const int32_t* colT, const float* valT, const float* out_deg,
                         const int64_t* graph_ptr, const int32_t* graph_of, int64_t G, int64_t n, float d, float eps,
                         int max_iter, float* p, int* iters, void* ws, size_t ws_bytes, void* stream);
/* InverseSampling.compute_sample_prob (:6-19) with degree_centrality_algorithm (:50-56): importance = alpha * pagerank +
 * (1 - alpha) * col_sum / (N_g - 1); prob = (1 / (importance + eps)) / (sum over graph g).  col_sum = column sums of the
 * adjacency (ragraph_csr_row_sums_f32 of its transposed CSR). */
int ragraph_sample_prob_f32(const float* pagerank, const float* col_sum, const int64_t* graph_ptr, int64_t G, float alpha,
                            float eps, float* prob, void* stream);
/* out[r] = sum of row r of a CSR matrix, sequential fp32 adds in CSR order (torch.sum(adj, dim=1) / dim=0 on the
 * transposed CSR: InverseSampling.py:25,53). */
int ragraph_csr_row_sums_f32(const int64_t* rowptr, const float* val, int64_t n, float* out, void* stream);
/* PositionAwareEncoder.encode_position_aware_code (PositionAwareEncoder.py:6-24) for G graphs of n <= 64 nodes at once
 * (the sampled toy graphs, ToyGraphBase.py:114: n = num_inverse_sample = 10): adj [G,n,n] dense, anchors [G,A] int64
 * (drawn by the caller: the reference uses torch.randint), codes [G,n,A]; dist_out [G,n,n] optional (the Floyd-Warshall
 * closure of :27-48). */
int ragraph_position_codes_batch_f32(const float* adj, int64_t G, int n, const int64_t* anchors, int A, float dis_q,
                                     float* dist_out, float* codes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Graph ingestion (the step before the path on the data side).
 *
 * utility.py:19-26,45-66 (normalize_adj of the batch's block-diagonal adjacency + I, built densely through scipy):
 *   D^-1/2 (A + I) D^-1/2 directly as CSR from an edge list.  row/col [E] int64 (batch-global node ids; duplicate edges
 *   SUM, as sp.coo_matrix(...).todense() does); n nodes.  Outputs: rowptr [n+1] int64, out_col / out_val with capacity
 *   E + n, *nnz (device) = number of stored entries.  normalize_adj returns (A D)^T D, i.e. entry (i,j) = d_i A[j][i] d_j
 *   with d from A's ROW sums: the transposed pattern is emitted, so a directed edge list gives the reference's matrix
 *   too.  Values are computed in float64 and cast to fp32 last (scipy float64 -> torch.FloatTensor).  Columns ascend
 *   within a row.  ws: ragraph_ingest_workspace_bytes(E + n, n). */
size_t ragraph_ingest_workspace_bytes(int64_t max_keys, int64_t n);
int ragraph_csr_sym_normalized_f32(const int64_t* row, const int64_t* col, int64_t E, int64_t n, int64_t* rowptr,
                                   int32_t* out_col, float* out_val, int64_t* nnz, void* ws, size_t ws_bytes, void* stream);
/* base_model.py:34-52 (_make_binorm_adj) + dataloader.py:94,108-113: interactions (users[e], items[e], step[e]) ->
 *   the symmetric, binarised, bi-normalised bipartite adjacency as a COO edge list over the joint id space (items offset
 *   by num_users), both directions, ordered by (destination, source) -- the order (A D)^T D .tocoo() leaves --, with
 *   norm = d^-1/2[src] d^-1/2[dst] (float64 product cast to fp32) and, per edge, the time step of the LAST occurrence of
 *   its (user, item) pair.  edges [2E,2] / norm [2E] / times [2E] have capacity 2E; *nedges (device) = edges written.
 *   ws: ragraph_ingest_workspace_bytes(2E, num_users + num_items). */
int ragraph_binorm_edges_f32(const int64_t* users, const int64_t* items, const int64_t* step, int64_t E, int64_t num_users,
                             int64_t num_items, int64_t* edges, float* norm, int64_t* times, int64_t* nedges, void* ws,
                             size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAGRAPH_HIP_H */
