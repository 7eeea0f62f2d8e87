// Host side: the level schedule of a filtered call (filter_schedule; exported as ragraph_topk_cosine_filtered_plan).
// Part of csrc/topk_filter.hip (textually included there, inside its namespace / after its helpers): split out in round 6 so
// that the ring, the candidate path and the launch plumbing can be read -- and changed -- apart.  No include guard on purpose:
// these are not stand-alone headers.

// Schedule of a call: exact fp32 top-k over the first n0 keys (its k-th score is the first bound), then bf16 filter +
// exact rescoring over [0, e1), [e1, e2), ... [.., N).  A level's k-th exact score is the next level's bound, so a level
// lets through ~1.3 k (its end / the previous end) keys per query.
//   * Large batches (the bench's 100 k queries): n0 = N/256, ends N/32, N/4, N -- ~100, ~100 and ~40 candidates per
//     query; level 0 is the fp32 tile kernel.  The matrix work dominates, three levels keep the rescoring at ~8 %.
//   * Small and medium batches (B <= 16384): a level costs ~60 us whatever it filters (launches, ring prologue, the
//     rescoring kernel's latency) while candidates are cheap, so fewer, steeper levels win; and level 0 is a slab --
//     the dense kernel (same fmaf chains as everything else) writes the B x n0 scores, topk_rows selects -- which
//     spreads over the whole chip where the tile kernel would run one query tile on a few CUs.  n0 and the number of
//     levels minimise   slab(B, n0) + L (60 us + B * 1.3 k r * 0.4 ns),  r = (N / n0)^(1/L),  under 1.3 k r <= cap / 2.
constexpr int FILTER_MAX_LEVELS = 3;
static int64_t filter_round_up(int64_t n) { return (n + FILTER_PAD_KEYS - 1) / FILTER_PAD_KEYS * FILTER_PAD_KEYS; }

struct FilterSchedule {
  int64_t bound_keys;               // > 0: no exact level 0 -- the first bound comes from a bf16 pass over keys [0, bound_keys)
  int64_t n0;                       // level 0: exact top-k over keys [0, n0)   (bound_keys == 0)
  int slab0;                        // level 0 by dense kernel + topk_rows (needs B * n0 floats of workspace)
  int nlev;                         // filter levels
  int64_t ends[FILTER_MAX_LEVELS];  // their ends (multiples of 256 except the last = N)
  int i8_levels;                    // the last i8_levels levels run on the int8 copy (filter_common.h)
};

constexpr int64_t FILTER_SLAB_MAX_B = 16384;
// up to this many queries the prepare launch also leaves the queries as bf16 (and int8) B operands in fragment order: the direct
// kernel's image (<= 256), and the ring kernel's operand load -- 32 independent 16-byte loads per lane instead of eight
// dependent batches of fp32 loads + conversions (13 us per segment at D = 256), which short launches cannot amortise
// (every filtered call: KeyIndex cuts batches at 262 144 queries.  Large batches have long segments on ONE GPU -- the images
// save ~0.7 % of the bench step -- but the short launches of a key-sharded rank do not: the bound launch of one rank of 8
// spent a quarter of its 0.27 ms converting operands.  The images are 3 D bytes per query: 77 MB at 100 000 queries.)
constexpr int64_t FILTER_QB_MAX_B = 262144;
constexpr int64_t FILTER_SLAB_MAX_SCORES = (int64_t)1 << 26;  // 256 MiB of scores

// Banks of >= 8192 keys (KeyIndex sends >= 16384) take their first bound from the BOUND pass instead of an
// exact level 0: the filter kernel itself runs over the first bound_keys keys and records, per query, the best approximate
// score of each of k consecutive parts; the smallest of the k maxima, minus eps, bounds the final k-th best from below
// (filter_prepare_kernel).  As a bound it is worth the exact k-th best of ~bound_keys / (ln k + 1) keys, and it costs a
// bf16 pass with no lists, no inserts and no fp32 matrix work: 0.7 ms instead of the tile kernel's 3.2 ms for the
// bench's 100 k queries, 40 us instead of the slab's 110 us for 256.  RAGRAPH_FILTER_EXACT_LEVEL0=1 keeps the exact
// level 0 (A/B).
static bool filter_bound_pass_enabled() {
  static const bool on = [] {
    const char* e = getenv("RAGRAPH_FILTER_EXACT_LEVEL0");
    return !(e && atoi(e) != 0);
  }();
  return on;
}

// Parts of the bound pass's prefix: 4 k (at most 128), as many as the prefix has stages (a part is at least one ring stage;
// sub-tiles of the direct kernel are finer), never fewer than k.
static int filter_bound_parts(int k, int64_t bound_keys, int D, int64_t B = 1 << 20, int n_shards = 1) {
  if (B <= 64) return k;  // a handful of queries: the minimum of k part maxima, taken inside the filter launch's prologue
                          // (filter_threshold) -- the extra selection launch would cost more than the shorter prefix saves
  int64_t g = 4 * (int64_t)k;
  if (g > 128) g = 128;
  if (n_shards > 1) g = (g + n_shards - 1) / n_shards;  // (pooled through the exchange: 4 k parts over all shards)
  const int64_t avail = bound_keys == INT64_MAX ? g : bound_keys / (FILTER_STAGE_BYTES / (2 * D));
  if (g > avail) g = avail;
  return (int)(g < k ? k : g);
}
// prefix keys per key of exact sample the bound is worth (see filter_bound_scores_kernel)
static double filter_bound_eff(int k, int parts) {
  if (parts >= 4 * k || parts >= 128) return 1.2;
  if (parts >= 2 * k) return 1.5;
  return log((double)k) + 1.0;
}

// D = 64: a stage of the int8 copy holds 512 keys (32 KB / 64 B) and a level starts at a whole stage, so the inner level ends
// are multiples of 512 -- a level that started at an odd multiple of 256 would begin with the previous level's last 256
// keys again, and a key listed twice breaks the selection (distinct pairs are what its ranks count).
static void filter_align_ends(FilterSchedule& sc, int D) {
  if (D != 64) return;
  for (int l = 0; l + 1 < sc.nlev; ++l) {
    const int64_t e = sc.ends[l] / 512 * 512;
    if (e >= 512 && (l == 0 || e > sc.ends[l - 1])) sc.ends[l] = e;
  }
  if (sc.bound_keys > sc.ends[0]) sc.bound_keys = sc.ends[0] / FILTER_PAD_KEYS * FILTER_PAD_KEYS;
}

// n_shards > 1 (row-sharded bank, N = the largest shard): the shards pool their first samples through the exchange, so
// the sample is planned for the WHOLE bank and every shard scans its share of the prefix.
static int rescore_slices(int64_t B, int k);
// Sharded banks of up to this many shards keep the SCORED lists on their int8 levels (and the schedule that goes with them):
// a shard's own round-1 bound comes from 1 / G of the keys while the level's threshold was pooled over all shards' earlier
// levels -- at G = 2 the shard's bound is still the sharper one (half of the bank against a quarter), from G = 4 it is not and
// the second round only adds latency (profiles/r3_emul.txt).  RAGRAPH_FILTER_SCORED_SHARDS: A/B.
static int filter_scored_shards() {
  static const int v = [] { const char* e = getenv("RAGRAPH_FILTER_SCORED_SHARDS"); return e ? atoi(e) : 2; }();
  return v;
}
static FilterSchedule filter_schedule(int64_t B, int64_t N, int D, int k, int n_shards = 1) {
  FilterSchedule sc{};
  const int cap = 2048;
  // scored lists (one bank, >= 2048 queries): an int8 level's rescoring fetches about a third of its candidates' rows, which
  // makes int8 pay on EVERY level (the bench step, 2 / 3 int8 levels: 24.3 / 23.85 ms; without the scores 26.9 / 27.7)
  const bool scored = (n_shards == 1 || n_shards <= filter_scored_shards()) && B >= 2048 && filter_scored_lists(B, D, k);  // (below 2048 queries the plain lists' plans
                                                                                   // stay: a smaller first sample measured slower)
  // (the model's price of an int8 candidate under scored lists, relative to the plain lists'; fitted: 0.6 moves 8192+ queries
  // x 1M keys from two levels to three, all int8 -- 8192: 2.37 -> 2.33 ms, 16384: 4.30 -> 4.13 -- while 0.45 also shrank the
  // first sample of 2048 - 8192 queries, which measured 2 - 4 % slower; RAGRAPH_FILTER_SCORED_CAND: A/B)
  static const double scored_cand = [] { const char* e = getenv("RAGRAPH_FILTER_SCORED_CAND"); return e ? atof(e) : 0.6; }();
  const bool bound = N >= 8192 && filter_bound_pass_enabled();
  // int8 levels (filter_common.h): D = 128 / 256, the ring kernel's batch sizes, banks long enough to be matrix-bound (an
  // int8 level quantises its queries from the fp32 rows per segment where the bf16 levels of up to 16384 queries load a
  // prepared image -- Cora-sized 2708 x 10 000 x 128: 0.087 -> 0.100 ms)
  static const bool i8_d64 = [] { const char* e = getenv("RAGRAPH_FILTER_I8_D64"); return !e || atoi(e) != 0; }();  // A/B
  // (D = 64, the edge flavour: one MFMA per 16-key half and query group, so the epilogue weighs more -- 65 536 x 4M x 64:
  // 22.5 -> 15.5 ms with eight groups per wave; eps is the same 0.02 but the scores' spread is 1/8: fewer extra candidates)
  // (with the prepared int8 operand image and the scored lists, D = 256 also pays on banks of 32 768+ keys from 2048 queries:
  // 4096 x 40 000: 0.189 -> 0.160 ms, 2100 x 60 000: 0.180 -> 0.150, 16 384 x 50 000: 0.64 -> 0.49; not at D = 128 -- 8192 x
  // 50 000: 0.237 -> 0.244 -- nor on shorter banks -- 8192 x 20 000 x 256: 0.221 -> 0.238)
  const bool i8_ok = (D == 128 || D == 256 || (D == 64 && i8_d64)) && B > 256 &&
                     (N * n_shards >= 65536 || (D == 256 && B >= 2048 && N * n_shards >= 32768));
  const bool mid_i8 = i8_ok && N * n_shards < 65536;
  static const bool i8_direct_env = [] { const char* e = getenv("RAGRAPH_FILTER_I8_DIRECT"); return !e || atoi(e) != 0; }();  // A/B
  // (D = 64, round 5: the edge flavour's calls of up to 256 queries -- half the stream, the scores' spread 1/8 against the
  // same eps; RAGRAPH_FILTER_I8_DIRECT_D64=0: A/B)
  static const bool i8_direct_d64 = [] { const char* e = getenv("RAGRAPH_FILTER_I8_DIRECT_D64"); return !e || atoi(e) != 0; }();
  const bool i8_direct = (D == 128 || D == 256 || (D == 64 && i8_direct_d64)) && B <= 256 && N * n_shards >= 65536 && i8_direct_env;
  // bound_keys / eff_div ~ the exact sample the bound is worth: planned for 4 k parts, corrected below if the prefix is
  // too short for that many
  const double eff_div = filter_bound_eff(k, B <= 64 ? k : 4 * k);
  auto prefix_for = [&](int64_t n0) {  // prefix whose bound is worth the exact k-th best of n0 keys
    int64_t nA = filter_round_up((int64_t)((double)n0 * eff_div));
    const int parts = filter_bound_parts(k, nA, D, B);
    if (parts < 4 * k && parts < 128) nA = filter_round_up((int64_t)((double)n0 * filter_bound_eff(k, parts)));
    return nA;
  };
  if (B > FILTER_SLAB_MAX_B || N < 4 * 4096) {
    // (the first sample: with 4 k parts the bound pass is cheap enough for N / 64 -- fewer candidates at the first level,
    // whose sub-tiles otherwise nearly all take the candidate path: 100k x 1M: 38.0 -> 36.5 ms against N / 256;
    // RAGRAPH_FILTER_N0DIV: A/B)
    static const int64_t n0div = [] { const char* e = getenv("RAGRAPH_FILTER_N0DIV"); return e ? (int64_t)atoll(e) : (int64_t)64; }();
    int64_t n0 = N * n_shards / n0div;  // (over all shards)
    if (n0 < 4096) n0 = 4096;
    int64_t nA = bound ? prefix_for(n0) : 0;
    if (n_shards > 1) {  // this shard's share
      n0 /= n_shards;
      nA = filter_round_up(nA / n_shards);
      const int64_t min_keys = filter_round_up((int64_t)k * (FILTER_STAGE_BYTES / (2 * D)));
      if (nA < min_keys) nA = min_keys;
    }
    if (n0 > N) n0 = N;
    if (n0 < k) n0 = k < N ? k : N;
    if (bound) {
      if (nA > N / 4) nA = N / 4 / FILTER_PAD_KEYS * FILTER_PAD_KEYS;
      sc.bound_keys = nA;
    }
    sc.n0 = n0;
    sc.slab0 = 0;  // (measured at 100 k queries: slabs of 16384 cost 3.6 ms -- dense kernel 104 TFLOP/s, topk_rows bound
                   // by its list inserts -- against the tile kernel's 3.2 ms)
    sc.nlev = 0;
    int64_t prev = n0;
    // (RAGRAPH_FILTER_FRACS="a,b": the first ends as fractions N/a, N/b of the bank -- schedule experiments)
    int64_t fracs[2] = {32, 4};
    int nfr = 2;
    if (const char* fe = getenv("RAGRAPH_FILTER_FRACS")) {
      long long a = 0, b = 0;
      nfr = sscanf(fe, "%lld,%lld", &a, &b);
      if (nfr < 1 || a < 2) nfr = 0;
      fracs[0] = a;
      fracs[1] = b;
      if (nfr == 2 && b < 2) nfr = 1;
    }
    for (int fi = 0; fi < nfr; ++fi) {
      const int64_t frac = fracs[fi];
      int64_t e = filter_round_up(N / frac);
      if (e < 4 * prev) e = filter_round_up(4 * prev);  // a level is at least 4x what came before
      if (e * 2 >= N) break;                            // too close to the end: the last level takes the rest
      sc.ends[sc.nlev++] = e;
      prev = e;
    }
    sc.ends[sc.nlev++] = N;
    // the k keys behind the bound must lie inside the first level (it has to find at least k candidates)
    if (sc.bound_keys > sc.ends[0]) sc.bound_keys = sc.ends[0] / FILTER_PAD_KEYS * FILTER_PAD_KEYS;
    if (sc.bound_keys / (FILTER_STAGE_BYTES / (2 * D)) < k) sc.bound_keys = 0;  // every part needs a stage of its own
    sc.i8_levels = i8_ok && B >= 1024 ? (scored ? 3 : 2) : 0;  // (banks below 4 x 4096 keys come here with any batch)
    filter_align_ends(sc, D);
    return sc;
  }
  double best = 1e30;
  int64_t best_n0 = 4096, best_nA = 0;
  int best_L = FILTER_MAX_LEVELS, best_i8 = 0;
  // RAGRAPH_FILTER_FORCE_N0 / _L: schedule experiments (n0 = the exact sample the first bound is worth, L levels)
  static const int64_t force_n0 = [] { const char* e = getenv("RAGRAPH_FILTER_FORCE_N0"); return e ? (int64_t)atoll(e) : (int64_t)0; }();
  static const int force_L = [] { const char* e = getenv("RAGRAPH_FILTER_FORCE_L"); return e ? atoi(e) : 0; }();
  const int stage_keys = FILTER_STAGE_BYTES / (2 * D);
  const double tiles = (double)((B + 511) / 512);
  for (int64_t n0 = 4096; n0 * 4 <= N; n0 *= 2) {
    if (force_n0 > 0 && n0 != force_n0) continue;
    double first;  // cost of the first bound, us
    int64_t nA = 0;
    if (bound) {
      nA = prefix_for(n0);
      if (nA * 4 > N) break;
      if (B <= 256)  // direct kernel: the prefix streams at ~5 TB/s (8.7 / 22 / 40 us for 54 k / 216 k / 216 k keys x 1 / 16 / 256 queries)
        first = 6.0 + (double)nA * 2.0 * D / 5.0e6 * (1.0 + (double)B / 320.0);
      else
        first = 35.0 + (double)nA * 2.0 * D / 3.0e6 + tiles * (double)(nA / stage_keys) * 3.1 / 256.0;
    } else {
      if (B * n0 > FILTER_SLAB_MAX_SCORES) break;
      first = 30.0 + (double)B * (double)n0 * (2.0 * D / 1.0e8 + 4.0 / 3.0e6);
    }
    for (int L = 1; L <= FILTER_MAX_LEVELS; ++L) {
      if (force_L > 0 && L != force_L) continue;
      const double r = pow((double)N / (double)n0, 1.0 / L);
      if (1.3 * k * r > cap / 2 && !(force_n0 > 0 && force_L > 0)) continue;
      // a level: launches + the rescoring kernels' latency floor, plus ~0.4 - 0.5 ns per candidate (1 KB row gather each)
      if (B <= 256) {
        // Direct kernel.  On the int8 copy its pass streams half the bytes and does half the matrix work (one query: 78 ->
        // 40 us of kernel; 256: 124 -> ~75) while ~3x the candidates come back: the same model with 3.9 k r candidates per
        // level and that saving decides between the two -- and moves n0 up when int8 wins.
        // (int8 wins at every batch size from 1 to 256 on the 1M x 256 bank -- 0.106 -> 0.074, 0.124 -> 0.091, 0.139 -> 0.109,
        // 0.192 -> 0.153 ms -- so where it is eligible the model only chooses ITS schedule; the two constants are not
        // comparable across the dtypes)
        for (int q8 = i8_direct ? 1 : 0; q8 <= (i8_direct ? 1 : 0); ++q8) {
          const double cands = 1.3 * k * r * (q8 ? 3.0 : 1.0);
          // (a handful of queries keep S sub-lists of `cap` slots each: filter_cap)
          if (cands > cap * (q8 ? rescore_slices(B, k) : 1) / 2 && !(force_n0 > 0 && force_L > 0)) continue;
          // (the sliced / wide rescoring of a small call is a latency chain: measured 1.2 - 3.7 ns per candidate on the int8
          // schedules -- forced n0 at 1 / 16 / 64 queries, profiles/r3_i8_ab.txt -- where round 2 fitted 0.5 to its bf16 ones)
          const double cost = first + L * (25.0 + (double)B * cands * (q8 ? 2.0e-3 : 0.5e-3)) + (L - 1) * (q8 ? 30.0 : 15.0)  /* (a second pass start-up; 256 queries, int8: one level 0.156, two 0.162 ms) */
                              - (q8 ? 32.0 + 0.08 * (double)B : 0.0);
          if (cost < best) {
            best = cost;
            best_n0 = n0;
            best_nA = nA;
            best_L = L;
            best_i8 = q8 ? L : 0;
          }
        }
        continue;
      }
      // Ring kernel: + the matrix work of each level -- 2 B keys D at ~1.25 PFLOP/s on the bf16 copy, ~2.4 Pop/s on the
      // int8 copy, whose ~5x wider bound passes ~3x the candidates -- for 0, 1 or 2 trailing int8
      // levels.  (Without int8 the matrix term is the same for every (n0, L): the choice among those is round 2's.)
      // Measured against forced schedules at 512 / 1024 / 2048 queries x 1M keys (profiles/r3_i8_ab.txt).
      // (a candidate costs ~0.4 ns while a level's rescoring is a latency chain -- up to ~1000 queries -- and ~0.18 ns once it
      // is bound by the row gathers: 100 000 queries x ~130 candidates x 1 KiB in 1.8 ms)
      const double per_cand = 0.18e-3 + 0.22e-3 * (B <= 1024 ? 1.0 : 1024.0 / (double)B);
      // (int8 candidates per bf16 candidate.  3.0 until the copy got its two scales and the calls their speculative bounds; 2.0
      // fits what tools/i8_rule_grid.py measures now -- 105 shapes of 300 .. 16 384 queries x 70 k .. 1 M keys x D = 64 / 128 /
      // 256, KeyIndex in its steady state, geomean 0.971 of the old rule's time; the banks of 150 k - 500 k keys that moved to
      // int8 0.77 - 0.9 (1100 x 300 k x 256: 0.188 -> 0.144 ms); 1.5 loses up to 1.6 x on 70 k-key banks.  profiles/r5_i8_rule_grid.txt)
      static const double i8_candf = [] { const char* e = getenv("RAGRAPH_FILTER_I8_CANDF"); return e ? atof(e) : 2.0; }();  // A/B
      // (mid_i8 -- D = 256 banks of 32 768 .. 65 535 keys: the constants below were fitted on million-key banks and overprice
      // these shapes' candidates; what measured faster there is the bf16 plan with every level moved to int8: see below)
      for (int i8 = 0; i8 <= (i8_ok && !mid_i8 ? (L < 2 || scored ? L : 2) : 0); ++i8) {
        double cost = first, e_prev = 0.0, e = (double)n0;
        bool fits = true;
        for (int l = 0; l < L; ++l) {
          e = l + 1 == L ? (double)N : e * r;
          const bool q8 = l >= L - i8;
          const double cands = 1.3 * k * r * (q8 ? i8_candf : 1.0);
          if (cands > cap / 2 && !(force_n0 > 0 && force_L > 0)) fits = false;
          cost += 60.0 + (e - e_prev) * (double)B * 2.0 * D / (q8 ? 2.4e9 : 1.25e9) +
                  (double)B * cands * per_cand * (q8 && scored ? scored_cand : 1.0);
          e_prev = e;
        }
        if (fits && cost < best) {
          best = cost;
          best_n0 = n0;
          best_nA = nA;
          best_L = L;
          best_i8 = i8;
        }
      }
    }
  }
  sc.bound_keys = bound ? (best_nA ? best_nA : prefix_for(4096)) : 0;
  if (bound && B > 128 && B <= 256 && n_shards == 1) {
    // the direct kernel deals the prefix's 16-KiB units over all waves of the chip in contiguous runs: 2.4 units per wave take
    // as long as 3 -- a prefix of whole rounds (8 waves x CUs units) costs what it reads: 256 queries x 1M 0.1406 -> 0.1381 ms,
    // 192: 0.1198 -> 0.1180 (up to 128 queries, whose pass is cheaper per key, the shorter prefix loses more than it saves:
    // 64 queries 0.110 -> 0.116).  RAGRAPH_FILTER_BOUND_ROUNDS=0: A/B
    static const int align_env = [] { const char* e = getenv("RAGRAPH_FILTER_BOUND_ROUNDS"); return e ? atoi(e) : 1; }();
    if (align_env) {
      const int64_t round_keys = (int64_t)8 * filter_device_cus() * (16384 / (2 * D));
      int64_t r = (sc.bound_keys + round_keys / 2) / round_keys;
      if (r < 1) r = 1;
      if (r * round_keys * 4 <= N && r * round_keys >= (int64_t)k * 4 * (FILTER_STAGE_BYTES / (2 * D))) sc.bound_keys = r * round_keys;
    }
  }
  sc.n0 = best_n0;
  sc.i8_levels = mid_i8 ? best_L : best_i8;
  sc.slab0 = 1;
  sc.nlev = 0;
  const double r = pow((double)N / (double)best_n0, 1.0 / best_L);
  double e = (double)best_n0;
  for (int l = 0; l + 1 < best_L; ++l) {
    e *= r;
    const int64_t ei = filter_round_up((int64_t)e);
    if (ei * 2 >= N) break;
    sc.ends[sc.nlev++] = ei;
  }
  sc.ends[sc.nlev++] = N;
  if (sc.bound_keys > sc.ends[0]) sc.bound_keys = sc.ends[0] / FILTER_PAD_KEYS * FILTER_PAD_KEYS;
  if (sc.bound_keys / stage_keys < k) sc.bound_keys = 0;  // every part needs a stage of its own: else the exact slab
  if (sc.bound_keys == 0 && B * sc.n0 > FILTER_SLAB_MAX_SCORES) sc.slab0 = 0;
  filter_align_ends(sc, D);
  return sc;
}
