// Exhaustive host-side check of ragraph_amd/csrc/segment_plan.h: for a sweep of (tiles, stages, workgroups) every
// tile's stage range [0, NS) must be covered exactly once, slots must be distinct, in stage order and below
// max_slots(), and the `last` flag must sit on the segment that ends at NS.  Prints the worst load imbalance seen.
//   g++ -O2 -std=c++17 -I ragraph_amd/csrc tools/check_segment_plan.cpp -o /tmp/check_segment_plan && /tmp/check_segment_plan
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "segment_plan.h"

using ragraph::Segment;
using ragraph::SegmentWalker;

static long checked = 0;
static double worst_ratio = 0;
static long worst_cfg[4];

static bool check(int64_t nq, int64_t NS, int W, int lb_min, int warm, int depth) {
  struct Piece { int64_t st0, st1; int slot, last; };
  std::vector<std::vector<Piece>> tiles(nq);
  const SegmentWalker::Choice ch = SegmentWalker::choose_depth(nq, NS, W, lb_min, warm);
  if (depth < 0) depth = ch.depth;
  int64_t P;
  {
    SegmentWalker w(nq, NS, W, lb_min, warm, depth, -1);
    Segment s;
    while (w.next(s)) {
    }
    P = w.used;
  }
  int64_t max_load = 0, total = 0;
  for (int c = 0; c < W; ++c) {
    SegmentWalker w(nq, NS, W, lb_min, warm, depth, c);
    Segment s;
    int64_t load = 0;
    int nseg = 0;
    while (w.next(s)) {
      if (s.tile < 0 || s.tile >= nq || s.st0 < 0 || s.st1 > NS || s.st0 >= s.st1) {
        printf("bad segment nq=%ld NS=%ld W=%d c=%d: tile %ld [%ld,%ld)\n", (long)nq, (long)NS, W, c, (long)s.tile, (long)s.st0, (long)s.st1);
        return false;
      }
      if (s.slot < 0 || s.slot >= P) {
        printf("slot %d outside [0,%ld) nq=%ld NS=%ld W=%d c=%d\n", s.slot, (long)P, (long)nq, (long)NS, W, c);
        return false;
      }
      tiles[s.tile].push_back({s.st0, s.st1, s.slot, s.last});
      load += (s.st1 - s.st0) + warm;
      if (++nseg > 100000) { printf("runaway walker\n"); return false; }
    }
    max_load = std::max(max_load, load);
    total += load;
  }
  for (int64_t t = 0; t < nq; ++t) {
    auto& v = tiles[t];
    std::sort(v.begin(), v.end(), [](const Piece& a, const Piece& b) { return a.st0 < b.st0; });
    int64_t pos = 0;
    for (size_t i = 0; i < v.size(); ++i) {
      if (v[i].st0 != pos) { printf("gap/overlap nq=%ld NS=%ld W=%d tile %ld at %ld (segment starts %ld)\n", (long)nq, (long)NS, W, (long)t, (long)pos, (long)v[i].st0); return false; }
      pos = v[i].st1;
      if (i && v[i].slot <= v[i - 1].slot) { printf("slots not increasing nq=%ld NS=%ld W=%d tile %ld\n", (long)nq, (long)NS, W, (long)t); return false; }
      if (v[i].last != (v[i].st1 == NS)) { printf("last flag wrong nq=%ld NS=%ld W=%d tile %ld\n", (long)nq, (long)NS, W, (long)t); return false; }
    }
    if (pos != NS) { printf("tile %ld covered to %ld of %ld (nq=%ld W=%d)\n", (long)t, (long)pos, (long)NS, (long)nq, W); return false; }
  }
  const double ideal = (double)(nq * NS) / W;
  if (warm > 0 && ideal > 50.0 * warm) {  // imbalance only means something when streams dwarf the warm-up
    const double ratio = max_load / ideal;
    if (ratio > worst_ratio) { worst_ratio = ratio; worst_cfg[0] = nq; worst_cfg[1] = NS; worst_cfg[2] = W; worst_cfg[3] = P; }
  }
  ++checked;
  return true;
}

int main() {
  const int Ws[] = {1, 2, 3, 5, 8, 32, 256};
  const int64_t NSs[] = {1, 2, 3, 4, 7, 31, 94, 157, 625, 1000, 31250, 250000};
  for (int W : Ws)
    for (int64_t NS : NSs)
      for (int64_t nq = 1; nq <= 3 * W + 2 && nq <= 600; ++nq)
        for (int lb_min : {1, 4, 50})
          for (int warm : {0, 33, 131}) {
            if (NS * nq > (int64_t)4e9) continue;
            for (int depth : {-1, 0, 1, 2, 5, 79})
              if (!check(nq, NS, W, lb_min, warm, depth)) return 1;
          }
  printf("segment plans checked: %ld, worst load / ideal (streams > 50 warm-ups): %.4f at nq=%ld NS=%ld W=%ld (slots %ld)\n", checked,
         worst_ratio, worst_cfg[0], worst_cfg[1], worst_cfg[2], worst_cfg[3]);
  // the shapes the planner is tuned for: 1M-key bank (31250 stages), XCD groups of 32 and the single group of 256
  for (int W : {32, 256}) {
    worst_ratio = 0;
    for (int64_t nq = 1; nq <= (W == 32 ? 200 : 63); ++nq) check(nq, 31250, W, 4, 131, -1);
    printf("NS=31250 W=%d: worst load / ideal %.4f at nq=%ld (slots %ld)\n", W, worst_ratio, worst_cfg[0], worst_cfg[3]);
  }
  // the c2 plan, for the record
  for (int c : {0, 16, 17, 31}) {
    SegmentWalker w(49, 31250, 32, 4, 131, SegmentWalker::choose_depth(49, 31250, 32, 4, 131).depth, c);
    Segment s;
    printf("c2 group of 49 tiles, workgroup %2d:", c);
    while (w.next(s)) printf(" tile %ld [%ld,%ld) slot %d%s;", (long)s.tile, (long)s.st0, (long)s.st1, s.slot, s.last ? " last" : "");
    printf("\n");
  }
  return 0;
}
