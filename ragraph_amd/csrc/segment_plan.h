// Work plan of the tile kernel (topk_cosine.hip): which (query tile, stage range) SEGMENTS each persistent workgroup
// of a group walks.  Plain C++ so the same code runs in the kernel, in the host-side planner, and in the CPU coverage
// test (tests/test_cpu_abi_and_host.py builds tools/check_segment_plan.cpp with g++).
//
// A group is W workgroups that share an L2 (the 32 of one XCD with the XCD-aware mapping, else all of them) and owns nq
// query tiles; a pass over the bank is NS stages.  Goals, in this order: (1) every workgroup gets the same load -- there
// is one workgroup per CU and nothing rebalances later; a segment costs its stages plus one list WARM-UP (~k ln(n/k)
// inserts per query, measured ~420 k keys' worth of streaming); (2) workgroups that run side by side read the SAME
// stages, so a stage comes from HBM once per group instead of once per workgroup.
//   * full rounds: nq / W times, workgroup c streams the whole bank for tile round*W + c (all W in lockstep);
//   * the left = nq % W tiles: budget = left*NS/W stages per workgroup.  Up to `depth` lockstep steps of Euclid's
//     algorithm on (left, W): with T tiles remaining (all with the same unread range [lo, hi)) and C workgroups with
//     `bud` stages each,
//       T <  C : the first T workgroups read the next `bud` stages of one tile each, in lockstep, and are done
//                (or the whole rest, when what would remain is not worth another cut);
//       T >= C : every workgroup reads the whole rest of one tile (hi - lo <= bud), in lockstep, and keeps bud - (hi-lo);
//   * then the remainder (T tiles x [lo, hi), C workgroups) is either cut tile by tile into C / T equal pieces
//     (ALIGNED: one segment per workgroup, C % T workgroups idle) or laid end to end and cut into C equal pieces
//     (LINEAR: perfectly even, but a piece that straddles a tile boundary is two segments), whichever costs less.
//   choose_depth() simulates the loads for every depth and takes the deepest one within 0.4 % (of the launch) of the best: lockstep
//   steps save re-reads, but late ones leave a few workgroups with many short remainders.
// Lists: a tile's segments get slots 0, 1, ... in stage order; `last` marks the segment that ends at NS.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RG_HD __host__ __device__
#else
#define RG_HD
#endif

namespace ragraph {

struct Segment {
  int64_t tile;      // group-local query tile
  int64_t st0, st1;  // stage range [st0, st1)
  int slot;          // which of the tile's partial lists this segment fills
  int last;          // 1: st1 == NS (the tile's unused slots follow this one)
};

struct SegmentWalker {
  static constexpr int MIN_SHARE = 4;  // a lockstep step is worth a cut only if it saves this many re-reads

  // plan constants
  int64_t NS;
  int W, c, lb_min, warm, depth;
  // full rounds
  int64_t rounds, round;
  // Euclid state over the leftover tiles
  int64_t t0, nT, c0, nC, lo, hi, bud;
  int slot0;  // slots already used by every remaining tile
  int steps;  // lockstep steps taken
  int phase;  // 1: lockstep steps, 2: remainder, 3: done
  // remainder
  int aligned;
  int64_t m, per;            // aligned: pieces per tile, stages per piece
  int64_t Lb, lin, lin_end;  // linear: stages per workgroup, this workgroup's span of the line
  // bookkeeping for the planner (host)
  int used;              // most slots any tile has used so far
  int64_t carry, worst;  // load (stages + warm-ups) of the workgroups still unfinished / of the most loaded finished one

  // c < 0 walks no workgroup: only the step bookkeeping (planner).
  RG_HD SegmentWalker(int64_t nq, int64_t NS_, int W_, int lb_min_, int warm_, int depth_, int c_)
      : NS(NS_), W(W_), c(c_), lb_min(lb_min_), warm(warm_), depth(depth_) {
    rounds = nq / W;
    round = 0;
    t0 = rounds * W;
    nT = nq % W;
    c0 = 0;
    nC = W;
    lo = 0;
    hi = NS;
    bud = nT ? (nT * NS + W - 1) / W : 0;
    if (bud < lb_min) bud = lb_min;
    slot0 = steps = 0;
    used = 1;
    carry = worst = 0;
    phase = nT ? 1 : 3;
    aligned = 0;
    m = per = Lb = lin = lin_end = 0;
  }

  RG_HD void enter_remainder() {
    phase = 2;
    const int64_t len = hi - lo;
    Lb = (nT * len + nC - 1) / nC;
    if (Lb < lb_min) Lb = lb_min;
    // segments of the most loaded workgroup under the linear cut: pieces shorter than a tile straddle at most one
    // boundary, longer ones cross Lb / len of them
    const int64_t lin_segs = Lb < len ? (nT > 1 ? 2 : 1) : Lb / len + 1;
    const int64_t lin_cost = Lb + lin_segs * warm;
    m = nC / nT;  // aligned needs at least one workgroup per tile
    int64_t al_cost = -1;
    if (m >= 1) {
      const int64_t mmax = len / lb_min > 1 ? len / lb_min : 1;  // no piece shorter than lb_min stages
      if (m > mmax) m = mmax;
      per = (len + m - 1) / m;
      al_cost = per + warm;
    }
    aligned = (al_cost >= 0 && al_cost <= lin_cost) ? 1 : 0;
    const int64_t cost = aligned ? al_cost : lin_cost;
    if (carry + cost > worst) worst = carry + cost;
    const int rem_slots = slot0 + (aligned ? (int)m : (int)((len + Lb - 1) / Lb) + 1);
    if (rem_slots > used) used = rem_slots;
    const int64_t ci = c - c0;
    lin = ci >= 0 ? ci * Lb : nT * len;
    lin_end = lin + Lb < nT * len ? lin + Lb : nT * len;
  }

  RG_HD bool next(Segment& s) {
    if (c >= 0 && round < rounds) {
      s.tile = round * W + c;
      s.st0 = 0;
      s.st1 = NS;
      s.slot = 0;
      s.last = 1;
      ++round;
      return true;
    }
    while (phase == 1) {
      if (nT <= 0 || nC <= 0) {
        if (carry > worst) worst = carry;
        phase = 3;
        break;
      }
      const int64_t len = hi - lo;
      if (steps >= depth || (nT < nC ? nT : nC) < MIN_SHARE) {
        enter_remainder();
        break;
      }
      ++steps;
      if (slot0 + 1 > used) used = slot0 + 1;
      if (nT >= nC) {  // every remaining workgroup reads the whole rest of one tile
        const bool mine = c >= c0 && c < c0 + nC;
        s.tile = t0 + (c - c0);
        s.st0 = lo;
        s.st1 = hi;
        s.slot = slot0;
        s.last = 1;
        t0 += nC;
        nT -= nC;
        bud -= len;
        carry += len + warm;
        if (mine) return true;
      } else {  // the first nT workgroups read the next `take` stages of one tile each and are done
        // Leaving r = len - bud stages per tile hands nT*(r + warm-up) to the nC - nT workgroups that remain; when
        // that overloads them by more than r (few workgroups left for many short remainders), or r is negligible,
        // this step's workgroups take the whole rest instead and run r stages long.
        const int64_t r = len - bud, rest = nC - nT;
        const int64_t tiny = warm < bud / 8 ? warm : bud / 8;
        const bool absorb = r <= tiny || rest <= 0 || r <= (nT * (r + warm)) / rest - bud;
        const int64_t take = absorb ? len : bud;
        const bool mine = c >= c0 && c < c0 + nT;
        s.tile = t0 + (c - c0);
        s.st0 = lo;
        s.st1 = lo + take;
        s.slot = slot0;
        s.last = (lo + take == hi);
        c0 += nT;
        nC -= nT;
        lo += take;
        ++slot0;
        if (carry + take + warm > worst) worst = carry + take + warm;
        if (lo >= hi) nT = 0;
        if (mine) {
          phase = 3;
          return true;
        }
      }
    }
    if (phase == 2) {
      const int64_t len = hi - lo;
      if (aligned) {
        phase = 3;  // one segment per workgroup
        const int64_t ci = c - c0;
        if (ci < 0) return false;
        const int64_t jt = ci / m, piece = ci % m;
        const int64_t a = piece * per;
        if (jt >= nT || a >= len) return false;
        s.tile = t0 + jt;
        s.st0 = lo + a;
        s.st1 = a + per < len ? lo + a + per : hi;
        s.slot = slot0 + (int)piece;
        s.last = (s.st1 == hi);
        return true;
      }
      if (c < c0 || lin >= lin_end) {
        phase = 3;
        return false;
      }
      const int64_t jt = lin / len;
      const int64_t a = lin - jt * len;
      const int64_t b = (a + (lin_end - lin) < len) ? a + (lin_end - lin) : len;
      s.tile = t0 + jt;
      s.st0 = lo + a;
      s.st1 = lo + b;
      s.slot = slot0 + (int)((c - c0) - (jt * len) / Lb);  // workgroup (jt*len)/Lb holds the tile's first piece
      s.last = (lo + b == hi);
      lin += b - a;
      return true;
    }
    return false;
  }

  struct Choice {
    int depth;
    int64_t slots;  // upper bound of the slots any tile uses
    int64_t load;   // stages + warm-ups of the most loaded workgroup, full rounds excluded
  };

  // Host side: loads of every depth by running the steps without a workgroup (O(depth^2), depth <= W / MIN_SHARE).
  static inline Choice choose_depth(int64_t nq, int64_t NS, int W, int lb_min, int warm) {
    Choice best = {0, 1, 0};
    int64_t min_load = -1;
    int64_t loads[80], slots[80];
    int nd = 0;
    for (int d = 0; d < 80; ++d) {
      SegmentWalker w(nq, NS, W, lb_min, warm, d, -1);
      Segment s;
      while (w.next(s)) {
      }
      loads[d] = w.worst;
      slots[d] = w.used;
      nd = d + 1;
      if (min_load < 0 || w.worst < min_load) min_load = w.worst;
      if (w.steps < d) break;  // no further step exists
    }
    const int64_t whole = (nq / W) * (NS + warm) + min_load;  // with the full rounds every workgroup also runs
    for (int d = nd - 1; d >= 0; --d)
      if (loads[d] <= min_load + whole / 256) {  // deepest plan within 0.4 % (of the whole launch) of the lightest
        best.depth = d;
        best.slots = slots[d];
        best.load = loads[d];
        break;
      }
    return best;
  }
};

}  // namespace ragraph
