// greb_step_order.h -- host side: what a row of the 384-wide circulation sub-step costs a wavefront, and how the rows of
// a field are cut into strips under a cost cap.  Shared by the launch-order builders of greb_step_rows.hip (one launch
// per sub-step) and greb_circ_rows.hip (one launch per circulation call).
#pragma once
#include <algorithm>
#include <vector>

#include "greb_kernels.h"

namespace greb {
namespace {

// What a row costs, in cycles (tools/stamp_step_rows.py, tools/step_timeline.py; profiles/r03_g384_substep_*):
//   issue  the issue slots it takes on its SIMD -- one instruction per 4 cycles, shared by the SIMD's two wavefronts:
//          ~480 instructions for a streamed row, 36 per chain sweep (33 without the clamp minimum), ~210 to set a chain up;
//   wall   what the row takes a wavefront that has the SIMD to itself: a streamed row waits for memory (3 500 cycles for
//          1 900 of issue -- 2 200 as it shares the SIMD, the figure used), a chain never waits.
// Two strips on one SIMD end after max(their walls, the sum of their issues): measured 94 000 cycles for a polar strip of
// 63 500 beside a 16-row streaming strip (30 400 of issue), 61 000-70 000 for two such streaming strips.
struct RowCost { int issue, wall; };
constexpr int kRowIssue = 1900, kRowWall = 3500, kSweepCycles = 147, kChainSetupCycles = 850, kFillIssue = 700, kFillWall = 3800;
RowCost step_row_cost(const RowTables& t, int k) {
  const int d = t.dif_time2[k], a = t.adv_time2[k];
  static const int row_issue = tuning_int("GREB_STEP_ROWCOST", kRowIssue); // -DGREB_TUNING builds only
  const int chains = (d > 1 ? kChainSetupCycles + kSweepCycles * d : 0) + (a > 1 ? kChainSetupCycles + kSweepCycles * a : 0);
  return {row_issue + chains, kRowWall + chains};
}


struct Strip { int field, k0, k1; long long issue, wall; };
// Rows [a, b) of a field with row table t: as few strips as the two caps allow, cut where the cumulative issue crosses
// equal shares (a greedy cut leaves every strip some way below its cap: more strips, or a larger S, than needed)
inline void cut_rows(const RowTables& t, int a, int b, long long cap_issue, long long cap_wall, std::vector<Strip>& mine) {
  long long fi = 0, fw = 0;
  for (int k = a; k < b; ++k) { const RowCost c = step_row_cost(t, k); fi += c.issue; fw += c.wall; }
  const long long ci = std::max<long long>(1, cap_issue - kFillIssue), cw = std::max<long long>(1, cap_wall - kFillWall);
  const int n = (int)std::min<long long>(b - a, std::max((fi + ci - 1) / ci, (fw + cw - 1) / cw));
  long long acc = 0, issue = kFillIssue, wall = kFillWall;
  int start = a, cut = 1;
  for (int k = a; k < b; ++k) {
    const RowCost c = step_row_cost(t, k);
    // the share boundary cut * fi / n lies nearer the start of row k than its end: close the strip before it
    if (k > start && cut < n && 2 * n * acc + (long long)n * c.issue >= 2 * fi * cut) {
      mine.push_back({0, start, k, issue, wall});
      start = k; issue = kFillIssue; wall = kFillWall;
      while (cut < n && 2 * n * acc + (long long)n * c.issue >= 2 * fi * cut) ++cut; // (a dear row may span shares)
    }
    acc += c.issue; issue += c.issue; wall += c.wall;
  }
  mine.push_back({0, start, b, issue, wall});
}

} // namespace
} // namespace greb
