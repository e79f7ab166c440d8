// guard.h -- the sync guard of the TSDR_FAST frame loop.
//
// FAST-mode pixels differ from the reference's by a few ulp and its projection sums are added in tile order, so its
// beta values differ from the oracle's at the 1e-7 level.  The frame-sync decision is findmax over beta
// (FrameSynchronisation.jl:66,76): whenever the best blank-band column leads the best OTHER column by less than that,
// a non-bit-exact evaluation may pick the neighbour.  The guard makes "identical frame-sync indices" unconditional:
// k_beta also reports, per workgroup, its largest column maximum and the largest one of any other column; a frame whose
// relative top-2 margin on either axis is below `thr` (default 2e-5, an order of magnitude above the largest FAST-vs-EXACT
// beta difference measured, tools/measure_beta_error.py) is re-evaluated in the reference's exact operation sequence --
// its 600x800 image (down_fused_body<EXACT>), its projections in the reference's order (proj_wg) and its beta scan (beta_wg) --
// by ONE launch (k_guard, sync.hip): a small persistent grid whose workgroups all leave at once when no frame is flagged,
// and otherwise work through a dependency-ordered queue of (frame, tile / row block / centre block) items.  Such a frame then carries EXACT pixels and the
// oracle's indices; all other frames keep FAST pixels and have margins no rounding difference can bridge.
#pragma once
#include <hip/hip_runtime.h>

namespace tsdr {

struct GuardArgs {
  const uint2 *top2 = nullptr;  // [frames][nbx + nby] {best, second-best-other-column} words from k_beta; null: no guard
  int nbx = 0, nby = 0;         // k_beta workgroups per frame along x / y
  float thr = 0.f;              // relative margin below which a frame is re-evaluated
  int *flags = nullptr;         // [frames] out: 1 = re-evaluated
  unsigned long long *stats = nullptr;  // [0] frames checked, [1] frames flagged (running totals; resettable), [2] the same
                                        // two as one never-reset word: checked << 32 | flagged (both mod 2^32)
  unsigned long long *host = nullptr;   // pinned ring entry of THIS call: host_tag << 48 | frames << 24 | flagged (what the
  unsigned long long host_tag = 0;      // adaptive route's decision folds in submission order, frames.hip:guard_auto_update)
  int count_only = 0;                   // 1: the buffer was computed in the exact sequence as a whole; only count
  // a call of more than kGuardChunk frames is several launches: the earlier ones add their counts to `acc` ({frames, flagged},
  // per pipeline lane, zero between calls; host == null for them), the last one publishes the call's totals and clears it
  unsigned *acc = nullptr;
};

constexpr int kGuardChunk = 256;   // frames per guard_image launch

#ifdef __HIPCC__
// the scan of one axis' records: largest column maximum, largest one of any other column -> flagged?
template <int NB>
__device__ inline bool guard_axis(const uint2 *e, int nb, float thr) {
  unsigned gb = 0u, gs = 0u;
  if (NB > 0) {  // compile-time count: every record is requested before the first compare
    uint2 v[NB > 0 ? NB : 1];
#pragma unroll
    for (int i = 0; i < NB; ++i) v[i] = e[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (v[i].x > gb) { gs = max(gb, v[i].y); gb = v[i].x; } else gs = max(gs, v[i].x);
    }
  } else {
    for (int i = 0; i < nb; ++i) {
      const uint2 v = e[i];
      if (v.x > gb) { gs = max(gb, v.y); gb = v.x; }      // new leader: the old one and the newcomer's runner-up compete
      else gs = max(gs, v.x);                              // (v.x == gb: another column holds the same maximum, an exact tie)
    }
  }
  const float b = __uint_as_float(gb), s2 = __uint_as_float(gs);
  return !((b - s2) > thr * b);  // also catches NaN / Inf / all-zero images
}

// true when frame f needs the exact re-evaluation (one lane does a whole frame: two short scans of the records)
__device__ inline bool guard_eval(const GuardArgs &g, int f, bool *y_too = nullptr) {
  const uint2 *base = g.top2 + (size_t)f * (size_t)(g.nbx + g.nby);
  bool bx, by;  // (both axes are always evaluated: no divergent second scan)
  if (g.nbx == 13 && g.nby == 10) {  // the 600 x 800 rendering image (the only SyncXY the frame loop accepts)
    bx = guard_axis<13>(base, 13, g.thr);
    by = guard_axis<10>(base + 13, 10, g.thr);
  } else {
    bx = guard_axis<0>(base, g.nbx, g.thr);
    by = guard_axis<0>(base + g.nbx, g.nby, g.thr);
  }
  if (y_too) *y_too = by;
  return bx || by;
}
#endif

}  // namespace tsdr
