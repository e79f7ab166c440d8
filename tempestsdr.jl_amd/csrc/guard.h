// guard.h -- the sync guard of the TSDR_FAST frame loop.
//
// FAST-mode pixels differ from the reference's by a few ulp and its projection sums are added in tile order, so its
// beta values differ from the oracle's at the 1e-7 level.  The frame-sync decision is findmax over beta
// (FrameSynchronisation.jl:66,76): whenever the best blank-band column leads the best OTHER column by less than that,
// a non-bit-exact evaluation may pick the neighbour.  The guard makes "identical frame-sync indices" unconditional:
// k_beta also reports, per workgroup, its largest column maximum and the largest one of any other column; a frame whose
// relative top-2 margin on either axis is below `thr` (default 2e-5, an order of magnitude above the largest FAST-vs-EXACT
// beta difference measured, tools/measure_beta_error.py) is re-evaluated in the reference's exact operation sequence --
// its 600x800 image (k_down_fused<EXACT>), its projections in the reference's order (k_proj) and its beta scan (k_beta) --
// by three launches: the first, a small persistent grid, works out which frames are flagged and walks only their tiles;
// the workgroups of the other two exit at once for every other frame.  Such a frame then carries EXACT pixels and the
// oracle's indices; all other frames keep FAST pixels and have margins no rounding difference can bridge.
#pragma once
#include <hip/hip_runtime.h>

namespace tsdr {

struct GuardArgs {
  const uint2 *top2 = nullptr;  // [frames][nbx + nby] {best, second-best-other-column} words from k_beta; null: no guard
  int nbx = 0, nby = 0;         // k_beta workgroups per frame along x / y
  float thr = 0.f;              // relative margin below which a frame is re-evaluated
  int *flags = nullptr;         // [frames] out: 1 = re-evaluated
  unsigned long long *stats = nullptr;  // [0] frames checked, [1] frames re-evaluated (running totals)
};

constexpr int kGuardChunk = 256;   // frames per guard_image launch

#ifdef __HIPCC__
// true when frame f needs the exact re-evaluation (one lane does a whole frame: two short scans of the records)
__device__ inline bool guard_eval(const GuardArgs &g, int f) {
  bool bad = false;
  const uint2 *base = g.top2 + (size_t)f * (size_t)(g.nbx + g.nby);
  for (int axis = 0; axis < 2; ++axis) {
    const int nb = axis == 0 ? g.nbx : g.nby;
    const uint2 *e = base + (axis == 0 ? 0 : g.nbx);
    unsigned gb = 0u, gs = 0u;  // largest column maximum; largest one of any other column
    for (int i = 0; i < nb; ++i) {
      const uint2 v = e[i];
      if (v.x > gb) { gs = max(gb, v.y); gb = v.x; }      // new leader: the old one and the newcomer's runner-up compete
      else gs = max(gs, v.x);                              // (v.x == gb: another column holds the same maximum, an exact tie)
    }
    const float b = __uint_as_float(gb), s2 = __uint_as_float(gs);
    if (!((b - s2) > g.thr * b)) bad = true;  // also catches NaN / Inf / all-zero images
  }
  return bad;
}
#endif

}  // namespace tsdr
