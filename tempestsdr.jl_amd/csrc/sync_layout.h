// sync_layout.h -- layout of the projection partial sums handed from their producer (k_proj, or the raster
// kernel's in-walk sums in TSDR_FAST mode) to k_fold.  Per frame: colpart[ncp][x_t] | rowpart[nrp][y_t]; the
// partials of one element are added in index order.
#pragma once
#include <cstddef>

namespace tsdr {

struct ProjLayout {
  int ncp = 0;  // partial sums per image column (0: nothing produced yet)
  int nrp = 0;  // partial sums per image row
};

static inline size_t proj_floats(int y_t, int x_t, ProjLayout pl) { return (size_t)pl.ncp * x_t + (size_t)pl.nrp * y_t; }

}  // namespace tsdr
