// scg_debug.h — option bits of scg_forward that exist for the library's OWN A/B tests and parity checks, not for callers (they
// are not in include/scg_raster.h; bits 0 and 1 of `options` are reserved for them there).  The parity suite renders with and
// without each fusion and demands bit-identical outputs (tests/test_gpu_parity.py); tools/ab_inproc.py times them against each
// other in one process.
#pragma once

enum {
    SCG_DEBUG_SEPARATE_SORT = 1,   // per-tile sort kernel + one-wave blend kernel instead of the forward blend that sorts its own tiles
    SCG_DEBUG_SEPARATE_HIST = 2    // geometry_forward_kernel + tile_hist_kernel instead of the geometry kernel that builds the histograms
};
