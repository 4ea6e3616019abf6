// sl3d_capi_internal.h -- what the translation units behind include/sl3d.h share (not part of the ABI; every symbol is hidden):
//   sl3d_capi_context.cpp  errors, sl3d_create / sl3d_destroy, calibration (T0 and the per-calibration tables), stream helpers
//   sl3d_capi_inputs.cpp   selection masks (prepared now or deferred to the launch: H0 / S3b / S3d) and frames in, [col][row] layouts
//   sl3d_capi_run.cpp      the per-stage entry points, every fused launch (run_fused), getters, the host-buffer pipeline
//   sl3d_capi_clouds.cpp   O1 / N2 / N3: ordered clouds, their consumers, colour, registration
//   sl3d_capi_next.cpp     N1 / N4: patterns, synthetic captures, cvUndistort2, plain device copies
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "sl3d_ctx.h"

#define SL3D_INTERNAL __attribute__((visibility("hidden")))

template <typename T>
inline int dev_alloc(sl3d_ctx *c, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T));
    if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? SL3D_E_NOMEM : SL3D_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    c->allocs.push_back(q);
    *p = (T *)q;
    return SL3D_OK;
}

SL3D_INTERNAL int launched(sl3d_ctx *x, int hip_err);   // a launch's hipError_t -> status (+ the context's error text)
SL3D_INTERNAL int need_keep(sl3d_ctx *x);               // SL3D_E_STATE unless the context keeps the stage planes
SL3D_INTERNAL int check_view(sl3d_ctx *x, int view, int n = 1);
// what kind of memory `p` is: 0 = pageable host, 1 = pinned host, 2 = device (*device = its ordinal)
SL3D_INTERNAL int memory_kind(const void *p, int *device = nullptr);
SL3D_INTERNAL bool is_pinned_host(const void *p);
SL3D_INTERNAL int ensure_colrow(sl3d_ctx *x, size_t bytes);

// the part of the mask plane that holds source pixels: the window + 2-pixel halo, clipped to the frame
struct MaskRegion {
    int gx0, gx1, gy0, gy1;
};
SL3D_INTERNAL MaskRegion mask_region(const KParams &P, MaskSrc &S);
// selected quads of a view as its last preparation counted them / are all views of a launch known to be sparsely selected
SL3D_INTERNAL bool quads_known(const sl3d_ctx *x, int view, unsigned *quads);
SL3D_INTERNAL bool sparse_views(const sl3d_ctx *x, int first, int n);
// deferred masks (sl3d_capi_inputs.cpp): prepare what is still deferred of views [first_view, first_view + n_views) (callers_only: only
// masks that lie in the CALLER's memory); drop / prepare what a new mask of those views supersedes; drain the stream for the caller
SL3D_INTERNAL int flush_masks(sl3d_ctx *x, int first_view, int n_views, bool callers_only = false);
SL3D_INTERNAL int supersede_masks(sl3d_ctx *x, int first_view, int n_views, int slots);
SL3D_INTERNAL int sync_for_caller(sl3d_ctx *x);
#define SYNC_FOR_CALLER(x)                     \
    do {                                       \
        const int rc_ = sync_for_caller(x);    \
        if (rc_) return rc_;                   \
    } while (0)

// every fused launch of the library (sl3d_capi_run.cpp)
SL3D_INTERNAL bool maskin_launch(const sl3d_ctx *x, int first_view, int n_views, bool keep);
SL3D_INTERNAL int small_launch_overlaps(sl3d_ctx *x, int first_view, int n_views, bool *overlap);
SL3D_INTERNAL int run_fused(sl3d_ctx *x, int first_view, int n_views, bool keep, int cmode, bool may_overlap = false);
