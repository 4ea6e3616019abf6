// sl3d_kernels.hip -- gfx950 (MI355X, wave64) kernels beside the fused hot path (which lives in sl3d_fused.h / sl3d_fused_*.hip):
//   k_mask_prepare                      : H0 / S3b / S3d -- selection mask -> 0/1 plane + the valid map after the boundary removal
//   k_wrap / k_unwrap / k_corr / k_tri  : the path cut at the reference's stage boundaries (parity mode; they write the planes
//                                         the reference keeps in globals, including its 8-bit known-answer debug images)
//   k_cam_table / k_proj_table          : T1 per calibration (what the reference tabulates per scan)
//   k_seg_scan / k_seg_close / k_compact_* / k_register : consumers of the clouds (O1, N2, N3)
//   k_undist_* (N4), k_pattern / k_synth (N1), k_to_colrow / k_mask_from_colrow (the reference's own [col][row] layouts),
//   k_atan_selfcheck (the device-side proof that the lattice atan2 equals the host's libm)
// Shared device arithmetic: sl3d_device.h.  Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "sl3d_device.h"
#include "sl3d_maskbits.h"

namespace sl3d {

// sl3d_set_mask(s) on the device (H0 / S3b / S3d).  The source is either the staging plane the host copy filled or the caller's
// own device-resident mask (MaskSrc: address of plane row 0 / byte 0, row stride, and the part of the plane that holds source
// bytes -- window + 2-pixel halo clipped to the frame; everything else counts as unselected and is never loaded).
// One lane = OWN pixels of a plane row x R rows: per row ONE load of OWN bytes (+ the dword to its left and to its right), `byte == 1`
// of those OWN + 8 bytes becomes OWN + 8 bits of one register, and the closed form of the boundary removal (3/wrapped_phase.cpp:
// 253-279) is evaluated on those bits (sl3d_maskbits.h) -- no per-pixel loads, no per-pixel branches.  It writes the 0/1 plane (what
// the per-stage kernel k_wrap reads) and the `band` plane (final valid bytes of every window pixel: what the fused kernel reads),
// and counts the quads that hold a valid pixel: ONE 8-byte store per block {seq, count} into host memory mapped into the device --
// no device atomics, no memset, no copy behind the kernel; the host adds the words up when it needs the number (sparse_views,
// sl3d_capi_inputs.cpp) and knows by the sequence number whether every block of THIS preparation has landed.
// Two shapes (launch_mask_prepare; both measured per view with rocprofv3, profiles/r05_mask_variants.txt):
//   OWN = 4,  256-thread blocks: many short waves -- the latency of a ~2-Mpx mask (1080p: 5.7 us against 6.5)
//   OWN = 16, one wave per block: 16-byte loads and stores, a third of the arithmetic per pixel -- larger masks (12 Mpx: 13.0 us
//             against 16.4) and several views per launch
template <int R, int OWN, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_mask_prepare(const KParams P, int first_view, const MaskSrc S, unsigned long long *__restrict__ partials,
                                                        unsigned seq)
{
    static_assert(OWN == 4 || OWN == 16, "one dword or one 16-byte quad per lane and row");
    constexpr int ND = OWN / 4;
    constexpr unsigned OWN_MASK = (1u << OWN) - 1u;
    typedef unsigned u32x4a __attribute__((ext_vector_type(4), aligned(4)));  // (a caller's device mask is only 4-byte aligned)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ unsigned s_quads;
    if (BLOCK > 64) {
        if (threadIdx.x == 0) s_quads = 0u;
        __syncthreads();
    }
    const int view = first_view + (int)blockIdx.y;
    const int cpr = P.mpitch / OWN, rows = P.H + 2 * SL3D_MASK_HALO;  // OWN-byte columns of a plane row
    const unsigned t = blockIdx.x * (unsigned)BLOCK + threadIdx.x;
    const int strip = (int)(t / (unsigned)cpr), xc = (int)(t - (unsigned)strip * (unsigned)cpr);
    const int pr0 = strip * R;  // first plane row of the lane
    unsigned wave_quads = 0u;
    if (pr0 < rows) {
        const MbCols c = mb_cols(OWN * xc, OWN, P.col0, SL3D_MASK_LPAD, P.fullW, S.bx0, S.bx1);
        const unsigned reg_own = (c.REG >> 4) & OWN_MASK;
        const bool whole = reg_own == OWN_MASK;  // all own bytes hold source pixels: one load
        unsigned own_bytes[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) own_bytes[k] = mb_expand_nibble(reg_own >> (4 * k)) * 0xffu;
        const unsigned outw = (mb_range_bits(SL3D_MASK_LPAD, SL3D_MASK_LPAD + P.W, OWN * xc - 4, OWN + 8) >> 4) & OWN_MASK;  // own pixels inside the window
        const uintptr_t src = S.origin + (uintptr_t)blockIdx.y * S.view_stride;
        MbRow row[R + 3];
        unsigned own[R][ND];
#pragma unroll
        for (int a = 0; a < R + 3; a++) {
            const int pr = pr0 + a - 2;
            unsigned dl = 0u, dr = 0u, d[ND];
#pragma unroll
            for (int k = 0; k < ND; k++) d[k] = 0u;
            if (pr >= S.r0 && pr < S.r1) {
                const uintptr_t p = src + (uintptr_t)pr * S.stride + (uintptr_t)(OWN * xc);
                if (ND == 4 && whole) {
                    const u32x4 q = *(const u32x4a *)p;
                    d[0] = q.x; d[ND > 1 ? 1 : 0] = q.y; d[ND > 2 ? 2 : 0] = q.z; d[ND > 3 ? 3 : 0] = q.w;
                } else {  // (one dword per lane, or the lanes at the region's edge: dword by dword, nothing outside the region is touched)
#pragma unroll
                    for (int k = 0; k < ND; k++)
                        if (reg_own & (0xfu << (4 * k))) d[k] = *(const unsigned *)(p + 4 * k);
                }
                if (c.REG & 0xfu) dl = *(const unsigned *)(p - 4);
                if (c.REG & (0xfu << (OWN + 4))) dr = *(const unsigned *)(p + OWN);
            }
            unsigned V = mb_pack_nibble(mb_eq1_bytes(dl)) | (mb_pack_nibble(mb_eq1_bytes(dr)) << (OWN + 4));
#pragma unroll
            for (int k = 0; k < ND; k++) {
                const unsigned bk = mb_eq1_bytes(d[k]);
                V |= mb_pack_nibble(bk) << (4 + 4 * k);
                if (a >= 2 && a < R + 2) own[a >= 2 && a < R + 2 ? a - 2 : 0][k] = bk & own_bytes[k];
            }
            row[a] = mb_row(V & c.REG, c, P.row0 + pr - SL3D_MASK_HALO, P.fullH);
        }
        unsigned L[R + 3], OK[R + 3];
#pragma unroll
        for (int a = 1; a < R + 2; a++) {
            L[a] = mb_L(row[a], row[a + 1]);
            OK[a] = mb_OK(row[a], L[a], row[a - 1]);
        }
        uint8_t *mask = (uint8_t *)P.mask + (size_t)view * P.mask_view_stride;
        uint8_t *band = (uint8_t *)P.band + (size_t)view * P.px_view_stride;
        const int xb = xc - SL3D_MASK_LPAD / OWN;  // OWN-byte column of the band row
        const bool in_band = xb >= 0 && xb < P.pitch / OWN;
#pragma unroll
        for (int a = 2; a < R + 2; a++) {
            const int pr = pr0 + a - 2, wr = pr - SL3D_MASK_HALO;
            unsigned v = 0u;
            const bool has_band = in_band && wr >= 0 && wr < P.H;
            if (has_band) v = (mb_valid(row[a], L[a], OK[a - 1], OK[a]) >> 4) & outw;
            if (ND == 4) {
                if (pr < rows) {
                    const u32x4 o = {own[a - 2][0], own[a - 2][ND > 1 ? 1 : 0], own[a - 2][ND > 2 ? 2 : 0], own[a - 2][ND > 3 ? 3 : 0]};
                    *(u32x4 *)(mask + (size_t)pr * P.mpitch + (size_t)xc * 16) = o;
                }
                if (has_band) {
                    const u32x4 o = {mb_expand_nibble(v), mb_expand_nibble(v >> 4), mb_expand_nibble(v >> 8), mb_expand_nibble(v >> 12)};
                    *(u32x4 *)(band + (size_t)wr * P.pitch + (size_t)xb * 16) = o;
                }
            } else {
                if (pr < rows) *(unsigned *)(mask + (size_t)pr * P.mpitch + (size_t)xc * 4) = own[a - 2][0];
                if (has_band) *(unsigned *)(band + (size_t)wr * P.pitch + (size_t)xb * 4) = mb_expand_nibble(v);
            }
#pragma unroll
            for (int k = 0; k < ND; k++) wave_quads += (unsigned)__popcll(__ballot(((v >> (4 * k)) & 0xfu) != 0u));
        }
    }
    // (lane 0 of a wave has the wave's smallest strip: if it is past the last row, so is every lane of the wave)
    if (BLOCK > 64) {
        if ((threadIdx.x & 63u) == 0u && wave_quads) atomicAdd(&s_quads, wave_quads);  // LDS
        __syncthreads();
        wave_quads = s_quads;
    }
    if (threadIdx.x == 0) partials[(size_t)view * gridDim.x + blockIdx.x] = ((unsigned long long)seq << 32) | wave_quads;
}

#ifndef SL3D_MASK_ROWS_PER_LANE
#define SL3D_MASK_ROWS_PER_LANE 4
#endif
// The shape is a property of the context (the host reads mask_prepare_blocks(P) words per view, whatever a call covers): masks of up
// to SL3D_MASK_WIDE_PX pixels take the many-short-waves shape, larger ones the 16-byte one
#ifndef SL3D_MASK_WIDE_PX
#define SL3D_MASK_WIDE_PX (6l << 20)
#endif
static bool mask_wide(const KParams &P) { return (long)P.pitch * P.H > SL3D_MASK_WIDE_PX; }
int mask_prepare_blocks(const KParams &P)
{
    const long strips = (P.H + 2 * SL3D_MASK_HALO + SL3D_MASK_ROWS_PER_LANE - 1) / SL3D_MASK_ROWS_PER_LANE;
    return mask_wide(P) ? (int)(((long)(P.mpitch >> 4) * strips + 63) / 64) : (int)(((long)(P.mpitch >> 2) * strips + 255) / 256);
}

int launch_mask_prepare(const KParams &P, int first_view, int n_views, const MaskSrc &S, unsigned long long *partials, unsigned seq, void *stream)
{
    (void)hipGetLastError();
    const dim3 grid((unsigned)mask_prepare_blocks(P), (unsigned)n_views);
    if (mask_wide(P))
        hipLaunchKernelGGL((k_mask_prepare<SL3D_MASK_ROWS_PER_LANE, 16, 64>), grid, dim3(64), 0, (hipStream_t)stream, P, first_view, S, partials, seq);
    else
        hipLaunchKernelGGL((k_mask_prepare<SL3D_MASK_ROWS_PER_LANE, 4, 256>), grid, dim3(256), 0, (hipStream_t)stream, P, first_view, S, partials, seq);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// staged kernels: one pixel per lane, stage boundaries as in the reference
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool px_coords(const KParams &P, int &col, int &row)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    row = (int)(t / P.pitch);
    col = (int)(t - (long)row * P.pitch);
    return row < P.H;
}

__device__ __forceinline__ int load_px(const KParams &P, int view, int plane, int row, int col)
{
    return P.frames[(size_t)view * P.view_stride + (size_t)plane * P.plane_stride + (size_t)row * P.pitch + col];
}

// stage 3: compute_wrapped_phase  3/wrapped_phase.cpp:402-467
__global__ __launch_bounds__(256) void k_wrap(const KParams P, int view, int axis)
{
    int col, row;
    if (!px_coords(P, col, row)) return;
    const size_t px = (size_t)view * P.px_view_stride + (size_t)row * P.pitch + col;
    const MaskView mv = mask_view(P, view);
    const int gx = P.col0 + col, gy = P.row0 + row;
    const bool inwin = col < P.W;
    // check_I_mod_criteria :106-115 -- for F==5 the assignment is commented out: nothing is valid
    const bool sel = inwin && P.F != 5 && mv.V(gx, gy);
    float phi = 0.f;
    uint8_t dbg = 0;
    if (sel) {
        const int base = axis == 0 ? 0 : (P.F + 2 * P.Nv);
        const int i0 = load_px(P, view, base + 0, row, col), i1 = load_px(P, view, base + 1, row, col);
        const int i2 = load_px(P, view, base + 2, row, col), i3 = P.F == 4 ? load_px(P, view, base + 3, row, col) : 0;
        phi = wrapped_phase<false>(P.F, i0, i1, i2, i3, nullptr, atan_consts<false>());  // :175 / :198
        // t3 = 128.0f+127.0f*(phi/(Pi)) (:178), 4-step 127.0f+128.0f*(...) (:199): double arithmetic, rounded to float, then to uchar
        const float t3 = P.F == 3 ? (float)(128.0f + 127.0f * (phi / (PI_REF))) : (float)(127.0f + 128.0f * (phi / (PI_REF)));
        dbg = (uint8_t)(int)t3;
    }
    const bool v = sel && mv.valid(gx, gy);  // boundary removal :253-279
    P.wrapped[axis][px] = phi;
    P.valid_axis[axis][px] = v ? 1 : 0;
    P.dbg3[axis][px] = v ? dbg : 0;  // :274 clears the debug pixel of every removed pixel
}

// stage 4: unwrap_phase  4/phase_unwrap.cpp:367-393 (Gray-code mode, count == 1)
__global__ __launch_bounds__(256) void k_unwrap(const KParams P, int view, int axis)
{
    int col, row;
    if (!px_coords(P, col, row)) return;
    const size_t px = (size_t)view * P.px_view_stride + (size_t)row * P.pitch + col;
    const int gx = P.col0 + col, gy = P.row0 + row;
    const bool v = P.valid_axis[axis][px] == 1;
    int code = -1;  // :143 / :211
    float unw = 0.f;
    uint8_t dbg = 0;
    if (v) {
        const int N = axis == 0 ? P.Nv : P.Nh;
        const int base = (axis == 0 ? 0 : (P.F + 2 * P.Nv)) + P.F;
        int b = 0;
        code = 0;
        for (int i = 0; i < N; i++) {
            const int g = load_px(P, view, base + i, row, col) - load_px(P, view, base + N + i, row, col) >= 0 ? 1 : 0;  // :183
            b ^= g;                                                                                                    // :187-191
            code = code * 2 + b;                                                                                       // :193
        }
        const bool in_range = axis == 0 ? (gx >= 1 && gx <= P.fullW - 2) : (gy >= 1 && gy <= P.fullH - 2);  // :285 / :304
        if (in_range) {
            float w = P.wrapped[axis][px];
            w = (float)((double)w + PI_REF);  // wrapped += Pi   :290 / :308
            P.wrapped[axis][px] = w;
            unw = unwrap_value(w, code);  // :291 / :309
        }
        // save_unwrap_phase_image :334-335 / :353-354: t is float, t*255 is float, cast to uchar (x86 wraps)
        const int ncodes = axis == 0 ? P.ncodes_v : P.ncodes_h;
        const float t = (float)(unw / (2.0 * PI_REF * ncodes));
        dbg = (uint8_t)(int)(t * 255);
    }
    P.code[axis][px] = code;
    P.unwrapped[axis][px] = unw;
    P.dbg4[axis][px] = dbg;
}

// stage 5: compute_c_p_map  5/compute_correspondance.cpp:630-679
__global__ __launch_bounds__(256) void k_corr(const KParams P, int view)
{
    int col, row;
    if (!px_coords(P, col, row)) return;
    const size_t px = (size_t)view * P.px_view_stride + (size_t)row * P.pitch + col;
    bool v = P.valid_axis[0][px] == 1 && P.valid_axis[1][px] == 1;  // merge_valid_maps :60-77
    long cx = 0, cy = 0;
    if (v) {
        double dx, dy;
        const bool okx = correspond(P.unwrapped[0][px], P.fwv, P.PW, cx, dx);
        const bool oky = correspond(P.unwrapped[1][px], P.fwh, P.PH, cy, dy);
        v = okx && oky;
    }
    P.valid[px] = v ? 1 : 0;
    P.cpmap[2 * px + 0] = v ? cx : 0;
    P.cpmap[2 * px + 1] = v ? cy : 0;
}

// stage 7: triangulate  7/triangulation.cpp:1444-1561 (method 3 only; the dead precomputations are not reproduced)
__global__ __launch_bounds__(256) void k_tri(const KParams P, const DevCal C, int view)
{
    int col, row;
    if (!px_coords(P, col, row)) return;
    const size_t px = (size_t)view * P.px_view_stride + (size_t)row * P.pitch + col;
    const float nanv = __builtin_nanf("");
    float x = nanv, y = nanv, z = nanv;
    double X[3] = {0, 0, 0};
    if (P.valid[px] == 1) {
        double u, v, up, vp;
        undistort_reproject((double)(P.col0 + col), (double)(P.row0 + row), C.cam, u, v);
        undistort_reproject((double)P.cpmap[2 * px], (double)P.cpmap[2 * px + 1], C.proj, up, vp);
        PinnedRows PR;
        for (int j = 0; j < 4; j++) { PR.c2[j] = C.Ac[8 + j]; PR.p2[j] = C.Ap[8 + j]; }
        triangulate_px(C, PR, u, v, up, vp, X);
        x = (float)X[0];
        y = (float)X[1];
        z = (float)X[2];
    }
    P.ipoints[3 * px + 0] = X[0];
    P.ipoints[3 * px + 1] = X[1];
    P.ipoints[3 * px + 2] = X[2];
    P.points[3 * px + 0] = x;
    P.points[3 * px + 1] = y;
    P.points[3 * px + 2] = z;
}

// T1 for the projector as a table (RIG 2): displacement of the undistorted + re-projected point from the projector pixel
// itself, for every projector pixel (7/triangulation.cpp:363-378 builds the same table, per scan).  float2: the
// displacement is a few tens of pixels, so its float rounding is ~3e-6 px (1e-9 relative in the 3-D point).
__global__ __launch_bounds__(256) void k_proj_table(const DevCal *__restrict__ C, int PW, int PH, float2 *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= PW) return;
    double u, v;
    undistort_reproject((double)x, (double)y, C->proj, u, v);
    out[(size_t)y * PW + x] = make_float2((float)(u - (double)x), (float)(v - (double)y));
}

// T1 for the camera as a table: what the fused kernel's per-item prologue iterates (5 fixed-point iterations per pixel), once
// per calibration for every pixel of the window (7/triangulation.cpp:252-307 builds cam_undist_points_mat the same way, per
// scan), in a form from which the kernel recovers the SAME doubles: a purely radial model (kind 1, the reference's camera)
// yields (x0*icd, y0*icd) with icd the factor of the last iteration -- one double per pixel; with tangential terms (kind 2)
// the normalised point itself, two doubles per pixel.  The re-projection of the general rig is applied in the kernel.
__global__ __launch_bounds__(256) void k_cam_table(const KParams P, const DevCal *__restrict__ C, int kind, double *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= P.pitch) return;
    double u, v, icd;
    undistort_normalized((double)(P.col0 + x), (double)(P.row0 + y), C->cam, u, v, &icd);
    const size_t i = (size_t)y * P.pitch + x;
    if (kind == 1) out[i] = icd;
    else {
        out[2 * i] = u;
        out[2 * i + 1] = v;
    }
}

int launch_cam_table(const KParams &P, const DevCal *d_cal, int kind, double *out, void *stream)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_cam_table, dim3((P.pitch + 255) / 256, P.H), dim3(256), 0, (hipStream_t)stream, P, d_cal, kind, out);
    return (int)hipGetLastError();
}

// The radial table of a purely radial distortion model (RadEntry, sl3d_internal.h): node i sits at r0^2 = i * r2max / (NODES - 1);
// its quadratic interpolates s = (last iteration's factor) - 1 at the node and at both cell boundaries (half a node spacing away),
// each evaluated by the very iteration cvUndistortPoints runs (radial_factor_m1).
__global__ __launch_bounds__(SL3D_RAD_NODES) void k_radial_table(const DevCal *__restrict__ C, int which, double node_spacing, RadEntry *__restrict__ out)
{
    const int i = (int)threadIdx.x;
    const Intr &I = which == 0 ? C->cam : C->proj;
    const double r2 = (double)i * node_spacing;
    const double s0 = radial_factor_m1(r2, I), sm = radial_factor_m1(r2 - 0.5 * node_spacing, I), sp = radial_factor_m1(r2 + 0.5 * node_spacing, I);
    RadEntry e;
    e.c0 = s0;
    e.c1 = (float)(sp - sm);                    // s(w) = s0 + w*(s+ - s-) + 2*w^2*(s+ + s- - 2*s0),  w in [-1/2, 1/2]
    e.c2 = (float)(2.0 * (sp + sm - 2.0 * s0));
    out[(size_t)blockIdx.x * SL3D_RAD_STRIDE + i] = e;
}

int launch_radial_table(const DevCal *d_cal, int which, double r2max, RadEntry *out, void *stream)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_radial_table, dim3(SL3D_RAD_COPIES), dim3(SL3D_RAD_NODES), 0, (hipStream_t)stream, d_cal, which, r2max / (double)(SL3D_RAD_NODES - 1), out);
    return (int)hipGetLastError();
}

int launch_proj_table(const DevCal *d_cal, int PW, int PH, float2 *out, void *stream)
{
    hipLaunchKernelGGL(k_proj_table, dim3((PW + 255) / 256, PH), dim3(256), 0, (hipStream_t)stream, d_cal, PW, PH, out);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// N4: cvUndistort2 (2/project_pattern.cpp:220,232,...), OpenCV 2.4.0's algorithm (parity unpinned):
// k_undist_map: one lane per image row.  The map of a row is a RECURRENCE along the row in OpenCV ((_x,_y,_w) advance
//   by (ir0,ir3,ir6) per column, accumulated in double), so a row is walked sequentially to round exactly as it does;
//   rows are independent (each restarts from i*ir1+ir2 ... of its stripe, whose matrix has cy - y0).
// k_undist_remap: one lane per pixel, INTER_LINEAR in OpenCV's fixed point (1/32-pixel positions, weights summing to
//   2^15), BORDER_CONSTANT 0.
// This file is compiled with -ffp-contract=off and IEEE division, so the doubles round as on the host.
// ------------------------------------------------------------------------------------------------
struct UndistParams {
    double K[9], d[5];
    int width, height, cn, stripe;
};

__global__ __launch_bounds__(64) void k_undist_map(UndistParams U, short *__restrict__ m1, unsigned short *__restrict__ m2)
{
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= U.height) return;
    const int y0 = (row / U.stripe) * U.stripe, i = row - y0;
    // iR = inverse of K with cy - y0: closed-form 3x3 (adjugate / determinant), the expressions of cvInvert's n == 3 case
    double S[9];
    for (int k = 0; k < 9; k++) S[k] = U.K[k];
    S[5] = U.K[5] - y0;
#define Sd(a, b) S[(a) * 3 + (b)]
    double det = Sd(0, 0) * (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) - Sd(0, 1) * (Sd(1, 0) * Sd(2, 2) - Sd(1, 2) * Sd(2, 0)) +
                 Sd(0, 2) * (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0));
    double ir[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (det != 0.) {
        det = 1. / det;
        ir[0] = (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) * det;
        ir[1] = (Sd(0, 2) * Sd(2, 1) - Sd(0, 1) * Sd(2, 2)) * det;
        ir[2] = (Sd(0, 1) * Sd(1, 2) - Sd(0, 2) * Sd(1, 1)) * det;
        ir[3] = (Sd(1, 2) * Sd(2, 0) - Sd(1, 0) * Sd(2, 2)) * det;
        ir[4] = (Sd(0, 0) * Sd(2, 2) - Sd(0, 2) * Sd(2, 0)) * det;
        ir[5] = (Sd(0, 2) * Sd(1, 0) - Sd(0, 0) * Sd(1, 2)) * det;
        ir[6] = (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0)) * det;
        ir[7] = (Sd(0, 1) * Sd(2, 0) - Sd(0, 0) * Sd(2, 1)) * det;
        ir[8] = (Sd(0, 0) * Sd(1, 1) - Sd(0, 1) * Sd(1, 0)) * det;
    }
#undef Sd
    const double u0 = U.K[2], v0 = U.K[5], fx = U.K[0], fy = U.K[4];
    const double k1 = U.d[0], k2 = U.d[1], p1 = U.d[2], p2 = U.d[3], k3 = U.d[4];
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    short *r1 = m1 + (size_t)row * U.width * 2;
    unsigned short *r2_ = m2 + (size_t)row * U.width;
    for (int j = 0; j < U.width; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double w = 1. / _w, x = _x * w, y = _y * w;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((0. * r2 + 0.) * r2 + 0.) * r2);
        const double u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0;
        const double v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0;
        const int iu = __double2int_rn(u * 32), iv = __double2int_rn(v * 32);  // cvRound
        r1[j * 2] = (short)(iu >> 5);
        r1[j * 2 + 1] = (short)(iv >> 5);
        r2_[j] = (unsigned short)((iv & 31) * 32 + (iu & 31));
    }
}

__global__ __launch_bounds__(256) void k_undist_remap(const uint8_t *__restrict__ src, size_t sstride, UndistParams U, const short *__restrict__ m1,
                                                      const unsigned short *__restrict__ m2, uint8_t *__restrict__ dst, size_t dstride)
{
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dx >= U.width) return;
    const size_t o = (size_t)dy * U.width + dx;
    const int sx = m1[2 * o], sy = m1[2 * o + 1], fxq = m2[o] & 31, fyq = (m2[o] >> 5) & 31;
    int w[4] = {(32 - fyq) * (32 - fxq) * 32, (32 - fyq) * fxq * 32, fyq * (32 - fxq) * 32, fyq * fxq * 32};
    if (w[0] == 32768) { w[0] = 32767; w[3] = 1; }  // saturate_cast<short>(32768) and the table's sum fix-up
    for (int k = 0; k < U.cn; k++) {
        int sum = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int xx = sx + (t & 1), yy = sy + (t >> 1);
            const int v = (xx >= 0 && xx < U.width && yy >= 0 && yy < U.height) ? src[(size_t)yy * sstride + (size_t)xx * U.cn + k] : 0;
            sum += v * w[t];
        }
        const int r = (sum + (1 << 14)) >> 15;
        dst[(size_t)dy * dstride + (size_t)dx * U.cn + k] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
    }
}

// the same remap for a stack of single-channel planes that share one map (the frames of a view): the map entry and the four
// weights are read / formed once per pixel, then every plane costs 4 taps and one byte
__global__ __launch_bounds__(256) void k_undist_remap_planes(const uint8_t *__restrict__ src, size_t spitch, size_t splane, int width, int height,
                                                             int n_planes, const short *__restrict__ m1, const unsigned short *__restrict__ m2,
                                                             uint8_t *__restrict__ dst, size_t dpitch, size_t dplane)
{
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dx >= width) return;
    const size_t o = (size_t)dy * width + dx;
    const int sx = m1[2 * o], sy = m1[2 * o + 1], fxq = m2[o] & 31, fyq = (m2[o] >> 5) & 31;
    int w[4] = {(32 - fyq) * (32 - fxq) * 32, (32 - fyq) * fxq * 32, fyq * (32 - fxq) * 32, fyq * fxq * 32};
    if (w[0] == 32768) { w[0] = 32767; w[3] = 1; }
    size_t off[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int xx = sx + (t & 1), yy = sy + (t >> 1);
        const bool in = xx >= 0 && xx < width && yy >= 0 && yy < height;
        off[t] = in ? (size_t)yy * spitch + (size_t)xx : 0;
        if (!in) w[t] = 0;  // BORDER_CONSTANT 0: the tap contributes nothing
    }
    for (int p = 0; p < n_planes; p++) {
        const uint8_t *sp = src + (size_t)p * splane;
        const int sum = sp[off[0]] * w[0] + sp[off[1]] * w[1] + sp[off[2]] * w[2] + sp[off[3]] * w[3];
        const int r = (sum + (1 << 14)) >> 15;
        dst[(size_t)p * dplane + (size_t)dy * dpitch + dx] = (uint8_t)(r > 255 ? 255 : r);
    }
}

int launch_undistort_planes(const uint8_t *src, size_t spitch, size_t splane, uint8_t *dst, size_t dpitch, size_t dplane, int width, int height,
                            int n_planes, const double K[9], const double dist[5], short *m1, unsigned short *m2, bool build_map, void *stream)
{
    UndistParams U;
    for (int k = 0; k < 9; k++) U.K[k] = K[k];
    for (int k = 0; k < 5; k++) U.d[k] = dist[k];
    U.width = width; U.height = height; U.cn = 1;
    int stripe = 4096 / (width > 1 ? width : 1);
    U.stripe = stripe < 1 ? 1 : (stripe > height ? height : stripe);
    hipStream_t st = (hipStream_t)stream;
    if (build_map) hipLaunchKernelGGL(k_undist_map, dim3((height + 63) / 64), dim3(64), 0, st, U, m1, m2);
    hipLaunchKernelGGL(k_undist_remap_planes, dim3((width + 255) / 256, height), dim3(256), 0, st, src, spitch, splane, width, height, n_planes, m1, m2,
                       dst, dpitch, dplane);
    return (int)hipGetLastError();
}

// build_map = false: m1 / m2 already hold the map of this (K, dist, width, height) -- every frame of a scan shares it
int launch_undistort(const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride, int width, int height, int cn, const double K[9],
                     const double dist[5], short *m1, unsigned short *m2, bool build_map, void *stream)
{
    UndistParams U;
    for (int k = 0; k < 9; k++) U.K[k] = K[k];
    for (int k = 0; k < 5; k++) U.d[k] = dist[k];
    U.width = width; U.height = height; U.cn = cn;
    int stripe = 4096 / (width > 1 ? width : 1);
    U.stripe = stripe < 1 ? 1 : (stripe > height ? height : stripe);
    hipStream_t st = (hipStream_t)stream;
    if (build_map) hipLaunchKernelGGL(k_undist_map, dim3((height + 63) / 64), dim3(64), 0, st, U, m1, m2);
    hipLaunchKernelGGL(k_undist_remap, dim3((width + 255) / 256, height), dim3(256), 0, st, src, sstride, U, m1, m2, dst, dstride);
    return (int)hipGetLastError();
}

// N1: one projector pattern (1/pattern_generator.cpp).  Every pattern is constant along one axis, so the host evaluates
// the reference's expression once per column (or row) with the libm the reference calls (cosf, pow: sl3d_generate_pattern)
// and this kernel replicates the profile: 16 bytes per lane, write-only, HBM bound (PW*PH bytes per pattern).
__global__ __launch_bounds__(256) void k_pattern(uint8_t *__restrict__ dst, size_t pitch, int PW, int PH, int axis,
                                                 const uint8_t *__restrict__ profile)
{
    const int c16 = blockIdx.x * blockDim.x + threadIdx.x;  // 16-byte column group
    const int r = blockIdx.y;
    if (c16 * 16 >= PW) return;
    uint4 v;
    if (axis == 0) {
        v = *(const uint4 *)(profile + (size_t)c16 * 16);  // the profile buffer is padded to a multiple of 16
    } else {
        const unsigned b = profile[r] * 0x01010101u;
        v = make_uint4(b, b, b, b);
    }
    *(uint4 *)(dst + (size_t)r * pitch + (size_t)c16 * 16) = v;  // pitch is a multiple of 16: the padding takes the spill-over
}

int launch_pattern(uint8_t *dst, size_t pitch, int PW, int PH, int axis, const uint8_t *profile, void *stream)
{
    const int groups = (PW + 15) / 16;
    hipLaunchKernelGGL(k_pattern, dim3((groups + 255) / 256, PH), dim3(256), 0, (hipStream_t)stream, dst, pitch, PW, PH, axis, profile);
    return (int)hipGetLastError();
}

// Exhaustive self-check of atan2_lattice / shift_pi against the host-libm table (see sl3d_create)
__global__ __launch_bounds__(256) void k_atan_selfcheck(const float *tab_phi, const float *tab_shift, unsigned *mismatches)
{
    __shared__ double s_rcp[SL3D_RCP_TAB];
    fill_rcp_table(s_rcp);
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= SL3D_ATAN_T1 * SL3D_ATAN_T2) return;
    const int t1 = i / SL3D_ATAN_T2 - 255, t2 = i % SL3D_ATAN_T2 - 510;
    // both reciprocal sources (LDS table: fused kernel; rcp + Newton: per-stage kernel) must reproduce the table
    const float phi = atan2_lattice<true>(t1, t2, s_rcp, atan_consts<true>()), phi2 = atan2_lattice<false>(t1, t2, nullptr, atan_consts<false>());
    const float sh = shift_pi(phi);
    // bit comparison: also catches a wrong sign of zero
    if (__float_as_uint(phi) != __float_as_uint(tab_phi[i]) || __float_as_uint(phi2) != __float_as_uint(tab_phi[i]) ||
        __float_as_uint(sh) != __float_as_uint(tab_shift[i]) || __float_as_uint(shift_pi_if(phi, true)) != __float_as_uint(tab_shift[i]) ||
        __float_as_uint(shift_pi_if(phi, false)) != __float_as_uint(phi))
        atomicAdd(mismatches, 1u);
}

int launch_atan_selfcheck(const float *tab_phi, const float *tab_shift, unsigned *mismatches, void *stream)
{
    const int n = SL3D_ATAN_T1 * SL3D_ATAN_T2;
    hipLaunchKernelGGL(k_atan_selfcheck, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, tab_phi, tab_shift, mismatches);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// O1 / N2: compaction of the dense cloud in the reference's row-major scan order
// (8/save_point_cloud.cpp:33-37 counts the valid pixels, :85-104 appends them).  Three launches on the
// context's stream: per-block counts (wave ballots), an exclusive scan of the block counts by one block,
// and the scatter.  A block covers 1024 consecutive pixels of the pitch-padded plane; padding pixels are
// never valid, so the scan order of the valid pixels is exactly the reference's.
// ------------------------------------------------------------------------------------------------
// blockIdx.y = view of a batch (strides in elements; 0 strides for a single view)
__global__ __launch_bounds__(256) void k_compact_count(const uint8_t *valid, size_t n_px, unsigned *block_counts, size_t valid_stride, int nb)
{
    valid += (size_t)blockIdx.y * valid_stride;
    block_counts += (size_t)blockIdx.y * nb;
    __shared__ unsigned s_cnt[4];
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x * 4;
    unsigned w = base < n_px ? *(const unsigned *)(valid + base) : 0u;  // 4 valid bytes (0/1)
    unsigned c = __popc(w & 0x01010101u);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// exclusive scan of n counts by a single 1024-thread block (n is a few thousand .. tens of thousands)
__global__ __launch_bounds__(1024) void k_compact_scan(const unsigned *counts, unsigned long long *offsets, int n, unsigned long long *total)
{
    counts += (size_t)blockIdx.x * n;   // one block per view of a batch
    offsets += (size_t)blockIdx.x * n;
    total += blockIdx.x;
    __shared__ unsigned long long s_part[1024];
    const int per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, n);
    unsigned long long sum = 0;
    for (int i = lo; i < hi; i++) sum += counts[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
        unsigned long long v = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned long long run = threadIdx.x == 0 ? 0 : s_part[threadIdx.x - 1];
    for (int i = lo; i < hi; i++) { offsets[i] = run; run += counts[i]; }
    if (threadIdx.x == 1023) *total = s_part[1023];
}

// texture (may be NULL): the BGR camera image save_point_cloud() colours the cloud with (8/save_point_cloud.cpp:46-52,
// 70-72), [row][pitch][3] bytes; rgb_out receives r,g,b per compacted point
__global__ __launch_bounds__(256) void k_compact_scatter(const uint8_t *valid, const float *points, size_t n_px,
                                                         const unsigned long long *block_offsets, float *cloud, const uint8_t *texture,
                                                         uint8_t *rgb_out, size_t view_stride, int nb)
{
    valid += (size_t)blockIdx.y * view_stride;
    points += 3 * (size_t)blockIdx.y * view_stride;
    cloud += 3 * (size_t)blockIdx.y * view_stride;
    block_offsets += (size_t)blockIdx.y * nb;
    __shared__ unsigned s_wave[4];
    __shared__ __attribute__((aligned(16))) float s_pts[1024 * 3];  // the block's valid points, compacted
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x * 4;
    const unsigned w = base < n_px ? (*(const unsigned *)(valid + base) & 0x01010101u) : 0u;
    const unsigned c = __popc(w);
    // exclusive prefix of c over the block: wave scan by shuffles, then the 4 wave totals
    unsigned incl = c;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off, 64);
        if ((threadIdx.x & 63) >= off) incl += t;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned wave_base = 0;
    for (int i = 0; i < (int)(threadIdx.x >> 6); i++) wave_base += s_wave[i];
    const unsigned block_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    unsigned local = wave_base + (incl - c);
    const unsigned long long block_off = block_offsets[blockIdx.x];
    if (w) {
        // the 4 pixels of a lane are 48 contiguous bytes: three 16-B loads, then the valid ones go to LDS in scan order
        const float4 *p4 = (const float4 *)(points + 3 * base);
        const float4 a = p4[0], b = p4[1], d = p4[2];
        const float q[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((w >> (8 * k)) & 1u) {
                s_pts[3 * local + 0] = q[3 * k + 0];
                s_pts[3 * local + 1] = q[3 * k + 1];
                s_pts[3 * local + 2] = q[3 * k + 2];
                if (texture) {
                    const uint8_t *t = texture + 3 * (base + k);  // b, g, r
                    uint8_t *o = rgb_out + 3 * (block_off + local);
                    o[0] = t[2]; o[1] = t[1]; o[2] = t[0];
                }
                local++;
            }
    }
    __syncthreads();
    // the block's segment of the cloud is contiguous: coalesced dword stores
    float *dst = cloud + 3 * block_off;
    for (unsigned i = threadIdx.x; i < 3 * block_total; i += 256) dst[i] = s_pts[i];
}

int launch_compact(const KParams &P, int view, unsigned *block_counts, unsigned long long *block_offsets, unsigned long long *total,
                   float *cloud, const uint8_t *texture, uint8_t *rgb_out, void *stream)
{
    const size_t n_px = P.px_view_stride;
    const int nb = (int)((n_px + 1023) / 1024);
    const uint8_t *valid = P.valid + (size_t)view * P.px_view_stride;
    const float *points = P.points + 3 * (size_t)view * P.px_view_stride;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(256), 0, st, valid, n_px, block_counts, (size_t)0, nb);
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, st, block_counts, block_offsets, nb, total);
    hipLaunchKernelGGL(k_compact_scatter, dim3(nb), dim3(256), 0, st, valid, points, n_px, block_offsets, cloud, texture, rgb_out, (size_t)0, nb);
    return (int)hipGetLastError();
}

// the same three kernels over a batch of views: view v's compacted cloud starts at clouds + 3*v*px_view_stride
int launch_compact_views(const KParams &P, int first_view, int n_views, unsigned *block_counts, unsigned long long *block_offsets,
                         unsigned long long *totals, float *clouds, void *stream)
{
    const size_t n_px = P.px_view_stride;
    const int nb = (int)((n_px + 1023) / 1024);
    const uint8_t *valid = P.valid + (size_t)first_view * P.px_view_stride;
    const float *points = P.points + 3 * (size_t)first_view * P.px_view_stride;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_compact_count, dim3(nb, n_views), dim3(256), 0, st, valid, n_px, block_counts, n_px, nb);
    hipLaunchKernelGGL(k_compact_scan, dim3(n_views), dim3(1024), 0, st, block_counts, block_offsets, nb, totals);
    hipLaunchKernelGGL(k_compact_scatter, dim3(nb, n_views), dim3(256), 0, st, valid, points, n_px, block_offsets, clouds, (const uint8_t *)nullptr,
                       (uint8_t *)nullptr, n_px, nb);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Consumers of the SEGMENTED clouds the fused kernel writes (k_fused<..., CMODE = 2>): a view's cloud is the concatenation of
// its segments' first `count` points.  One wave per segment, one point (12 bytes) per lane and step.
// ------------------------------------------------------------------------------------------------
// Exclusive scan of one view's segment counts (1024-thread blocks), behind every segmented launch: it is on the critical path of
// sl3d_run_clouds, so it is written for latency.  A view's segments are scanned by SL3D_SCAN_PARTS blocks (one block per view was a
// latency chain of ~10 us on a 256-CU machine): block (view, part) SUMS the counts of the parts in front of it -- the same
// coalesced reads every one of them does anyway, at most n dwords from the L2 -- and scans its own part from that carry; no block
// waits for another.  The last part's block leaves the view's total in the mapped host word sl3d_get_cloud_counts reads.
// ONE memory round trip per block (round 4): it requests its own counts -- 4 consecutive ones per thread, one 16-byte load, a wave
// reads 1 KB contiguous -- AND the counts in front of its part in the same breath, sums the latter, scans the former in registers
// (the 4 entries, 6 wave shuffles, the 16 wave totals through LDS: one block barrier per chunk) and writes 4 offsets per thread
// (32 contiguous bytes); the next chunk of a long part travels while the current one is scanned.  (Until then: the carry first,
// then the chunk through a padded LDS array with five barriers -- 6.1 us for the 8,100 counts of one 1080p view;
// profiles/r04_seg_scan_ab.txt.  Rounds 2-3: k_compact_scan, 32 strided dwords per thread, 14.3 us for 16 x 32,400 counts; one
// block per view with runs of 32: 10.2 us.)
#define SL3D_SCAN_RUN 4
#define SL3D_SCAN_PARTS 8
__global__ __launch_bounds__(1024) void k_seg_scan(const unsigned *__restrict__ counts, unsigned long long *__restrict__ offsets, int n,
                                                   unsigned long long *total)
{
    const int view = (int)blockIdx.y, part = (int)blockIdx.x;
    counts += (size_t)view * n;   // (n = 4 * tiles: every row of counts is 16-byte aligned, every row of offsets 32-byte aligned)
    offsets += (size_t)view * n;
    constexpr int CHUNK = 1024 * SL3D_SCAN_RUN;
    // parts are whole chunks, so that every chunk of a part is scanned by the same code path
    const int part_len = (((n + SL3D_SCAN_PARTS - 1) / SL3D_SCAN_PARTS + CHUNK - 1) / CHUNK) * CHUNK;
    const int begin = min(part * part_len, n), end = min(begin + part_len, n);
    __shared__ unsigned long long s_part[16];
    __shared__ unsigned s_wave[2][16];
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint4 zero = {0u, 0u, 0u, 0u};
    auto mine = [&](int base) { return base + SL3D_SCAN_RUN * t < end ? *(const uint4 *)(counts + base + SL3D_SCAN_RUN * t) : zero; };
    uint4 c = mine(begin);
    unsigned long long carry;
    {   // the carry into this part: the sum of everything in front of it (requested together with the part's first chunk)
        unsigned long long acc = 0ull;
        for (int i = SL3D_SCAN_RUN * t; i < begin; i += CHUNK) {
            const uint4 f = *(const uint4 *)(counts + i);
            acc += (unsigned long long)f.x + f.y + f.z + f.w;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) s_part[wave] = acc;
        __syncthreads();
        carry = 0ull;
#pragma unroll
        for (int w = 0; w < 16; w++) carry += s_part[w];
    }
    int buf = 0;
    for (int base = begin;; base += CHUNK, buf ^= 1) {
        const uint4 next = base + CHUNK < end ? mine(base + CHUNK) : zero;  // (the next chunk travels while this one is scanned)
        const unsigned e1 = c.x, e2 = e1 + c.y, e3 = e2 + c.z, run = e3 + c.w;
        unsigned incl = run;  // inclusive scan of the run totals over the wave, then over the 16 waves
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) s_wave[buf][wave] = incl;
        __syncthreads();  // (two buffers: the next chunk's totals do not overwrite what a slower wave still reads)
        unsigned wbase = 0, chunk_total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const unsigned v = s_wave[buf][w];
            wbase += w < wave ? v : 0u;
            chunk_total += v;
        }
        if (base + SL3D_SCAN_RUN * t < end) {
            const unsigned long long o = carry + (unsigned long long)(wbase + (incl - run));
            ulonglong2 *dst = (ulonglong2 *)(offsets + base + SL3D_SCAN_RUN * t);
            dst[0] = make_ulonglong2(o, o + e1);
            dst[1] = make_ulonglong2(o + e2, o + e3);
        }
        carry += (unsigned long long)chunk_total;
        if (base + CHUNK >= end) break;
        c = next;
    }
    if (t == 0 && part == SL3D_SCAN_PARTS - 1) total[view] = carry;
}

int launch_seg_scan(const KParams &P, int first_view, int n_views, void *stream)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_seg_scan, dim3(SL3D_SCAN_PARTS, (unsigned)n_views), dim3(1024), 0, (hipStream_t)stream,
                       P.seg_counts + (size_t)first_view * P.n_segs, P.seg_offsets + (size_t)first_view * P.n_segs, P.n_segs,
                       P.cloud_totals + first_view);
    return (int)hipGetLastError();
}

// REG = false: plain copy (closing the gaps); true: the rigid transform of k_register on the way (9/register_point_clouds.cpp:109-117)
// SCAN = false: the segments' offsets come from k_seg_scan.  true: the consumer scans on entry -- a block (4 segments) adds up the
// counts in front of it itself (at most n_segs dwords from the L2, 16 bytes per lane and step: what every part of k_seg_scan does
// for its carry), so a launch of a few views needs NO scan launch between the fused kernel and its consumer; the last block leaves
// the view's total in total_out[view] (host memory mapped into the device: sl3d_get_cloud_counts' word).  Points whose index in the
// closed cloud is >= capacity are not written (a destination smaller than the cloud takes its first `capacity` points).
template <bool REG, bool SCAN>
__global__ __launch_bounds__(256) void k_seg_close(const float *__restrict__ seg_xyz, const unsigned *__restrict__ counts,
                                                   const unsigned long long *__restrict__ offsets, int n_segs, size_t src_view_stride, float *dst,
                                                   size_t dst_view_stride, float r00, float r02, float r20, float r22, float tx, float ty, float tz,
                                                   unsigned long long *total_out, unsigned long long capacity)
{
    typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));
    const int wave = (int)(threadIdx.x >> 6);
    const int seg = blockIdx.x * 4 + wave, lane = (int)(threadIdx.x & 63u), v = blockIdx.y;
    const unsigned cnt = seg < n_segs ? counts[(size_t)v * n_segs + seg] : 0u;
    unsigned long long off;
    if (SCAN) {
        __shared__ unsigned long long s_front[4];
        __shared__ unsigned s_cnt[4];
        const unsigned *cv = counts + (size_t)v * n_segs;
        const int n_front = (int)blockIdx.x * 4;  // (whole 16-byte groups: n_segs = 4 * tiles, every row of counts is 16-byte aligned)
        unsigned long long acc = 0ull;
        for (int i = 4 * (int)threadIdx.x; i < n_front; i += 1024) {
            const uint4 f = *(const uint4 *)(cv + i);
            acc += (unsigned long long)f.x + f.y + f.z + f.w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) {
            s_front[wave] = acc;
            s_cnt[wave] = cnt;
        }
        __syncthreads();
        off = s_front[0] + s_front[1] + s_front[2] + s_front[3];
#pragma unroll
        for (int w = 0; w < 4; w++) off += w < wave ? s_cnt[w] : 0u;
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 192) total_out[v] = off + cnt;  // (wave 3 of the last block: everything in front + its own)
    } else {
        if (seg >= n_segs) return;
        off = offsets[(size_t)v * n_segs + seg];
    }
    // (a 3-float vector type is PADDED to 16 bytes: points are addressed through float pointers, 12 bytes apart)
    const float *src = seg_xyz + 3 * ((size_t)v * src_view_stride + (size_t)seg * SL3D_SEG_POINTS);
    float *out = dst + 3 * ((size_t)v * dst_view_stride + (size_t)off);
    const unsigned long long room = off < capacity ? capacity - off : 0ull;
    const unsigned n = SCAN ? (unsigned)(room < cnt ? room : cnt) : cnt;
    for (unsigned i = (unsigned)lane; i < n; i += 64u) {
        f32x3 p = *(const f32x3 *)(src + 3 * (size_t)i);
        if (REG) {
            const float x = p.x - tx, y = p.y - ty, z = p.z - tz;
            const float X = (float)(((double)r00 * (double)x + 0.0 * (double)y) + (double)r02 * (double)z);
            const float Y = (float)((0.0 * (double)x + 1.0 * (double)y) + 0.0 * (double)z);
            const float Z = (float)(((double)r20 * (double)x + 0.0 * (double)y) + (double)r22 * (double)z);
            p.x = X + tx; p.y = Y + ty; p.z = Z + tz;
        }
        *(f32x3 *)(out + 3 * (size_t)i) = p;
    }
}

int launch_seg_close(const KParams &P, int first_view, int n_views, float *dst, size_t dst_view_stride_points, void *stream)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_seg_close<false, false>), dim3((unsigned)((P.n_segs + 3) / 4), (unsigned)n_views), dim3(256), 0, (hipStream_t)stream,
                       P.clouds + 3 * (size_t)first_view * P.px_view_stride, P.seg_counts + (size_t)first_view * P.n_segs,
                       P.seg_offsets + (size_t)first_view * P.n_segs, P.n_segs, P.px_view_stride, dst, dst_view_stride_points, 0.f, 0.f, 0.f, 0.f, 0.f,
                       0.f, 0.f, (unsigned long long *)nullptr, ~0ull);
    return (int)hipGetLastError();
}

// the same for views whose segment counts have NOT been scanned (sl3d_run_clouds over a few views leaves the scan to its consumer):
// view first_view + k to dst + 3 * k * dst_view_stride_points, at most capacity_points points of each; the views' totals go to
// P.cloud_totals
int launch_seg_close_scan(const KParams &P, int first_view, int n_views, float *dst, size_t dst_view_stride_points, unsigned long long capacity_points,
                          void *stream)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_seg_close<false, true>), dim3((unsigned)((P.n_segs + 3) / 4), (unsigned)n_views), dim3(256), 0, (hipStream_t)stream,
                       P.clouds + 3 * (size_t)first_view * P.px_view_stride, P.seg_counts + (size_t)first_view * P.n_segs,
                       (const unsigned long long *)nullptr, P.n_segs, P.px_view_stride, dst, dst_view_stride_points, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                       P.cloud_totals + first_view, capacity_points);
    return (int)hipGetLastError();
}

int launch_seg_register(const KParams &P, int view, float *out, const float R4[4], float tx, float ty, float tz, void *stream)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_seg_close<true, false>), dim3((unsigned)((P.n_segs + 3) / 4), 1u), dim3(256), 0, (hipStream_t)stream,
                       P.clouds + 3 * (size_t)view * P.px_view_stride, P.seg_counts + (size_t)view * P.n_segs, P.seg_offsets + (size_t)view * P.n_segs,
                       P.n_segs, P.px_view_stride, out, (size_t)0, R4[0], R4[1], R4[2], R4[3], tx, ty, tz, (unsigned long long *)nullptr, ~0ull);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// N3: turntable registration, 9/register_point_clouds.cpp:83-128.  Per point, in the reference's types:
// p -= t (float), p = R*p with the float GEMM of cvMatMul (double accumulator, k ascending, rounded to float on
// store), p += t (float).  R = rotation about Y by theta (row 1 and the last column are the identity's).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_register(const float *in, float *out, long n, float r00, float r02, float r20, float r22,
                                                  float tx, float ty, float tz)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = in[3 * i + 0] - tx, y = in[3 * i + 1] - ty, z = in[3 * i + 2] - tz;  // :109-111
    // rows of R: (r00, 0, r02, 0), (0, 1, 0, 0), (r20, 0, r22, 0); the products with exact zeros add nothing
    const float X = (float)(((double)r00 * (double)x + 0.0 * (double)y) + (double)r02 * (double)z);  // :113
    const float Y = (float)((0.0 * (double)x + 1.0 * (double)y) + 0.0 * (double)z);
    const float Z = (float)(((double)r20 * (double)x + 0.0 * (double)y) + (double)r22 * (double)z);
    out[3 * i + 0] = X + tx;  // :115-117
    out[3 * i + 1] = Y + ty;
    out[3 * i + 2] = Z + tz;
}

int launch_register(const float *in, float *out, long n, const float R4[4], float tx, float ty, float tz, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_register, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, n, R4[0], R4[1], R4[2], R4[3], tx, ty, tz);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// N1: synthetic captures generated in place.  A plane Z = z0 + a*X + b*Y in the world frame is
// seen by the calibrated camera/projector pair; each camera pixel's ray (5-iteration undistortion, as stage 7
// applies it) is intersected with the plane and projected into the projector -> (xp, yp); the projected patterns
// follow the reference's generator: fringe k = 127 + 128*cosf((p/fw)*2*Pi - Pi - Pi/2 + k*Pi/2), Pi = 22/7
// (1/pattern_generator.cpp:302,313), Gray bit i of floor(p/fw), MSB first, x255 (:80-105), inverse = 255 - pattern
// (:497); then gain, offset and counter-hash noise.  Host twin: 3dscan_amd/synth.py (same formulas; the trig
// functions differ in the last ulp, so a few bytes per million differ by one grey level).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ unsigned synth_camera(unsigned I, bool lit, const SynthParams &S, unsigned long long key, int gx, int gy)
{
    int nz = 0;
    if (S.noise > 0) {
        const unsigned long long idx = ((unsigned long long)gy << 20) + (unsigned long long)gx;
        nz = (int)(splitmix64(idx ^ key) % (unsigned long long)(2 * S.noise + 1)) - S.noise;
    }
    const float lin = (lit ? S.gain * (float)I : 0.0f) + S.offset;
    const double v = floor((double)lin + (double)nz + 0.5);
    return (unsigned)fmin(fmax(v, 0.0), 255.0);
}

__global__ __launch_bounds__(256) void k_synth(const KParams P, const DevCal C, const SynthParams S, int view)
{
    const int qpr = P.pitch >> 2;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(q / qpr), cq = (int)(q - (long)row * qpr);
    if (row >= P.H) return;
    uint8_t *vb = (uint8_t *)P.frames + (size_t)view * P.view_stride + (size_t)row * P.pitch + (size_t)cq * 4;
    const int gy = P.row0 + row;
    double xp[4], yp[4];
    bool lit[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int gx = P.col0 + cq * 4 + k;
        // normalised undistorted ray of the camera pixel (the first half of undistort_reproject)
        const Intr &I = C.cam;
        const double x0 = ((double)gx - I.cx) * I.ifx, y0 = ((double)gy - I.cy) * I.ify;
        double x = x0, y = y0;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icd = 1.0 / (1.0 + ((I.k3 * r2 + I.k2) * r2 + I.k1) * r2);
            const double dx = 2.0 * I.p1 * x * y + I.p2 * (r2 + 2.0 * x * x), dy = I.p1 * (r2 + 2.0 * y * y) + 2.0 * I.p2 * x * y;
            x = (x0 - dx) * icd;
            y = (y0 - dy) * icd;
        }
        // world ray: origin ow = -Rc^T tc, direction dw = Rc^T (x,y,1); plane normal n = (-a,-b,1), n.X = z0
        double dw[3], ow[3];
        for (int i = 0; i < 3; i++) {
            dw[i] = S.Rc[0 * 3 + i] * x + S.Rc[1 * 3 + i] * y + S.Rc[2 * 3 + i];
            ow[i] = -(S.Rc[0 * 3 + i] * S.tc[0] + S.Rc[1 * 3 + i] * S.tc[1] + S.Rc[2 * 3 + i] * S.tc[2]);
        }
        const double lam = (S.z0 - (-S.a * ow[0] - S.b * ow[1] + ow[2])) / (-S.a * dw[0] - S.b * dw[1] + dw[2]);
        double Xw[3], Xq[3];
        for (int i = 0; i < 3; i++) Xw[i] = ow[i] + lam * dw[i];
        for (int i = 0; i < 3; i++) Xq[i] = S.Rp[i * 3 + 0] * Xw[0] + S.Rp[i * 3 + 1] * Xw[1] + S.Rp[i * 3 + 2] * Xw[2] + S.tp[i];
        xp[k] = S.Kp[0] * Xq[0] / Xq[2] + S.Kp[2];
        yp[k] = S.Kp[4] * Xq[1] / Xq[2] + S.Kp[5];
        lit[k] = xp[k] >= 0.0 && xp[k] < (double)P.PW && yp[k] >= 0.0 && yp[k] < (double)P.PH;
    }
    int plane = 0;
#pragma unroll 1
    for (int axis = 0; axis < 2; axis++) {
        const int N = axis == 0 ? P.Nv : P.Nh, fw = axis == 0 ? P.fwv : P.fwh;
        float pf[4];
        int gray[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            pf[k] = (float)(axis == 0 ? xp[k] : yp[k]);
            int code = (int)floorf(pf[k] / (float)fw);
            code = min(max(code, 0), (1 << N) - 1);
            gray[k] = code ^ (code >> 1);
        }
        int fidx = axis * 1000;  // frame index inside the noise key, as in synth.py
        for (int fr = 0; fr < P.F + 2 * N; fr++, fidx++, plane++) {
            const unsigned long long key =
                splitmix64(S.seed * 0x100000001B3ull + (unsigned long long)S.view_id * 1000003ull + (unsigned long long)fidx);
            unsigned word = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                unsigned I;
                if (fr < P.F) {
                    const double arg = (double)(pf[k] / (float)fw) * 2.0 * PI_REF - PI_REF - ((PI_REF) / 2.0) + (PI_REF / 2.0) * (double)fr;
                    const float t = 127.0f + 128.0f * cosf((float)arg);
                    I = (unsigned)fminf(fmaxf(t, 0.0f), 255.0f);
                } else {
                    const int i = (fr - P.F) % N;
                    const unsigned bit = (unsigned)(gray[k] >> (N - 1 - i)) & 1u;
                    I = (fr - P.F) < N ? bit * 255u : 255u - bit * 255u;
                }
                word |= synth_camera(I, lit[k], S, key, P.col0 + cq * 4 + k, gy) << (8 * k);
            }
            *(unsigned *)(vb + (size_t)plane * P.plane_stride) = word;
        }
    }
}

int launch_synth(const KParams &P, const DevCal &C, const SynthParams &S, int view, void *stream)
{
    const long quads = (long)(P.pitch >> 2) * P.H;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, C, S, view);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The reference's global arrays in the reference's OWN layout: every image-shaped global is indexed [col][row]
// (PROJECT_GLOBAL/common_variables.h:12-21,56-62), i.e. the transpose of the row-major planes the kernels write.  The transpose
// is done HERE, through LDS tiles (32 x 32 elements: reads coalesced along the row of the source, writes coalesced along the
// column-major destination), with the element conversion the reference's types ask for (valid bytes -> int), so that a global
// reaches the caller as ONE contiguous device-to-host copy instead of a strided pass of the host over every plane
// (3/wrapped_phase.cpp:165-175 is that access pattern, and SURVEY blames it for the reference's own slowness).
// ------------------------------------------------------------------------------------------------
template <typename TI, typename TO, int C>
__global__ __launch_bounds__(256) void k_to_colrow(const TI *__restrict__ src, TO *__restrict__ dst, int W, int H, int pitch)
{
    __shared__ TO tile[32][32 * C + 1];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        if (r < H && c < W) {
#pragma unroll
            for (int k = 0; k < C; k++) tile[j][tx * C + k] = (TO)src[((size_t)r * pitch + c) * C + k];
        }
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = bx + j, r = by + tx;
        if (c < W && r < H) {
#pragma unroll
            for (int k = 0; k < C; k++) dst[((size_t)c * H + r) * C + k] = tile[tx][j * C + k];
        }
    }
}

// which: 0..2 valid maps (vertical, horizontal, merged) -> int; 3,4 wrapped; 5,6 unwrapped -> float; 7,8 code -> int;
// 9 intersection_points -> double[3]; 10 the dense f32 result widened -> double[3].  dst: [W][H] elements of the window.
int launch_to_colrow(const KParams &P, int view, int which, void *dst, void *stream)
{
    const dim3 grid((unsigned)((P.W + 31) / 32), (unsigned)((P.H + 31) / 32)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const size_t off = (size_t)view * P.px_view_stride;
    (void)hipGetLastError();
    switch (which) {
    case 0: case 1: hipLaunchKernelGGL((k_to_colrow<uint8_t, int, 1>), grid, block, 0, st, P.valid_axis[which] + off, (int *)dst, P.W, P.H, P.pitch); break;
    case 2: hipLaunchKernelGGL((k_to_colrow<uint8_t, int, 1>), grid, block, 0, st, P.valid + off, (int *)dst, P.W, P.H, P.pitch); break;
    case 3: case 4: hipLaunchKernelGGL((k_to_colrow<float, float, 1>), grid, block, 0, st, P.wrapped[which - 3] + off, (float *)dst, P.W, P.H, P.pitch); break;
    case 5: case 6: hipLaunchKernelGGL((k_to_colrow<float, float, 1>), grid, block, 0, st, P.unwrapped[which - 5] + off, (float *)dst, P.W, P.H, P.pitch); break;
    case 7: case 8: hipLaunchKernelGGL((k_to_colrow<int32_t, int, 1>), grid, block, 0, st, P.code[which - 7] + off, (int *)dst, P.W, P.H, P.pitch); break;
    case 9: hipLaunchKernelGGL((k_to_colrow<double, double, 3>), grid, block, 0, st, P.ipoints + 3 * off, (double *)dst, P.W, P.H, P.pitch); break;
    case 10: hipLaunchKernelGGL((k_to_colrow<float, double, 3>), grid, block, 0, st, P.points + 3 * off, (double *)dst, P.W, P.H, P.pitch); break;
    default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

// selected_region as the reference holds it -- int [col][row] (m_tech_project_console.cpp:146-238) -- into the byte staging plane
// k_mask_prepare reads: `sel` holds columns [gx0, gx0 + ncols) x rows [gy0, gy0 + nrows) of the frame, [col][row]; a pixel is
// selected iff its int == 1.  Tiled the other way round: reads coalesced along the rows of a column, writes along the row.
__global__ __launch_bounds__(256) void k_mask_from_colrow(const KParams P, const int *__restrict__ sel, int gx0, int gy0, int ncols, int nrows,
                                                          uint8_t *__restrict__ raw)
{
    __shared__ uint8_t tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int c = bx + j, r = by + tx;
        if (c < ncols && r < nrows) tile[j][tx] = sel[(size_t)c * nrows + r] == 1 ? 1 : 0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        if (c < ncols && r < nrows)
            raw[(size_t)(gy0 + r - P.row0 + SL3D_MASK_HALO) * P.mpitch + SL3D_MASK_LPAD + (gx0 + c - P.col0)] = tile[tx][j];
    }
}

int launch_mask_from_colrow(const KParams &P, const int *sel, int gx0, int gy0, int ncols, int nrows, uint8_t *raw, void *stream)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_mask_from_colrow, dim3((unsigned)((ncols + 31) / 32), (unsigned)((nrows + 31) / 32)), dim3(256), 0, (hipStream_t)stream, P, sel, gx0,
                       gy0, ncols, nrows, raw);
    return (int)hipGetLastError();
}

static dim3 px_grid(const KParams &P) { return dim3((unsigned)(((long)P.pitch * P.H + 255) / 256), 1, 1); }

int launch_wrap(const KParams &P, int view, int axis, void *stream)
{
    hipLaunchKernelGGL(k_wrap, px_grid(P), dim3(256), 0, (hipStream_t)stream, P, view, axis);
    return (int)hipGetLastError();
}
int launch_unwrap(const KParams &P, int view, int axis, void *stream)
{
    hipLaunchKernelGGL(k_unwrap, px_grid(P), dim3(256), 0, (hipStream_t)stream, P, view, axis);
    return (int)hipGetLastError();
}
int launch_corr(const KParams &P, int view, void *stream)
{
    hipLaunchKernelGGL(k_corr, px_grid(P), dim3(256), 0, (hipStream_t)stream, P, view);
    return (int)hipGetLastError();
}
int launch_tri(const KParams &P, const DevCal &C, int view, void *stream)
{
    hipLaunchKernelGGL(k_tri, px_grid(P), dim3(256), 0, (hipStream_t)stream, P, C, view);
    return (int)hipGetLastError();
}

}  // namespace sl3d
