// sl3d_kernels.hip -- gfx950 (MI355X, wave64) kernels for the structured-light hot path.
//
//   k_fused : stages 3(v) 3(h) 4(v) 4(h) 5 7 + the f32 cast of stage 8 in one pass.  One lane owns
//             4 horizontally adjacent pixels = one dword of every 8-bit plane, so a wave reads
//             256 contiguous bytes of each of the 2F+2Nv+2Nh planes (coalesced) and writes
//             3 KiB of xyz + 256 B of valid.  No MFMA: this is a per-pixel map bounded by HBM
//             bandwidth and by fp64 VALU.
//   k_wrap / k_unwrap / k_corr / k_tri : the same arithmetic cut at the reference's stage
//             boundaries (parity mode; writes the planes the reference keeps in globals).
//
// Arithmetic contract:
//   * everything up to the correspondence (x,y) is BIT-EXACT with the reference's C expressions:
//     atan2 comes from a table built by the host with the double-precision libm atan2 the reference
//     calls (3/wrapped_phase.cpp:175); the +Pi, +code*2.0*Pi, /(2.0*Pi), *fw, lrint chain is
//     evaluated in fp64 with the reference's operation order and Pi = 22.0/7.0; this file is
//     compiled with -ffp-contract=off so no FMA is formed behind our back.
//   * stage 7 (fp64 4x3 least squares) only has to match within 1e-5; it uses explicit fma().
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "sl3d_internal.h"

namespace sl3d {

#define PI_REF 22.0 / 7.0 /* PROJECT_GLOBAL/global_cv.h:62: unparenthesised on purpose */

// ------------------------------------------------------------------------------------------------
// Selection mask -> valid map  (3/wrapped_phase.cpp:106-115 then :253-279 / :306-318)
//
// The reference's boundary removal scans the interior row-major; a pixel is cleared if any
// 8-neighbour is `!= 1 && !visited`, and every pixel that satisfies the test (valid or not) is
// marked visited.  That is NOT a symmetric erosion.  Writing V = selected, and for a pixel q
//   later(q)   = {E, SW, S, SE}   (scanned after q)      earlier(q) = {NW, N, NE, W}
//   L(q) = some later neighbour unselected
//   B(q) = some earlier neighbour lies on the frame border and is unselected (border pixels are
//          never scanned, hence never visited)
// the scan has the closed form (validated against the literal loop in tests/test_oracle.py):
//   interior p :  valid(p) = V(p) & !L(p) & AND_{n in earlier(p)} [ V(n) | (interior(n) & (L(n) | B(n))) ]
//   border   p :  valid(p) = V(p)
// because an unselected interior pixel with an unselected later neighbour is always visited.
// ------------------------------------------------------------------------------------------------
struct MaskView {
    const uint8_t *base;  // address of window pixel (0,0)
    int mpitch;
    int col0, row0, fullW, fullH;
    // V at frame coordinates; pixels outside the frame are never consulted for in-frame results
    __device__ __forceinline__ bool V(int gx, int gy) const
    {
        if (gx < 0 || gy < 0 || gx >= fullW || gy >= fullH) return false;
        return base[(ptrdiff_t)(gy - row0) * mpitch + (gx - col0)] == 1;
    }
    __device__ __forceinline__ bool interior(int gx, int gy) const
    {
        return gx >= 1 && gx <= fullW - 2 && gy >= 1 && gy <= fullH - 2;
    }
    __device__ __forceinline__ bool L(int gx, int gy) const
    {
        return !V(gx + 1, gy) || !V(gx - 1, gy + 1) || !V(gx, gy + 1) || !V(gx + 1, gy + 1);
    }
    __device__ __forceinline__ bool borderUnsel(int gx, int gy) const
    {
        if (gx < 0 || gy < 0 || gx >= fullW || gy >= fullH) return false;
        return !interior(gx, gy) && !V(gx, gy);
    }
    __device__ __forceinline__ bool B(int gx, int gy) const
    {
        return borderUnsel(gx - 1, gy - 1) || borderUnsel(gx, gy - 1) || borderUnsel(gx + 1, gy - 1) || borderUnsel(gx - 1, gy);
    }
    __device__ __forceinline__ bool OK(int gx, int gy) const
    {
        return V(gx, gy) || (interior(gx, gy) && (L(gx, gy) || B(gx, gy)));
    }
    // generic (any position) evaluation of the closed form
    __device__ bool valid(int gx, int gy) const
    {
        if (!V(gx, gy)) return false;
        if (!interior(gx, gy)) return true;
        if (L(gx, gy)) return false;
        return OK(gx - 1, gy - 1) && OK(gx, gy - 1) && OK(gx + 1, gy - 1) && OK(gx - 1, gy);
    }
};

__device__ __forceinline__ MaskView mask_view(const KParams &P, int view)
{
    MaskView m;
    m.base = P.mask + (size_t)view * P.mask_view_stride + (size_t)SL3D_MASK_HALO * P.mpitch + SL3D_MASK_LPAD;
    m.mpitch = P.mpitch;
    m.col0 = P.col0; m.row0 = P.row0; m.fullW = P.fullW; m.fullH = P.fullH;
    return m;
}

// Valid bits of the 4 pixels (cq*4 .. cq*4+3, row) of a window; bit k = pixel k.
// Fast path: 3 rows x 3 aligned dwords of 0/1 bytes, byte-parallel logic (all neighbours are
// interior pixels of the frame, so B == false and interior == true).  Pixels within 3 of the
// frame border take the generic path.
__device__ __forceinline__ unsigned quad_valid_bits(const KParams &P, const MaskView &mv, int cq, int row)
{
    const int c = cq * 4, gx = P.col0 + c, gy = P.row0 + row;
    unsigned bits = 0;
    if (gy >= 3 && gy <= P.fullH - 4 && gx >= 4 && gx + 3 <= P.fullW - 5) {
        const uint8_t *r1 = mv.base + (ptrdiff_t)row * mv.mpitch + c;  // row y, pixel c
        const uint8_t *r0 = r1 - mv.mpitch, *r2 = r1 + mv.mpitch;
        const unsigned bP = *(const unsigned *)(r0 - 4), bC = *(const unsigned *)r0, bN = *(const unsigned *)(r0 + 4);
        const unsigned cP = *(const unsigned *)(r1 - 4), cC = *(const unsigned *)r1, cN = *(const unsigned *)(r1 + 4);
        const unsigned dP = *(const unsigned *)(r2 - 4), dC = *(const unsigned *)r2, dN = *(const unsigned *)(r2 + 4);
        // X(d): bytes of row X at columns c+k+d, k=0..3
#define SHL2(Pw, Cw) __builtin_amdgcn_alignbyte(Cw, Pw, 2)
#define SHL1(Pw, Cw) __builtin_amdgcn_alignbyte(Cw, Pw, 3)
#define SHR1(Cw, Nw) __builtin_amdgcn_alignbyte(Nw, Cw, 1)
#define SHR2(Cw, Nw) __builtin_amdgcn_alignbyte(Nw, Cw, 2)
        const unsigned ONE = 0x01010101u;
        const unsigned Bm1 = SHL1(bP, bC), B0 = bC, B1 = SHR1(bC, bN), B2 = SHR2(bC, bN);
        const unsigned Cm2 = SHL2(cP, cC), Cm1 = SHL1(cP, cC), C0 = cC, C1 = SHR1(cC, cN), C2 = SHR2(cC, cN);
        const unsigned Dm2 = SHL2(dP, dC), Dm1 = SHL1(dP, dC), D0 = dC, D1 = SHR1(dC, dN);
        unsigned v = C0 & C1 & Dm1 & D0 & D1;                 // V(p) & !L(p)
        v &= Bm1 | ((B0 & Cm2 & Cm1) ^ ONE);                   // OK(NW) given the line above
        v &= B0 | ((B1 & Cm1) ^ ONE);                          // OK(N)
        v &= B1 | ((B2 & C2) ^ ONE);                           // OK(NE)
        v &= Cm1 | (Dm2 ^ ONE);                                // OK(W)
#undef SHL2
#undef SHL1
#undef SHR1
#undef SHR2
        bits = (v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u);
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (mv.valid(gx + k, gy)) bits |= 1u << k;
    }
    // pixels of the pitch padding are not part of the window
    const int inside = P.W - c;  // number of window pixels in this quad (may be <= 0 or >= 4)
    if (inside < 4) bits &= inside <= 0 ? 0u : ((1u << inside) - 1u);
    return bits;
}

// ------------------------------------------------------------------------------------------------
// bit-exact phase chain
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int atan_index(int t1, int t2) { return (t1 + 255) * SL3D_ATAN_T2 + (t2 + 510); }

// t1,t2 of create_wrapped_phase: 3-step 3/wrapped_phase.cpp:171-172, 4-step :195-196 (exact small integers)
__device__ __forceinline__ int fringe_index(int F, int i0, int i1, int i2, int i3)
{
    if (F == 3) return atan_index(i0 - i2, 2 * i1 - i0 - i2);
    return atan_index(i3 - i1, i0 - i2);
}

// unwrapped = wrapped(+Pi already applied) + code*2.0*Pi          4/phase_unwrap.cpp:290-291, :308-309
__device__ __forceinline__ float unwrap_value(float wrapped_shifted, int code)
{
    return (float)((double)wrapped_shifted + (double)code * 2.0 * PI_REF);
}

// lrint(fw*(phi/(2.0*Pi))) with the FE_INVALID and range rejections   5/compute_correspondance.cpp:648-675
// returns true if the coordinate is accepted
__device__ __forceinline__ bool correspond(float unwrapped, int fw, int limit, long &out)
{
    const double a = (double)fw * ((double)unwrapped / (2.0 * PI_REF));
    const double r = rint(a);  // round-half-even, the default rounding mode lrint runs under
    // FE_INVALID <=> NaN, inf or outside long; those and out-of-range values both clear the pixel
    const bool ok = (r >= 0.0) && (r <= (double)(limit - 1));
    out = ok ? (long)r : 0;
    return ok;
}

// ------------------------------------------------------------------------------------------------
// stage 7 (tolerance path: explicit fma, fp64)
// ------------------------------------------------------------------------------------------------
// T1: cvUndistortPoints (5 fixed-point iterations) then K*(x,y,1) and the homogeneous divide
//     7/triangulation.cpp:290-307 (camera), :363-378 (projector)
__device__ __forceinline__ void undistort_reproject(double px, double py, const Intr &I, double &u, double &v)
{
    const double x0 = (px - I.cx) * I.ifx, y0 = (py - I.cy) * I.ify;
    double x = x0, y = y0;
    if (I.has_dist) {
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double r2 = fma(x, x, y * y);
            const double icdist = 1.0 / fma(fma(fma(I.k3, r2, I.k2), r2, I.k1), r2, 1.0);
            const double dx = fma(2.0 * I.p1 * x, y, I.p2 * fma(2.0 * x, x, r2));
            const double dy = fma(I.p1, fma(2.0 * y, y, r2), 2.0 * I.p2 * x * y);
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
    }
    double uh = fma(I.K[0], x, fma(I.K[1], y, I.K[2]));
    double vh = fma(I.K[3], x, fma(I.K[4], y, I.K[5]));
    if (!I.affine) {
        const double wh = fma(I.K[6], x, fma(I.K[7], y, I.K[8]));
        uh /= wh;
        vh /= wh;
    }
    u = uh;
    v = vh;
}

// T2 + T3: P (4x3), F (4x1), V = (P^T P)^-1 P^T F   7/triangulation.cpp:1152-1168,1181-1188,1202-1206
// evaluated as adj(P^T P) (P^T F) / det(P^T P) (symmetric normal matrix; within 1e-12 of the literal order)
__device__ __forceinline__ void triangulate_px(const DevCal &C, double u, double v, double up, double vp, double X[3])
{
    double p[4][3], f[4];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        p[0][j] = fma(-u, C.Ac[8 + j], C.Ac[0 + j]);
        p[1][j] = fma(-v, C.Ac[8 + j], C.Ac[4 + j]);
        p[2][j] = fma(-up, C.Ap[8 + j], C.Ap[0 + j]);
        p[3][j] = fma(-vp, C.Ap[8 + j], C.Ap[4 + j]);
    }
    f[0] = fma(C.Ac[11], u, -C.Ac[3]);
    f[1] = fma(C.Ac[11], v, -C.Ac[7]);
    f[2] = fma(C.Ap[11], up, -C.Ap[3]);
    f[3] = fma(C.Ap[11], vp, -C.Ap[7]);
    double m00 = 0, m01 = 0, m02 = 0, m11 = 0, m12 = 0, m22 = 0, g0 = 0, g1 = 0, g2 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        m00 = fma(p[i][0], p[i][0], m00);
        m01 = fma(p[i][0], p[i][1], m01);
        m02 = fma(p[i][0], p[i][2], m02);
        m11 = fma(p[i][1], p[i][1], m11);
        m12 = fma(p[i][1], p[i][2], m12);
        m22 = fma(p[i][2], p[i][2], m22);
        g0 = fma(p[i][0], f[i], g0);
        g1 = fma(p[i][1], f[i], g1);
        g2 = fma(p[i][2], f[i], g2);
    }
    const double c00 = fma(m11, m22, -m12 * m12);
    const double c01 = fma(m02, m12, -m01 * m22);
    const double c02 = fma(m01, m12, -m02 * m11);
    const double c11 = fma(m00, m22, -m02 * m02);
    const double c12 = fma(m01, m02, -m00 * m12);
    const double c22 = fma(m00, m11, -m01 * m01);
    const double det = fma(m00, c00, fma(m01, c01, m02 * c02));
    // cvInvert returns a zero matrix when det == 0 (then V = 0)
    const double rdet = det != 0.0 ? 1.0 / det : 0.0;
    X[0] = fma(c00, g0, fma(c01, g1, c02 * g2)) * rdet;
    X[1] = fma(c01, g0, fma(c11, g1, c12 * g2)) * rdet;
    X[2] = fma(c02, g0, fma(c12, g1, c22 * g2)) * rdet;
}

// ------------------------------------------------------------------------------------------------
// fused kernel
// ------------------------------------------------------------------------------------------------
struct PixelResult {
    float x, y, z;
    bool valid;
};

template <bool KEEP>
__device__ __forceinline__ PixelResult pixel_chain(const KParams &P, const DevCal &C, int gx, int gy, int idx_v, int idx_h,
                                                   int code_v, int code_h, size_t keep_off)
{
    PixelResult R;
    const float nanv = __builtin_nanf("");
    R.x = R.y = R.z = nanv;
    // stage 4: the in-place +Pi and the unwrap skip the first/last column (v) or row (h) of the frame
    const bool in_v = gx >= 1 && gx <= P.fullW - 2;  // 4/phase_unwrap.cpp:285
    const bool in_h = gy >= 1 && gy <= P.fullH - 2;  // 4/phase_unwrap.cpp:304
    const float wv = in_v ? P.atab_shift[idx_v] : P.atab_phi[idx_v];
    const float wh = in_h ? P.atab_shift[idx_h] : P.atab_phi[idx_h];
    const float uv = in_v ? unwrap_value(wv, code_v) : 0.0f;  // unwrapped stays unset (0 here) outside the loop range
    const float uh = in_h ? unwrap_value(wh, code_h) : 0.0f;
    long cx, cy;
    const bool okx = correspond(uv, P.fwv, P.PW, cx);
    const bool oky = correspond(uh, P.fwh, P.PH, cy);
    R.valid = okx && oky;
    if (KEEP) {
        P.wrapped[0][keep_off] = wv;
        P.wrapped[1][keep_off] = wh;
        P.unwrapped[0][keep_off] = uv;
        P.unwrapped[1][keep_off] = uh;
        P.code[0][keep_off] = code_v;
        P.code[1][keep_off] = code_h;
        // c_p_map keeps whatever lrint produced even when the pixel is then rejected by the range
        // test; rejected pixels are never compared, store 0 for those
        P.cpmap[2 * keep_off + 0] = R.valid ? cx : 0;
        P.cpmap[2 * keep_off + 1] = R.valid ? cy : 0;
    }
    if (R.valid) {
        double u, v, up, vp, X[3];
        undistort_reproject((double)gx, (double)gy, C.cam, u, v);
        undistort_reproject((double)cx, (double)cy, C.proj, up, vp);
        triangulate_px(C, u, v, up, vp, X);
        R.x = (float)X[0];  // 8/save_point_cloud.cpp:100-102
        R.y = (float)X[1];
        R.z = (float)X[2];
        if (KEEP) {
            P.ipoints[3 * keep_off + 0] = X[0];
            P.ipoints[3 * keep_off + 1] = X[1];
            P.ipoints[3 * keep_off + 2] = X[2];
        }
    }
    return R;
}

template <bool KEEP>
__global__ __launch_bounds__(256) void k_fused(const KParams P, const DevCal C, int first_view)
{
    const int qpr = P.pitch >> 2;  // quads per row, pitch padding included
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(q / qpr), cq = (int)(q - (long)row * qpr);
    if (row >= P.H) return;
    const int view = first_view + blockIdx.y;
    const MaskView mv = mask_view(P, view);
    // F == 5: check_I_mod_criteria's assignment is commented out (3/wrapped_phase.cpp:117-129): nothing is valid
    const unsigned vbits = P.F == 5 ? 0u : quad_valid_bits(P, mv, cq, row);

    const size_t px = (size_t)view * P.px_view_stride + (size_t)row * P.pitch + (size_t)cq * 4;  // first pixel of the quad
    float4 *out_xyz = (float4 *)(P.points + 3 * px);
    const float nanv = __builtin_nanf("");
    float o[12];
#pragma unroll
    for (int i = 0; i < 12; i++) o[i] = nanv;
    unsigned vout = 0;

    if (KEEP) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            for (int a = 0; a < 2; a++) {
                P.wrapped[a][px + k] = 0.f;
                P.unwrapped[a][px + k] = 0.f;
                P.code[a][px + k] = -1;  // 4/phase_unwrap.cpp:143
                P.valid_axis[a][px + k] = (vbits >> k) & 1u;
            }
            P.cpmap[2 * (px + k)] = 0;
            P.cpmap[2 * (px + k) + 1] = 0;
            P.ipoints[3 * (px + k)] = P.ipoints[3 * (px + k) + 1] = P.ipoints[3 * (px + k) + 2] = 0.0;
        }
    }

    if (vbits != 0) {
        const uint8_t *fb = P.frames + (size_t)view * P.view_stride + (size_t)row * P.pitch + (size_t)cq * 4;
        const size_t ps = P.plane_stride;
        // ---- vertical axis planes: fringe F, gray Nv, inverse Nv ----
        unsigned f[2][4];
        int code[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        const uint8_t *pl = fb;
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int N = a == 0 ? P.Nv : P.Nh;
            f[a][0] = *(const unsigned *)(pl);
            f[a][1] = *(const unsigned *)(pl + ps);
            f[a][2] = *(const unsigned *)(pl + 2 * ps);
            f[a][3] = P.F == 4 ? *(const unsigned *)(pl + 3 * ps) : 0u;
            pl += (size_t)P.F * ps;
            unsigned b = 0;  // running binary bit per pixel, bit k
#pragma unroll 2
            for (int i = 0; i < N; i++) {
                const unsigned g = *(const unsigned *)(pl + (size_t)i * ps);
                const unsigned iv = *(const unsigned *)(pl + (size_t)(N + i) * ps);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // G_i = (gray - inverse >= 0); B_0 = G_0, B_i = B_{i-1} xor G_i; code = sum B_i 2^(N-1-i)
                    const unsigned ge = ((g >> (8 * k)) & 255u) >= ((iv >> (8 * k)) & 255u) ? 1u : 0u;  // 4/phase_unwrap.cpp:183
                    b ^= ge << k;
                    code[a][k] = code[a][k] * 2 + (int)((b >> k) & 1u);
                }
            }
            pl += (size_t)2 * N * ps;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((vbits >> k) & 1u) {
                const int sh = 8 * k;
                const int iv = fringe_index(P.F, (f[0][0] >> sh) & 255, (f[0][1] >> sh) & 255, (f[0][2] >> sh) & 255, (f[0][3] >> sh) & 255);
                const int ih = fringe_index(P.F, (f[1][0] >> sh) & 255, (f[1][1] >> sh) & 255, (f[1][2] >> sh) & 255, (f[1][3] >> sh) & 255);
                const PixelResult R = pixel_chain<KEEP>(P, C, P.col0 + cq * 4 + k, P.row0 + row, iv, ih, code[0][k], code[1][k], px + k);
                o[3 * k + 0] = R.x;
                o[3 * k + 1] = R.y;
                o[3 * k + 2] = R.z;
                vout |= (R.valid ? 1u : 0u) << (8 * k);
            }
        }
    }
    out_xyz[0] = make_float4(o[0], o[1], o[2], o[3]);
    out_xyz[1] = make_float4(o[4], o[5], o[6], o[7]);
    out_xyz[2] = make_float4(o[8], o[9], o[10], o[11]);
    *(unsigned *)(P.valid + px) = vout;
}

int launch_fused(const KParams &P, const DevCal &C, int first_view, int n_views, bool keep, void *stream)
{
    const long quads = (long)(P.pitch >> 2) * P.H;
    dim3 grid((unsigned)((quads + 255) / 256), (unsigned)n_views, 1), block(256, 1, 1);
    if (keep)
        hipLaunchKernelGGL(k_fused<true>, grid, block, 0, (hipStream_t)stream, P, C, first_view);
    else
        hipLaunchKernelGGL(k_fused<false>, grid, block, 0, (hipStream_t)stream, P, C, first_view);
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
        phi = P.atab_phi[fringe_index(P.F, i0, i1, i2, i3)];  // :175 / :198
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
        const bool okx = correspond(P.unwrapped[0][px], P.fwv, P.PW, cx);
        const bool oky = correspond(P.unwrapped[1][px], P.fwh, P.PH, cy);
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
        triangulate_px(C, u, v, up, vp, X);
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
