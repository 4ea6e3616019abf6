// sl3d_device.h -- device-side arithmetic shared by every kernel translation unit of the library (sl3d_kernels.hip: the per-stage
// and auxiliary kernels; sl3d_fused.h + sl3d_fused_*.hip: the fused hot path).  Everything here is __device__ __forceinline__.
//
// Arithmetic contract:
//   * everything up to the correspondence (x,y) is BIT-EXACT with the reference's C expressions:
//     (float)atan2 is evaluated in the kernel on the integer lattice its arguments live on and equals
//     the double-precision libm atan2 the reference calls (3/wrapped_phase.cpp:175) on every lattice
//     point (atan2_lattice4; proven by exhaustion on the CPU and again on the device at sl3d_create);
//     the +Pi, +code*2.0*Pi, /(2.0*Pi), *fw, lrint chain is evaluated in fp64 with the reference's
//     operation order and Pi = 22.0/7.0; every kernel file is compiled with -ffp-contract=off so no FMA is
//     formed behind our back.
//   * stage 7 (fp64 4x3 least squares) only has to match within 1e-5; it uses explicit fma().
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "sl3d_internal.h"
#include "sl3d_atan_coeffs.h"

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
// the scan has the closed form (validated against the literal loop on the CPU by the test suite):
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
    // generic (any position) evaluation of the closed form (used by the per-stage kernel k_wrap)
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
// Validity after stage 3's boundary removal is a function of the selection mask alone, so it is evaluated once per
// sl3d_set_mask for every pixel of the window (k_mask_prepare: the generic closed form above, MaskView::valid) into the
// `band` plane -- one 0/1 byte per pixel, 0 in the pitch padding -- and the fused kernel reads ONE dword per quad and view
// instead of 3 rows x 12 mask bytes plus ~45 instructions of byte-parallel logic (round 1 evaluated only the quads within
// 3 pixels of the frame border ahead of time).  The load (MaskQuad) is separate from its use so that the next view's
// dword can be requested a view ahead.
struct MaskQuad {
    unsigned band;
};

__device__ __forceinline__ MaskQuad load_mask_quad(const KParams &P, int view, unsigned lane_off);  // (below, with the addressing helpers)

__device__ __forceinline__ unsigned mask_quad_bits(const MaskQuad &m)
{
    const unsigned w = m.band;
    return (w & 1u) | ((w >> 7) & 2u) | ((w >> 14) & 4u) | ((w >> 21) & 8u);
}
// ------------------------------------------------------------------------------------------------
// bit-exact phase chain
// ------------------------------------------------------------------------------------------------
// 1/d to ~1 ulp: v_rcp_f64 seed + two Newton steps
__device__ __forceinline__ double recip(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    return fma(r, e, r);
}

// 1/d for the tolerance path (stage 7) and for the lattice atan2 without the LDS table: v_rcp_f64 is good to 2^-24.4,
// one Newton step brings it to 2.2e-15 (tools/valubench measures both over 2^26 doubles), the second one (recip) to
// the correctly rounded value
__device__ __forceinline__ double recip1(double d)
{
    const double r = __builtin_amdgcn_rcp(d);
    return fma(r, fma(-d, r, 1.0), r);
}

// Wrapped phase without a table: (float)atan2((double)t1,(double)t2) for the small integers the
// fringe frames produce (|t1| <= 255, |t2| <= 510), evaluated in fp64 so that, after rounding to
// float, it equals the double-precision libm atan2 the reference calls (3/wrapped_phase.cpp:175)
// on EVERY point of that lattice.  That equality is not assumed: tests/native/exact_arith_check.c proves it
// on the CPU with the same constants (sl3d_atan_coeffs.h), and sl3d_create() runs k_atan_selfcheck over all
// 521,731 points against a table built with the host's libm and refuses to create a context if a single
// value differs (tests/test_gpu_stage_parity.py repeats the check).
// Method: octant reduction on the integers, a second reduction lo/hi > 70/169 -> (hi-lo)/(hi+lo) (still a
// quotient of small integers, so there is exactly one division), atan(r) = r + r*z*Q(z), z = r^2, Horner.
// Accuracy budget: the true atan2 of a lattice point stays >= 6.7e-14 (relative, ~300 ulp of a double) away
// from every float rounding boundary, so the quotient needs no correctly rounded division (n * RN(1/d) is
// within 1 ulp), pi/4 and pi need no low words, and Q needs degree 8, not 10 (sl3d_atan_coeffs.h).
// A 2 MB gather table costs more than this arithmetic: every wave-level gather pulls 64 separate
// 128-B lines through the vector L1 for 256 useful bytes (tools/membench.hip, flags=4: -50%).
// rcp_tab: optional LDS table of correctly rounded 1/d, d = 0..767 (entry 0 holds 1); nullptr = compute it
#define SL3D_RCP_TAB 768
__device__ __forceinline__ void fill_rcp_table(double *tab)
{
    for (int i = threadIdx.x; i < SL3D_RCP_TAB; i += blockDim.x) tab[i] = 1.0 / (double)(i == 0 ? 1 : i);  // IEEE division
}

// Horner coefficients of Q, highest degree first.  SGPR = true pins each one in a scalar register pair right
// where it is called: an fp64 FMA can take one scalar operand, so every Horner step is a single v_fma_f64.
// (Left to itself the compiler hoists the constants into VGPR pairs for the whole kernel and issues a
// v_mov_b64 + v_fmac_f64 pair per step.)
struct AtanK {
    double c[SL3D_ATAN_DEG + 1];
};
template <bool SGPR>
__device__ __forceinline__ AtanK atan_consts()
{
    AtanK K = {SL3D_ATAN_Q};
    if (SGPR) {
#pragma unroll
        for (int j = 0; j <= SL3D_ATAN_DEG; j++) asm volatile("" : "+s"(K.c[j]));
    }
    return K;
}

// TAB: rcp_tab (LDS, SL3D_RCP_TAB entries) supplies the correctly rounded reciprocal; otherwise rcp + Newton.
// The arguments come as differences of small non-negative integers, t1 = a - b, t2 = c - d (a..d < 2^16): the absolute
// values are one v_sad_u16 each and the signs one compare each.
template <bool TAB>
__device__ __forceinline__ float atan2_lattice4(unsigned a, unsigned b, unsigned c, unsigned d, const double *rcp_tab, const AtanK &K)
{
    const unsigned ay = __builtin_amdgcn_sad_u16(a, b, 0u), ax = __builtin_amdgcn_sad_u16(c, d, 0u);
    const bool neg1 = a < b, neg2 = c < d;  // t1 < 0, t2 < 0
    const unsigned lo = min(ay, ax), hi = max(ay, ax);
    const bool swap = ay > ax;
    const bool red = __umul24(169u, lo) > __umul24(70u, hi);  // lo/hi > 0.414201 (just below tan(pi/8)); full-rate 24-bit multiplies
    const unsigned num = red ? hi - lo : lo, den = red ? hi + lo : hi;
    // num/den to 1 ulp (den == 0 only for t1 == t2 == 0, where num == 0 as well: use 0/1); den <= 255 + 510
    const unsigned den1 = max(den, 1u);
    const double r = (double)num * (TAB ? rcp_tab[den1] : recip1((double)den1));
    const double z = r * r;
    double p = K.c[0];
#pragma unroll
    for (int j = 1; j <= SL3D_ATAN_DEG; j++) p = fma(p, z, K.c[j]);
    const double at = fma(r, z * p, r);
    // first octant pair:  !swap,!red: at | !swap,red: pi/4 - at | swap,red: pi/4 + at | swap,!red: pi/2 - at
    // i.e. phi1 = k*(pi/4) + s*at with k = red ? 1 : (swap ? 2 : 0), s = -1 iff swap != red;
    // t2 < 0: phi2 = pi - phi1 = (4-k)*(pi/4) - s*at, ONE fma on an integer multiplier and a sign-adjusted at
    // (k*pi/4 + at is rounded once); t1 < 0 flips the sign of the (non-negative) float result.
    const int k1 = red ? 1 : (swap ? 2 : 0);
    const int k2 = neg2 ? 4 - k1 : k1;
    const bool nega = (swap != red) != neg2;
    const float phi = (float)fma((double)k2, SL3D_PIO4, nega ? -at : at);
    return neg1 ? -phi : phi;
}

template <bool TAB>
__device__ __forceinline__ float atan2_lattice(int t1, int t2, const double *rcp_tab, const AtanK &K)
{
    return atan2_lattice4<TAB>((unsigned)max(t1, 0), (unsigned)max(-t1, 0), (unsigned)max(t2, 0), (unsigned)max(-t2, 0), rcp_tab, K);
}

// (t1,t2) of create_wrapped_phase: 3-step 3/wrapped_phase.cpp:171-172, 4-step :195-196 (exact small integers)
template <bool TAB>
__device__ __forceinline__ float wrapped_phase(int F, unsigned i0, unsigned i1, unsigned i2, unsigned i3, const double *rcp_tab, const AtanK &K)
{
    if (F == 3) return atan2_lattice4<TAB>(i0, i2, 2u * i1, i0 + i2, rcp_tab, K);
    return atan2_lattice4<TAB>(i3, i1, i0, i2, rcp_tab, K);
}

// the value wrapped_phi holds after stage 4's in-place `+= Pi` (4/phase_unwrap.cpp:290,308)
__device__ __forceinline__ float shift_pi(float phi) { return (float)((double)phi + PI_REF); }
// same, applied only where stage 4's loop runs: adding 0.0 in double and rounding back returns phi itself
__device__ __forceinline__ float shift_pi_if(float phi, bool in_range) { return (float)((double)phi + (in_range ? PI_REF : 0.0)); }

// Correctly rounded division by a constant without the IEEE divide expansion (Markstein): with
// y = RN(1/c), q0 = RN(a*y), r = a - q0*c (exact, one fma), q = RN(q0 + r*y) equals RN(a/c).
// tests/native/exact_arith_check.c (run by tests/test_exact_arith.py) proves q == a/c exhaustively for the two uses below: c = 7 over every
// a = 44*code, code < 2^20, and c = 44/7 over every float a in [5e-4, 6e4] (all absolute phases).
__device__ __forceinline__ double div_exact(double a, double c, double y)
{
    const double q0 = a * y;
    const double r = fma(-q0, c, a);
    return fma(r, y, q0);
}

// unwrapped = wrapped(+Pi already applied) + code*2.0*Pi          4/phase_unwrap.cpp:290-291, :308-309
// code*2.0*Pi expands to ((code*2.0)*22.0)/7.0; the two products are exact integers (= 44*code)
__device__ __forceinline__ float unwrap_value(float wrapped_shifted, int code)
{
    const double k = div_exact((double)(code * 44), 7.0, 1.0 / 7.0);
    return (float)((double)wrapped_shifted + k);
}

// lrint(fw*(phi/(2.0*Pi))) with the FE_INVALID and range rejections   5/compute_correspondance.cpp:648-675
// returns true if the coordinate is accepted.  phi is 0 (unset) or a positive finite absolute phase.
__device__ __forceinline__ bool correspond(float unwrapped, int fw, int limit, long &out, double &out_d)
{
    const double c = 2.0 * PI_REF;  // (2.0*Pi) -> (2.0*22.0)/7.0, folded at compile time exactly as on the host
    const double a = (double)fw * div_exact((double)unwrapped, c, 1.0 / c);
    const double r = rint(a);  // round-half-even, the default rounding mode lrint runs under
    // FE_INVALID <=> NaN, inf or outside long; those and out-of-range values both clear the pixel.  The range test is made on
    // the double itself (NaN compares false), so no out-of-range value is ever converted to an integer.
    const bool ok = r >= 0.0 && r <= (double)(limit - 1);
    out = ok ? (long)(int)r : 0;
    out_d = r;  // the same integer as a double (exact), for stage 7
    return ok;
}

// ------------------------------------------------------------------------------------------------
// stage 7 (tolerance path: explicit fma, fp64)
// ------------------------------------------------------------------------------------------------
// T1: cvUndistortPoints (5 fixed-point iterations) then K*(x,y,1) and the homogeneous divide
//     7/triangulation.cpp:290-307 (camera), :363-378 (projector)
// Terms whose coefficient is exactly zero are skipped through wave-uniform flags; each skipped term is
// an exact zero in the reference's arithmetic, so the value is unchanged.
// the 5 fixed-point iterations of cvUndistortPoints on normalised coordinates
// icd (optional): the factor of the last iteration of a purely radial model, for which the result is exactly (x0*icd, y0*icd)
template <typename IntrT>
__device__ __forceinline__ void undistort_normalized(double px, double py, const IntrT &I, double &xo, double &yo, double *icd = nullptr)
{
    const double x0 = (px - I.cx) * I.ifx, y0 = (py - I.cy) * I.ify;
    double x = x0, y = y0;
    if (icd) *icd = 1.0;
    if (I.has_dist) {
        if (I.has_tan) {
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const double r2 = fma(x, x, y * y);
                const double icdist = recip(fma(fma(fma(I.k3, r2, I.k2), r2, I.k1), r2, 1.0));
                const double dx = fma(2.0 * I.p1 * x, y, I.p2 * fma(2.0 * x, x, r2));
                const double dy = fma(I.p1, fma(2.0 * y, y, r2), 2.0 * I.p2 * x * y);
                x = (x0 - dx) * icdist;
                y = (y0 - dy) * icdist;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const double r2 = fma(x, x, y * y);
                const double icdist = recip(fma(fma(fma(I.k3, r2, I.k2), r2, I.k1), r2, 1.0));
                x = x0 * icdist;
                y = y0 * icdist;
                if (icd) *icd = icdist;
            }
        }
    }
    xo = x;
    yo = y;
}

// The same iteration for a purely radial model as a function of r0^2 = x0^2 + y0^2 alone: every iterate is (x0, y0) times a
// factor, so r_j^2 = r0^2 * icd_(j-1)^2 and the result is (x0, y0) * icd_5.  Returns icd_5 - 1 (radial tables, RadEntry).
template <typename IntrT>
__device__ __forceinline__ double radial_factor_m1(double r0sq, const IntrT &I)
{
    double r2 = r0sq, icd = 1.0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        icd = recip(fma(fma(fma(I.k3, r2, I.k2), r2, I.k1), r2, 1.0));
        r2 = r0sq * icd * icd;
    }
    return icd - 1.0;
}

// s(r0^2) from a radial table (RadEntry, here in LDS): the node nearest to r0^2 * scale, one 16-byte read, one float FMA + multiply,
// one double add.  The node coordinate t and the distance w to the node are formed in fp64 (r0^2 is a double anyway): in fp32, t up
// to 255 carries 1.5e-5 node spacings of rounding error into w, i.e. ~1e-9 |k1| into s -- several times the interpolation remainder
// the table was sized for; in fp64 only that remainder (~1e-10) is left.
__device__ __forceinline__ double radial_lookup(const RadEntry *tab, float scale, double r0sq)
{
    const double t = r0sq * (double)scale;
    const int i = min(__double2int_rn(t), SL3D_RAD_NODES - 1);  // (a rejected pixel's harmless index 0 and anything past the table end: clamped)
    const float w = (float)(t - (double)i);
    const RadEntry e = tab[i];
    return e.c0 + (double)(fmaf(e.c2, w, e.c1) * w);
}

// K * (x, y, 1) and the homogeneous divide
template <typename IntrT>
__device__ __forceinline__ void reproject(double x, double y, const IntrT &I, double &u, double &v)
{
    double uh, vh;
    if (I.plain) {  // K = [fx 0 cx; 0 fy cy; 0 0 1]
        uh = fma(I.K[0], x, I.K[2]);
        vh = fma(I.K[4], y, I.K[5]);
    } else {
        uh = fma(I.K[0], x, fma(I.K[1], y, I.K[2]));
        vh = fma(I.K[3], x, fma(I.K[4], y, I.K[5]));
        if (!I.affine) {
            const double iw = recip(fma(I.K[6], x, fma(I.K[7], y, I.K[8])));
            uh *= iw;
            vh *= iw;
        }
    }
    u = uh;
    v = vh;
}

template <typename IntrT>
__device__ __forceinline__ void undistort_reproject(double px, double py, const IntrT &I, double &u, double &v)
{
    double x, y;
    undistort_normalized(px, py, I, x, y);
    reproject(x, y, I, u, v);
}

// T2 + T3: P (4x3), F (4x1), V = (P^T P)^-1 P^T F   7/triangulation.cpp:1152-1168,1181-1188,1202-1206
// evaluated as adj(P^T P) (P^T F) / det(P^T P) (symmetric normal matrix; within 1e-12 of the literal order)
// The third row of each projection matrix (A[2][0..3]) multiplies the variable in every entry of P and F.
// An fp64 FMA can read only one scalar register, so with all of A in SGPRs every entry costs an extra
// v_mov_b64; the kernel therefore keeps these 8 doubles in VGPRs (PinnedRows), loaded once per lane.
struct PinnedRows {
    double c2[4], p2[4];  // A_cam[2][0..3], A_proj[2][0..3]
    double t[3];          // fast rig only: tcn (an addend the compiler would otherwise copy into a VGPR pair per use)
};

template <typename AP>
__device__ __forceinline__ void tri_row(AP A, const double a2[4], double t, double &m00, double &m01, double &m02, double &m11,
                                        double &m12, double &m22, double &g0, double &g1, double &g2, int r)
{
    // row of P: A[r][0..2] - t*A[2][0..2]; entry of F: A[2][3]*t - A[r][3]
    const double p0 = fma(-t, a2[0], A[4 * r + 0]), p1 = fma(-t, a2[1], A[4 * r + 1]), p2 = fma(-t, a2[2], A[4 * r + 2]);
    const double f = fma(a2[3], t, -A[4 * r + 3]);
    m00 = fma(p0, p0, m00); m01 = fma(p0, p1, m01); m02 = fma(p0, p2, m02);
    m11 = fma(p1, p1, m11); m12 = fma(p1, p2, m12); m22 = fma(p2, p2, m22);
    g0 = fma(p0, f, g0); g1 = fma(p1, f, g1); g2 = fma(p2, f, g2);
}

template <typename CalT>
__device__ __forceinline__ void triangulate_px(const CalT &C, const PinnedRows &R, double u, double v, double up, double vp, double X[3])
{
    double m00 = 0, m01 = 0, m02 = 0, m11 = 0, m12 = 0, m22 = 0, g0 = 0, g1 = 0, g2 = 0;
    tri_row(C.Ac, R.c2, u, m00, m01, m02, m11, m12, m22, g0, g1, g2, 0);
    tri_row(C.Ac, R.c2, v, m00, m01, m02, m11, m12, m22, g0, g1, g2, 1);
    tri_row(C.Ap, R.p2, up, m00, m01, m02, m11, m12, m22, g0, g1, g2, 0);
    tri_row(C.Ap, R.p2, vp, m00, m01, m02, m11, m12, m22, g0, g1, g2, 1);
    const double c00 = fma(m11, m22, -m12 * m12);
    const double c01 = fma(m02, m12, -m01 * m22);
    const double c02 = fma(m01, m12, -m02 * m11);
    const double c11 = fma(m00, m22, -m02 * m02);
    const double c12 = fma(m01, m02, -m00 * m12);
    const double c22 = fma(m00, m11, -m01 * m01);
    const double det = fma(m00, c00, fma(m01, c01, m02 * c02));
    // cvInvert returns a zero matrix when det == 0 (then V = 0)
    const double rdet = det != 0.0 ? recip(det) : 0.0;
    X[0] = fma(c00, g0, fma(c01, g1, c02 * g2)) * rdet;
    X[1] = fma(c01, g0, fma(c11, g1, c12 * g2)) * rdet;
    X[2] = fma(c02, g0, fma(c12, g1, c22 * g2)) * rdet;
}

// The same least-squares problem in the camera frame (DevCal::Apc): the camera rows fx*(1,0,-xn), fy*(0,1,-yn) have a
// closed-form normal matrix, only the two projector rows are accumulated, and the solution is rotated back to world
// coordinates with the numerator (X = Rct*(adj*g)/det + tcn).  69 fp64 operations instead of 84, and the camera's third
// row needs no pinned registers.  det == 0 (cvInvert's zero matrix, V = 0) is reported through `singular`.
template <typename CalT>
__device__ __forceinline__ void triangulate_camframe(const CalT &C, const PinnedRows &R, double xn, double yn, double up, double vp, double X[3],
                                                     bool &singular)
{
    // camera rows fx*(1,0,-xn) + s*(0,1,-yn) and fy*(0,1,-yn): their outer products, with q = fx*s and r = s^2 + fy^2
    // (q = 0 for the usual K: a and b are then fx^2*xn and fy^2*yn, bit for bit what round 2 computed)
    const double a = fma(C.fxs, yn, C.fx2 * xn), b = fma(C.fxs, xn, C.fy2 * yn);
    double m00 = C.fx2, m01 = C.fxs, m02 = -a, m11 = C.fy2, m12 = -b, m22 = fma(a, xn, b * yn), g0 = 0, g1 = 0, g2 = 0;
    tri_row(C.Apc, R.p2, up, m00, m01, m02, m11, m12, m22, g0, g1, g2, 0);
    tri_row(C.Apc, R.p2, vp, m00, m01, m02, m11, m12, m22, g0, g1, g2, 1);
    const double c00 = fma(m11, m22, -m12 * m12);
    const double c01 = fma(m02, m12, -m01 * m22);
    const double c02 = fma(m01, m12, -m02 * m11);
    const double c11 = fma(m00, m22, -m02 * m02);
    const double c12 = fma(m01, m02, -m00 * m12);
    const double c22 = fma(m00, m11, -m01 * m01);
    const double det = fma(m00, c00, fma(m01, c01, m02 * c02));
    singular = det == 0.0;
    const double rdet = recip1(det);
    const double n0 = fma(c00, g0, fma(c01, g1, c02 * g2));
    const double n1 = fma(c01, g0, fma(c11, g1, c12 * g2));
    const double n2 = fma(c02, g0, fma(c12, g1, c22 * g2));
    X[0] = fma(fma(C.Rct[0], n0, fma(C.Rct[1], n1, C.Rct[2] * n2)), rdet, R.t[0]);
    X[1] = fma(fma(C.Rct[3], n0, fma(C.Rct[4], n1, C.Rct[5] * n2)), rdet, R.t[1]);
    X[2] = fma(fma(C.Rct[6], n0, fma(C.Rct[7], n1, C.Rct[8] * n2)), rdet, R.t[2]);
}

// ------------------------------------------------------------------------------------------------
// fused kernel
// ------------------------------------------------------------------------------------------------
// Hide a wave-uniform pointer from the optimiser: loads through it can neither be hoisted out of the
// enclosing loop nor strength-reduced into dozens of live scalar registers.  (Without this the 46 plane
// addresses and the 60 fp64 calibration constants are kept in SGPRs across the loops; gfx950 has 102, the
// overflow is spilled to VGPR lanes and re-read with v_readlane -- measured at ~30 % of all VALU issue.)
#define GLOBAL_AS __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const GLOBAL_AS T *opaque(const T *p)
{
    asm volatile("" : "+s"(p));
    return (const GLOBAL_AS T *)p;  // the asm hides the provenance: restate that this is global memory
}
// same for read-only constants: the constant address space tells the compiler the memory is never written
// while the kernel runs, so wave-uniform loads become scalar loads (s_load, scalar cache) instead of
// vector loads that every pixel iteration would have to wait for with vmcnt(0)
#define CONST_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const CONST_AS T *opaque_const(const T *p)
{
    asm volatile("" : "+s"(p));
    return (const CONST_AS T *)p;
}
__device__ __forceinline__ unsigned opaque_u32(unsigned v)
{
    asm volatile("" : "+s"(v));
    return v;
}
// dword at (wave-uniform base) + (32-bit lane offset): the saddr + voffset form of global_load_dword.
// The base is hidden behind an empty asm: otherwise the optimiser re-associates (view base + lane offset) + plane
// offset and spends one 64-bit VALU add per load (46 v_lshl_add_u64 per quad) instead of two SALU adds.
// The planes are read exactly once, by exactly one CU: the loads carry the non-temporal hint (+1.1 %).  (gfx950's sc0 / sc1 scope
// bits were probed through raw buffer loads in round 3: +-0 in the fused kernel, profiles/r03_membench_scope.txt.)
__device__ __forceinline__ unsigned ldg32(const GLOBAL_AS uint8_t *base, unsigned off)
{
    asm volatile("" : "+s"(base));
    return __builtin_nontemporal_load((const GLOBAL_AS unsigned *)(base + (size_t)off));
}

// the same form for a cached load and for streaming stores: (wave-uniform base in SGPRs) + (32-bit lane offset); with the base
// hidden, nothing of the address is loop-invariant VGPR state (hoisted base + lane offset pairs were what the register allocator
// spilled in the small-launch instantiations)
__device__ __forceinline__ unsigned ldg32_cached(const GLOBAL_AS uint8_t *base, unsigned off)
{
    asm volatile("" : "+s"(base));
    return *(const GLOBAL_AS unsigned *)(base + (size_t)off);
}
// (the builtin is int -> int: widened directly, its result would be SIGN-extended)
__device__ __forceinline__ unsigned first_lane_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
template <typename T>
__device__ __forceinline__ void stg_nt(GLOBAL_AS uint8_t *base, unsigned off, T v)
{
    asm volatile("" : "+s"(base));
    __builtin_nontemporal_store(v, (GLOBAL_AS T *)(base + (size_t)off));
}
template <typename T>
__device__ __forceinline__ GLOBAL_AS uint8_t *opaque_out(T *p)
{
    // (wave-uniform by construction; said explicitly, because the divergence analysis does not always see it -- folded away where it does)
    unsigned long long u = (unsigned long long)p;
    u = ((unsigned long long)first_lane_u32((unsigned)(u >> 32)) << 32) | (unsigned long long)first_lane_u32((unsigned)u);
    asm volatile("" : "+s"(u));
    return (GLOBAL_AS uint8_t *)u;
}

// a float2 at (wave-uniform base) + (32-bit byte offset), cached (the projector table of rig class 2: neighbouring lanes share lines)
__device__ __forceinline__ float2 ldg_f2(const GLOBAL_AS uint8_t *base, unsigned off)
{
    asm volatile("" : "+s"(base));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = *(const GLOBAL_AS f32x2 *)(base + (size_t)off);
    return make_float2(v.x, v.y);
}

// the valid-map dword of the lane's quad (lane_off = byte offset of the quad inside any plane of a view)
__device__ __forceinline__ MaskQuad load_mask_quad(const KParams &P, int view, unsigned lane_off)
{
    MaskQuad m;
    // read once per view, by one lane (the non-temporal hint on it was measured: 361.8-362.8 us against 359.4-359.9)
    m.band = ldg32_cached(opaque(P.band + (size_t)view * P.px_view_stride), lane_off);
    return m;
}

}  // namespace sl3d
