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
//     (float)atan2 is evaluated in the kernel on the integer lattice its arguments live on and equals
//     the double-precision libm atan2 the reference calls (3/wrapped_phase.cpp:175) on every lattice
//     point (atan2_lattice4; proven by exhaustion on the CPU and again on the device at sl3d_create);
//     the +Pi, +code*2.0*Pi, /(2.0*Pi), *fw, lrint chain is evaluated in fp64 with the reference's
//     operation order and Pi = 22.0/7.0; this file is compiled with -ffp-contract=off so no FMA is
//     formed behind our back.
//   * stage 7 (fp64 4x3 least squares) only has to match within 1e-5; it uses explicit fma().
//
// The SL3D_* macros below are compile-time switches for measured A/B builds (tools/ab.sh); the
// defaults are the shipped configuration and every rejected alternative is recorded in DESIGN.md.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "sl3d_internal.h"
#include "sl3d_atan_coeffs.h"

// the timed pixel loop handles the 4 pixels of a lane as 2 pairs: 1 = rolled pair loop, 2 = both pairs unrolled
#ifndef SL3D_PAIR_UNROLL
#define SL3D_PAIR_UNROLL 1
#endif
#ifndef SL3D_OCC
#define SL3D_OCC 4 /* waves per SIMD the fused kernel is compiled for */
#endif
#ifndef SL3D_BLOCK
#define SL3D_BLOCK 256 /* threads per block of the fused kernel (tools/ab.sh: 128 and 512 measured) */
#endif
// 1: the fused kernel reads 1/d of the atan2 quotient from an LDS table (6 KB per block); 0: v_rcp_f64 + one Newton step
#ifndef SL3D_RCP_LDS
#define SL3D_RCP_LDS 1
#endif
// non-temporal hints (tools/ab.sh): the planes are read exactly once by exactly one CU (nt loads: +1.1 %); the results are
// written once too, but nt stores lose the L2's merging of the three partial-line stores of a wave (-11 %)
#ifndef SL3D_NT_LOADS
#define SL3D_NT_LOADS 1
#endif
#ifndef SL3D_NT_STORES
#define SL3D_NT_STORES 0
#endif
// dense xyz stores: 0 = three 16-byte stores per lane at a 48-byte lane stride (the L2 merges the partial lines), 1 = read back
// across the wave's lanes from LDS so that every store instruction writes 1 KiB of whole lines for EVERY launch (round 1: -1.2 %,
// round 3: -2.4 % at 16 views; the one-view launch takes the coalesced form whatever this switch says, see store_quad)
#ifndef SL3D_NT_COALESCED
#define SL3D_NT_COALESCED 1 /* the coalesced store path (whole 1-KiB runs per instruction) carries the non-temporal hint, see store_quad */
#endif
#ifndef SL3D_NT_AUX
#define SL3D_NT_AUX 0 /* the hint on the valid-byte dword of a quad as well: measured, 361.8-362.8 us against 359.4-359.9 (profiles/r03_pipe_ntloads_ab.txt) */
#endif
#ifndef SL3D_NT_SEG
#define SL3D_NT_SEG 1 /* the same hint on the segment stores (and the valid dword) of the segmented clouds */
#endif
#ifndef SL3D_COALESCED_STORES
#define SL3D_COALESCED_STORES 1
#endif
#ifndef SL3D_EARLY_PLANES
#define SL3D_EARLY_PLANES 1 /* small-launch instantiation: the first view's planes requested before the mask is known (see EARLY in k_fused) */
#endif
// 1: XCD-banded tile order.  Workgroups go round-robin to the 8 XCDs (each with its own L2); with the natural order the
// three tiles that share a mask row (vertical neighbours are 1.9 tiles apart) land on three different L2s, and the mask
// is what the measured 1.046x traffic over the algorithmic bytes consists of.  With 1, XCD x walks the x-th eighth of
// the window top to bottom, so vertically adjacent tiles share an L2 -- but HBM then sees 8 distant streams per plane
// instead of one: measured -4 % (4 x 4000 launches, alternating).  The natural order stays: for a streaming kernel
// with 2 % of shared bytes, DRAM locality across XCDs is worth more than L2 locality inside one.
#ifndef SL3D_XCD_BANDS
#define SL3D_XCD_BANDS 0
#endif
// timed kernels: 1 = the pixel loop is cut in two phases (correspondences of all 4 pixels, then stage 7), 0 = one chain per pixel
#ifndef SL3D_SPLIT
#define SL3D_SPLIT 1
#endif
// timed kernels: 1 = the next view's planes are requested between the two phases of the current view (needs SL3D_SPLIT)
#ifndef SL3D_PIPE
#define SL3D_PIPE 1
#endif
#ifndef SL3D_PIPE_RIG0
#define SL3D_PIPE_RIG0 0 /* the general rig's stage 7 is register-hungry: pipelined it spills (measured below) */
#endif
#ifndef SL3D_MASK_PREFETCH
#define SL3D_MASK_PREFETCH 1
#endif

// measurement only (tools/ab.sh builds with -DSL3D_MEASURE -DSL3D_ABLATE=n): 1 = no per-pixel arithmetic (xyz made of the raw
// decode results), 2 = no mask reads, 4 = no xyz stores.  Results are wrong by construction; the shipped build has neither
// the compile-time switch nor the run-time hooks (SL3D_VPT / SL3D_ABLATE environment variables, KParams::ablate).
#if !defined(SL3D_MEASURE) || !defined(SL3D_ABLATE)
#undef SL3D_ABLATE
#define SL3D_ABLATE 0
#endif
// measurement builds only (-DSL3D_MEASURE -DSL3D_CX=bits): parts of the in-kernel compaction switched off or instrumented
// (results wrong by construction for 1 / 2 / 4 / 512; the A/B tables of DESIGN.md 4b come from these)
//   1 no look-back (prefix = tile * 1024)     2 no barrier before the stores     4 no barrier after the pixel loop
//   64 look-back counters (calls, rounds, re-polls, ticks) printed at sl3d_destroy     128 clock stamps per tile (tools/lb_trace.py)
//   512 no look-back for the LAST view of a block
#if !defined(SL3D_MEASURE) || !defined(SL3D_CX)
#undef SL3D_CX
#define SL3D_CX 0
#endif
// measurement builds only (-DSL3D_MEASURE -DSL3D_TRACE): wall-clock stamps (100 MHz) of every wave of the dense timed kernel at
// its phase boundaries, first view of the item: 0 entry, 1 reciprocal table filled, 2 item set up (camera table entries, first
// mask dword requested), 3 plane loads issued, 4 planes landed + decoded, 5 phase A done, 6 phase B done, 7 stores issued
// -> KParams::dbg [block][wave][8] (tools/phase_trace.py)
#if defined(SL3D_MEASURE) && defined(SL3D_TRACE)
#define SL3D_STAMP(k)                                                                                                               \
    do {                                                                                                                            \
        /* every lane of the wave stores the same (scalar) clock to the same word: no divergent branch in the instrumented code */   \
        if (CMODE == 0 && !KEEP && P.dbg)                                                                                           \
            P.dbg[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 8 + (k)] = wall_clock64();              \
    } while (0)
#else
#define SL3D_STAMP(k)
#endif
#ifdef SL3D_MEASURE
#define SL3D_ABLATE_RT(P) ((P).ablate)
#else
#define SL3D_ABLATE_RT(P) 0
#endif
#define SL3D_PRAGMA_(x) _Pragma(#x)
#define SL3D_UNROLL(n) SL3D_PRAGMA_(unroll n)

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

__device__ __forceinline__ MaskQuad load_mask_quad(const KParams &P, int view, int cq, int row)
{
    MaskQuad m;
    const unsigned *bp = (const unsigned *)(P.band + (size_t)view * P.px_view_stride + (size_t)row * P.pitch + cq * 4);
    m.band = SL3D_NT_AUX ? __builtin_nontemporal_load(bp) : *bp;  // read once per view, by one lane
    return m;
}

__device__ __forceinline__ unsigned mask_quad_bits(const MaskQuad &m)
{
    const unsigned w = m.band;
    return (w & 1u) | ((w >> 7) & 2u) | ((w >> 14) & 4u) | ((w >> 21) & 8u);
}

// sl3d_set_mask on the device: `raw` holds the caller's bytes of the window + 2-pixel halo (clipped to the frame) in the
// layout of the mask plane itself (row r of the plane = window row r - 2, byte SL3D_MASK_LPAD + c = window column c).
// One lane per dword of the plane: normalises the bytes to 0/1 (selected iff byte == 1; outside the frame or the halo: 0; the
// per-stage kernel k_wrap evaluates the boundary removal on this plane) and evaluates the generic closed form of the
// boundary removal (MaskView::valid on the raw bytes) for EVERY pixel of the window into the band plane the fused kernel reads.
__global__ __launch_bounds__(256) void k_mask_prepare(const KParams P, int view, const uint8_t *__restrict__ raw)
{
    const int dwords_per_row = P.mpitch >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(t / dwords_per_row), x = (int)(t - (long)r * dwords_per_row);
    if (r >= P.H + 2 * SL3D_MASK_HALO) return;
    MaskView m;
    m.base = raw + (size_t)SL3D_MASK_HALO * P.mpitch + SL3D_MASK_LPAD;
    m.mpitch = P.mpitch;
    m.col0 = P.col0; m.row0 = P.row0; m.fullW = P.fullW; m.fullH = P.fullH;
    const int c0 = x * 4 - SL3D_MASK_LPAD, wr = r - SL3D_MASK_HALO;  // window column of byte 0, window row
    const int gy = P.row0 + wr;
    unsigned norm = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c = c0 + k;
        if (c >= -SL3D_MASK_HALO && c < P.W + SL3D_MASK_HALO && m.V(P.col0 + c, gy)) norm |= 1u << (8 * k);
    }
    uint8_t *dst = (uint8_t *)P.mask + (size_t)view * P.mask_view_stride;
    *(unsigned *)(dst + (size_t)r * P.mpitch + (size_t)x * 4) = norm;
    if (wr >= 0 && wr < P.H && c0 >= 0 && c0 < P.pitch) {
        unsigned band = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (c0 + k < P.W && m.valid(P.col0 + c0 + k, gy)) band |= 1u << (8 * k);
        *(unsigned *)((uint8_t *)P.band + (size_t)view * P.px_view_stride + (size_t)wr * P.pitch + (size_t)c0) = band;
    }
}

int launch_mask_prepare(const KParams &P, int view, const uint8_t *raw, void *stream)
{
    const long n = (long)(P.mpitch >> 2) * (P.H + 2 * SL3D_MASK_HALO);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_mask_prepare, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, view, raw);
    return (int)hipGetLastError();
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
// value differs (tests/test_gpu_parity.py repeats the check).
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
// A/B (SL3D_BUF_LOADS): the same dword through a raw buffer load -- resource of the view, lane offset, SCALAR plane offset -- whose
// cache-policy operand reaches the scope bits the global-load builtins do not (aux: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef SL3D_BUF_LOADS
#define SL3D_BUF_LOADS 0
#endif
#ifndef SL3D_BUF_AUX
#define SL3D_BUF_AUX 19
#endif
__device__ __forceinline__ unsigned ldb32(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned plane_off)
{
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)plane_off, SL3D_BUF_AUX);
}
__device__ __forceinline__ unsigned ldg32(const GLOBAL_AS uint8_t *base, unsigned off)
{
    asm volatile("" : "+s"(base));
    if (SL3D_NT_LOADS) return __builtin_nontemporal_load((const GLOBAL_AS unsigned *)(base + (size_t)off));
    return *(const GLOBAL_AS unsigned *)(base + (size_t)off);
}


// ------------------------------------------------------------------------------------------------
// Ordered compaction inside the fused kernel (O1 / N2: 8/save_point_cloud.cpp:33-37 counts the valid pixels, :85-104 appends
// them in row-major scan order): single pass, decoupled look-back over the 1024-pixel tiles of a view.
// A tile (= one block) publishes the number of valid pixels it found in a view as an AGGREGATE word, later its inclusive
// PREFIX; a tile's exclusive prefix is the sum of the aggregates of its predecessors back to the nearest prefix.
// Status word: epoch << 34 | flag << 32 | count -- one naturally aligned 8-byte word written by ONE agent-scope store and
// polled with agent-scope loads, so it needs no fence (data and tag travel together); words of an older launch generation
// (epoch) read as "not ready", so the array is never cleared between launches.  Tiles are chained in blockIdx.x order
// inside one view; a tile only ever waits for tiles with a lower linear block index, which the dispatcher started earlier.
// ------------------------------------------------------------------------------------------------
#define SL3D_ST_AGG 1ull
#define SL3D_ST_PREFIX 2ull
#define SL3D_ST_FAILED 3ull /* a look-back gave up (time-out): published INSTEAD of a prefix, and contagious -- whoever sees it gives up
                               at once and passes it on, so a failed launch ends quickly and never hands out a made-up prefix */
#define SL3D_LOOKBACK_SPINS (1 << 22) /* polls before a look-back gives up and raises KParams::lookback_err (seconds) */

__device__ __forceinline__ unsigned long long status_word(unsigned epoch, unsigned long long flag, unsigned count)
{
    return ((unsigned long long)epoch << 34) | (flag << 32) | (unsigned long long)count;
}
__device__ __forceinline__ void status_publish(unsigned long long *w, unsigned long long v)
{
    __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned wave_sum(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// exclusive prefix of tile `tile` in one view's status row; executed by ONE whole wave.  A round looks at the
// SL3D_LB_LANES * SL3D_LB_WORDS nearest predecessors (word k of lane j: tile hi - k*LANES - j, so every load is one contiguous
// run).  Measured on the 16 x 1080p batch (profiles/README.md, round 2): the nearest known prefix is ~10-20 tiles back, so a
// small window suffices, and what a look-back costs is its polls -- agent-scope 8-byte loads that go to the memory side of the
// L2 every time -- and above all WAITING for predecessor tiles that are still computing the view: with the look-back right
// behind the tile's own count 57 % of the calls had to wait (3.6 us per call); deferred by one whole view (the points wait in registers) 19 %
// (1.5 us, one round trip that rides behind the next view's plane loads).  Wider windows only add polls: 64 lanes -3 %,
// 256 tiles per round -25 %.
#ifndef SL3D_LB_WORDS
#define SL3D_LB_WORDS 1
#endif
// COMPACT kernel, where a block's work item comes from: 0 = blockIdx (relies on in-order dispatch), 1 = one ticket per block
// (blocks come and go as in the dense kernel, but a look-back can never wait for a tile that has not started), 2 = persistent
// blocks that keep drawing tickets
// COMPACT kernel: views between a tile's count and its look-back: 1 = one (points wait in registers), 2 = two (a second LDS
// staging area in between; fits beside the first at 3 blocks per CU)
#ifndef SL3D_SLACK
#define SL3D_SLACK 2
#endif
#ifndef SL3D_PERSIST
#define SL3D_PERSIST 1
#endif
#ifndef SL3D_LB_SLEEP
#define SL3D_LB_SLEEP 8 /* x64 clocks between two polls of a window that is not ready */
#endif
#ifndef SL3D_LB_LANES
#define SL3D_LB_LANES 16 /* lanes of the wave that poll (a round covers SL3D_LB_LANES * SL3D_LB_WORDS tiles) */
#endif
#if SL3D_CX & 64
#define SL3D_LB_STATS_ARG , unsigned long long (&g_lb_stats)[6]
#define SL3D_LB_STATS_PASS , lb_stats
#else
#define SL3D_LB_STATS_ARG
#define SL3D_LB_STATS_PASS
#endif
// the lane's status words of the first look-back window of tile `tile`: requested early, consumed later
struct LbWords {
    unsigned long long w[SL3D_LB_WORDS];
};
__device__ __forceinline__ LbWords lookback_poll(const unsigned long long *row, int tile, unsigned epoch)
{
    const int lane = (int)(threadIdx.x & 63u);
    LbWords r;
#pragma unroll
    for (int k = 0; k < SL3D_LB_WORDS; k++) {
        const int idx = tile - 1 - SL3D_LB_LANES * k - lane;
        r.w[k] = lane < SL3D_LB_LANES ? status_word(epoch, SL3D_ST_PREFIX, 0u) : status_word(epoch, SL3D_ST_AGG, 0u);
        if (idx >= 0 && lane < SL3D_LB_LANES) r.w[k] = __hip_atomic_load(row + (size_t)idx * SL3D_ST_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return r;
}

// have_first: `first` holds the words lookback_poll fetched for the first window (no load for that round)
// failed: the look-back timed out or met a SL3D_ST_FAILED word; the return value is then meaningless (the launch is void and
// reported through KParams::lookback_flag) and the caller publishes SL3D_ST_FAILED instead of a prefix
__device__ __forceinline__ unsigned tile_lookback(const unsigned long long *row, int tile, unsigned epoch, int *err, bool have_first,
                                                  const LbWords &first, bool &failed SL3D_LB_STATS_ARG)
{
    failed = false;
    const int lane = (int)(threadIdx.x & 63u);
    unsigned sum = 0;
    int hi = tile - 1;  // nearest predecessor of the current window
    int spins = 0;
#if SL3D_CX & 64
    const unsigned long long t_begin = wall_clock64();
    unsigned rounds = 0;
    auto stats = [&](unsigned result) {
        g_lb_stats[0] += 1ull;
        g_lb_stats[1] += (unsigned long long)rounds;
        g_lb_stats[2] += (unsigned long long)spins;
        g_lb_stats[3] += wall_clock64() - t_begin;
        g_lb_stats[4] += (unsigned long long)(tile - 1 - hi);
        if (spins > 0) g_lb_stats[5] += 1ull;
        return result;
    };
#else
    auto stats = [&](unsigned result) { return result; };
#endif
    for (;;) {
#if SL3D_CX & 64
        rounds++;
#endif
        unsigned long long w[SL3D_LB_WORDS];
#pragma unroll
        for (int k = 0; k < SL3D_LB_WORDS; k++) {
            const int idx = hi - SL3D_LB_LANES * k - lane;  // word k of every lane: one contiguous run of tiles per load
            // tiles before the first one: an inclusive prefix of 0; lanes beyond the polling window: an empty aggregate
            w[k] = lane < SL3D_LB_LANES ? status_word(epoch, SL3D_ST_PREFIX, 0u) : status_word(epoch, SL3D_ST_AGG, 0u);
            if (have_first) w[k] = first.w[k];
            else if (idx >= 0 && lane < SL3D_LB_LANES) w[k] = __hip_atomic_load(row + (size_t)idx * SL3D_ST_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        have_first = false;
        // nearest first = word 0 of lanes 0..L-1, then word 1 of lanes 0..L-1, ...: walk the words until one holds a prefix
        bool done = false, retry = false;
        unsigned add = 0;
#pragma unroll
        for (int k = 0; k < SL3D_LB_WORDS; k++) {
            if (!done && !retry) {
                const unsigned flag = (unsigned)(w[k] >> 32) & 3u;
                const bool ready = (unsigned)(w[k] >> 34) == epoch && flag != 0u;
                const unsigned long long R = __ballot(ready), Pm = __ballot(ready && flag == (unsigned)SL3D_ST_PREFIX);
                if (__ballot(ready && flag == (unsigned)SL3D_ST_FAILED) != 0ull) {  // a predecessor gave up: so does this tile
                    failed = true;
                    if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    return stats(0u);
                }
                if (Pm != 0ull) {
                    const int p = __ffsll((long long)Pm) - 1;  // the lane that holds the nearest known inclusive prefix
                    const unsigned long long need = p == 63 ? ~0ull : ((1ull << (p + 1)) - 1ull);
                    if ((R & need) == need) {
                        add += lane <= p ? (unsigned)w[k] : 0u;
                        done = true;
                    } else retry = true;
                } else if (R == ~0ull) add += (unsigned)w[k];  // a run of aggregates: add them, go on to the next word
                else retry = true;
            }
        }
        if (done) return stats(sum + wave_sum(add));
        if (!retry) {  // the whole window held aggregates and no prefix: look further back
            sum += wave_sum(add);
            hi -= SL3D_LB_LANES * SL3D_LB_WORDS;
            continue;
        }
        if (++spins > SL3D_LOOKBACK_SPINS) {  // never expected: report instead of hanging the GPU
            if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (host-mapped flag)
            failed = true;
            return sum;
        }
        __builtin_amdgcn_s_sleep(SL3D_LB_SLEEP);
    }
}

// one pixel, everything after the byte loads: stage 4 unwrap, stage 5, stage 7, stage 8 cast.
// (cu,cv) = undistorted camera pixel coordinates of this pixel (T1, depends on the pixel only);
// (wv,wh) = wrapped phases, already shifted by +Pi where stage 4 shifts them.
struct PixelResult {
    float x, y, z;
    bool valid;
};

// RIG (stage 7 of the timed fused kernel, chosen by launch_fused from the calibration):
//   0  general: any K, any distortion, everything evaluated in the kernel with the reference's operation order
//      (also what the parity mode and the per-stage kernels run)
//   1  camera K upper triangular + affine (fx, skew, fy, cx, cy), projector without distortion and with a plain K (the reference's
//      own calibration): camera-frame least squares, the projector point is the correspondence itself
//   2  the same camera, any other projector: camera-frame least squares; the undistorted projector point comes from the
//      per-calibration table KParams::proj_disp (one float2 displacement per projector pixel, built by k_proj_table with
//      the same 5-iteration undistortion) -- the reference also tabulates it (7/triangulation.cpp:363-378), per scan
template <bool KEEP, int RIG, typename CalP>
__device__ __forceinline__ PixelResult pixel_chain(const KParams &P, CalP Cp, const PinnedRows &PR, int gx, int gy, double cu, double cv,
                                                   float wv, float wh, int code_v, int code_h, size_t keep_off)
{
    PixelResult R;
    const float nanv = __builtin_nanf("");
    R.x = R.y = R.z = nanv;
    // stage 4: the unwrap skips the first/last column (v) or row (h) of the frame; unwrapped stays unset (0 here)
    const bool in_v = gx >= 1 && gx <= P.fullW - 2;  // 4/phase_unwrap.cpp:285
    const bool in_h = gy >= 1 && gy <= P.fullH - 2;  // 4/phase_unwrap.cpp:304
    float uvv = unwrap_value(wv, code_v), uhv = unwrap_value(wh, code_h);
    if (!KEEP) {
        // timed mode: keep the two phase chains out of divergent branches (the optimiser would sink each atan2 into
        // its own `if (in range)` block and serialise them) so that they interleave in one basic block
        asm volatile("" : "+v"(uvv));
        asm volatile("" : "+v"(uhv));
    }
    float uv = in_v ? uvv : 0.0f;
    float uh = in_h ? uhv : 0.0f;
    if (!KEEP) {  // select the 32-bit value (the optimiser would move the select behind the conversion to double: 2 ops each)
        asm volatile("" : "+v"(uv));
        asm volatile("" : "+v"(uh));
    }
    long cx, cy;
    double cxd, cyd;
    const bool okx = correspond(uv, P.fwv, P.PW, cx, cxd);
    const bool oky = correspond(uh, P.fwh, P.PH, cy, cyd);
    R.valid = okx && oky;
    if (KEEP) {
        P.wrapped[0][keep_off] = wv;
        P.wrapped[1][keep_off] = wh;
        P.unwrapped[0][keep_off] = uv;
        P.unwrapped[1][keep_off] = uh;
        P.code[0][keep_off] = code_v;
        P.code[1][keep_off] = code_h;
        // rejected pixels are never compared; store 0 for those
        P.cpmap[2 * keep_off + 0] = R.valid ? cx : 0;
        P.cpmap[2 * keep_off + 1] = R.valid ? cy : 0;
    }
    if (KEEP ? R.valid : true) {  // timed mode: branch-free (an invalid pixel's result is discarded by the caller)
        double up, vp, X[3];
        bool singular = false;
        const auto &C = *Cp;
        if (RIG == 1) {
            // the projector's undistort + re-project is fx*((x-cx)*(1/fx)) + cx, i.e. x itself up to 2-3 ulp (1e-13 px),
            // and (cu,cv) are the camera's undistorted NORMALISED coordinates for the camera-frame solve
            triangulate_camframe(C, PR, cu, cv, cxd, cyd, X, singular);
        } else if (RIG == 2) {
            // neighbouring camera pixels see neighbouring projector pixels: the gather stays within a few cache lines
            // per wave (cx, cy are 0 for a rejected pixel, whose result is discarded)
            const float2 d = P.proj_disp[(size_t)(int)cy * (size_t)P.PW + (size_t)(int)cx];
            triangulate_camframe(C, PR, cu, cv, cxd + (double)d.x, cyd + (double)d.y, X, singular);
        } else {
            if (C.proj.identity) {
                up = cxd;
                vp = cyd;
            } else if (!KEEP && P.proj_disp) {  // timed mode: the per-calibration table of the same values (see RIG 2)
                const float2 d = P.proj_disp[(size_t)(int)cy * (size_t)P.PW + (size_t)(int)cx];
                up = cxd + (double)d.x;
                vp = cyd + (double)d.y;
            } else {
                undistort_reproject(cxd, cyd, C.proj, up, vp);
            }
            triangulate_px(C, PR, cu, cv, up, vp, X);
        }
        R.x = (float)X[0];  // 8/save_point_cloud.cpp:100-102
        R.y = (float)X[1];
        R.z = (float)X[2];
        if (RIG != 0 && singular) R.x = R.y = R.z = 0.0f;  // cvInvert's zero matrix: V = 0
        if (KEEP) {
            P.ipoints[3 * keep_off + 0] = X[0];
            P.ipoints[3 * keep_off + 1] = X[1];
            P.ipoints[3 * keep_off + 2] = X[2];
        }
    }
    return R;
}

// The two halves of pixel_chain as the timed kernels use them (SL3D_SPLIT): stages 4 + 5 of one pixel -> its correspondence
// (bit-exact chain, both axes in one basic block), and stage 7 + the cast of stage 8 from that correspondence.
__device__ __forceinline__ bool correspond_px(const KParams &P, int gx, int gy, float wv, float wh, int code_v, int code_h, int &cx, int &cy)
{
    const bool in_v = gx >= 1 && gx <= P.fullW - 2;  // 4/phase_unwrap.cpp:285
    const bool in_h = gy >= 1 && gy <= P.fullH - 2;  // 4/phase_unwrap.cpp:304
    float uvv = unwrap_value(wv, code_v), uhv = unwrap_value(wh, code_h);
    asm volatile("" : "+v"(uvv));
    asm volatile("" : "+v"(uhv));
    float uv = in_v ? uvv : 0.0f;
    float uh = in_h ? uhv : 0.0f;
    asm volatile("" : "+v"(uv));
    asm volatile("" : "+v"(uh));
    long lx, ly;
    double dxd, dyd;
    const bool okx = correspond(uv, P.fwv, P.PW, lx, dxd);
    const bool oky = correspond(uh, P.fwh, P.PH, ly, dyd);
    cx = (int)lx;
    cy = (int)ly;
    return okx && oky;
}

template <int RIG, typename CalP>
__device__ __forceinline__ void triangulate_from(const KParams &P, CalP Cp, const PinnedRows &PR, double cu, double cv, int cx, int cy, float2 d, bool table,
                                                 float &x, float &y, float &z)
{
    const auto &C = *Cp;
    const double cxd = (double)cx, cyd = (double)cy;
    double X[3];
    bool singular = false;
    if (RIG == 1) {
        triangulate_camframe(C, PR, cu, cv, cxd, cyd, X, singular);
    } else if (RIG == 2) {
        triangulate_camframe(C, PR, cu, cv, cxd + (double)d.x, cyd + (double)d.y, X, singular);
    } else {
        double up = cxd, vp = cyd;
        if (table) {
            up = cxd + (double)d.x;
            vp = cyd + (double)d.y;
        } else if (!C.proj.identity) {
            undistort_reproject(cxd, cyd, C.proj, up, vp);
        }
        triangulate_px(C, PR, cu, cv, up, vp, X);
    }
    x = (float)X[0];  // 8/save_point_cloud.cpp:100-102
    y = (float)X[1];
    z = (float)X[2];
    if (RIG != 0 && singular) x = y = z = 0.0f;  // cvInvert's zero matrix: V = 0
}

// grid.x covers the quads (4 pixels) of one window, grid.y covers groups of `vpt` views: a lane keeps
// its 4 pixels and walks through the views of its group, so the camera-side undistortion (the most
// expensive per-pixel constant of stage 7) is computed once per pixel, not once per pixel per view.
//
// Memory-level parallelism: all 2F+2Nv+2Nh plane dwords of a view are requested back to back before
// the first one is consumed (NMAX is the compile-time unroll bound of the Gray planes; an axis with fewer
// planes skips the surplus loads through a wave-uniform test).
// A wave therefore has ~12 KiB of HBM requests in flight instead of a round trip per pair of bit planes.
// Every plane is addressed as (wave-uniform 64-bit plane base) + (one 32-bit lane offset).
//
// The 4 pixels of a lane are processed by a ROLLED loop (one copy of the fp64 chain, low VGPR count);
// per-pixel operands are picked by shifts / selects, and the 48 B of xyz a lane produces are staged
// through LDS so they leave as three 16-B stores per lane (a wave writes 3 KiB contiguous).
//
// FGEN = false: 3-step fringes (the reference's configuration) with the F test folded at compile time.
//
// COMPACT = true (sl3d_run_clouds, timed mode only): instead of the dense xyz plane the kernel writes the compacted cloud of
// every view -- the valid points in row-major scan order (8/save_point_cloud.cpp:85-104) -- in the same pass: a block is a
// 1024-pixel tile of the scan, tile prefixes come from a decoupled look-back (tile_lookback), and the points of view v
// leave while the planes of view v+1 are in flight (their look-back overlaps that latency).  The valid map is still written.
// the COMPACT kernel keeps a second view's points in registers: 3 waves per SIMD leave it 168 VGPRs (150 used,
// no scratch); squeezed into the 128 of 4 waves per SIMD it spills 88 bytes per lane and loses 15 %
#ifndef SL3D_OCC_COMPACT
#define SL3D_OCC_COMPACT 3
#endif
// CMODE = 2 (sl3d_run_clouds, the default): the SEGMENTED ordered cloud -- no dependency between tiles at all.  A wave owns 256
// consecutive pixels of the scan; it compacts ITS valid points (4 ballots + mbcnt, no block barrier, no LDS exchange) into its
// own fixed slot of the cloud buffer -- points [256*seg, 256*seg + count) with seg = 4*tile + wave -- and stores the count.
// Scan order is preserved inside a segment and across segments, so the cloud of a view is the concatenation of its segments;
// k_compact_scan turns the counts into offsets, and the consumers that exist anyway close the gaps while they do their own
// work (k_seg_close into a contiguous device / mapped host buffer, k_register_seg, the pack before an RCCL send).
// Same traffic as the look-back kernel (47 + 1 + 12*valid_fraction B/px), none of its waiting.
#ifndef SL3D_SEG_LDS
#define SL3D_SEG_LDS 1 /* 1: a wave compacts its points inside its own 3 KB of the LDS staging area and stores whole 16-byte chunks (coalesced); 0: 12-byte stores per point */
#endif
#define SL3D_SEG_POINTS 256 /* pixels (point slots) per segment = one wave of the fused kernel */
// RCPT = false: the instantiation for SMALL launches (a handful of views: the reference's one scan per call): 1/d of the atan2
// quotient by v_rcp_f64 + one Newton step instead of the LDS table, whose fill (768 IEEE divisions and a block barrier per block)
// nothing amortises when a block lives for one or two views.  Both ways are proven equal to the host's libm on the whole lattice
// by the device self-check.  Round 3, alternating on one box: 1 view 30.3 against 31.6 us (rocprofv3 kernel durations), 2 views
// -2.7 %, 4 views -1 %; at 16 views per launch the table is as fast (dense) or 1.4 % faster (clouds) -- profiles/r03_rcp_table_ab*.txt.
template <bool KEEP, int NMAX, bool FGEN, bool EXACT, int RIG, int CMODE = 0, bool RCPT = true>
__global__ __launch_bounds__(SL3D_BLOCK, CMODE == 1 ? SL3D_OCC_COMPACT : SL3D_OCC) void k_fused(const KParams P, const DevCal *__restrict__ Cglobal, int first_view, int n_views, int vpt)
{
    constexpr bool COMPACT = CMODE == 1;  // the single-pass look-back compaction (everything named COMPACT below)
    constexpr bool SEG = CMODE == 2;      // the segmented compaction
    static_assert(!(KEEP && CMODE != 0), "the parity mode writes dense planes");
    static_assert(!SEG || SL3D_BLOCK == 256, "a segment is one wave of a 256-thread block: 4 segments per 1024-pixel tile");
    static_assert(!COMPACT || SL3D_BLOCK != 256 || !SL3D_XCD_BANDS, "the look-back chains 1024-pixel tiles in ticket order");  // (other block sizes: A/B builds of the dense kernel only)
    __shared__ __attribute__((aligned(16))) float s_xyz[SL3D_BLOCK * 12];
    // COMPACT: valid pixels per wave of the current view, double-buffered by the parity of the block's view counter (a wave that
    // runs ahead into the next view writes the OTHER half; it cannot reach the view after that before every wave has passed the
    // next view's barrier, i.e. has read this half); exclusive prefix of the tile being flushed
    __shared__ unsigned s_wtot[2][4], s_base;
    unsigned wt_par = 0;
    // COMPACT with SL3D_SLACK >= 2: SL3D_SLACK - 1 more staging areas, for the views that wait between the pixel loop and the registers
    constexpr int NMID = COMPACT ? SL3D_SLACK - 1 : 0;
    __shared__ __attribute__((aligned(16))) float s_mid[NMID > 0 ? NMID * SL3D_BLOCK * 12 : 4];
    __shared__ __attribute__((aligned(16))) double s_cam[SL3D_BLOCK * 8];  // undistorted camera coordinates of the lane's 4 pixels

    // (the compacting kernel with three views of slack gives the table's 6 KB to its staging areas and computes 1/d: -1 %)
    constexpr bool RCP_TAB = RCPT && SL3D_RCP_LDS != 0 && !(COMPACT && SL3D_SLACK >= 3);
    __shared__ __attribute__((aligned(16))) double s_rcp[RCP_TAB ? SL3D_RCP_TAB : 1];  // 1/d for the atan2 quotient
    SL3D_STAMP(0);
#ifdef SL3D_MEASURE
    // experiment (tools/ab.sh, env SL3D_STAGGER = mode * 256 + units): the blocks of the FIRST round (the ones that find the machine
    // empty) start `slot * units` sleeps of ~0.45 us late, slot = the block's generation on its CU taken from the dispatch order
    // (mode 0: blockIdx.x / n_cus) or the wave's slot on its SIMD (mode 1: HW_ID.wave_id) -- so that the first slots' planes land
    // early and their arithmetic runs under the later slots' loads instead of every wave of the round loading at once
    if (P.stagger != 0 && CMODE == 0 && blockIdx.y == 0 && blockIdx.x < (unsigned)(P.n_cus > 0 ? P.n_cus : 256) * 4u) {
        const unsigned units = (unsigned)P.stagger & 255u, mode = (unsigned)P.stagger >> 8;
        const unsigned slot = mode == 0 ? blockIdx.x / (unsigned)(P.n_cus > 0 ? P.n_cus : 256) : (__builtin_amdgcn_s_getreg(6148) & 3u);  // HW_REG_HW_ID[3:0]
        for (unsigned i = 0; i < slot * units; i++) __builtin_amdgcn_s_sleep(16);
    }
#endif
    if (RCP_TAB) {
        fill_rcp_table(s_rcp);
        __syncthreads();
    }
    SL3D_STAMP(1);
    const int F = FGEN ? P.F : 3;
    const int qpr = P.pitch >> 2;  // quads per row, pitch padding included
    // A work ITEM is one 1024-pixel tile (256 lanes x 4 pixels) of the window for one group of `vpt` views.
    //   dense kernels : one item per block, item = (blockIdx.x, blockIdx.y); gridDim.x is a multiple of 8 (launch_fused),
    //                   so blockIdx.x % 8 is the XCD whatever blockIdx.y is
    //   COMPACT       : the blocks are persistent and draw items from a ticket counter (item k = tile k % n_tiles of view
    //                   group k / n_tiles): whoever holds ticket k started after tickets < k were handed out, so a look-back
    //                   only ever waits for tiles that are running or done -- no assumption on the dispatch order -- and the
    //                   flush pipeline (two views behind the one being computed) runs on across items, so a block waits for
    //                   its predecessors' last counts only once, at the very end of the kernel
    unsigned tile = 0;
    int row_q = 0, cq = 0, row = 0, gx0 = 0, gy = 0, v_begin = 0, v_end = 0;
    bool alive = true;
    unsigned lane_off = 0;  // byte offset of the quad inside any plane
    const float nanv = __builtin_nanf("");
    float *my_xyz = s_xyz + threadIdx.x * 12;
    const unsigned ps = (unsigned)P.plane_stride;
    double *my_cam = s_cam + threadIdx.x * 8;
    __shared__ unsigned s_ticket[2];
    unsigned item = 0, n_items = 0, item_parity = 0;
    auto take_ticket = [&]() -> unsigned { return atomicAdd(P.ticket, 1u) - P.ticket_base; };
    if (COMPACT) {
        n_items = (unsigned)P.n_tiles * (unsigned)((n_views + vpt - 1) / vpt);
        if (SL3D_PERSIST == 0) {
            item = blockIdx.x;
        } else {
            if (threadIdx.x == 0) s_ticket[0] = take_ticket();
            __syncthreads();
            item = __builtin_amdgcn_readfirstlane(s_ticket[0]);  // block-uniform: keep it in a scalar register
        }
    }
    // everything of an item that depends on the pixel only; false if this lane has nothing to do in a dense kernel
    MaskQuad mq_first = {0u};
    // EARLY (the small-launch instantiation of the pipelined dense kernels): the first view's planes are requested UNCONDITIONALLY,
    // right behind the item's mask / camera-table requests and before any of those is waited for -- one round trip instead of two
    // (mask -> valid bits -> plane loads) in front of the first decode, at the price of plane loads for quads that turn out to be
    // masked off.  It pays where a block lives for one or two views (SL3D_EARLY_PLANES, profiles/README.md).
    constexpr bool EARLY = SL3D_EARLY_PLANES && !RCPT && CMODE == 0 && SL3D_PIPE && SL3D_SPLIT && !KEEP && RIG != 0;
    double camt[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // EARLY: the lane's camera-table entries between their request and finish_cam
    auto begin_item = [&](unsigned tile_, int group) -> bool {
        tile = tile_;
        v_begin = first_view + group * vpt;  // (block-uniform values first: nothing below may make them look divergent)
        v_end = min(v_begin + vpt, first_view + n_views);
        const long q = (long)tile * SL3D_BLOCK + threadIdx.x;
        row_q = (int)(q / qpr);
        cq = (int)(q - (long)row_q * qpr);
        // COMPACT: a block keeps all its lanes (block barriers in the view loop); lanes past the last row work on a clamped
        // address and have no valid pixel
        // SEG: a wave stores its segment with all 64 lanes (whole 16-byte chunks, lane after lane), so the lanes past the last row
        // stay too, without a valid pixel; only the blocks the grid was padded with leave (they own no segment)
        if (SEG && tile_ >= (unsigned)P.n_tiles) return false;
        if (!COMPACT && !SEG && row_q >= P.H) return false;
        alive = row_q < P.H;
        row = (COMPACT || SEG) ? min(row_q, P.H - 1) : row_q;
        gx0 = P.col0 + cq * 4;
        gy = P.row0 + row;
        lane_off = (unsigned)row * (unsigned)P.pitch + (unsigned)cq * 4u;
        // the valid bits of the item's first view are requested now, so that they travel together with the camera table
        // entries below instead of after them (one round trip less before the first plane loads can leave)
        mq_first = load_mask_quad(P, min(v_begin, first_view + n_views - 1), cq, row);
        // T1 for the camera depends on the pixel only: once per lane and item, kept in LDS so the rolled pixel loop can
        // index it (each lane reads back only what it wrote: no barrier).
        // (Round 3 measured the other order -- these loads requested before the reciprocal-table fill, their entries consumed
        // behind the first view's plane loads, so that no set-up round trip precedes the 11.5 KB of plane requests: one-view
        // launch 31.8-32.2 us against 31.1-31.6, 16 views +-0 (profiles/r03_prologue_ab.txt).  The phase trace says why: what a
        // cold launch waits for in its first 5 us is the memory system's ramp under 4096 waves asking at once, not this
        // dependency.)
        if (EARLY && P.use_cam_table) {  // requested only; finish_cam turns them into coordinates behind the plane requests
            const size_t i0 = (size_t)row * P.pitch + (size_t)cq * 4;
            if (P.use_cam_table == 1) {
                const double2 *tp = (const double2 *)(P.cam_tab + i0);
                const double2 a = tp[0], b = tp[1];
                camt[0] = a.x; camt[1] = a.y; camt[2] = b.x; camt[3] = b.y;
            } else {
                const double2 *tp = (const double2 *)(P.cam_tab + 2 * i0);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double2 a = tp[k];
                    camt[2 * k] = a.x;
                    camt[2 * k + 1] = a.y;
                }
            }
            return true;
        }
        if (P.use_cam_table) {
            // the per-calibration table (k_cam_table) holds what the loop below iterates; the doubles that come out are the same
            const auto &I = opaque_const(Cglobal)->cam;
            const size_t i0 = (size_t)row * P.pitch + (size_t)cq * 4;
            const double y0 = ((double)gy - I.cy) * I.ify;
            double t[8];
            if (P.use_cam_table == 1) {
                const double2 *tp = (const double2 *)(P.cam_tab + i0);
                const double2 a = tp[0], b = tp[1];
                t[0] = a.x; t[1] = a.y; t[2] = b.x; t[3] = b.y;
            } else {
                const double2 *tp = (const double2 *)(P.cam_tab + 2 * i0);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double2 a = tp[k];
                    t[2 * k] = a.x;
                    t[2 * k + 1] = a.y;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                double xn, yn;
                if (P.use_cam_table == 1) {
                    xn = (((double)(gx0 + k)) - I.cx) * I.ifx * t[k];
                    yn = y0 * t[k];
                } else {
                    xn = t[2 * k];
                    yn = t[2 * k + 1];
                }
                if (RIG == 0) reproject(xn, yn, I, xn, yn);
                my_cam[2 * k] = xn;
                my_cam[2 * k + 1] = yn;
            }
            return true;
        }
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            double cu = 0.0, cv = 0.0;
            if (cq * 4 < P.W && !(SL3D_ABLATE_RT(P) & 4)) {
                if (RIG != 0) undistort_normalized((double)(gx0 + k), (double)gy, opaque_const(Cglobal)->cam, cu, cv);  // camera-frame solve
                else undistort_reproject((double)(gx0 + k), (double)gy, opaque_const(Cglobal)->cam, cu, cv);
            }
            my_cam[2 * k] = cu;
            my_cam[2 * k + 1] = cv;
        }
        return true;
    };
    auto finish_cam = [&]() {  // EARLY: the table entries requested by begin_item -> camera coordinates in LDS (same arithmetic)
        if (!P.use_cam_table) return;
        const auto &I = opaque_const(Cglobal)->cam;
        const double y0 = ((double)gy - I.cy) * I.ify;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double xn, yn;
            if (P.use_cam_table == 1) {
                xn = (((double)(gx0 + k)) - I.cx) * I.ifx * camt[k];
                yn = y0 * camt[k];
            } else {
                xn = camt[2 * k];
                yn = camt[2 * k + 1];
            }
            if (RIG == 0) reproject(xn, yn, I, xn, yn);
            my_cam[2 * k] = xn;
            my_cam[2 * k + 1] = yn;
        }
    };

    PinnedRows PR;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        PR.c2[j] = RIG != 0 ? 0.0 : Cglobal->Ac[8 + j];
        PR.p2[j] = RIG != 0 ? Cglobal->Apc[8 + j] : Cglobal->Ap[8 + j];
        if (RIG == 0) asm volatile("" : "+v"(PR.c2[j]));
        asm volatile("" : "+v"(PR.p2[j]));  // stay in VGPRs (see PinnedRows)
        if (j < 3) {
            PR.t[j] = RIG != 0 ? Cglobal->tcn[j] : 0.0;
            if (RIG != 0) asm volatile("" : "+v"(PR.t[j]));
        }
    }
    // EXACT: both axes have exactly NMAX Gray planes (the usual case): the plane clamps and the per-plane tests fold away
    const int Nv = EXACT ? NMAX : P.Nv, Nh = EXACT ? NMAX : P.Nh;
    // ---- building blocks of one view ------------------------------------------------------------------------------
    // planes of a view: vertical axis (fringe F, gray Nv, inverse Nv), then the horizontal axis.  Plane offsets are added
    // to the scalar view base (SALU); every load uses the same 32-bit VGPR offset.  Instruction selection works per basic
    // block: the zero-extension of the lane offset has to happen in the block of the loads for them to select the
    // (SGPR base + 32-bit VGPR offset) form, hence the per-call copy behind an empty asm.
    auto issue_fringe = [&](int view, unsigned (&f)[2][4]) {
        const GLOBAL_AS uint8_t *vb = opaque(P.frames + (size_t)view * P.view_stride);
        const unsigned psv = opaque_u32(ps);  // per-view copy: plane offsets are recomputed, not kept live
        unsigned lo = lane_off;
        asm volatile("" : "+v"(lo));
        if (SL3D_BUF_LOADS) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(const uint8_t *)vb, 0, (int)P.view_stride, 0x00020000);
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const unsigned p0 = a == 0 ? 0u : (unsigned)(F + 2 * Nv) * psv;
                f[a][0] = ldb32(r, lo, p0);
                f[a][1] = ldb32(r, lo, p0 + psv);
                f[a][2] = ldb32(r, lo, p0 + 2u * psv);
                f[a][3] = (FGEN && F == 4) ? ldb32(r, lo, p0 + 3u * psv) : 0u;
            }
            return;
        }
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const unsigned p0 = a == 0 ? 0u : (unsigned)(F + 2 * Nv) * psv;
            f[a][0] = ldg32(vb + (size_t)p0, lo);
            f[a][1] = ldg32(vb + (size_t)(p0 + psv), lo);
            f[a][2] = ldg32(vb + (size_t)(p0 + 2u * psv), lo);
            f[a][3] = (FGEN && F == 4) ? ldg32(vb + (size_t)(p0 + 3u * psv), lo) : 0u;
        }
    };
    auto issue_gray = [&](int view, unsigned (&g)[2][NMAX], unsigned (&iv)[2][NMAX]) {
        const GLOBAL_AS uint8_t *vb = opaque(P.frames + (size_t)view * P.view_stride);
        const unsigned psv = opaque_u32(ps);
        unsigned lo = lane_off;
        asm volatile("" : "+v"(lo));
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)(const uint8_t *)vb, 0, (int)P.view_stride, 0x00020000);
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int N = a == 0 ? Nv : Nh;
            const unsigned pg = (unsigned)((a == 0 ? 0 : F + 2 * Nv) + F) * psv;
#pragma unroll
            for (int i = 0; i < NMAX; i++) {
                // an axis with fewer than NMAX planes: the surplus loads are not issued (wave-uniform test; decode ignores them)
                g[a][i] = iv[a][i] = 0u;
                if (EXACT || i < N) {
                    if (SL3D_BUF_LOADS) {
                        g[a][i] = ldb32(rg, lo, pg + (unsigned)i * psv);
                        iv[a][i] = ldb32(rg, lo, pg + (unsigned)(N + i) * psv);
                    } else {
                        g[a][i] = ldg32(vb + (size_t)(pg + (unsigned)i * psv), lo);
                        iv[a][i] = ldg32(vb + (size_t)(pg + (unsigned)(N + i) * psv), lo);
                    }
                }
            }
        }
    };
    // Gray decode, byte-parallel over the 4 pixels of the lane.
    // G_i = (gray - inverse >= 0) (4/phase_unwrap.cpp:183) for 4 bytes at once: the low 7 bits are compared by a
    // borrow-protected subtraction, bit 7 decides unless the top bits are equal (one v_bitop3 on x, y, t).
    // B_0 = G_0, B_i = B_{i-1} xor G_i (:187-191) is a running xor of the masks; the code sum B_i 2^(N-1-i) (:193) is
    // accumulated per byte, the LAST 8 planes in `lo`, the ones before them in `hi`, so that the 16-bit code of a pixel
    // is (hi byte, lo byte) and one v_perm per pixel pair builds it: code[a][j] = codes of pixels 2j (low half), 2j+1.
    auto decode = [&](const unsigned (&g)[2][NMAX], const unsigned (&iv)[2][NMAX], unsigned (&code)[2][2]) {
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int N = a == 0 ? Nv : Nh;
            const unsigned H = 0x80808080u;
            unsigned bacc = 0;  // running binary bit of pixel k at bit 8k+7
            unsigned hi = 0, lo = 0;
#pragma unroll
            for (int i = 0; i < NMAX; i++) {
                if (i < N) {
                    const unsigned x = g[a][i], y = iv[a][i];
                    const unsigned t = (x | H) - (y & ~H);                           // bit 8k+7: (x & 0x7f) >= (y & 0x7f)
                    const unsigned ge = __builtin_amdgcn_bitop3_b32(x, y, t, 0xB2);  // (x & ~y) | (~(x ^ y) & t): byte x >= byte y
                    bacc = __builtin_amdgcn_bitop3_b32(bacc, ge, H, 0x78);           // bacc ^ (ge & H)
                    if (i < N - 8) hi = (hi << 1) | (bacc >> 7);
                    else lo = (lo << 1) | (bacc >> 7);
                }
            }
            code[a][0] = __builtin_amdgcn_perm(hi, lo, 0x05010400u);  // bytes (lo0, hi0, lo1, hi1); selectors 0-3 = lo, 4-7 = hi
            code[a][1] = __builtin_amdgcn_perm(hi, lo, 0x07030602u);  // bytes (lo2, hi2, lo3, hi3)
        }
    };
    // one pixel of the timed mode: no divergent branch inside (an invalid pixel's result is replaced by NaN at the end)
    // `i` (0 or 1, compile-time after inlining) is the pixel's place in the CURRENT pair: the pair loop shifts the fringe
    // dwords, the code words and the valid bits down after each pair, so every operand sits at a fixed byte / half-word
    // (static sub-dword selects instead of shifts by a loop counter); k = 2*pair + i only addresses LDS and the frame.
    auto pixel = [&](int i, int k, size_t px, unsigned vbits, const unsigned (&f)[2][4], const unsigned (&code)[2][2], unsigned &ok,
                     float &ox, float &oy, float &oz) {
        const int sh = 8 * i;
        if (SL3D_ABLATE & 1) {
            ox = __uint_as_float(((code[0][0] ^ f[0][0] ^ f[1][1]) >> sh) | 0x3f800000u);
            oy = __uint_as_float(((code[1][0] ^ f[0][1] ^ f[1][2]) >> sh) | 0x3f800000u);
            oz = __uint_as_float(((code[0][1] ^ code[1][1] ^ f[0][2] ^ f[1][0]) >> sh) | 0x3f800000u);
            ok = 1u;
            return;
        }
        const int code_v = (int)((code[0][0] >> (16 * i)) & 0xffffu);
        const int code_h = (int)((code[1][0] >> (16 * i)) & 0xffffu);
        const AtanK AK = atan_consts<true>();
        float wv = wrapped_phase<RCP_TAB>(F, (f[0][0] >> sh) & 255, (f[0][1] >> sh) & 255, (f[0][2] >> sh) & 255, (f[0][3] >> sh) & 255, s_rcp, AK);
        float wh = wrapped_phase<RCP_TAB>(F, (f[1][0] >> sh) & 255, (f[1][1] >> sh) & 255, (f[1][2] >> sh) & 255, (f[1][3] >> sh) & 255, s_rcp, AK);
        // stage 4 shifts by +Pi only inside its loop range (4/phase_unwrap.cpp:285,290,304,308); outside it the
        // unwrapped value is 0 whatever the wrapped one is (pixel_chain), and the timed mode does not keep wrapped
        wv = shift_pi(wv);
        wh = shift_pi(wh);
        const double cu = my_cam[2 * k], cv = my_cam[2 * k + 1];
        const PixelResult R = pixel_chain<false, RIG>(P, opaque_const(Cglobal), PR, gx0 + k, gy, cu, cv, wv, wh, code_v, code_h, px + k);
        const bool okpx = ((vbits >> i) & 1u) && R.valid;
        ox = okpx ? R.x : nanv;
        oy = okpx ? R.y : nanv;
        oz = okpx ? R.z : nanv;
        ok = okpx ? 1u : 0u;
    };
    // the 4 pixels of a lane as two pairs: two independent fp64 dependency chains per iteration for the scheduler to
    // interleave (tools/ab.sh: 1 pixel per iteration -3 %, all 4 unrolled +1 % but 12 more VGPRs)
    auto pixel_pairs = [&](size_t px, unsigned vbits, unsigned (&f)[2][4], unsigned (&code)[2][2]) -> unsigned {
        unsigned vout = 0;
        SL3D_UNROLL(SL3D_PAIR_UNROLL)
        for (int j = 0; j < 2; j++) {
            unsigned ok0, ok1;
            pixel(0, 2 * j, px, vbits, f, code, ok0, my_xyz[6 * j + 0], my_xyz[6 * j + 1], my_xyz[6 * j + 2]);
            pixel(1, 2 * j + 1, px, vbits, f, code, ok1, my_xyz[6 * j + 3], my_xyz[6 * j + 4], my_xyz[6 * j + 5]);
            vout = (vout >> 16) | (ok0 << 16) | (ok1 << 24);  // after two pairs: valid byte of pixel k at byte k
#pragma unroll
            for (int a = 0; a < 2; a++) {
#pragma unroll
                for (int p = 0; p < 4; p++) f[a][p] >>= 16;
                code[a][0] = code[a][1];
            }
            vbits >>= 2;
        }
        return vout;
    };
    // ---- the same work cut in two phases (SL3D_SPLIT): A = stages 3..5 of the lane's 4 pixels, the correspondences parked in the
    // LDS staging area (slots 3k, 3k+1 of pixel k, which its own result overwrites later); B = stage 7.  Between the two the
    // plane registers are dead -- that is where a table rig asks for its 4 projector-table entries at once (instead of one
    // dependent gather inside every pixel's chain), and where the pipelined loop requests the next view's planes.
    int *my_cp = (int *)my_xyz;
    auto pixel_A = [&](int i, int k, unsigned vbits, const unsigned (&f)[2][4], const unsigned (&code)[2][2]) -> unsigned {
        const int sh = 8 * i;
        const int code_v = (int)((code[0][0] >> (16 * i)) & 0xffffu);
        const int code_h = (int)((code[1][0] >> (16 * i)) & 0xffffu);
        const AtanK AK = atan_consts<true>();
        float wv = wrapped_phase<RCP_TAB>(F, (f[0][0] >> sh) & 255, (f[0][1] >> sh) & 255, (f[0][2] >> sh) & 255, (f[0][3] >> sh) & 255, s_rcp, AK);
        float wh = wrapped_phase<RCP_TAB>(F, (f[1][0] >> sh) & 255, (f[1][1] >> sh) & 255, (f[1][2] >> sh) & 255, (f[1][3] >> sh) & 255, s_rcp, AK);
        wv = shift_pi(wv);
        wh = shift_pi(wh);
        int cx, cy;
        const bool ok = correspond_px(P, gx0 + k, gy, wv, wh, code_v, code_h, cx, cy) && ((vbits >> i) & 1u);
        my_cp[3 * k] = ok ? cx : 0;  // a rejected pixel keeps a harmless table index
        my_cp[3 * k + 1] = ok ? cy : 0;
        return ok ? 1u : 0u;
    };
    auto phase_A = [&](unsigned vbits, unsigned (&f)[2][4], unsigned (&code)[2][2]) -> unsigned {
        unsigned vout = 0;
#pragma unroll 1
        for (int j = 0; j < 2; j++) {
            const unsigned ok0 = pixel_A(0, 2 * j, vbits, f, code), ok1 = pixel_A(1, 2 * j + 1, vbits, f, code);
            vout = (vout >> 16) | (ok0 << 16) | (ok1 << 24);  // after two pairs: valid byte of pixel k at byte k
#pragma unroll
            for (int a = 0; a < 2; a++) {
#pragma unroll
                for (int p = 0; p < 4; p++) f[a][p] >>= 16;
                code[a][0] = code[a][1];
            }
            vbits >>= 2;
        }
        return vout;
    };
    const bool proj_table = RIG == 2 || (RIG == 0 && !KEEP && P.proj_disp != nullptr);
    auto gather_B = [&](float2 (&d)[4]) {
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = make_float2(0.f, 0.f);
        if (proj_table) {
            // neighbouring camera pixels see neighbouring projector pixels: the 4 gathers stay within a few cache lines per wave
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = P.proj_disp[(size_t)my_cp[3 * k + 1] * (size_t)P.PW + (size_t)my_cp[3 * k]];
        }
    };
    auto phase_B = [&](unsigned vout, float2 (&d)[4]) {
        unsigned vb = vout;
#pragma unroll 1
        for (int j = 0; j < 2; j++) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int k = 2 * j + i;
                float x, y, z;
                triangulate_from<RIG>(P, opaque_const(Cglobal), PR, my_cam[2 * k], my_cam[2 * k + 1], my_cp[3 * k], my_cp[3 * k + 1], d[i], proj_table, x, y, z);
                const bool ok = ((vb >> (8 * i)) & 1u) != 0u;
                my_xyz[3 * k + 0] = ok ? x : nanv;
                my_xyz[3 * k + 1] = ok ? y : nanv;
                my_xyz[3 * k + 2] = ok ? z : nanv;
            }
            d[0] = d[2];
            d[1] = d[3];
            vb >>= 16;
        }
    };
    // the 48 B of xyz a lane produces are staged in LDS (the rolled pixel loop indexes them).  A full wave reads its 3 KB of the
    // staging area back ACROSS lanes (a wave's LDS instructions execute in order: no barrier) and stores whole 1-KiB runs with the
    // non-temporal hint; a wave that is not whole (the last rows of a window), and the parity mode, store three 16-B pieces per lane,
    // each lane what it wrote itself.
    auto store_quad = [&](size_t px, unsigned vout) {
        float4 *out_xyz = (float4 *)(P.points + 3 * px);
        const float4 *sx = (const float4 *)my_xyz;
        if (!(SL3D_ABLATE & 4) || KEEP || sx[0].x == 12345.f) {
            if (SL3D_NT_STORES && !KEEP) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4 *sv = (const f32x4 *)my_xyz;
                f32x4 *ov = (f32x4 *)out_xyz;
                __builtin_nontemporal_store(sv[0], ov);
                __builtin_nontemporal_store(sv[1], ov + 1);
                __builtin_nontemporal_store(sv[2], ov + 2);
            } else if ((SL3D_COALESCED_STORES || (!RCPT && n_views == 1)) && !KEEP && __ballot(true) == ~0ull) {
                // a full wave's 64 x 48 B of results are 3 KB contiguous in LDS AND in the dense plane (quads are consecutive in the
                // pitch-padded layout): every store instruction writes 1 KiB of whole lines, lane after lane, as the segmented kernel
                // does -- and because they are whole lines they can carry the non-temporal hint: nothing is left for the L2 to merge,
                // the lines stream out instead of sitting dirty in the L2 until they are evicted or the kernel ends.
                // Round 3, alternating (profiles/r03_nt_coalesced_ab.txt): 16 views per launch +6...8 % (0.634 -> 0.683 on a slow box,
                // 0.66-0.67 -> 0.70-0.72 on a fast one), one view 29.2 -> 27.3 us, other rigs +7 %; each half alone LOSES
                // (coalesced without the hint -2.4 %, profiles/r03_coalesced_stores_ab.txt; the hint on the 16-byte pieces below
                // -11 %, round 1: every piece becomes a memory write of its own).
                const unsigned lane_ = threadIdx.x & 63u;
                const float4 *wb4 = (const float4 *)(s_xyz + (threadIdx.x >> 6) * (64u * 12u));
                float4 *o4 = (float4 *)(P.points + 3 * (px - 4u * lane_));
                if (SL3D_NT_COALESCED) {
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    const f32x4 *w4 = (const f32x4 *)wb4;
                    f32x4 *q4 = (f32x4 *)o4;
                    __builtin_nontemporal_store(w4[lane_], q4 + lane_);
                    __builtin_nontemporal_store(w4[64u + lane_], q4 + 64u + lane_);
                    __builtin_nontemporal_store(w4[128u + lane_], q4 + 128u + lane_);
                    __builtin_nontemporal_store(vout, (unsigned *)(P.valid + px));
                    return;
                }
                o4[lane_] = wb4[lane_];
                o4[64u + lane_] = wb4[64u + lane_];
                o4[128u + lane_] = wb4[128u + lane_];
            } else {
                out_xyz[0] = sx[0];
                out_xyz[1] = sx[1];
                out_xyz[2] = sx[2];
            }
        }
        if (SL3D_NT_STORES && !KEEP) __builtin_nontemporal_store(vout, (unsigned *)(P.valid + px));
        else *(unsigned *)(P.valid + px) = vout;
    };
    auto fill_nan = [&]() {
#pragma unroll
        for (int i = 0; i < 12; i++) my_xyz[i] = nanv;
    };
    // SEG: the wave's valid points of this view, compacted in scan order into the wave's own segment of the cloud buffer, and
    // their count.  Needs nothing from any other wave; lanes past the last row take part with no valid pixel.
    auto store_segment = [&](int view, size_t px, unsigned vout) {
        if (alive) {
            if (SL3D_NT_SEG) __builtin_nontemporal_store(vout, (unsigned *)(P.valid + px));
            else *(unsigned *)(P.valid + px) = vout;
        }
        const unsigned long long b0 = __ballot((vout & 0x00000001u) != 0u), b1 = __ballot((vout & 0x00000100u) != 0u),
                                 b2 = __ballot((vout & 0x00010000u) != 0u), b3 = __ballot((vout & 0x01000000u) != 0u);
        auto below = [](unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
        unsigned rank = below(b0) + below(b1) + below(b2) + below(b3);  // valid pixels of the lanes below this one
        const unsigned total = (unsigned)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
        const unsigned lane_ = threadIdx.x & 63u, wave_ = threadIdx.x >> 6;
        const unsigned seg = tile * 4u + wave_;
        if (lane_ == 0u) P.seg_counts[(size_t)view * (size_t)P.n_segs + seg] = total;
        float *slot = P.clouds + 3 * ((size_t)view * P.px_view_stride + (size_t)seg * SL3D_SEG_POINTS);
        if (SL3D_SEG_LDS) {
            // in place, inside the wave's own 3 KB of the staging area: every lane first reads its 12 floats, then writes its
            // valid points at their compacted position (<= its own: a wave's LDS instructions execute in order, so no lane's
            // data is overwritten before it was read), then the wave stores ceil(3*total/4) whole 16-byte chunks, lane after
            // lane: 1 KiB per store instruction.  (A chunk may run up to 3 floats past the last point: still inside the slot.)
            const float4 *sx = (const float4 *)my_xyz;
            const float4 a = sx[0], b = sx[1], c = sx[2];
            const float q[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
            float *wbase = s_xyz + wave_ * (64u * 12u);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((vout >> (8 * k)) & 1u) {
                    wbase[3 * rank + 0] = q[3 * k + 0];
                    wbase[3 * rank + 1] = q[3 * k + 1];
                    wbase[3 * rank + 2] = q[3 * k + 2];
                    rank++;
                }
            const unsigned chunks = (3u * total + 3u) >> 2;
            const float4 *wb4 = (const float4 *)wbase;
            float4 *out4 = (float4 *)slot;
#pragma unroll
            for (int c3 = 0; c3 < 3; c3++) {
                const unsigned i = (unsigned)c3 * 64u + lane_;
                if (i < chunks) {
                    if (SL3D_NT_SEG) {
                        typedef float f32x4 __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(((const f32x4 *)wb4)[i], (f32x4 *)out4 + i);
                    } else {
                        out4[i] = wb4[i];
                    }
                }
            }
        } else {
            typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));
            float *dst = slot + 3 * (size_t)rank;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((vout >> (8 * k)) & 1u) {
                    f32x3 pt;
                    pt.x = my_xyz[3 * k + 0]; pt.y = my_xyz[3 * k + 1]; pt.z = my_xyz[3 * k + 2];
                    *(f32x3 *)dst = pt;
                    dst += 3;
                }
        }
    };

    // F == 5: check_I_mod_criteria's assignment is commented out (3/wrapped_phase.cpp:117-129): nothing is valid
    auto valid_bits = [&](const MaskQuad &m) -> unsigned {
        if ((COMPACT || SEG) && !alive) return 0u;
        return (FGEN && F == 5) ? 0u : (!KEEP && (SL3D_ABLATE & 2)) ? 0xfu : mask_quad_bits(m);
    };
    // ---- COMPACT: up to three views of this lane's loop are in flight behind the one being computed ----------------------------
    //   fresh : the view computed last; its points are still in the LDS staging area (my_xyz), its tile count is published
    //   mid   : (SL3D_SLACK >= 2) the view(s) before it, parked in further staging areas (s_mid)
    //   held  : the oldest; its points sit in 12 registers.  Its first look-back window is requested behind a batch of plane
    //           loads and consumed behind the decode that waits for those planes, one or two whole iterations after its count
    //           was published -- by then its predecessors have normally published theirs, so the look-back finds its words
    //           ready instead of polling for them; then its points are stored at their final, compacted position.
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
#if SL3D_CX & 64
    unsigned long long lb_stats[6] = {0, 0, 0, 0, 0, 0};
#endif
    bool have_fresh = false, have_held = false;
    bool draining = false;  // (measurement builds: -DSL3D_CX=512 skips the look-back of the block's last view)
    int fview = 0, hview = 0;
    unsigned ftile = 0, htile = 0;
    unsigned fvout = 0, frank = 0, ftotal = 0;  // valid bytes of the lane's quad, its exclusive rank inside the tile, the tile's count
    unsigned hvout = 0, hrank = 0, htotal = 0;
    // SL3D_SLACK >= 2: the views in between, oldest last (points in s_mid[k])
    int mview[NMID > 0 ? NMID : 1] = {0};
    unsigned mtile[NMID > 0 ? NMID : 1] = {0}, mvout[NMID > 0 ? NMID : 1] = {0}, mrank[NMID > 0 ? NMID : 1] = {0}, mtotal[NMID > 0 ? NMID : 1] = {0};
    bool have_mid[NMID > 0 ? NMID : 1] = {false};
    float *my_mid = s_mid + (NMID > 0 ? threadIdx.x * 12 : 0);  // area k at my_mid + k * SL3D_BLOCK * 12
    auto any_mid = [&]() {
        bool a = false;
#pragma unroll
        for (int k = 0; k < NMID; k++) a = a || have_mid[k];
        return a;
    };
    float held[12];
    LbWords lb_first = {};  // wave 0: the held view's first look-back window, requested by poll_held
    bool poll_pending = false;
    // fresh (staging area) -> [mid (second staging area) ->] held (registers): called when the held slot is free and the staging
    // area is about to be overwritten.  With the middle stage a view's look-back starts two whole iterations after its count
    // was published instead of one.
    auto hold_fresh = [&]() {
        if (NMID > 0) {
            // the oldest parked view moves into the (free) registers, the others move up one area, the fresh one is parked
            if (have_mid[NMID - 1]) {
                const float4 *sm = (const float4 *)(my_mid + (NMID - 1) * SL3D_BLOCK * 12);
                const float4 a = sm[0], b = sm[1], c = sm[2];
                held[0] = a.x; held[1] = a.y; held[2] = a.z; held[3] = a.w; held[4] = b.x; held[5] = b.y; held[6] = b.z; held[7] = b.w;
                held[8] = c.x; held[9] = c.y; held[10] = c.z; held[11] = c.w;
                hview = mview[NMID - 1]; htile = mtile[NMID - 1]; hvout = mvout[NMID - 1]; hrank = mrank[NMID - 1]; htotal = mtotal[NMID - 1];
                have_held = true;
                have_mid[NMID - 1] = false;
            }
#pragma unroll
            for (int k = NMID - 1; k > 0; k--) {
                if (have_mid[k - 1]) {
                    const float4 *src = (const float4 *)(my_mid + (k - 1) * SL3D_BLOCK * 12);
                    float4 *dst = (float4 *)(my_mid + k * SL3D_BLOCK * 12);
                    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
                    mview[k] = mview[k - 1]; mtile[k] = mtile[k - 1]; mvout[k] = mvout[k - 1]; mrank[k] = mrank[k - 1]; mtotal[k] = mtotal[k - 1];
                    have_mid[k] = true;
                    have_mid[k - 1] = false;
                }
            }
            if (have_fresh) {
                const float4 *sx = (const float4 *)my_xyz;
                float4 *sm = (float4 *)my_mid;
                sm[0] = sx[0]; sm[1] = sx[1]; sm[2] = sx[2];
                mview[0] = fview; mtile[0] = ftile; mvout[0] = fvout; mrank[0] = frank; mtotal[0] = ftotal;
                have_mid[0] = true;
                have_fresh = false;
            }
            return;
        }
        const float4 *sx = (const float4 *)my_xyz;
        const float4 a = sx[0], b = sx[1], c = sx[2];
        held[0] = a.x; held[1] = a.y; held[2] = a.z; held[3] = a.w; held[4] = b.x; held[5] = b.y; held[6] = b.z; held[7] = b.w;
        held[8] = c.x; held[9] = c.y; held[10] = c.z; held[11] = c.w;
        hview = fview; htile = ftile; hvout = fvout; hrank = frank; htotal = ftotal;
        have_held = true;
        have_fresh = false;
    };
    // the first look-back window of the held view is REQUESTED right behind a batch of plane loads and CONSUMED (flush_held)
    // right behind the decode that waits for those planes anyway: its round trip costs nothing unless it has to be repeated
    auto poll_held = [&]() {
        if (wave == 0 && !(SL3D_CX & 1)) lb_first = lookback_poll(P.tile_status + (size_t)hview * (size_t)P.n_tiles * SL3D_ST_STRIDE, (int)htile, P.epoch);
        poll_pending = true;
    };
    auto flush_held = [&]() {
        unsigned long long *row_st = P.tile_status + (size_t)hview * (size_t)P.n_tiles * SL3D_ST_STRIDE;
        if (wave == 0) {
#if SL3D_CX & 128
            if (lane == 0) P.dbg[((size_t)hview * P.n_tiles + htile) * 4 + 1] = wall_clock64();
#endif
            bool lb_failed = false;
            const unsigned base = ((SL3D_CX & 1) || ((SL3D_CX & 512) && draining))
                                      ? htile * 1024u
                                      : tile_lookback(row_st, (int)htile, P.epoch, P.lookback_flag, poll_pending, lb_first, lb_failed SL3D_LB_STATS_PASS);
            if (lane == 0) {
#if SL3D_CX & 128
                P.dbg[((size_t)hview * P.n_tiles + htile) * 4 + 2] = wall_clock64();
                P.dbg[((size_t)hview * P.n_tiles + htile) * 4 + 3] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF814 /* HW_REG_XCC_ID */) | ((unsigned long long)base << 32);
#endif
                s_base = base;
                if (htile != 0u)
                    status_publish(row_st + (size_t)htile * SL3D_ST_STRIDE, lb_failed ? status_word(P.epoch, SL3D_ST_FAILED, 0u) : status_word(P.epoch, SL3D_ST_PREFIX, base + htotal));
                if ((int)htile == P.n_tiles - 1) P.cloud_totals[hview] = (unsigned long long)(base + htotal);
            }
        }
        poll_pending = false;
        if (!(SL3D_CX & 2)) __syncthreads();
        float *dst = P.clouds + 3 * ((size_t)hview * P.px_view_stride + (size_t)(((SL3D_CX & 2) ? htile * 1024u : s_base) + hrank));
        typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((hvout >> (8 * k)) & 1u) {
                f32x3 pt;
                pt.x = held[3 * k + 0]; pt.y = held[3 * k + 1]; pt.z = held[3 * k + 2];
                *(f32x3 *)dst = pt;
                dst += 3;
            }
        have_held = false;
    };

    // The mask of the NEXT view is requested before the current view's planes, so a wave never waits a full memory
    // round trip for 36 bytes before it can ask for its 11.5 KB.  (Going further -- the next view's planes in flight
    // during the pixel loop, landing in the registers the decode has freed -- was measured: 142 VGPRs, 3 waves/SIMD,
    // -6 %; squeezed into 128 with spills, -14 %.  Occupancy hides the latency better than in-wave pipelining.)
    for (;;) {  // items of this block (dense kernels: exactly one)
    unsigned f[2][4], g[2][NMAX], iv[2][NMAX], code[2][2];
    if (COMPACT) {
        if (item >= n_items) break;  // block-uniform
        begin_item(item % (unsigned)P.n_tiles, (int)(item / (unsigned)P.n_tiles));
    } else if (!begin_item(SL3D_XCD_BANDS ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x, (int)blockIdx.y)) {
        return;
    }
    if (EARLY) {  // planes of the first view right behind the set-up requests; then the set-up results are consumed
        issue_fringe(v_begin, f);
        issue_gray(v_begin, g, iv);
        finish_cam();
    }
    SL3D_STAMP(2);
    unsigned next_ticket = 0;
    MaskQuad mq = mq_first;
    // PIPE (timed kernels): the planes of view v+1 are requested in the middle of view v -- after phase A, when the plane
    // registers of view v are dead, before stage 7 -- so a wave's own arithmetic runs under its own memory requests
    // (the general rig's stage 7 is too register-hungry for it: 44 bytes of scratch per lane, -9 %)
    constexpr bool PIPE = SL3D_PIPE && SL3D_SPLIT && !KEEP && (RIG != 0 || SL3D_PIPE_RIG0);
    unsigned vb_next = 0;
    // (Round 3 read the ISA of this loop: the wait-count pass puts an s_waitcnt vmcnt(0) at the pipeline point and at the loop
    // latch -- vmcnt is ONE in-order counter for loads and stores, so the first makes a wave wait for the acknowledgement of the
    // previous view's stores before its next 46 loads leave, the second holds this view's stores back until the next view's
    // planes have landed.  A schedule without either (mask dwords turned into valid bits in front of the stores, an explicit
    // wait in front of the loop so that only VALU values cross the back edge) was built and measured: 16 views +-0.3 %, one view
    // 32.1 against 31.6 us (profiles/r03_mask_early_ab.txt).  Neither wait is on the critical path; the simpler code stays.)
    if (PIPE) {
        vb_next = valid_bits(mq);
        if (v_begin + 1 < v_end) mq = load_mask_quad(P, v_begin + 1, cq, row);
        if (!EARLY && vb_next != 0) {  // (EARLY: they are in flight already)
            issue_fringe(v_begin, f);
            issue_gray(v_begin, g, iv);
        }
    }
    for (int view = v_begin; view < v_end; view++) {
        unsigned vbits;
        if (PIPE) {
            vbits = vb_next;
        } else {
            if (!SL3D_MASK_PREFETCH && view > v_begin) mq = load_mask_quad(P, view, cq, row);
            vbits = valid_bits(mq);
            if (SL3D_MASK_PREFETCH && view + 1 < v_end) mq = load_mask_quad(P, view + 1, cq, row);
        }
        const size_t px = (size_t)view * P.px_view_stride + (size_t)lane_off;  // first pixel of the quad
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
        if (!COMPACT && !SEG && (KEEP || vbits == 0)) fill_nan();

        if (!PIPE && vbits != 0) {
            // every load of the view is issued before the first one is consumed
            issue_fringe(view, f);
            issue_gray(view, g, iv);
        }
        // the ticket of this block's NEXT item is drawn behind the plane loads of the item's first view and handed to the
        // other waves through LDS; they read it after the view loop (a block barrier per view lies in between)
        if (COMPACT && SL3D_PERSIST == 2 && view == v_begin && threadIdx.x == 0) next_ticket = take_ticket();
        // COMPACT: the previous view's points leave now, behind this view's loads (its look-back overlaps their latency)
        if (COMPACT && !PIPE && have_held && !poll_pending) poll_held();  // behind this view's plane loads
        if (COMPACT && SL3D_PERSIST == 2 && view == v_begin && threadIdx.x == 0) s_ticket[(item_parity + 1u) & 1u] = next_ticket;
        if (view == v_begin) SL3D_STAMP(3);
        if (vbits != 0) decode(g, iv, code);  // waits for the planes of this view
        if (view == v_begin) SL3D_STAMP(4);
        if (COMPACT) {
            // the view computed two steps ago leaves (its look-back window arrived with the planes), then the previous view's
            // points move from the staging area -- about to be overwritten -- into registers
            if (have_held) flush_held();
            if (have_fresh || any_mid()) hold_fresh();
        }
        if (vbits != 0) {
            if (KEEP) {
#pragma unroll 1
                for (int k = 0; k < 4; k++) {
                    if ((vbits >> k) & 1u) {
                        const int sh = 8 * k;
                        const int code_v = (int)((code[0][k >> 1] >> (16 * (k & 1))) & 0xffffu);
                        const int code_h = (int)((code[1][k >> 1] >> (16 * (k & 1))) & 0xffffu);
                        // stage 3: wrapped phase of both axes; stage 4 shifts it by +Pi inside its loop range
                        const AtanK AK = atan_consts<true>();
                        float wv = wrapped_phase<RCP_TAB>(F, (f[0][0] >> sh) & 255, (f[0][1] >> sh) & 255, (f[0][2] >> sh) & 255, (f[0][3] >> sh) & 255, s_rcp, AK);
                        float wh = wrapped_phase<RCP_TAB>(F, (f[1][0] >> sh) & 255, (f[1][1] >> sh) & 255, (f[1][2] >> sh) & 255, (f[1][3] >> sh) & 255, s_rcp, AK);
                        wv = shift_pi_if(wv, gx0 + k >= 1 && gx0 + k <= P.fullW - 2);  // 4/phase_unwrap.cpp:285,290
                        wh = shift_pi_if(wh, gy >= 1 && gy <= P.fullH - 2);            // 4/phase_unwrap.cpp:304,308
                        const double cu = my_cam[2 * k], cv = my_cam[2 * k + 1];
                        const PixelResult R = pixel_chain<true, 0>(P, opaque_const(Cglobal), PR, gx0 + k, gy, cu, cv, wv, wh, code_v, code_h, px + k);
                        if (R.valid) {
                            my_xyz[3 * k + 0] = R.x;
                            my_xyz[3 * k + 1] = R.y;
                            my_xyz[3 * k + 2] = R.z;
                            vout |= 1u << (8 * k);
                        }
                    }
                }
            } else if (SL3D_SPLIT) {
                vout = phase_A(vbits, f, code);
            } else {
                vout = pixel_pairs(px, vbits, f, code);
            }
        }
        if (view == v_begin) SL3D_STAMP(5);
        // (Round 3 read the ISA of the table rigs: their 4 projector-table entries are requested BEHIND the next view's 46 plane
        // loads, so -- vmcnt counts in issue order -- phase B starts only once those planes have landed.  Requesting them first,
        // waiting, and parking them in the pixel's spare staging slots was built and measured: distorted rig 77.8-78.2 Gpx/s
        // against 79.0-79.5 for this order (profiles/r03_gather_first_ab.txt).  The wave waits for those planes at the next decode
        // anyway; an L2 round trip of its own in front of the plane issue is what costs.)
        if (PIPE && view + 1 < v_end) {
            vb_next = valid_bits(mq);
            if (view + 2 < v_end) mq = load_mask_quad(P, view + 2, cq, row);
            if (vb_next != 0) {
                issue_fringe(view + 1, f);
                issue_gray(view + 1, g, iv);
            }
        }
        if (COMPACT && PIPE && have_held && !poll_pending) poll_held();  // behind the next view's plane loads
        if (!KEEP && SL3D_SPLIT && vbits != 0) {
            float2 d[4];
            gather_B(d);
            phase_B(vout, d);
        }
        if (SEG) {
            store_segment(view, px, vout);
            continue;
        }
        if (!COMPACT) {
            if (view == v_begin) SL3D_STAMP(6);
            store_quad(px, vout);
            if (view == v_begin) SL3D_STAMP(7);
            continue;
        }
        if (alive) *(unsigned *)(P.valid + px) = vout;
        // rank of the lane's first valid pixel inside the wave (4 ballots, one per pixel of the quad), wave totals through LDS
        const unsigned long long b0 = __ballot((vout & 0x00000001u) != 0u), b1 = __ballot((vout & 0x00000100u) != 0u),
                                 b2 = __ballot((vout & 0x00010000u) != 0u), b3 = __ballot((vout & 0x01000000u) != 0u);
        auto below = [](unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
        const unsigned rank_w = below(b0) + below(b1) + below(b2) + below(b3);
        if (lane == 0) s_wtot[wt_par][wave] = (unsigned)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
        if (!(SL3D_CX & 4)) __syncthreads();
        const unsigned t0 = s_wtot[wt_par][0], t1 = s_wtot[wt_par][1], t2 = s_wtot[wt_par][2], t3 = s_wtot[wt_par][3];
        wt_par ^= 1u;
        ftotal = t0 + t1 + t2 + t3;
        frank = rank_w + (wave > 0 ? t0 : 0u) + (wave > 1 ? t1 : 0u) + (wave > 2 ? t2 : 0u);
        fvout = vout;
        fview = view;
        ftile = tile;
        have_fresh = true;
        // the tile's count becomes visible to its successors right away; the first tile of a view knows its prefix already
#if SL3D_CX & 128
        if (threadIdx.x == 0) P.dbg[((size_t)view * P.n_tiles + tile) * 4 + 0] = wall_clock64();
#endif
        if (threadIdx.x == 0)
            status_publish(P.tile_status + ((size_t)view * (size_t)P.n_tiles + tile) * SL3D_ST_STRIDE, status_word(P.epoch, tile == 0u ? SL3D_ST_PREFIX : SL3D_ST_AGG, ftotal));
    }
    if (!COMPACT || SL3D_PERSIST != 2) break;
    item_parity ^= 1u;
    item = __builtin_amdgcn_readfirstlane(s_ticket[item_parity & 1u]);
    }
    if (COMPACT) {  // drain: the view before last, then the last one
        if (have_held) flush_held();
        while (have_fresh || any_mid()) {  // (block-uniform)
            draining = !have_fresh;
            __syncthreads();  // (inside the view loop a barrier separates two flushes: every wave has read s_base of the previous one)
            hold_fresh();
            if (have_held) flush_held();
        }
#if SL3D_CX & 64
        if (threadIdx.x == 0)
            for (int i = 0; i < 6; i++) atomicAdd((unsigned long long *)(P.lookback_err + 2) + i, lb_stats[i]);
#endif
    }
}

// number of 1024-pixel tiles (= blocks along x that own pixels) of one view
int fused_tiles(const KParams &P)
{
    const long quads = (long)(P.pitch >> 2) * P.H;
    return (int)((quads + SL3D_BLOCK - 1) / SL3D_BLOCK);
}

// Instantiations: the timed 3-step kernel exists for every N = 6..12 with both axes equal (EXACT: plane tests fold away)
// and for the unroll bounds 8 / 12 / 16 otherwise; the parity mode and the 4-/5-step fringes use the bounds only.
// launches of at most this many views take the instantiation without the LDS reciprocal table (dense 3-step timed kernels)
// (re-measured with the streaming stores: 8 views 185.6-187.5 us through it against 183.8-184.7, 16 views +-0: stays at 4)
#ifndef SL3D_SMALL_LAUNCH_VIEWS
#define SL3D_SMALL_LAUNCH_VIEWS 4
#endif
template <bool KEEP, bool FGEN, int RIG, int COMPACT>
static void launch_fused_n(int nv, int nh, dim3 grid, dim3 block, hipStream_t st, const KParams &P, const DevCal *C, int first_view, int n_views, int vpt)
{
    const int nmax = nv > nh ? nv : nh;
    constexpr bool HAS_SMALL = !KEEP && !FGEN && COMPACT == 0 && SL3D_RCP_LDS != 0;  // (the only family that has the second instantiation)
    const bool small = HAS_SMALL && n_views <= SL3D_SMALL_LAUNCH_VIEWS;
#define SL3D_LAUNCH(NM, EX)                                                                                                              \
    do {                                                                                                                                 \
        if (small) hipLaunchKernelGGL((k_fused<KEEP, NM, FGEN, EX, RIG, COMPACT, !HAS_SMALL>), grid, block, 0, st, P, C, first_view, n_views, vpt); \
        else hipLaunchKernelGGL((k_fused<KEEP, NM, FGEN, EX, RIG, COMPACT>), grid, block, 0, st, P, C, first_view, n_views, vpt);         \
    } while (0)
    if (!KEEP && !FGEN && nv == nh && nv >= 6 && nv <= 12) {
        switch (nv) {
        case 6: SL3D_LAUNCH(6, true); break;
        case 7: SL3D_LAUNCH(7, true); break;
        case 8: SL3D_LAUNCH(8, true); break;
        case 9: SL3D_LAUNCH(9, true); break;
        case 10: SL3D_LAUNCH(10, true); break;
        case 11: SL3D_LAUNCH(11, true); break;
        default: SL3D_LAUNCH(12, true); break;
        }
    } else if (nmax <= 8) SL3D_LAUNCH(8, false);
    else if (nmax <= 12) SL3D_LAUNCH(12, false);
    else SL3D_LAUNCH(SL3D_MAX_GRAY, false);
#undef SL3D_LAUNCH
}

template <bool FGEN, int COMPACT>
static void launch_fused_rig(int rig, dim3 grid, dim3 block, hipStream_t st, const KParams &P, const DevCal *C, int first_view, int n_views, int vpt)
{
    if (rig == 1) launch_fused_n<false, FGEN, 1, COMPACT>(P.Nv, P.Nh, grid, block, st, P, C, first_view, n_views, vpt);
    else if (rig == 2 && P.proj_disp) launch_fused_n<false, FGEN, 2, COMPACT>(P.Nv, P.Nh, grid, block, st, P, C, first_view, n_views, vpt);
    else launch_fused_n<false, FGEN, 0, COMPACT>(P.Nv, P.Nh, grid, block, st, P, C, first_view, n_views, vpt);
}

// views per lane: as many as possible up to SL3D_VPT_MAX (amortises the set-up of a block and the camera table entries) while
// the grid still has >= ~8 blocks per CU to balance the tail.  (Rounds 1-2: 8.  With the stores streaming past the L2 the
// optimum moved: 16 views per launch 354.5 us at 4 against 359.6 at 8 and 359.3 at 2, 372.7 at 16; 32 views 700 against 708,
// profiles/r03_vpt_sweep4.txt; on another box, production builds alternating: 360.7-361.7 against 362.6-363.1.)
#ifndef SL3D_VPT_MAX
#define SL3D_VPT_MAX 4
#endif
static int views_per_lane(unsigned bx, int n_views, int cam_table_kind)
{
    int vpt = 1;
    // (a two-double camera table -- tangential terms -- costs a block 16 B/px: those rigs keep 8 views per lane, measured -0.6 % at 4)
    const int cap = cam_table_kind == 2 ? 8 : SL3D_VPT_MAX;
    while (vpt < cap && vpt < n_views && (long)bx * ((n_views + 2 * vpt - 1) / (2 * vpt)) >= 2048) vpt *= 2;
#ifdef SL3D_MEASURE
    if (getenv("SL3D_VPT") && atoi(getenv("SL3D_VPT")) >= 1) vpt = atoi(getenv("SL3D_VPT"));
#endif
    return vpt;
}

// rig: 0 / 1 / 2, see pixel_chain (the host knows the calibration; folded at compile time in the timed kernels).
// compact: 1 = the timed kernel writes contiguous compacted clouds by a decoupled look-back (KParams::clouds / tile_status /
// cloud_totals must be set), 2 = segmented clouds (KParams::clouds / seg_counts), instead of the dense xyz plane (0).
// Returns the hipError_t of THIS launch.
int launch_fused(const KParams &P_, const DevCal *d_cal, int rig, int first_view, int n_views, bool keep, int compact, void *stream,
                 unsigned *tickets_drawn)
{
    KParams P = P_;
    unsigned drawn = 0;
    const long quads = (long)(P.pitch >> 2) * P.H;
    const unsigned bx = ((unsigned)((quads + SL3D_BLOCK - 1) / SL3D_BLOCK) + 7u) & ~7u;  // a multiple of 8: see the tile order in k_fused
    const int vpt = views_per_lane(bx, n_views, P.cam_tab != nullptr ? P.cam_tab_kind : 0);
    dim3 grid(bx, (unsigned)((n_views + vpt - 1) / vpt), 1), block(SL3D_BLOCK, 1, 1);
    // the timed kernels read the camera-side T1 from the per-calibration table whatever the batch is: with 8 views per lane it
    // costs nothing (1 B/px/view), with 1..4 it saves the iteration (+2..13 %), and a view's result does not depend on the
    // batch it was launched in
    P.use_cam_table = P.cam_tab != nullptr ? P.cam_tab_kind : 0;
#ifdef SL3D_MEASURE
    if (getenv("SL3D_CAMTAB") && atoi(getenv("SL3D_CAMTAB")) == 0) P.use_cam_table = 0;
    P.stagger = getenv("SL3D_STAGGER") ? atoi(getenv("SL3D_STAGGER")) : 0;
#endif
    if (compact == 1) {
        // persistent blocks that draw (tile, view group) items from the context's ticket counter: as many as the GPU holds at
        // once (more would only queue), each draws one ticket per item plus the one that tells it to stop
        const unsigned n_items = (unsigned)P.n_tiles * grid.y, slots = (unsigned)(P.n_cus > 0 ? P.n_cus : 256) * SL3D_OCC_COMPACT;
        grid = dim3(SL3D_PERSIST == 2 && slots < n_items ? slots : n_items, 1, 1);
        P.ticket_base = *tickets_drawn;
        drawn = SL3D_PERSIST == 2 ? n_items + grid.x : SL3D_PERSIST == 1 ? n_items : 0u;
        *tickets_drawn += drawn;
    }
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();  // an earlier sticky error of another library is not this launch's
    if (keep) {
        if (P.F == 3) launch_fused_n<true, false, 0, 0>(P.Nv, P.Nh, grid, block, st, P, d_cal, first_view, n_views, vpt);
        else launch_fused_n<true, true, 0, 0>(P.Nv, P.Nh, grid, block, st, P, d_cal, first_view, n_views, vpt);
    } else if (P.F != 3) {  // 4-step (and the all-invalid 5-step) fringes: the F test stays a run-time branch
        if (compact == 1) launch_fused_rig<true, 1>(rig, grid, block, st, P, d_cal, first_view, n_views, vpt);
        else if (compact == 2) launch_fused_rig<true, 2>(rig, grid, block, st, P, d_cal, first_view, n_views, vpt);
        else launch_fused_rig<true, 0>(rig, grid, block, st, P, d_cal, first_view, n_views, vpt);
    } else {
        if (compact == 1) launch_fused_rig<false, 1>(rig, grid, block, st, P, d_cal, first_view, n_views, vpt);
        else if (compact == 2) launch_fused_rig<false, 2>(rig, grid, block, st, P, d_cal, first_view, n_views, vpt);
        else launch_fused_rig<false, 0>(rig, grid, block, st, P, d_cal, first_view, n_views, vpt);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && tickets_drawn) *tickets_drawn -= drawn;  // a launch that did not happen drew no tickets
    return (int)e;
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

int launch_proj_table(const DevCal *d_cal, int PW, int PH, float2 *out, void *stream)
{
    hipLaunchKernelGGL(k_proj_table, dim3((PW + 255) / 256, PH), dim3(256), 0, (hipStream_t)stream, d_cal, PW, PH, out);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// N4: cvUndistort2 (2/project_pattern.cpp:220,232,...), OpenCV 2.4.0's algorithm (parity unpinned, see oracle/):
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
// sl3d_run_clouds, so it is written for latency.  The counts are taken through LDS in chunks of SL3D_SCAN_RUN x 1024: coalesced
// loads into a padded LDS array (rows of RUN + 1 words: the per-thread runs below are conflict-free), every thread scans its RUN
// consecutive entries, the 1024 run totals are scanned by wave shuffles + 16 wave totals, and the offsets leave coalesced again;
// a running carry links the chunks (any frame size).  The view's total goes straight into the mapped host word
// sl3d_get_cloud_counts reads.  (k_compact_scan -- 32 strided dwords per thread straight from global memory -- took 14.3 us for
// 16 x 32,400 counts; this scan with ONE block per view and runs of 32: 10.2 us.)
// Since the end of round 3 a view's segments are scanned by SL3D_SCAN_PARTS blocks instead of one (16 blocks on a 256-CU machine were
// a latency chain of ~10 us behind every sl3d_run_clouds): block (view, part) first SUMS the counts of the parts in front of it --
// the same coalesced reads every one of them does anyway, at most n dwords from the L2 -- and then scans its own part from that
// carry; no block waits for another.  The last part's block leaves the view's total.
#define SL3D_SCAN_RUN 4
#define SL3D_SCAN_PARTS 8
__global__ __launch_bounds__(1024) void k_seg_scan(const unsigned *__restrict__ counts, unsigned long long *__restrict__ offsets, int n,
                                                   unsigned long long *total)
{
    const int view = (int)blockIdx.y, part = (int)blockIdx.x;
    counts += (size_t)view * n;
    offsets += (size_t)view * n;
    constexpr int CHUNK = 1024 * SL3D_SCAN_RUN;
    // parts are whole chunks, so that every chunk of a part is scanned by the same code path
    const int part_len = (((n + SL3D_SCAN_PARTS - 1) / SL3D_SCAN_PARTS + CHUNK - 1) / CHUNK) * CHUNK;
    const int begin = min(part * part_len, n), end = min(begin + part_len, n);
    __shared__ unsigned s_val[1024 * (SL3D_SCAN_RUN + 1)];
    __shared__ unsigned s_wave[16];
    __shared__ unsigned long long s_carry;
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    {   // the carry into this part: the sum of everything in front of it
        unsigned long long acc = 0ull;
        for (int i = t; i < begin; i += 1024) acc += counts[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        __shared__ unsigned long long s_part[16];
        if (lane == 0) s_part[wave] = acc;
        __syncthreads();
        if (t == 0) {
            unsigned long long c = 0ull;
            for (int w = 0; w < 16; w++) c += s_part[w];
            s_carry = c;
        }
    }
    for (int base = begin; base < end; base += CHUNK) {
        const int m = min(end - base, CHUNK);
        __syncthreads();  // the previous chunk's LDS values have been written out; s_carry is up to date
        for (int i = t; i < CHUNK; i += 1024) s_val[i + i / SL3D_SCAN_RUN] = i < m ? counts[base + i] : 0u;
        __syncthreads();
        unsigned *mine = s_val + t * (SL3D_SCAN_RUN + 1);
        unsigned run = 0;
#pragma unroll
        for (int j = 0; j < SL3D_SCAN_RUN; j++) {  // in place: each entry becomes the exclusive prefix inside the thread's run
            const unsigned c = mine[j];
            mine[j] = run;
            run += c;
        }
        unsigned incl = run;  // inclusive scan of the run totals over the wave, then over the 16 waves
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        unsigned wbase = 0;
        for (int w = 0; w < wave; w++) wbase += s_wave[w];
        const unsigned long long carry = s_carry;
        mine[SL3D_SCAN_RUN] = wbase + (incl - run);  // the run's exclusive prefix inside the chunk, parked in the row's padding word
        __syncthreads();
        for (int i = t; i < m; i += 1024) {
            const int r = i / SL3D_SCAN_RUN;
            offsets[base + i] = carry + (unsigned long long)(s_val[r * (SL3D_SCAN_RUN + 1) + SL3D_SCAN_RUN] + s_val[i + r]);
        }
        if (t == 1023) s_carry = carry + (unsigned long long)(wbase + incl);
    }
    __syncthreads();
    if (t == 0 && part == SL3D_SCAN_PARTS - 1) total[view] = s_carry;
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
template <bool REG>
__global__ __launch_bounds__(256) void k_seg_close(const float *__restrict__ seg_xyz, const unsigned *__restrict__ counts,
                                                   const unsigned long long *__restrict__ offsets, int n_segs, size_t src_view_stride, float *dst,
                                                   size_t dst_view_stride, float r00, float r02, float r20, float r22, float tx, float ty, float tz)
{
    typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));
    const int seg = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u), v = blockIdx.y;
    if (seg >= n_segs) return;
    const unsigned cnt = counts[(size_t)v * n_segs + seg];
    // (a 3-float vector type is PADDED to 16 bytes: points are addressed through float pointers, 12 bytes apart)
    const float *src = seg_xyz + 3 * ((size_t)v * src_view_stride + (size_t)seg * SL3D_SEG_POINTS);
    float *out = dst + 3 * ((size_t)v * dst_view_stride + (size_t)offsets[(size_t)v * n_segs + seg]);
    for (unsigned i = (unsigned)lane; i < cnt; i += 64u) {
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
    hipLaunchKernelGGL(k_seg_close<false>, dim3((unsigned)((P.n_segs + 3) / 4), (unsigned)n_views), dim3(256), 0, (hipStream_t)stream,
                       P.clouds + 3 * (size_t)first_view * P.px_view_stride, P.seg_counts + (size_t)first_view * P.n_segs,
                       P.seg_offsets + (size_t)first_view * P.n_segs, P.n_segs, P.px_view_stride, dst, dst_view_stride_points, 0.f, 0.f, 0.f, 0.f, 0.f,
                       0.f, 0.f);
    return (int)hipGetLastError();
}

int launch_seg_register(const KParams &P, int view, float *out, const float R4[4], float tx, float ty, float tz, void *stream)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_seg_close<true>, dim3((unsigned)((P.n_segs + 3) / 4), 1u), dim3(256), 0, (hipStream_t)stream,
                       P.clouds + 3 * (size_t)view * P.px_view_stride, P.seg_counts + (size_t)view * P.n_segs, P.seg_offsets + (size_t)view * P.n_segs,
                       P.n_segs, P.px_view_stride, out, (size_t)0, R4[0], R4[1], R4[2], R4[3], tx, ty, tz);
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
// 9 intersection_points -> double[3].  dst: [W][H] elements of the window.
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
