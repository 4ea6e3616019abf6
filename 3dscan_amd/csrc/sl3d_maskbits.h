// sl3d_maskbits.h -- H0 / S3b / S3d as bit-plane arithmetic: the selection mask -> the valid map after stage 3's boundary
// removal (3/wrapped_phase.cpp:106-115, then :253-279 / :306-318), 16 pixels x R rows per lane, no per-pixel loads.
//
// Shared by k_mask_prepare (sl3d_kernels.hip) and by the CPU emulation the test suite compares with the literal scan of the reference's loop
// (tests/native/mask_bits_emul.c): plain C, no HIP types.
//
// The closed form (sl3d_device.h, MaskView) written per ROW of OWN + 8 neighbouring pixels held as bits of one register -- bit i of
// a row word is the pixel at plane byte (first own byte) - 4 + i, i.e. the lane's OWN pixels (16: one 16-byte load; the emulation also
// runs 4) are bits 4 .. 4 + OWN - 1 and the dword to their left / right is bits 0..3 / OWN + 4 .. OWN + 7 (everything a result
// depends on lies within 2 columns):
//   V      selected (byte == 1), 0 outside the staged region (which never leaves the frame)
//   nV3    = nV | nV<<1 | nV>>1          some pixel of the 3 columns around is unselected            (nV = ~V)
//   L(r)   = nV(r)>>1 | nV3(r+1)         E, SW, S or SE neighbour unselected: those are scanned LATER than the pixel
//   bu(r)  = ROWM(r) & nV(r)             frame-border pixel, unselected (never scanned, hence never `visited`)
//   B(r)   = bu3(r-1) | bu(r)<<1         NW, N, NE or W neighbour is such a pixel
//   OK(r)  = V | INT(r) & (L | B)        a neighbour that does not clear the pixels scanned after it
//   valid(y) = V & ( ~INT(y) | ~L(y) & OK3(y-1) & OK(y)<<1 )
// with INT(r) = the interior columns if row r is an interior row (else 0) and ROWM(r) = the frame's border columns on an interior
// row, every in-frame column on the frame's first / last row, 0 outside the frame.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define SL3D_MB_FN __host__ __device__ __forceinline__
#else
#define SL3D_MB_FN static inline
#endif

// bits i of an nbits-bit row word (nbits <= 24) whose coordinate base + i lies in [lo, hi)
SL3D_MB_FN unsigned mb_range_bits(int lo, int hi, int base, int nbits)
{
    int a = lo - base, b = hi - base;
    a = a < 0 ? 0 : (a > nbits ? nbits : a);
    b = b < 0 ? 0 : (b > nbits ? nbits : b);
    return b > a ? ((1u << b) - (1u << a)) : 0u;
}

// 0x01 in every byte of d that equals 1 (exact: no borrow runs between the bytes)
SL3D_MB_FN unsigned mb_eq1_bytes(unsigned d)
{
    const unsigned t = d ^ 0x01010101u;
    return (~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u) >> 7;
}
// 0/1 bytes -> 4 bits (byte k -> bit k): the four partial products land on distinct bits, so nothing carries
SL3D_MB_FN unsigned mb_pack_nibble(unsigned bytes01) { return (bytes01 * 0x01020408u) >> 24; }
// 4 bits -> 0/1 bytes
SL3D_MB_FN unsigned mb_expand_nibble(unsigned n) { return ((n & 0xfu) * 0x00204081u) & 0x01010101u; }

// per-lane column constants
typedef struct MbCols {
    unsigned REG;   // bits inside the staged region's byte columns [bx0, bx1)
    unsigned INF;   // in-frame columns
    unsigned INTC;  // interior columns: 1 <= gx <= fullW - 2
} MbCols;

SL3D_MB_FN MbCols mb_cols(int own_byte0 /* plane byte of the lane's first own pixel */, int own, int col0, int lpad, int fullW, int bx0, int bx1)
{
    MbCols c;
    const int b0 = own_byte0 - 4;        // plane byte of bit 0
    const int gx0 = col0 - lpad + b0;    // its frame column
    c.REG = mb_range_bits(bx0, bx1, b0, own + 8);
    c.INF = mb_range_bits(0, fullW, gx0, own + 8);
    c.INTC = mb_range_bits(1, fullW - 1, gx0, own + 8);
    return c;
}

// per-row derived words
typedef struct MbRow {
    unsigned V, nV, nV3, bu, bu3, INT;
} MbRow;

// V: the 12 selected bits of frame row gy (already limited to the staged region)
SL3D_MB_FN MbRow mb_row(unsigned V, const MbCols c, int gy, int fullH)
{
    MbRow r;
    const int inframe = gy >= 0 && gy < fullH, interior = gy >= 1 && gy <= fullH - 2;
    const unsigned rowm = !inframe ? 0u : (interior ? (c.INF & ~c.INTC) : c.INF);
    r.V = V;
    r.nV = ~V;
    r.nV3 = r.nV | (r.nV << 1) | (r.nV >> 1);
    r.bu = rowm & r.nV;
    r.bu3 = r.bu | (r.bu << 1) | (r.bu >> 1);
    r.INT = interior ? c.INTC : 0u;
    return r;
}

SL3D_MB_FN unsigned mb_L(const MbRow r, const MbRow below) { return (r.nV >> 1) | below.nV3; }
SL3D_MB_FN unsigned mb_OK(const MbRow r, unsigned L, const MbRow above) { return r.V | (r.INT & (L | above.bu3 | (r.bu << 1))); }
// valid bits of row y (bits 4 .. 4 + OWN - 1 are the lane's own pixels) from OK of the row above and of the row itself
SL3D_MB_FN unsigned mb_valid(const MbRow r, unsigned L, unsigned OK_above, unsigned OK_row)
{
    const unsigned ok3 = OK_above & (OK_above << 1) & (OK_above >> 1);
    return r.V & (~r.INT | (~L & ok3 & (OK_row << 1)));
}

// ---- one quad per lane, one row per lane: the fused kernel's MASKIN launches (sl3d_fused.h) --------------------------------------------
// A result bit depends on nothing beyond 2 columns either side (the recurrences above shift by at most one column twice), so a lane
// that owns ONE quad needs 8 selection bytes per row -- plane bytes own-2 .. own+5 -- which arrive as ONE 8-byte load at own + delta:
// delta = -2, or 0 / -4 where that load would begin before / end behind the readable part [lo, hi) of the row (the frame's first /
// last quad of a caller's own device-resident mask; hi - lo >= 8).  The bytes that are then missing lie outside the frame.
SL3D_MB_FN int mb_quad_delta(int own_byte0, int lo, int hi) { return own_byte0 - 2 < lo ? 0 : (own_byte0 + 6 > hi ? -4 : -2); }
// the two dwords of that load -> the 12-bit row word (bit i = plane byte own - 4 + i; bits the load does not cover: 0)
SL3D_MB_FN unsigned mb_quad_word(unsigned w0, unsigned w1, int delta)
{
    const unsigned b = mb_pack_nibble(mb_eq1_bytes(w0)) | (mb_pack_nibble(mb_eq1_bytes(w1)) << 4);  // bit i = plane byte own + delta + i
    return (b << (4 + delta)) & 0xfffu;
}
// the lane's own dword / the one to its left / to its right out of the same load, as far as it covers them (the rest lies outside
// every region: see mb_quad_delta) -- what the normalised 0/1 plane stores
SL3D_MB_FN unsigned mb_quad_own(unsigned w0, unsigned w1, int delta) { return delta == 0 ? w0 : delta == -4 ? w1 : (w0 >> 16) | (w1 << 16); }
SL3D_MB_FN unsigned mb_quad_left(unsigned w0, unsigned w1, int delta) { return delta == 0 ? 0u : delta == -4 ? w0 : w0 << 16; }
SL3D_MB_FN unsigned mb_quad_right(unsigned w0, unsigned w1, int delta) { return delta == 0 ? w1 : delta == -4 ? 0u : w1 >> 16; }
// does frame row gy - 2 matter to the quad's results?  Only through unselected FRAME-BORDER pixels (bu): all of the frame's first row, else
// the frame's first / last column
SL3D_MB_FN int mb_quad_top_needed(const MbCols c, int gy, int fullH)
{
    const int ty = gy - 2, inframe = ty >= 0 && ty < fullH, interior = ty >= 1 && ty <= fullH - 2;
    const unsigned rowm = !inframe ? 0u : (interior ? (c.INF & ~c.INTC) : c.INF);
    return (rowm & 0x3fcu) != 0u;
}
// valid bits (bit k = pixel k of the quad) of frame row gy from the row words of gy-2 (0 if not needed), gy-1, gy, gy+1 (each already
// limited to the staged region)
SL3D_MB_FN unsigned mb_quad_valid(unsigned Vtop, unsigned Vm1, unsigned V0, unsigned Vp1, const MbCols c, int gy, int fullH)
{
    const MbRow r0 = mb_row(Vtop, c, gy - 2, fullH), r1 = mb_row(Vm1, c, gy - 1, fullH), r2 = mb_row(V0, c, gy, fullH), r3 = mb_row(Vp1, c, gy + 1, fullH);
    const unsigned L1 = mb_L(r1, r2), L2 = mb_L(r2, r3);
    const unsigned OK1 = mb_OK(r1, L1, r0), OK2 = mb_OK(r2, L2, r1);
    return (mb_valid(r2, L2, OK1, OK2) >> 4) & 0xfu;
}

// The same for a quad in the PLAIN interior: the 8 columns own-2 .. own+5 are interior frame columns inside the region and rows gy-1 ..
// gy+1 are interior frame rows, gy-2 is not the frame's first row (mb_quad_plain).  Then no frame-border pixel has a say (bu = 0), INT
// is all ones and nothing needs masking: ~25 instead of ~110 integer instructions, and no column constants (mb_cols) at all.  Row words
// straight from mb_quad_word with delta = -2 (bits 2..9).
SL3D_MB_FN int mb_quad_plain(int gx0 /* frame column of the quad's first pixel */, int gy, int fullW, int fullH, int reg_x0, int reg_x1 /* region columns [x0, x1) */)
{
    return gx0 - 2 >= 1 && gx0 + 5 <= fullW - 2 && gx0 - 2 >= reg_x0 && gx0 + 5 < reg_x1 && gy >= 3 && gy + 1 <= fullH - 2;
}
SL3D_MB_FN unsigned mb_quad_valid_plain(unsigned Vm1, unsigned V0, unsigned Vp1)
{
    const unsigned n0 = ~Vm1, n1 = ~V0, n2 = ~Vp1;
    const unsigned L1 = (n0 >> 1) | n1 | (n1 << 1) | (n1 >> 1), L2 = (n1 >> 1) | n2 | (n2 << 1) | (n2 >> 1);
    const unsigned OK1 = Vm1 | L1, OK2 = V0 | L2;
    return ((V0 & ~L2 & OK1 & (OK1 << 1) & (OK1 >> 1) & (OK2 << 1)) >> 4) & 0xfu;
}
