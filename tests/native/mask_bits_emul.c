/* CPU emulation of k_mask_prepare (3dscan_amd/csrc/sl3d_kernels.hip): the SAME bit-plane arithmetic (sl3d_maskbits.h, shared
 * with the kernel) driven by the same lane / strip indexing, so that the closed form, its 12-bit row words and every frame /
 * window / halo edge case are checked against the oracle's literal scan of 3/wrapped_phase.cpp:253-279 on the CPU
 * (tests/test_mask_bits.py).  Test infrastructure only: the product never runs this. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "sl3d_maskbits.h"

#define HALO 2
#define LPAD 16

/* mask: full-frame bytes (fullW x fullH, row stride `stride`).  Window (col0,row0,W,H).  R rows per lane, OWN (4 or 16) pixels per lane
 * and row -- the kernel runs 16.  Outputs: norm plane [(H+4)][mpitch] (mpitch = pitch + 32, pitch = W rounded up to 16), band [H][pitch];
 * returns the quads with a valid pixel. */
long emul_mask_prepare(const uint8_t *mask, size_t stride, int fullW, int fullH, int col0, int row0, int W, int H, int R, int OWN, uint8_t *norm,
                       uint8_t *band)
{
    const int pitch = (W + 15) & ~15, mpitch = pitch + 2 * LPAD, rows = H + 2 * HALO, cpr = mpitch / OWN, nd = OWN / 4;
    /* the staging plane: zero outside the copied region, exactly what sl3d_set_masks builds */
    uint8_t *raw = (uint8_t *)calloc((size_t)mpitch * rows + 32, 1);
    const int gy0 = row0 - HALO < 0 ? 0 : row0 - HALO, gy1 = row0 + H + HALO > fullH ? fullH : row0 + H + HALO;
    const int gx0 = col0 - HALO < 0 ? 0 : col0 - HALO, gx1 = col0 + W + HALO > fullW ? fullW : col0 + W + HALO;
    const int bx0 = LPAD + gx0 - col0, bx1 = LPAD + gx1 - col0, r0 = gy0 - row0 + HALO, r1 = gy1 - row0 + HALO;
    for (int gy = gy0; gy < gy1; gy++) memcpy(raw + (size_t)(gy - row0 + HALO) * mpitch + bx0, mask + (size_t)gy * stride + gx0, (size_t)(gx1 - gx0));
    long quads = 0;
    const int strips = (rows + R - 1) / R;
    MbRow *row = (MbRow *)malloc(sizeof(MbRow) * (R + 3));
    unsigned *own = (unsigned *)malloc(sizeof(unsigned) * (R + 3) * 4), *L = (unsigned *)malloc(sizeof(unsigned) * (R + 3)),
             *OK = (unsigned *)malloc(sizeof(unsigned) * (R + 3));
    const unsigned own_mask = OWN == 16 ? 0xffffu : 0xfu;
    for (int strip = 0; strip < strips; strip++)
        for (int x = 0; x < cpr; x++) {
            const int pr0 = strip * R;
            const MbCols c = mb_cols(OWN * x, OWN, col0, LPAD, fullW, bx0, bx1);
            const unsigned reg_own = (c.REG >> 4) & own_mask;
            const unsigned outw = (mb_range_bits(LPAD, LPAD + W, OWN * x - 4, OWN + 8) >> 4) & own_mask;
            for (int a = 0; a < R + 3; a++) {
                const int pr = pr0 + a - 2;
                unsigned dl = 0, dr = 0, d[4] = {0, 0, 0, 0};
                if (pr >= r0 && pr < r1) {
                    const uint8_t *p = raw + (size_t)pr * mpitch + OWN * x;
                    for (int k = 0; k < nd; k++)
                        if (reg_own & (0xfu << (4 * k))) memcpy(&d[k], p + 4 * k, 4);
                    if (c.REG & 0xfu) memcpy(&dl, p - 4, 4);
                    if (c.REG & (0xfu << (OWN + 4))) memcpy(&dr, p + OWN, 4);
                }
                unsigned V = mb_pack_nibble(mb_eq1_bytes(dl)) | (mb_pack_nibble(mb_eq1_bytes(dr)) << (OWN + 4));
                for (int k = 0; k < nd; k++) {
                    const unsigned bk = mb_eq1_bytes(d[k]);
                    V |= mb_pack_nibble(bk) << (4 + 4 * k);
                    own[4 * a + k] = bk & (mb_expand_nibble(reg_own >> (4 * k)) * 0xffu);
                }
                row[a] = mb_row(V & c.REG, c, row0 + pr - HALO, fullH);
            }
            for (int a = 1; a < R + 2; a++) {
                L[a] = mb_L(row[a], row[a + 1]);
                OK[a] = mb_OK(row[a], L[a], row[a - 1]);
            }
            const int xb = x - LPAD / OWN;
            for (int a = 2; a < R + 2; a++) {
                const int pr = pr0 + a - 2, wr = pr - HALO;
                if (pr < rows) memcpy(norm + (size_t)pr * mpitch + (size_t)x * OWN, &own[4 * a], (size_t)OWN);
                if (xb >= 0 && xb < pitch / OWN && wr >= 0 && wr < H) {
                    const unsigned v = (mb_valid(row[a], L[a], OK[a - 1], OK[a]) >> 4) & outw;
                    for (int k = 0; k < nd; k++) {
                        const unsigned b = mb_expand_nibble(v >> (4 * k));
                        memcpy(band + (size_t)wr * pitch + (size_t)xb * OWN + 4 * k, &b, 4);
                        quads += ((v >> (4 * k)) & 0xfu) != 0;
                    }
                }
            }
        }
    free(row); free(own); free(L); free(OK); free(raw);
    return quads;
}

/* the byte helpers, exhaustively: every dword pattern of bytes in {0,1,2,0x80,0xff,0x81,0x7f,0x00} x 4 and all 16 nibbles */
int emul_check_byte_helpers(void)
{
    static const unsigned vals[8] = {0, 1, 2, 0x80, 0xff, 0x81, 0x7f, 0x01};
    for (int i = 0; i < 4096; i++) {
        unsigned d = 0, expect_bytes = 0, expect_nib = 0;
        for (int k = 0; k < 4; k++) {
            const unsigned b = vals[(i >> (3 * k)) & 7];
            d |= b << (8 * k);
            if (b == 1) { expect_bytes |= 1u << (8 * k); expect_nib |= 1u << k; }
        }
        if (mb_eq1_bytes(d) != expect_bytes) return 1;
        if (mb_pack_nibble(expect_bytes) != expect_nib) return 2;
        if (mb_expand_nibble(expect_nib) != expect_bytes) return 3;
    }
    for (int base = -30; base < 30; base++)
        for (int lo = -20; lo < 20; lo++)
            for (int hi = -20; hi < 24; hi++) {
                unsigned e = 0;
                for (int nbits = 12; nbits <= 24; nbits += 12) {
                    e = 0;
                    for (int i = 0; i < nbits; i++)
                        if (base + i >= lo && base + i < hi) e |= 1u << i;
                    if (mb_range_bits(lo, hi, base, nbits) != e) return 4;
                }
            }
    return 0;
}

/* CPU emulation of a MASKIN launch's mask evaluation (3dscan_amd/csrc/sl3d_fused.h: maskin_lane / maskin_request / maskin_finish): one
 * lane per quad and window row, the 8-byte row loads with their delta, the rows y-2 / y+2 asked for only where they matter, the halo
 * duties of the window's first / last row and quad -- through the shared header's mb_quad_* functions.  direct = 0: the source is the
 * staging plane (as emul_mask_prepare builds it); direct = 1: the caller's own full-frame mask is read in place (needs col0, fullW and
 * stride multiples of 4, fullW >= 8), and every byte read must lie inside the mask (returns -1 - <count of bytes read outside>).
 * Outputs as emul_mask_prepare; returns the quads with a valid pixel. */
static long g_oob;
long g_plain_quads; /* quads the short form was checked on, over all calls */
static void load8(const uint8_t *base, long off, const uint8_t *lo, const uint8_t *hi, unsigned *w0, unsigned *w1)
{
    const uint8_t *p = base + off;
    if (p < lo || p + 8 > hi) { g_oob += 1; *w0 = *w1 = 0; return; }
    memcpy(w0, p, 4);
    memcpy(w1, p + 4, 4);
}

long emul_maskin(const uint8_t *mask, size_t stride, int fullW, int fullH, int col0, int row0, int W, int H, int direct, uint8_t *norm, uint8_t *band)
{
    const int pitch = (W + 15) & ~15, mpitch = pitch + 2 * LPAD, rows = H + 2 * HALO;
    uint8_t *raw = (uint8_t *)calloc((size_t)mpitch * rows + 32, 1);
    const int gy0 = row0 - HALO < 0 ? 0 : row0 - HALO, gy1 = row0 + H + HALO > fullH ? fullH : row0 + H + HALO;
    const int gx0 = col0 - HALO < 0 ? 0 : col0 - HALO, gx1 = col0 + W + HALO > fullW ? fullW : col0 + W + HALO;
    const int bx0 = LPAD + gx0 - col0, bx1 = LPAD + gx1 - col0, r0 = gy0 - row0 + HALO, r1 = gy1 - row0 + HALO;
    for (int gy = gy0; gy < gy1; gy++) memcpy(raw + (size_t)(gy - row0 + HALO) * mpitch + bx0, mask + (size_t)gy * stride + gx0, (size_t)(gx1 - gx0));
    /* the kernel's view of the source: origin = address of plane row 0, byte 0 */
    const uint8_t *origin = direct ? mask + ((long)row0 - HALO) * (long)stride + col0 - LPAD : raw;
    const long sstride = direct ? (long)stride : mpitch;
    const int lo = direct ? LPAD - col0 : 0, hi = direct ? LPAD - col0 + fullW : mpitch;
    const uint8_t *mem_lo = direct ? mask : raw, *mem_hi = direct ? mask + (size_t)(fullH - 1) * stride + fullW : raw + (size_t)mpitch * rows;
    g_oob = 0;
    long quads = 0, plain_mismatch = 0, n_plain = 0;
    /* planes the kernel never writes stay what the context's creation left them: zero */
    memset(norm, 0, (size_t)mpitch * rows);
    for (int row = 0; row < H; row++)
        for (int cq = 0; cq < pitch / 4; cq++) {
            const int own = LPAD + cq * 4;
            const MbCols c = mb_cols(own, 4, col0, LPAD, fullW, bx0, bx1);
            const int delta = mb_quad_delta(own, lo, hi), gy = row0 + row;
            unsigned w[5][2];   /* plane rows row .. row + 4 (window rows y-2 .. y+2) */
            int have[5] = {0, 1, 1, 1, 0};
            const int first = row == 0, last = row == H - 1;
            if (first || mb_quad_top_needed(c, gy, fullH)) have[0] = 1;
            if (last) have[4] = 1;
            for (int a = 0; a < 5; a++) {
                w[a][0] = w[a][1] = 0;
                const int pr = row + a;
                if (have[a] && pr >= r0 && pr < r1 && (c.REG & 0x0f0u)) load8(origin, (long)pr * sstride + own + delta, mem_lo, mem_hi, &w[a][0], &w[a][1]);
            }
            unsigned V[4];
            for (int a = 0; a < 4; a++) V[a] = (a == 0 && !have[0]) ? 0u : (mb_quad_word(w[a][0], w[a][1], delta) & c.REG);
            unsigned outw = mb_range_bits(0, W, cq * 4, 4);
            const unsigned v = mb_quad_valid(V[0], V[1], V[2], V[3], c, gy, fullH) & outw;
            /* the short form of the plain interior (what whole waves of such lanes evaluate) must say the same */
            if (mb_quad_plain(col0 + cq * 4, gy, fullW, fullH, gx0, gx1) && own - 2 >= lo && own + 6 <= hi && !first && !last && cq != 0 && cq != pitch / 4 - 1) {
                n_plain++;
                if (delta != -2 || have[0] || (c.REG & 0x3fcu) != 0x3fcu ||
                    mb_quad_valid_plain(mb_quad_word(w[1][0], w[1][1], -2), mb_quad_word(w[2][0], w[2][1], -2), mb_quad_word(w[3][0], w[3][1], -2)) != v)
                    plain_mismatch++;
            }
            const unsigned vb = mb_expand_nibble(v);
            memcpy(band + (size_t)row * pitch + (size_t)cq * 4, &vb, 4);
            quads += v != 0;
            for (int a = 0; a < 5; a++) {
                if (!(a == 2 || (first && a < 2) || (last && a > 2))) continue;
                uint8_t *q = norm + (size_t)(row + a) * mpitch + own;
                unsigned o = mb_eq1_bytes(mb_quad_own(w[a][0], w[a][1], delta)) & (mb_expand_nibble(c.REG >> 4) * 0xffu);
                memcpy(q, &o, 4);
                if (cq == 0) { o = mb_eq1_bytes(mb_quad_left(w[a][0], w[a][1], delta)) & (mb_expand_nibble(c.REG) * 0xffu); memcpy(q - 4, &o, 4); }
                if (cq == pitch / 4 - 1) { o = mb_eq1_bytes(mb_quad_right(w[a][0], w[a][1], delta)) & (mb_expand_nibble(c.REG >> 8) * 0xffu); memcpy(q + 4, &o, 4); }
            }
        }
    free(raw);
    g_plain_quads += n_plain;
    if (plain_mismatch) return -1000000 - plain_mismatch;
    return g_oob ? -1 - g_oob : quads;
}
