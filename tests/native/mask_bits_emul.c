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

/* mask: full-frame bytes (fullW x fullH, row stride `stride`).  Window (col0,row0,W,H).  R rows per lane.
 * Outputs: norm plane [(H+4)][mpitch] (mpitch = pitch + 32, pitch = W rounded up to 16), band [H][pitch]; returns the quads
 * with a valid pixel. */
long emul_mask_prepare(const uint8_t *mask, size_t stride, int fullW, int fullH, int col0, int row0, int W, int H, int R, uint8_t *norm, uint8_t *band)
{
    const int pitch = (W + 15) & ~15, mpitch = pitch + 2 * LPAD, rows = H + 2 * HALO, dwpr = mpitch >> 2;
    /* the staging plane: zero outside the copied region, exactly what sl3d_set_masks builds */
    uint8_t *raw = (uint8_t *)calloc((size_t)mpitch * rows + 8, 1);
    const int gy0 = row0 - HALO < 0 ? 0 : row0 - HALO, gy1 = row0 + H + HALO > fullH ? fullH : row0 + H + HALO;
    const int gx0 = col0 - HALO < 0 ? 0 : col0 - HALO, gx1 = col0 + W + HALO > fullW ? fullW : col0 + W + HALO;
    const int bx0 = LPAD + gx0 - col0, bx1 = LPAD + gx1 - col0, r0 = gy0 - row0 + HALO, r1 = gy1 - row0 + HALO;
    for (int gy = gy0; gy < gy1; gy++) memcpy(raw + (size_t)(gy - row0 + HALO) * mpitch + bx0, mask + (size_t)gy * stride + gx0, (size_t)(gx1 - gx0));
    long quads = 0;
    const int strips = (rows + R - 1) / R;
    MbRow *row = (MbRow *)malloc(sizeof(MbRow) * (R + 3));
    unsigned *own = (unsigned *)malloc(sizeof(unsigned) * (R + 3)), *L = (unsigned *)malloc(sizeof(unsigned) * (R + 3)),
             *OK = (unsigned *)malloc(sizeof(unsigned) * (R + 3));
    for (int strip = 0; strip < strips; strip++)
        for (int x = 0; x < dwpr; x++) {
            const int pr0 = strip * R;
            const MbCols c = mb_cols(x, col0, LPAD, fullW, bx0, bx1);
            const unsigned own_bytes = mb_expand_nibble(c.REG >> 4) * 0xffu;
            const unsigned outw = (mb_range_bits(LPAD, LPAD + W, 4 * x - 4) >> 4) & 0xfu;
            for (int a = 0; a < R + 3; a++) {
                const int pr = pr0 + a - 2;
                unsigned dl = 0, dc = 0, dr = 0;
                if (pr >= r0 && pr < r1) {
                    const uint8_t *p = raw + (size_t)pr * mpitch + 4 * x;
                    if (c.REG & 0x00fu) memcpy(&dl, p - 4, 4);
                    if (c.REG & 0x0f0u) memcpy(&dc, p, 4);
                    if (c.REG & 0xf00u) memcpy(&dr, p + 4, 4);
                }
                const unsigned bc = mb_eq1_bytes(dc);
                const unsigned V = (mb_pack_nibble(mb_eq1_bytes(dl)) | (mb_pack_nibble(bc) << 4) | (mb_pack_nibble(mb_eq1_bytes(dr)) << 8)) & c.REG;
                row[a] = mb_row(V, c, row0 + pr - HALO, fullH);
                own[a] = bc & own_bytes;
            }
            for (int a = 1; a < R + 2; a++) {
                L[a] = mb_L(row[a], row[a + 1]);
                OK[a] = mb_OK(row[a], L[a], row[a - 1]);
            }
            const int xb = x - (LPAD >> 2);
            for (int a = 2; a < R + 2; a++) {
                const int pr = pr0 + a - 2, wr = pr - HALO;
                if (pr < rows) memcpy(norm + (size_t)pr * mpitch + (size_t)x * 4, &own[a], 4);
                if (xb >= 0 && xb < (pitch >> 2) && wr >= 0 && wr < H) {
                    const unsigned v = (mb_valid(row[a], L[a], OK[a - 1], OK[a]) >> 4) & outw;
                    const unsigned b = mb_expand_nibble(v);
                    memcpy(band + (size_t)wr * pitch + (size_t)xb * 4, &b, 4);
                    quads += v != 0;
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
                for (int i = 0; i < 12; i++)
                    if (base + i >= lo && base + i < hi) e |= 1u << i;
                if (mb_range_bits(lo, hi, base) != e) return 4;
            }
    return 0;
}
