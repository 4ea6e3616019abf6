/* CPU proof of the exact-arithmetic shortcuts the HIP kernels use (IEEE double ops and fma behave
 * identically on the CPU and on gfx950):
 *  1. div_exact(a, c, 1/c): q0 = a*y, r = fma(-q0,c,a), q = fma(r,y,q0) equals a/c
 *       - c = 7        for every a = 44*code, code < 2^20            (unwrap: code*2.0*Pi)
 *       - c = 2*22/7   for every float a in [5e-4, 6e4] and a = 0     (correspondence: phi/(2.0*Pi))
 *  2. atan2_lattice (octant + pi/8 reduction on the integers, quotient n * RN(1/d), Horner polynomial of
 *     3dscan_amd/csrc/sl3d_atan_coeffs.h, pi/4 and pi as plain doubles) rounded to float equals (float)atan2 of
 *     libm on all 511 x 1021 lattice points; also prints how close the true value (long double) and the
 *     computed one come to a float rounding boundary (the accuracy budget of the polynomial).
 *     The device evaluates the very same IEEE operations (the LDS table holds RN(1/d)); its rcp + Newton variant
 *     may differ from RN(1/d) by an ulp, ~300x below the margin; the GPU self-check in sl3d_create covers both.
 *  -DATAN_ONLY skips part 1 (used to check the other polynomial degrees of the header).
 * Prints "OK" and exits 0 when everything holds. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sl3d_atan_coeffs.h"

static double div_exact(double a, double c, double y) { double q0 = a * y; double r = fma(-q0, c, a); return fma(r, y, q0); }

static double atan2_lattice(int t1, int t2)
{
    static const double Q[SL3D_ATAN_DEG + 1] = SL3D_ATAN_Q;
    int ay = abs(t1), ax = abs(t2), lo = ay < ax ? ay : ax, hi = ay < ax ? ax : ay, swap = ay > ax;
    int red = 169 * lo > 70 * hi;
    int num = red ? hi - lo : lo, den = red ? hi + lo : hi;
    double r = (double)num * (1.0 / (double)(den == 0 ? 1 : den));
    double z = r * r, p = Q[0];
    for (int i = 1; i <= SL3D_ATAN_DEG; i++) p = fma(p, z, Q[i]);
    double a = fma(r, z * p, r);
    int k1 = red ? 1 : (swap ? 2 : 0), k2 = t2 < 0 ? 4 - k1 : k1;
    int nega = (swap != red) != (t2 < 0);
    /* the kernel rounds the non-negative value to float and then applies the sign of t1: same float */
    double phi2 = fma((double)k2, SL3D_PIO4, nega ? -a : a);
    return t1 < 0 ? -phi2 : phi2;
}

/* relative distance of v to the nearest float rounding boundary */
static long double boundary_distance(long double v)
{
    float f = (float)v, up = nextafterf(f, INFINITY), dn = nextafterf(f, -INFINITY);
    long double m1 = ((long double)f + (long double)up) / 2, m2 = ((long double)f + (long double)dn) / 2;
    long double d1 = fabsl(v - m1), d2 = fabsl(v - m2);
    return (d1 < d2 ? d1 : d2) / fabsl(v);
}

int main(void)
{
    long bad = 0, bad7 = 0;
#ifndef ATAN_ONLY
    long n = 0;
    const double c = 2.0 * 22.0 / 7.0, y = 1.0 / c;
    uint32_t lo, hi;
    float flo = 0.0005f, fhi = 60000.0f;
    memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
    for (uint32_t b = lo; b <= hi; b++) {
        float f; memcpy(&f, &b, 4);
        if (div_exact((double)f, c, y) != (double)f / c) bad++;
        n++;
    }
    if (div_exact(0.0, c, y) != 0.0) bad++;
    printf("div by 2*22/7: %ld floats, %ld mismatches\n", n, bad);
    for (long code = 0; code < (1 << 20); code++) {
        double a = ((double)code * 2.0) * 22.0;
        if ((double)(code * 44) != a) bad7++;
        if (div_exact(a, 7.0, 1.0 / 7.0) != a / 7.0) bad7++;
    }
    printf("div by 7: %ld mismatches over 2^20 codes\n", bad7);
#endif
    long bada = 0;
    long double margin_true = 1, margin_comp = 1;
    for (int t1 = -255; t1 <= 255; t1++)
        for (int t2 = -510; t2 <= 510; t2++) {
            float ref = (float)atan2((double)(float)t1, (double)(float)t2), got = (float)atan2_lattice(t1, t2);
            uint32_t a, b; memcpy(&a, &ref, 4); memcpy(&b, &got, 4);
            if (a != b) bada++;
            if (t1 != 0) {  /* t1 == 0: exactly 0 or pi */
                long double dt = boundary_distance(atan2l((long double)t1, (long double)t2)), dc = boundary_distance((long double)atan2_lattice(t1, t2));
                if (dt < margin_true) margin_true = dt;
                if (dc < margin_comp) margin_comp = dc;
            }
        }
    printf("atan2 lattice, degree %d: %ld mismatches over %d points; closest approach to a float rounding boundary: true %.2Le, computed %.2Le (relative)\n",
           SL3D_ATAN_DEG, bada, 511 * 1021, margin_true, margin_comp);
    if (margin_comp < 1e-14L) bada++;  /* an ulp of difference in the reciprocal must not matter */
    if (bad || bad7 || bada) return 1;
    printf("OK\n");
    return 0;
}
