/* CPU proof of the exact-arithmetic shortcuts the HIP kernels use (IEEE double ops and fma behave
 * identically on the CPU and on gfx950):
 *  1. div_exact(a, c, 1/c): q0 = a*y, r = fma(-q0,c,a), q = fma(r,y,q0) equals a/c
 *       - c = 7        for every a = 44*code, code < 2^20            (unwrap: code*2.0*Pi)
 *       - c = 2*22/7   for every float a in [5e-4, 6e4] and a = 0     (correspondence: phi/(2.0*Pi))
 *  2. atan2_lattice (octant + pi/8 reduction, degree-10 Horner polynomial, hi/lo constants) rounded to
 *     float equals (float)atan2 of libm on all 511 x 1021 lattice points (the device version replaces the
 *     single division by rcp + Newton + Markstein; the GPU self-check in sl3d_create covers that).
 * Prints "OK" and exits 0 when everything holds. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double div_exact(double a, double c, double y) { double q0 = a * y; double r = fma(-q0, c, a); return fma(r, y, q0); }

static double atan2_lattice(int t1, int t2)
{
    static const double Q[11] = {-0x1.5555555555555p-2, 0x1.999999999934ap-3, -0x1.24924924360cbp-3, 0x1.c71c7185314cbp-4,
                                 -0x1.745d0b26b83e7p-4, 0x1.3b1262d95579ep-4, -0x1.10fa75382537fp-4, 0x1.dfe61e80903d2p-5,
                                 -0x1.a098bb6ba4941p-5, 0x1.41603647c7a7cp-5, -0x1.3a2b7a07caea9p-6};
    const double PIO4_HI = 0x1.921fb54442d18p-1, PIO4_LO = 0x1.1a62633145c07p-55, PI_HI = 0x1.921fb54442d18p+1, PI_LO = 0x1.1a62633145c07p-53;
    int ay = abs(t1), ax = abs(t2), lo = ay < ax ? ay : ax, hi = ay < ax ? ax : ay, swap = ay > ax;
    int red = 169 * lo > 70 * hi;
    int num = red ? hi - lo : lo, den = red ? hi + lo : hi;
    double r = (double)num / (double)(den == 0 ? 1 : den);
    double z = r * r, p = Q[10];
    for (int i = 9; i >= 0; i--) p = fma(p, z, Q[i]);
    double a = fma(r, z * p, r);
    double kq = (double)((red ? 1 : 0) + ((swap && !red) ? 2 : 0));
    double sa = (swap != red) ? -a : a;
    double phi = fma(kq, PIO4_HI, 0.0) + (sa + kq * PIO4_LO);
    phi = t2 < 0 ? PI_HI - (phi - PI_LO) : phi;
    return t1 < 0 ? -phi : phi;
}

int main(void)
{
    long bad = 0, n = 0;
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
    long bad7 = 0;
    for (long code = 0; code < (1 << 20); code++) {
        double a = ((double)code * 2.0) * 22.0;
        if ((double)(code * 44) != a) bad7++;
        if (div_exact(a, 7.0, 1.0 / 7.0) != a / 7.0) bad7++;
    }
    printf("div by 7: %ld mismatches over 2^20 codes\n", bad7);
    long bada = 0;
    for (int t1 = -255; t1 <= 255; t1++)
        for (int t2 = -510; t2 <= 510; t2++) {
            float ref = (float)atan2((double)(float)t1, (double)(float)t2), got = (float)atan2_lattice(t1, t2);
            uint32_t a, b; memcpy(&a, &ref, 4); memcpy(&b, &got, 4);
            if (a != b) bada++;
        }
    printf("atan2 lattice: %ld mismatches over %d points\n", bada, 511 * 1021);
    if (bad || bad7 || bada) return 1;
    printf("OK\n");
    return 0;
}
