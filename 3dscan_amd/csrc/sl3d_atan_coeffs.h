/* sl3d_atan_coeffs.h -- polynomial of the lattice atan2 (shared by the HIP kernels and by the exhaustive CPU
 * proof tests/native/exact_arith_check.c, so the proof covers exactly the constants the kernels are built with).
 *
 * atan(r) = r + r*z*Q(z), z = r^2, 0 <= r <= 99/239 (after the octant and tan(pi/8) reductions).
 * Q is the Chebyshev-node interpolant of (atan(r)/r - 1)/z on [0, (99/239)^2] (tools/fit_atan.py, long double):
 *   degree  6:  |atan error| <= 2.8e-12 relative        degree  8: 3.6e-15        degree 10: 7e-18
 * What is REQUIRED is only that (float)phi equals (float)atan2() of libm on the 521,731 lattice points
 * |t1| <= 255, |t2| <= 510.  The true atan2 of a lattice point never comes closer than 6.7e-14 (relative) to a
 * float rounding boundary (tests/native/exact_arith_check.c prints the margin), so degree 8 is safe analytically
 * with a factor 18 to spare, and every degree listed here is PROVEN by exhaustion (CPU test + device self-check
 * at the first sl3d_create); degree 5 fails on 28 points.  Coefficients are listed highest degree first. */
#ifndef SL3D_ATAN_COEFFS_H
#define SL3D_ATAN_COEFFS_H

#ifndef SL3D_ATAN_DEG
#define SL3D_ATAN_DEG 8
#endif

#if SL3D_ATAN_DEG == 6
#define SL3D_ATAN_Q { -0x1.4b362ba10a5b0p-5, 0x1.2442bd1eef1bep-4, -0x1.71d3a76a99b92p-4, 0x1.c6f685282f675p-4, \
                      -0x1.2491c081b1849p-3, 0x1.9999982575ce2p-3, -0x1.5555555502103p-2 }
#elif SL3D_ATAN_DEG == 7
#define SL3D_ATAN_Q { 0x1.0df924d8fe6e7p-5, -0x1.edc0883d027a3p-5, 0x1.377df8419ffbdp-4, -0x1.7415b2886a0bfp-4, \
                      0x1.c719635ce2e93p-4, -0x1.249240da0b078p-3, 0x1.999999885d13fp-3, -0x1.5555555552612p-2 }
#elif SL3D_ATAN_DEG == 8
#define SL3D_ATAN_Q { -0x1.be2abca282e03p-6, 0x1.a76cf3c1543d2p-5, -0x1.0c53100a4eae4p-4, 0x1.3a9d928344ae3p-4, \
                      -0x1.74563d8e51db4p-4, 0x1.c71c3825d8da8p-4, -0x1.249248aa6eaf8p-3, 0x1.99999998d1640p-3, \
                      -0x1.55555555553a3p-2 }
#elif SL3D_ATAN_DEG == 10
#define SL3D_ATAN_Q { -0x1.3a2b7a07caea9p-6, 0x1.41603647c7a7cp-5, -0x1.a098bb6ba4941p-5, 0x1.dfe61e80903d2p-5, \
                      -0x1.10fa75382537fp-4, 0x1.3b1262d95579ep-4, -0x1.745d0b26b83e7p-4, 0x1.c71c7185314cbp-4, \
                      -0x1.24924924360cbp-3, 0x1.999999999934ap-3, -0x1.5555555555555p-2 }
#else
#error "SL3D_ATAN_DEG must be 6, 7, 8 or 10"
#endif

/* pi/4 and pi to the nearest double (k*pi/4, k <= 2, is then exact too) */
#define SL3D_PIO4 0x1.921fb54442d18p-1
#define SL3D_PI 0x1.921fb54442d18p+1

#endif
