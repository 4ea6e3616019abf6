// valubench.hip -- issue cost of the VALU instructions the fused kernel is made of (cycles per wave64
// instruction per SIMD), measured with 8 independent chains per lane and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096, CH = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int n)
{
    double a[CH]; unsigned u[CH]; float f[CH];
    for (int i = 0; i < CH; i++) { a[i] = 1.0 + threadIdx.x * 1e-9 + i; u[i] = threadIdx.x + i; f[i] = 1.f + i; }
    const double c = out[0] + 1.0000001, d = 1e-9;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (OP == 0) a[i] = fma(a[i], c, d);                       // v_fma_f64
            if (OP == 1) a[i] = a[i] * c;                              // v_mul_f64
            if (OP == 2) a[i] = a[i] + c;                              // v_add_f64
            if (OP == 3) u[i] = u[i] * 3u + (unsigned)it;              // v_mad_u32_u24 / mul+add
            if (OP == 4) u[i] = (u[i] ^ (u[i] >> 3)) + 1u;             // v_xor + v_lshr + v_add (3 ops)
            if (OP == 5) f[i] = fmaf(f[i], 1.0001f, 0.5f);             // v_fma_f32
            if (OP == 6) a[i] = (double)(float)a[i] + d;               // cvt_f32_f64 + cvt_f64_f32 + add
            if (OP == 7) a[i] = __builtin_amdgcn_rcp(a[i]) + c;        // v_rcp_f64 + add
            if (OP == 8) a[i] = (u[i] & 1u) ? a[i] : c;                // 2 x v_cndmask_b32
            if (OP == 9) a[i] = __builtin_rint(a[i] * c);              // mul + rndne
        }
    }
    double s = 0; unsigned t = 0; float g = 0;
    for (int i = 0; i < CH; i++) { s += a[i]; t += u[i]; g += f[i]; }
    out[1 + blockIdx.x * 256 + threadIdx.x] = s + t + g;
}

// accuracy of v_rcp_f64 alone, with one Newton step (recip1 of the kernels) and with two (recip): max relative error
// against the IEEE division over 2^26 doubles spread over [2^-20, 2^20) (splitmix-style hash per element)
__global__ void k_rcp_accuracy(double *maxerr)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    const double m = 1.0 + (double)(z >> 12) * 0x1p-52;                 // [1, 2)
    const double d = ldexp(m, (int)(z & 63u) % 40 - 20) * ((z & 64u) ? -1.0 : 1.0);
    const double ref = 1.0 / d;
    const double r0 = __builtin_amdgcn_rcp(d);
    const double r1 = fma(r0, fma(-d, r0, 1.0), r0);
    const double r2 = fma(r1, fma(-d, r1, 1.0), r1);
    const double e[3] = {fabs(r0 - ref) / fabs(ref), fabs(r1 - ref) / fabs(ref), fabs(r2 - ref) / fabs(ref)};
    for (int j = 0; j < 3; j++) {
        double v = e[j];
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
        if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&maxerr[j], (unsigned long long)__double_as_longlong(v));  // non-negative doubles order as integers
    }
}

int main()
{
    {
        double *me; CHK(hipMalloc(&me, 24)); CHK(hipMemset(me, 0, 24));
        hipLaunchKernelGGL(k_rcp_accuracy, dim3(1 << 18), dim3(256), 0, 0, me);
        double h[3]; CHK(hipMemcpy(h, me, 24, hipMemcpyDeviceToHost));
        printf("max relative error over 2^26 doubles: v_rcp_f64 %.3e (2^%.1f), + 1 Newton step %.3e, + 2 steps %.3e\n", h[0], log2(h[0]), h[1], h[2]);
        CHK(hipFree(me));
    }
    double *out; CHK(hipMalloc(&out, (1 + 1024 * 256) * 8)); CHK(hipMemset(out, 0, 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    int clk = 0; CHK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    const char *names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "int mad (mul+add)", "xor+lshr+add (3 int ops)", "v_fma_f32",
                           "cvt f64->f32->f64 + add (3 ops)", "v_rcp_f64 + add (2 ops)", "2 x v_cndmask_b32", "mul_f64 + rndne_f64 (2 ops)"};
    const int nops[] = {1, 1, 1, 2, 3, 1, 3, 2, 2, 2};
    for (int op = 0; op < 10; op++) {
        float best = 1e9;
        for (int r = 0; r < 5; r++) {
            CHK(hipEventRecord(e0));
            const dim3 g(1024), b(256);  // 4 blocks per CU = 4 waves per SIMD
            switch (op) {
            case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, ITER); break; case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, ITER); break;
            case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, ITER); break; case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, ITER); break;
            case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, out, ITER); break; case 5: hipLaunchKernelGGL(k<5>, g, b, 0, 0, out, ITER); break;
            case 6: hipLaunchKernelGGL(k<6>, g, b, 0, 0, out, ITER); break; case 7: hipLaunchKernelGGL(k<7>, g, b, 0, 0, out, ITER); break;
            case 8: hipLaunchKernelGGL(k<8>, g, b, 0, 0, out, ITER); break; case 9: hipLaunchKernelGGL(k<9>, g, b, 0, 0, out, ITER); break;
            }
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // wave-instructions per SIMD = 4 waves * ITER * CH * nops ; time -> cycles at the reported clock
        const double insts = 4.0 * ITER * CH * nops[op];
        const double ns_per = best * 1e6 / insts;
        printf("%-36s %7.3f ms  %6.2f ns per wave-instruction per SIMD  (= %5.2f cycles at %.2f GHz nominal)\n", names[op], best, ns_per,
               ns_per * clk / 1e6, clk / 1e6);
    }
    return 0;
}
