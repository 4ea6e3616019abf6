// membench.hip -- access-pattern microbenchmark behind the fused kernel's data-path design.
// Streams P byte planes of n pixels (like the 47 planes of a view) and writes 13 B/pixel, with
//   mode 0: one dword  (4 px)  per lane per plane  -> 256 B per wave-instruction
//   mode 1: one dwordx4 (16 px) per lane per plane  -> 1 KiB per wave-instruction, registers
//   mode 2: dwordx4 global->LDS DMA (global_load_lds), tile of 1024 px per 256-thread block, then ds_read_b32
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench ; run: tools/membench [planes] [Mpx]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int P>
__global__ __launch_bounds__(256) void k_dword(const uint8_t *in, size_t plane, float4 *out, unsigned *outv, size_t nquads)
{
    size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquads) return;
    unsigned v[P];
#pragma unroll
    for (int p = 0; p < P; p++) v[p] = *(const unsigned *)(in + p * plane + q * 4);
    unsigned a = 0, b = 0, c = 0;
#pragma unroll
    for (int p = 0; p < P; p++) { a ^= v[p]; b += v[p]; c |= v[p] >> (p & 7); }
    float fa = __uint_as_float((a & 0x007fffffu) | 0x3f800000u), fb = __uint_as_float((b & 0x007fffffu) | 0x3f800000u),
          fc = __uint_as_float((c & 0x007fffffu) | 0x3f800000u);
    out[q * 3 + 0] = make_float4(fa, fb, fc, fa);
    out[q * 3 + 1] = make_float4(fb, fc, fa, fb);
    out[q * 3 + 2] = make_float4(fc, fa, fb, fc);
    outv[q] = a;
}

template <int P>
__global__ __launch_bounds__(256) void k_dwordx4(const uint8_t *in, size_t plane, float4 *out, uint4 *outv, size_t n16)
{
    size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;  // 16 px per lane
    if (q >= n16) return;
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
#pragma unroll
    for (int p = 0; p < P; p++) {
        uint4 v = *(const uint4 *)(in + p * plane + q * 16);
        a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w;
        b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
    unsigned w[4] = {a.x, a.y, a.z, a.w}, u[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float fa = __uint_as_float((w[k] & 0x007fffffu) | 0x3f800000u), fb = __uint_as_float((u[k] & 0x007fffffu) | 0x3f800000u);
        out[q * 12 + 3 * k + 0] = make_float4(fa, fb, fa, fb);
        out[q * 12 + 3 * k + 1] = make_float4(fb, fa, fb, fa);
        out[q * 12 + 3 * k + 2] = make_float4(fa, fa, fb, fb);
    }
    outv[q] = a;
}

template <int P>
__global__ __launch_bounds__(256) void k_lds(const uint8_t *in, size_t plane, float4 *out, unsigned *outv, size_t nquads)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char tile[];  // [P][1024]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t tile0 = (size_t)blockIdx.x * 1024;  // first pixel byte of the tile
    for (int p = wave; p < P; p += 4) {
        const uint8_t *src = in + p * plane + tile0 + lane * 16;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src,
                                         (void __attribute__((address_space(3))) *)(tile + p * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned a = 0, b = 0, c = 0;
#pragma unroll
    for (int p = 0; p < P; p++) { unsigned v = *(const unsigned *)(tile + p * 1024 + threadIdx.x * 4); a ^= v; b += v; c |= v >> (p & 7); }
    if (q >= nquads) return;
    float fa = __uint_as_float((a & 0x007fffffu) | 0x3f800000u), fb = __uint_as_float((b & 0x007fffffu) | 0x3f800000u),
          fc = __uint_as_float((c & 0x007fffffu) | 0x3f800000u);
    out[q * 3 + 0] = make_float4(fa, fb, fc, fa);
    out[q * 3 + 1] = make_float4(fb, fc, fa, fb);
    out[q * 3 + 2] = make_float4(fc, fa, fb, fc);
    outv[q] = a;
}

// incremental model of the fused kernel's data path: feature flags add one element at a time
//   1: loop over 4 "views" per lane (planes of view v at +v*P*plane)   2: 9 mask dwords first, loads depend on them
//   4: 8 dependent table gathers (2 MB table)   8: LDS round trip for the 48 B of output   16: ~600 dependent fp64 fma per quad
//   32: tile-interleaved frame layout [view][tile of 64 quads][plane][256 B]: the P loads of a wave hit ONE contiguous P*256-B region
//   64: 8 views per lane instead of 4 (with flag 1)
template <int P, int OCC>
__global__ __launch_bounds__(256, OCC) void k_steps(const uint8_t *in, size_t plane, size_t view_stride, int nviews, const uint8_t *mask,
                                                    const float *tab, float4 *out, unsigned *outv, size_t nquads, int flags)
{
    __shared__ __attribute__((aligned(16))) float sx[256 * 12];
    size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquads) return;
    const int vpl = (flags & 64) ? 8 : 4;
    const int v0 = (flags & 1) ? blockIdx.y * vpl : blockIdx.y, v1 = (flags & 1) ? v0 + vpl : v0 + 1;
    for (int view = v0; view < v1 && view < nviews; view++) {
        const uint8_t *base = in + (size_t)view * view_stride + q * 4;
        size_t pstride = plane;
        if (flags & 32) {
            base = in + (size_t)view * view_stride + (q >> 6) * (size_t)(P * 256) + (q & 63) * 4;
            pstride = 256;
        }
        unsigned m = 0xf;
        if (flags & 2) {
            const unsigned *mp = (const unsigned *)(mask + (size_t)view * nquads * 4 + q * 4);
            unsigned acc = 0xffffffffu;
#pragma unroll
            for (int i = -4; i <= 4; i++) acc &= mp[(q + i < nquads && (long)q + i >= 0) ? i : 0] | 0x01010101u;
            m = acc & 0xf;
        }
        float4 o0 = make_float4(0, 0, 0, 0), o1 = o0, o2 = o0;
        unsigned a = 0, b = 0, c = 0;
        if (m) {
            unsigned v[P];
#pragma unroll
            for (int p = 0; p < P; p++) v[p] = *(const unsigned *)(base + p * pstride);
            float w = 0;
            if (flags & 4) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    unsigned idx = ((v[k % 6] >> (8 * (k & 3))) & 255) * 1021 + (((v[(k + 1) % 6] >> (8 * (k & 3))) & 255) * 3 + 100);
                    w += tab[idx];
                }
            }
#pragma unroll
            for (int p = 0; p < P; p++) { a ^= v[p]; b += v[p]; c |= v[p] >> (p & 7); }
            double acc = (double)w + (double)a;
            if (flags & 16) {
#pragma unroll 8
                for (int i = 0; i < 600; i++) acc = fma(acc, 1.0000001, 1e-9);
            }
            float fa = __uint_as_float((a & 0x007fffffu) | 0x3f800000u) + w, fb = __uint_as_float((b & 0x007fffffu) | 0x3f800000u),
                  fc = __uint_as_float((c & 0x007fffffu) | 0x3f800000u) + (float)acc;
            o0 = make_float4(fa, fb, fc, fa); o1 = make_float4(fb, fc, fa, fb); o2 = make_float4(fc, fa, fb, fc);
        }
        if (flags & 8) {
            float4 *my = (float4 *)(sx + threadIdx.x * 12);
            my[0] = o0; my[1] = o1; my[2] = o2;
            o0 = my[0]; o1 = my[1]; o2 = my[2];
        }
        float4 *op = out + ((size_t)view * nquads + q) * 3;
        op[0] = o0; op[1] = o1; op[2] = o2;
        outv[(size_t)view * nquads + q] = a;
    }
}

// Round 3: what a PERSISTENT, VIEW-MAJOR walk of the same bytes would give the data path: as many blocks as the machine holds
// (slots), every block walks the views in the same order (outer loop) and, inside a view, its share of the tiles (inner loop:
// tile b, b + slots, ...), so that at any moment the whole machine streams the 47 planes of about ONE view instead of 8 views at
// once; per step the lane also reads 32 B of a per-pixel table (the camera table, which nothing amortises in this order; it is
// re-read per view, from L2 / the Infinity Cache).  Compare with flags 0 (one view per block) and 65 (8 views per lane).
template <int P, int OCC>
__global__ __launch_bounds__(256, OCC) void k_viewmajor(const uint8_t *in, size_t plane, size_t view_stride, int nviews, const double *camtab, float4 *out,
                                                        unsigned *outv, size_t nquads, int use_tab)
{
    const size_t ntiles = (nquads + 255) / 256;
    for (int view = 0; view < nviews; view++)
        for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const size_t q = tile * 256 + threadIdx.x;
            if (q >= nquads) continue;
            const uint8_t *base = in + (size_t)view * view_stride + q * 4;
            unsigned v[P];
#pragma unroll
            for (int p = 0; p < P; p++) v[p] = *(const unsigned *)(base + p * plane);
            double t = 0;
            if (use_tab) {
                const double2 *tp = (const double2 *)(camtab + q * 4);
                const double2 a = tp[0], b = tp[1];
                t = a.x + a.y + b.x + b.y;
            }
            unsigned a = 0, b = 0, c = 0;
#pragma unroll
            for (int p = 0; p < P; p++) { a ^= v[p]; b += v[p]; c |= v[p] >> (p & 7); }
            float fa = __uint_as_float((a & 0x007fffffu) | 0x3f800000u) + (float)t, fb = __uint_as_float((b & 0x007fffffu) | 0x3f800000u),
                  fc = __uint_as_float((c & 0x007fffffu) | 0x3f800000u);
            float4 *op = out + ((size_t)view * nquads + q) * 3;
            op[0] = make_float4(fa, fb, fc, fa); op[1] = make_float4(fb, fc, fa, fb); op[2] = make_float4(fc, fa, fb, fc);
            outv[(size_t)view * nquads + q] = a;
        }
}

// Round 3: persistent WAVES that draw their work from a ticket counter, items in address order (view-major, then tile, 256
// pixels = one wave's quads per item): the set of tiles in flight stays the narrow moving window the dispatcher gives short
// blocks (flags 0) although the waves loop -- which is what would let a looping kernel keep its in-wave pipeline AND the DRAM
// locality of one-view-per-block.  The next ticket is drawn while the current item is processed; `prefetch` also requests the
// next item's planes before the current item's stores (two register sets).
template <int P, int OCC, bool PREFETCH>
__global__ __launch_bounds__(256, OCC) void k_ticket(const uint8_t *in, size_t plane, size_t view_stride, int nviews, const double *camtab, float4 *out,
                                                     unsigned *outv, size_t nquads, int use_tab, unsigned *counter)
{
    const unsigned lane = threadIdx.x & 63u;
    const size_t nw = (nquads + 63) / 64, total = nw * (size_t)nviews;  // wave items per view, items in all
    auto draw = [&]() -> unsigned {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(counter, 1u);
        return __builtin_amdgcn_readfirstlane(t);
    };
    auto issue = [&](unsigned item, unsigned (&v)[P], double2 (&ct)[2]) {
        const size_t view = item / nw, q = (item % nw) * 64 + lane;
        const size_t qq = q < nquads ? q : nquads - 1;
        const uint8_t *base = in + view * view_stride + qq * 4;
#pragma unroll
        for (int p = 0; p < P; p++) v[p] = *(const unsigned *)(base + p * plane);
        if (use_tab) {
            const double2 *tp = (const double2 *)(camtab + qq * 4);
            ct[0] = tp[0];
            ct[1] = tp[1];
        }
    };
    auto finish = [&](unsigned item, const unsigned (&v)[P], const double2 (&ct)[2]) {
        const size_t view = item / nw, q = (item % nw) * 64 + lane;
        if (q >= nquads) return;
        const double t = use_tab ? ct[0].x + ct[0].y + ct[1].x + ct[1].y : 0.0;
        unsigned a = 0, b = 0, c = 0;
#pragma unroll
        for (int p = 0; p < P; p++) { a ^= v[p]; b += v[p]; c |= v[p] >> (p & 7); }
        float fa = __uint_as_float((a & 0x007fffffu) | 0x3f800000u) + (float)t, fb = __uint_as_float((b & 0x007fffffu) | 0x3f800000u),
              fc = __uint_as_float((c & 0x007fffffu) | 0x3f800000u);
        float4 *op = out + (view * nquads + q) * 3;
        op[0] = make_float4(fa, fb, fc, fa); op[1] = make_float4(fb, fc, fa, fb); op[2] = make_float4(fc, fa, fb, fc);
        outv[view * nquads + q] = a;
    };
    unsigned item = draw();
    if (item >= total) return;
    if constexpr (!PREFETCH) {
        for (;;) {
            unsigned va[P];
            double2 ca[2] = {};
            const unsigned next = draw();
            issue(item, va, ca);
            finish(item, va, ca);
            if (next >= total) return;
            item = next;
        }
    }
    unsigned va[P], vb[P];
    double2 ca[2] = {}, cb[2] = {};
    issue(item, va, ca);
    for (;;) {
        const unsigned next = draw();
        if (PREFETCH) {
            if (next < total) issue(next, vb, cb);
            finish(item, va, ca);
            if (next >= total) break;
            item = next;
            const unsigned next2 = draw();
            if (next2 < total) issue(next2, va, ca);
            finish(item, vb, cb);
            if (next2 >= total) break;
            item = next2;
            // (the ticket drawn at the top of the loop for this item is next2's successor: handled by the next iteration)
        } else {
            finish(item, va, ca);
            if (next >= total) break;
            item = next;
            issue(item, va, ca);
        }
    }
}

// Round 3: the same static view-major walk with the NEXT item's loads issued before the current item's stores (two register
// sets): a looping wave then never has to wait for the acknowledgement of its own stores before it can consume loads (vmcnt
// counts loads and stores in one in-order counter on this ISA), which is what separates a looping kernel from short blocks.
template <int P, int OCC>
__global__ __launch_bounds__(256, OCC) void k_viewmajor_pipe(const uint8_t *in, size_t plane, size_t view_stride, int nviews, float4 *out, unsigned *outv,
                                                             size_t nquads)
{
    const size_t ntiles = (nquads + 255) / 256, total = ntiles * (size_t)nviews;
    auto issue = [&](size_t item, unsigned (&v)[P]) {
        const size_t view = item / ntiles, q = (item % ntiles) * 256 + threadIdx.x, qq = q < nquads ? q : nquads - 1;
        const uint8_t *base = in + view * view_stride + qq * 4;
#pragma unroll
        for (int p = 0; p < P; p++) v[p] = *(const unsigned *)(base + p * plane);
    };
    auto finish = [&](size_t item, const unsigned (&v)[P]) {
        const size_t view = item / ntiles, q = (item % ntiles) * 256 + threadIdx.x;
        if (q >= nquads) return;
        unsigned a = 0, b = 0, c = 0;
#pragma unroll
        for (int p = 0; p < P; p++) { a ^= v[p]; b += v[p]; c |= v[p] >> (p & 7); }
        float fa = __uint_as_float((a & 0x007fffffu) | 0x3f800000u), fb = __uint_as_float((b & 0x007fffffu) | 0x3f800000u),
              fc = __uint_as_float((c & 0x007fffffu) | 0x3f800000u);
        float4 *op = out + (view * nquads + q) * 3;
        op[0] = make_float4(fa, fb, fc, fa); op[1] = make_float4(fb, fc, fa, fb); op[2] = make_float4(fc, fa, fb, fc);
        outv[view * nquads + q] = a;
    };
    size_t item = blockIdx.x;
    if (item >= total) return;
    unsigned va[P], vb[P];
    issue(item, va);
    for (;;) {
        size_t next = item + gridDim.x;
        if (next < total) issue(next, vb);
        finish(item, va);
        if (next >= total) break;
        item = next;
        next = item + gridDim.x;
        if (next < total) issue(next, va);
        finish(item, vb);
        if (next >= total) break;
        item = next;
    }
}

// one-view launches (round 3): the data path of ONE 1080p view per launch, back to back (the working set stays in the Infinity
// Cache), at 8 waves/SIMD (every wave of the launch resident at once: one round) against 4 (LDS-limited: two rounds, like the
// fused kernel at 128 VGPRs), and with the planes requested in two dependent halves (what a 64-VGPR kernel would have to do)
// store form of k_one: 0 = three 16-B pieces per lane (48-B stride), 1 = the same with the non-temporal hint, 2 = 1-KiB runs of
// whole lines per instruction (lane after lane), 3 = the same with the hint (what the fused kernel ships since the end of round 3)
template <int P, bool SPLIT, int ST = 0>
__global__ __launch_bounds__(256, 8) void k_one(const uint8_t *in, size_t plane, float4 *out, unsigned *outv, size_t nquads, int lds_words)
{
    extern __shared__ unsigned pad[];
    size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquads) return;
    unsigned a = 0, b = 0, c = 0;
    if (lds_words) pad[threadIdx.x] = (unsigned)q;  // keeps the allocation alive
    if (SPLIT) {
        constexpr int H1 = P / 2;
        unsigned v[H1];
#pragma unroll
        for (int p = 0; p < H1; p++) v[p] = *(const unsigned *)(in + p * plane + q * 4);
#pragma unroll
        for (int p = 0; p < H1; p++) { a ^= v[p]; b += v[p]; c |= v[p] >> (p & 7); }
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
        unsigned w[P - H1];
        const uint8_t *in2 = in + (a == 0x12345u ? 4 : 0);  // the second half's addresses depend on the first half's data
#pragma unroll
        for (int p = H1; p < P; p++) w[p - H1] = *(const unsigned *)(in2 + p * plane + q * 4);
#pragma unroll
        for (int p = H1; p < P; p++) { a ^= w[p - H1]; b += w[p - H1]; c |= w[p - H1] >> (p & 7); }
    } else {
        unsigned v[P];
#pragma unroll
        for (int p = 0; p < P; p++) v[p] = *(const unsigned *)(in + p * plane + q * 4);
#pragma unroll
        for (int p = 0; p < P; p++) { a ^= v[p]; b += v[p]; c |= v[p] >> (p & 7); }
    }
    if (lds_words) a ^= pad[threadIdx.x ^ 1] & 1u;
    float fa = __uint_as_float((a & 0x007fffffu) | 0x3f800000u), fb = __uint_as_float((b & 0x007fffffu) | 0x3f800000u),
          fc = __uint_as_float((c & 0x007fffffu) | 0x3f800000u);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 v0 = {fa, fb, fc, fa}, v1 = {fb, fc, fa, fb}, v2 = {fc, fa, fb, fc};
    f32x4 *o = (f32x4 *)out;
    if (ST >= 2) {
        const size_t w0 = (q & ~(size_t)63) * 3, l = q & 63;   // (the data would come across lanes from LDS: free in the fused kernel)
        if (ST == 3) {
            __builtin_nontemporal_store(v0, o + w0 + l);
            __builtin_nontemporal_store(v1, o + w0 + 64 + l);
            __builtin_nontemporal_store(v2, o + w0 + 128 + l);
            __builtin_nontemporal_store(a, outv + q);
        } else {
            o[w0 + l] = v0; o[w0 + 64 + l] = v1; o[w0 + 128 + l] = v2;
            outv[q] = a;
        }
    } else if (ST == 1) {
        __builtin_nontemporal_store(v0, o + q * 3);
        __builtin_nontemporal_store(v1, o + q * 3 + 1);
        __builtin_nontemporal_store(v2, o + q * 3 + 2);
        __builtin_nontemporal_store(a, outv + q);
    } else {
        o[q * 3 + 0] = v0; o[q * 3 + 1] = v1; o[q * 3 + 2] = v2;
        outv[q] = a;
    }
}

template <int ST>
static void run_store_form(const uint8_t *fr, size_t vpx, int nviews, float4 *o, unsigned *ov, hipEvent_t e0, hipEvent_t e1, const char *name)
{
    constexpr int P = 47;
    const size_t nq = vpx * nviews / 4;   // the views as ONE run of pixels: plane stride = all of them
    const int launches = nviews == 1 ? 300 : 40;
    float best = 1e9;
    for (int pass = 0; pass < 3; pass++) {
        CHK(hipEventRecord(e0));
        for (int i = 0; i < launches; i++) hipLaunchKernelGGL((k_one<P, false, ST>), dim3((nq + 255) / 256), dim3(256), 0, 0, fr, vpx * nviews, o, ov, nq, 0);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (pass && ms < best) best = ms;
    }
    CHK(hipGetLastError());
    const double us = best * 1e3 / launches, bytes = (double)vpx * nviews * (P + 13);
    printf("store form %-44s %2d view(s) per launch: %8.2f us  %6.0f GB/s  %5.1f Gpx/s\n", name, nviews, us, bytes / us / 1e3, vpx * nviews / us / 1e3);
}

// scope / cache-policy bits of gfx950 global loads and stores on the fused kernel's data path (one dword per lane and plane in,
// whole 1-KiB runs out): LD 0 plain, 1 nt (the builtin), 2 "sc1", 3 "sc0 sc1", 4 "sc0 sc1 nt"; ST 0 nt (the builtin), 1 "sc1",
// 2 "sc0 sc1", 3 "sc0 sc1 nt", 4 plain.  The asm forms wait with an explicit s_waitcnt.
#define MB_LOAD_ASM(BITS)  asm volatile("global_load_dword %0, %1, off " BITS : "=v"(v[p]) : "v"(a) : "memory")
#define MB_STORE_ASM(BITS, ptr, val) asm volatile("global_store_dwordx4 %0, %1, off " BITS :: "v"(ptr), "v"(val) : "memory")
template <int P, int LD, int ST>
__global__ __launch_bounds__(256, 8) void k_scope(const uint8_t *in, size_t plane, float4 *out, unsigned *outv, size_t nquads)
{
    size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquads) return;
    unsigned v[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        const unsigned *a = (const unsigned *)(in + p * plane + q * 4);
        if (LD == 0) v[p] = *a;
        else if (LD == 1) v[p] = __builtin_nontemporal_load(a);
        else if (LD == 2) MB_LOAD_ASM("sc1");
        else if (LD == 3) MB_LOAD_ASM("sc0 sc1");
        else MB_LOAD_ASM("sc0 sc1 nt");
    }
    if (LD >= 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < P; p++) asm volatile("" : "+v"(v[p]));
    }
    unsigned a = 0, b = 0, c = 0;
#pragma unroll
    for (int p = 0; p < P; p++) { a ^= v[p]; b += v[p]; c |= v[p] >> (p & 7); }
    float fa = __uint_as_float((a & 0x007fffffu) | 0x3f800000u), fb = __uint_as_float((b & 0x007fffffu) | 0x3f800000u),
          fc = __uint_as_float((c & 0x007fffffu) | 0x3f800000u);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 v0 = {fa, fb, fc, fa}, v1 = {fb, fc, fa, fb}, v2 = {fc, fa, fb, fc};
    f32x4 *o = (f32x4 *)out;
    const size_t w0 = (q & ~(size_t)63) * 3, l = q & 63;
    f32x4 *p0 = o + w0 + l, *p1 = o + w0 + 64 + l, *p2 = o + w0 + 128 + l;
    if (ST == 0) {
        __builtin_nontemporal_store(v0, p0); __builtin_nontemporal_store(v1, p1); __builtin_nontemporal_store(v2, p2);
        __builtin_nontemporal_store(a, outv + q);
    } else if (ST == 4) {
        *p0 = v0; *p1 = v1; *p2 = v2; outv[q] = a;
    } else {
        if (ST == 1) { MB_STORE_ASM("sc1", p0, v0); MB_STORE_ASM("sc1", p1, v1); MB_STORE_ASM("sc1", p2, v2); }
        else if (ST == 2) { MB_STORE_ASM("sc0 sc1", p0, v0); MB_STORE_ASM("sc0 sc1", p1, v1); MB_STORE_ASM("sc0 sc1", p2, v2); }
        else { MB_STORE_ASM("sc0 sc1 nt", p0, v0); MB_STORE_ASM("sc0 sc1 nt", p1, v1); MB_STORE_ASM("sc0 sc1 nt", p2, v2); }
        __builtin_nontemporal_store(a, outv + q);
    }
}


// round 4: the same data path with the planes of a view stored PLANAR ([plane][row][pitch]: a wave's 46 loads go to 46 regions
// 2 MB apart) or ROW-INTERLEAVED ([row][plane][pitch]: the 46 loads of a wave lie 1920 B apart inside one 88-KB run, consecutive
// tiles read consecutive memory) -- address = plane * plane_stride + row * row_pitch + 4 * quad column
template <int P>
__global__ __launch_bounds__(256, 8) void k_layout(const uint8_t *in, size_t view_stride, unsigned plane_stride, unsigned row_pitch, unsigned qpr, float4 *out,
                                                   unsigned *outv, size_t nquads_view)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquads_view) return;
    const unsigned row = (unsigned)(q / qpr), cq = (unsigned)(q - (size_t)row * qpr);
    const uint8_t *base = in + (size_t)blockIdx.y * view_stride + (size_t)row * row_pitch + cq * 4u;
    unsigned v[P];
#pragma unroll
    for (int p = 0; p < P; p++) v[p] = __builtin_nontemporal_load((const unsigned *)(base + (size_t)p * plane_stride));
    unsigned a = 0, b = 0, c = 0;
#pragma unroll
    for (int p = 0; p < P; p++) { a ^= v[p]; b += v[p]; c |= v[p] >> (p & 7); }
    float fa = __uint_as_float((a & 0x007fffffu) | 0x3f800000u), fb = __uint_as_float((b & 0x007fffffu) | 0x3f800000u),
          fc = __uint_as_float((c & 0x007fffffu) | 0x3f800000u);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 v0 = {fa, fb, fc, fa}, v1 = {fb, fc, fa, fb}, v2 = {fc, fa, fb, fc};
    const size_t qg = (size_t)blockIdx.y * nquads_view + q;
    f32x4 *o = (f32x4 *)out;
    const size_t w0 = (qg & ~(size_t)63) * 3, l = qg & 63;
    __builtin_nontemporal_store(v0, o + w0 + l); __builtin_nontemporal_store(v1, o + w0 + 64 + l); __builtin_nontemporal_store(v2, o + w0 + 128 + l);
    __builtin_nontemporal_store(a, outv + qg);
}

template <int LD, int ST>
static void run_scope(const uint8_t *fr, size_t vpx, int nviews, float4 *o, unsigned *ov, hipEvent_t e0, hipEvent_t e1)
{
    constexpr int P = 47;
    const size_t nq = vpx * nviews / 4;
    const int launches = 40;
    float best = 1e9;
    for (int pass = 0; pass < 3; pass++) {
        CHK(hipEventRecord(e0));
        for (int i = 0; i < launches; i++) hipLaunchKernelGGL((k_scope<P, LD, ST>), dim3((nq + 255) / 256), dim3(256), 0, 0, fr, vpx * nviews, o, ov, nq);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (pass && ms < best) best = ms;
    }
    CHK(hipGetLastError());
    static const char *ld[] = {"plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt"}, *st[] = {"nt", "sc1", "sc0 sc1", "sc0 sc1 nt", "plain"};
    const double us = best * 1e3 / launches, bytes = (double)vpx * nviews * (P + 13);
    printf("loads %-11s stores %-11s %2d views per launch: %8.2f us  %6.0f GB/s  %5.1f Gpx/s\n", ld[LD], st[ST], nviews, us, bytes / us / 1e3, vpx * nviews / us / 1e3);
}

int main(int argc, char **argv)
{
    constexpr int P = 47;
    if (argc > 1 && !strcmp(argv[1], "scope")) {
        const size_t vpx = 1920 * 1080;
        uint8_t *fr; float4 *o; unsigned *ov;
        CHK(hipMalloc(&fr, vpx * 47 * 16 + 64)); CHK(hipMalloc(&o, vpx * 12 * 16)); CHK(hipMalloc(&ov, vpx * 16));
        CHK(hipMemset(fr, 0x5a, vpx * 47 * 16 + 64));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++) {
            run_scope<0, 0>(fr, vpx, 16, o, ov, e0, e1);
            run_scope<1, 0>(fr, vpx, 16, o, ov, e0, e1);
            run_scope<2, 0>(fr, vpx, 16, o, ov, e0, e1);
            run_scope<3, 0>(fr, vpx, 16, o, ov, e0, e1);
            run_scope<4, 0>(fr, vpx, 16, o, ov, e0, e1);
            run_scope<1, 4>(fr, vpx, 16, o, ov, e0, e1);
            run_scope<1, 1>(fr, vpx, 16, o, ov, e0, e1);
            run_scope<1, 2>(fr, vpx, 16, o, ov, e0, e1);
            run_scope<1, 3>(fr, vpx, 16, o, ov, e0, e1);
            run_scope<4, 3>(fr, vpx, 16, o, ov, e0, e1);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "stores")) {  // the data path alone with each store form, 1 and 16 views of 1080p per launch
        const size_t vpx = 1920 * 1080;
        uint8_t *fr; float4 *o; unsigned *ov;
        CHK(hipMalloc(&fr, vpx * 47 * 16 + 64)); CHK(hipMalloc(&o, vpx * 12 * 16)); CHK(hipMalloc(&ov, vpx * 16));
        CHK(hipMemset(fr, 0x5a, vpx * 47 * 16 + 64));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++)
            for (int nviews : {1, 16}) {
                run_store_form<0>(fr, vpx, nviews, o, ov, e0, e1, "16-B pieces");
                run_store_form<1>(fr, vpx, nviews, o, ov, e0, e1, "16-B pieces, non-temporal");
                run_store_form<2>(fr, vpx, nviews, o, ov, e0, e1, "whole 1-KiB runs");
                run_store_form<3>(fr, vpx, nviews, o, ov, e0, e1, "whole 1-KiB runs, non-temporal");
            }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "layout")) {
        // planar vs row-interleaved frames: one view per launch from HBM (8 views round robin) and 16 views per launch (steady)
        const unsigned Wd = 1920, Hh = 1080, qpr = Wd / 4;
        const size_t vpx = (size_t)Wd * Hh, nq = vpx / 4;
        const int NV = 16;
        uint8_t *fr; float4 *o; unsigned *ov;
        CHK(hipMalloc(&fr, vpx * 47 * NV + 64)); CHK(hipMalloc(&o, vpx * 12 * NV)); CHK(hipMalloc(&ov, vpx * NV));
        CHK(hipMemset(fr, 0x5a, vpx * 47 * NV + 64));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; rep++)
            for (int inter = 0; inter < 2; inter++)
                for (int batch : {1, 16}) {
                    const unsigned plane_stride = inter ? Wd : (unsigned)vpx, row_pitch = inter ? 47u * Wd : Wd;
                    const int launches = batch == 1 ? 800 : 100;
                    float ms = 0;
                    for (int pass = 0; pass < 2; pass++) {
                        CHK(hipEventRecord(e0));
                        for (int i = 0; i < launches; i++) {
                            const int v = batch == 1 ? i % 8 : 0;
                            hipLaunchKernelGGL((k_layout<47>), dim3((nq + 255) / 256, batch), dim3(256), 0, 0, fr + (size_t)v * vpx * 47, vpx * 47, plane_stride, row_pitch,
                                               qpr, (float4 *)((float *)o + (size_t)v * vpx * 3), ov + (size_t)v * vpx / 4, nq);
                        }
                        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                        CHK(hipEventElapsedTime(&ms, e0, e1));
                    }
                    CHK(hipGetLastError());
                    const double us = ms * 1e3 / launches;
                    printf("%-16s %2d view(s) per launch%s: %8.2f us  %6.0f GB/s  frac of 8 TB/s %.3f\n", inter ? "row-interleaved" : "planar", batch,
                           batch == 1 ? " (8 views round robin, HBM)" : "", us, 60.0 * vpx * batch / us / 1e3, 60.0 * vpx * batch / us / 1e3 / 8000.0);
                }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "oneview_cold")) {
        // round 4: the data path of a one-view launch FROM HBM -- 8 resident views (8 x 97.5 MB of planes + 8 x 27 MB of results:
        // three times the 256 MiB Infinity Cache), one view per launch, a different one each launch; the shipped access forms
        // (nt dword loads, whole 1-KiB nt stores: k_scope<P, 1, 0>) and the plain ones; next to it the same view over and over
        // (cache resident, what `oneview` measures).  P = 47 planes (46 + the valid-map byte) or 55 (+ the 8 B/px camera table).
        const size_t vpx = 1920 * 1080, nq = vpx / 4;
        const int NV = 8;
        uint8_t *fr; float4 *o; unsigned *ov;
        CHK(hipMalloc(&fr, vpx * 55 * NV + 64)); CHK(hipMalloc(&o, vpx * 12 * NV)); CHK(hipMalloc(&ov, vpx * NV));
        CHK(hipMemset(fr, 0x5a, vpx * 55 * NV + 64));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++)
            for (int planes : {47, 55})
                for (int form = 0; form < 2; form++)
                    for (int cold = 0; cold < 2; cold++) {
                        const int launches = 400;
                        float ms = 0;
                        for (int pass = 0; pass < 2; pass++) {
                            CHK(hipEventRecord(e0));
                            for (int i = 0; i < launches; i++) {
                                const int v = cold ? i % NV : 0;
                                const uint8_t *f = fr + (size_t)v * vpx * 55;
                                float4 *oo = (float4 *)((float *)o + (size_t)v * vpx * 3);
                                unsigned *vv = ov + (size_t)v * vpx / 4;
                                if (planes == 47 && form == 0) hipLaunchKernelGGL((k_scope<47, 1, 0>), dim3((nq + 255) / 256), dim3(256), 0, 0, f, vpx, oo, vv, nq);
                                if (planes == 47 && form == 1) hipLaunchKernelGGL((k_scope<47, 0, 4>), dim3((nq + 255) / 256), dim3(256), 0, 0, f, vpx, oo, vv, nq);
                                if (planes == 55 && form == 0) hipLaunchKernelGGL((k_scope<55, 1, 0>), dim3((nq + 255) / 256), dim3(256), 0, 0, f, vpx, oo, vv, nq);
                                if (planes == 55 && form == 1) hipLaunchKernelGGL((k_scope<55, 0, 4>), dim3((nq + 255) / 256), dim3(256), 0, 0, f, vpx, oo, vv, nq);
                            }
                            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                            CHK(hipEventElapsedTime(&ms, e0, e1));
                        }
                        CHK(hipGetLastError());
                        const double us = ms * 1e3 / launches, bytes = (double)vpx * (planes + 13);
                        printf("one view per launch, %d B/px read, %s, %s: %6.2f us per launch  %6.0f GB/s moved  frac of 8 TB/s on 60 B/px %.3f\n", planes,
                               form == 0 ? "nt loads + whole-line nt stores" : "plain loads + plain whole-line stores", cold ? "8 views round robin (HBM)" : "same view (Infinity Cache)",
                               us, bytes / us / 1e3, 60.0 * vpx / us / 1e3 / 8000.0);
                    }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "oneview")) {
        const size_t vpx = 1920 * 1080, nq = vpx / 4;
        uint8_t *fr; float4 *o; unsigned *ov;
        CHK(hipMalloc(&fr, vpx * P + 64)); CHK(hipMalloc(&o, vpx * 12)); CHK(hipMalloc(&ov, vpx));
        CHK(hipMemset(fr, 0x5a, vpx * P + 64));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++)
            for (int split = 0; split < 2; split++)
                for (int lds_kb : {0, 20, 40, 52}) {  // 160 KB per CU: 0 -> 8 blocks (VGPR-limited), 20 -> 8, 40 -> 4, 52 -> 3 blocks of 4 waves
                    const int launches = 300;
                    for (int pass = 0; pass < 2; pass++) {
                        CHK(hipEventRecord(e0));
                        for (int i = 0; i < launches; i++) {
                            if (split) hipLaunchKernelGGL((k_one<P, true>), dim3((nq + 255) / 256), dim3(256), lds_kb * 1024, 0, fr, vpx, o, ov, nq, lds_kb);
                            else hipLaunchKernelGGL((k_one<P, false>), dim3((nq + 255) / 256), dim3(256), lds_kb * 1024, 0, fr, vpx, o, ov, nq, lds_kb);
                        }
                        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                        if (pass) printf("one view per launch, %s, %2d KB LDS per block: %6.2f us per launch\n", split ? "planes in two dependent halves" : "all planes at once", lds_kb, ms * 1e3 / launches);
                    }
                    CHK(hipGetLastError());
                }
        return 0;
    }
    size_t npx = (size_t)(argc > 1 ? atof(argv[1]) : 33.1776) * 1000000;
    npx = (npx + 4095) / 4096 * 4096;
    uint8_t *in; float4 *out; unsigned *outv;
    CHK(hipMalloc(&in, npx * P)); CHK(hipMalloc(&out, npx * 12)); CHK(hipMalloc(&outv, npx));
    std::vector<uint8_t> h(1 << 20);
    for (auto &x : h) x = rand();
    for (size_t o = 0; o < npx * P; o += h.size()) CHK(hipMemcpy(in + o, h.data(), std::min(h.size(), npx * P - o), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const double bytes = (double)npx * (P + 13);
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9;
        for (int it = 0; it < 12; it++) {
            CHK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_dword<P>, dim3((npx / 4 + 255) / 256), dim3(256), 0, 0, in, npx, out, outv, npx / 4);
            if (mode == 1) hipLaunchKernelGGL(k_dwordx4<P>, dim3((npx / 16 + 255) / 256), dim3(256), 0, 0, in, npx, out, (uint4 *)outv, npx / 16);
            if (mode == 2) hipLaunchKernelGGL(k_lds<P>, dim3(npx / 1024), dim3(256), P * 1024, 0, in, npx, out, outv, npx / 4);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2 && ms < best) best = ms;
        }
        CHK(hipGetLastError());
        const char *names[] = {"dword/lane (256 B per wave-load)", "dwordx4/lane (1 KiB per wave-load)", "dwordx4 global->LDS DMA + ds_read_b32"};
        printf("mode %d %-40s %8.3f ms  %7.1f GB/s  %6.1f Gpx/s  (%.1f%% of 8 TB/s)\n", mode, names[mode], best, bytes / best / 1e6, npx / best / 1e6,
               bytes / best / 1e6 / 80.0);
    }
    // ---- incremental model: 16 views of 1920x1080 ----
    {
        const int nviews = 16; const size_t vpx = 1920 * 1080, nq = vpx / 4;
        uint8_t *fr, *mk; float *tab; float4 *o; unsigned *ov;
        CHK(hipMalloc(&fr, vpx * P * nviews)); CHK(hipMalloc(&mk, vpx * nviews + 64)); CHK(hipMalloc(&tab, 511 * 1021 * 4));
        CHK(hipMalloc(&o, vpx * 12 * nviews)); CHK(hipMalloc(&ov, vpx * nviews));
        for (size_t off = 0; off < vpx * P * nviews; off += h.size()) CHK(hipMemcpy(fr + off, h.data(), std::min(h.size(), vpx * P * nviews - off), hipMemcpyHostToDevice));
        CHK(hipMemset(mk, 1, vpx * nviews + 64)); CHK(hipMemset(tab, 0, 511 * 1021 * 4));
        const double by = (double)vpx * nviews * (P + 14);
        for (int occ = 0; occ < 2; occ++)
            for (int flags : {0, 1, 65, 32, 33, 97, 11, 43, 107, 16, 2, 8}) {
                float best = 1e9;
                for (int it = 0; it < 8; it++) {
                    dim3 grid((nq + 255) / 256, (flags & 1) ? nviews / ((flags & 64) ? 8 : 4) : nviews);
                    CHK(hipEventRecord(e0));
                    if (occ == 0) hipLaunchKernelGGL((k_steps<P, 8>), grid, dim3(256), 0, 0, fr, vpx, vpx * P, nviews, mk + 32, tab, o, ov, nq, flags);
                    else hipLaunchKernelGGL((k_steps<P, 4>), grid, dim3(256), 0, 0, fr, vpx, vpx * P, nviews, mk + 32, tab, o, ov, nq, flags);
                    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                    if (it >= 2 && ms < best) best = ms;
                }
                CHK(hipGetLastError());
                printf("steps occ>=%d flags=%2d  %8.3f ms  %7.1f GB/s  %6.1f Gpx/s\n", occ == 0 ? 8 : 4, flags, best, by / best / 1e6, vpx * nviews / best / 1e6);
            }
        // persistent, view-major walk (round 3), 4 waves/SIMD like the fused kernel: with and without the per-step table read
        double *ct;
        CHK(hipMalloc(&ct, vpx * 8));
        CHK(hipMemset(ct, 0, vpx * 8));
        for (int slots : {1024, 2048, 512})
            for (int use_tab = 0; use_tab < 2; use_tab++) {
                float best = 1e9;
                for (int it = 0; it < 8; it++) {
                    CHK(hipEventRecord(e0));
                    hipLaunchKernelGGL((k_viewmajor<P, 4>), dim3(slots), dim3(256), 0, 0, fr, vpx, vpx * P, nviews, ct, o, ov, nq, use_tab);
                    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                    if (it >= 2 && ms < best) best = ms;
                }
                CHK(hipGetLastError());
                printf("view-major persistent, %4d blocks, table %d  %8.3f ms  %7.1f GB/s  %6.1f Gpx/s\n", slots, use_tab, best, by / best / 1e6, vpx * nviews / best / 1e6);
            }
        for (int slots : {768, 1024, 512}) {
            float best = 1e9;
            for (int it = 0; it < 8; it++) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL((k_viewmajor_pipe<P, 3>), dim3(slots), dim3(256), 0, 0, fr, vpx, vpx * P, nviews, o, ov, nq);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 2 && ms < best) best = ms;
            }
            CHK(hipGetLastError());
            printf("view-major persistent, next loads before the stores, %4d blocks  %8.3f ms  %7.1f GB/s  %6.1f Gpx/s\n", slots, best, by / best / 1e6, vpx * nviews / best / 1e6);
        }
        // ticketed persistent waves (round 3)
        unsigned *cnt;
        CHK(hipMalloc(&cnt, 4));
        for (int pre = 0; pre < 2; pre++)
            for (int slots : {1024})
                for (int use_tab = 0; use_tab < 1; use_tab++) {
                    float best = 1e9;
                    for (int it = 0; it < 4; it++) {
                        CHK(hipMemsetAsync(cnt, 0, 4, 0));
                        CHK(hipEventRecord(e0));
                        if (pre) hipLaunchKernelGGL((k_ticket<P, 3, true>), dim3(slots), dim3(256), 0, 0, fr, vpx, vpx * P, nviews, ct, o, ov, nq, use_tab, cnt);
                        else hipLaunchKernelGGL((k_ticket<P, 4, false>), dim3(slots), dim3(256), 0, 0, fr, vpx, vpx * P, nviews, ct, o, ov, nq, use_tab, cnt);
                        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                        if (it >= 2 && ms < best) best = ms;
                    }
                    CHK(hipGetLastError());
                    printf("ticketed persistent waves, prefetch %d, %4d blocks, table %d  %8.3f ms  %7.1f GB/s  %6.1f Gpx/s\n", pre, slots, use_tab, best, by / best / 1e6,
                           vpx * nviews / best / 1e6);
                }
    }
    return 0;
}
