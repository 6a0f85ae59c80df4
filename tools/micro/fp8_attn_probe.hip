// Probe (round 4) for the fp8 attention kernel: everything the design assumes about gfx950 that the two guides do not state.
//   1. operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands (lane l: row / column l & 31, k = 32 (l >> 5) + byte)
//   2. which lane's scale byte applies to which (row, 32-element k-block), for A and for B; op_sel / op_sel_hi byte selection
//   3. v_cvt_pk_u8_f32 rounding and saturation; v_cvt_pk_fp8_f32 rounding / saturation; v_cvt_scalef32_pk_fp8_f32
//   4. e4m3 decode inside the MFMA (subnormals)
//   5. issue-rate micro-benchmark: cycles per scaled 32x32x64 MFMA with F x (v_fma_f32 + v_cvt_pk_u8_f32) or
//      F x (v_fma_f32 + v_exp_f32 + 1/2 v_cvt_pk_fp8_f32) beside it, one and two waves per SIMD
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/fp8_attn_probe.hip -o tools/micro/fp8_attn_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

static float e4m3(uint8_t v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x;
    if (e == 0) x = ldexpf((float)m / 8.f, -6);
    else if (e == 15 && m == 7) x = NAN;
    else x = ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}

// D[32][32] = A[32][64] . B[32][64]^T with per-lane scale registers and op_sel values given at compile time
template <int OA, int OB>
__global__ void mfma_probe(const uint8_t* A, const uint8_t* B, const int* sa, const int* sb, float* D) {
    const int l = threadIdx.x;
    v8i a, b;
    const int* ap = (const int*)(A + (l & 31) * 64 + (l >> 5) * 32);
    const int* bp = (const int*)(B + (l & 31) * 64 + (l >> 5) * 32);
    for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OA, sa[l], OB, sb[l]);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

__global__ void cvt_probe(const float* x, int n, unsigned* u8out, unsigned* fp8out, unsigned* sfp8out, float scale) {
    const int i = threadIdx.x;
    if (i >= n) return;
    unsigned u = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(u) : "v"(x[i]));
    u8out[i] = u;
    fp8out[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.f, 0, false) & 0xFFu;
    s16x2 old = {0, 0};
    s16x2 q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[i], 0.f, scale, false);
    sfp8out[i] = (unsigned)q[0] & 0xFFu;
}

// ---- issue-rate micro-benchmark ----
// MODE 0: F x (v_fma_f32, v_cvt_pk_u8_f32) per MFMA; MODE 1: F x (v_fma_f32, v_exp_f32) + F/2 x v_cvt_pk_fp8_f32 per MFMA
template <int MODE, int F>
__global__ __launch_bounds__(512) void rate_probe(float* sink, unsigned long long* cyc, int iters, float seed) {
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x * 0x01000100 + i; b[i] = 0x30303030 + i * 0x00010001; }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed + (float)(threadIdx.x & 7) * 0.125f + i;
    unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float c0 = seed * 0.5f, c1 = seed * 0.25f;
    __syncthreads();
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                         : "+v"(acc[q]) : "v"(a), "v"(b), "v"(0x7F7F7F7F));
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int j = (q * F + f) & 15;
                if (MODE == 0) {
                    float y;
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(x[j]), "v"(c0), "v"(c1));
                    asm volatile("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(pk[j & 7]) : "v"(y), "n"(0));
                } else {
                    float y, z;
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(x[j]), "v"(c0), "v"(c1));
                    asm volatile("v_exp_f32 %0, %1" : "=v"(z) : "v"(y));
                    x[j] = z;
                    if (f & 1) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "=v"(pk[j & 7]) : "v"(x[j]), "v"(x[(j + 15) & 15]));
                }
            }
        }
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) s += acc[q][e];
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += (float)pk[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int F>
static void run_rate(float* sink, unsigned long long* cyc, int threads) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((rate_probe<MODE, F>), dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters, 1.0f);
    hipLaunchKernelGGL((rate_probe<MODE, F>), dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    s /= blocks;
    printf("  mode %d  F=%2d  waves/SIMD %d : %.1f cycles per MFMA (per wave)   -> per SIMD per MFMA %.1f\n", MODE, F, threads / 256,
           s / (iters * 4.0), s / (iters * 4.0) / (threads / 256));
}

int main() {
    // ---------- 1. data layout ----------
    std::vector<uint8_t> A(32 * 64), B(32 * 64);
    srand(3);
    auto small = [] { const uint8_t t[7] = {0x00, 0x38, 0x40, 0x44, 0xB8, 0xC0, 0x30}; return t[rand() % 7]; };   // 0, 1, 2, 3, -1, -2, .5
    for (auto& v : A) v = small();
    for (auto& v : B) v = small();
    uint8_t *dA, *dB; float* dD; int *dsa, *dsb;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 1024 * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    std::vector<int> sa(64, 0x7F7F7F7F), sb(64, 0x7F7F7F7F);
    std::vector<float> D(1024);
    auto run = [&](int oa, int ob) {
        hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        if (oa == 0 && ob == 0) hipLaunchKernelGGL((mfma_probe<0, 0>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        if (oa == 1 && ob == 0) hipLaunchKernelGGL((mfma_probe<1, 0>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        if (oa == 2 && ob == 0) hipLaunchKernelGGL((mfma_probe<2, 0>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        if (oa == 3 && ob == 0) hipLaunchKernelGGL((mfma_probe<3, 0>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        if (oa == 0 && ob == 3) hipLaunchKernelGGL((mfma_probe<0, 3>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    };
    run(0, 0);
    {
        double e1 = 0, e2 = 0, nrm = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int k = 0; k < 64; ++k) s += (double)e4m3(A[i * 64 + k]) * e4m3(B[j * 64 + k]);
            e1 += fabs(D[i * 32 + j] - s); e2 += fabs(D[j * 32 + i] - s); nrm += fabs(s);
        }
        printf("[1] 32x32x64 e4m3: err if D[row=a][col=b] %.3e ; transposed %.3e  (nrm %.1f)\n", e1 / nrm, e2 / nrm, nrm);
    }
    // ---------- 2. scales: all ones data; A scale of lane l = 127 + (l>>5 ? 3 : 0) + ((l&31)==5 ? 1 : 0) ----------
    for (auto& v : A) v = 0x38;
    for (auto& v : B) v = 0x38;
    for (int l = 0; l < 64; ++l) sa[l] = 127 + ((l >> 5) ? 3 : 0) + (((l & 31) == 5) ? 1 : 0);
    run(0, 0);
    printf("[2a] A scale by lane (row 5: x2, k-block 1: x8): D[0][0] %.0f  D[5][0] %.0f  D[0][5] %.0f  (expect 288 576 288)\n", D[0], D[5 * 32], D[5]);
    // zero the SECOND 16 bytes of every row's first 32 (k = 16..31): under the contiguous hypothesis D = 16 + 32*8 = 272
    for (int i = 0; i < 32; ++i) for (int k = 16; k < 32; ++k) A[i * 64 + k] = 0;
    run(0, 0);
    printf("[2b] A[:, 16:32] = 0: D[0][0] %.0f (expect 272: k 16..31 belongs to block 0)\n", D[0]);
    for (auto& v : A) v = 0x38;
    for (int l = 0; l < 64; ++l) sa[l] = 0x7F7F7F7F;
    for (int l = 0; l < 64; ++l) sb[l] = 127 + ((l >> 5) ? 2 : 0) + (((l & 31) == 7) ? 1 : 0);
    run(0, 0);
    printf("[2c] B scale by lane (col 7: x2, k-block 1: x4): D[0][0] %.0f  D[0][7] %.0f  D[7][0] %.0f  (expect 160 320 160)\n", D[0], D[7], D[7 * 32]);
    // op_sel: scale register bytes {1, 2, 4, 8} -> D = 64 * scale
    for (int l = 0; l < 64; ++l) { sa[l] = 0x7F | (0x80 << 8) | (0x81 << 16) | (0x82 << 24); sb[l] = 0x7F7F7F7F; }
    for (int o = 0; o < 4; ++o) { run(o, 0); printf("[2d] opsel_a %d -> D[0][0] %.0f (expect %d)\n", o, D[0], 64 << o); }
    for (int l = 0; l < 64; ++l) { sb[l] = 0x7F | (0x80 << 8) | (0x81 << 16) | (0x82 << 24); sa[l] = 0x7F7F7F7F; }
    run(0, 3); printf("[2e] opsel_b 3 -> D[0][0] %.0f (expect 512)\n", D[0]);
    // ---------- 4. decode of every positive byte: A[row r][k 0] = byte, B[col][k 0] = 1 ----------
    for (int l = 0; l < 64; ++l) { sa[l] = 0x7F7F7F7F; sb[l] = 0x7F7F7F7F; }
    int bad = 0;
    for (int t = 0; t < 4; ++t) {
        for (auto& v : A) v = 0;
        for (auto& v : B) v = 0;
        for (int r = 0; r < 32; ++r) { A[r * 64] = (uint8_t)(t * 32 + r); B[r * 64] = 0x38; }
        if (t == 3) A[31 * 64] = 0;       // 0x7F = NaN
        run(0, 0);
        for (int r = 0; r < 32; ++r) {
            const float want = e4m3(A[r * 64]);
            if (D[r * 32] != want) { ++bad; printf("    byte 0x%02x decodes to %g, expected %g\n", A[r * 64], D[r * 32], want); }
        }
    }
    printf("[4] e4m3 decode inside the MFMA: %d of 127 bytes differ from the OCP table\n", bad);
    // ---------- 3. conversions ----------
    {
        const float xs[] = {0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 126.5f, 254.4f, 254.5f, 255.5f, 300.f, -0.6f, -5.f, 1e9f,
                            448.f, 449.f, 463.9f, 464.f, 480.f, 500.f, 1e6f, 0.001953125f, 0.0009765625f, 0.0029296875f, 0.00146484375f, 17.f, 17.1f, 18.9f, 19.f,
                            0.0625f, 0.017f};
        const int n = sizeof xs / sizeof xs[0];
        float* dx; unsigned *d1, *d2, *d3;
        hipMalloc(&dx, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4); hipMalloc(&d3, n * 4);
        hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, n, d1, d2, d3, 4.0f);
        std::vector<unsigned> h1(n), h2(n), h3(n);
        hipMemcpy(h1.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), d2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h3.data(), d3, n * 4, hipMemcpyDeviceToHost);
        printf("[3] x -> v_cvt_pk_u8_f32 | v_cvt_pk_fp8_f32 (value) | v_cvt_scalef32_pk_fp8_f32 scale 4 (value)\n");
        for (int i = 0; i < n; ++i)
            printf("    %-14g -> %3u | 0x%02x (%g) | 0x%02x (%g)\n", xs[i], h1[i] & 0xFF, h2[i], e4m3((uint8_t)h2[i]), h3[i], e4m3((uint8_t)h3[i]));
    }
    // ---------- 5. issue rates ----------
    {
        float* sink; unsigned long long* cyc;
        hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
        printf("[5] cycles per v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3) with VALU fillers, all 256 CUs busy\n");
        for (int threads : {256, 512}) {
            run_rate<0, 0>(sink, cyc, threads);
            run_rate<0, 2>(sink, cyc, threads);
            run_rate<0, 4>(sink, cyc, threads);
            run_rate<0, 6>(sink, cyc, threads);
            run_rate<0, 8>(sink, cyc, threads);
            run_rate<0, 12>(sink, cyc, threads);
            run_rate<1, 2>(sink, cyc, threads);
            run_rate<1, 4>(sink, cyc, threads);
            run_rate<1, 6>(sink, cyc, threads);
            run_rate<1, 8>(sink, cyc, threads);
        }
    }
    return 0;
}
