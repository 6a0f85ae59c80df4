// Probe 2 (round 4): issue rates that decide the fp8 attention / fp8 GEMM schedules.
//   a. MFMA issue rate by shape with ONE and TWO waves per SIMD (cycles per MFMA per SIMD, wall-clock TFLOP/s, held clock):
//      v_mfma_scale_f32_32x32x64_f8f6f4, v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3), v_mfma_f32_32x32x16_bf16
//   b. VALU instruction rates alone: v_add_f32, v_cvt_pknorm_u16_f32, v_perm_b32, v_cvt_pk_u8_f32, v_exp_f32, v_cvt_pk_fp8_f32, v_max3_f32
//   c. rounding of v_cvt_pknorm_u16_f32
//   d. 32x32x64 MFMA with G groups of {4 v_add_f32, 2 v_cvt_pknorm_u16_f32, 1 v_perm_b32} beside it
// Build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/micro/fp8_attn_probe2.hip -o tools/micro/fp8_attn_probe2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// SHAPE 0: 32x32x64 f8f6f4, 1: 16x16x128 f8f6f4, 2: 32x32x16 bf16.  G: filler groups per MFMA (SHAPE 0 only)
template <int SHAPE, int G>
__global__ __launch_bounds__(512) void mfma_rate(float* sink, unsigned long long* cyc, int iters, unsigned seed) {
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        unsigned x = seed * 2654435761u + threadIdx.x * 40503u + i * 97u;
        x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
        a[i] = (int)(x & 0x77777777u);               // random e4m3 / bf16 bit patterns, finite
        x *= 0x9E3779B1u; x ^= x >> 11;
        b[i] = (int)(x & 0x77777777u);
    }
    f32x16 acc[4];
    f32x4 acc4[4];
    for (int q = 0; q < 4; ++q) { for (int e = 0; e < 16; ++e) acc[q][e] = 0.f; for (int e = 0; e < 4; ++e) acc4[q][e] = 0.f; }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = (float)(threadIdx.x & 7) * 0.0001f + i * 0.00001f;
    unsigned pk[4] = {0, 0, 0, 0};
    const float kc = 0.0003f;
    __syncthreads();
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (SHAPE == 0)
                asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[q]) : "v"(a), "v"(b), "v"(0x7F7F7F7F));
            else if (SHAPE == 1)
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc4[q]) : "v"(a), "v"(b), "v"(0x7F7F7F7F));
            else {
                const bf16x8 aa = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, a, 0, 1, 2, 3));
                const bf16x8 bb = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b, b, 0, 1, 2, 3));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(aa), "v"(bb));
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int j = ((q * G + g) * 4) & 15;
                float y0, y1, y2, y3;
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(y0) : "v"(x[j]), "v"(kc));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(y1) : "v"(x[j + 1]), "v"(kc));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(y2) : "v"(x[j + 2]), "v"(kc));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(y3) : "v"(x[j + 3]), "v"(kc));
                unsigned u0, u1;
                asm volatile("v_cvt_pknorm_u16_f32 %0, %1, %2" : "=v"(u0) : "v"(y0), "v"(y1));
                asm volatile("v_cvt_pknorm_u16_f32 %0, %1, %2" : "=v"(u1) : "v"(y2), "v"(y3));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pk[g & 3]) : "v"(u1), "v"(u0), "v"(0x06040200));
            }
        }
    }
    const unsigned long long t1 = now();
    float s = 0.f;
    for (int q = 0; q < 4; ++q) { for (int e = 0; e < 16; ++e) s += acc[q][e]; for (int e = 0; e < 4; ++e) s += acc4[q][e]; }
    for (int i = 0; i < 4; ++i) s += (float)pk[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int G>
static void run_mfma(float* sink, unsigned long long* cyc, int threads, const char* name, double flop_per_mfma) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_rate<SHAPE, G>), dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters, 1u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mfma_rate<SHAPE, G>), dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters, 2u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    s /= blocks;
    const double nm = iters * 4.0, wps = threads / 256.0;
    const double tf = flop_per_mfma * nm * (threads / 64.0) * blocks / (ms * 1e-3) / 1e12;
    printf("  %-22s G=%d waves/SIMD %.0f : %.1f cyc per MFMA per wave, %.1f per SIMD ; wall %.2f ms = %.0f TFLOP/s ; clock %.2f GHz\n", name, G, wps,
           s / nm, s / nm / wps, ms, tf, s / (ms * 1e-3) / 1e9);
}

// OP: 0 v_add_f32, 1 v_cvt_pknorm_u16_f32, 2 v_perm_b32, 3 v_cvt_pk_u8_f32, 4 v_exp_f32, 5 v_cvt_pk_fp8_f32, 6 v_max3_f32, 7 v_fma_f32
template <int OP>
__global__ __launch_bounds__(512) void valu_rate(float* sink, unsigned long long* cyc, int iters) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.25f + threadIdx.x * 1e-4f + i * 1e-3f;
    const float kc = 0.37f;
    __syncthreads();
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (OP == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[(j + 5) & 15]), "v"(kc));
            if (OP == 1) asm volatile("v_cvt_pknorm_u16_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[(j + 5) & 15]), "v"(kc));
            if (OP == 2) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x[j]) : "v"(x[(j + 5) & 15]), "v"(kc), "v"(0x06040200));
            if (OP == 3) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(x[j]) : "v"(x[(j + 5) & 15]), "v"(kc));
            if (OP == 4) asm volatile("v_exp_f32 %0, %1" : "=v"(x[j]) : "v"(x[(j + 5) & 15]));
            if (OP == 5) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[(j + 5) & 15]), "v"(kc));
            if (OP == 6) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(x[j]) : "v"(x[(j + 5) & 15]), "v"(kc), "v"(x[(j + 9) & 15]));
            if (OP == 7) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[j]) : "v"(x[(j + 5) & 15]), "v"(kc), "v"(x[(j + 9) & 15]));
        }
    }
    const unsigned long long t1 = now();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
static void run_valu(float* sink, unsigned long long* cyc, const char* name) {
    const int iters = 4000, blocks = 256;
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL((valu_rate<OP>), dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        s /= blocks;
        printf("  %-24s waves/SIMD %d : %.2f cycles per instruction per wave, %.2f per SIMD\n", name, threads / 256, s / (iters * 16.0),
               s / (iters * 16.0) / (threads / 256));
    }
}

__global__ void pknorm_probe(const float* x, int n, unsigned* out) {
    const int i = threadIdx.x;
    if (i >= n) return;
    u16x2 u = __builtin_amdgcn_cvt_pknorm_u16(x[i] * (1.0f / 65535.0f), 0.f);
    out[i] = u[0];
}

int main() {
    float* sink; unsigned long long* cyc;
    hipMalloc(&sink, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    printf("[a] MFMA issue rates (random finite operand bits), all 256 CUs\n");
    for (int threads : {256, 512}) {
        run_mfma<0, 0>(sink, cyc, threads, "scale_32x32x64 e4m3", 2.0 * 32 * 32 * 64);
        run_mfma<1, 0>(sink, cyc, threads, "scale_16x16x128 e4m3", 2.0 * 16 * 16 * 128);
        run_mfma<2, 0>(sink, cyc, threads, "32x32x16 bf16", 2.0 * 32 * 32 * 16);
    }
    printf("[d] 32x32x64 e4m3 MFMA + G x {4 v_add_f32, 2 v_cvt_pknorm_u16_f32, 1 v_perm_b32}\n");
    for (int threads : {256, 512}) {
        run_mfma<0, 1>(sink, cyc, threads, "scale_32x32x64 e4m3", 2.0 * 32 * 32 * 64);
        run_mfma<0, 2>(sink, cyc, threads, "scale_32x32x64 e4m3", 2.0 * 32 * 32 * 64);
        run_mfma<0, 3>(sink, cyc, threads, "scale_32x32x64 e4m3", 2.0 * 32 * 32 * 64);
        run_mfma<0, 4>(sink, cyc, threads, "scale_32x32x64 e4m3", 2.0 * 32 * 32 * 64);
    }
    printf("[b] VALU instruction rates alone\n");
    run_valu<0>(sink, cyc, "v_add_f32");
    run_valu<7>(sink, cyc, "v_fma_f32");
    run_valu<1>(sink, cyc, "v_cvt_pknorm_u16_f32");
    run_valu<2>(sink, cyc, "v_perm_b32");
    run_valu<3>(sink, cyc, "v_cvt_pk_u8_f32");
    run_valu<4>(sink, cyc, "v_exp_f32");
    run_valu<5>(sink, cyc, "v_cvt_pk_fp8_f32");
    run_valu<6>(sink, cyc, "v_max3_f32");
    {
        const float xs[] = {0.f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 3.5f, 119.5f, 120.5f, 120.49f, 126.f, 255.4f, -3.f, 70000.f};
        const int n = sizeof xs / sizeof xs[0];
        float* dx; unsigned* d1;
        hipMalloc(&dx, n * 4); hipMalloc(&d1, n * 4);
        hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(pknorm_probe, dim3(1), dim3(64), 0, 0, dx, n, d1);
        std::vector<unsigned> h(n);
        hipMemcpy(h.data(), d1, n * 4, hipMemcpyDeviceToHost);
        printf("[c] v_cvt_pknorm_u16_f32(x / 65535):");
        for (int i = 0; i < n; ++i) printf("  %g->%u", xs[i], h[i]);
        printf("\n");
    }
    return 0;
}
