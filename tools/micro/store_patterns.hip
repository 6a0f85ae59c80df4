// Micro-benchmark (round 2): how fast can one CU store a 256 x 256 bf16 tile (128 KiB) of a row-major [M][ldc] matrix, by the
// shape of a store instruction?  One workgroup of 512 threads per CU walks `tiles` tiles; patterns:
//   0: 8 B per lane, a wave instruction = 16 rows x 32 B   (MFMA fragment layout)
//   1: 16 B per lane, 16 rows x 64 B                       (after the lane-group exchange)
//   2: 16 B per lane, 8 rows x 128 B                       (full lines, needs a transpose through LDS)
//   3: 16 B per lane, 2 rows x 512 B
// build: hipcc -O3 --offload-arch=gfx950 store_patterns.hip -o store_patterns ; run on the GPU box: store_patterns [workgroups [tiles per workgroup]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <int PAT>
__global__ __launch_bounds__(512) void k(uint16_t* C, int64_t ldc, int tiles_per_wg, int ntn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int id = blockIdx.x * tiles_per_wg + t;
        const int tm = id / ntn, tn = id % ntn;
        uint16_t* base = C + (int64_t)tm * 256 * ldc + tn * 256;
        if (PAT == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = wr * 128 + i * 16 + (lane & 15), n = wc * 64 + j * 16 + (lane >> 4) * 4;
                    *(uint2*)(base + (int64_t)m * ldc + n) = uint2{(unsigned)m, (unsigned)n};
                }
        } else if (PAT == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const int g = lane >> 4;
                    const int m = wr * 128 + i * 16 + (lane & 15), n = wc * 64 + (jp * 2 + (g & 1)) * 16 + (g >> 1) * 8;
                    *(uint4*)(base + (int64_t)m * ldc + n) = uint4{(unsigned)m, (unsigned)n, 0u, 0u};
                }
        } else if (PAT == 2) {
#pragma unroll
            for (int r8 = 0; r8 < 16; ++r8) {
                const int m = wr * 128 + r8 * 8 + (lane >> 3), n = wc * 64 + (lane & 7) * 8;
                *(uint4*)(base + (int64_t)m * ldc + n) = uint4{(unsigned)m, (unsigned)n, 0u, 0u};
            }
        } else {
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                const int m = wave * 32 + r2 * 2 + (lane >> 5), n = (lane & 31) * 8;
                *(uint4*)(base + (int64_t)m * ldc + n) = uint4{(unsigned)m, (unsigned)n, 0u, 0u};
            }
        }
    }
}

int main(int argc, char** argv) {
    const int M = 65536, N = 5120, ntn = N / 256, ntiles = (M / 256) * ntn;
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, per = argc > 2 ? atoi(argv[2]) : ntiles / wgs;   // few workgroups: the per-CU store path, not HBM
    uint16_t* C;
    hipMalloc(&C, (size_t)M * N * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 4; ++pat) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            switch (pat) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(512), 0, 0, C, (int64_t)N, per, ntn); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(512), 0, 0, C, (int64_t)N, per, ntn); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(512), 0, 0, C, (int64_t)N, per, ntn); break;
                default: hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(512), 0, 0, C, (int64_t)N, per, ntn); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("pattern %d: %.3f ms for %d tiles per CU -> %.2f us per 128 KiB tile, %.2f TB/s aggregate\n", pat, best, per,
               best * 1e3 / per, (double)wgs * per * 131072 / best / 1e9);
    }
    return 0;
}
