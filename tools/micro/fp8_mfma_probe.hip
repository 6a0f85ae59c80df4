#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// hypothesis: A operand of 16x16x128: lane l holds row (l & 15), bytes k = (l >> 4) * 32 .. +31 ; B operand likewise for columns;
// D[i][j] with the standard 16x16 map: col = lane & 15 (from B?), row = (lane >> 4) * 4 + r
__global__ void probe(const uint8_t* A /*[16][128]*/, const uint8_t* B /*[16][128] (n-major)*/, float* D /*[16][16]*/) {
    const int l = threadIdx.x;
    v8i a, b;
    const int* ap = (const int*)(A + (l & 15) * 128 + (l >> 4) * 32);
    const int* bp = (const int*)(B + (l & 15) * 128 + (l >> 4) * 32);
    for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
static float e4m3(uint8_t v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x;
    if (e == 0) x = ldexpf((float)m / 8.f, -6);
    else if (e == 15 && m == 7) x = NAN;
    else x = ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}
int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    srand(1);
    for (auto& v : A) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v = 0x30; }
    for (auto& v : B) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v = 0x30; }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    float D[256];
    hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
    // candidates: D[i][j] = sum_k A[i][k] B[j][k]  or transposed
    double e1 = 0, e2 = 0, nrm = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0;
        for (int k = 0; k < 128; ++k) s += (double)e4m3(A[i * 128 + k]) * e4m3(B[j * 128 + k]);
        e1 += fabs(D[i * 16 + j] - s); e2 += fabs(D[j * 16 + i] - s); nrm += fabs(s);
    }
    printf("rel err if D[row=a][col=b]: %.3e   if transposed: %.3e\n", e1 / nrm, e2 / nrm);
    return 0;
}
