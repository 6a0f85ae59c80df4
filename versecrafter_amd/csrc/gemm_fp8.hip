// fp8 operands for the GEMM (BASELINE config 5 names "fp8 MFMA"; the reference has no fp8 compute path -- its fp8 mode is weight
// STORAGE with bf16 arithmetic, CLI.py:292-301 -- so this is a capability of this build, off by default, parity unpinned by nature).
//   vc_op_quantize_rows_fp8   bf16 [M, K] -> OCP e4m3 [M, K] + one fp32 scale per row: scale = amax / 448 (1 for an all-zero row),
//                             q = round_to_e4m3(x / scale).  Per token for activations, per output channel for nn.Linear weights.
//   vc_op_gemm_fp8            C = epilogue((A_q W_q^T) * a_scale[m] * w_scale[n]) through the ping-pong kernel's FP8 instantiation
//                             (gemm_bf16.hip: v_mfma_scale_f32_16x16x128_f8f6f4 at unit block scales, fp32 accumulation).
#include <stdint.h>
#include <string.h>

#include "../../include/vcengine.h"
#include "vc_common.h"
#include "vc_kernels.h"

namespace {

// one 256-thread workgroup per row; 16 bytes (8 bf16) per thread and trip.  The row stays in registers between the amax pass and the
// conversion (NCH trips of 16 bytes per thread: K <= 2048 NCH), so it is read once and every load is issued before the first store (a
// load behind a store is waited for together with the store: in-order vmcnt).  K > 16384 takes the two-pass form (NCH = 0).
template <int NCH>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q, int64_t ldq,
                                                                float* __restrict__ scale, int M, int K) {
    __shared__ float red[4];
    const int nch = K / 8;
    for (int64_t m = blockIdx.x; m < M; m += gridDim.x) {
        const bf16_t* row = x + m * ldx;
        float amax = 0.f;
        uint4 keep[NCH > 0 ? NCH : 1];
        auto amax8 = [&](const uint4& v) {
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                amax = fmaxf(amax, fabsf(__uint_as_float(w[e] << 16)));
                amax = fmaxf(amax, fabsf(__uint_as_float(w[e] & 0xFFFF0000u)));
            }
        };
        if (NCH > 0) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = threadIdx.x + i * 256;
                keep[i] = c < nch ? *(const uint4*)(row + c * 8) : uint4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) amax8(keep[i]);
        } else {
            for (int c = threadIdx.x; c < nch; c += 256) amax8(*(const uint4*)(row + c * 8));
        }
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_down(amax, o));
        __syncthreads();                                   // red[] of the previous row has been read
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
        const float inv = 1.0f / sc;
        if (threadIdx.x == 0) scale[m] = sc;
        uint8_t* qrow = q + m * ldq;
        auto cvt8 = [&](const uint4& v, int c) {
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
            unsigned out[2];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = __uint_as_float(w[e] << 16) * inv, hi = __uint_as_float(w[e] & 0xFFFF0000u) * inv;
                const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(lo, hi, 0, false);        // two e4m3 in the low 16 bits
                if (e & 1) out[e >> 1] |= ((unsigned)pk & 0xFFFFu) << 16;
                else out[e >> 1] = (unsigned)pk & 0xFFFFu;
            }
            *(uint2*)(qrow + c * 8) = uint2{out[0], out[1]};
        };
        if (NCH > 0) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = threadIdx.x + i * 256;
                if (c < nch) cvt8(keep[i], c);
            }
        } else {
            for (int c = threadIdx.x; c < nch; c += 256) cvt8(*(const uint4*)(row + c * 8), c);
        }
    }
}

}  // namespace

int vc_launch_quantize_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int M, int K, hipStream_t stream) {
    if (!x || !q || !scale || M <= 0 || K <= 0) return VC_E_INVALID;
    if (K % 8 || ldx % 8 || ldq % 8 || ((uintptr_t)x & 15) || ((uintptr_t)q & 7)) return VC_E_UNSUPPORTED;
    const int grid = M < (1 << 20) ? M : (1 << 20);
    const int trips = (K / 8 + 255) / 256;
#define VC_QROWS(N) hipLaunchKernelGGL(quantize_rows_fp8_kernel<N>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, ldx, (uint8_t*)q, ldq, scale, M, K)
    if (trips <= 1) VC_QROWS(1);
    else if (trips <= 3) VC_QROWS(3);          // d = 5120: 640 chunks
    else if (trips <= 8) VC_QROWS(8);          // ffn 13824: 1728 chunks
    else VC_QROWS(0);
#undef VC_QROWS
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}

extern "C" {

int vc_op_quantize_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, void* scale, int M, int K, void* stream) {
    return vc_launch_quantize_rows_fp8(x, ldx, q, ldq, (float*)scale, M, K, (hipStream_t)stream);
}

int vc_op_gemm_fp8(const void* A, int64_t lda, const void* a_scale, const void* W, int64_t ldw, const void* w_scale, void* C, int64_t ldc,
                   const void* bias, int M, int N, int K, int epilogue, const void* resid, int64_t ldr, const void* gate,
                   int64_t gate_bstride, int rows_per_batch, int a_rows_padded, void* stream) {
    VcGemmParams p;
    memset(&p, 0, sizeof p);
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc; p.bias = bias; p.M = M; p.N = N; p.K = K;
    p.epilogue = epilogue; p.resid = resid; p.ldr = ldr; p.gate = gate; p.gate_bstride = gate_bstride;
    p.rows_per_batch = rows_per_batch; p.valid_rows = -1; p.a_rows_padded = a_rows_padded;
    p.fp8 = 1; p.a_scale = (const float*)a_scale; p.w_scale = (const float*)w_scale;
    return vc_launch_gemm(p, (hipStream_t)stream);
}

}  // extern "C"
