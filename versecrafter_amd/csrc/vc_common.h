// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libvcengine.
// Wave size is 64 everywhere; bf16 storage, fp32 math.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define VC_DEVICE __device__ __forceinline__

// Host: hipFuncSetAttribute(max dynamic LDS) once per (kernel, device).  `done` is a function-local static of the launcher: one
// bit per device ordinal (a process that drives several GPUs sets the attribute on each of them).
#include <atomic>
inline bool vc_set_lds_once(std::atomic<uint64_t>& done, const void* kernel, int lds_bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
}

// ---- bf16 <-> f32 -------------------------------------------------------------------------
VC_DEVICE float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
VC_DEVICE float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// round-to-nearest-even pair pack (lowers to v_cvt_pk_bf16_f32 on gfx950; NaN stays NaN)
VC_DEVICE uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
VC_DEVICE float round_bf16(float x) { return (float)((__bf16)x); }

VC_DEVICE void unpack8(const uint4& u, float (&f)[8]) {
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x);
    f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
    f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z);
    f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
}
VC_DEVICE uint4 pack8(const float (&f)[8]) {
    uint4 u;
    u.x = pack_bf16x2(f[0], f[1]);
    u.y = pack_bf16x2(f[2], f[3]);
    u.z = pack_bf16x2(f[4], f[5]);
    u.w = pack_bf16x2(f[6], f[7]);
    return u;
}
VC_DEVICE void unpack4(const uint2& u, float (&f)[4]) {
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x);
    f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
}
VC_DEVICE uint2 pack4(const float (&f)[4]) {
    uint2 u;
    u.x = pack_bf16x2(f[0], f[1]);
    u.y = pack_bf16x2(f[2], f[3]);
    return u;
}

// ---- 64-lane wave reductions -------------------------------------------------------------
VC_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
VC_DEVICE float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

VC_DEVICE float gelu_tanh_f(float x) {
    // nn.GELU(approximate='tanh'): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (x + k1 * x * x * x);
    // tanh(u) = 1 - 2/(exp(2u)+1) on the hardware exp2 / rcp (1 ulp each; the result is rounded to bf16 by every caller):
    // exp overflow -> inf -> rcp 0 -> tanh = 1, underflow -> 0 -> tanh = -1.  An IEEE division here costs ~10 instructions
    // per element of every FFN-1 output tile.
    const float e = __builtin_amdgcn_exp2f(u * 2.8853900817779268f);
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
    return 0.5f * x * (1.0f + t);
}
VC_DEVICE float silu_f(float x) { return x / (1.0f + __expf(-x)); }
