// fp8 self-attention for gfx950 (round 4; BASELINE config 5 names "fp8 MFMA" -- an opt-in mode of this build, the reference computes
// attention in bf16 through flash-attn, wan_transformer3d.py:394-399):  out = softmax(q k^T / sqrt(D)) v, head dim 128, non-causal,
// keys >= k_len masked, with Q, K, V AND the softmax weights P held as OCP e4m3 under MX-style block scales (one E8M0 power of two per
// 32 elements along the contraction), both products on v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulation.
//
// Two kernels:
//  1. attn_quant_fp8_kernel: one pass over the bf16 q / k / v of a (batch, head, 64-token tile) that writes
//       Q8  [B][H][tile][64][128]  e4m3, natural d order, pre-multiplied by scale * log2(e) * 8 / 65535 (the softmax constant and the
//                                  byte mapping of P below are folded into Q before it is quantised); Qs [..][64][4] scale bytes
//       KV  [B][H][nTk + 3][19456] one RECORD per key tile j = what the attention kernel copies into LDS in one beat, as it lies in LDS:
//         bytes 0 .. 9215      the K image of tile j: 64 rows of 128 B at a pitch of 144 B (9 sixteen-byte slots: the 16 lanes of a
//                              ds_read_b128 group then fall on 16 different slots of the 256-byte bank row with NO xor swizzle, so a
//                              lane's two chunks stay adjacent and land in one 8-register tuple without moves); the first 4 of the 16 pad
//                              bytes of row l hold the four block-scale bytes MFMA lane l supplies;
//         bytes 9216 .. 19455  V of tile j - 1 TRANSPOSED (K runs one tile ahead of V in the kernel): 128 rows (d) of 64 B at a pitch of
//                              80 B, the 64 keys of the tile in the order the P fragment holds them (below); pad bytes 0..3 of rows
//                              0..63: the scale bytes of lane l.
//       Record 0 has no V part, records nTk .. nTk+2 no K part (the copy runs three records ahead and is not clamped): never computed on.
//     so the attention kernel's LDS-DMA is ONE linear copy per beat and no transposed LDS read is needed.
//     Scale blocks follow the MFMA: the instruction's k-block b (32 of its 64 k) is bytes 16b .. 16b+15 of BOTH lane halves, and the
//     scale of (row, block b) is supplied by lane 32 b + row (probed: tools/micro/fp8_attn_probe.hip).  A lane reads 32 contiguous bytes
//     (2s+h)*32.. of its q / k row for k-step s, so block (s, b) of a q / k row is d in {64s+16b .. +15} u {64s+32+16b .. +15}; a block
//     of V^T is one d column over the 32 keys kb*32 .. kb*32+31.  A block's scale is the smallest power of two 2^e with amax <= 448 2^e
//     (no saturation: v_cvt_pk_fp8_f32 turns values >= 480 into NaN); elements are rounded to nearest even.
//  2. attn_fp8_kernel: the structure of attn_fwd_pipe_kernel (8 waves x 32 query rows, 64-key tiles, S^T = K Q^T with the query on the
//     lane, O^T += V^T P^T, software pipeline MFMA(S(t+1)) || VALU(P(t)), MFMA(PV(t)) || VALU(max(t+1)), deferred rescale) with
//       * 4 + 4 MFMAs per tile instead of 16 + 16, half the LDS bytes, whole records through a 4-deep LDS ring filled two tiles
//         ahead by LDS-DMA (issued by waves 0-3 only: five instructions each, no branches) behind a counted vmcnt(5) and a raw s_barrier;
//       * P per (row, tile) block-scaled: e = ceil(log2 of the tile's largest weight relative to the row's reference), P 2^(8-e) in
//         e4m3 (largest byte in (112, 120]), the block scale 2^(e-8) goes into the MFMA's scale operand -- a tile far below the running
//         maximum keeps full relative precision instead of flushing to zero;
//       * the row sum comes from a ninth MFMA against an all-ones A operand: l is the sum of exactly the P values the PV product saw;
//       * PMODE 1 ("log-domain", default): P's byte is produced WITHOUT an exponential: byte = round(8 log2(P 2^(8-e)) + 56) is an e4m3
//         bit pattern whose value is 2^floor(.) (1 + frac / 8) -- the piecewise-linear 2^x, at most 6.1 % above the exponential and
//         identical for numerator and denominator; with the constants folded into Q it is one v_add_f32 per element plus 2
//         v_cvt_pknorm_u16_f32 + 1 v_perm_b32 per four (the fp8 loop is VALU-bound: tools/micro/fp8_attn_probe*.hip);
//         PMODE 0 ("exact"): v_exp_f32 and v_cvt_pk_fp8_f32.
// Restated on the CPU in oracle/attn_fp8_oracle.py (the definition the tests pin; parity with the reference is "within the fp8 error
// of exact attention", bounds in tests/test_gpu_attention_fp8.py).
#include <stdlib.h>

#include <type_traits>

#include "vc_common.h"
#include "vc_kernels.h"

#ifdef F8_TRACE     // tools/trace_attn_fp8.py: per-wave sums of the core clock spent in each section of a beat
__device__ uint64_t* vc_f8_trace_buf = nullptr;
extern "C" int vc_debug_set_attn_fp8_trace(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(vc_f8_trace_buf), &buf, sizeof buf) == hipSuccess ? 0 : -1;
}
#endif

namespace {

constexpr int KT = 64;
constexpr int KPITCH = 144, VPITCH = 80;          // row pitch of the K / V^T tile images (128 / 64 data bytes + 16 of padding)
constexpr int KTILE = KT * KPITCH, VTILE = 128 * VPITCH;      // bytes of a K image (9 KiB) and of a V^T image (10 KiB)
constexpr int REC = KTILE + VTILE;                // one record: K image of tile j | V^T image of tile j - 1 (19 KiB)
constexpr int REC_PAD = 3;                        // records past the last tile that the kernel's copy may touch
constexpr int NST = 4;                            // LDS ring depth (records)
constexpr int F8_ONES = NST * REC;                // 32 bytes of e4m3 1.0: the A operand of the row-sum MFMA, read like a fragment (broadcast)
constexpr int F8_LDS = F8_ONES + 64;              // 76 KiB
constexpr int WPIECE = REC / 4;                   // bytes of a record one of the four copying waves moves: 4 x 1024 + 768
typedef i32x8 __attribute__((aligned(16))) i32x8_a16;
constexpr float K1 = 65535.0f / 8.0f;             // raw accumulator units -> log2 units
constexpr float DEFER_T = 8.0f;                   // a row's reference follows its maximum once it is 2^8 behind (as the bf16 kernel)

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;

// smallest e with amax <= 448 * 2^e, as the E8M0 byte e + 127 clamped to [0, 254]  (448 = 1.75 * 2^8)
VC_DEVICE int mx_scale_byte(float amax) {
    const unsigned u = __float_as_uint(amax);
    const int E = (int)((u >> 23) & 0xFFu);
    int sb = E - 8 + ((u & 0x7FFFFFu) > 0x600000u ? 1 : 0);
    sb = sb < 0 ? 0 : sb;
    return sb > 254 ? 254 : sb;
}
VC_DEVICE float mx_inv_scale(int sb) { return __uint_as_float((unsigned)(254 - sb) << 23); }     // 2^(127 - sb)

// 16 floats (already multiplied by the block's inverse scale) -> 16 e4m3 bytes, element j in byte j
VC_DEVICE uint4 cvt16_fp8(const float (&x)[16]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i], x[4 * i + 1], 0, false);
        r = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i + 2], x[4 * i + 3], r, true);
        w[i] = (unsigned)r;
    }
    return uint4{w[0], w[1], w[2], w[3]};
}

// ---------------------------------------------------------------------------------------------------------------------------------
// quantiser: one workgroup of 256 threads per (batch, head, 64-token tile)
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_quant_fp8_kernel(VcAttnFp8Params p, int nTq, int nTk, int ntile) {
    const int tile = blockIdx.x % ntile, bh = blockIdx.x / ntile;
    const int b = bh / p.H, head = bh - b * p.H;
    const int tid = threadIdx.x;
    const bf16_t* qp = (const bf16_t*)p.q + (int64_t)b * p.q_bs + (int64_t)head * p.q_hs;
    const bf16_t* kp = (const bf16_t*)p.k + (int64_t)b * p.k_bs + (int64_t)head * p.k_hs;
    const bf16_t* vp = (const bf16_t*)p.v + (int64_t)b * p.v_bs + (int64_t)head * p.v_hs;
    char* ws = (char*)p.ws;
    char* rec0 = ws + p.off_kv + (int64_t)bh * (nTk + REC_PAD) * REC;
    __shared__ unsigned ksb[64], vsb[64];               // the scale dwords of the 64 MFMA lanes (bytes come from four threads each)
    const int k_len = (p.k_len > 0 && p.k_len < p.Lk) ? p.k_len : p.Lk;      // keys past it are quantised as zeros: they never reach a sum
    // ---- q and k rows: thread = (token r, block (s, bb)) ----
    {
        const int r = tid >> 2, blk = tid & 3, s = blk >> 1, bb = blk & 1;
        const int d0 = 64 * s + 16 * bb, d1 = d0 + 32;
        const int tok = tile * KT + r;
#pragma unroll
        for (int which = 0; which < 2; ++which) {           // 0: q, 1: k
            const bool isq = which == 0;
            const int L = isq ? p.Lq : k_len, nT = isq ? nTq : nTk;
            if (tile >= nT) continue;
            float x[32];
            if (tok < L) {
                const bf16_t* row = (isq ? qp + (int64_t)tok * p.q_ts : kp + (int64_t)tok * p.k_ts);
                const uint4 a0 = *(const uint4*)(row + d0), a1 = *(const uint4*)(row + d0 + 8);
                const uint4 c0 = *(const uint4*)(row + d1), c1 = *(const uint4*)(row + d1 + 8);
                float t8[8];
                unpack8(a0, t8);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = t8[i];
                unpack8(a1, t8);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[8 + i] = t8[i];
                unpack8(c0, t8);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[16 + i] = t8[i];
                unpack8(c1, t8);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[24 + i] = t8[i];
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) x[i] = 0.f;
            }
            if (isq) {
#pragma unroll
                for (int i = 0; i < 32; ++i) x[i] *= p.qfold;             // softmax constant and byte mapping folded into Q
            }
            float amax = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(x[i]));
            const int sb = mx_scale_byte(amax);
            const float inv = mx_inv_scale(sb);
            float lo[16], hi[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { lo[i] = x[i] * inv; hi[i] = x[16 + i] * inv; }
            const uint4 qlo = cvt16_fp8(lo), qhi = cvt16_fp8(hi);
            if (isq) {
                char* q8 = ws + p.off_q8 + ((int64_t)bh * nTq + tile) * (KT * 128) + r * 128;
                *(uint4*)(q8 + d0) = qlo;
                *(uint4*)(q8 + d1) = qhi;
                (ws + p.off_qs + ((int64_t)bh * nTq + tile) * (KT * 4))[r * 4 + 2 * s + bb] = (char)sb;
            } else {
                char* k8 = rec0 + (int64_t)tile * REC + r * KPITCH;
                *(uint4*)(k8 + d0) = qlo;
                *(uint4*)(k8 + d1) = qhi;
                ((char*)ksb)[(bb * 32 + (r & 31)) * 4 + (r >> 5) * 2 + s] = (char)sb;      // lane 32 bb + (r & 31): byte kb * 2 + s
            }
        }
    }
    // ---- v: thread = (column d, key block kb) ----
    if (tile < nTk) {
        const int d = tid & 127, kb = tid >> 7;
        float x[32];
        float amax = 0.f;
        // the tile of V goes through LDS: 16 bytes per lane from global memory (a key row's 128 channels are contiguous), columns read back
        // per thread; rows whose strides or base are not 16-byte aligned take the 2-byte loads directly
        __shared__ __attribute__((aligned(16))) unsigned short vt[KT][136];
        const bool wide_v = ((p.v_ts | p.v_hs | p.v_bs) & 7) == 0 && ((uintptr_t)p.v & 15) == 0;
        if (wide_v) {
            const int kr = tid >> 2, qd = tid & 3, key = tile * KT + kr;
            const bf16_t* src = vp + (int64_t)key * p.v_ts + qd * 32;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *(uint4*)&vt[kr][qd * 32 + 8 * i] = key < k_len ? *(const uint4*)(src + 8 * i) : uint4{0u, 0u, 0u, 0u};
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                x[i] = __uint_as_float((unsigned)vt[kb * 32 + i][d] << 16);
                amax = fmaxf(amax, fabsf(x[i]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int key = tile * KT + kb * 32 + i;
                x[i] = key < k_len ? (float)vp[(int64_t)key * p.v_ts + d] : 0.f;
                amax = fmaxf(amax, fabsf(x[i]));
            }
        }
        const int sb = mx_scale_byte(amax);
        const float inv = mx_inv_scale(sb);
        char* v8 = rec0 + (int64_t)(tile + 1) * REC + KTILE + d * VPITCH;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float y[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) y[j] = x[(j & 3) + 8 * (j >> 2) + 4 * hh] * inv;     // the key order of the P fragment
            *(uint4*)(v8 + 32 * hh + 16 * kb) = cvt16_fp8(y);
        }
        if (kb == 0 && d >= 64) *(uint4*)(v8 + 64) = uint4{0u, 0u, 0u, 0u};     // the pads travel with the image: keep them defined
        ((char*)vsb)[(kb * 32 + (d & 31)) * 4 + (d >> 5)] = (char)sb;            // lane 32 kb + (d & 31): byte db
    }
    __syncthreads();
    if (tile < nTk && tid < 128) {                       // the pad of image row l: lane l's scale dword, then zeros
        const int l = tid & 63;
        if (tid < 64) *(uint4*)(rec0 + (int64_t)tile * REC + l * KPITCH + 128) = uint4{ksb[l], 0u, 0u, 0u};
        else *(uint4*)(rec0 + (int64_t)(tile + 1) * REC + KTILE + l * VPITCH + 64) = uint4{vsb[l], 0u, 0u, 0u};
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// attention
// ---------------------------------------------------------------------------------------------------------------------------------
VC_DEVICE void f8_glds16(unsigned voff, const void* sbase, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst_uniform)
                 : "memory");
}
VC_DEVICE void f8_glds4(unsigned voff, const void* sbase, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst_uniform)
                 : "memory");
}

// The scaled MFMA is issued by inline asm with the accumulator TIED (through the builtin hipcc, ROCm 7.2, leaves vdst != srcC for part
// of them: the accumulators migrate between tuples and ~260 VGPRs spill into the loop -- the round-3 finding of the fp8 GEMM again).
// What hipcc then no longer does for these instructions is done by hand:
//   * operands that a VALU instruction may just have written (the P fragment, its scale): `s_nop 1` in front of the first MFMA of the group;
//   * an accumulator is read by VALU code only behind F8_MFMA_SETTLE (20 wait states for a 16-pass MFMA) or behind a later group of
//     MFMAs of the same wave (the matrix pipe is in order);
//   * ds_read results feeding an asm operand are waited for by the compiler's own lgkmcnt pass.
// OPS: op_sel / op_sel_hi select the scale BYTE of the two scale registers (x = a's byte, y = b's byte: op_sel:[x&1,y&1,0] op_sel_hi:[x>>1,y>>1,0]).
#define F8_OPS_00 "op_sel_hi:[0,0,0]"
#define F8_OPS_10 "op_sel:[1,0,0] op_sel_hi:[0,0,0]"
#define F8_OPS_20 "op_sel:[0,0,0] op_sel_hi:[1,0,0]"
#define F8_OPS_30 "op_sel:[1,0,0] op_sel_hi:[1,0,0]"
#define F8_OPS_12 "op_sel:[1,0,0] op_sel_hi:[0,1,0]"
#define F8_OPS_32 "op_sel:[1,0,0] op_sel_hi:[1,1,0]"
#ifndef F8_POST
#define F8_POST ""
#endif
#ifndef F8_ABLATE
#define F8_ABLATE 0         // timing-only builds (WRONG results): 1 no DMA wait / barrier, 2 no LDS-DMA issue, 3 both, 4 no MFMA, 5 no conversion of P
#endif                      // (tools/build_variant.sh + tools/time_attn_fp8_core.py; what they can and cannot show: profiles/r04_attn_fp8_ablate.txt)
#ifndef F8_ASM_GAPS
#define F8_ASM_GAPS 0       // pmode 1: a gap's fourteen conversion instructions as ONE asm statement (0: C++ with an opaque copy of the constant per gap)
#endif
#ifndef F8_PRIO
#define F8_PRIO 1           // s_setprio 1 around: 1 the QK^T phase, 2 the PV phase (A/B builds)
#endif
#ifndef F8_STAGGER
#define F8_STAGGER 1        // waves 4-7 run the two phases of a beat in the other order (0: all eight waves in step)
#endif
#ifndef F8_PRE
#define F8_PRE "s_nop 1\n\t"
#endif
#if F8_ABLATE == 4
#define F8_MFMA(acc, a, b, sa, sb, OPS) asm volatile("" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define F8_MFMA_NOP(acc, a, b, sa, sb, OPS) asm volatile("" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define F8_MFMA_ZERO(acc, a, b, sa, sb, OPS) asm volatile("" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define F8_MFMA_ZERO_NOP(acc, a, b, sa, sb, OPS) asm volatile("" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#else
#define F8_MFMA(acc, a, b, sa, sb, OPS) \
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 " OPS F8_POST : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define F8_MFMA_NOP(acc, a, b, sa, sb, OPS) \
    asm volatile(F8_PRE "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 " OPS F8_POST : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define F8_MFMA_ZERO(acc, a, b, sa, sb, OPS) \
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 " OPS F8_POST : "=&v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define F8_MFMA_ZERO_NOP(acc, a, b, sa, sb, OPS) \
    asm volatile(F8_PRE "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 " OPS F8_POST : "=&v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#endif
#define F8_MFMA_SETTLE4(a0, a1, a2, a3) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3))

template <int PMODE>
__global__ __launch_bounds__(512, 2) void attn_fp8_kernel(VcAttnFp8Params p, int nQ, int nwork, int nTq, int nTk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int per_xcd = gridDim.x >> 3;
    const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (id >= nwork) return;
    const int bh = id / nQ, qb = id - bh * nQ;
    const int b = bh / p.H, head = bh - b * p.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    const char* ws = (const char*)p.ws;
    const char* q8 = ws + p.off_q8 + (int64_t)bh * nTq * (KT * 128);
    const char* qs = ws + p.off_qs + (int64_t)bh * nTq * (KT * 4);
    const char* rec0 = ws + p.off_kv + (int64_t)bh * (nTk + REC_PAD) * REC;
    bf16_t* op = (bf16_t*)p.out + (int64_t)b * p.o_bs + (int64_t)head * p.o_hs;

    const int k_len = (p.k_len > 0 && p.k_len < p.Lk) ? p.k_len : p.Lk;
    const int nt = (k_len + KT - 1) / KT;
    const bool tail_partial = (k_len & (KT - 1)) != 0;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
    const unsigned lane16 = (unsigned)(lane * 16);     // the wave's 1-KiB piece is selected through the scalar base

    // One wave-instruction of LDS-DMA moves 1 KiB; a record is 19 of them.  Waves 0-3 copy a quarter of the record each -- four whole
    // instructions and one of 48 lanes under one M0 value, no branch -- and waves 4-7 none: the staggered half is the critical
    // one (tools/trace_attn_fp8.py: with every wave issuing 2-5 pieces behind per-wave conditions the issue alone was 350-460 of a beat's
    // 2530 clocks, and waves 0-3 then idled 500 at the barrier).  Beat t (right after its barrier) requests record t+3 = K(t+3) | V(t+2)
    // into ring slot (t+3) & 3; at the top of the next beat vmcnt(5) leaves only that request in flight on the copying waves: a record has
    // two beats to land.  The workspace holds three records past the last tile, so nothing is clamped.
    // (the instruction offset of global_load_lds is added to the global AND to the LDS address: one M0 value serves the five pieces)
    const unsigned lds_w = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * WPIECE + 1024u);
    const char* src_w = rec0 + (int64_t)wave * WPIECE + 1024;          // + 1024: the five instruction offsets are -1024 .. 3072
    auto copy_record = [&](int j) {
        const char* src = src_w + (int64_t)j * REC;
        const unsigned dst = lds_w + (unsigned)(j & 3) * REC;
        unsigned keep;
        uint64_t ex;
        asm volatile(
            "s_mov_b32 %[keep], m0\n\t"
            "s_mov_b32 m0, %[dst]\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[v], %[src] offset:-1024\n\t"
            "global_load_lds_dwordx4 %[v], %[src]\n\t"
            "global_load_lds_dwordx4 %[v], %[src] offset:1024\n\t"
            "global_load_lds_dwordx4 %[v], %[src] offset:2048\n\t"
            "s_mov_b64 %[ex], exec\n\ts_mov_b32 exec_hi, 0xffff\n\t"            // lanes 0-47 (exec is all ones here: wave-uniform code)
            "global_load_lds_dwordx4 %[v], %[src] offset:3072\n\t"
            "s_mov_b64 exec, %[ex]\n\t"
            "s_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep), [ex] "=&s"(ex)
            : [dst] "s"(dst), [v] "v"(lane16), [src] "s"(src)
            : "memory", "scc");
    };
    // lane * 4 ... are derived from lane * 16 where they are needed, by an instruction hipcc cannot hoist out of the tile loop: as loop
    // invariants they (and the LDS addresses built on them) ended up spilled, and a spill reload inside the loop waits vmcnt(0) -- draining
    // the LDS-DMA ring
    auto ksc_off = [&]() { unsigned a; asm volatile("v_mul_u32_u24 %0, 9, %1" : "=v"(a) : "v"(lane16)); return a + 128u; };   // lane * 144 + 128
    auto vsc_off = [&]() { unsigned a; asm volatile("v_mul_u32_u24 %0, 5, %1" : "=v"(a) : "v"(lane16)); return a + 64u + (unsigned)KTILE; };
    auto issue = [&](int t) { if (wave < 4) copy_record(t + 3); };
    // ---- prologue loads: records 0 and 1 (K(0), K(1), V(0)), drained once; then what "beat -1" would have requested ----
    if (wave < 4) {
        copy_record(0);
        copy_record(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        copy_record(2);
    }

    // ---- Q fragments and scales (registers for the whole kernel) ----
    const int q_row = qb * 256 + wave * 32 + r;
    const int q_row_c = q_row < p.Lq ? q_row : p.Lq - 1;
    i32x8 qf[2];
    int qsc;
    {
        const char* qrow = q8 + (int64_t)q_row_c * 128 + 32 * h;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const i32x4 lo = *(const i32x4*)(qrow + 64 * s), hi = *(const i32x4*)(qrow + 64 * s + 16);
            qf[s] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        qsc = (int)((*(const unsigned*)(qs + (int64_t)q_row_c * 4)) >> (8 * h));      // byte 0: block (0, h), byte 2: block (1, h)
    }
    // fragment addresses: K row kb*32 + r, bytes (2s+h)*32 .. +31; V^T row db*32 + r, bytes 32h .. +31 (pitches 144 / 80: no swizzle)
    const unsigned koff = (unsigned)(r * KPITCH + 32 * h), voff = (unsigned)(r * VPITCH + 32 * h);
    if (tid < 8) *(int*)(smem + F8_ONES + tid * 4) = 0x38383838;  // e4m3 1.0 (visible after the prologue's barrier)
    if (tid == 8) *(int*)(smem + F8_ONES + 32) = 0x7F7F7F7F;      // and the unit E8M0 scale that goes with it

    f32x16 O[4];
#pragma unroll
    for (int e = 0; e < 16; ++e) { O[0][e] = 0.f; O[1][e] = 0.f; O[2][e] = 0.f; O[3][e] = 0.f; }
    float l_run = 0.f;
    float m_run = -1e30f, m_new = -1e30f, m_tile = -1e30f;       // raw accumulator units (x K1 = log2 units)

    auto qk = [&](int t, f32x16 (&S)[2]) {                        // S(t) = K(t) Q^T : 4 MFMAs
        const char* kbuf = smem + (t & 3) * REC;
        const int ksc = *(const int*)(kbuf + ksc_off());
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            i32x8 kf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) kf[s] = *(const i32x8_a16*)(kbuf + kb * (32 * KPITCH) + koff + 64 * s);
            if (kb == 0) {
                F8_MFMA_ZERO(S[0], kf[0], qf[0], ksc, qsc, F8_OPS_00);
                F8_MFMA(S[0], kf[1], qf[1], ksc, qsc, F8_OPS_12);
            } else {
                F8_MFMA_ZERO(S[1], kf[0], qf[0], ksc, qsc, F8_OPS_20);
                F8_MFMA(S[1], kf[1], qf[1], ksc, qsc, F8_OPS_32);
            }
        }
    };
    auto mask_tail = [&](f32x16 (&S)[2], int t) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = t * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (key >= k_len) S[kb][e] = -1e30f;
            }
    };
    auto row_max = [&](const f32x16 (&S)[2]) -> float {
        float mx = S[0][0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, S[0][e]);
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, S[1][e]);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };

    // ---- prologue: S(0) and its row maximum ----
    f32x16 Sa[2], Sb[2];
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                                 // the copying waves' pieces of records 0, 1 (K(0), K(1), V(0)) landed before their vmcnt(0) above
    __builtin_amdgcn_sched_barrier(0);
    qk(0, Sa);
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(Sa[0]), "+v"(Sa[1]));
    if (nt == 1 && tail_partial) mask_tail(Sa, 0);
    m_tile = row_max(Sa);
    m_new = m_tile;

    // A tile's work is two phases (below).  ONE instance of each per S-buffer role, run-time flags for "a next tile exists" and "it is the
    // masked last one" (wave-uniform branches): the bf16 kernel's specialised tail instances are not an option here -- hipcc spills
    // accumulators around them, and a spill store that follows an inline-asm MFMA reads its result before the matrix pipe has written it
    // (the compiler does not know the statement is an MFMA); found as wrong outputs for short key sequences; tests/test_build_resources.py
    // requires 0 scratch.
    //   phase 1 (t): [deferred rescale] ; exponent of P(t) ; MFMA S(t+1) = K(t+1) Q^T (4)  ||  VALU: the e4m3 bytes of P(t)
    //   phase 2 (t): MFMA row sum, O += V(t)^T P(t)^T (1 + 4)                                 ||  VALU: row maximum of S(t+1)
    // The nine MFMAs are inline asm, which hipcc keeps in order but does not interleave by itself, so the VALU / LDS work is placed BETWEEN
    // them by hand: a quarter of P(t)'s conversion (14 VALU) and the fragment reads of the MFMA after next behind each QK^T MFMA, a quarter
    // of the row maximum behind each PV MFMA.  An accumulator is read only after a LATER group of MFMAs of this wave has been issued.
    // Waves 4-7 run the two phases of a beat (= the stretch between two workgroup barriers) in the OTHER order -- phase 2 of the previous
    // tile, then phase 1 of this one -- so that on every SIMD one wave is in its VALU-heavy QK^T phase (and its serial head: barrier, DMA
    // issue, rescale test) while its partner feeds the matrix pipe from the VALU-light PV phase (MI355X_MICROARCH.md "Two waves per SIMD",
    // item 9; same per-row operation order, so both halves give bit-identical rows).
    // Fragment reads sit one or two MFMA gaps ahead of the MFMA that takes them; the reads of a phase's FIRST MFMA are issued before the
    // phase: K(t+1)'s first half right behind the beat's barrier (in front of the DMA issue and the rescale arithmetic, or -- staggered half
    // -- of the whole phase 2), V(t)'s first fragment and scales in the last gap of phase 1.  (Reading a WHOLE phase ahead -- all K
    // fragments behind the barrier, all V^T fragments in the QK^T gaps, 243 VGPRs -- was measured and is slower: 23.2 ms against 21.1 ms
    // at the bench shape, profiles/r04_attn_fp8_ablate.txt.)
    i32x8 pf, k00, k01, vf0, vf1;
    int pscale = 0, ksc = 0, vsc = 0;
    const i32x8 ones = *(const i32x8_a16*)(smem + F8_ONES);        // every lane the same 32 bytes (written before the prologue's barrier)
    const int unit = *(const int*)(smem + F8_ONES + 32);
#define F8_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef F8_TRACE
    uint64_t trc_acc[5] = {0, 0, 0, 0, 0}, trc_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(trc_last) :: "memory");
    const uint64_t trc_t0 = trc_last;
    uint64_t trc_r0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(trc_r0) :: "memory");        // 100 MHz: the in-kernel clock is d(memtime) / d(memrealtime) x 100 MHz
    auto mark = [&](int i) {
        uint64_t now;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        trc_acc[i] += now - trc_last;
        trc_last = now;
    };
#ifdef F8_CLOCK_ONLY       // stamps around the whole loop only: the clock the UNPERTURBED loop holds
#define F8_MARK(i) do {} while (0)
#else
#define F8_MARK(i) mark(i)
#endif
#else
#define F8_MARK(i) do {} while (0)
#endif
    auto beat_head = [&](int t) {
        F8_FENCE();
        F8_MARK(4);
        if (F8_ABLATE != 1 && F8_ABLATE != 3) {
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // (copying waves) all but the youngest record: record t+1 = K(t+1) | V(t) is in LDS
            __builtin_amdgcn_s_barrier();                         // ... for every wave; and every wave is past its reads of the slots refilled now
        }
        F8_FENCE();
        F8_MARK(0);
        if (t + 1 < nt) {
            const char* kbuf = smem + ((t + 1) & 3) * REC;
            ksc = *(const int*)(kbuf + ksc_off());
            k00 = *(const i32x8_a16*)(kbuf + koff);
            k01 = *(const i32x8_a16*)(kbuf + koff + 64);
        }
        F8_FENCE();
        if (F8_ABLATE != 2 && F8_ABLATE != 3) issue(t);
        F8_MARK(1);
    };
    auto phase1 = [&](int t, f32x16 (&Sc)[2], f32x16 (&Sn)[2]) {
        const bool MORE = t + 1 < nt;
        // deferred rescale (as attn_fwd_pipe_kernel): the row's reference follows its running maximum only after 2^DEFER_T
        {
            const bool moved = (m_new - m_run) * K1 > DEFER_T;
            if (__any(moved)) {
                F8_MFMA_SETTLE4(O[0], O[1], O[2], O[3]);          // the PV MFMAs of the previous tile may still be in the pipe
                const float m_ref = moved ? m_new : m_run;
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_ref) * K1);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) O[i][e] *= alpha;
                m_run = m_ref;
            }
        }
        // block scale of P(t): 2^(et - 8), et = ceil(log2 of the tile's largest weight against the reference) (<= DEFER_T)
        const float et = fmaxf(__builtin_ceilf((m_tile - m_run) * K1), -100.f);
        pscale = 119 + (int)et;
        const float kc = PMODE == 1 ? (120.f - 8.f * et) * (1.0f / 65535.0f) - m_run : 8.f - et - m_run * K1;
        // one dword of P(t): four weights.  `k` is the gap's own copy of the constant (an opaque copy made BEHIND the MFMA that opens the
        // gap), and the gap ends in a statement that consumes the dwords it produced: the conversion can neither rise above the gap's
        // MFMA nor sink below the next one (sched_barrier alone does not hold it: the adds are emitted before the first fence otherwise)
        auto pconv = [&](int kb, int i, float k) -> int {
            if (PMODE == 1) {
                // byte = round(8 (log2 units of s - reference - et + 8) + 56) = round(65535 (s - m_run) + 120 - 8 et): v_cvt_pknorm_u16_f32
                // computes round(65535 clamp(x, 0, 1)); the result is < 256, the low bytes of the two halves are gathered by v_perm_b32
                const u16x2 u01 = __builtin_amdgcn_cvt_pknorm_u16(Sc[kb][4 * i] + k, Sc[kb][4 * i + 1] + k);
                const u16x2 u23 = __builtin_amdgcn_cvt_pknorm_u16(Sc[kb][4 * i + 2] + k, Sc[kb][4 * i + 3] + k);
                return (int)__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, u23), __builtin_bit_cast(unsigned, u01), 0x06040200u);
            } else {
                float pe[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) pe[j] = __builtin_amdgcn_exp2f(Sc[kb][4 * i + j] * K1 + k);
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(pe[0], pe[1], 0, false);
                return __builtin_amdgcn_cvt_pk_fp8_f32(pe[2], pe[3], w, true);
            }
        };
        auto pgap = [&](int kb, int i0) {                         // dwords i0, i0 + 1 of half kb
            if (F8_ABLATE == 5) { pf[kb * 4 + i0] = __float_as_int(Sc[kb][4 * i0]); pf[kb * 4 + i0 + 1] = __float_as_int(Sc[kb][4 * i0 + 4]); return; }
            if (PMODE == 1 && F8_ASM_GAPS) {
                // the same fourteen instructions as one asm statement: ordered against the MFMA statements by being volatile, and no
                // per-gap copy of the constant (the opaque-copy form below costs a v_mov and an s_nop per gap)
                int w0, w1;
                float t0, t1, t2, t3, t4, t5, t6, t7;
                asm volatile(
                    "v_add_f32 %[t0], %[s0], %[k]\n\tv_add_f32 %[t1], %[s1], %[k]\n\tv_add_f32 %[t2], %[s2], %[k]\n\tv_add_f32 %[t3], %[s3], %[k]\n\t"
                    "v_add_f32 %[t4], %[s4], %[k]\n\tv_add_f32 %[t5], %[s5], %[k]\n\tv_add_f32 %[t6], %[s6], %[k]\n\tv_add_f32 %[t7], %[s7], %[k]\n\t"
                    "v_cvt_pknorm_u16_f32 %[t0], %[t0], %[t1]\n\tv_cvt_pknorm_u16_f32 %[t2], %[t2], %[t3]\n\t"
                    "v_cvt_pknorm_u16_f32 %[t4], %[t4], %[t5]\n\tv_cvt_pknorm_u16_f32 %[t6], %[t6], %[t7]\n\t"
                    "v_perm_b32 %[w0], %[t2], %[t0], %[sel]\n\tv_perm_b32 %[w1], %[t6], %[t4], %[sel]"
                    : [w0] "=&v"(w0), [w1] "=&v"(w1), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4),
                      [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7)
                    : [s0] "v"(Sc[kb][4 * i0]), [s1] "v"(Sc[kb][4 * i0 + 1]), [s2] "v"(Sc[kb][4 * i0 + 2]), [s3] "v"(Sc[kb][4 * i0 + 3]),
                      [s4] "v"(Sc[kb][4 * i0 + 4]), [s5] "v"(Sc[kb][4 * i0 + 5]), [s6] "v"(Sc[kb][4 * i0 + 6]), [s7] "v"(Sc[kb][4 * i0 + 7]),
                      [k] "v"(kc), [sel] "s"(0x06040200));
                pf[kb * 4 + i0] = w0;
                pf[kb * 4 + i0 + 1] = w1;
                return;
            }
            float k = kc;
            asm volatile("" : "+v"(k));
            const int w0 = pconv(kb, i0, k), w1 = pconv(kb, i0 + 1, k);
            asm volatile("" :: "v"(w0), "v"(w1));
            pf[kb * 4 + i0] = w0;
            pf[kb * 4 + i0 + 1] = w1;
        };
        if (MORE) {
            const char* kbuf = smem + ((t + 1) & 3) * REC;
            F8_FENCE();
            if (F8_PRIO == 1) __builtin_amdgcn_s_setprio(1);
            F8_MFMA_ZERO(Sn[0], k00, qf[0], ksc, qsc, F8_OPS_00);
            const i32x8 k10 = *(const i32x8_a16*)(kbuf + 32 * KPITCH + koff);
            pgap(0, 0);
            F8_FENCE();
            F8_MFMA(Sn[0], k01, qf[1], ksc, qsc, F8_OPS_12);
            const i32x8 k11 = *(const i32x8_a16*)(kbuf + 32 * KPITCH + koff + 64);
            pgap(0, 2);
            F8_FENCE();
            F8_MFMA_ZERO(Sn[1], k10, qf[0], ksc, qsc, F8_OPS_20);
            vsc = *(const int*)(smem + ((t + 1) & 3) * REC + vsc_off());          // V(t) travels in record t + 1
            vf0 = *(const i32x8_a16*)(smem + ((t + 1) & 3) * REC + KTILE + voff);
            pgap(1, 0);
            F8_FENCE();
            F8_MFMA(Sn[1], k11, qf[1], ksc, qsc, F8_OPS_32);
            vf1 = *(const i32x8_a16*)(smem + ((t + 1) & 3) * REC + KTILE + 32 * VPITCH + voff);
            pgap(1, 2);
            if (F8_PRIO == 1) __builtin_amdgcn_s_setprio(0);
            F8_FENCE();
        } else {
            vsc = *(const int*)(smem + ((t + 1) & 3) * REC + vsc_off());
            vf0 = *(const i32x8_a16*)(smem + ((t + 1) & 3) * REC + KTILE + voff);
            vf1 = *(const i32x8_a16*)(smem + ((t + 1) & 3) * REC + KTILE + 32 * VPITCH + voff);
            pgap(0, 0); pgap(0, 2); pgap(1, 0); pgap(1, 2);
            F8_FENCE();
        }
    };
    auto phase2 = [&](int t, f32x16 (&Sc)[2], f32x16 (&Sn)[2]) {
        const bool MORE = t + 1 < nt;
        const bool MASK = MORE && tail_partial && (t + 2 == nt);
        const char* vbuf = smem + ((t + 1) & 3) * REC + KTILE;     // V(t) travels in record t + 1
        i32x8 vf[4];
        vf[0] = vf0;
        vf[1] = vf1;
        vf[2] = *(const i32x8_a16*)(vbuf + 64 * VPITCH + voff);
        // PV gaps: a quarter of S(t+1)'s row maximum each, unconditionally (no branch, no copies at merge points: on the last tile and on the
        // masked one -- which takes its maximum again behind the mask -- the result is simply not used)
        float mxa = -1e30f;
        // the tile's row sum FIRST, into the registers of S(t) (dead: P(t) has been packed): 1^T P^T has 32 equal rows, element 0 is
        // added to the running sum once the four MFMAs behind it have been issued
        F8_FENCE();
        if (F8_PRIO == 2) __builtin_amdgcn_s_setprio(1);
        F8_MFMA_ZERO_NOP(Sc[0], ones, pf, unit, pscale, F8_OPS_00);
        vf[3] = *(const i32x8_a16*)(vbuf + 96 * VPITCH + voff);
        asm volatile("" : "+v"(Sn[0]));                           // written at least two MFMAs before the one just issued
#pragma unroll
        for (int e = 0; e < 8; ++e) mxa = fmaxf(mxa, Sn[0][e]);
        asm volatile("" : "+v"(mxa));
        F8_FENCE();
        F8_MFMA(O[0], vf[0], pf, vsc, pscale, F8_OPS_00);
#pragma unroll
        for (int e = 8; e < 16; ++e) mxa = fmaxf(mxa, Sn[0][e]);
        asm volatile("" : "+v"(mxa));
        F8_FENCE();
        F8_MFMA(O[1], vf[1], pf, vsc, pscale, F8_OPS_10);
        asm volatile("" : "+v"(Sn[1]));                           // at least three MFMAs behind the last QK^T one
#pragma unroll
        for (int e = 0; e < 8; ++e) mxa = fmaxf(mxa, Sn[1][e]);
        asm volatile("" : "+v"(mxa));
        F8_FENCE();
        F8_MFMA(O[2], vf[2], pf, vsc, pscale, F8_OPS_20);
#pragma unroll
        for (int e = 8; e < 16; ++e) mxa = fmaxf(mxa, Sn[1][e]);
        asm volatile("" : "+v"(mxa));
        F8_FENCE();
        F8_MFMA(O[3], vf[3], pf, vsc, pscale, F8_OPS_30);
        if (F8_PRIO == 2) __builtin_amdgcn_s_setprio(0);
        F8_FENCE();
        asm volatile("" : "+v"(Sc[0]));
        l_run += Sc[0][0];
        if (MORE) {
            if (MASK) {
                mask_tail(Sn, t + 1);
                m_tile = row_max(Sn);
            } else {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mxa), __float_as_uint(mxa), false, false);
                m_tile = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            m_new = fmaxf(m_new, m_tile);
        }
    };
#undef F8_FENCE
    // one loop for both halves of the workgroup (wave-uniform branches on `stag`); beat t = [barrier t .. barrier t+1):
    //   waves 0-3: phase1(t) phase2(t)         waves 4-7: phase2(t-1) phase1(t), and phase2(nt-1) behind the last beat
    const bool stag = F8_STAGGER != 0 && wave >= 4;
    for (int t = 0; t <= nt; t += 2) {
        if (t < nt) beat_head(t);
        if (stag && t > 0) { phase2(t - 1, Sb, Sa); F8_MARK(3); }
        if (t < nt) {
            phase1(t, Sa, Sb);
            F8_MARK(2);
            if (!stag) { phase2(t, Sa, Sb); F8_MARK(3); }
        }
        if (t + 1 <= nt) {
            if (t + 1 < nt) beat_head(t + 1);
            if (stag && t < nt) { phase2(t, Sa, Sb); F8_MARK(3); }
            if (t + 1 < nt) {
                phase1(t + 1, Sb, Sa);
                F8_MARK(2);
                if (!stag) { phase2(t + 1, Sb, Sa); F8_MARK(3); }
            }
        }
    }
#ifdef F8_TRACE
    if (vc_f8_trace_buf && lane == 0) {
        uint64_t* o = vc_f8_trace_buf + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = trc_acc[i];
        uint64_t trc_c1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(trc_c1) :: "memory");
        o[5] = trc_c1 - trc_t0;
        o[6] = (uint64_t)nt;
        uint64_t trc_r1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(trc_r1) :: "memory");
        o[7] = trc_r1 - trc_r0;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // nothing may still be landing in LDS when the workgroup retires
    F8_MFMA_SETTLE4(O[0], O[1], O[2], O[3]);

    // Epilogue.  The lane's row and half are derived AGAIN here, from an instruction hipcc cannot move above the loop: kept alive across it
    // they were the values it chose to spill (the kernel must have no scratch at all: tests/test_build_resources.py).
    unsigned lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int r_e = (int)(lane_e & 31u), h_e = (int)(lane_e >> 5);
    const int q_row_e = qb * 256 + wave * 32 + r_e;
    const int q_row_ce = q_row_e < p.Lq ? q_row_e : p.Lq - 1;
    const float inv = 1.0f / l_run;
    const bool wide = (((p.o_ts | p.o_hs | p.o_bs) & 7) == 0) && (((uintptr_t)p.out & 15) == 0);
    if (wide) {
        bf16_t* orow = op + (int64_t)q_row_ce * p.o_ts + 8 * h_e;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float va[4], vb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { va[e] = O[db][8 * j + e] * inv; vb[e] = O[db][8 * j + 4 + e] * inv; }
                const uint2 pa = pack4(va), pb = pack4(vb);
                const auto sx = __builtin_amdgcn_permlane32_swap(pa.x, pb.x, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(pa.y, pb.y, false, false);
                if (q_row_e < p.Lq) *(uint4*)(orow + db * 32 + 16 * j) = uint4{sx[0], sy[0], sx[1], sy[1]};
            }
    } else if (q_row_e < p.Lq) {
        bf16_t* orow = op + (int64_t)q_row_e * p.o_ts + 4 * h_e;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v[4] = {O[db][4 * g4] * inv, O[db][4 * g4 + 1] * inv, O[db][4 * g4 + 2] * inv, O[db][4 * g4 + 3] * inv};
                *(uint2*)(orow + db * 32 + 8 * g4) = pack4(v);
            }
    }
}

inline int64_t up256(int64_t v) { return (v + 255) / 256 * 256; }

}  // namespace

int64_t vc_attention_fp8_workspace_bytes(int B, int H, int Lq, int Lk) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
    const int64_t nTq = (Lq + KT - 1) / KT, nTk = (Lk + KT - 1) / KT, bh = (int64_t)B * H;
    return up256(bh * nTq * KT * 128) + up256(bh * nTq * KT * 4) + up256(bh * (nTk + REC_PAD) * REC);
}

static int fp8_attn_fill(VcAttnFp8Params& p, int64_t ws_bytes) {
    if (!p.q || !p.k || !p.v || !p.out || !p.ws || p.B <= 0 || p.H <= 0 || p.Lq <= 0 || p.Lk <= 0) return VC_E_INVALID;
    if ((p.q_ts | p.k_ts | p.q_hs | p.k_hs | p.q_bs | p.k_bs) % 8) return VC_E_UNSUPPORTED;      // 16-byte row loads of q and k
    if ((p.o_ts | p.o_hs | p.o_bs) % 4 || ((uintptr_t)p.ws & 255)) return VC_E_UNSUPPORTED;
    if (p.pmode != 0 && p.pmode != 1) return VC_E_INVALID;
    if (ws_bytes < vc_attention_fp8_workspace_bytes(p.B, p.H, p.Lq, p.Lk)) return VC_E_NOMEM;
    const int64_t nTq = (p.Lq + KT - 1) / KT, nTk = (p.Lk + KT - 1) / KT, bh = (int64_t)p.B * p.H;
    int64_t off = 0;
    p.off_q8 = off; off += up256(bh * nTq * KT * 128);
    p.off_qs = off; off += up256(bh * nTq * KT * 4);
    p.off_kv = off;
    p.qfold = (float)((double)p.scale * 1.4426950408889634 * 8.0 / 65535.0);
    return VC_OK;
}

int vc_launch_attention_fp8_quant(VcAttnFp8Params p, int64_t ws_bytes, hipStream_t stream) {
    const int rc = fp8_attn_fill(p, ws_bytes);
    if (rc != VC_OK) return rc;
    const int nTq = (p.Lq + KT - 1) / KT, nTk = (p.Lk + KT - 1) / KT, ntile = nTq > nTk ? nTq : nTk;
    hipLaunchKernelGGL(attn_quant_fp8_kernel, dim3((unsigned)((int64_t)p.B * p.H * ntile)), dim3(256), 0, stream, p, nTq, nTk, ntile);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}

int vc_launch_attention_fp8_core(VcAttnFp8Params p, int64_t ws_bytes, hipStream_t stream) {
    const int rc = fp8_attn_fill(p, ws_bytes);
    if (rc != VC_OK) return rc;
    const int nTq = (p.Lq + KT - 1) / KT, nTk = (p.Lk + KT - 1) / KT;
    const int nQ = (p.Lq + 255) / 256;
    const int nwork = p.B * p.H * nQ;
    const int grid = (nwork + 7) / 8 * 8;
    static std::atomic<uint64_t> done0{0}, done1{0};
    if (p.pmode == 1) {
        if (!vc_set_lds_once(done1, (const void*)attn_fp8_kernel<1>, F8_LDS)) return VC_E_HIP;
        hipLaunchKernelGGL(attn_fp8_kernel<1>, dim3(grid), dim3(512), F8_LDS, stream, p, nQ, nwork, nTq, nTk);
    } else {
        if (!vc_set_lds_once(done0, (const void*)attn_fp8_kernel<0>, F8_LDS)) return VC_E_HIP;
        hipLaunchKernelGGL(attn_fp8_kernel<0>, dim3(grid), dim3(512), F8_LDS, stream, p, nQ, nwork, nTq, nTk);
    }
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}

int vc_launch_attention_fp8(const VcAttnFp8Params& p, int64_t ws_bytes, hipStream_t stream) {
    const int rc = vc_launch_attention_fp8_quant(p, ws_bytes, stream);
    if (rc != VC_OK) return rc;
    return vc_launch_attention_fp8_core(p, ws_bytes, stream);
}
