// Flash-style attention forward for gfx950, head dim 128, non-causal, optional key-length mask.
//
// Replaces videox_fun.models.attention_utils.attention (flash-attn) as called by the reference at
// wan_transformer3d.py:394-399 (self-attention, k_lens = seq_lens) and :425-430 (T5 cross-attention,
// 512 keys, no mask):  out = softmax(q k^T / sqrt(D)) v, keys >= k_len masked.
//
// Structure (CDNA4, 64-lane waves, v_mfma_f32_32x32x16_bf16):
//  * workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows, its Q
//    fragment lives in registers for the whole kernel.
//  * K/V tiles of 64 keys are staged HBM -> registers -> LDS (double buffer, one barrier per tile);
//    the loads of tile t+1 are issued before the math of tile t and written after it.
//  * S^T = K . Q^T ("swapped" product): the 32x32 accumulator has the query row on the LANE and the key
//    index in the registers, so row max / row sum are per-lane loops plus one exchange with lane^32,
//    and the accumulator is already the B operand of the next product  O^T = V^T . P^T  (no LDS trip).
//  * V^T fragments come from the row-major LDS image through ds_read_b64_tr_b16 (hardware transpose).
//  * LDS images: 256-byte rows; K chunks XOR (row&15) (conflict-free ds_read_b128),
//    V chunks XOR (((row&3)<<2)|((row>>2)&3)) (conflict-free transposed reads).
//  * workgroup -> (batch*head, q block) map is XCD-contiguous so the workgroups sharing one XCD's L2 walk
//    the same K/V.
#include <stdlib.h>

#include "vc_common.h"
#include "vc_kernels.h"

namespace {

constexpr int D = 128;
constexpr int QB = 128;     // query rows per workgroup
constexpr int KT = 64;      // keys per tile
constexpr int TILE_BYTES = KT * D * 2;          // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;     // K + V
constexpr int LDS_BYTES = 2 * STAGE_BYTES;      // double buffer = 64 KiB

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

VC_DEVICE int k_off(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }
VC_DEVICE int v_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
VC_DEVICE int v_off(int row, int ch) { return row * 256 + ((ch ^ v_swz(row)) << 4); }

template <bool SEG>
VC_DEVICE int64_t tok_off(int t, int64_t ts, int seg_len, int64_t ss) {
    if (!SEG) return (int64_t)t * ts;
    const int s = t / seg_len;
    return (int64_t)s * ss + (int64_t)(t - s * seg_len) * ts;
}

typedef __attribute__((address_space(3))) char lds_char;

// One LDS-DMA of 16 B per lane (LDS dest = M0 base + lane*16), issued through inline asm so that hipcc does not
// see a pending LDS write: otherwise it drains vmcnt(0) in front of the next ds_read and the prefetch of tile
// t+1 stops overlapping the math of tile t.  The wave waits for it itself (s_waitcnt vmcnt(0) at the loop top).
VC_DEVICE void glds16_asm(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_uniform)
                 : "memory");
}

// K/V tile t -> LDS stage `buf`: one wave-instruction fills 4 rows of 256 B; the chunk swizzle is applied to the
// per-lane SOURCE address (the LDS image of a wave-instruction is lane-linear).
template <bool SEG>
VC_DEVICE void stage_kv(const bf16_t* kp, const bf16_t* vp, const VcAttnParams& p, int t, char* buf, int wave, int lane) {
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)buf;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row0 = __builtin_amdgcn_readfirstlane(wave * 16 + i * 4);
        const int row = row0 + (lane >> 4), pc = lane & 15;
        int key = t * KT + row;
        key = key < p.Lk ? key : p.Lk - 1;
        const bf16_t* ks = kp + tok_off<SEG>(key, p.k_ts, p.seg_len, p.k_ss) + ((pc ^ (row & 15)) << 3);
        const bf16_t* vs = vp + tok_off<SEG>(key, p.v_ts, p.seg_len, p.v_ss) + ((pc ^ v_swz(row)) << 3);
        glds16_asm(ks, __builtin_amdgcn_readfirstlane(lds0 + row0 * 256));
        glds16_asm(vs, __builtin_amdgcn_readfirstlane(lds0 + TILE_BYTES + row0 * 256));
    }
}

struct AttnLaneConst {
    unsigned koff[8];     // K image byte offset of this lane's row r, chunk ks*2+h          (+ kb*8192 + stage)
    unsigned voff[4][2];  // V image byte offset for (db, half): keys +0..3 / +8..11 of step 0 (+ s*4096 + stage)
};

// one KV tile: S^T = K.Q^T, online softmax, O^T += V^T.P^T.  STAGE selects the LDS stage statically so that every
// LDS read is (loop-invariant lane offset register) + (immediate).
template <int STAGE, int VARIANT>
VC_DEVICE void attn_tile(const char* smem, const AttnLaneConst& lc, const bf16x8 (&qf)[8], f32x16 (&O)[4], float& m_run,
                         float& l_run, float c, int t, int k_len, int h) {
    const char* kbuf = smem + STAGE * STAGE_BYTES;
    const char* vbuf = kbuf + TILE_BYTES;
    // ---- all 16 K fragments first, then the MFMA chain ----
    bf16x8 kf[2][8];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) kf[kb][ks] = *(const bf16x8*)(kbuf + lc.koff[ks] + kb * 8192);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 S[2];
    if (VARIANT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) S[kb][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][ks], qf[ks], S[kb], 0, 0, 0);
    }
    if (VARIANT & 1) __builtin_amdgcn_s_setprio(0);
    // V^T fragments of key steps 0,1 fly under the softmax
    bf16x8 vfa[2][4], vfb[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + lc.voff[db][0] + s * 4096));
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + lc.voff[db][1] + s * 4096));
            vfa[s][db] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    __builtin_amdgcn_sched_barrier(0);
    if ((t + 1) * KT > k_len) {   // tile straddles k_len: mask keys >= k_len (block-uniform branch)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = t * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (key >= k_len) S[kb][e] = -1e30f;
            }
    }
    // ---- online softmax (row = lane&31, duplicated on lane^32) ----
    float mx = S[0][0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mx = fmaxf(mx, S[0][e]);
#pragma unroll
    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, S[1][e]);
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (__any(mx > m_run)) {      // some row of this wave has a new maximum: rescale (exact; else alpha == 1)
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[i][e] *= alpha;
    }
    const float mc = m_run * c;
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float pe = __builtin_amdgcn_exp2f(S[kb][e] * c - mc);
            S[kb][e] = pe;
            ps += pe;
        }
    l_run += ps;
    // ---- O^T[d][q] += V^T[d][key] . P^T[key][q] ----
    bf16x8 pf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[s][j] = (__bf16)S[s >> 1][8 * (s & 1) + j];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + lc.voff[db][0] + (s + 2) * 4096));
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + lc.voff[db][1] + (s + 2) * 4096));
            vfb[s][db] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    if (VARIANT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int db = 0; db < 4; ++db) O[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfa[s][db], pf[s], O[db], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int db = 0; db < 4; ++db) O[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfb[s][db], pf[s + 2], O[db], 0, 0, 0);
    if (VARIANT & 1) __builtin_amdgcn_s_setprio(0);
}

template <bool SEG, int VARIANT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(VcAttnParams p, int nQ, int nwork) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int per_xcd = gridDim.x >> 3;
    const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (id >= nwork) return;
    const int bh = id / nQ, qb = id - bh * nQ;
    const int b = bh / p.H, head = bh - b * p.H;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const bf16_t* qp = (const bf16_t*)p.q + (int64_t)b * p.q_bs + (int64_t)head * p.q_hs;
    const bf16_t* kp = (const bf16_t*)p.k + (int64_t)b * p.k_bs + (int64_t)head * p.k_hs;
    const bf16_t* vp = (const bf16_t*)p.v + (int64_t)b * p.v_bs + (int64_t)head * p.v_hs;
    bf16_t* op = (bf16_t*)p.out + (int64_t)b * p.o_bs + (int64_t)head * p.o_hs;

    const int k_len = (p.k_len > 0 && p.k_len < p.Lk) ? p.k_len : p.Lk;
    const int nt = (k_len + KT - 1) / KT;

    stage_kv<SEG>(kp, vp, p, 0, smem, wave, lane);

    // ---- Q fragment: B operand of S^T = K.Q^T : lane holds Q[q = r][d = ks*16 + 8h + 0..7] ----
    const int q_row = qb * QB + wave * 32 + r;
    const int q_row_c = q_row < p.Lq ? q_row : p.Lq - 1;
    bf16x8 qf[8];
    {
        const bf16_t* qrow = qp + tok_off<SEG>(q_row_c, p.q_ts, p.seg_len, p.q_ss) + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qrow + ks * 16);
        // retire these (compiler-counted) loads here: a vmcnt wait left inside the loop would also drain the
        // hand-issued LDS-DMA prefetch of the next tile
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
    }

    // ---- loop-invariant per-lane LDS offsets ----
    AttnLaneConst lc;
    {
        const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;   // transposed-read roles (T10)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) lc.koff[ks] = k_off(r, ks * 2 + h);
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                lc.voff[db][hf] = v_off(4 * (g >> 1) + q4 + 8 * hf, db * 4 + 2 * (g & 1) + (p4 >> 1)) + 8 * (p4 & 1);
    }

    f32x16 O[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[i][e] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float c = p.scale * 1.4426950408889634f;

    for (int t = 0; t < nt; t += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();     // tile t landed for every wave; every wave finished reading stage 1
        if (t + 1 < nt) stage_kv<SEG>(kp, vp, p, t + 1, smem + STAGE_BYTES, wave, lane);
        attn_tile<0, VARIANT>(smem, lc, qf, O, m_run, l_run, c, t, k_len, h);
        if (t + 1 >= nt) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 2 < nt) stage_kv<SEG>(kp, vp, p, t + 2, smem, wave, lane);
        attn_tile<1, VARIANT>(smem, lc, qf, O, m_run, l_run, c, t + 1, k_len, h);
    }

    // ---- epilogue: lane holds O[q = r][d = db*32 + 8*g4 + 4h + 0..3] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < p.Lq) {
        bf16_t* orow = op + tok_off<SEG>(q_row, p.o_ts, p.seg_len, p.o_ss) + 4 * h;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v[4] = {O[db][4 * g4] * inv, O[db][4 * g4 + 1] * inv, O[db][4 * g4 + 2] * inv,
                              O[db][4 * g4 + 3] * inv};
                *(uint2*)(orow + db * 32 + 8 * g4) = pack4(v);
            }
    }
}

template <bool SEG, int VARIANT>
int launch_attn(const VcAttnParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attn_fwd_kernel<SEG, VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES) != hipSuccess)
            return VC_E_HIP;
        attr_set = true;
    }
    const int nQ = (p.Lq + QB - 1) / QB;
    const int nwork = p.B * p.H * nQ;
    const int grid = (nwork + 7) / 8 * 8;
    hipLaunchKernelGGL((attn_fwd_kernel<SEG, VARIANT>), dim3(grid), dim3(256), LDS_BYTES, stream, p, nQ, nwork);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}

}  // namespace

int vc_launch_attention(const VcAttnParams& p, hipStream_t stream) {
    if (!p.q || !p.k || !p.v || !p.out || p.B <= 0 || p.H <= 0 || p.Lq <= 0 || p.Lk <= 0) return VC_E_INVALID;
    if ((p.q_ts | p.k_ts | p.v_ts | p.q_hs | p.k_hs | p.v_hs | p.q_bs | p.k_bs | p.v_bs) % 8) return VC_E_UNSUPPORTED;
    if ((p.o_ts | p.o_hs | p.o_bs) % 4) return VC_E_UNSUPPORTED;
    if (p.seg_len < 0 || (p.seg_len > 0 && ((p.q_ss | p.k_ss | p.v_ss) % 8 || p.o_ss % 4))) return VC_E_UNSUPPORTED;
    static const int variant = getenv("VC_ATTN_VARIANT") ? atoi(getenv("VC_ATTN_VARIANT")) : 0;
    if (p.seg_len > 0) return launch_attn<true, 0>(p, stream);
    return variant == 1 ? launch_attn<false, 1>(p, stream) : launch_attn<false, 0>(p, stream);
}
