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

VC_DEVICE int k_off(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }
VC_DEVICE int v_off(int row, int ch) { return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }

VC_DEVICE int64_t tok_off(int t, int64_t ts, int seg_len, int64_t ss) {
    if (seg_len == 0) return (int64_t)t * ts;
    const int s = t / seg_len;
    return (int64_t)s * ss + (int64_t)(t - s * seg_len) * ts;
}

VC_DEVICE void issue_loads(const bf16_t* kp, const bf16_t* vp, const VcAttnParams& p, int t,
                           int st_row, int st_ch, uint4 (&kreg)[4], uint4 (&vreg)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int key = t * KT + st_row + 16 * i;
        key = key < p.Lk ? key : p.Lk - 1;
        kreg[i] = *(const uint4*)(kp + tok_off(key, p.k_ts, p.seg_len, p.k_ss) + st_ch * 8);
        vreg[i] = *(const uint4*)(vp + tok_off(key, p.v_ts, p.seg_len, p.v_ss) + st_ch * 8);
    }
}
VC_DEVICE void write_lds(char* buf, int st_row, int st_ch, const uint4 (&kreg)[4], const uint4 (&vreg)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = st_row + 16 * i;
        *(uint4*)(buf + k_off(row, st_ch)) = kreg[i];
        *(uint4*)(buf + TILE_BYTES + v_off(row, st_ch)) = vreg[i];
    }
}

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

    // ---- Q fragment: B operand of S^T = K.Q^T : lane holds Q[q = r][d = ks*16 + 8h + 0..7] ----
    const int q_row = qb * QB + wave * 32 + r;
    const int q_row_c = q_row < p.Lq ? q_row : p.Lq - 1;
    bf16x8 qf[8];
    {
        const bf16_t* qrow = qp + tok_off(q_row_c, p.q_ts, p.seg_len, p.q_ss) + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qrow + ks * 16);
    }

    // ---- staging map: thread -> (row, 16-byte chunk) x 4 ----
    const int st_row = tid >> 4, st_ch = tid & 15;   // rows st_row + 16*i
    uint4 kreg[4], vreg[4];
    f32x16 O[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[i][e] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float c = p.scale * 1.4426950408889634f;

    issue_loads(kp, vp, p, 0, st_row, st_ch, kreg, vreg);
    write_lds(smem, st_row, st_ch, kreg, vreg);
    __syncthreads();

    // transposed-read lane constants: group g = lane>>4 -> (h = g>>1, d half = g&1); in group: q4 = row, p4 = col quad
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;

    for (int t = 0; t < nt; ++t) {
        char* buf = smem + (t & 1) * STAGE_BYTES;
        // next tile's loads fly under this tile's math (after the last tile: a harmless re-load)
        issue_loads(kp, vp, p, (t + 1 < nt) ? t + 1 : t, st_row, st_ch, kreg, vreg);

        // ---- S^T[key][q] for the two 32-key blocks ----
        f32x16 S[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) S[kb][e] = 0.f;
            const int krow = kb * 32 + r;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(buf + k_off(krow, ks * 2 + h));
                S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], S[kb], 0, 0, 0);
            }
        }
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
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        const float mc = m_new * c;
        m_run = m_new;
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = __builtin_amdgcn_exp2f(S[kb][e] * c - mc);
                S[kb][e] = pe;
                ps += pe;
            }
        l_run = l_run * alpha + ps;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[i][e] *= alpha;

        // ---- O^T[d][q] += V^T[d][key] . P^T[key][q] ----
        const char* vbuf = buf + TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kb = s >> 1, s2 = s & 1;
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (__bf16)S[kb][8 * s2 + j];
            const int key0 = s * 16 + 4 * (g >> 1) + q4;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const int ch = db * 4 + 2 * (g & 1) + (p4 >> 1);
                const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (lds_bf16x4*)(vbuf + v_off(key0, ch) + 8 * (p4 & 1)));
                const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (lds_bf16x4*)(vbuf + v_off(key0 + 8, ch) + 8 * (p4 & 1)));
                const bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                O[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, O[db], 0, 0, 0);
            }
        }

        write_lds(smem + ((t + 1) & 1) * STAGE_BYTES, st_row, st_ch, kreg, vreg);
        __syncthreads();
    }

    // ---- epilogue: lane holds O[q = r][d = db*32 + 8*g4 + 4h + 0..3] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < p.Lq) {
        bf16_t* orow = op + tok_off(q_row, p.o_ts, p.seg_len, p.o_ss) + 4 * h;
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

}  // namespace

int vc_launch_attention(const VcAttnParams& p, hipStream_t stream) {
    if (!p.q || !p.k || !p.v || !p.out || p.B <= 0 || p.H <= 0 || p.Lq <= 0 || p.Lk <= 0) return VC_E_INVALID;
    if ((p.q_ts | p.k_ts | p.v_ts | p.q_hs | p.k_hs | p.v_hs | p.q_bs | p.k_bs | p.v_bs) % 8) return VC_E_UNSUPPORTED;
    if ((p.o_ts | p.o_hs | p.o_bs) % 4) return VC_E_UNSUPPORTED;
    if (p.seg_len < 0 || (p.seg_len > 0 && ((p.q_ss | p.k_ss | p.v_ss) % 8 || p.o_ss % 4))) return VC_E_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES) != hipSuccess)
            return VC_E_HIP;
        attr_set = true;
    }
    const int nQ = (p.Lq + QB - 1) / QB;
    const int nwork = p.B * p.H * nQ;
    const int grid = (nwork + 7) / 8 * 8;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(grid), dim3(256), LDS_BYTES, stream, p, nQ, nwork);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}
