// Attention forward for medium key counts (5-8 tiles of 64 keys), head dim 128: T5 cross-attention over a zero-padded prompt whose
// folded key sequence does not fit the LDS-resident kernel (attn_short_kernel, <= 256 keys), i.e. prompts of 257-511 tokens
// (wan_transformer3d.py:425-430 after wan_transformer3d_versecrafter.py:358-363; the padded keys are identical rows and fold into
// one key of multiplicity n: attention.hip, MERGE).
//
// Round 3: this replaces the <MERGE, 4-wave> instantiation of the software-pipelined kernel, which at these sizes was all
// prologue / tail and spilled 893 VGPRs.  Plain structure, nothing to spill (about 190 registers):
//  * workgroup = 4 waves = 128 query rows of one (batch, head); Q fragments in registers;
//  * K / V tiles are double-buffered in LDS (same swizzled images as attention.hip), register-staged: the global loads of tile
//    t+1 are issued before tile t's QK^T chain and written to the other buffer right after it; one barrier per tile;
//  * per tile: S^T = K.Q^T (16 v_mfma_f32_32x32x16_bf16), exact online softmax, O^T += V^T.P^T (16 MFMA);
//  * epilogue: lane pairs (r, r + 32) trade halves so that every lane stores 16 contiguous bytes.
#include <stdlib.h>

#include "vc_common.h"
#include "vc_kernels.h"

namespace {

constexpr int D = 128;
constexpr int KT = 64;
constexpr int TILE_BYTES = KT * D * 2;          // 16 KiB
constexpr int LDS_BYTES = 4 * TILE_BYTES;       // 2 stages of K | V

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // plain vector type: arrays of HIP's uint4 struct stay in scratch here

VC_DEVICE int ks_off(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }
VC_DEVICE int vs_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
VC_DEVICE int vs_off(int row, int ch) { return row * 256 + ((ch ^ vs_swz(row)) << 4); }

template <bool MERGE>
__global__ __launch_bounds__(256, 2) void attn_stream_kernel(VcAttnParams p, int nQ, int nwork) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int per_xcd = gridDim.x >> 3;
    const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (id >= nwork) return;
    const int bh = id / nQ, qb = id - bh * nQ;
    const int b = bh / p.H, head = bh - b * p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    const bf16_t* qp = (const bf16_t*)p.q + (int64_t)b * p.q_bs + (int64_t)head * p.q_hs;
    const bf16_t* kp = (const bf16_t*)p.k + (int64_t)b * p.k_bs + (int64_t)head * p.k_hs;
    const bf16_t* vp = (const bf16_t*)p.v + (int64_t)b * p.v_bs + (int64_t)head * p.v_hs;
    bf16_t* op = (bf16_t*)p.out + (int64_t)b * p.o_bs + (int64_t)head * p.o_hs;

    int k_len = (p.k_len > 0 && p.k_len < p.Lk) ? p.k_len : p.Lk;
    float pad_bias = 0.f;              // added to the raw logit of key k_len - 1 (MERGE)
    if (MERGE) {
        const int from = p.pad_from[b];
        if (from >= 0 && from < p.Lk - 1) {
            k_len = from + 1;
            pad_bias = log2f((float)(p.Lk - from)) / (p.scale * 1.4426950408889634f);
        }
    }
    const int nt = (k_len + KT - 1) / KT;

    // ---- staging: thread -> 4 (row, chunk) pairs per operand and tile: chunk = tid & 15, rows tid / 16 + 16 j ----
    const int s_ch = tid & 15, s_row = tid >> 4;
    u32x4 kreg[4], vreg[4];
    auto load_tile = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int key = t * KT + s_row + 16 * j;
            key = key < p.Lk ? key : p.Lk - 1;
            kreg[j] = *(const u32x4*)(kp + (int64_t)key * p.k_ts + s_ch * 8);
            vreg[j] = *(const u32x4*)(vp + (int64_t)key * p.v_ts + s_ch * 8);
        }
    };
    auto store_tile = [&](int st) __attribute__((always_inline)) {
        char* kimg = smem + st * 2 * TILE_BYTES;
        char* vimg = kimg + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = s_row + 16 * j;
            *(u32x4*)(kimg + ks_off(row, s_ch)) = kreg[j];
            *(u32x4*)(vimg + vs_off(row, s_ch)) = vreg[j];
        }
    };
    load_tile(0);

    // ---- Q fragment, per-lane LDS offsets (as attention.hip) ----
    const int q_row = qb * 128 + wave * 32 + r;
    const int q_row_c = q_row < p.Lq ? q_row : p.Lq - 1;
    bf16x8 qf[8];
    {
        const bf16_t* qrow = qp + (int64_t)q_row_c * p.q_ts + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qrow + ks * 16);
    }
    unsigned koff0, voff[4][2];        // K chunk 2 ks + h of row r: (2 ks + h) ^ (r & 15) = (h ^ (r & 15)) ^ 2 ks -> one register, XOR per use
    {
        const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
        koff0 = ks_off(r, h);
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                voff[db][hf] = vs_off(4 * (g >> 1) + q4 + 8 * hf, db * 4 + 2 * (g & 1) + (p4 >> 1)) + 8 * (p4 & 1);
    }
    store_tile(0);
    __syncthreads();

    f32x16 O[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[i][e] = 0.f;
    const float c = p.scale * 1.4426950408889634f;
    float m_run = -1e30f, l_run = 0.f;

    for (int t = 0; t < nt; ++t) {
        const bool more = t + 1 < nt;
        if (more) load_tile(t + 1);                        // in flight during this tile's arithmetic
        const char* kbuf = smem + (t & 1) * 2 * TILE_BYTES;
        const char* vbuf = kbuf + TILE_BYTES;
        f32x16 S[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) S[kb][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kbuf + (koff0 ^ (unsigned)(ks << 5)) + kb * 8192);
                S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], S[kb], 0, 0, 0);
            }
        }
        // tile t+1 -> the other stage (its last readers passed the barrier that ended iteration t-1): written here, after the QK^T
        // chain has covered the loads' latency and before the softmax needs the registers
        if (more) store_tile((t + 1) & 1);
        if (!more) {                                       // keys >= k_len of the last tile; the folded key's multiplicity
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = t * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (key >= k_len) S[kb][e] = -1e30f;
                    else if (MERGE && key == k_len - 1) S[kb][e] += pad_bias;
                }
        }
        float mx = S[0][0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, S[0][e]);
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, S[1][e]);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        const float m_new = fmaxf(m_run, fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) O[i][e] *= alpha;
            m_run = m_new;
        }
        const float mc = m_run * c;
        bf16x8 pf[4];
        float ps = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pe = __builtin_amdgcn_exp2f(S[s >> 1][8 * (s & 1) + j] * c - mc);
                ps += pe;
                pf[s][j] = (__bf16)pe;
            }
        l_run += ps;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + voff[db][0] + s * 4096));
                const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + voff[db][1] + s * 4096));
                const bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                O[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], O[db], 0, 0, 0);
            }
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    // the output row is recomputed from the lane id here (an opaque read, so that nothing lane-derived is kept -- and spilled --
    // across the tile loop for the epilogue's sake)
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int q_row_e = qb * 128 + wave * 32 + (lane_e & 31);
    bf16_t* orow = op + (int64_t)(q_row_e < p.Lq ? q_row_e : p.Lq - 1) * p.o_ts + 8 * (lane_e >> 5);
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
}

}  // namespace

// plain layout; the caller (vc_launch_attention) has validated strides; 16-byte output stores need o strides % 8 == 0
int vc_launch_attention_stream(const VcAttnParams& p, hipStream_t stream) {
    if (p.seg_len != 0 || (p.o_ts | p.o_hs | p.o_bs) % 8) return VC_E_UNSUPPORTED;
    const int nQ = (p.Lq + 127) / 128;
    const int nwork = p.B * p.H * nQ;
    const int grid = (nwork + 7) / 8 * 8;
    if (p.pad_merge) {
        static std::atomic<uint64_t> done{0};
        if (!vc_set_lds_once(done, (const void*)attn_stream_kernel<true>, LDS_BYTES)) return VC_E_HIP;
        hipLaunchKernelGGL(attn_stream_kernel<true>, dim3(grid), dim3(256), LDS_BYTES, stream, p, nQ, nwork);
    } else {
        static std::atomic<uint64_t> done{0};
        if (!vc_set_lds_once(done, (const void*)attn_stream_kernel<false>, LDS_BYTES)) return VC_E_HIP;
        hipLaunchKernelGGL(attn_stream_kernel<false>, dim3(grid), dim3(256), LDS_BYTES, stream, p, nQ, nwork);
    }
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}
