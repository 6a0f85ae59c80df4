// Self-attention forward with v_mfma_f32_16x16x32_bf16 -- the MFMA-shape A/B of the pipelined kernel in attention.hip
// (same workgroup, same 32 query rows per wave, same 64-key tiles, same LDS bytes per FLOP; cdna_hip_programming.md rule 28 /
// MI355X_MICROARCH.md "DVFS give-back (7)": where the chip holds its clock down under an MFMA-dense loop the clock it holds can
// depend on the MFMA shape, so both shapes are built at the same output tile per wave and the faster one by wall time is kept).
//
// Replaces videox_fun.models.attention_utils.attention as called at wan_transformer3d.py:394-399 (self-attention, plain
// [B][L][heads][128] layout, key-length mask).  Selected through VcAttnParams::variant (tests / tools) or by the launcher's default.
//
// What changes against the 32x32x16 form:
//  * a wave's 32 query rows are two blocks of 16; S^T = K.Q^T is built from 16-key x 16-query blocks: A = K block (16 keys x 32 d,
//    one ds_read_b128 per lane, shared by both query blocks), B = Q (registers).  Accumulator S[qb][kb]: lane (i = lane & 15,
//    g = lane >> 4) holds query i of block qb and keys 16 kb + 4 g + e, e = 0..3.
//  * a query's 64 logits of a tile sit on the 4 lanes i, i+16, i+32, i+48: row maxima of both query blocks are reduced together
//    with two v_permlane16_swap and one v_permlane32_swap; row sums stay per lane until the epilogue.
//  * O^T += V^T.P^T per 32-key step s: the contraction order is free as long as both operands agree, so position 8 g + j of step
//    s is key 32 s + 4 g + j (j < 4) or 32 s + 16 + 4 g + j - 4 (j >= 4) -- exactly the keys lane group g holds in S[.][2s] and
//    S[.][2s+1]: P is packed in place (no lane movement), and the V^T fragment is two ds_read_b64_tr_b16 of 4-key x 16-d blocks.
//  * V image swizzle (row & 7) << 1 on the 16-byte chunk index: a 32-lane half of a transposed read covers 8 consecutive keys x
//    16 d = all 64 banks once; K image as in attention.hip (chunk ^ (row & 15)): conflict-free for the 16-lane groups of ds_read_b128.
//  * epilogue: lanes g, g^1 hold neighbouring 8-byte pieces of a row; one v_permlane16_swap per dword pairs them into 16-byte stores.
#include <stdlib.h>

#include <type_traits>

#include "vc_common.h"
#include "vc_kernels.h"

namespace {

#ifndef VC_ATTN_DEFER_MAX
#define VC_ATTN_DEFER_MAX 8
#endif
#ifndef VC_ATTN16_QK_FENCE
#define VC_ATTN16_QK_FENCE 1
#endif
#ifndef VC_ATTN16_XOR_ADDR
#define VC_ATTN16_XOR_ADDR 1
#endif

constexpr int D = 128;
constexpr int KT = 64;
constexpr int TILE_BYTES = KT * D * 2;          // 16 KiB
constexpr int LDS_BYTES = 4 * TILE_BYTES;       // Kst[2] | Vst[2]
constexpr int P_KST = 0, P_VST = 2 * TILE_BYTES;

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) char lds_char;

VC_DEVICE int k16_off(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }
VC_DEVICE int v16_swz(int row) { return (row & 7) << 1; }
VC_DEVICE int v16_off(int row, int ch) { return row * 256 + ((ch ^ v16_swz(row)) << 4); }

VC_DEVICE void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst_uniform)
                 : "memory");
}

// max (or sum) of a and b over the four lanes i, i+16, i+32, i+48 of every i, both results in every lane
template <bool MAX>
VC_DEVICE void reduce4_pair(float& a, float& b) {
    auto op = [](float x, float y) { return MAX ? fmaxf(x, y) : x + y; };
    const auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    const float t = op(__uint_as_float(s1[0]), __uint_as_float(s1[1]));       // rows: a(0,1) b(0,1) a(2,3) b(2,3)
    const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    const float u = op(__uint_as_float(s2[0]), __uint_as_float(s2[1]));       // rows: A B A B
    const auto s3 = __builtin_amdgcn_permlane16_swap(__float_as_uint(u), __float_as_uint(u), false, false);
    a = __uint_as_float(s3[0]);
    b = __uint_as_float(s3[1]);
}

template <int NW, bool LSE = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_pipe16_kernel(VcAttnParams p, int nQ, int nwork) {
    constexpr int QB = NW * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int per_xcd = gridDim.x >> 3;
    const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (id >= nwork) return;
    const int bh = id / nQ, qblk = id - bh * nQ;
    const int b = bh / p.H, head = bh - b * p.H;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;

    const bf16_t* qp = (const bf16_t*)p.q + (int64_t)b * p.q_bs + (int64_t)head * p.q_hs;
    const bf16_t* kp = (const bf16_t*)p.k + (int64_t)b * p.k_bs + (int64_t)head * p.k_hs;
    const bf16_t* vp = (const bf16_t*)p.v + (int64_t)b * p.v_bs + (int64_t)head * p.v_hs;
    bf16_t* op = (bf16_t*)p.out + (int64_t)b * p.o_bs + (int64_t)head * p.o_hs;

    const int k_len = (p.k_len > 0 && p.k_len < p.Lk) ? p.k_len : p.Lk;
    const int nt = (k_len + KT - 1) / KT;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
    constexpr int RPW = KT / NW;
    const int pc = lane & 15;

    // staging as in attention.hip: a wave's pieces (4 tile rows = 1 KiB per wave-instruction) are 16 rows apart, so one per-lane
    // byte offset per operand serves all of them; tile and piece go through the wave-uniform 64-bit base
    const unsigned k_ts2 = (unsigned)(p.k_ts * 2), v_ts2 = (unsigned)(p.v_ts * 2);
    const int qbase = __builtin_amdgcn_readfirstlane(NW == 8 ? (wave & 3) + 8 * (wave >> 2) : wave);
    const int lrow = 4 * qbase + (lane >> 4);
    const unsigned klane = (unsigned)lrow * k_ts2 + ((pc ^ (lrow & 15)) << 4);
    const unsigned vlane = (unsigned)lrow * v_ts2 + ((pc ^ v16_swz(lrow)) << 4);
    unsigned kposA = 0u, vposA = 0u;                      // byte offset of the next tile's first key
    auto stage_one = [&](int t, bool is_k, int st, unsigned& posA) {
        const unsigned ts2 = is_k ? k_ts2 : v_ts2;
        const char* base = (const char*)(is_k ? kp : vp);
        const unsigned dst = lds0 + (is_k ? P_KST : P_VST) + st * TILE_BYTES + qbase * 1024;
        if ((t + 1) * KT <= p.Lk) {
#pragma unroll
            for (int j = 0; j < RPW / 4; ++j)
                glds16_s(is_k ? klane : vlane, base + (posA + (unsigned)(16 * j) * ts2), __builtin_amdgcn_readfirstlane(dst + j * 4096));
        } else {                                           // last, partial tile: rows clamped to Lk - 1
#pragma unroll
            for (int j = 0; j < RPW / 4; ++j) {
                const int row = lrow + 16 * j;
                int key = t * KT + row;
                key = key < p.Lk ? key : p.Lk - 1;
                const unsigned sw = is_k ? (unsigned)(row & 15) : (unsigned)v16_swz(row);
                glds16_s((unsigned)key * ts2 + ((pc ^ sw) << 4), base, __builtin_amdgcn_readfirstlane(dst + j * 4096));
            }
        }
        posA += (unsigned)KT * ts2;
    };
    auto stage = [&](int t, bool do_k, bool do_v, int kst, int vst) {
        if (do_k) stage_one(t, true, kst, kposA);
        if (do_v) stage_one(t, false, vst, vposA);
    };

    stage(0, true, true, 0, 0);
    if (nt > 1) stage(1, true, false, 1, 0);

    // ---- Q fragments: B operand, lane (i, g) holds query 16 qb + i, d = 32 ks + 8 g .. + 7 ----
    const int q_row0 = qblk * QB + wave * 32 + li;          // query block qb: + 16 qb
    bf16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int rc = q_row0 + 16 * qb < p.Lq ? q_row0 + 16 * qb : p.Lq - 1;
        const bf16_t* qrow = qp + (int64_t)rc * p.q_ts + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const bf16x8*)(qrow + ks * 32);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qb][ks]));

    unsigned koff[4], voff[8];
    {
        const int q4 = li >> 2, p4 = li & 3;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) koff[ks] = k16_off(li, 4 * ks + g);                       // + kb * 4096 + stage
#pragma unroll
        for (int db = 0; db < 8; ++db) voff[db] = v16_off(4 * g + q4, 2 * db + (p4 >> 1)) + 8 * (p4 & 1);   // + s * 8192 + half * 4096 + stage
#if VC_ATTN16_XOR_ADDR      // voff[db] = voff[0] ^ (db << 5): one register and a v_xor per use instead of eight registers
        asm volatile("" : "+v"(voff[0]), "+v"(koff[0]));
#endif
    }
    auto koff_of = [&](int ks) -> unsigned {
#if VC_ATTN16_XOR_ADDR
        return ks ? koff[0] ^ (unsigned)(ks << 6) : koff[0];      // (4 ks + g) ^ li = (g ^ li) ^ 4 ks: ks sits in bits 6-7 of the byte offset
#else
        return koff[ks];
#endif
    };
    auto voff_of = [&](int db) -> unsigned {
#if VC_ATTN16_XOR_ADDR
        return db ? voff[0] ^ (unsigned)(db << 5) : voff[0];
#else
        return voff[db];
#endif
    };

    f32x4 O[2][8];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int db = 0; db < 8; ++db)
#pragma unroll
            for (int e = 0; e < 4; ++e) O[qb][db][e] = 0.f;
    const float c = p.scale * 1.4426950408889634f;
    float m_run[2] = {-1e30f, -1e30f}, m_new[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

    struct STile { f32x4 s[2][4]; };          // [query block][key block]

    // loop order follows the address registers (one per ks / per db, everything else is an immediate): hipcc groups the reads of
    // one base register anyway, and with the other order it kept 8 extra fragments live (spills and vmcnt(0) drains in the loop)
    auto qk = [&](const char* kbuf, STile& S) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int e = 0; e < 4; ++e) S.s[qb][kb][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned ko = koff_of(ks);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const bf16x8 kf = *(const bf16x8*)(kbuf + ko + kb * 4096);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    S.s[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][ks], S.s[qb][kb], 0, 0, 0);
            }
#if VC_ATTN16_QK_FENCE
            __builtin_amdgcn_sched_barrier(0x0406);       // K fragments at most one k-step ahead of their MFMAs
#endif
        }
    };
    auto mask_tail = [&](STile& S, int t) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = t * KT + 16 * kb + 4 * g + e;
                if (key >= k_len) { S.s[0][kb][e] = -1e30f; S.s[1][kb][e] = -1e30f; }
            }
    };
    auto row_max = [&](const STile& S, float& mx0, float& mx1) {
        float a = S.s[0][0][0], bq = S.s[1][0][0];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (kb + e > 0) { a = fmaxf(a, S.s[0][kb][e]); bq = fmaxf(bq, S.s[1][kb][e]); }
            }
        reduce4_pair<true>(a, bq);
        mx0 = a; mx1 = bq;
    };

    STile Sa, Sb;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    qk(smem + P_KST, Sa);
    if (nt == 1) mask_tail(Sa, 0);
    {
        float a, bq;
        row_max(Sa, a, bq);
        m_new[0] = fmaxf(m_new[0], a);
        m_new[1] = fmaxf(m_new[1], bq);
    }

    auto body = [&](int t, STile& Sc, STile& Sn, auto par_tag, auto more_tag, auto mask_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool MORE = decltype(more_tag)::value, MASK = decltype(mask_tag)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // K(t+1), V(t) landed; all waves are past QK(t) [Kst[PAR]] and PV(t-1) [Vst[PAR^1]]
        if (t + 2 < nt) stage(t + 2, true, false, PAR, 0);
        if (MORE) stage(t + 1, false, true, 0, PAR ^ 1);
        {   // deferred rescale, per row (see attention.hip); m_run / m_new are equal on the 4 lanes of a query
            bool moved[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
                moved[qb] = VC_ATTN_DEFER_MAX > 0 ? (m_new[qb] - m_run[qb]) * c > (float)VC_ATTN_DEFER_MAX : m_new[qb] > m_run[qb];
            if (__any(moved[0] || moved[1])) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const float m_ref = moved[qb] ? m_new[qb] : m_run[qb];
                    const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_ref) * c);
                    l_run[qb] *= alpha;
#pragma unroll
                    for (int db = 0; db < 8; ++db)
#pragma unroll
                        for (int e = 0; e < 4; ++e) O[qb][db][e] *= alpha;
                    m_run[qb] = m_ref;
                }
            }
        }
        const float mc0 = m_run[0] * c, mc1 = m_run[1] * c;
        // ---- phase 1: MFMA S(t+1) = K(t+1).Q^T  ||  VALU P(t) = exp2(S(t) c - m c), row sums, bf16 pack ----
        if (MORE) qk(smem + P_KST + (PAR ^ 1) * TILE_BYTES, Sn);
        bf16x8 pf[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float mc = qb ? mc1 : mc0;
            float ps = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pe = __builtin_amdgcn_exp2f(Sc.s[qb][2 * s + (j >> 2)][j & 3] * c - mc);
                    ps += pe;
                    pf[qb][s][j] = (__bf16)pe;
                }
            l_run[qb] += ps;
        }
        // ---- phase 2: MFMA O += V(t)^T.P(t)^T  ||  VALU row maxima of S(t+1) ----
        const char* vbuf = smem + P_VST + PAR * TILE_BYTES;
        // V^T fragments one d-block ahead of their MFMAs, and no further: left alone hipcc issues most of the phase's 32 transposed
        // reads up front (44 registers of fragments in flight) and spills a Q fragment around the loop, whose reload drains vmcnt
        auto read_v = [&](int db, bf16x8 (&vf)[2]) {
            const unsigned vo = voff_of(db);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + vo + s * 8192));
                const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + vo + s * 8192 + 4096));
                vf[s] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        bf16x8 vfa[2], vfb[2];
        read_v(0, vfa);
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            bf16x8 (&cur)[2] = (db & 1) ? vfb : vfa;
            bf16x8 (&nxt)[2] = (db & 1) ? vfa : vfb;
            if (db + 1 < 8) read_v(db + 1, nxt);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    O[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[s], pf[qb][s], O[qb][db], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x0406);       // VALU / SALU / transcendentals may cross; MFMA and LDS reads stay in order
        }
        if (MORE) {
            if (MASK) mask_tail(Sn, t + 1);
            float a, bq;
            row_max(Sn, a, bq);
            m_new[0] = fmaxf(m_new[0], a);
            m_new[1] = fmaxf(m_new[1], bq);
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 3 < nt; t += 2) {
        body(t, Sa, Sb, P0{}, T_{}, F_{});
        body(t + 1, Sb, Sa, P1{}, T_{}, F_{});
    }
    if (nt - t == 3) {
        body(t, Sa, Sb, P0{}, T_{}, F_{});
        body(t + 1, Sb, Sa, P1{}, T_{}, T_{});
        body(t + 2, Sa, Sb, P0{}, F_{}, F_{});
    } else if (nt - t == 2) {
        body(t, Sa, Sb, P0{}, T_{}, T_{});
        body(t + 1, Sb, Sa, P1{}, F_{}, F_{});
    } else {
        body(t, Sa, Sb, P0{}, F_{}, F_{});
    }

    reduce4_pair<false>(l_run[0], l_run[1]);
    if (LSE && g == 0) {        // P = exp2((s - m_run) c) and l = sum P, so sum_k exp2(s c) = l 2^(m_run c); one lane of a query's four writes it
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
            if (q_row0 + 16 * qb < p.Lq)
                p.lse[((int64_t)b * p.H + head) * p.Lq + q_row0 + 16 * qb] = log2f(l_run[qb]) + m_run[qb] * c;
    }
    // Epilogue: lane (i, g) holds d = 16 db + 4 g .. + 3 of query i; lanes g, g^1 hold the neighbouring 8 bytes of the same row.
    // One v_permlane16_swap per dword over a (db, db+1) pair leaves every lane with 16 contiguous bytes of one row:
    // g even -> block db, g odd -> block db + 1, at d = 16 (db + (g & 1)) + 8 (g >> 1).
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float inv = 1.0f / l_run[qb];
        bf16_t* orow = op + (int64_t)(q_row0 + 16 * qb) * p.o_ts + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
        for (int db = 0; db < 8; db += 2) {
            float va[4], vb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = O[qb][db][e] * inv; vb[e] = O[qb][db + 1][e] * inv; }
            const uint2 pa = pack4(va), pb = pack4(vb);
            const auto sx = __builtin_amdgcn_permlane16_swap(pa.x, pb.x, false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(pa.y, pb.y, false, false);
            if (q_row0 + 16 * qb < p.Lq) *(uint4*)(orow + 16 * db) = uint4{sx[0], sy[0], sx[1], sy[1]};
        }
    }
}

template <int NW, bool LSE>
int launch_attn_pipe16(const VcAttnParams& p, hipStream_t stream) {
    constexpr int QB = NW * 32;
    static std::atomic<uint64_t> attr_done{0};
    if (!vc_set_lds_once(attr_done, (const void*)attn_fwd_pipe16_kernel<NW, LSE>, LDS_BYTES)) return VC_E_HIP;
    const int nQ = (p.Lq + QB - 1) / QB;
    const int nwork = p.B * p.H * nQ;
    const int grid = (nwork + 7) / 8 * 8;
    hipLaunchKernelGGL((attn_fwd_pipe16_kernel<NW, LSE>), dim3(grid), dim3(NW * 64), LDS_BYTES, stream, p, nQ, nwork);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}

// ring attention: the outputs of R key blocks, each normalised by its own sum, re-weighted by their share of the total
__global__ __launch_bounds__(256) void attn_merge_kernel(VcAttnMergeParams p) {
    const int64_t rows = (int64_t)p.B * p.Lq * p.H;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;         // one thread per 8 output elements
    if (i >= rows * 16) return;
    const int64_t row = i >> 4;
    const int ch = (int)(i & 15);
    const int hh = (int)(row % p.H);
    const int64_t bq = row / p.H;
    const int q = (int)(bq % p.Lq), b = (int)(bq / p.Lq);
    const int64_t li = ((int64_t)b * p.H + hh) * p.Lq + q;
    float l[8], mx = -INFINITY;
    for (int r = 0; r < p.R; ++r) { l[r] = p.lse[r][li]; mx = fmaxf(mx, l[r]); }
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tot = 0.f;
    for (int r = 0; r < p.R; ++r) {
        const float w = l[r] == -INFINITY ? 0.f : exp2f(l[r] - mx);
        if (w == 0.f) continue;
        tot += w;
        float f[8];
        unpack8(*(const uint4*)((const bf16_t*)p.part[r] + row * 128 + ch * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += w * f[e];
    }
    const float inv = tot > 0.f ? 1.0f / tot : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= inv;
    *(uint4*)((bf16_t*)p.out + (int64_t)b * p.o_bs + (int64_t)q * p.o_ts + (int64_t)hh * p.o_hs + ch * 8) = pack8(acc);
}

}  // namespace

// plain layout only (seg_len == 0, no padded-key folding); the caller (vc_launch_attention) has validated strides and spans
int vc_launch_attention_mfma16(const VcAttnParams& p, hipStream_t stream) {
    if (p.seg_len != 0 || p.pad_merge) return VC_E_UNSUPPORTED;
    if ((p.o_ts | p.o_hs | p.o_bs) % 8) return VC_E_UNSUPPORTED;           // 16-byte output stores
    if (p.lse) return p.Lk >= 2048 ? launch_attn_pipe16<8, true>(p, stream) : launch_attn_pipe16<4, true>(p, stream);
    return p.Lk >= 2048 ? launch_attn_pipe16<8, false>(p, stream) : launch_attn_pipe16<4, false>(p, stream);
}

int vc_launch_attention_merge(const VcAttnMergeParams& p, hipStream_t stream) {
    if (p.R < 1 || p.R > 8 || !p.out || p.B <= 0 || p.H <= 0 || p.Lq <= 0 || (p.o_bs | p.o_ts | p.o_hs) % 8) return VC_E_INVALID;
    for (int r = 0; r < p.R; ++r)
        if (!p.part[r] || !p.lse[r]) return VC_E_INVALID;
    const int64_t n = (int64_t)p.B * p.Lq * p.H * 16;
    hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}
