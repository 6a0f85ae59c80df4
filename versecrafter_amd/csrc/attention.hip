// Flash-style attention forward for gfx950, head dim 128, non-causal, optional key-length mask.
//
// Replaces videox_fun.models.attention_utils.attention (flash-attn) as called by the reference at
// wan_transformer3d.py:394-399 (self-attention, k_lens = seq_lens) and :425-430 (T5 cross-attention,
// 512 keys, no mask):  out = softmax(q k^T / sqrt(D)) v, keys >= k_len masked.
//
// Structure (CDNA4, 64-lane waves, v_mfma_f32_32x32x16_bf16):
//  * workgroup = 8 waves = 256 query rows of one (batch, head) (4 waves / 128 rows when there are < 2048 keys); each
//    wave owns 32 query rows, its Q fragment lives in registers for the whole kernel.
//  * K/V tiles of 64 keys go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4 issued through inline asm with a
//    hand-placed s_waitcnt, so hipcc does not drain it early), 2-deep rings, one barrier per tile;
//  * software pipeline inside each wave: the MFMA chain of S(t+1) = K(t+1).Q^T is issued together with the
//    exp2 / row-sum / bf16-pack VALU work of tile t, and the O += V(t)^T.P(t)^T chain together with the row
//    maxima of tile t+1; the loop is unrolled by two so that every LDS address is register + immediate;
//  * online softmax with a deferred rescale: a row's exponent reference follows its running maximum only when the row has
//    outgrown it by more than 2^8 (per row, so results do not depend on wave composition); O / l is unchanged by the
//    choice of reference, the rescale of O (64 multiplies per lane) runs on a few tiles per sequence instead of ~40 %;
//  * S^T = K . Q^T ("swapped" product): the 32x32 accumulator has the query row on the LANE and the key
//    index in the registers, so row max / row sum are per-lane loops plus one exchange with lane^32,
//    and the accumulator is already the B operand of the next product  O^T = V^T . P^T  (no LDS trip).
//  * V^T fragments come from the row-major LDS image through ds_read_b64_tr_b16 (hardware transpose).
//  * LDS images: 256-byte rows; K chunks XOR (row&15) (conflict-free ds_read_b128),
//    V chunks XOR (((row&3)<<2)|((row>>2)&3)) (conflict-free transposed reads).
//  * workgroup -> (batch*head, q block) map is XCD-contiguous so the workgroups sharing one XCD's L2 walk
//    the same K/V.
#include <stdlib.h>

#include <type_traits>

#include "vc_common.h"
#include "vc_kernels.h"

#ifdef VC_ATTN_CLOCK   // tools/clock_attn.py: s_memtime / s_memrealtime around the tile loop of attn_fwd_pipe_kernel (diagnostic build only)
__device__ uint64_t* vc_attn_clock_buf = nullptr;
extern "C" int vc_debug_set_attn_clock(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(vc_attn_clock_buf), &buf, sizeof buf) == hipSuccess ? 0 : -1;
}
#endif

namespace {

#ifndef VC_ATTN_DEFER_MAX
#define VC_ATTN_DEFER_MAX 8
#endif
#ifndef VC_ATTN_ROWSUM
#define VC_ATTN_ROWSUM 0    // measured (round 2, tools/ab_attn.sh): 0: 1144 TF, 1: 1115, 2: 1126, 4: 1116 -- pinning the sums costs
#endif
#ifndef VC_ATTN_PACKED
#define VC_ATTN_PACKED 0    // measured (round 2, tools/ab_attn.sh, same box): 0: 1204 TF, 1 (v_pk_fma_f32 / v_pk_add_f32 pairs): 1140 TF
#endif
#ifndef VC_ATTN_ABLATE
#define VC_ATTN_ABLATE 0
#endif
#ifndef VC_ATTN_DEFAULT_SHAPE
#define VC_ATTN_DEFAULT_SHAPE 32    // 32: v_mfma_f32_32x32x16_bf16 (this file); 16: v_mfma_f32_16x16x32_bf16 (attention16.hip)
#endif
#ifndef VC_ATTN_FOLD
#define VC_ATTN_FOLD 0      // round 4: the softmax constant folded into Q (once per workgroup) and the row's exponent reference into the C
#endif                      // operand of the first QK^T MFMA of a tile: P = exp2(S) with no v_fma_f32 per element (A/B: tools/ab_attn.sh)
#ifndef VC_ATTN_QK_INTERLEAVE
#define VC_ATTN_QK_INTERLEAVE 0
#endif

constexpr int D = 128;
constexpr int KT = 64;      // keys per tile
constexpr int TILE_BYTES = KT * D * 2;          // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;     // K + V
constexpr int LDS_BYTES = 2 * STAGE_BYTES;      // double buffer = 64 KiB

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) char lds_char;

VC_DEVICE int k_off(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }
VC_DEVICE int v_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
VC_DEVICE int v_off(int row, int ch) { return row * 256 + ((ch ^ v_swz(row)) << 4); }

template <bool SEG>
VC_DEVICE int64_t tok_off(int t, int64_t ts, int seg_len, int64_t ss) {
    if (!SEG) return (int64_t)t * ts;
    const int s = t / seg_len;
    return (int64_t)s * ss + (int64_t)(t - s * seg_len) * ts;
}


struct AttnLaneConst {
    unsigned koff[8];     // K image byte offset of this lane's row r, chunk ks*2+h          (+ kb*8192 + stage)
    unsigned voff[4][2];  // V image byte offset for (db, half): keys +0..3 / +8..11 of step 0 (+ s*4096 + stage)
};

// =====================================================================================================
// Software pipelining: while the matrix pipe runs S(t+1) = K(t+1).Q^T the same wave's VALU does the
// exponentials / row sums / bf16 packing of tile t, and while it runs O += V(t)^T.P(t)^T the VALU does the row
// maxima of tile t+1 (MFMA and VALU are separate pipes: MI355X_MICROARCH "Two waves per SIMD").
// K and V tiles have separate 2-deep LDS rings because their lifetimes differ by half an iteration.
// Loads use 32-bit per-lane byte offsets against a wave-uniform 64-bit base (global_load_lds ... saddr).
// =====================================================================================================
constexpr int P_KST = 0, P_VST = 2 * TILE_BYTES;   // Kst[2] | Vst[2], 64 KiB total

VC_DEVICE void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst_uniform)
                 : "memory");
}

struct PipeLoad {
    unsigned koff[4], voff[4];   // byte offsets of this lane's 4 rows of the NEXT tile to load (K / V), incl. chunk
    int within[4];               // SEG only: row index inside its segment
};

template <bool SEG>
VC_DEVICE unsigned row_byte_off(int key, int64_t ts, int seg_len, int64_t ss) {
    if (!SEG) return (unsigned)key * (unsigned)(ts * 2);     // < 2^32 (checked by the launcher)
    const int sg = key / seg_len;
    return (unsigned)(((int64_t)sg * ss + (int64_t)(key - sg * seg_len) * ts) * 2);
}

// MERGE (T5 cross-attention): the keys from pad_from[b] on are IDENTICAL rows (the reference zero-pads every prompt to 512
// tokens before text_embedding, VC.py:358-363, and attends over all of them, WT.py:425-430): n equal keys contribute
// n * exp(s) to numerator and denominator alike, so one of them is kept with log2(n) added to its exponent and the rest are
// skipped -- the same softmax with 512 - len fewer keys.
template <bool SEG, int NW, bool MERGE = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_pipe_kernel(VcAttnParams p, int nQ, int nwork) {
    constexpr int QB = NW * 32;          // query rows per workgroup
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

    int k_len = (p.k_len > 0 && p.k_len < p.Lk) ? p.k_len : p.Lk;
    float pad_bias = 0.f;              // added to the raw logit of key k_len - 1 (MERGE)
    if (MERGE) {
        const int from = p.pad_from[b];
        if (from >= 0 && from < p.Lk - 1) {
            k_len = from + 1;
            pad_bias = VC_ATTN_FOLD ? log2f((float)(p.Lk - from)) : log2f((float)(p.Lk - from)) / (p.scale * 1.4426950408889634f);
        }
    }
    const int nt = (k_len + KT - 1) / KT;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
    constexpr int RPW = KT / NW;                          // tile rows staged per wave (16 or 8)
    const int pc = lane & 15;

    // stage K(t) / V(t): RPW/4 wave-instructions (pieces of 4 tile rows) each.  A wave's pieces are 16 rows apart, so the
    // swizzled chunk of a lane is the same in all of them: ONE per-lane byte offset per operand; the tile and the piece
    // are selected through the wave-uniform 64-bit base (SALU only -- no vector instruction per piece).  Tiles are
    // staged in order (K: 0, 1, 2, ...; V: 0, 1, ...), so the byte offset of a tile's first key is carried in scalar
    // registers (StreamPos).  Segmented token axis (Ulysses receive layout, segments of seg_len >= 64 keys): a tile
    // straddles at most one segment boundary; pieces past it take the next segment's base, the piece on it and the last,
    // partial tile (and seg_len < 64) go through the generic per-row path with rows clamped to Lk-1.
    struct StreamPos { unsigned A; int w0; };          // A: byte offset of the tile's first key; w0: its index in its segment
    const unsigned k_ts2 = (unsigned)(p.k_ts * 2), v_ts2 = (unsigned)(p.v_ts * 2);
    const unsigned k_jump = SEG ? (unsigned)((p.k_ss - (int64_t)p.seg_len * p.k_ts) * 2) : 0u;
    const unsigned v_jump = SEG ? (unsigned)((p.v_ss - (int64_t)p.seg_len * p.v_ts) * 2) : 0u;
    const bool fast_ok = !SEG || p.seg_len >= KT;
    const int qbase = __builtin_amdgcn_readfirstlane(NW == 8 ? (wave & 3) + 8 * (wave >> 2) : wave);   // pieces qbase + 4j
    const int lrow = 4 * qbase + (lane >> 4);          // this lane's tile row in piece j = 0 (piece j: + 16 j)
    const unsigned klane = (unsigned)lrow * k_ts2 + ((pc ^ (lrow & 15)) << 4);
    const unsigned vlane = (unsigned)lrow * v_ts2 + ((pc ^ v_swz(lrow)) << 4);
    auto stage_one = [&](int t, bool is_k, int st, StreamPos& sp) {
        const unsigned ts2 = is_k ? k_ts2 : v_ts2, jump = is_k ? k_jump : v_jump;
        const char* base = (const char*)(is_k ? kp : vp);
        const unsigned dst = lds0 + (is_k ? P_KST : P_VST) + st * TILE_BYTES + qbase * 1024;
        const int bnd = SEG ? p.seg_len - sp.w0 : KT;              // first tile row that belongs to the next segment
        bool fast = fast_ok && (t + 1) * KT <= p.Lk;
        if (SEG) {                                                 // a piece ON the boundary needs per-row addresses
#pragma unroll
            for (int j = 0; j < RPW / 4; ++j) {
                const int row0 = 4 * qbase + 16 * j;
                fast = fast && !(bnd > row0 && bnd < row0 + 4);
            }
        }
        if (fast) {
#pragma unroll
            for (int j = 0; j < RPW / 4; ++j) {
                const int row0 = 4 * qbase + 16 * j;               // first tile row of the piece (wave-uniform)
                unsigned a = sp.A + (unsigned)(16 * j) * ts2;
                if (SEG) a += row0 >= bnd ? jump : 0u;
                glds16_s(is_k ? klane : vlane, base + a, __builtin_amdgcn_readfirstlane(dst + j * 4096));
            }
        } else {
#pragma unroll
            for (int j = 0; j < RPW / 4; ++j) {
                const int row = lrow + 16 * j;
                int key = t * KT + row;
                key = key < p.Lk ? key : p.Lk - 1;
                const unsigned sw = is_k ? (unsigned)(row & 15) : (unsigned)v_swz(row);
                glds16_s(row_byte_off<SEG>(key, is_k ? p.k_ts : p.v_ts, p.seg_len, is_k ? p.k_ss : p.v_ss) + ((pc ^ sw) << 4),
                         base, __builtin_amdgcn_readfirstlane(dst + j * 4096));
            }
        }
        sp.A += (unsigned)KT * ts2;
        if (SEG && fast_ok) {
            sp.w0 += KT;
            if (sp.w0 >= p.seg_len) { sp.w0 -= p.seg_len; sp.A += jump; }
        }
    };
    StreamPos kpos{0u, 0}, vpos{0u, 0};
    auto stage = [&](int t, bool do_k, bool do_v, int kst, int vst) {
        if (do_k) stage_one(t, true, kst, kpos);
        if (do_v) stage_one(t, false, vst, vpos);
    };

    stage(0, true, true, 0, 0);
    if (nt > 1) stage(1, true, false, 1, 0);

    // ---- Q fragment ----
    const int q_row = qb * QB + wave * 32 + r;
    const int q_row_c = q_row < p.Lq ? q_row : p.Lq - 1;
    bf16x8 qf[8];
    {
        const bf16_t* qrow = qp + tok_off<SEG>(q_row_c, p.q_ts, p.seg_len, p.q_ss) + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qrow + ks * 16);
#if VC_ATTN_FOLD
        // Q' = bf16(q * scale * log2 e): the accumulator of K Q'^T is the logit in log2 units (one more bf16 rounding of q, 2^-9 relative:
        // 1e-3 of a log2 unit on a logit, a tenth of the rounding P itself gets)
        const float cq = p.scale * 1.4426950408889634f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[ks][j] = (__bf16)((float)qf[ks][j] * cq);
#endif
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
    }
    AttnLaneConst lc;
    {
        const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
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
    const float c = VC_ATTN_FOLD ? 1.0f : p.scale * 1.4426950408889634f;     // FOLD: S, m_run, m_new are in log2 units already
    float m_run = -1e30f, m_new = -1e30f, l_run = 0.f;
#if VC_ATTN_FOLD
    // sixteen copies of -m_run: the C operand of the first MFMA of every 32-key half, so that S arrives as (logit - reference) and
    // the weights are exp2(S) without a multiply-add per element.  Invariant at the top of body(t): Sc = S(t) - m_run.  The reference
    // moves rarely (deferred rescale): the branch that moves it also shifts the S(t) at hand and rewrites the sixteen copies.
    f32x16 bias;
#pragma unroll
    for (int e = 0; e < 16; ++e) bias[e] = 0.f;
#endif

    auto qk = [&](const char* kbuf, f32x16 (&S)[2]) {
#if VC_ATTN_QK_INTERLEAVE      // the two 32-key halves alternate: every MFMA depends on the one TWO back, not on its predecessor
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) S[kb][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 kf = *(const bf16x8*)(kbuf + lc.koff[ks] + kb * 8192);
                S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], S[kb], 0, 0, 0);
            }
#else
#if VC_ATTN_ABLATE == 2        // timing ablation only (wrong results): half of the K fragment reads (= a quarter of all LDS reads) removed
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) S[kb][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(kbuf + lc.koff[ks]);
            S[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], S[0], 0, 0, 0);
            S[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[(ks + 1) & 7], S[1], 0, 0, 0);
        }
#elif VC_ATTN_ABLATE == 4      // timing ablation only (wrong results): half of the QK^T MFMAs and K fragment reads = the matrix-pipe cycles and LDS bytes of an fp8 QK^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) S[kb][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kbuf + lc.koff[ks] + kb * 8192);
                S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], S[kb], 0, 0, 0);
            }
        }
#elif VC_ATTN_FOLD
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kbuf + lc.koff[ks] + kb * 8192);
                S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? bias : S[kb], 0, 0, 0);
            }
        }
#else
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) S[kb][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kbuf + lc.koff[ks] + kb * 8192);
                S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], S[kb], 0, 0, 0);
            }
        }
#endif
#endif
    };
    auto mask_tail = [&](f32x16 (&S)[2], int t) {      // keys >= k_len of the (last) tile t
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = t * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (key >= k_len) S[kb][e] = -1e30f;
                else if (MERGE && key == k_len - 1) S[kb][e] += pad_bias;
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

    // ---- prologue: S(0) and its row maxima ----
    f32x16 Sa[2], Sb[2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    qk(smem + P_KST, Sa);
    if (nt == 1) mask_tail(Sa, 0);
    m_new = fmaxf(m_new, row_max(Sa));
#if VC_ATTN_FOLD
    m_run = m_new;                      // O and l are still zero: the first reference is simply taken
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) Sa[kb][e] -= m_run;
#pragma unroll
    for (int e = 0; e < 16; ++e) bias[e] = -m_run;
    asm volatile("" : "+v"(bias));      // opaque: or hipcc rebuilds the sixteen copies in front of every MFMA that takes them
#endif

#ifdef VC_ATTN_TRACE     // (with VC_ATTN_CLOCK) s_memtime sums per section of a tile: the fences change hipcc's schedule -- indicative only
    uint64_t trc_acc[5] = {0, 0, 0, 0, 0}, trc_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(trc_last) :: "memory");
    auto trc_mark = [&](int i) {
        uint64_t now;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        trc_acc[i] += now - trc_last;
        trc_last = now;
    };
#define VC_ATTN_MARK(i) trc_mark(i)
#else
#define VC_ATTN_MARK(i) do {} while (0)
#endif
    // one pipelined iteration; PAR = t & 1 (static LDS stages); MORE: tile t+1 exists (compute S(t+1) into Sn);
    // MASK: tile t+1 is the last one.  Sc = S(t) on entry; the caller swaps the roles of the two buffers.
    auto body = [&](int t, f32x16 (&Sc)[2], f32x16 (&Sn)[2], auto par_tag, auto more_tag, auto mask_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool MORE = decltype(more_tag)::value, MASK = decltype(mask_tag)::value;
        VC_ATTN_MARK(4);
#if VC_ATTN_ABLATE != 5    // 5: timing ablation only (a data race): what the wait for the tiles requested ONE beat earlier costs = what a deeper ring could win
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();   // K(t+1), V(t) landed; all waves are past QK(t) [Kst[PAR]] and PV(t-1) [Vst[PAR^1]]
        VC_ATTN_MARK(0);
        if (t + 2 < nt) stage(t + 2, true, false, PAR, 0);
        if (MORE) stage(t + 1, false, true, 0, PAR ^ 1);
        VC_ATTN_MARK(1);
        // Deferred rescale (VC_ATTN_DEFER_MAX = T > 0): a row's exponent reference m_run follows its true running maximum
        // m_new only once the row has outgrown it by more than 2^T; until then P = exp2((S - m_run) c) may exceed 1
        // (< 2^T: harmless in bf16 / fp32).  O and l carry the same factor, so the quotient is unchanged; only the
        // rounding points of P move.  The decision is per row (lane), so a row's result does not depend on which other
        // rows share its wave.  T = 0: classic form (reference follows the maximum; exact skip when no row moved).
        {
            const bool moved = VC_ATTN_DEFER_MAX > 0 ? (m_new - m_run) * c > (float)VC_ATTN_DEFER_MAX : m_new > m_run;
            if (__any(moved)) {
                const float m_ref = moved ? m_new : m_run;
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_ref) * c);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) O[i][e] *= alpha;
#if VC_ATTN_FOLD
                const float shift = m_run - m_ref;               // S(t) was accumulated on top of -m_run
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) Sc[kb][e] += shift;
#pragma unroll
                for (int e = 0; e < 16; ++e) bias[e] = -m_ref;
                asm volatile("" : "+v"(bias));
#endif
                m_run = m_ref;
            }
        }
        [[maybe_unused]] const float mc = m_run * c;
        // ---- phase 1: MFMA S(t+1) = K(t+1).Q^T  ||  VALU P(t) = exp2(S(t) c - m c), row sums, bf16 pack ----
        if (MORE) qk(smem + P_KST + (PAR ^ 1) * TILE_BYTES, Sn);
        // row sums.  hipcc sinks this 32-add chain behind the next barrier; VC_ATTN_ROWSUM = n > 0 builds n independent partial
        // sums pinned to this phase instead (A/B knob: every pinned form measured 1.5-2.5 % SLOWER, so 0 is the default)
        constexpr int NPS = VC_ATTN_ROWSUM > 0 ? VC_ATTN_ROWSUM : 1;
        float ps[NPS];
#pragma unroll
        for (int i = 0; i < NPS; ++i) ps[i] = 0.f;
        bf16x8 pf[4];
#if VC_ATTN_PACKED
        // two elements per VALU instruction where the ISA has a packed fp32 form (v_pk_fma_f32, v_pk_add_f32): the scale /
        // subtract and the row sums take 16 + 16 instructions per tile instead of 32 + 32 on the issue port the MFMAs share;
        // the row sum becomes two interleaved partial sums (even / odd keys), added once per tile
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t c2 = {c, c}, nmc2 = {-mc, -mc};
        f32x2_t ps2 = {0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f32x2_t x = {Sc[s >> 1][8 * (s & 1) + j], Sc[s >> 1][8 * (s & 1) + j + 1]};
                const f32x2_t y = __builtin_elementwise_fma(x, c2, nmc2);
                const f32x2_t pe = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
                ps2 += pe;
                pf[s][j] = (__bf16)pe[0];
                pf[s][j + 1] = (__bf16)pe[1];
            }
        l_run += ps2[0] + ps2[1];
#else
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#if VC_ATTN_ABLATE == 1        // timing ablation only (wrong results): the transcendental replaced by a plain VALU op
                const float pe = Sc[s >> 1][8 * (s & 1) + j] * c - mc;
#elif VC_ATTN_FOLD
                const float pe = __builtin_amdgcn_exp2f(Sc[s >> 1][8 * (s & 1) + j]);
#else
                const float pe = __builtin_amdgcn_exp2f(Sc[s >> 1][8 * (s & 1) + j] * c - mc);
#endif
                ps[(s * 8 + j) % NPS] += pe;
                pf[s][j] = (__bf16)pe;
            }
        {
            float tot = ps[0];
#pragma unroll
            for (int i = 1; i < NPS; ++i) tot += ps[i];
            l_run += tot;
        }
#endif
        if (VC_ATTN_ROWSUM > 0) asm volatile("" : "+v"(l_run));
        VC_ATTN_MARK(2);
        // ---- phase 2: MFMA O += V(t)^T.P(t)^T  ||  VALU row maxima of S(t+1) ----
        const char* vbuf = smem + P_VST + PAR * TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
#if VC_ATTN_ABLATE == 3        // timing ablation only (wrong results): K halved as in 2 AND half of the V fragment reads removed: half of all LDS reads
                const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + lc.voff[db][0] + (s & 1) * 4096));
                const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + lc.voff[db][1] + (s & 1) * 4096));
#else
                const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + lc.voff[db][0] + s * 4096));
                const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + lc.voff[db][1] + s * 4096));
#endif
                const bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                O[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], O[db], 0, 0, 0);
            }
        if (MORE) {
            if (MASK) mask_tail(Sn, t + 1);
            m_new = fmaxf(m_new, row_max(Sn) + (VC_ATTN_FOLD ? m_run : 0.f));       // true running maximum (>= m_run)
        }
        VC_ATTN_MARK(3);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
#ifdef VC_ATTN_CLOCK
    uint64_t clk_c0, clk_r0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk_c0), "=s"(clk_r0) :: "memory");
#endif
    // tiles 0 .. nt-3 in pairs (S(t) alternates between Sa and Sb), then the masked and the final tile
    int t = 0;
    for (; t + 3 < nt; t += 2) {
        body(t, Sa, Sb, P0{}, T_{}, F_{});
        body(t + 1, Sb, Sa, P1{}, T_{}, F_{});
    }
    // here t is even and nt - t is 1, 2 or 3
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

#ifdef VC_ATTN_CLOCK
    {
        uint64_t clk_c1, clk_r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk_c1), "=s"(clk_r1) :: "memory");
        if (vc_attn_clock_buf && lane == 0) {
            uint64_t* o = vc_attn_clock_buf + ((size_t)blockIdx.x * NW + wave) * 8;
            o[0] = clk_c1 - clk_c0;
            o[1] = clk_r1 - clk_r0;
#ifdef VC_ATTN_TRACE
            for (int i = 0; i < 5; ++i) o[2 + i] = trc_acc[i];
            o[7] = (uint64_t)nt;
#endif
        }
    }
#endif
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    // Epilogue (round 4: the widened store tail of attn_short_kernel, cdna_hip_programming.md T21): a lane holds 8-byte pieces (4 dims)
    // at dims db*32 + 8*g4 + 4*h and lane r + 32 holds the neighbouring pieces of the same row; one permlane32 swap per register pairs
    // them up so that every lane stores 16 contiguous bytes -- 8 dwordx4 instead of 16 dwordx2 (the row-per-lane store tail is
    // issue-bound).  Rows whose stride is not a multiple of 16 bytes keep the 8-byte form.
    const bool wide = (((p.o_ts | p.o_hs | p.o_bs | p.o_ss) & 7) == 0) && (((uintptr_t)p.out & 15) == 0);
    const int q_row_s = q_row < p.Lq ? q_row : p.Lq - 1;
    if (wide) {
        bf16_t* orow = op + tok_off<SEG>(q_row_s, p.o_ts, p.seg_len, p.o_ss) + 8 * h;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float va[4], vb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { va[e] = O[db][8 * j + e] * inv; vb[e] = O[db][8 * j + 4 + e] * inv; }
                const uint2 pa = pack4(va), pb = pack4(vb);       // g4 = 2j and g4 = 2j + 1 of this half
                const auto sx = __builtin_amdgcn_permlane32_swap(pa.x, pb.x, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(pa.y, pb.y, false, false);
                // lanes 0-31: {own 2j, partner's 2j}; lanes 32-63: {partner's 2j+1, own 2j+1}
                if (q_row < p.Lq) *(uint4*)(orow + db * 32 + 16 * j) = uint4{sx[0], sy[0], sx[1], sy[1]};
            }
    } else if (q_row < p.Lq) {
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

// =====================================================================================================
// Short key sequences (T5 cross-attention, WT.py:425-430: 512 prompt positions of which the zero-padded tail folds into one key,
// so 2-4 tiles of 64 keys remain): the whole K and V of one (batch, head) fit in LDS.  A workgroup of 4 waves stages them ONCE
// (swizzled images as above) and then walks a chunk of the query axis, 32 rows per wave per trip, with no further barrier:
// per tile QK^T (16 MFMA), exact online softmax, PV (16 MFMA).  No software pipeline and nothing to spill -- the kernel is bound
// by reading Q and writing O once (the pipelined kernel's 2-tile case was all prologue / tail and spilled 893 VGPRs).
// =====================================================================================================
constexpr int SHORT_MAX_TILES = 4;

template <bool MERGE>
__global__ __launch_bounds__(256, 2) void attn_short_kernel(VcAttnParams p, int nchunks, int chunk_rows, int nt_lds, int nwork) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int per_xcd = gridDim.x >> 3;
    const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (id >= nwork) return;
    const int bh = id / nchunks, chunk = id - bh * nchunks;
    const int b = bh / p.H, head = bh - b * p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const bf16_t* qp = (const bf16_t*)p.q + (int64_t)b * p.q_bs + (int64_t)head * p.q_hs;
    const bf16_t* kp = (const bf16_t*)p.k + (int64_t)b * p.k_bs + (int64_t)head * p.k_hs;
    const bf16_t* vp = (const bf16_t*)p.v + (int64_t)b * p.v_bs + (int64_t)head * p.v_hs;
    bf16_t* op = (bf16_t*)p.out + (int64_t)b * p.o_bs + (int64_t)head * p.o_hs;

    int k_len = (p.k_len > 0 && p.k_len < p.Lk) ? p.k_len : p.Lk;
    float pad_bias = 0.f;
    if (MERGE) {
        const int from = p.pad_from[b];
        if (from >= 0 && from < p.Lk - 1) {
            k_len = from + 1;
            pad_bias = log2f((float)(p.Lk - from)) / (p.scale * 1.4426950408889634f);
        }
    }
    const int nt = (k_len + KT - 1) / KT;              // <= nt_lds (launcher)
    char* kimg = smem;
    char* vimg = smem + nt_lds * TILE_BYTES;
    // ---- stage K and V of this (batch, head): 16-byte chunks, consecutive lanes on consecutive chunks of a row ----
    for (int idx = tid; idx < nt * KT * 16; idx += 256) {
        const int row = idx >> 4, ch = idx & 15;       // row = key index (tile-major: tile = row / 64)
        const int key = row < p.Lk ? row : p.Lk - 1;
        const int tr = row & (KT - 1), tile = row >> 6;
        *(uint4*)(kimg + tile * TILE_BYTES + k_off(tr, ch)) = *(const uint4*)(kp + (int64_t)key * p.k_ts + ch * 8);
        *(uint4*)(vimg + tile * TILE_BYTES + v_off(tr, ch)) = *(const uint4*)(vp + (int64_t)key * p.v_ts + ch * 8);
    }
    __syncthreads();

    AttnLaneConst lc;
    {
        const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) lc.koff[ks] = k_off(r, ks * 2 + h);
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                lc.voff[db][hf] = v_off(4 * (g >> 1) + q4 + 8 * hf, db * 4 + 2 * (g & 1) + (p4 >> 1)) + 8 * (p4 & 1);
    }
    const float c = p.scale * 1.4426950408889634f;
    const int row_end = min(p.Lq, (chunk + 1) * chunk_rows);
    // the Q rows of the NEXT trip are requested before this trip's arithmetic (the kernel is bound by these reads and the O
    // writes: one more trip of loads in flight per wave)
    auto load_q = [&](int q0, bf16x8 (&dst)[8]) {
        const int row = q0 + r < p.Lq ? q0 + r : p.Lq - 1;
        const bf16_t* qrow = qp + (int64_t)row * p.q_ts + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) dst[ks] = *(const bf16x8*)(qrow + ks * 16);
    };
    bf16x8 qf[8], qn[8];
    const int q_first = chunk * chunk_rows + wave * 32;
    if (q_first < row_end) load_q(q_first, qf);
    for (int q0 = q_first; q0 < row_end; q0 += 128) {
        const int q_row = q0 + r;
        const bool more = q0 + 128 < row_end;
        if (more) {
            load_q(q0 + 128, qn);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qn[ks]));      // issue the loads here, not where they are used
        }
        f32x16 O[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[i][e] = 0.f;
        float m_run = -1e30f, l_run = 0.f;
        for (int t = 0; t < nt; ++t) {
            f32x16 S[2];
            const char* kbuf = kimg + t * TILE_BYTES;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) S[kb][e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const bf16x8 kf = *(const bf16x8*)(kbuf + lc.koff[ks] + kb * 8192);
                    S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], S[kb], 0, 0, 0);
                }
            }
            if (t == nt - 1) {
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
            const char* vbuf = vimg + t * TILE_BYTES;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + lc.voff[db][0] + s * 4096));
                    const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vbuf + lc.voff[db][1] + s * 4096));
                    const bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                    O[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], O[db], 0, 0, 0);
                }
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        // Epilogue: a lane holds 8-byte pieces (4 dims) at dims db*32 + 8*g4 + 4*h; the two halves of the wave (lane r, r + 32)
        // hold the neighbouring pieces of the same row.  One permlane32 swap per register pairs them up -- lanes 0-31 keep the
        // even g4 of both halves, lanes 32-63 the odd ones -- so that every lane stores 16 contiguous bytes: 8 dwordx4 stores
        // instead of 16 dwordx2 (the row-per-lane store tail is issue-bound: MI355X_MICROARCH.md, attention epilogue).
        {
            bf16_t* orow = op + (int64_t)q_row * p.o_ts + 8 * h;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float va[4], vb[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { va[e] = O[db][8 * j + e] * inv; vb[e] = O[db][8 * j + 4 + e] * inv; }
                    const uint2 pa = pack4(va), pb = pack4(vb);       // g4 = 2j and g4 = 2j + 1 of this half
                    const auto sx = __builtin_amdgcn_permlane32_swap(pa.x, pb.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(pa.y, pb.y, false, false);
                    // lanes 0-31: {own 2j, partner's 2j}; lanes 32-63: {partner's 2j+1, own 2j+1}
                    if (q_row < p.Lq) *(uint4*)(orow + db * 32 + 16 * j) = uint4{sx[0], sy[0], sx[1], sy[1]};
                }
        }
        if (more) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qf[ks] = qn[ks];
        }
    }
}

template <bool MERGE>
int launch_attn_short(const VcAttnParams& p, int nt_lds, hipStream_t stream) {
    const int lds = nt_lds * 2 * TILE_BYTES;
    static std::atomic<uint64_t> attr_done{0};
    if (!vc_set_lds_once(attr_done, (const void*)attn_short_kernel<MERGE>, SHORT_MAX_TILES * 2 * TILE_BYTES)) return VC_E_HIP;
    // ~10 workgroups per CU (two are resident at a time: five rounds, so the ragged last round costs little; staging K / V
    // again per workgroup is 32-64 KB against 256 KB of Q + O per trip); a chunk is a whole number of 128-row trips
    const int bh = p.B * p.H;
    int nchunks = (2560 + bh - 1) / bh;
    const int max_chunks = (p.Lq + 127) / 128;
    if (nchunks > max_chunks) nchunks = max_chunks;
    if (nchunks < 1) nchunks = 1;
    const int chunk_rows = ((p.Lq + nchunks - 1) / nchunks + 127) / 128 * 128;
    nchunks = (p.Lq + chunk_rows - 1) / chunk_rows;
    const int nwork = bh * nchunks;
    const int grid = (nwork + 7) / 8 * 8;
    hipLaunchKernelGGL(attn_short_kernel<MERGE>, dim3(grid), dim3(256), lds, stream, p, nchunks, chunk_rows, nt_lds, nwork);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}

template <bool SEG, int NW, bool MERGE = false>
int launch_attn_pipe(const VcAttnParams& p, hipStream_t stream) {
    constexpr int QB = NW * 32;
    static std::atomic<uint64_t> attr_done{0};
    if (!vc_set_lds_once(attr_done, (const void*)attn_fwd_pipe_kernel<SEG, NW, MERGE>, LDS_BYTES)) return VC_E_HIP;
    const int nQ = (p.Lq + QB - 1) / QB;
    const int nwork = p.B * p.H * nQ;
    const int grid = (nwork + 7) / 8 * 8;
    hipLaunchKernelGGL((attn_fwd_pipe_kernel<SEG, NW, MERGE>), dim3(grid), dim3(NW * 64), LDS_BYTES, stream, p, nQ, nwork);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}

}  // namespace

int vc_launch_attention(const VcAttnParams& p, hipStream_t stream) {
    if (!p.q || !p.k || !p.v || !p.out || p.B <= 0 || p.H <= 0 || p.Lq <= 0 || p.Lk <= 0) return VC_E_INVALID;
    if ((p.q_ts | p.k_ts | p.v_ts | p.q_hs | p.k_hs | p.v_hs | p.q_bs | p.k_bs | p.v_bs) % 8) return VC_E_UNSUPPORTED;
    if ((p.o_ts | p.o_hs | p.o_bs) % 4) return VC_E_UNSUPPORTED;
    if (p.seg_len < 0 || (p.seg_len > 0 && ((p.q_ss | p.k_ss | p.v_ss) % 8 || p.o_ss % 4))) return VC_E_UNSUPPORTED;
    // the kernel addresses K/V rows with 32-bit byte offsets from the per-(batch, head) base
    const int64_t span_k = (p.seg_len > 0 ? (int64_t)((p.Lk - 1) / p.seg_len) * p.k_ss + (int64_t)p.seg_len * p.k_ts
                                          : (int64_t)p.Lk * p.k_ts) * 2;
    const int64_t span_v = (p.seg_len > 0 ? (int64_t)((p.Lk - 1) / p.seg_len) * p.v_ss + (int64_t)p.seg_len * p.v_ts
                                          : (int64_t)p.Lk * p.v_ts) * 2;
    if (span_k >= (1ll << 32) || span_v >= (1ll << 32) || p.k_ts < 0 || p.v_ts < 0) return VC_E_UNSUPPORTED;
    // 256 query rows per workgroup (8 waves) halve the K/V LDS-DMA per FLOP on long sequences; short key sequences
    // (T5 cross-attention, 512 keys) are prologue-dominated and run better with twice as many, smaller workgroups
    if (p.lse) {          // log-sum-exp output (ring attention): the 16x16x32 kernel only, plain layout
        if (p.seg_len != 0 || p.pad_merge || (p.o_ts | p.o_hs | p.o_bs) % 8) return VC_E_UNSUPPORTED;
        return vc_launch_attention_mfma16(p, stream);
    }
    if (p.pad_merge) {
        if (p.seg_len > 0 || p.k_len > 0 || p.B > 8) return VC_E_UNSUPPORTED;
        int nt_max = 0;                                  // key tiles left after folding the padded tail, worst sample
        for (int i = 0; i < p.B; ++i) {
            const int from = p.pad_from[i];
            const int kl = (from >= 0 && from < p.Lk - 1) ? from + 1 : p.Lk;
            nt_max = nt_max > (kl + KT - 1) / KT ? nt_max : (kl + KT - 1) / KT;
        }
        if (nt_max <= SHORT_MAX_TILES && p.Lq >= 128) return launch_attn_short<true>(p, nt_max, stream);
        return vc_launch_attention_stream(p, stream);        // 5-8 tiles: plain double-buffered kernel (attention_stream.hip)
    }
    if (p.seg_len == 0 && p.Lk <= SHORT_MAX_TILES * KT && p.Lq >= 1024) return launch_attn_short<false>(p, (p.Lk + KT - 1) / KT, stream);
    // MFMA shape of the pipelined kernel (round 3 A/B): variant 16 / 32 force one, 0 takes VC_ATTN_DEFAULT_SHAPE
    if (p.variant != 0 && p.variant != 16 && p.variant != 32) return VC_E_INVALID;
    if ((p.variant ? p.variant : VC_ATTN_DEFAULT_SHAPE) == 16 && p.seg_len == 0 && (p.o_ts | p.o_hs | p.o_bs) % 8 == 0)
        return vc_launch_attention_mfma16(p, stream);
    if (p.Lk >= 2048) return p.seg_len > 0 ? launch_attn_pipe<true, 8>(p, stream) : launch_attn_pipe<false, 8>(p, stream);
    return p.seg_len > 0 ? launch_attn_pipe<true, 4>(p, stream) : launch_attn_pipe<false, 4>(p, stream);
}
