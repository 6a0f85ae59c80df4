// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias)
//
// Replaces what the reference gets from cuBLAS through nn.Linear / nn.Conv3d(k=s=(1,2,2)):
//   q/k/v/o projections      wan_transformer3d.py:366-369, 385-387, 404, 420-422, 435
//   FFN                      wan_transformer3d.py:557-559, 606
//   before_proj / after_proj wan_transformer3d_versecrafter.py:104-110, 114, 121
//   patch embeddings         wan_transformer3d.py:758-759 ; wan_transformer3d_versecrafter.py:199-201
//   text embedding, head     wan_transformer3d.py:760-762, 626
// with the elementwise work that follows each of them fused into the epilogue
// (bias, tanh-GELU, residual, adaLN gate, GeoAdapter hint injection).
//
// Structure: BMxBNx64 block tile, v_mfma_f32_16x16x32_bf16, both operands K-contiguous,
// staged HBM->LDS with global_load_lds_dwordx4 (no VGPR round trip) into a double buffer.
// LDS image: [rows][64 bf16] = 128-byte rows, 16-byte chunks XOR-swizzled by ((row>>1)&7) so that
// every 16-lane group of a ds_read_b128 fragment read hits 16 distinct 16-byte slots of the 256-byte
// bank row.  global_load_lds writes lane-linear, so the swizzle is applied to the per-lane SOURCE
// address and again on the read.
// The MFMA is issued as D = Wfrag . Afrag^T, so a lane holds 4 CONSECUTIVE n of one row m:
// the epilogue loads/stores 8-byte bf16x4 pieces.
// Workgroup -> tile map: XCD-contiguous (blocks b, b+8, ... share an L2) and grouped 8 tile-rows deep so
// that the 32 tiles resident on one XCD share A / W panels.
#include "vc_common.h"
#include "vc_kernels.h"

#ifndef VC_GEMM_SMALL_TILES
#define VC_GEMM_SMALL_TILES 256     // fewer 256 x 256 tiles than CUs: the 128 x 128 kernel is used (see vc_launch_gemm)
#endif

#ifdef VC_PP_TRACE
__device__ uint64_t* vc_pp_trace_buf = nullptr;
extern "C" int vc_debug_set_gemm_trace(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(vc_pp_trace_buf), &buf, sizeof buf) == hipSuccess ? 0 : -1;
}
#endif

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int BM_, int BN_, int WM_, int WN_>
struct GemmCfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int BK = 64;
    static constexpr int THREADS = WM * WN * 64;
    static constexpr int WTM = BM / WM, WTN = BN / WN;   // per-wave output tile
    static constexpr int MI = WTM / 16, NI = WTN / 16;   // 16x16 MFMA tiles per wave
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static constexpr int A_LOADS = A_BYTES / (THREADS * 16);
    static constexpr int B_LOADS = B_BYTES / (THREADS * 16);
    static_assert(A_BYTES % (THREADS * 16) == 0 && B_BYTES % (THREADS * 16) == 0, "tile/threads");
};

// one 16-byte-per-lane LDS-DMA: LDS destination = wave-uniform base + lane*16
VC_DEVICE void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}

typedef __attribute__((address_space(3))) char lds_char_t;

// LDS-DMA with a wave-uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset; inline asm keeps it out of hipcc's
// vmcnt bookkeeping (the kernel waits for it itself, right before the barrier that publishes the stage)
VC_DEVICE void glds16_sbase(unsigned voff, const void* sbase, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst_uniform)
                 : "memory");
}

template <class Cfg>
VC_DEVICE void stage_tile(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
                          int m0, int n0, int M, int N, int k0, char* buf, int tid, int wave) {
#pragma unroll
    for (int i = 0; i < Cfg::A_LOADS; ++i) {
        const int q = i * Cfg::THREADS + tid;
        const int row = q >> 3, pc = q & 7;
        const int c = pc ^ ((row >> 1) & 7);
        int gr = m0 + row;
        gr = gr < M ? gr : M - 1;
        glds16(A + (int64_t)gr * lda + k0 + c * 8, buf + (i * Cfg::THREADS + wave * 64) * 16);
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_LOADS; ++i) {
        const int q = i * Cfg::THREADS + tid;
        const int row = q >> 3, pc = q & 7;
        const int c = pc ^ ((row >> 1) & 7);
        int gr = n0 + row;
        gr = gr < N ? gr : N - 1;
        glds16(W + (int64_t)gr * ldw + k0 + c * 8, buf + Cfg::A_BYTES + (i * Cfg::THREADS + wave * 64) * 16);
    }
}

template <class Cfg, bool SB, int EPI>
__global__ __launch_bounds__(Cfg::THREADS) void gemm_bf16_kernel(VcGemmParams p, int nTm, int nTn, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MI = Cfg::MI, NI = Cfg::NI;

    // ---- workgroup -> tile (XCD-contiguous, grouped) ----
    const int per_xcd = gridDim.x >> 3;
    const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (id >= ntiles) return;
    constexpr int GROUP_M = 8;
    const int width = GROUP_M * nTn;
    const int group = id / width;
    const int first_m = group * GROUP_M;
    const int gsz = min(nTm - first_m, GROUP_M);
    const int tm = first_m + (id % width) % gsz;
    int tn = (id % width) / gsz;
    // grouped launch: the column-tile axis runs over the problems back to back (they share the A panel in L2)
    const int nTn1 = p.ngroups > 1 ? nTn / p.ngroups : nTn;
    const int grp = tn / nTn1;
    tn -= grp * nTn1;
    const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)(grp == 0 ? p.W : p.Wg[grp - 1]);

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / Cfg::BK;
    // SB: 32-bit per-lane byte offsets computed once + a scalar base advanced per K-step
    unsigned aoff[Cfg::A_LOADS], boff[Cfg::B_LOADS];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char_t*)smem;
    if (SB) {
#pragma unroll
        for (int i = 0; i < Cfg::A_LOADS; ++i) {
            const int q = i * Cfg::THREADS + tid, row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
            int gr = m0 + row;
            gr = gr < p.M ? gr : p.M - 1;
            aoff[i] = (unsigned)(((int64_t)gr * p.lda + c * 8) * 2);
        }
#pragma unroll
        for (int i = 0; i < Cfg::B_LOADS; ++i) {
            const int q = i * Cfg::THREADS + tid, row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
            int gr = n0 + row;
            gr = gr < p.N ? gr : p.N - 1;
            boff[i] = (unsigned)(((int64_t)gr * p.ldw + c * 8) * 2);
        }
    }
    // all 8 pieces of a K-step in ONE asm statement: M0 (LDS destination of the wave) is saved / restored once and
    // stepped by 8 KiB between pieces (piece i of a wave lands at stage + wave*1 KiB + i*8 KiB; the W tile follows A)
    static_assert(!SB || (Cfg::A_LOADS == 4 && Cfg::B_LOADS == 4 && Cfg::THREADS * 16 == 0x2000),
                  "stage_sb assumes 256x256x64 / 512 threads");
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
    auto stage_sb = [&](int kstep) {
        const unsigned start = lds_wave + (kstep & 1) * Cfg::STAGE_BYTES;
        const bf16_t* ab = A + kstep * Cfg::BK;
        const bf16_t* wb = W + kstep * Cfg::BK;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %[keep], m0\n\t"
            "s_mov_b32 m0, %[start]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[a0], %[ab]\n\t"
            "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[a1], %[ab]\n\t"
            "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[a2], %[ab]\n\t"
            "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[a3], %[ab]\n\t"
            "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[b0], %[wb]\n\t"
            "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[b1], %[wb]\n\t"
            "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[b2], %[wb]\n\t"
            "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[b3], %[wb]\n\t"
            "s_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep)
            : [start] "s"(start), [ab] "s"(ab), [wb] "s"(wb), [a0] "v"(aoff[0]), [a1] "v"(aoff[1]), [a2] "v"(aoff[2]),
              [a3] "v"(aoff[Cfg::A_LOADS - 1]), [b0] "v"(boff[0]), [b1] "v"(boff[1]), [b2] "v"(boff[2]),
              [b3] "v"(boff[Cfg::B_LOADS - 1])
            : "memory", "scc");
    };
    if (SB) stage_sb(0);
    else stage_tile<Cfg>(A, p.lda, W, p.ldw, m0, n0, p.M, p.N, 0, smem, tid, wave);

    // per-lane fragment addressing (swizzled chunk for ks = 0, 1)
    const int frow = lane & 15;
    const int sw = (lane >> 1) & 7;
    const int pc0 = (((lane >> 4)) ^ sw) << 4;
    const int pc1 = ((4 + (lane >> 4)) ^ sw) << 4;
    const int a_row_off = (wm * Cfg::WTM + frow) * 128;
    const int b_row_off = Cfg::A_BYTES + (wn * Cfg::WTN + frow) * 128;

    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // tile kt has landed for every wave; every wave is done reading the other buffer
        char* cur = smem + (kt & 1) * Cfg::STAGE_BYTES;
        if (kt + 1 < nk) {
            if (SB) stage_sb(kt + 1);
            else stage_tile<Cfg>(A, p.lda, W, p.ldw, m0, n0, p.M, p.N, (kt + 1) * Cfg::BK,
                                 smem + ((kt + 1) & 1) * Cfg::STAGE_BYTES, tid, wave);
        }
        // all fragment reads of this K-step first (2 k-substeps x (MI + NI) ds_read_b128), then the MFMA chain:
        // the LDS latency is paid once per K-step instead of once per group of MFMAs
        bf16x8 af[2][MI], bfr[2][NI];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pc = ks ? pc1 : pc0;
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[ks][j] = *(const bf16x8*)(cur + b_row_off + j * 16 * 128 + pc);
#pragma unroll
            for (int i = 0; i < MI; ++i) af[ks][i] = *(const bf16x8*)(cur + a_row_off + i * 16 * 128 + pc);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + (lane>>4)*4 + 0..3] ----
    // As in the ping-pong kernel below: what a chunk of (at most 4) row blocks READS -- residual, hint, gate -- is loaded as one
    // batch before the chunk's first store (C may alias the residual, so interleaved loads would each wait for the store in front
    // of them); the bias depends on the column only and is loaded once.  Same arithmetic per element.
    const bf16_t* bias = (const bf16_t*)(grp == 0 ? p.bias : p.biasg[grp - 1]);
    const bf16_t* resid = (const bf16_t*)p.resid;
    const bf16_t* gate = (const bf16_t*)p.gate;
    const bf16_t* hint = (const bf16_t*)p.hint;
    bf16_t* C = (bf16_t*)(grp == 0 ? p.C : p.Cg[grp - 1]);
    const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
    constexpr bool NEED_R = EPI == VC_EPI_BIAS_RESID || EPI == VC_EPI_GELU_MUL || EPI == VC_EPI_BIAS_GATE_RESID;
    constexpr int CH = MI < 4 ? MI : 4;                      // row blocks per chunk
    const int nb = n0 + wn * Cfg::WTN + (lane >> 4) * 4;
    uint2 bbp[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) bbp[j] = (bias && nb + j * 16 < p.N) ? *(const uint2*)(bias + nb + j * 16) : uint2{0u, 0u};
#pragma unroll
    for (int i0 = 0; i0 < MI; i0 += CH) {
        uint2 rr[CH * NI], hh[CH * NI], gq[CH * NI];
        if (NEED_R) {
#pragma unroll
            for (int ii = 0; ii < CH; ++ii) {
                const int m = m0 + wm * Cfg::WTM + (i0 + ii) * 16 + (lane & 15);
                if (m < p.M) {
                    const int b = m / rpb;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int n = nb + j * 16;
                        if (n < p.N) {
                            rr[ii * NI + j] = *(const uint2*)(resid + (int64_t)m * p.ldr + n);
                            if (EPI == VC_EPI_BIAS_GATE_RESID) {
                                gq[ii * NI + j] = *(const uint2*)(gate + (int64_t)b * p.gate_bstride + n);
                                if (hint) hh[ii * NI + j] = *(const uint2*)(hint + (int64_t)m * p.ldh + n);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ii = 0; ii < CH; ++ii) {
            const int i = i0 + ii;
            const int m = m0 + wm * Cfg::WTM + i * 16 + (lane & 15);
            if (m >= p.M) continue;
            const int b = m / rpb;
            const bool dead = p.valid_rows >= 0 && (m - b * rpb) >= p.valid_rows;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = nb + j * 16;
                if (n >= p.N) continue;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (bias) {
                    float bb[4];
                    unpack4(bbp[j], bb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bb[e];
                }
                if (EPI == VC_EPI_BIAS_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(round_bf16(v[e]));
                } else if (EPI == VC_EPI_BIAS_RESID) {
                    float r[4];
                    unpack4(rr[ii * NI + j], r);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = r[e] + round_bf16(v[e]);
                } else if (EPI == VC_EPI_GELU_MUL) {
                    float r[4];
                    unpack4(rr[ii * NI + j], r);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = round_bf16(gelu_tanh_f(round_bf16(v[e]))) * r[e];
                } else if (EPI == VC_EPI_BIAS_GATE_RESID) {
                    float r[4], g[4];
                    unpack4(rr[ii * NI + j], r);
                    unpack4(gq[ii * NI + j], g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = r[e] + round_bf16(round_bf16(v[e]) * g[e]);
                    if (hint) {
                        float h[4];
                        unpack4(hh[ii * NI + j], h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = round_bf16(v[e]) + round_bf16(h[e] * p.hint_scale);
                    }
                }
                if (dead) v[0] = v[1] = v[2] = v[3] = 0.f;
                *(uint2*)(C + (int64_t)m * p.ldc + n) = pack4(v);
            }
        }
    }
}

// =====================================================================================================
// Ping-pong kernel (production kernel for the engine's large GEMMs): 256x256x64 tile, 8 waves (2 x 4, 128x64 each),
// 2 LDS stages of 64 KiB.  The two waves of a SIMD (wave w and w+4: groups G0 = rows 0-127, G1 = rows 128-255) run the
// same program one barrier apart, so that at any time one of them is in a 16-MFMA section (one quadrant of its output
// x K=64) and the other in a "load" section (4-8 ds_read_b128 fragment reads + the 2 LDS-DMA pieces that are its share
// of one half-tile of a later K-tile).  Eight phases = two K-tiles per loop trip; per phase
//     reads ; DMA ; [vmcnt] ; barrier ; MFMA x16 ; barrier
// K-tile image in LDS (per stage): [A half0 | A half1 | B half0 | B half1], 16 KiB each, where half h of A holds, for
// both wave groups, the 64 rows each wave uses in its quadrants h (and likewise 32-row slices of B for the 4 wave
// columns): a half is dead once the phase that reads it has passed, which is what lets its successor (K-tile + 2) be
// staged while the K-tile is still being computed:
//     phase 1: reads A0 of E      stages Ah1 of O      phase 5: reads A0 of O      stages Ah1 of E'
//     phase 2: reads B1 of E      stages Bh0 of E'     phase 6: reads B1 of O      stages Bh0 of O'
//     phase 3: reads A1 of E      stages Ah0 of E'     phase 7: reads A1 of O      stages Ah0 of O'
//              vmcnt(4): O landed                               vmcnt(4): E' landed
//     phase 4: reads B0 of O      stages Bh1 of E'     phase 8: reads B0 of E'     stages Bh1 of O'
// (E/O = K-tile in the even/odd stage, ' = two K-tiles later).  The B0 fragments of a K-tile are read during the last
// phase of the previous one, into the register set that held its B1 (dead after phase 3): 8/4/8/4 reads per phase.
// WAR: a half is restaged two phases after its last read.  RAW: the counted vmcnt (all but the two youngest half-tiles)
// sits before a barrier that every reader passes before its first read of that K-tile, one phase later.
// Rows past M are read (never stored): the caller guarantees they are readable (a_rows_padded / M % 256 == 0).
// The epilogue kind is a template parameter (here and in the other kernels): with all five kinds inline behind run-time
// branches this kernel was 52 KB of code, most of it unrolled epilogue, against a 64 KB instruction cache shared by two
// CUs; per kind it is 11-24 KB (round 2: 0-2 % at the cfg-3 shapes, same-box A/B).
// =====================================================================================================
// FP8 (BASELINE config 5): the same kernel on OCP e4m3 operands.  A 128-byte LDS row then holds K = 128 elements instead of 64, and ONE
// v_mfma_scale_f32_16x16x128_f8f6f4 (32 bytes of each operand per lane: the 16-byte chunks 2g, 2g+1 of lane group g; twice the cycles of
// a bf16 16x16x32) replaces the two bf16 MFMAs of a row: identical bytes, DMA pieces, phases and matrix-pipe cycles per K-tile at twice
// the multiply-adds.  Block scales are 1 (e8m0 127); the per-token and per-channel scales are applied to the fp32 accumulator.
template <int EPI, bool HINT = false, bool FP8 = false>      // HINT: the gated-residual epilogue also adds a hint (VC.py:146-147)
__global__ __launch_bounds__(512) void gemm_pp_kernel(VcGemmParams p, int nTm, int nTn, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int per_xcd = gridDim.x >> 3;
    const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (id >= ntiles) return;
    constexpr int GROUP_M = 8;
    const int width = GROUP_M * nTn;
    const int group = id / width;
    const int first_m = group * GROUP_M;
    const int gsz = min(nTm - first_m, GROUP_M);
    int tm = first_m + (id % width) % gsz;
    int tn = (id % width) / gsz;
    if (p.tile_map == 1) {
        // A/B of the round-3 review's proposal (tile id 6, tools only; same tiles, same arithmetic, bit-identical C): the eight XCDs walk
        // the SAME super-band of 64 tile rows -- XCD x takes rows 64 s + 8 x .. + 7 of super-band s, all tile columns -- instead of eight
        // bands 32 tile rows apart (nTm % 64 == 0: the launcher checks)
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int sb = j / width, q = j - sb * width;
        tm = 64 * sb + 8 * x + (q & 7);
        tn = q >> 3;
    }
    const int nTn1 = p.ngroups > 1 ? nTn / p.ngroups : nTn;
    const int grp = tn / nTn1;
    tn -= grp * nTn1;
    const int m0 = tm * 256, n0 = tn * 256;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
#ifdef VC_PP_TRACE      // tools/trace_gemm.py: per-workgroup timestamps (100 MHz) of entry / first MFMA / loop end / stores issued
    auto now = [] { uint64_t t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; };
    uint64_t* trc = vc_pp_trace_buf ? vc_pp_trace_buf + (size_t)blockIdx.x * 8 : nullptr;
    if (trc && tid == 0) {
        trc[0] = now();
        trc[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // HW_ID
        trc[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));      // XCC_ID
        trc[6] = id;
    }
#endif

    auto uniform_ptr = [](const char* q) {     // force a wave-uniform address into an SGPR pair
        const uint64_t u = (uint64_t)q;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return (const char*)(((uint64_t)hi << 32) | lo);
    };
    const int64_t lda2 = p.lda * (FP8 ? 1 : 2), ldw2 = p.ldw * (FP8 ? 1 : 2);      // row pitches in bytes
    // staging: per half-tile a wave moves pieces `wave` and `wave + 8` (8 LDS rows of 128 B each); both have the
    // parity of `wave`, so the lane's swizzled chunk is one constant
    const int r8 = lane >> 3;
    const int chunk = (lane & 7) ^ ((r8 >> 1) + 4 * (wave & 1));
    const unsigned voff_a = (unsigned)(r8 * lda2 + chunk * 16);
    const unsigned voff_w = (unsigned)(r8 * ldw2 + chunk * 16);
    // first source row of piece `wave` in half 0:  A: 8*wave (group 0; piece wave+8 = group 1, +128 rows)
    //                                              W: wave column wave>>2, 8*(wave&3) (piece wave+8: column +2, +128 rows)
    const char* a_src = uniform_ptr((const char*)p.A + ((int64_t)m0 + 8 * wave) * lda2);
    const char* w_src = uniform_ptr((const char*)(grp == 0 ? p.W : p.Wg[grp - 1]) +
                                    ((int64_t)n0 + (wave >> 2) * 64 + (wave & 3) * 8) * ldw2);
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_char_t*)smem + wave * 1024);
    const int nk = p.K >> (FP8 ? 7 : 6);                   // K-tiles of 128 bytes per row

    // region: 0 Ah0, 1 Ah1, 2 Bh0, 3 Bh1
    auto stage_half = [&](int stage, int region, int kt) {
        const bool isA = region < 2;
        const int hh = region & 1;
        const char* s0 = isA ? a_src + (int64_t)hh * 64 * lda2 : w_src + (int64_t)hh * 32 * ldw2;
        s0 += (int64_t)kt * 128;
        const char* s1 = s0 + 128 * (isA ? lda2 : ldw2);
        // readfirstlane: in the FP8 instantiations hipcc (ROCm 7.2) hands this "s" operand over in a VGPR copy of the SGPR otherwise
        const unsigned d0 = FP8 ? __builtin_amdgcn_readfirstlane(lds_wave + stage * 65536 + region * 16384) : lds_wave + stage * 65536 + region * 16384;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %[keep], m0\n\t"
            "s_mov_b32 m0, %[d0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v], %[s0]\n\t"
            "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v], %[s1]\n\t"
            "s_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep)
            : [d0] "s"(d0), [v] "v"(isA ? voff_a : voff_w), [s0] "s"(s0), [s1] "s"(s1)
            : "memory", "scc");
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment reads: LDS row (within a half) of A = wr*64 + i*16 + frow, of B = wc*32 + j*16 + frow
    const int frow = lane & 15;
    const int sw = (lane >> 1) & 7;
    // bf16: k-step ks reads chunk 4 ks + g; fp8: the lane's 32 consecutive bytes are the chunks 2g, 2g + 1 (g = lane >> 4)
    const int pc0 = ((FP8 ? 2 * (lane >> 4) : (lane >> 4)) ^ sw) << 4, pc1 = ((FP8 ? 2 * (lane >> 4) + 1 : 4 + (lane >> 4)) ^ sw) << 4;
    const char* a_rd = smem + (wr * 64 + frow) * 128;
    const char* b_rd = smem + 32768 + (wc * 32 + frow) * 128;
    bf16x8 af[4][2], bf[2][2][2];       // af[i][ks] (half in use), bf[half][j][ks]
    i32x8 a8[4], b8[2][2];              // the fp8 form: one 32-byte fragment per (i) / (half, j)
    auto read_a = [&](int stage, int hh) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (FP8) {
                const i32x4 lo = *(const i32x4*)(a_rd + stage * 65536 + hh * 16384 + i * 2048 + pc0);
                const i32x4 hi = *(const i32x4*)(a_rd + stage * 65536 + hh * 16384 + i * 2048 + pc1);
                a8[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
                af[i][0] = *(const bf16x8*)(a_rd + stage * 65536 + hh * 16384 + i * 2048 + pc0);
                af[i][1] = *(const bf16x8*)(a_rd + stage * 65536 + hh * 16384 + i * 2048 + pc1);
            }
        }
    };
    // B fragments of half hh of the K-tile in `stage` live in register set hh ^ stage: the set that held B1 of one K-tile
    // is dead after that K-tile's third phase and receives B0 of the NEXT K-tile during its fourth (see the loop)
    auto read_b = [&](int stage, int hh) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (FP8) {
                const i32x4 lo = *(const i32x4*)(b_rd + stage * 65536 + hh * 16384 + j * 2048 + pc0);
                const i32x4 hi = *(const i32x4*)(b_rd + stage * 65536 + hh * 16384 + j * 2048 + pc1);
                b8[hh ^ stage][j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
                bf[hh ^ stage][j][0] = *(const bf16x8*)(b_rd + stage * 65536 + hh * 16384 + j * 2048 + pc0);
                bf[hh ^ stage][j][1] = *(const bf16x8*)(b_rd + stage * 65536 + hh * 16384 + j * 2048 + pc1);
            }
        }
    };
    const int unit_scale = 0x7F7F7F7F;           // e8m0 127 = 1.0 for every 32-element block (fp8 form)
    auto mma = [&](int ah, int bh, int stage) {   // quadrant (A half ah, B half bh) x K=64: 16 MFMA
        const int bs = bh ^ stage;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    // asm with a tied accumulator: through the builtin hipcc (ROCm 7.2) leaves the scaled MFMA untied (vdst != srcC for
                    // part of them), the accumulators migrate between tuples and 110-350 VGPRs spill into the loop.  Operands come from
                    // ds_read (the waitcnt pass covers asm inputs) and the accumulators are next read in the epilogue.
                    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                                 : "+v"(acc[ah * 4 + i][bh * 2 + j])
                                 : "v"(b8[bs][j]), "v"(a8[i]), "v"(unit_scale));
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ah * 4 + i][bh * 2 + j] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[bs][j][ks], af[i][ks], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
#define VC_PP_BARRIER()  do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VC_PP_WAIT(str)  do { __builtin_amdgcn_sched_barrier(0); asm volatile(str ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

    // ---- prologue: K-tile 0 complete, K-tile 1 minus its Ah1 in flight ----
    stage_half(0, 2, 0); stage_half(0, 0, 0); stage_half(0, 3, 0); stage_half(0, 1, 0);
    stage_half(1, 2, 1); stage_half(1, 0, 1); stage_half(1, 3, 1);
    VC_PP_WAIT("s_waitcnt vmcnt(6)");
    VC_PP_BARRIER();
    if (wr == 1) VC_PP_BARRIER();          // group 1 runs one barrier behind group 0
    read_b(0, 0);                          // B0 of K-tile 0 (later ones are read during the previous K-tile's 4th phase)
    __builtin_amdgcn_sched_barrier(0);

    // One pair of K-tiles (8 phases).  Loads issued during the pair fill K-tiles kt+1 (its Ah1), kt+2 and kt+3 (minus its Ah1);
    // the LAST pair of the tile stages nothing past K-tile nk-1 and waits for everything instead of counting.
    auto pair = [&](int kt, auto last_c) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_c)::value;
        auto stage_ahead = [&](int stage, int region, int off) __attribute__((always_inline)) {
            if (!(LAST && off >= 2)) stage_half(stage, region, kt + off);
        };
#pragma unroll
        for (int st = 0; st < 2; ++st) {     // st = 0: K-tile kt (even stage), st = 1: K-tile kt+1 (odd stage)
            // phase 1 / 5: reads A0
            read_a(st, 0);
            __builtin_amdgcn_sched_barrier(0);
            stage_ahead(st ^ 1, 1, 1 + st);                     // Ah1 of the other stage's next K-tile
            VC_PP_BARRIER();
            mma(0, 0, st);
            VC_PP_BARRIER();
            // phase 2 / 6: reads B1
            read_b(st, 1);
            __builtin_amdgcn_sched_barrier(0);
            stage_ahead(st, 2, 2 + st);                         // Bh0 of this stage's next K-tile
            VC_PP_BARRIER();
            mma(0, 1, st);
            VC_PP_BARRIER();
            // phase 3 / 7: reads A1; the other stage's K-tile has landed (this wave's pieces) once all but the two
            // youngest half-tiles are retired
            read_a(st, 1);
            __builtin_amdgcn_sched_barrier(0);
            stage_ahead(st, 0, 2 + st);                         // Ah0
            if (!LAST) VC_PP_WAIT("s_waitcnt vmcnt(4)");
            else if (st == 0) VC_PP_WAIT("s_waitcnt vmcnt(0)");  // nothing younger in flight: the last K-tile has landed
            VC_PP_BARRIER();
            mma(1, 1, st);
            VC_PP_BARRIER();
            // phase 4 / 8: reads B0 of the NEXT K-tile (other stage; its set of registers held this K-tile's B1)
            read_b(st ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            stage_ahead(st, 3, 2 + st);                         // Bh1
            VC_PP_BARRIER();
            mma(1, 0, st);
            VC_PP_BARRIER();
        }
    };
#ifdef VC_PP_TRACE
    if (trc && tid == 0) trc[1] = now();
#endif
    for (int kt = 0; kt < nk - 2; kt += 2) pair(kt, std::false_type{});
    pair(nk - 2, std::true_type{});
    if (wr == 0) VC_PP_BARRIER();          // equalise the barrier count
    if constexpr (FP8) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // asm MFMAs: no compiler-managed wait states before the accumulators are read
    VC_PP_WAIT("s_waitcnt vmcnt(0)");      // nothing may still be landing in LDS when the workgroup retires
#undef VC_PP_BARRIER
#undef VC_PP_WAIT
#ifdef VC_PP_TRACE
    if (trc && tid == 0) trc[2] = now();
#endif

    // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + (lane>>4)*4 + 0..3] ----
    // Everything the epilogue READS is fetched before its first store: the bias and the gate of the tile's (at most two) samples
    // once per wave (they do not depend on the row), the residual (and hint) rows of a whole pass as one batch of loads.  C may
    // alias the residual (the engine adds in place), so with loads and stores interleaved fragment by fragment every load had
    // to wait for the store before it: 32 memory round trips in a row, 27.6 us per tile for the gated-residual epilogue against
    // 4.2 us for a plain store (tools/trace_gemm.py, round 2).  Same arithmetic per element, same results.  The launcher
    // guarantees 16-byte aligned rows of C and rows_per_batch >= 256 (a tile of 256 rows then touches at most two samples).
    const bf16_t* bias = (const bf16_t*)(grp == 0 ? p.bias : p.biasg[grp - 1]);
    const bf16_t* resid = (const bf16_t*)p.resid;
    const bf16_t* gate = (const bf16_t*)p.gate;
    const bf16_t* hint = (const bf16_t*)p.hint;
    bf16_t* C = (bf16_t*)(grp == 0 ? p.C : p.Cg[grp - 1]);
    const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
    const int b_first = m0 / rpb;                                         // wave-uniform (scalar) division, once per tile
    const int m_next = (b_first + 1) * rpb;                               // first row of the next sample
    const int b_last = (p.M - 1) / rpb;
    constexpr bool NEED_R = EPI == VC_EPI_BIAS_RESID || EPI == VC_EPI_GELU_MUL || EPI == VC_EPI_BIAS_GATE_RESID;
    const int nb = n0 + wc * 64 + (lane >> 4) * 4;                        // this lane's first column in block j = 0
    const int g16 = lane >> 4;
    const int nst = n0 + wc * 64 + (g16 & 1) * 16 + (g16 >> 1) * 8;       // first column of this lane's 16-byte store in pair jp = 0
    uint2 bbp[4], ggp[2][4];   // kept packed (bf16 x 4): the gated-residual form needs every register for its batch of loads
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bbp[j] = bias ? *(const uint2*)(bias + nb + j * 16) : uint2{0u, 0u};
        if (EPI == VC_EPI_BIAS_GATE_RESID) {
            ggp[0][j] = *(const uint2*)(gate + (int64_t)b_first * p.gate_bstride + nb + j * 16);
            ggp[1][j] = *(const uint2*)(gate + (int64_t)(b_first < b_last ? b_first + 1 : b_last) * p.gate_bstride + nb + j * 16);
        }
    }
    const float* wscale = FP8 ? (grp == 0 ? p.w_scale : p.w_scaleg[grp - 1]) + nb : nullptr;   // fp8: output-channel scales of this lane's columns
    // fp8: the sixteen channel scales of this lane's columns and (per pass, below) the token scales of its rows are read BEFORE the first
    // store, like everything else the epilogue reads: a load issued behind a store can only be waited for together with that store (the
    // vmcnt queue is in order), and with the scales re-read row by row that was eight store round trips in a row -- the K sweep read
    // 14.3 us per output tile for the fp8 kernel against 9-12 for the bf16 one (profiles/r04_gemm_fp8_ksweep.txt)
    f32x4 wsv[4];
    if constexpr (FP8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wsv[j] = *(const f32x4*)(wscale + j * 16);
    }
    uint2 rr[HINT ? 16 : 32], hh[16];      // residual fragments of a pass (32 without a hint, 16 + 16 hint fragments with one)
    auto pass = [&](auto i0_c, auto i1_c) __attribute__((always_inline)) {
        constexpr int I0 = decltype(i0_c)::value, I1 = decltype(i1_c)::value;
        float ascv[I1 - I0];
        if constexpr (FP8) {
#pragma unroll
            for (int i = I0; i < I1; ++i) {
                const int m = m0 + wr * 128 + i * 16 + (lane & 15);
                ascv[i - I0] = p.a_scale[m < p.M ? m : p.M - 1];
            }
            if (!NEED_R) __builtin_amdgcn_sched_barrier(0);
        }
        if (NEED_R) {
#pragma unroll
            for (int i = I0; i < I1; ++i) {
                const int m = m0 + wr * 128 + i * 16 + (lane & 15);
                if (m < p.M) {
                    // 16 bytes per lane in the layout of the stores below (64 contiguous bytes per row and instruction), turned
                    // back into the accumulator's fragment layout by the same half exchange between lane groups g and g ^ 1
                    const bf16_t* rrow = resid + (int64_t)m * p.ldr + nst;
                    const bf16_t* hrow = HINT ? hint + (int64_t)m * p.ldh + nst : nullptr;
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        const uint4 L = *(const uint4*)(rrow + jp * 32);
                        const auto x0 = __builtin_amdgcn_permlane16_swap(L.x, L.z, false, false);
                        const auto x1 = __builtin_amdgcn_permlane16_swap(L.y, L.w, false, false);
                        rr[(i - I0) * 4 + jp * 2] = uint2{x0[0], x1[0]};
                        rr[(i - I0) * 4 + jp * 2 + 1] = uint2{x0[1], x1[1]};
                        if (HINT) {
                            const uint4 Hh = *(const uint4*)(hrow + jp * 32);
                            const auto y0 = __builtin_amdgcn_permlane16_swap(Hh.x, Hh.z, false, false);
                            const auto y1 = __builtin_amdgcn_permlane16_swap(Hh.y, Hh.w, false, false);
                            hh[(i - I0) * 4 + jp * 2] = uint2{y0[0], y1[0]};
                            hh[(i - I0) * 4 + jp * 2 + 1] = uint2{y0[1], y1[1]};
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);         // no store of this pass before its last load has been issued
        }
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            const int m = m0 + wr * 128 + i * 16 + (lane & 15);
            if (m >= p.M) continue;
            const bool second = m >= m_next;                          // row of the tile's second sample
            const bool dead = p.valid_rows >= 0 && (m - (second ? m_next : b_first * rpb)) >= p.valid_rows;
            const unsigned keep = dead ? 0u : 0xFFFFFFFFu;            // rows past valid_rows are written as +0.0
            bf16_t* crow = C + (int64_t)m * p.ldc + nst;
            float asc = 1.f;
            if constexpr (FP8) asc = ascv[i - I0];
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                uint2 pk[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = jp * 2 + jj;
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    if constexpr (FP8) {
                        const f32x4 wsj = wsv[j];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= asc * wsj[e];
                    }
                    if (bias) {
                        float bb[4];
                        unpack4(bbp[j], bb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bb[e];
                    }
                    if (EPI == VC_EPI_BIAS_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(round_bf16(v[e]));
                    } else if (EPI == VC_EPI_BIAS_RESID) {
                        float r[4];
                        unpack4(rr[(i - I0) * 4 + j], r);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = r[e] + round_bf16(v[e]);
                    } else if (EPI == VC_EPI_GELU_MUL) {
                        float r[4];
                        unpack4(rr[(i - I0) * 4 + j], r);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = round_bf16(gelu_tanh_f(round_bf16(v[e]))) * r[e];
                    } else if (EPI == VC_EPI_BIAS_GATE_RESID) {
                        float r[4], g4[4];
                        unpack4(rr[(i - I0) * 4 + j], r);
                        const uint2 gsel = uint2{second ? ggp[1][j].x : ggp[0][j].x, second ? ggp[1][j].y : ggp[0][j].y};
                        unpack4(gsel, g4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = r[e] + round_bf16(round_bf16(v[e]) * g4[e]);
                        if (HINT) {
                            float hv[4];
                            unpack4(hh[(i - I0) * 4 + j], hv);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = round_bf16(v[e]) + round_bf16(hv[e] * p.hint_scale);
                        }
                    }
                    pk[jj] = pack4(v);
                    pk[jj].x &= keep;
                    pk[jj].y &= keep;
                }
                // two neighbouring 16-column blocks -> one 16-byte store: the lane groups (rows of 16 lanes) g and g^1 trade halves
                // so that even groups hold 8 consecutive columns of block 2jp, odd groups of block 2jp+1 (v_permlane16_swap: odd
                // rows of the first operand <-> even rows of the second); a store covers 64 contiguous bytes per row instead of 32
                const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
                *(uint4*)(crow + jp * 32) = uint4{s0[0], s1[0], s0[1], s1[1]};
            }
        }
    };
    using I0_ = std::integral_constant<int, 0>;
    using I4_ = std::integral_constant<int, 4>;
    using I8_ = std::integral_constant<int, 8>;
    if (HINT) {
        pass(I0_{}, I4_{});
        pass(I4_{}, I8_{});
    } else {
        pass(I0_{}, I8_{});
    }
#ifdef VC_PP_TRACE
    if (trc && tid == 0) {
        trc[3] = now();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        trc[7] = now();                                            // this wave's stores acknowledged
    }
#endif
}

template <int EPI, bool HINT = false, bool FP8 = false>
int launch_pp_e(const VcGemmParams& p, hipStream_t stream) {
    constexpr int LDS = 2 * 65536;
    static std::atomic<uint64_t> attr_done{0};
    if (!vc_set_lds_once(attr_done, (const void*)gemm_pp_kernel<EPI, HINT, FP8>, LDS)) return VC_E_HIP;
    const int ng = p.ngroups > 1 ? p.ngroups : 1;
    const int nTm = (p.M + 255) / 256, nTn = ng * (p.N / 256);
    const int ntiles = nTm * nTn;
    const int grid = (ntiles + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_pp_kernel<EPI, HINT, FP8>), dim3(grid), dim3(512), LDS, stream, p, nTm, nTn, ntiles);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}
int launch_pp_fp8(const VcGemmParams& p, hipStream_t stream) {
    switch (p.epilogue) {
        case VC_EPI_BIAS: return launch_pp_e<VC_EPI_BIAS, false, true>(p, stream);
        case VC_EPI_BIAS_GELU: return launch_pp_e<VC_EPI_BIAS_GELU, false, true>(p, stream);
        case VC_EPI_BIAS_RESID: return launch_pp_e<VC_EPI_BIAS_RESID, false, true>(p, stream);
        case VC_EPI_BIAS_GATE_RESID:
            return p.hint ? launch_pp_e<VC_EPI_BIAS_GATE_RESID, true, true>(p, stream) : launch_pp_e<VC_EPI_BIAS_GATE_RESID, false, true>(p, stream);
    }
    return VC_E_UNSUPPORTED;
}
int launch_pp(const VcGemmParams& p, hipStream_t stream) {
    switch (p.epilogue) {
        case VC_EPI_BIAS: return launch_pp_e<VC_EPI_BIAS>(p, stream);
        case VC_EPI_BIAS_GELU: return launch_pp_e<VC_EPI_BIAS_GELU>(p, stream);
        case VC_EPI_BIAS_RESID: return launch_pp_e<VC_EPI_BIAS_RESID>(p, stream);
        case VC_EPI_BIAS_GATE_RESID:
            return p.hint ? launch_pp_e<VC_EPI_BIAS_GATE_RESID, true>(p, stream) : launch_pp_e<VC_EPI_BIAS_GATE_RESID>(p, stream);
        case VC_EPI_GELU_MUL: return launch_pp_e<VC_EPI_GELU_MUL>(p, stream);
    }
    return VC_E_INVALID;
}

// =====================================================================================================
// One-wave-per-SIMD kernel (round 2 experiment, tile id 5): 256x256x64 tile, 4 waves (2 x 2, 128x128 each), accumulators
// in the 256 AGPRs, 2 LDS stages of 64 KiB, ONE barrier per K-tile.  Where the ping-pong kernel alternates whole 16-MFMA
// sections with load sections across the two waves of a SIMD (16 barriers per two K-tiles), here a single wave interleaves
// its own fragment reads and DMA issue between MFMAs (1 read per 4 MFMA): the 128x128 wave tile reads 32 KiB of fragments
// per K-tile instead of 2 x 24 KiB, and nothing but the barrier ever idles the matrix pipe.
//   K-tile kt (stage s = kt & 1), two steps of 64 MFMA (k = 0..31, 32..63):
//     step 0: MFMA on F0                     | reads F1 <- (kt, k 32..63) from stage s
//             vmcnt(0): K-tile kt+1 landed (issued one K-tile ago) ; barrier   (every wave is past its reads of stage s)
//     step 1: MFMA on F1 | DMA K-tile kt+2 -> stage s | reads F0 <- (kt+1, k 0..31) from stage s^1
// Same LDS image per row as the ping-pong kernel (128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7), stage =
// [A rows 0..255 | B rows 0..255]; same K order per accumulator -> bit-identical results.
// =====================================================================================================
template <int EPI>
__global__ __launch_bounds__(256) void gemm_sw_kernel(VcGemmParams p, int nTm, int nTn, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int per_xcd = gridDim.x >> 3;
    const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (id >= ntiles) return;
    constexpr int GROUP_M = 8;
    const int width = GROUP_M * nTn;
    const int group = id / width;
    const int first_m = group * GROUP_M;
    const int gsz = min(nTm - first_m, GROUP_M);
    const int tm = first_m + (id % width) % gsz;
    int tn = (id % width) / gsz;
    const int nTn1 = p.ngroups > 1 ? nTn / p.ngroups : nTn;
    const int grp = tn / nTn1;
    tn -= grp * nTn1;
    const int m0 = tm * 256, n0 = tn * 256;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    auto uniform_ptr = [](const char* q) {
        const uint64_t u = (uint64_t)q;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return (const char*)(((uint64_t)hi << 32) | lo);
    };
    const int64_t lda2 = p.lda * 2, ldw2 = p.ldw * 2;
    // staging: a K-tile is 64 pieces of 8 rows (32 of A, 32 of B); wave w moves pieces w, w+4, ... of each (8 + 8 per K-tile);
    // all have the parity of w, so the lane's swizzled chunk is one constant
    const int r8 = lane >> 3;
    const int chunk = (lane & 7) ^ ((r8 >> 1) + 4 * (wave & 1));
    const unsigned voff_a = (unsigned)(r8 * lda2 + chunk * 16);
    const unsigned voff_w = (unsigned)(r8 * ldw2 + chunk * 16);
    unsigned voff_a_t[8], voff_w_t[8];      // piece t of a K-tile: 32 rows further down (lane offsets: one scalar base per K-tile)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        voff_a_t[t] = voff_a + (unsigned)(t * 32 * lda2);
        voff_w_t[t] = voff_w + (unsigned)(t * 32 * ldw2);
    }
    const char* a_src = uniform_ptr((const char*)p.A + ((int64_t)m0 + 8 * wave) * lda2);
    const char* w_src = uniform_ptr((const char*)(grp == 0 ? p.W : p.Wg[grp - 1]) + ((int64_t)n0 + 8 * wave) * ldw2);
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_char_t*)smem + wave * 1024);
    const int nk = p.K >> 6;

    // piece t (0..7) of A (isA) or B of K-tile kt into `stage`
    auto dma = [&](int stage, bool isA, int t, int kt) {
        const char* s0 = (isA ? a_src + (int64_t)t * 32 * lda2 : w_src + (int64_t)t * 32 * ldw2) + (int64_t)kt * 128;
        const unsigned d0 = lds_wave + stage * 65536 + (isA ? 0 : 32768) + t * 4096;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %[keep], m0\n\t"
            "s_mov_b32 m0, %[d0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v], %[s0]\n\t"
            "s_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep)
            : [d0] "s"(d0), [v] "v"(isA ? voff_a : voff_w), [s0] "s"(s0)
            : "memory", "scc");
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int sw = (lane >> 1) & 7;
    const int pc[2] = {((lane >> 4) ^ sw) << 4, ((4 + (lane >> 4)) ^ sw) << 4};
    const char* a_rd = smem + (wr * 128 + frow) * 128;
    const char* b_rd = smem + 32768 + (wc * 128 + frow) * 128;
    bf16x8 fa[2][8], fb[2][8];              // fragment sets F0 / F1: fa[set][i], fb[set][j]
    // fragment number f (0..15) of K-half ks of the K-tile in `stage` into set `set`: f < 8 -> B block f, else A block f-8
    // (the order the next step's MFMAs first need them in: all of B and A block 0 within its first 8 MFMAs)
    auto read_frag = [&](int set, int stage, int ks, int f) {
        if (f < 8) fb[set][f] = *(const bf16x8*)(b_rd + stage * 65536 + f * 2048 + pc[ks]);
        else       fa[set][f - 8] = *(const bf16x8*)(a_rd + stage * 65536 + (f - 8) * 2048 + pc[ks]);
    };
    // MFMA number q (0..63) of a step, ordered so that consecutive MFMAs share neither accumulator nor (where possible) wait
    // on the same late fragment: row block i = q / 8, column block j = q % 8
    auto mma1 = [&](int set, int q) {
        const int i = q >> 3, j = q & 7;
        // accumulator tied to itself in AGPRs by hand: with all 256 in use the register allocator otherwise routes every
        // MFMA's C operand through a scratch tuple (4 v_accvgpr_mov per MFMA)
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fb[set][j]), "v"(fa[set][i]));
    };
#define VC_SW_FENCE() __builtin_amdgcn_sched_barrier(0)

    // ---- prologue: K-tiles 0 and 1 in flight, F0 <- (0, k 0..31) ----
#pragma unroll
    for (int t = 0; t < 8; ++t) { dma(0, true, t, 0); dma(0, false, t, 0); }
    if (nk > 1) {
#pragma unroll
        for (int t = 0; t < 8; ++t) { dma(1, true, t, 1); dma(1, false, t, 1); }
        VC_SW_FENCE(); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); VC_SW_FENCE();
    } else {
        VC_SW_FENCE(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); VC_SW_FENCE();
    }
    __builtin_amdgcn_s_barrier();
    VC_SW_FENCE();
#pragma unroll
    for (int f = 0; f < 16; ++f) read_frag(0, 0, 0, f);

    auto ktile = [&](int kt, auto more_c) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more_c)::value;
        const int s = kt & 1;
        // ---- step 0: MFMA on F0; F1 <- second K-half of this K-tile ----
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            read_frag(1, s, 1, g);
            VC_SW_FENCE();
#pragma unroll
            for (int q = 0; q < 4; ++q) mma1(0, g * 4 + q);
            VC_SW_FENCE();
        }
        // K-tile kt+1 (this wave's pieces) landed; after the barrier: everyone's pieces, and nobody reads stage s any more
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VC_SW_FENCE();
        __builtin_amdgcn_s_barrier();
        VC_SW_FENCE();
        // ---- step 1: MFMA on F1; K-tile kt+2 -> stage s; F0 <- first K-half of K-tile kt+1 ----
        // the 16 pieces of K-tile kt+2 land at consecutive 4 KiB steps of this wave's LDS window (8 of A, then 8 of B): M0 is
        // set once and stepped, and each load sits between two MFMAs (the M0 write needs a wait state before its use)
        const char* a_k = a_src + (int64_t)(kt + 2) * 128;
        const char* w_k = w_src + (int64_t)(kt + 2) * 128;
        if (MORE) asm volatile("s_mov_b32 m0, %0" :: "s"(lds_wave + s * 65536 - 4096) : "memory");
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            read_frag(0, s ^ 1, 0, g);
            VC_SW_FENCE();
            if (MORE) {
                const int q0 = g * 4;
                asm volatile(
                    "s_add_u32 m0, m0, 0x1000\n\t"
                    "v_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\t"
                    "global_load_lds_dwordx4 %12, %13\n\t"
                    "v_mfma_f32_16x16x32_bf16 %1, %5, %9, %1\n\t"
                    "v_mfma_f32_16x16x32_bf16 %2, %6, %10, %2\n\t"
                    "v_mfma_f32_16x16x32_bf16 %3, %7, %11, %3"
                    : "+a"(acc[(q0 + 0) >> 3][(q0 + 0) & 7]), "+a"(acc[(q0 + 1) >> 3][(q0 + 1) & 7]),
                      "+a"(acc[(q0 + 2) >> 3][(q0 + 2) & 7]), "+a"(acc[(q0 + 3) >> 3][(q0 + 3) & 7])
                    : "v"(fb[1][(q0 + 0) & 7]), "v"(fb[1][(q0 + 1) & 7]), "v"(fb[1][(q0 + 2) & 7]), "v"(fb[1][(q0 + 3) & 7]),
                      "v"(fa[1][(q0 + 0) >> 3]), "v"(fa[1][(q0 + 1) >> 3]), "v"(fa[1][(q0 + 2) >> 3]), "v"(fa[1][(q0 + 3) >> 3]),
                      "v"(g < 8 ? voff_a_t[g & 7] : voff_w_t[g & 7]), "s"(g < 8 ? a_k : w_k)
                    : "memory", "scc");
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) mma1(1, g * 4 + q);
            }
            VC_SW_FENCE();
        }
    };
    for (int kt = 0; kt < nk - 2; ++kt) ktile(kt, std::true_type{});
    if (nk > 1) ktile(nk - 2, std::false_type{});
    ktile(nk - 1, std::false_type{});
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the hand-issued MFMAs' results are read by VALU code below
#undef VC_SW_FENCE

    // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + (lane>>4)*4 + 0..3] ----
    const bf16_t* bias = (const bf16_t*)(grp == 0 ? p.bias : p.biasg[grp - 1]);
    const bf16_t* resid = (const bf16_t*)p.resid;
    const bf16_t* gate = (const bf16_t*)p.gate;
    const bf16_t* hint = (const bf16_t*)p.hint;
    bf16_t* C = (bf16_t*)(grp == 0 ? p.C : p.Cg[grp - 1]);
    const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const int b = m / rpb;
        const bool dead = p.valid_rows >= 0 && (m - b * rpb) >= p.valid_rows;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wc * 128 + j * 16 + (lane >> 4) * 4;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (bias) {
                float bb[4];
                unpack4(*(const uint2*)(bias + n), bb);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bb[e];
            }
            if (EPI == VC_EPI_BIAS_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(round_bf16(v[e]));
            } else if (EPI == VC_EPI_BIAS_RESID) {
                float r[4];
                unpack4(*(const uint2*)(resid + (int64_t)m * p.ldr + n), r);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = r[e] + round_bf16(v[e]);
            } else if (EPI == VC_EPI_GELU_MUL) {
                float r[4];
                unpack4(*(const uint2*)(resid + (int64_t)m * p.ldr + n), r);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = round_bf16(gelu_tanh_f(round_bf16(v[e]))) * r[e];
            } else if (EPI == VC_EPI_BIAS_GATE_RESID) {
                float r[4], gg[4];
                unpack4(*(const uint2*)(resid + (int64_t)m * p.ldr + n), r);
                unpack4(*(const uint2*)(gate + (int64_t)b * p.gate_bstride + n), gg);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = r[e] + round_bf16(round_bf16(v[e]) * gg[e]);
                if (hint) {
                    float hv[4];
                    unpack4(*(const uint2*)(hint + (int64_t)m * p.ldh + n), hv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = round_bf16(v[e]) + round_bf16(hv[e] * p.hint_scale);
                }
            }
            if (dead) v[0] = v[1] = v[2] = v[3] = 0.f;
            *(uint2*)(C + (int64_t)m * p.ldc + n) = pack4(v);
        }
    }
}

template <int EPI>
int launch_sw_e(const VcGemmParams& p, hipStream_t stream) {
    constexpr int LDS = 2 * 65536;
    static std::atomic<uint64_t> attr_done{0};
    if (!vc_set_lds_once(attr_done, (const void*)gemm_sw_kernel<EPI>, LDS)) return VC_E_HIP;
    const int ng = p.ngroups > 1 ? p.ngroups : 1;
    const int nTm = (p.M + 255) / 256, nTn = ng * (p.N / 256);
    const int ntiles = nTm * nTn;
    const int grid = (ntiles + 7) / 8 * 8;
    hipLaunchKernelGGL(gemm_sw_kernel<EPI>, dim3(grid), dim3(256), LDS, stream, p, nTm, nTn, ntiles);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}
int launch_sw(const VcGemmParams& p, hipStream_t stream) {
    switch (p.epilogue) {
        case VC_EPI_BIAS: return launch_sw_e<VC_EPI_BIAS>(p, stream);
        case VC_EPI_BIAS_GELU: return launch_sw_e<VC_EPI_BIAS_GELU>(p, stream);
        case VC_EPI_BIAS_RESID: return launch_sw_e<VC_EPI_BIAS_RESID>(p, stream);
        case VC_EPI_BIAS_GATE_RESID: return launch_sw_e<VC_EPI_BIAS_GATE_RESID>(p, stream);
        case VC_EPI_GELU_MUL: return launch_sw_e<VC_EPI_GELU_MUL>(p, stream);
    }
    return VC_E_INVALID;
}

template <class Cfg, bool SB, int EPI>
int launch_cfg_e(const VcGemmParams& p, hipStream_t stream) {
    static std::atomic<uint64_t> attr_done{0};
    if (!vc_set_lds_once(attr_done, (const void*)gemm_bf16_kernel<Cfg, SB, EPI>, Cfg::LDS_BYTES)) return VC_E_HIP;
    const int ng = p.ngroups > 1 ? p.ngroups : 1;
    const int nTm = (p.M + Cfg::BM - 1) / Cfg::BM, nTn = ng * ((p.N + Cfg::BN - 1) / Cfg::BN);
    const int ntiles = nTm * nTn;
    const int grid = (ntiles + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_bf16_kernel<Cfg, SB, EPI>), dim3(grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, p, nTm, nTn,
                       ntiles);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}
template <class Cfg, bool SB = false>
int launch_cfg(const VcGemmParams& p, hipStream_t stream) {
    switch (p.epilogue) {
        case VC_EPI_BIAS: return launch_cfg_e<Cfg, SB, VC_EPI_BIAS>(p, stream);
        case VC_EPI_BIAS_GELU: return launch_cfg_e<Cfg, SB, VC_EPI_BIAS_GELU>(p, stream);
        case VC_EPI_BIAS_RESID: return launch_cfg_e<Cfg, SB, VC_EPI_BIAS_RESID>(p, stream);
        case VC_EPI_BIAS_GATE_RESID: return launch_cfg_e<Cfg, SB, VC_EPI_BIAS_GATE_RESID>(p, stream);
        case VC_EPI_GELU_MUL: return launch_cfg_e<Cfg, SB, VC_EPI_GELU_MUL>(p, stream);
    }
    return VC_E_INVALID;
}

}  // namespace

// shapes / alignments the fp8 instantiation of the ping-pong kernel takes (operand POINTERS other than A, W, C are not looked at: the
// engine asks before it has quantised anything)
bool vc_gemm_fp8_eligible(const VcGemmParams& p) {
    const bool rows_ok8 = (p.M % 256 == 0) || p.a_rows_padded;
    bool ok = rows_ok8 && p.M > 0 && p.N % 256 == 0 && p.K % 256 == 0 && p.lda % 16 == 0 && p.ldw % 16 == 0 && p.ldc % 8 == 0 &&
              ((uintptr_t)p.C & 15) == 0 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0 &&
              (p.rows_per_batch == 0 || p.rows_per_batch >= 256) && p.lda * 256 < (1ll << 31) && p.ldw * 256 < (1ll << 31);
    if (p.resid) ok = ok && p.ldr % 8 == 0 && ((uintptr_t)p.resid & 15) == 0;
    if (p.hint) ok = ok && p.ldh % 8 == 0 && ((uintptr_t)p.hint & 15) == 0;
    for (int g = 1; g < p.ngroups; ++g) ok = ok && ((uintptr_t)p.Cg[g - 1] & 15) == 0;
    return ok && (p.epilogue == VC_EPI_BIAS || p.epilogue == VC_EPI_BIAS_GELU || p.epilogue == VC_EPI_BIAS_RESID || p.epilogue == VC_EPI_BIAS_GATE_RESID);
}

int vc_launch_gemm(const VcGemmParams& p, hipStream_t stream) {
    if (!p.A || !p.W || !p.C || p.M <= 0 || p.N <= 0 || p.K <= 0) return VC_E_INVALID;
    if (p.fp8) {      // e4m3 operands: the ping-pong kernel or nothing
        if (!p.a_scale || !p.w_scale) return VC_E_INVALID;
        for (int g = 1; g < p.ngroups; ++g)
            if (!p.Wg[g - 1] || !p.Cg[g - 1] || !p.w_scaleg[g - 1]) return VC_E_INVALID;
        if ((p.epilogue == VC_EPI_BIAS_RESID || p.epilogue == VC_EPI_BIAS_GATE_RESID) && !p.resid) return VC_E_INVALID;
        if (p.epilogue == VC_EPI_BIAS_GATE_RESID && !p.gate) return VC_E_INVALID;
        return vc_gemm_fp8_eligible(p) && ((uintptr_t)p.w_scale & 15) == 0 ? launch_pp_fp8(p, stream) : VC_E_UNSUPPORTED;
    }
    if (p.K % 64 != 0 || p.N % 4 != 0) return VC_E_UNSUPPORTED;
    if ((p.lda % 8) || (p.ldw % 8) || (p.ldc % 4)) return VC_E_UNSUPPORTED;
    if ((p.epilogue == VC_EPI_BIAS_RESID || p.epilogue == VC_EPI_BIAS_GATE_RESID || p.epilogue == VC_EPI_GELU_MUL) && (!p.resid || p.ldr % 4))
        return VC_E_INVALID;
    if (p.epilogue == VC_EPI_BIAS_GATE_RESID && !p.gate) return VC_E_INVALID;
    if (p.ngroups < 0 || p.ngroups > 3) return VC_E_INVALID;
    for (int g = 1; g < p.ngroups; ++g)
        if (!p.Wg[g - 1] || !p.Cg[g - 1]) return VC_E_INVALID;
    // p.tile (tests / tuning): 0 auto, 1 -> 128x128, 2 -> 256x256, 3 -> 256x256 with 64-bit DMA addresses, 4 -> ping-pong kernel,
    // 5 -> one-wave-per-SIMD kernel
    bool big = (p.M >= 1024 && p.N >= 256);
    // small problems (a 1.3B model on a 9-frame clip: M = 3840, N = 1536 -> 90 tiles of 256 x 256 for 256 CUs): 128 x 128 tiles,
    // two workgroups per CU, fill the chip where the big tile leaves two thirds of it idle; same K order, bit-identical results
    const int64_t t256 = (int64_t)((p.M + 255) / 256) * ((p.N + 255) / 256) * (p.ngroups > 1 ? p.ngroups : 1);
    if (t256 < VC_GEMM_SMALL_TILES) big = false;
    if (p.tile == 1) big = false;
    if (p.tile == 2 || p.tile == 5) big = true;
    // every tile row readable (M a multiple of 256 or padded buffers), N % 256 == 0, K % 128 == 0: ping-pong kernel
    const bool rows_ok = (p.M % 256 == 0) || p.a_rows_padded;
    // its epilogue stores 16 bytes per lane and holds the gate of at most two samples per tile
    bool epi_ok = p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0 && (p.rows_per_batch == 0 || p.rows_per_batch >= 256);
    if (p.resid) epi_ok = epi_ok && p.ldr % 8 == 0 && ((uintptr_t)p.resid & 15) == 0;        // ... and loads them the same way
    if (p.hint) epi_ok = epi_ok && p.ldh % 8 == 0 && ((uintptr_t)p.hint & 15) == 0;
    for (int g = 1; g < p.ngroups; ++g) epi_ok = epi_ok && ((uintptr_t)p.Cg[g - 1] & 15) == 0;
    if (big && rows_ok && epi_ok && p.N % 256 == 0 && p.K % 128 == 0 && p.lda * 512 < (1ll << 31) && p.ldw * 512 < (1ll << 31) &&
        (p.tile == 0 || p.tile == 4 || p.tile == 5 || p.tile == 6)) {
        if (p.tile == 6) {          // ping-pong kernel under the "shared super-band" tile map (A/B tool only)
            if (((p.M + 255) / 256) % 64) return VC_E_UNSUPPORTED;
            VcGemmParams q = p;
            q.tile_map = 1;
            return launch_pp(q, stream);
        }
        return p.tile == 5 ? launch_sw(p, stream) : launch_pp(p, stream);
    }
    // operands below 4 GiB (every shape of the engine): LDS-DMA with 32-bit lane offsets against a scalar base
    const bool fits32 = (int64_t)p.M * p.lda * 2 < (1ll << 32) && (int64_t)p.N * p.ldw * 2 < (1ll << 32);
    if (big && fits32 && p.tile != 3) return launch_cfg<GemmCfg<256, 256, 2, 4>, true>(p, stream);
    if (big) return launch_cfg<GemmCfg<256, 256, 2, 4>>(p, stream);
    return launch_cfg<GemmCfg<128, 128, 2, 2>>(p, stream);
}
