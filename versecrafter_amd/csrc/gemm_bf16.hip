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

template <class Cfg, bool SB>
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
    const bf16_t* bias = (const bf16_t*)(grp == 0 ? p.bias : p.biasg[grp - 1]);
    const bf16_t* resid = (const bf16_t*)p.resid;
    const bf16_t* gate = (const bf16_t*)p.gate;
    const bf16_t* hint = (const bf16_t*)p.hint;
    bf16_t* C = (bf16_t*)(grp == 0 ? p.C : p.Cg[grp - 1]);
    const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * Cfg::WTM + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const int b = m / rpb;
        const bool dead = p.valid_rows >= 0 && (m - b * rpb) >= p.valid_rows;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * Cfg::WTN + j * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (bias) {
                float bb[4];
                unpack4(*(const uint2*)(bias + n), bb);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bb[e];
            }
            if (p.epilogue == VC_EPI_BIAS_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(round_bf16(v[e]));
            } else if (p.epilogue == VC_EPI_BIAS_RESID) {
                float r[4];
                unpack4(*(const uint2*)(resid + (int64_t)m * p.ldr + n), r);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = r[e] + round_bf16(v[e]);
            } else if (p.epilogue == VC_EPI_BIAS_GATE_RESID) {
                float r[4], g[4];
                unpack4(*(const uint2*)(resid + (int64_t)m * p.ldr + n), r);
                unpack4(*(const uint2*)(gate + (int64_t)b * p.gate_bstride + n), g);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = r[e] + round_bf16(round_bf16(v[e]) * g[e]);
                if (hint) {
                    float h[4];
                    unpack4(*(const uint2*)(hint + (int64_t)m * p.ldh + n), h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = round_bf16(v[e]) + round_bf16(h[e] * p.hint_scale);
                }
            }
            if (dead) v[0] = v[1] = v[2] = v[3] = 0.f;
            *(uint2*)(C + (int64_t)m * p.ldc + n) = pack4(v);
        }
    }
}

template <class Cfg, bool SB = false>
int launch_cfg(const VcGemmParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<Cfg, SB>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return VC_E_HIP;
        attr_set = true;
    }
    const int ng = p.ngroups > 1 ? p.ngroups : 1;
    const int nTm = (p.M + Cfg::BM - 1) / Cfg::BM, nTn = ng * ((p.N + Cfg::BN - 1) / Cfg::BN);
    const int ntiles = nTm * nTn;
    const int grid = (ntiles + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_bf16_kernel<Cfg, SB>), dim3(grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, p, nTm, nTn,
                       ntiles);
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}

}  // namespace

int vc_gemm_tile_override = 0;   // 0 auto, 1 -> 128x128, 2 -> 256x256, 3 -> 256x256 with 64-bit DMA addresses (tests)

int vc_launch_gemm(const VcGemmParams& p, hipStream_t stream) {
    if (!p.A || !p.W || !p.C || p.M <= 0 || p.N <= 0 || p.K <= 0) return VC_E_INVALID;
    if (p.K % 64 != 0 || p.N % 4 != 0) return VC_E_UNSUPPORTED;
    if ((p.lda % 8) || (p.ldw % 8) || (p.ldc % 4)) return VC_E_UNSUPPORTED;
    if ((p.epilogue == VC_EPI_BIAS_RESID || p.epilogue == VC_EPI_BIAS_GATE_RESID) && (!p.resid || p.ldr % 4))
        return VC_E_INVALID;
    if (p.epilogue == VC_EPI_BIAS_GATE_RESID && !p.gate) return VC_E_INVALID;
    if (p.ngroups < 0 || p.ngroups > 3) return VC_E_INVALID;
    for (int g = 1; g < p.ngroups; ++g)
        if (!p.Wg[g - 1] || !p.Cg[g - 1]) return VC_E_INVALID;
    bool big = (p.M >= 1024 && p.N >= 256);
    if (vc_gemm_tile_override == 1) big = false;
    if (vc_gemm_tile_override == 2) big = true;
    // operands below 4 GiB (every shape of the engine): LDS-DMA with 32-bit lane offsets against a scalar base
    const bool fits32 = (int64_t)p.M * p.lda * 2 < (1ll << 32) && (int64_t)p.N * p.ldw * 2 < (1ll << 32);
    if (big && fits32 && vc_gemm_tile_override != 3) return launch_cfg<GemmCfg<256, 256, 2, 4>, true>(p, stream);
    if (big) return launch_cfg<GemmCfg<256, 256, 2, 4>>(p, stream);
    return launch_cfg<GemmCfg<128, 128, 2, 2>>(p, stream);
}
