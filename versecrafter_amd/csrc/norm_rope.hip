// Row kernels (HBM-bound): fp32 LayerNorm + adaLN modulate / affine, and full-dim RMSNorm + 3-axis RoPE.
//
//   WanLayerNorm            wan_transformer3d.py:326-346  (fp32 stats, biased variance, eps 1e-6)
//   adaLN modulate          wan_transformer3d.py:591, 603, 643   y = LN(x) * (1 + scale) + shift
//   norm3 (affine)          wan_transformer3d.py:548-550, 600
//   WanRMSNorm              wan_transformer3d.py:307-323  (reduction over the FULL model dim, not per head)
//   rope_apply              wan_transformer3d.py:143-172  (adjacent pairs, [22|21|21] split of 64 freqs)
//
// One 64-lane wave owns one row; the row stays in registers between the statistics pass and the
// normalise pass (one HBM read + one HBM write per element), 16-byte bf16x8 accesses per lane.
// Rounding points follow the reference under bf16 autocast (LN output, rsqrt, each bf16 multiply/add).
#include "vc_common.h"
#include "vc_kernels.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;

// Q8 (fp8 linear layers, vc_set_fp8_linear): the row is NOT written as bf16 but as the e4m3 operand of the GEMM it feeds -- bytes
// q[row][dim] + one scale per row, exactly what quantize_rows_fp8_kernel makes of the bf16 row (amax / 448, RNE) -- so the quantiser's
// pass over the row (2 bytes read + 1 written per element) and the bf16 write (2 bytes) disappear.
template <int MAXC, bool Q8 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                        int rows, int dim, int rows_per_batch, float eps, int mode,
                                                        const bf16_t* __restrict__ p0,
                                                        const bf16_t* __restrict__ p1, int64_t p_bstride,
                                                        uint8_t* __restrict__ q = nullptr, float* __restrict__ qscale = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * ROWS_PER_BLOCK + wave;
    if (row >= rows) return;
    const bf16_t* xr = x + (int64_t)row * dim;
    uint4 raw[MAXC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int idx = c * 512 + lane * 8;
        if (idx < dim) {
            raw[c] = *(const uint4*)(xr + idx);
            float f[8];
            unpack8(raw[c], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += f[e];
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    // keep the ROW (packed bf16, MAXC x 4 registers) live across the passes, not its fp32 expansion (x 2 the registers):
    // the opaque asm stops the compiler from reusing the unpacked values, which would cost 3 of 8 waves per SIMD
#pragma unroll
    for (int c = 0; c < MAXC; ++c) asm volatile("" : "+v"(raw[c].x), "+v"(raw[c].y), "+v"(raw[c].z), "+v"(raw[c].w));
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int idx = c * 512 + lane * 8;
        if (idx < dim) {
            float f[8];
            unpack8(raw[c], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; v += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)dim + eps);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) asm volatile("" : "+v"(raw[c].x), "+v"(raw[c].y), "+v"(raw[c].z), "+v"(raw[c].w));
    const int b = rows_per_batch > 0 ? row / rows_per_batch : 0;
    const bf16_t* q0 = mode == 0 ? p0 + (int64_t)b * p_bstride : p0;   // scale (mode 0) | weight (mode 1)
    const bf16_t* q1 = mode == 0 ? p1 + (int64_t)b * p_bstride : p1;   // shift (mode 0) | bias   (mode 1)
    bf16_t* yr = y + (int64_t)row * dim;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int idx = c * 512 + lane * 8;
        if (idx < dim) {
            float f[8], a[8], bb[8];
            unpack8(raw[c], f);
            unpack8(*(const uint4*)(q0 + idx), a);
            unpack8(*(const uint4*)(q1 + idx), bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float n = (f[e] - mean) * rstd;
                if (mode == 0)   // bf16(bf16(LN) * bf16(1 + scale)) + shift
                    f[e] = round_bf16(round_bf16(n) * round_bf16(1.0f + a[e])) + bb[e];
                else             // fp32 affine inside F.layer_norm, one rounding at the end
                    f[e] = n * a[e] + bb[e];
            }
            if constexpr (Q8) raw[c] = pack8(f);         // the bf16 row, kept in the registers that held the input row
            else *(uint4*)(yr + idx) = pack8(f);
        }
    }
    if constexpr (Q8) {
        float amax = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int idx = c * 512 + lane * 8;
            if (idx < dim) {
                float f[8];
                unpack8(raw[c], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
            }
        }
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
        const float inv = 1.0f / sc;
        if (lane == 0) qscale[row] = sc;
        uint8_t* qr = q + (int64_t)row * dim;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int idx = c * 512 + lane * 8;
            if (idx < dim) {
                float f[8];
                unpack8(raw[c], f);
                unsigned out[2];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[2 * e] * inv, f[2 * e + 1] * inv, 0, false);
                    if (e & 1) out[e >> 1] |= ((unsigned)pk & 0xFFFFu) << 16;
                    else out[e >> 1] = (unsigned)pk & 0xFFFFu;
                }
                *(uint2*)(qr + idx) = uint2{out[0], out[1]};
            }
        }
    }
}

template <int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(bf16_t* __restrict__ x, int64_t ld, int rows, int dim,
                                                           const bf16_t* __restrict__ w, float eps,
                                                           const float2* __restrict__ table, VcRopeGrid grid,
                                                           int use_rope) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * ROWS_PER_BLOCK + wave;
    if (row >= rows) return;
    bf16_t* xr = x + (int64_t)row * ld;
    uint4 raw[MAXC];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int idx = c * 512 + lane * 8;
        if (idx < dim) {
            raw[c] = *(const uint4*)(xr + idx);
            float f[8];
            unpack8(raw[c], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
        }
    }
    const float inv = round_bf16(rsqrtf(wave_sum(ss) / (float)dim + eps));   // rsqrt(...).to(bf16), WT.py:323
#pragma unroll
    for (int c = 0; c < MAXC; ++c) asm volatile("" : "+v"(raw[c].x), "+v"(raw[c].y), "+v"(raw[c].z), "+v"(raw[c].w));   // as in layernorm_kernel

    // RoPE multipliers: this lane's 4 pairs have the same in-head pair index j0..j0+3 in every chunk
    float cs[4], sn[4];
    bool rot = false;
    if (use_rope) {
        const int tok = grid.token_offset + (grid.rows_per_batch > 0 ? row % grid.rows_per_batch : row);
        if (tok < grid.F * grid.H * grid.W) {
            rot = true;
            const int hw = grid.H * grid.W;
            const int pf = tok / hw, ph = (tok / grid.W) % grid.H, pw = tok % grid.W;
            const int j0 = (lane & 15) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + e;
                const int pos = j < 22 ? pf : (j < 43 ? ph : pw);
                const float2 t = table[pos * 64 + j];
                cs[e] = t.x;
                sn[e] = t.y;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int idx = c * 512 + lane * 8;
        if (idx < dim) {
            float f[8], ww[8];
            unpack8(raw[c], f);
            unpack8(*(const uint4*)(w + idx), ww);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = round_bf16(round_bf16(f[e] * inv) * ww[e]);
            if (rot) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = f[2 * e], bq = f[2 * e + 1];
                    f[2 * e] = a * cs[e] - bq * sn[e];
                    f[2 * e + 1] = a * sn[e] + bq * cs[e];
                }
            }
            *(uint4*)(xr + idx) = pack8(f);
        }
    }
}

// Self-attention front (WT.py:385-392) in ONE pass over the q|k|v buffer [rows][3 dim]: blockIdx.y = 0 / 1: full-dim RMSNorm + RoPE
// of q / k (arithmetic of rmsnorm_rope_kernel, bit for bit), blockIdx.y = 2: v.  Written in place (send == nullptr; v is then
// not launched) or straight into the Ulysses exchange layout send[3][B][P_dst][Lloc][dim / P] -- which makes the separate pack pass
// (and its second read + write of q|k|v) disappear.
template <int MAXC, bool PACK>
__global__ __launch_bounds__(256) void qkv_front_kernel(bf16_t* __restrict__ qkv, int rows, int dim, const bf16_t* __restrict__ wq,
                                                        const bf16_t* __restrict__ wk, float eps, const float2* __restrict__ table,
                                                        VcRopeGrid grid, bf16_t* __restrict__ send, int P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * ROWS_PER_BLOCK + wave;
    const int which = blockIdx.y;
    if (row >= rows) return;
    bf16_t* xr = qkv + (int64_t)row * 3 * dim + (int64_t)which * dim;
    const int hd = dim / P;
    const float inv_hd = 1.0f / (float)hd;              // idx / hd for idx < 8192, hd >= 128: exact through the reciprocal
    // exchange layout: send[3 (q, k, v)][B][P_dst][Lloc][hd]  (row = b * Lloc + i; Lloc = grid.rows_per_batch)
    const int rpb = grid.rows_per_batch > 0 ? grid.rows_per_batch : rows;
    const int sb = row / rpb, si = row - sb * rpb;
    bf16_t* srow = PACK ? send + ((int64_t)which * P * rows + (int64_t)sb * P * rpb + si) * hd : nullptr;
    const int64_t peer_stride = (int64_t)rpb * hd;
    auto out_ptr = [&](int idx) -> bf16_t* {
        if (!PACK) return xr + idx;
        const int peer = (int)(((float)idx + 0.5f) * inv_hd);
        return srow + peer * peer_stride + (idx - peer * hd);
    };
    uint4 raw[MAXC];
    if (PACK && which == 2) {                          // v: move only
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int idx = c * 512 + lane * 8;
            if (idx < dim) *(uint4*)out_ptr(idx) = *(const uint4*)(xr + idx);
        }
        return;
    }
    const bf16_t* w = which == 0 ? wq : wk;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int idx = c * 512 + lane * 8;
        if (idx < dim) {
            raw[c] = *(const uint4*)(xr + idx);
            float f[8];
            unpack8(raw[c], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
        }
    }
    const float inv = round_bf16(rsqrtf(wave_sum(ss) / (float)dim + eps));   // rsqrt(...).to(bf16), WT.py:323
#pragma unroll
    for (int c = 0; c < MAXC; ++c) asm volatile("" : "+v"(raw[c].x), "+v"(raw[c].y), "+v"(raw[c].z), "+v"(raw[c].w));
    float cs[4], sn[4];
    bool rot = false;
    {
        const int tok = grid.token_offset + (grid.rows_per_batch > 0 ? row % grid.rows_per_batch : row);
        if (tok < grid.F * grid.H * grid.W) {
            rot = true;
            const int hw = grid.H * grid.W;
            const int pf = tok / hw, ph = (tok / grid.W) % grid.H, pw = tok % grid.W;
            const int j0 = (lane & 15) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + e;
                const int pos = j < 22 ? pf : (j < 43 ? ph : pw);
                const float2 t = table[pos * 64 + j];
                cs[e] = t.x;
                sn[e] = t.y;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int idx = c * 512 + lane * 8;
        if (idx < dim) {
            float f[8], ww[8];
            unpack8(raw[c], f);
            unpack8(*(const uint4*)(w + idx), ww);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = round_bf16(round_bf16(f[e] * inv) * ww[e]);
            if (rot) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = f[2 * e], bq = f[2 * e + 1];
                    f[2 * e] = a * cs[e] - bq * sn[e];
                    f[2 * e + 1] = a * sn[e] + bq * cs[e];
                }
            }
            *(uint4*)out_ptr(idx) = pack8(f);
        }
    }
}

}  // namespace

int vc_launch_qkv_front(void* qkv, int rows, int dim, const void* wq, const void* wk, float eps, const float2* rope_table,
                        const VcRopeGrid* grid, void* send, int P, hipStream_t stream) {
    if (!qkv || !wq || !wk || !rope_table || !grid || rows <= 0 || dim <= 0 || P <= 0) return VC_E_INVALID;
    if (dim % 128 || dim > 8192 || dim % P || (dim / P) % 128) return VC_E_UNSUPPORTED;
    const VcRopeGrid g = *grid;
    if (g.F > 1024 || g.H > 1024 || g.W > 1024 || g.F <= 0 || g.H <= 0 || g.W <= 0) return VC_E_INVALID;
    const dim3 gr((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, send ? 3 : 2), block(256);
#define QF_LAUNCH(MC)                                                                                                    \
    do {                                                                                                                 \
        if (send) hipLaunchKernelGGL((qkv_front_kernel<MC, true>), gr, block, 0, stream, (bf16_t*)qkv, rows, dim,        \
                                     (const bf16_t*)wq, (const bf16_t*)wk, eps, rope_table, g, (bf16_t*)send, P);        \
        else hipLaunchKernelGGL((qkv_front_kernel<MC, false>), gr, block, 0, stream, (bf16_t*)qkv, rows, dim,            \
                                (const bf16_t*)wq, (const bf16_t*)wk, eps, rope_table, g, (bf16_t*)send, P);             \
    } while (0)
    if (dim <= 512) QF_LAUNCH(1);
    else if (dim <= 2048) QF_LAUNCH(4);
    else if (dim <= 5120) QF_LAUNCH(10);
    else QF_LAUNCH(16);
#undef QF_LAUNCH
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}

int vc_launch_layernorm(const void* x, void* y, int rows, int dim, int rows_per_batch, float eps, int mode,
                        const void* p0, const void* p1, int64_t p_bstride, hipStream_t stream) {
    if (!x || !y || !p0 || !p1 || rows <= 0 || dim <= 0) return VC_E_INVALID;
    if (dim % 8 || dim > 8192 || (p_bstride % 8)) return VC_E_UNSUPPORTED;
    const dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
#define LN_LAUNCH(MC)                                                                                          \
    hipLaunchKernelGGL(layernorm_kernel<MC>, grid, block, 0, stream, (const bf16_t*)x, (bf16_t*)y, rows, dim,  \
                       rows_per_batch, eps, mode, (const bf16_t*)p0, (const bf16_t*)p1, p_bstride)
    if (dim <= 512) LN_LAUNCH(1);
    else if (dim <= 2048) LN_LAUNCH(4);
    else if (dim <= 5120) LN_LAUNCH(10);
    else LN_LAUNCH(16);
#undef LN_LAUNCH
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}

int vc_launch_layernorm_q8(const void* x, void* q, float* qscale, int rows, int dim, int rows_per_batch, float eps, int mode,
                           const void* p0, const void* p1, int64_t p_bstride, hipStream_t stream) {
    if (!x || !q || !qscale || !p0 || !p1 || rows <= 0 || dim <= 0) return VC_E_INVALID;
    if (dim % 8 || dim > 8192 || (p_bstride % 8) || ((uintptr_t)q & 7)) return VC_E_UNSUPPORTED;
    const dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
#define LNQ_LAUNCH(MC)                                                                                               \
    hipLaunchKernelGGL((layernorm_kernel<MC, true>), grid, block, 0, stream, (const bf16_t*)x, (bf16_t*)nullptr, rows, dim, \
                       rows_per_batch, eps, mode, (const bf16_t*)p0, (const bf16_t*)p1, p_bstride, (uint8_t*)q, qscale)
    if (dim <= 512) LNQ_LAUNCH(1);
    else if (dim <= 2048) LNQ_LAUNCH(4);
    else if (dim <= 5120) LNQ_LAUNCH(10);
    else LNQ_LAUNCH(16);
#undef LNQ_LAUNCH
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}

int vc_launch_rmsnorm_rope(void* x, int64_t ld, int rows, int dim, const void* w, float eps,
                           const float2* rope_table, const VcRopeGrid* grid, hipStream_t stream) {
    if (!x || !w || rows <= 0 || dim <= 0) return VC_E_INVALID;
    if (dim % 8 || dim > 8192 || ld % 8) return VC_E_UNSUPPORTED;
    const int use_rope = (rope_table && grid) ? 1 : 0;
    if (use_rope && dim % 128) return VC_E_UNSUPPORTED;
    VcRopeGrid g = use_rope ? *grid : VcRopeGrid{0, 0, 0, 0, 0};
    if (use_rope && (g.F > 1024 || g.H > 1024 || g.W > 1024 || g.F <= 0 || g.H <= 0 || g.W <= 0)) return VC_E_INVALID;
    const dim3 gr((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
#define RMS_LAUNCH(MC)                                                                                      \
    hipLaunchKernelGGL(rmsnorm_rope_kernel<MC>, gr, block, 0, stream, (bf16_t*)x, ld, rows, dim,            \
                       (const bf16_t*)w, eps, rope_table, g, use_rope)
    if (dim <= 512) RMS_LAUNCH(1);
    else if (dim <= 2048) RMS_LAUNCH(4);
    else if (dim <= 5120) RMS_LAUNCH(10);
    else RMS_LAUNCH(16);
#undef RMS_LAUNCH
    return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP;
}
