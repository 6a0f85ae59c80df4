// Small kernels of the denoise step: patchify / unpatchify gathers, the fp32 time-embedding MLPs,
// modulation tables, bf16 elementwise helpers.  None of them is on the roofline-relevant part of the step.
//
//   patch_embedding (Conv3d k=s=(1,2,2) == GEMM over 2x2 patches)  wan_transformer3d.py:758-759, VC.py:199-201
//   unpatchify                                                     wan_transformer3d.py:1127-1150
//   sinusoidal_embedding_1d / time_embedding / time_projection     wan_transformer3d.py:39-49, 764-766; VC.py:347-354
//   modulation + e                                                 wan_transformer3d.py:588, 641
#include "vc_common.h"
#include "vc_kernels.h"

namespace {

// thread = (b, c, tok): reads the 2x2 patch, writes 4 contiguous bf16 of row b*Lpad+tok at column c*4
__global__ void patchify_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ A, int B, int C, int T, int H,
                                int W, int Lrows, int tok_offset) {
    const int H2 = H / 2, W2 = W / 2;
    const int L = T * H2 * W2;
    const int64_t total = (int64_t)B * C * Lrows;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int li = (int)(i % Lrows);
        const int tok = tok_offset + li;
        const int c = (int)((i / Lrows) % C);
        const int b = (int)(i / ((int64_t)Lrows * C));
        uint2 o = make_uint2(0u, 0u);
        if (tok < L) {
            const int w2 = tok % W2, h2 = (tok / W2) % H2, f = tok / (W2 * H2);
            const bf16_t* src = x + ((((int64_t)b * C + c) * T + f) * H + 2 * h2) * W + 2 * w2;
            o.x = *(const uint32_t*)src;
            o.y = *(const uint32_t*)(src + W);
        }
        *(uint2*)(A + ((int64_t)b * Lrows + li) * (C * 4) + c * 4) = o;
    }
}

// thread = (b, c, tok): out[b, c, f, 2h+q, 2w+r] = y[b*Lpad + tok, (q*2+r)*C + c]
__global__ void unpatchify_kernel(const bf16_t* __restrict__ y, bf16_t* __restrict__ out, int B, int C, int T,
                                  int H2, int W2, int Lloc) {
    const int L = T * H2 * W2;
    const int64_t total = (int64_t)B * C * L;
    const int H = 2 * H2, W = 2 * W2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int tok = (int)(i % L);
        const int c = (int)((i / L) % C);
        const int b = (int)(i / ((int64_t)L * C));
        const int w2 = tok % W2, h2 = (tok / W2) % H2, f = tok / (W2 * H2);
        const int64_t row = (int64_t)(tok / Lloc) * B * Lloc + (int64_t)b * Lloc + tok % Lloc;
        const bf16_t* src = y + row * (4 * C) + c;
        bf16_t* dst = out + ((((int64_t)b * C + c) * T + f) * H + 2 * h2) * W + 2 * w2;
        dst[0] = src[0];
        dst[1] = src[C];
        dst[W] = src[2 * C];
        dst[W + 1] = src[3 * C];
    }
}

constexpr int SL_MAXB = 8;
// one wave per output n: y[b, n] = sum_k act(x[b, k]) * W[n, k] + bias[n]
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, const bf16_t* __restrict__ W,
                                                           const bf16_t* __restrict__ bias, float* __restrict__ y,
                                                           int B, int N, int K, int silu_input) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    float acc[SL_MAXB];
#pragma unroll
    for (int b = 0; b < SL_MAXB; ++b) acc[b] = 0.f;
    const bf16_t* wr = W + (int64_t)n * K;
    for (int k = lane * 8; k < K; k += 512) {
        float wf[8];
        unpack8(*(const uint4*)(wr + k), wf);
#pragma unroll
        for (int b = 0; b < SL_MAXB; ++b) {
            if (b < B) {
                const float4 x0 = *(const float4*)(x + (int64_t)b * K + k);
                const float4 x1 = *(const float4*)(x + (int64_t)b * K + k + 4);
                float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = silu_input ? silu_f(xv[e]) : xv[e];
                    acc[b] += a * wf[e];
                }
            }
        }
    }
    const float bv = bias ? (float)bias[n] : 0.f;
#pragma unroll
    for (int b = 0; b < SL_MAXB; ++b) {
        if (b < B) {
            const float s = wave_sum(acc[b]);
            if (lane == 0) y[(int64_t)b * N + n] = s + bv;
        }
    }
}

__global__ void sinusoid_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int freq_dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * freq_dim) return;
    const int b = i / freq_dim, j = i % freq_dim, half = freq_dim / 2;
    const int jj = j < half ? j : j - half;
    const double w = pow(10000.0, -(double)jj / (double)half);
    const double s = (double)t[b] * w;
    out[i] = (float)(j < half ? cos(s) : sin(s));
}

__global__ void modulation_kernel(const bf16_t* __restrict__ mod, const float* __restrict__ e, bf16_t* __restrict__ out,
                                  int B, int J, int dim, int64_t e_bstride, int64_t e_jstride) {
    const int64_t total = (int64_t)B * J * dim;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % dim);
        const int j = (int)((i / dim) % J);
        const int b = (int)(i / ((int64_t)J * dim));
        out[i] = (bf16_t)((float)mod[(int64_t)j * dim + col] + round_bf16(e[b * e_bstride + j * e_jstride + col]));
    }
}

__global__ void axpy_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out,
                            float s, int64_t n8, int mode) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float fa[8], fb[8];
        unpack8(*(const uint4*)(a + i * 8), fa);
        unpack8(*(const uint4*)(b + i * 8), fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[e] = mode == 0 ? fa[e] + round_bf16(fb[e] * s) : fa[e] - fb[e];
        *(uint4*)(out + i * 8) = pack8(fa);
    }
}

__global__ void pad_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int len, int total, int dim8) {
    const int64_t n = (int64_t)total * dim8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / dim8);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < len) v = ((const uint4*)src)[i];
        ((uint4*)dst)[i] = v;
    }
}

// Ulysses unpack: recv [B][P_src = head group][Lloc][hd] -> attn [B * Lloc, d]   (M = B * Lloc rows, rpb = Lloc)
__global__ void sp_unpack_o_kernel(const bf16_t* __restrict__ recv, bf16_t* __restrict__ attn, int M, int rpb, int d8, int hd8, int P) {
    const int64_t n = (int64_t)M * d8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % d8), row = (int)(i / d8);
        const int src = col / hd8, c = col - src * hd8;
        const int b = row / rpb, r = row - b * rpb;
        ((uint4*)attn)[i] = ((const uint4*)recv)[(((int64_t)b * P + src) * rpb + r) * hd8 + c];
    }
}

inline int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}
inline int ok() { return hipGetLastError() == hipSuccess ? VC_OK : VC_E_HIP; }


// geoada_context of one sample (pipeline_wan_versecrafter.py:440-488, ref_images = None):
//   out[0:64]          = z                                   (VAE latents of the control videos)
//   out[64 + dy*8+dx]  = mask[0][src(t)][8y+dy][8x+dx]       (8x8 pixel-unshuffle of mask channel 0, then
//                                                             F.interpolate(mode="nearest-exact") over frames)
// src(t) = min(floor((t + 0.5) * (F / T)), F - 1) evaluated in fp32 exactly as ATen's nearest-exact index.
template <typename MaskT>
__global__ __launch_bounds__(256) void geoada_context_kernel(const bf16_t* __restrict__ z, const MaskT* __restrict__ mask,
                                                             bf16_t* __restrict__ out, int T, int h, int w, int F,
                                                             float scale) {
    const int64_t plane = (int64_t)h * w, vol = (int64_t)T * plane;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 128 * vol) return;
    const int c = (int)(i / vol);
    const int64_t rem = i - (int64_t)c * vol;
    if (c < 64) { out[i] = z[i]; return; }
    const int t = (int)(rem / plane);
    const int yx = (int)(rem - (int64_t)t * plane);
    const int y = yx / w, x = yx - y * w;
    const int dy = (c - 64) >> 3, dx = (c - 64) & 7;
    int src = (int)floorf(((float)t + 0.5f) * scale);
    src = src < F - 1 ? src : F - 1;
    const int64_t W = (int64_t)w * 8;
    out[i] = (bf16_t)(float)mask[((int64_t)src * h * 8 + (y * 8 + dy)) * W + x * 8 + dx];
}


// One sampler update on the latent (pipeline_wan_versecrafter.py:903-909): classifier-free-guidance combine, flow
// prediction -> x0, UniPC corrector (uses the previous sample and x0 history) and UniPC predictor, in ONE pass.
// Every elementwise op of the torch formulation (utils/fm_solvers_unipc.py, restating the third-party
// FlowUniPCMultistepScheduler) rounds its result to bf16; the kernel reproduces those roundings op for op, so the result is
// bit-identical to the unfused path.  Scalars (sc[]) are computed on the host exactly as the scheduler computes them.
struct VcUnipcScalars {
    float guidance, sigma_t;
    float ca, cb, cBh, rk_c, rho0_c, rho1_c;      // corrector
    float pa, pb, pBh, rk_p, rho_p;               // predictor
};
__global__ __launch_bounds__(256) void unipc_update_kernel(const bf16_t* __restrict__ nu, const bf16_t* __restrict__ nc,
                                                           const bf16_t* __restrict__ sample, const bf16_t* __restrict__ last,
                                                           const bf16_t* __restrict__ m0p, const bf16_t* __restrict__ m1p,
                                                           bf16_t* __restrict__ x0_out, bf16_t* __restrict__ samp_out,
                                                           bf16_t* __restrict__ next_out, int64_t n, VcUnipcScalars k,
                                                           int flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool do_cfg = flags & 1, use_corr = flags & 2, corr2 = flags & 4, pred2 = flags & 8;
    auto r = [](float v) { return round_bf16(v); };
    const float c = (float)nc[i];
    float noise = c;
    if (do_cfg) {
        const float u = (float)nu[i];
        noise = r(u + r(k.guidance * r(c - u)));
    }
    const float s = (float)sample[i];
    const float x0 = r(s - r(k.sigma_t * noise));
    float samp = s;
    float m0 = 0.f;
    if (use_corr || pred2) m0 = (float)m0p[i];
    if (use_corr) {
        const float xt_ = r(r(k.ca * (float)last[i]) - r(k.cb * m0));
        const float e = r(k.rho1_c * r(x0 - m0));
        float inner = e;
        if (corr2) inner = r(r(k.rho0_c * r(r((float)m1p[i] - m0) / k.rk_c)) + e);
        samp = r(xt_ - r(k.cBh * inner));
    }
    const float xp_ = r(r(k.pa * samp) - r(k.pb * x0));
    float nxt = xp_;
    if (pred2) nxt = r(xp_ - r(k.pBh * r(k.rho_p * r(r(m0 - x0) / k.rk_p))));
    x0_out[i] = (bf16_t)x0;
    if (samp_out) samp_out[i] = (bf16_t)samp;
    next_out[i] = (bf16_t)nxt;
}

}  // namespace

int vc_launch_patchify(const void* x, void* A, int B, int C, int T, int H, int W, int Lrows, int tok_offset,
                       hipStream_t s) {
    if (!x || !A || B <= 0 || C <= 0 || T <= 0 || H <= 0 || W <= 0 || Lrows <= 0 || tok_offset < 0) return VC_E_INVALID;
    if ((H & 1) || (W & 1)) return VC_E_INVALID;
    const int64_t n = (int64_t)B * C * Lrows;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)A, B, C, T,
                       H, W, Lrows, tok_offset);
    return ok();
}

int vc_launch_unpatchify(const void* y, void* out, int B, int C, int T, int H2, int W2, int Lloc, hipStream_t s) {
    if (!y || !out || B <= 0 || C <= 0 || T <= 0 || H2 <= 0 || W2 <= 0 || Lloc <= 0) return VC_E_INVALID;
    const int64_t n = (int64_t)B * C * T * H2 * W2;
    hipLaunchKernelGGL(unpatchify_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, (const bf16_t*)y, (bf16_t*)out, B,
                       C, T, H2, W2, Lloc);
    return ok();
}

int vc_launch_small_linear(const float* x, const void* W, const void* bias, float* y, int B, int N, int K,
                           int silu_input, hipStream_t s) {
    if (!x || !W || !y || B <= 0 || B > SL_MAXB || N <= 0 || K <= 0) return VC_E_INVALID;
    if (K % 8) return VC_E_UNSUPPORTED;
    hipLaunchKernelGGL(small_linear_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, (const bf16_t*)W,
                       (const bf16_t*)bias, y, B, N, K, silu_input);
    return ok();
}

int vc_launch_sinusoid(const float* t, float* out, int B, int freq_dim, hipStream_t s) {
    if (!t || !out || B <= 0 || freq_dim <= 0 || (freq_dim & 1)) return VC_E_INVALID;
    hipLaunchKernelGGL(sinusoid_kernel, dim3((B * freq_dim + 255) / 256), dim3(256), 0, s, t, out, B, freq_dim);
    return ok();
}

int vc_launch_modulation(const void* mod, const float* e, void* out, int B, int J, int dim, int64_t e_bstride,
                         int64_t e_jstride, hipStream_t s) {
    if (!mod || !e || !out || B <= 0 || J <= 0 || dim <= 0) return VC_E_INVALID;
    hipLaunchKernelGGL(modulation_kernel, dim3(grid_for((int64_t)B * J * dim, 256)), dim3(256), 0, s,
                       (const bf16_t*)mod, e, (bf16_t*)out, B, J, dim, e_bstride, e_jstride);
    return ok();
}

int vc_launch_axpy(const void* a, const void* b, void* out, float sc, int64_t n, hipStream_t st) {
    if (!a || !b || !out || n <= 0 || n % 8) return VC_E_INVALID;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b,
                       (bf16_t*)out, sc, n / 8, 0);
    return ok();
}

int vc_launch_sub(const void* a, const void* b, void* out, int64_t n, hipStream_t st) {
    if (!a || !b || !out || n <= 0 || n % 8) return VC_E_INVALID;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b,
                       (bf16_t*)out, 0.f, n / 8, 1);
    return ok();
}

int vc_launch_pad_rows(const void* src, void* dst, int len, int total, int dim, hipStream_t st) {
    if (!dst || total <= 0 || dim <= 0 || dim % 8 || len < 0 || len > total || (len > 0 && !src)) return VC_E_INVALID;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for((int64_t)total * dim / 8, 256)), dim3(256), 0, st,
                       (const bf16_t*)src, (bf16_t*)dst, len, total, dim / 8);
    return ok();
}

// Occupies `st` for `usec` microseconds with one idle wave (what-if timing of an exchange's wire time: tools/sim_sp_rank.py).
// The wave polls the constant-rate wall clock and sleeps in between; it exits on the deadline or after a fixed number of polls.
__global__ void delay_kernel(uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    for (int i = 0; i < (1 << 24); ++i) {
        if (wall_clock64() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(64);
    }
}

int vc_launch_delay(double usec, hipStream_t st) {
    if (!(usec >= 0) || usec > 5e6) return VC_E_INVALID;
    static int khz = 0;
    if (!khz) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) { khz = 0; return VC_E_HIP; }
    }
    const uint64_t ticks = (uint64_t)(usec * 1e-3 * khz);
    if (ticks == 0) return VC_OK;
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, st, ticks);
    return ok();
}

int vc_launch_sp_unpack_o(const void* recv, void* attn, int M, int rows_per_batch, int d, int P, hipStream_t st) {
    if (!recv || !attn || M <= 0 || d <= 0 || P <= 0 || d % (8 * P)) return VC_E_INVALID;
    if (rows_per_batch <= 0 || M % rows_per_batch) return VC_E_INVALID;
    hipLaunchKernelGGL(sp_unpack_o_kernel, dim3(grid_for((int64_t)M * d / 8, 256)), dim3(256), 0, st,
                       (const bf16_t*)recv, (bf16_t*)attn, M, rows_per_batch, d / 8, d / P / 8, P);
    return ok();
}

int vc_launch_geoada_context(const void* z, const void* mask, int mask_is_f32, void* out, int T, int h, int w, int F,
                             hipStream_t st) {
    if (!z || !mask || !out || T <= 0 || h <= 0 || w <= 0 || F <= 0) return VC_E_INVALID;
    const int64_t n = (int64_t)128 * T * h * w;
    const float scale = (float)F / (float)T;
    if (mask_is_f32)
        hipLaunchKernelGGL(geoada_context_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const bf16_t*)z,
                           (const float*)mask, (bf16_t*)out, T, h, w, F, scale);
    else
        hipLaunchKernelGGL(geoada_context_kernel<bf16_t>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const bf16_t*)z,
                           (const bf16_t*)mask, (bf16_t*)out, T, h, w, F, scale);
    return ok();
}

int vc_launch_unipc_update(const void* noise_uncond, const void* noise_cond, const void* sample, const void* last,
                           const void* m0, const void* m1, void* x0_out, void* samp_out, void* next_out, int64_t n,
                           const float* sc, int flags, hipStream_t st) {
    if (!noise_cond || !sample || !x0_out || !next_out || !sc || n <= 0) return VC_E_INVALID;
    if ((flags & 1) && !noise_uncond) return VC_E_INVALID;
    if ((flags & 2) && (!last || !m0)) return VC_E_INVALID;
    if ((flags & 4) && (!(flags & 2) || !m1)) return VC_E_INVALID;
    if ((flags & 8) && !m0) return VC_E_INVALID;
    VcUnipcScalars k;
    k.guidance = sc[0]; k.sigma_t = sc[1];
    k.ca = sc[2]; k.cb = sc[3]; k.cBh = sc[4]; k.rk_c = sc[5]; k.rho0_c = sc[6]; k.rho1_c = sc[7];
    k.pa = sc[8]; k.pb = sc[9]; k.pBh = sc[10]; k.rk_p = sc[11]; k.rho_p = sc[12];
    hipLaunchKernelGGL(unipc_update_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, (const bf16_t*)noise_uncond,
                       (const bf16_t*)noise_cond, (const bf16_t*)sample, (const bf16_t*)last, (const bf16_t*)m0,
                       (const bf16_t*)m1, (bf16_t*)x0_out, (bf16_t*)samp_out, (bf16_t*)next_out, n, k, flags);
    return ok();
}
