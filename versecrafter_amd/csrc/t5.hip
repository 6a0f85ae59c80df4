// umT5 text encoder (the Wan2.1 "WanT5EncoderModel" the reference's pipeline calls once per prompt:
// pipeline_wan_versecrafter.py:221-282, 273; inference/versecrafter_inference.py:243-249; hyper-parameters
// config/wan2.1/wan_civitai.yaml:14-26: vocab 256384, dim 4096, 64 heads x 64, ffn 10240, 24 layers, 32 buckets).
//
// The class itself lives in the un-vendored videox_fun package (origin: Wan2.1 wan/modules/t5.py), so the algorithm is
// restated from the published T5 v1.1 / umT5 encoder and pinned against transformers' UMT5EncoderModel (same arithmetic):
//   x = token_embedding[ids]
//   per layer:  x += o( softmax( q(n1 x) k(n1 x)^T + rel_bias[bucket(j - i)] + key_mask ) v(n1 x) )      (no 1/sqrt(d) scale)
//               x += fc2( gelu_tanh(gate(n2 x)) * fc1(n2 x) )
//   out = norm(x)                 n*(x) = w * bf16(x * rsqrt(mean(x^2) + eps))   (T5LayerNorm: no mean subtraction, no bias)
// Linear layers have no bias.  Every layer owns its relative-position embedding (shared_pos = False).
//
// GEMMs run on the engine's bf16 MFMA kernels (gemm_bf16.hip; the gated GELU is a GEMM epilogue); the row kernels and the
// 512-token attention below are small: the encoder runs once per video (~10 TFLOP for a prompt pair).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/vcengine.h"
#include "vc_common.h"
#include "vc_kernels.h"

namespace {

// ---- kernels ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void t5_embed_kernel(const int32_t* __restrict__ ids, const bf16_t* __restrict__ table,
                                                       bf16_t* __restrict__ x, int rows, int dim8, int vocab) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * dim8) return;
    const int row = (int)(i / dim8), c = (int)(i - (int64_t)row * dim8);
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    ((uint4*)x)[i] = ((const uint4*)table)[(int64_t)id * dim8 + c];
}

// T5LayerNorm, one wave per row:  y = w * bf16(x * rsqrt(mean(x^2) + eps))
__global__ __launch_bounds__(256) void t5_rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                         bf16_t* __restrict__ y, int rows, int dim, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const bf16_t* xr = x + (int64_t)row * dim;
    float ss = 0.f;
    for (int idx = lane * 8; idx < dim; idx += 512) {
        float f[8];
        unpack8(*(const uint4*)(xr + idx), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)dim + eps);
    bf16_t* yr = y + (int64_t)row * dim;
    for (int idx = lane * 8; idx < dim; idx += 512) {
        float f[8], g[8];
        unpack8(*(const uint4*)(xr + idx), f);
        unpack8(*(const uint4*)(w + idx), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = g[e] * round_bf16(f[e] * rstd);
        *(uint4*)(yr + idx) = pack8(f);
    }
}

// Self-attention of one (batch, head, query row) per wave; head dim 64, L <= 1024 keys (multiple of 64).
//   s_j = bf16(q . k_j) + rel_emb[bucket[j - i + L - 1]][head]   (bf16 add), masked keys -> -inf; softmax in fp32, weights
//   rounded to bf16; out = sum_j p_j v_j.
constexpr int T5_MAXJ = 16;
__global__ __launch_bounds__(256) void t5_attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                           const bf16_t* __restrict__ rel_emb,   // [num_buckets, H]
                                                           const int32_t* __restrict__ bucket,   // [2L - 1]
                                                           const int32_t* __restrict__ mask,     // [B, L] or null
                                                           int B, int H, int L, int DA) {
    __shared__ float pbuf[4][T5_MAXJ * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    if (wid >= (int64_t)B * H * L) return;           // L % 4 == 0: the whole block leaves together
    const int qi = (int)(wid % L);
    const int h = (int)((wid / L) % H);
    const int b = (int)(wid / ((int64_t)L * H));
    const int64_t ld = 3 * (int64_t)DA;
    const bf16_t* qrow = qkv + ((int64_t)b * L + qi) * ld + h * 64;
    float q[64];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float f[8];
        unpack8(*(const uint4*)(qrow + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[c * 8 + e] = f[e];
    }
    const int nj = L >> 6;
    float sc[T5_MAXJ];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < T5_MAXJ; ++j) {
        sc[j] = -3.0e38f;
        if (j < nj) {
            const int key = j * 64 + lane;
            const bf16_t* krow = qkv + ((int64_t)b * L + key) * ld + DA + h * 64;
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float f[8];
                unpack8(*(const uint4*)(krow + c * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) dot += q[c * 8 + e] * f[e];
            }
            const float bias = (float)rel_emb[(int64_t)bucket[key - qi + L - 1] * H + h];
            float s = round_bf16(round_bf16(dot) + bias);
            if (mask && mask[(int64_t)b * L + key] == 0) s = -3.0e38f;
            sc[j] = s;
            mx = fmaxf(mx, s);
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < T5_MAXJ; ++j)
        if (j < nj) { sc[j] = __expf(sc[j] - mx); sum += sc[j]; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < T5_MAXJ; ++j)
        if (j < nj) pbuf[wave][j * 64 + lane] = round_bf16(sc[j] * inv);
    __syncthreads();
    // lane = output dim
    const bf16_t* vbase = qkv + (int64_t)b * L * ld + 2 * DA + h * 64 + lane;
    float acc = 0.f;
#pragma unroll 8
    for (int key = 0; key < L; ++key) acc += pbuf[wave][key] * (float)vbase[(int64_t)key * ld];
    out[((int64_t)b * L + qi) * DA + h * 64 + lane] = (bf16_t)acc;
}

inline int grid_for(int64_t n, int block) { return (int)((n + block - 1) / block); }

struct LayerW {
    const void *n1, *q, *k, *v, *o, *n2, *gate, *fc1, *fc2, *pos;
};

}  // namespace

struct vc_t5 {
    vc_t5_config cfg;
    std::unordered_map<std::string, std::pair<const void*, std::vector<int64_t>>> slots;
    std::vector<LayerW> layers;
    const void *emb = nullptr, *norm = nullptr;
    bool resolved = false;
    char* arena = nullptr;
    int64_t arena_bytes = 0;
    int ws_B = 0, ws_L = 0;
    void *x = nullptr, *t = nullptr, *qkv = nullptr, *a = nullptr, *u = nullptr;
    int32_t* bucket = nullptr;
    std::string err;
};

namespace {

thread_local std::string g_t5_create_error;

int t5_fail(vc_t5* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_t5_create_error = buf;
    return code;
}

// T5 relative position bucket, bidirectional (encoder): half of the buckets per sign, exact below max_exact, log-spaced
// up to max_distance (Wan t5.py T5RelativeEmbedding._relative_position_bucket == transformers T5Attention._relative_position_bucket)
int rel_bucket(int rel, int num_buckets, int max_distance) {
    const int nb = num_buckets / 2;
    int ret = rel > 0 ? nb : 0;
    int n = rel < 0 ? -rel : rel;
    const int max_exact = nb / 2;
    if (n < max_exact) return ret + n;
    // torch: max_exact + (log(n / max_exact) / log(max_distance / max_exact) * (nb - max_exact)).long(), clamped to nb - 1
    const float v = logf((float)n / (float)max_exact) / logf((float)max_distance / (float)max_exact) * (float)(nb - max_exact);
    int large = max_exact + (int)v;
    if (large > nb - 1) large = nb - 1;
    return ret + large;
}

const void* need(vc_t5* h, const std::string& key, std::vector<int64_t> shape, int* rc) {
    auto it = h->slots.find(key);
    if (it == h->slots.end()) { *rc = t5_fail(h, VC_E_STATE, "weight '%s' was never loaded (vc_t5_load_weight)", key.c_str()); return nullptr; }
    if (it->second.second != shape) { *rc = t5_fail(h, VC_E_INVALID, "weight '%s' has the wrong shape", key.c_str()); return nullptr; }
    return it->second.first;
}

int t5_resolve(vc_t5* h) {
    if (h->resolved) return VC_OK;
    const vc_t5_config& c = h->cfg;
    int rc = VC_OK;
    h->emb = need(h, "token_embedding.weight", {c.vocab, c.dim}, &rc);
    h->norm = need(h, "norm.weight", {c.dim}, &rc);
    h->layers.resize(c.num_layers);
    for (int i = 0; i < c.num_layers && rc == VC_OK; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        LayerW& w = h->layers[i];
        w.n1 = need(h, p + "norm1.weight", {c.dim}, &rc);
        w.q = need(h, p + "attn.q.weight", {c.dim_attn, c.dim}, &rc);
        w.k = need(h, p + "attn.k.weight", {c.dim_attn, c.dim}, &rc);
        w.v = need(h, p + "attn.v.weight", {c.dim_attn, c.dim}, &rc);
        w.o = need(h, p + "attn.o.weight", {c.dim, c.dim_attn}, &rc);
        w.n2 = need(h, p + "norm2.weight", {c.dim}, &rc);
        w.gate = need(h, p + "ffn.gate.0.weight", {c.dim_ffn, c.dim}, &rc);
        w.fc1 = need(h, p + "ffn.fc1.weight", {c.dim_ffn, c.dim}, &rc);
        w.fc2 = need(h, p + "ffn.fc2.weight", {c.dim, c.dim_ffn}, &rc);
        w.pos = need(h, p + "pos_embedding.embedding.weight", {c.num_buckets, c.num_heads}, &rc);
    }
    if (rc != VC_OK) return rc;
    h->resolved = true;
    return VC_OK;
}

void t5_free_ws(vc_t5* h) {
    if (h->arena) (void)hipFree(h->arena);
    if (h->bucket) (void)hipFree(h->bucket);
    h->arena = nullptr;
    h->bucket = nullptr;
    h->ws_B = h->ws_L = 0;
}

int t5_workspace(vc_t5* h, int B, int L, hipStream_t s) {
    if (h->ws_B == B && h->ws_L == L) return VC_OK;
    t5_free_ws(h);
    const vc_t5_config& c = h->cfg;
    const int64_t R = (int64_t)B * L;
    auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
    int64_t off = 0;
    auto take = [&](int64_t bytes) { int64_t o = off; off += al(bytes); return o; };
    const int64_t o_x = take(R * c.dim * 2), o_t = take(R * c.dim * 2), o_qkv = take(R * 3 * c.dim_attn * 2),
                  o_a = take(R * c.dim_attn * 2), o_u = take(R * c.dim_ffn * 2);
    int64_t widest = c.dim_ffn > 3 * c.dim_attn ? c.dim_ffn : 3 * c.dim_attn;
    if (c.dim > widest) widest = c.dim;
    (void)take(256 * widest * 2);      // the ping-pong GEMM reads (never stores) A rows up to the next multiple of 256
    if (hipMalloc(&h->arena, off) != hipSuccess) {
        (void)hipGetLastError();
        h->arena = nullptr;
        return t5_fail(h, VC_E_NOMEM, "hipMalloc of %lld-byte T5 workspace failed", (long long)off);
    }
    h->arena_bytes = off;
    h->x = h->arena + o_x; h->t = h->arena + o_t; h->qkv = h->arena + o_qkv; h->a = h->arena + o_a; h->u = h->arena + o_u;
    std::vector<int32_t> tab(2 * L - 1);
    for (int r = -(L - 1); r <= L - 1; ++r) tab[r + L - 1] = rel_bucket(r, c.num_buckets, c.max_distance);
    if (hipMalloc(&h->bucket, tab.size() * 4) != hipSuccess) {
        (void)hipGetLastError();
        t5_free_ws(h);
        return t5_fail(h, VC_E_NOMEM, "hipMalloc of the bucket table failed");
    }
    if (hipMemcpyAsync(h->bucket, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)      // `tab` is a stack object: finish the copy before it goes away
        return t5_fail(h, VC_E_HIP, "upload of the bucket table failed");
    h->ws_B = B; h->ws_L = L;
    return VC_OK;
}

VcGemmParams t5_gemm(const void* A, int64_t lda, const void* W, void* C, int64_t ldc, int M, int N, int K, int epi) {
    VcGemmParams p;
    memset(&p, 0, sizeof p);
    p.A = A; p.lda = lda; p.W = W; p.ldw = K; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.epilogue = epi;
    p.valid_rows = -1; p.a_rows_padded = 1;
    return p;
}

#define T5CHK(h, expr)                                                                              \
    do {                                                                                            \
        int rc_ = (expr);                                                                           \
        if (rc_ != VC_OK) return t5_fail(h, rc_, "%s failed with %d (%s)", #expr, rc_, hipGetErrorString(hipGetLastError())); \
    } while (0)

}  // namespace

extern "C" {

int vc_t5_create(const vc_t5_config* cfg, vc_t5** out) {
    if (!cfg || !out) return t5_fail(nullptr, VC_E_INVALID, "vc_t5_create: null argument");
    const vc_t5_config& c = *cfg;
    if (c.vocab <= 0 || c.dim <= 0 || c.dim % 64 || c.dim_attn <= 0 || c.dim_ffn <= 0 || c.dim_ffn % 64 || c.num_heads <= 0 ||
        c.num_layers <= 0 || c.num_buckets < 4 || c.num_buckets % 2 || c.max_distance <= c.num_buckets / 4)
        return t5_fail(nullptr, VC_E_INVALID, "vc_t5_create: bad configuration");
    if (c.dim_attn != c.num_heads * 64)
        return t5_fail(nullptr, VC_E_UNSUPPORTED, "vc_t5_create: head dim %d (only 64 is built: umT5-XXL has 64 x 64)",
                       c.dim_attn / c.num_heads);
    vc_t5* h = new vc_t5();
    h->cfg = c;
    if (h->cfg.eps <= 0.f) h->cfg.eps = 1e-6f;
    *out = h;
    return VC_OK;
}

int vc_t5_load_weight(vc_t5* h, const char* key, const void* dev_ptr, int ndim, const int64_t* shape) {
    if (!h || !key || !dev_ptr || ndim <= 0 || ndim > 2 || !shape) return t5_fail(h, VC_E_INVALID, "vc_t5_load_weight: bad argument");
    h->slots[key] = {dev_ptr, std::vector<int64_t>(shape, shape + ndim)};
    h->resolved = false;
    return VC_OK;
}

int vc_t5_encode(vc_t5* h, const int32_t* ids, const int32_t* mask, void* out, int B, int L, void* stream) {
    if (!h || !ids || !out) return t5_fail(h, VC_E_INVALID, "vc_t5_encode: null argument");
    if (B <= 0 || B > 64 || L <= 0 || L % 64 || L > 64 * T5_MAXJ)
        return t5_fail(h, VC_E_INVALID, "vc_t5_encode: B=%d L=%d (L must be a multiple of 64, at most %d)", B, L, 64 * T5_MAXJ);
    { int r = t5_resolve(h); if (r != VC_OK) return r; }
    hipStream_t s = (hipStream_t)stream;
    { int r = t5_workspace(h, B, L, s); if (r != VC_OK) return r; }
    const vc_t5_config& c = h->cfg;
    const int R = B * L, d = c.dim, da = c.dim_attn, f = c.dim_ffn, H = c.num_heads;
    hipLaunchKernelGGL(t5_embed_kernel, dim3(grid_for((int64_t)R * d / 8, 256)), dim3(256), 0, s, ids, (const bf16_t*)h->emb,
                       (bf16_t*)h->x, R, d / 8, c.vocab);
    auto rms = [&](const void* x, const void* w, void* y) {
        hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3((R + 3) / 4), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y,
                           R, d, c.eps);
    };
    for (int i = 0; i < c.num_layers; ++i) {
        const LayerW& w = h->layers[i];
        rms(h->x, w.n1, h->t);
        {   // q | k | v into [R, 3 da] (one grouped launch of three problems sharing A)
            VcGemmParams g = t5_gemm(h->t, d, w.q, h->qkv, 3 * (int64_t)da, R, da, d, VC_EPI_BIAS);
            g.ngroups = 3;
            g.Wg[0] = w.k; g.Cg[0] = (char*)h->qkv + (int64_t)da * 2;
            g.Wg[1] = w.v; g.Cg[1] = (char*)h->qkv + (int64_t)2 * da * 2;
            T5CHK(h, vc_launch_gemm(g, s));
        }
        hipLaunchKernelGGL(t5_attention_kernel, dim3(grid_for((int64_t)B * H * L, 4)), dim3(256), 0, s, (const bf16_t*)h->qkv,
                           (bf16_t*)h->a, (const bf16_t*)w.pos, h->bucket, mask, B, H, L, da);
        {   // x += o(attn)
            VcGemmParams g = t5_gemm(h->a, da, w.o, h->x, d, R, d, da, VC_EPI_BIAS_RESID);
            g.resid = h->x; g.ldr = d;
            T5CHK(h, vc_launch_gemm(g, s));
        }
        rms(h->x, w.n2, h->t);
        {   // u = fc1(t) ; u = gelu(gate(t)) * u ; x += fc2(u)
            VcGemmParams g = t5_gemm(h->t, d, w.fc1, h->u, f, R, f, d, VC_EPI_BIAS);
            T5CHK(h, vc_launch_gemm(g, s));
            g = t5_gemm(h->t, d, w.gate, h->u, f, R, f, d, VC_EPI_GELU_MUL);
            g.resid = h->u; g.ldr = f;
            T5CHK(h, vc_launch_gemm(g, s));
            g = t5_gemm(h->u, f, w.fc2, h->x, d, R, d, f, VC_EPI_BIAS_RESID);
            g.resid = h->x; g.ldr = d;
            T5CHK(h, vc_launch_gemm(g, s));
        }
    }
    rms(h->x, h->norm, out);
    if (hipGetLastError() != hipSuccess) return t5_fail(h, VC_E_HIP, "vc_t5_encode: kernel launch failed");
    return VC_OK;
}

int vc_t5_relative_bucket(int rel, int num_buckets, int max_distance) {
    if (num_buckets < 4 || num_buckets % 2 || max_distance <= num_buckets / 4) return -1;
    return rel_bucket(rel, num_buckets, max_distance);
}

const char* vc_t5_last_error(const vc_t5* h) { return h ? h->err.c_str() : g_t5_create_error.c_str(); }

int64_t vc_t5_workspace_bytes(const vc_t5* h) { return h ? h->arena_bytes : 0; }

void vc_t5_destroy(vc_t5* h) {
    if (!h) return;
    t5_free_ws(h);
    delete h;
}

}  // extern "C"
