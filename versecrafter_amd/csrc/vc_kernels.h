// Internal launcher interface of libvcengine (host side).  Every launcher enqueues on the
// given HIP stream, allocates nothing, never synchronises, and returns 0 or a VC_E_* code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// error codes (mirrored in include/vcengine.h)
#define VC_OK 0
#define VC_E_INVALID (-1)
#define VC_E_HIP (-2)
#define VC_E_STATE (-3)
#define VC_E_NOMEM (-4)
#define VC_E_UNSUPPORTED (-5)

// ---- GEMM: C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N]) ------------------------------
enum VcEpilogue {
    VC_EPI_BIAS = 0,        // C = acc + bias
    VC_EPI_BIAS_GELU = 1,   // C = gelu_tanh(acc + bias)
    VC_EPI_BIAS_RESID = 2,  // C = resid + (acc + bias)
    VC_EPI_BIAS_GATE_RESID = 3,  // C = resid + (acc + bias) * gate[b, n]  (+ hint * hint_scale)
    VC_EPI_GELU_MUL = 4,    // C = gelu_tanh(acc + bias) * resid      (T5 gated-GELU feed-forward: gate(x) * fc1(x))
};

struct VcGemmParams {
    const void* A;    int64_t lda;   // bf16 [M, K]
    const void* W;    int64_t ldw;   // bf16 [N, K]  (nn.Linear weight layout)
    void* C;          int64_t ldc;   // bf16 [M, N]
    const void* bias;                // bf16 [N] or nullptr
    int M, N, K;
    int epilogue;
    const void* resid; int64_t ldr;  // bf16 [M, N] (may alias C)
    const void* gate;  int64_t gate_bstride;  // bf16, gate[b*gate_bstride + n], b = m / rows_per_batch
    const void* hint;  int64_t ldh;  float hint_scale;  // optional second residual (GATE_RESID only)
    // grouped launch (q/k/v projections): up to 3 problems sharing A, M, N, K, ld*; group g > 0 uses Wg/biasg/Cg
    int ngroups;                     // 0 or 1 -> single problem
    const void* Wg[2]; const void* biasg[2]; void* Cg[2];
    int rows_per_batch;              // rows per sample (for gate and valid_rows); 0 -> M
    int a_rows_padded;               // rows of A up to the next multiple of 256 are readable (engine workspace)
    int valid_rows;                  // >= 0: rows with (m % rows_per_batch) >= valid_rows are written as 0; < 0: off
    int tile;                        // kernel selection for tests / tuning (0 = auto; see vc_launch_gemm)
    int tile_map;                    // ping-pong kernel: 0 = XCD-contiguous bands (production), 1 = all XCDs on one 64-row super-band (A/B, tile id 6)
    // fp8 operands (BASELINE config 5's dtype; block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 at unit block scales): A and W hold OCP
    // e4m3 bytes ([M, K] / [N, K]; lda / ldw in elements = bytes), C = (A W^T) * a_scale[m] * w_scale[n], then the epilogue as for bf16.
    // Ping-pong kernel only: M padded / % 256, N % 256, K % 256.
    int fp8;
    const float* a_scale;            // [M] per row (token) of A
    const float* w_scale;            // [N] per row (output channel) of W; groups g > 0: w_scaleg[g - 1]
    const float* w_scaleg[2];
};
int vc_launch_gemm(const VcGemmParams& p, hipStream_t stream);
bool vc_gemm_fp8_eligible(const VcGemmParams& p);     // with p.fp8 operands in place (A, W e4m3): would vc_launch_gemm take the shape?
int vc_launch_layernorm_q8(const void* x, void* q, float* qscale, int rows, int dim, int rows_per_batch, float eps, int mode,
                           const void* p0, const void* p1, int64_t p_bstride, hipStream_t stream);
int vc_launch_quantize_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int M, int K, hipStream_t stream);

// ---- attention: out[b, i, h, :] = softmax(q k^T * scale) v ----------------------------------
struct VcAttnParams {
    const void* q; int64_t q_bs, q_ts, q_hs;   // element strides: batch, token, head (D=128 contiguous)
    const void* k; int64_t k_bs, k_ts, k_hs;
    const void* v; int64_t v_bs, v_ts, v_hs;
    void* out;     int64_t o_bs, o_ts, o_hs;
    int B, H, Lq, Lk;
    int k_len;          // keys >= k_len are masked (k_len <= Lk); 0 -> Lk
    float scale;        // 1/sqrt(D)
    // optional segmented token axis (Ulysses receive layout [P_src][...][L/P][...]): token t lives at
    // (t / seg_len) * *_ss + (t % seg_len) * *_ts ; seg_len == 0 -> plain t * *_ts
    int seg_len;
    int64_t q_ss, k_ss, v_ss, o_ss;
    // identical trailing keys (zero-padded prompt positions): per batch b the keys pad_from[b] .. Lk-1 are equal rows and are
    // folded into one key with multiplicity Lk - pad_from[b]  (pad_merge != 0; B <= 8; pad_from[b] < 0 or >= Lk-1: nothing to fold)
    int pad_merge;
    int pad_from[8];
    // kernel selection for tests / tuning (a launch parameter, no global state): 0 = the launcher's choice, 32 = the
    // v_mfma_f32_32x32x16_bf16 pipelined kernel, 16 = the v_mfma_f32_16x16x32_bf16 one (attention16.hip; plain layout only)
    int variant;
    // optional: log2-domain log-sum-exp of every query row, lse[(b * H + head) * Lq + q] = log2(sum_k exp2(s_qk * scale * log2 e))
    // (float32).  Ring attention merges the outputs of several key blocks with it; only the 16x16x32 kernel writes it (plain layout).
    float* lse;
};
int vc_launch_attention(const VcAttnParams& p, hipStream_t stream);
int vc_launch_attention_mfma16(const VcAttnParams& p, hipStream_t stream);     // attention16.hip
// out[b, q, h, :] = sum_r w_r part_r[b, q, h, :],  w_r = exp2(lse_r - log2 sum_r' exp2(lse_r'))   (parts: bf16 [B][Lq][H][128]
// contiguous, lse: float32 [B][H][Lq]; a block that saw no key carries lse = -inf and is ignored); out with element strides
struct VcAttnMergeParams {
    const void* part[8]; const float* lse[8]; int R;
    void* out; int64_t o_bs, o_ts, o_hs;
    int B, H, Lq;
};
int vc_launch_attention_merge(const VcAttnMergeParams& p, hipStream_t stream);
int vc_launch_attention_stream(const VcAttnParams& p, hipStream_t stream);     // attention_stream.hip (5-8 key tiles, padded-key folding)

// ---- fp8 self-attention (attention_fp8.hip; opt-in mode of this build, BASELINE config 5's dtype) ----------------------------------
// q, k, v bf16 with element strides (batch, token, head), D = 128 contiguous; out bf16.  ws: 256-byte aligned device workspace of at least
// vc_attention_fp8_workspace_bytes(B, H, Lq, Lk) bytes (the e4m3 copies of q, k, v^T and their E8M0 block scales).
// pmode 1: P's e4m3 byte from the piecewise-linear 2^x (no exponential); pmode 0: v_exp_f32 + v_cvt_pk_fp8_f32.
struct VcAttnFp8Params {
    const void* q; int64_t q_bs, q_ts, q_hs;
    const void* k; int64_t k_bs, k_ts, k_hs;
    const void* v; int64_t v_bs, v_ts, v_hs;
    void* out;     int64_t o_bs, o_ts, o_hs;
    int B, H, Lq, Lk, k_len;
    float scale;
    int pmode;
    void* ws;
    // filled by the launcher
    int64_t off_q8, off_qs, off_kv;
    float qfold;
};
int64_t vc_attention_fp8_workspace_bytes(int B, int H, int Lq, int Lk);
int vc_launch_attention_fp8(const VcAttnFp8Params& p, int64_t ws_bytes, hipStream_t stream);          // quantise + attend
int vc_launch_attention_fp8_quant(VcAttnFp8Params p, int64_t ws_bytes, hipStream_t stream);           // the two halves (tests, profiling)
int vc_launch_attention_fp8_core(VcAttnFp8Params p, int64_t ws_bytes, hipStream_t stream);

// ---- row kernels --------------------------------------------------------------------------
// y = LN(x) * (1 + scale[b]) + shift[b]        (mode 0, WT.py:591,603; head WT.py:643)
// y = LN(x) * w + bias                        (mode 1, norm3, WT.py:548-550)
int vc_launch_layernorm(const void* x, void* y, int rows, int dim, int rows_per_batch, float eps, int mode,
                        const void* p0, const void* p1, int64_t p_bstride, hipStream_t stream);

// in place on a [rows, dim] slice with row stride ld:  x = rmsnorm(x) * w ; optional 3-axis RoPE
struct VcRopeGrid { int F, H, W; int token_offset; int rows_per_batch; };
int vc_launch_rmsnorm_rope(void* x, int64_t ld, int rows, int dim, const void* w, float eps,
                           const float2* rope_table /*[1024][64] (cos,sin)*/, const VcRopeGrid* grid,
                           hipStream_t stream);

// self-attention front in one pass over qkv [rows][3 dim]: RMSNorm + RoPE of q and k (in place when send == nullptr), or q, k
// (normed, rotated) and v written straight into the Ulysses exchange layout send[3][B][P][Lloc][dim / P] (rows = B * Lloc, Lloc = grid rows_per_batch)
int vc_launch_qkv_front(void* qkv, int rows, int dim, const void* wq, const void* wk, float eps, const float2* rope_table,
                        const VcRopeGrid* grid, void* send, int P, hipStream_t stream);

// ---- small kernels ------------------------------------------------------------------------
// A[b*Lrows + i, c*4 + p*2 + q] = x[b, c, f, 2h+p, 2w+q] for token tok = tok_offset + i = (f, h, w)
// (rows with tok >= F*H2*W2 are zero).  Lrows / tok_offset select a rank's sequence chunk (VC.py:366-367).
int vc_launch_patchify(const void* x, void* A, int B, int C, int T, int H, int W, int Lrows, int tok_offset,
                       hipStream_t s);
// out[b, c, f, 2h+q, 2w+r] = y[row(b, tok), (q*2+r)*C + c],  row = (tok / Lloc) * B * Lloc + b * Lloc + tok % Lloc
// (the [P][B][Lloc] row order a flat all-gather of per-rank [B, Lloc, 4C] buffers produces; P = 1: b*Lloc + tok)
int vc_launch_unpatchify(const void* y, void* out, int B, int C, int T, int H2, int W2, int Lloc, hipStream_t s);
// fp32 "small-M" linear: y[b, n] = act_in(x[b, :]) . W[n, :] + bias[n];  W,bias bf16; x,y fp32
int vc_launch_small_linear(const float* x, const void* W, const void* bias, float* y, int B, int N, int K,
                           int silu_input, hipStream_t s);
// sinusoidal_embedding_1d(freq_dim, t) -> fp32 [B, freq_dim]   (WT.py:39-49)
int vc_launch_sinusoid(const float* t, float* out, int B, int freq_dim, hipStream_t s);
// out_bf16[b, j, :] = bf16(mod[j, :] + bf16(e[b*e_bstride + j*e_jstride + :]))   (WT.py:588 / 641; e_jstride 0 = broadcast)
int vc_launch_modulation(const void* mod, const float* e, void* out, int B, int J, int dim, int64_t e_bstride,
                         int64_t e_jstride, hipStream_t s);
// elementwise bf16: out = a + b * s ; out = a - b
int vc_launch_axpy(const void* a, const void* b, void* out, float s, int64_t n, hipStream_t st);
int vc_launch_sub(const void* a, const void* b, void* out, int64_t n, hipStream_t st);
// zero-pad text rows: dst[b, i, :] = i < len[b] ? src_b[i, :] : 0      (VC.py:358-363)
int vc_launch_pad_rows(const void* src, void* dst, int len, int total, int dim, hipStream_t st);
int vc_launch_unipc_update(const void* noise_uncond, const void* noise_cond, const void* sample, const void* last,
                           const void* m0, const void* m1, void* x0_out, void* samp_out, void* next_out, int64_t n,
                           const float* sc, int flags, hipStream_t st);
int vc_launch_geoada_context(const void* z, const void* mask, int mask_is_f32, void* out, int T, int h, int w, int F,
                             hipStream_t st);
// Ulysses exchange buffers (layout contract: versecrafter_amd/dist.py).  q|k|v reach the send layout [3][B][P_dst][Lloc][d/P] through
// vc_launch_qkv_front; the return path needs:
//   unpack recv [B][P_src][Lloc][d/P] -> attn [M = B * Lloc, d]
int vc_launch_sp_unpack_o(const void* recv, void* attn, int M, int rows_per_batch, int d, int P, hipStream_t st);
// one idle wave holds the stream for `usec` microseconds (what-if timing only)
int vc_launch_delay(double usec, hipStream_t st);

// ---- RCCL transport of the Ulysses exchange (sp_rccl.hip); librccl is bound with dlopen at first use ----
#define VC_RCCL_UNIQUE_ID_BYTES 128
struct VcComm;
int vc_comm_available();                 // librccl can be bound (dlopen + every symbol) -- no device work
int vc_comm_unique_id(void* out128);
int vc_comm_create(VcComm** out, const void* id128, int world, int rank);   // blocks until all ranks have joined
int vc_comm_ranks(const VcComm* c);
void vc_comm_destroy(VcComm* c);
int vc_comm_all_to_all(VcComm* c, const void* send, void* recv, int64_t bytes_per_peer, hipStream_t s);
int vc_comm_all_to_all_n(VcComm* c, const void* send, void* recv, int64_t bytes_per_peer, int n, hipStream_t s);   // n slabs, one group
int vc_comm_all_gather(VcComm* c, const void* send, void* recv, int64_t bytes, hipStream_t s);
int vc_comm_all_to_all_sub_n(VcComm* c, const void* send, void* recv, int64_t bytes_per_peer, int n, int first, int count, hipStream_t s);
int vc_comm_sendrecv(VcComm* c, const void* send, int dst, void* recv, int src, int64_t bytes, hipStream_t s);
const char* vc_comm_error();    // message of the calling thread's last failed vc_comm_* call
