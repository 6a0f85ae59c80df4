/* libvcengine -- C ABI of the MI355X-native VerseCrafter denoising engine.
 *
 * The drop-in boundary for the hot path of ztitomir/VerseCrafter: one denoise step =
 * VerseCrafterWanTransformer3DModel.forward (Wan2.1 DiT + GeoAdapter), as driven by
 * WanVerseCrafterPipeline.__call__'s loop.  The reference is pure Python and has no FFI of its own;
 * each entry point below cites the reference interface it stands in for (paths relative to the
 * reference root; WT.py = versecrafter/models/wan_transformer3d.py,
 * VC.py = versecrafter/models/wan_transformer3d_versecrafter.py,
 * PIPE.py = versecrafter/pipeline/pipeline_wan_versecrafter.py).  The ctypes binding a maintainer would add
 * is shown in INTEGRATION.md and shipped as versecrafter_amd/_lib.py.
 *
 * Conventions
 *   - plain C types only; tensors are raw DEVICE pointers (bf16 unless stated), row-major, caller-owned.
 *   - the library borrows weight pointers until vc_destroy and owns only its workspace.
 *   - every call returns 0 (VC_OK) or a negative VC_E_* code; vc_last_error gives the message.
 *   - kernels are enqueued on the given hipStream_t (void*; NULL = default stream); no host sync,
 *     no allocation inside vc_forward.  Handles are not thread-safe.
 *   - gfx950 only.  There is no CPU fallback: without a HIP device every compute entry fails with VC_E_HIP.
 */
#ifndef VCENGINE_H
#define VCENGINE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VC_OK 0
#define VC_E_INVALID (-1)     /* bad argument / shape                               */
#define VC_E_HIP (-2)         /* HIP runtime error (message has hipGetErrorString)   */
#define VC_E_STATE (-3)       /* call order violated (e.g. forward before prepare)   */
#define VC_E_NOMEM (-4)
#define VC_E_UNSUPPORTED (-5) /* shape outside what the kernels implement            */

#define VC_ABI_VERSION 3
#define VC_MAX_GEOADA_LAYERS 64

typedef struct vc_engine vc_engine;

/* Hyper-parameters of VerseCrafterWanTransformer3DModel.__init__ (VC.py:153-170). */
typedef struct vc_config {
    int32_t dim, ffn_dim, num_heads, num_layers;
    int32_t in_dim, out_dim, geoada_in_dim;
    int32_t text_dim, text_len, freq_dim;
    float eps;
    int32_t num_geoada_layers;                       /* 0 -> range(0, num_layers, 2)  (VC.py:175) */
    int32_t geoada_layers[VC_MAX_GEOADA_LAYERS];
} vc_config;

/* vc_forward flags: the TeaCache branches of VC.py:384-411 */
#define VC_FWD_RUN_MAIN_BLOCKS 1u   /* run self.blocks (should_calc)                         */
#define VC_FWD_STORE_RESIDUAL 2u    /* keep x_out - x_in  (previous_residual_cond)           */
#define VC_FWD_USE_RESIDUAL 4u      /* x = x + previous_residual instead of the main blocks */
#define VC_FWD_RESIDUAL_UNCOND 16u  /* with STORE / USE: the previous_residual_uncond slot (cond_flag=False, VC.py:391-394,
                                       408-411) instead of previous_residual_cond.  A stored residual of B' samples serves a
                                       later USE with B <= B' samples by its LAST B samples (previous_residual[-B:], VC.py:396:
                                       what cfg_skip needs when it drops the unconditional half mid-sampling); the residuals
                                       survive vc_prepare_video as long as the token geometry is unchanged.  The first STORE
                                       of a slot allocates it (the only allocation vc_forward ever makes). */
#define VC_FWD_SHARED_CFG_INPUT 8u  /* caller asserts: every sample has the SAME x, t and geoada_context (the CFG pair of
                                       PIPE.py:878-887 differs only in the prompt): block 0 of both chains computes its
                                       self-attention half once; results are bit-identical to the unflagged call */

int vc_abi_version(void);
const char* vc_last_error(const vc_engine* h);     /* h may be NULL: last error of a failed vc_create */

/* VerseCrafterWanTransformer3DModel.__init__ (VC.py:151-201). */
int vc_create(const vc_config* cfg, vc_engine** out);
void vc_destroy(vc_engine* h);

/* load_state_dict (WT.py:1302-1311): key = reference state-dict name, shape = its shape.  dtype: 0 = bf16. */
int vc_load_weight(vc_engine* h, const char* key, const void* dev_ptr, int dtype, int ndim, const int64_t* shape);
/* number of state-dict keys still unset (0 = ready) */
int vc_missing_weights(const vc_engine* h);

/* self.freqs (WT.py:783-795; enable_riflex WT.py:873-888): HOST complex128 table [rows][cols] as (re, im) doubles */
int vc_set_rope_table(vc_engine* h, const double* cis, int rows, int cols);

/* enable_multi_gpus_inference (WT.py:901-921) + the sequence chunking of VC.py:269-270, 366-367.
 * The self-attention exchange (third-party usp_attn_forward, bound at WT.py:907-921) is delegated to the host
 * through two callbacks so that the collective itself stays in torch.distributed / RCCL:
 *   all_to_all(ctx, send, recv, bytes_per_peer, stream): peer r's slice is send[r*bytes_per_peer ...]
 *   all_gather(ctx, send, recv, bytes, stream)
 * One exchange of a block is several such all-to-alls on consecutive slabs of the engine's buffers -- one per (q|k|v tensor,
 * sample) before the attention, one per sample after it (layout: versecrafter_amd/dist.py) -- so that what arrives is the
 * attention kernel's plain [B][L][heads / world][128] layout; the callback is invoked once per slab.                       */
typedef int (*vc_all_to_all_fn)(void* ctx, const void* send, void* recv, int64_t bytes_per_peer, void* stream);
typedef int (*vc_all_gather_fn)(void* ctx, const void* send, void* recv, int64_t bytes, void* stream);
int vc_sp_init(vc_engine* h, int world, int rank, vc_all_to_all_fn a2a, vc_all_gather_fn ag, void* ctx);

/* The product transport for world > 1: RCCL over xGMI, owned by the engine (vc_sp_init(h, ncclUniqueId, rank,
 * world)).  Replaces set_multi_gpus_devices + usp_attn_forward's collectives (inference/versecrafter_inference.py:180;
 * WT.py:901-921; VC.py:432-433).  The engine creates ONE COMMUNICATOR PER BLOCK CHAIN (main blocks / GeoAdapter blocks) and
 * enqueues every exchange on the HIP stream of the chain that needs it, so the two chains overlap each other's exchanges
 * and nothing calls back into the host on the step path.  librccl.so.1 is bound with dlopen at the first of these calls (the
 * instance already loaded into the process is reused; VC_RCCL_LIB overrides the path).
 *   vc_rccl_available : binds librccl only.
 *   vc_rccl_unique_id : rank 0 fills out[128] (ncclGetUniqueId); the host ships the bytes to every rank (any side channel).
 *   vc_sp_init_rccl   : unique_ids = n_ids x 128 bytes, n_ids must be 2 (chain 0, chain 1); every rank of the world must
 *                       call it (ncclCommInitRank is a rendezvous).  flags: VC_SP_FORCE_EXCHANGE runs the exchange path
 *                       (pack, all-to-alls, attention, all-to-alls, unpack, all-gather) even at world == 1.  The slabs of one
 *                       exchange are enqueued as ONE RCCL group (one fused launch).
 *   vc_sp_comm_ranks  : ncclCommCount of chain 0's communicator (0: the RCCL transport is not active).
 *   vc_sp_all_to_all / vc_sp_all_gather : the engine's collectives in isolation (tests): byte buffers, chain 0 or 1. */
#define VC_RCCL_UNIQUE_ID_BYTES 128
#define VC_SP_FORCE_EXCHANGE 1u
/* Ulysses x ring hybrid (the reference's `--ulysses_degree U --ring_degree R`, CLI.py:59-62, inference.sh:66-67; third-party
 * xFuserLongContextAttention bound at WT.py:907-921).  After vc_sp_init / vc_sp_init_rccl on a world of U * R ranks, vc_sp_set_ring(R)
 * makes rank g * U + u exchange heads inside its Ulysses group of U neighbours (heads / U per rank, the group's U * Lloc tokens) and
 * pass K|V blocks round the ring of the R ranks that share u; the R partial attention outputs are merged by their log-sum-exps
 * (vc_op_attention_lse / vc_op_attention_merge).  Needed when num_heads is not divisible by the world size (Wan2.1-1.3B: 12 heads on 8
 * GPUs = U 4 x R 2); with R = 1 nothing changes.  On the RCCL transport both exchanges are grouped point-to-point on the world
 * communicator (callbacks may be NULL); on the callback transport:
 *   all_to_all_sub(ctx, send, recv, bytes_per_peer, first_rank, count, stream): slice j of send goes to rank first_rank + j, slice j of
 *                                                                               recv comes from it (count slices);
 *   sendrecv(ctx, send, dst_rank, recv, src_rank, bytes, stream).
 * vc_sp_init* reset the ring degree to 1. */
typedef int (*vc_all_to_all_sub_fn)(void* ctx, const void* send, void* recv, int64_t bytes_per_peer, int first_rank, int count, void* stream);
typedef int (*vc_sendrecv_fn)(void* ctx, const void* send, int dst_rank, void* recv, int src_rank, int64_t bytes, void* stream);
int vc_sp_set_ring(vc_engine* h, int ring_degree, vc_all_to_all_sub_fn all_to_all_sub, vc_sendrecv_fn sendrecv);
int vc_sp_ring_degree(const vc_engine* h);
int vc_rccl_available(void);      /* 0 when librccl can be bound on this rank (no device work): lets the ranks AGREE, before the
                                   * first step that can fail on one side only, whether the engine-owned transport is usable */
int vc_rccl_unique_id(void* out, int nbytes);
int vc_sp_init_rccl(vc_engine* h, int world, int rank, const void* unique_ids, int n_ids, uint32_t flags);
int vc_sp_comm_ranks(const vc_engine* h);
int vc_sp_all_to_all(vc_engine* h, int chain, const void* send, void* recv, int64_t bytes_per_peer, void* stream);
/* nslab all-to-alls on consecutive [world][bytes_per_peer] slabs as the step path issues them (ONE RCCL group on the engine-owned
 * transport, one callback per slab otherwise) */
int vc_sp_all_to_all_n(vc_engine* h, int chain, const void* send, void* recv, int64_t bytes_per_peer, int nslab, void* stream);
int vc_sp_all_gather(vc_engine* h, const void* send, void* recv, int64_t bytes, void* stream);
/* the hybrid's two exchanges in isolation (tests, bring-up probes): nslab all-to-alls among ranks first .. first + count - 1; the ring pass */
int vc_sp_all_to_all_sub(vc_engine* h, int chain, const void* send, void* recv, int64_t bytes_per_peer, int nslab, int first, int count,
                         void* stream);
int vc_sp_sendrecv(vc_engine* h, int chain, const void* send, int dst_rank, void* recv, int src_rank, int64_t bytes, void* stream);
/* What-if timing of ONE rank of a `world`-way run on a single GPU (tools/sim_sp_rank.py): every exchange is a local copy
 * followed by one idle wave that holds the chain's stream for (bytes leaving the rank) / egress_gbps -- the rank's compute
 * share, the pack / unpack passes, the two-chain schedule and the exposure of the wire time are real, the RESULTS ARE NOT
 * (no peer data).  Never used by the product path. */
int vc_sp_init_sim(vc_engine* h, int world, int rank, double egress_gbps);

/* Step-invariant part of forward, hoisted (once per video): geoada_patch_embedding (VC.py:262-267),
 * text_embedding (VC.py:358-363) and every block's cross-attention k/v (WT.py:421-422).
 *   geoada_context [B, geoada_in_dim, T, H, W]; text[i] = [text_lens[i], text_dim]; seq_len as PIPE.py:861-865. */
int vc_prepare_video(vc_engine* h, const void* geoada_context, const void* const* text, const int32_t* text_lens,
                     int B, int T, int H, int W, int seq_len, void* stream);

/* VerseCrafterWanTransformer3DModel.forward (VC.py:295-442) for the prepared video.
 *   x [B, in_dim, T, H, W] bf16, t [B] fp32 (device), out [B, out_dim, T, H, W] bf16.                        */
int vc_forward(vc_engine* h, const void* x, const float* t, void* out, float geoada_context_scale,
               uint32_t flags, void* stream);

/* time-embedding only: e0 [B, 6, dim] fp32 (device) for the TeaCache gate (WT.py:205-245), VC.py:347-354 */
int vc_time_embedding(vc_engine* h, const float* t, int B, float* e0_out, void* stream);

/* Live per-kernel-class timing for bench.py's roofline line: when enabled, every launch of a class inside
 * vc_forward is bracketed by HIP events on the launch stream.  vc_profile_read synchronises those events and
 * returns, per class, launch count, summed duration (ms), algorithmic FLOPs (GEMM 2MNK, attention 4 B H Lq Lk D)
 * and algorithmic bytes, then clears the records. */
#define VC_PROF_GEMM 0
#define VC_PROF_ATTN_SELF 1
#define VC_PROF_ATTN_CROSS 2
#define VC_PROF_ROW 3       /* LayerNorm / RMSNorm+RoPE row kernels */
#define VC_PROF_NCLASS 4
int vc_profile_enable(vc_engine* h, int on);
int vc_profile_read(vc_engine* h, int ncls, int64_t* count, double* ms, double* flops, double* bytes);

/* A new video through the same handle: forget the stored TeaCache residuals (TeaCache.reset() of the third-party class clears
 * previous_residual_cond / _uncond; here they live in the engine).  The slots stay allocated; VC_FWD_USE_RESIDUAL before the next
 * VC_FWD_STORE_RESIDUAL fails with VC_E_STATE instead of re-adding the previous video's residual. */
int vc_reset_residuals(vc_engine* h);
/* forwards served by a hipGraph replay since vc_create (small token counts; tests assert that a replay really happened) */
int64_t vc_graph_replays(const vc_engine* h);

/* bytes of library-owned device workspace currently allocated */
int64_t vc_workspace_bytes(const vc_engine* h);

/* ---- per-kernel entry points (parity tests call the hot kernels in isolation) ------------------------ */

/* nn.Linear (+ fused epilogue): C[M,N] = epi(A[M,K] . W[N,K]^T + bias).  epilogue: 0 bias, 1 bias+gelu(tanh),
 * 2 resid + y, 3 resid + y*gate[b] (+ hint*hint_scale).  rows_per_batch selects gate row b = m / rows_per_batch. */
int vc_op_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias,
                    int M, int N, int K, int epilogue, const void* resid, int64_t ldr, const void* gate,
                    int64_t gate_bstride, int rows_per_batch, const void* hint, int64_t ldh, float hint_scale,
                    int tile /*0 auto, 1: 128x128, 2: 256x256 two-stage, 3: the same with 64-bit DMA addresses, 4: ping-pong, 5: one wave per SIMD (4, 5: A readable up to the next multiple of 256 rows)*/, void* stream);

/* attention() of videox_fun as called at WT.py:394-399 / 425-430.  q,k,v,out: [B, L, H, 128] with element
 * strides (batch, token, head); keys >= k_len masked (0 = none). */
int vc_op_attention(const void* q, const void* k, const void* v, void* out, int B, int H, int Lq, int Lk,
                    const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                    const int64_t* o_strides, int k_len, float scale, void* stream);

/* vc_op_attention with the MFMA shape of the pipelined kernel forced (tests / A-B tools): variant 32 = v_mfma_f32_32x32x16_bf16,
 * 16 = v_mfma_f32_16x16x32_bf16, 0 = the library's default.  Same contract and tolerance as vc_op_attention. */
int vc_op_attention_variant(const void* q, const void* k, const void* v, void* out, int B, int H, int Lq, int Lk,
                            const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                            const int64_t* o_strides, int k_len, float scale, int variant, void* stream);

/* fp8 self-attention (this build; BASELINE config 5 names "fp8 MFMA" -- the reference computes attention in bf16 through flash-attn,
 * WT.py:394-399).  Same contract as vc_op_attention for the bf16 inputs and the bf16 output; inside, q, k, v AND the softmax weights are
 * OCP e4m3 with one E8M0 power-of-two scale per 32 elements along each contraction (MX-style), both products run on
 * v_mfma_scale_f32_32x32x64_f8f6f4 with fp32 accumulation (csrc/attention_fp8.hip; CPU restatement oracle/attn_fp8_oracle.py).
 *   pmode 1: the e4m3 byte of a weight is round(8 log2 w + 56), i.e. the piecewise-linear 2^x -- no exponential; pmode 0: v_exp_f32.
 *   stage 0: quantise + attend; 1: quantise q / k / v into the workspace only; 2: attend on a workspace stage 1 filled (same shape).
 *   workspace: 256-byte aligned device memory of >= vc_op_attention_fp8_workspace_bytes(B, H, Lq, Lk) bytes.
 * PARITY: "within the fp8 error of exact attention" (bounds in tests/test_gpu_attention_fp8.py); the definition itself is pinned. */
int64_t vc_op_attention_fp8_workspace_bytes(int B, int H, int Lq, int Lk);
int vc_op_attention_fp8(const void* q, const void* k, const void* v, void* out, int B, int H, int Lq, int Lk, const int64_t* q_strides,
                        const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, int k_len, float scale, int pmode,
                        int stage, void* workspace, int64_t workspace_bytes, void* stream);

/* Ring attention building blocks (the ring half of the reference's Ulysses x ring hybrid, third-party xFuserLongContextAttention bound at
 * WT.py:907-921; CLI.py:59-62): attention over ONE block of keys that also returns, per query row, the log2-domain log-sum-exp of its
 * logits (lse float32 [B][H][Lq]); and the merge of R such partial outputs (bf16 [B][Lq][H][128] contiguous) into
 * out = sum_r w_r part_r, w_r = exp2(lse_r - log2 sum exp2(lse)) -- equal to attention over the concatenated keys up to bf16 rounding of
 * the parts.  HOST arrays of R device pointers. */
int vc_op_attention_lse(const void* q, const void* k, const void* v, void* out, float* lse, int B, int H, int Lq, int Lk,
                        const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, int k_len,
                        float scale, void* stream);
int vc_op_attention_merge(const void* const* parts, const float* const* lses, int R, void* out, int B, int H, int Lq,
                          const int64_t* o_strides, void* stream);

/* The same attention on the Ulysses receive layout (token axis in segments of seg_len tokens, one per source rank):
 * strides are {batch, token within a segment, head, segment}; token t = (t / seg_len, t % seg_len).  Lq = Lk = L. */
int vc_op_attention_segmented(const void* q, const void* k, const void* v, void* out, int B, int H, int L,
                              const int64_t* q_strides4, const int64_t* k_strides4, const int64_t* v_strides4,
                              const int64_t* o_strides4, int seg_len, int k_len, float scale, void* stream);

/* Cross-attention over a zero-padded prompt (WT.py:425-430 after VC.py:358-363): per batch b the keys pad_from[b] .. Lk-1 are
 * IDENTICAL rows (HOST int32 array of B <= 8 entries); they are folded into one key of multiplicity Lk - pad_from[b]. */
int vc_op_attention_padmerge(const void* q, const void* k, const void* v, void* out, int B, int H, int Lq, int Lk,
                             const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                             const int64_t* o_strides, const int32_t* pad_from, float scale, void* stream);

/* WanLayerNorm + modulate (mode 0: y = LN(x)*(1+p0[b])+p1[b]) or affine (mode 1: y = LN(x)*p0+p1). */
int vc_op_layernorm(const void* x, void* y, int rows, int dim, int rows_per_batch, float eps, int mode,
                    const void* p0, const void* p1, int64_t p_bstride, void* stream);

/* WanRMSNorm (+ rope_apply when table != NULL), in place.  table: DEVICE float2 [1024][64] (cos, sin);
 * grid = {F, H, W, token_offset, rows_per_batch}. */
int vc_op_rmsnorm_rope(void* x, int64_t ld, int rows, int dim, const void* w, float eps, const void* table,
                       const int32_t* grid5, void* stream);

/* Self-attention front (WT.py:385-392) in one pass over qkv [rows][3 dim]: WanRMSNorm + rope_apply of q and k, in place
 * (send == NULL), or q, k (normed, rotated) and v written into the Ulysses exchange layout send[3][B][P][Lloc][dim / P] (rows = B * Lloc, Lloc = grid rows_per_batch). */
int vc_op_qkv_front(void* qkv, int rows, int dim, const void* wq, const void* wk, float eps, const void* table,
                    const int32_t* grid5, void* send, int P, void* stream);

/* Control-map front-end of one sample (PIPE.py:440-488, geoada_encode_masks + geoada_latent, ref_images = None):
 * out [128,T,h,w] bf16 = concat( z [64,T,h,w] bf16 ,  nearest-exact frame resize of the 8x8 pixel-unshuffle of
 * mask[0] [F,H,W] (bf16, or fp32 when mask_is_f32) ).  T = (F+3)/4, h = 2*(H/16) = H/8, w = W/8. */
int vc_op_geoada_context(const void* z, const void* mask, int mask_is_f32, void* out, int T, int h, int w, int F, int H,
                         int W, void* stream);

/* One sampler update on the latent (PIPE.py:903-909; third-party FlowUniPCMultistepScheduler.step, restated in
 * versecrafter_amd/utils/fm_solvers_unipc.py) fused into one pass over n bf16 elements:
 *   noise = u + g (c - u)  [flags&1]; x0 = sample - sigma_t noise; corrected sample [flags&2, order 2: flags&4];
 *   next sample = UniPC predictor [order 2: flags&8].  Each op rounds to bf16 exactly as the torch formulation does.
 * scalars13 (HOST): {guidance, sigma_t, ca, cb, cBh, rk_c, rho0_c, rho1_c, pa, pb, pBh, rk_p, rho_p}.
 * m0 / m1: the previous two x0 predictions; last: the previous (corrected) sample; samp_out may be NULL. */
int vc_op_unipc_update(const void* noise_uncond, const void* noise_cond, const void* sample, const void* last,
                       const void* m0, const void* m1, void* x0_out, void* samp_out, void* next_out, int64_t n,
                       const float* scalars13, int flags, void* stream);

/* ---- umT5 text encoder --------------------------------------------------------------------------
 * Replaces WanT5EncoderModel (videox_fun, un-vendored; origin Wan2.1 wan/modules/t5.py) as the reference's pipeline
 * uses it: `self.text_encoder(ids, attention_mask=mask)[0]` (pipeline_wan_versecrafter.py:273), constructed at
 * inference/versecrafter_inference.py:243-249 with config/wan2.1/wan_civitai.yaml:14-26.
 * Weights are borrowed device pointers (bf16) addressed by the upstream state-dict keys:
 *   token_embedding.weight [vocab,dim], norm.weight [dim], and per layer i = blocks.i.:
 *   norm1.weight, attn.{q,k,v}.weight [dim_attn,dim], attn.o.weight [dim,dim_attn], norm2.weight,
 *   ffn.gate.0.weight, ffn.fc1.weight [dim_ffn,dim], ffn.fc2.weight [dim,dim_ffn],
 *   pos_embedding.embedding.weight [num_buckets,num_heads]. */
typedef struct vc_t5_config {
    int32_t vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, max_distance;
    float eps;
} vc_t5_config;
typedef struct vc_t5 vc_t5;
int vc_t5_create(const vc_t5_config* cfg, vc_t5** out);
int vc_t5_load_weight(vc_t5* h, const char* key, const void* dev_ptr, int ndim, const int64_t* shape);
/* ids, mask: DEVICE int32 [B, L] (mask 1 = token, 0 = padding; NULL = no padding); out: DEVICE bf16 [B, L, dim].
 * L must be a multiple of 64 (the reference pads every prompt to text_length = 512). */
int vc_t5_encode(vc_t5* h, const int32_t* ids, const int32_t* mask, void* out, int B, int L, void* stream);
/* host-only: bidirectional T5 bucket of rel = key - query (what the encoder's table holds); -1 on bad arguments */
int vc_t5_relative_bucket(int rel, int num_buckets, int max_distance);
const char* vc_t5_last_error(const vc_t5* h);
int64_t vc_t5_workspace_bytes(const vc_t5* h);
void vc_t5_destroy(vc_t5* h);

/* ---- Wan2.1 video VAE ----------------------------------------------------------------------------
 * Replaces AutoencoderKLWan (videox_fun, un-vendored; origin Wan2.1 wan/modules/vae.py) as the reference's pipeline uses it:
 * `vae.encode(frames)[0].mode()` for each control video (pipeline_wan_versecrafter.py:397-438) and `vae.decode(latents).sample`
 * (pipeline_wan_versecrafter.py:550-555), constructed at inference/versecrafter_inference.py:220-236 with
 * config/wan2.1/wan_civitai.yaml:8-13.  PARITY UNPINNED: restated from the published architecture (oracle/vae_oracle.py).
 * Weights are borrowed device pointers (bf16) addressed by the upstream state-dict keys (without VideoX-Fun's "model." prefix):
 *   encoder.conv1, encoder.downsamples.N.{residual.{0,3}.gamma, residual.{2,6}, shortcut, resample.1, time_conv},
 *   encoder.middle.{0,2}.*, encoder.middle.1.{norm.gamma, to_qkv, proj}, encoder.head.{0.gamma, 2}, conv1, conv2, decoder.* ;
 * they are re-packed once (tap-major, channels padded to 64) into library-owned memory at the first encode / decode.
 * The workspace (five buffers of the largest activation: ~60 GB for an 81-frame 480p decode) is allocated at the first call that
 * needs it and kept -- a video takes four encodes and one decode, and allocating tens of GB costs seconds -- until
 * vc_vae_release_workspace or vc_vae_destroy; encode / decode return without a host sync. */
typedef struct vc_vae_config {
    int32_t dim, z_dim;                 /* 96, 16 */
    int32_t dim_mult[4];                /* 1, 2, 4, 4 */
    int32_t num_res_blocks;             /* 2 (the decoder uses one more per level) */
    int32_t temporal_downsample[3];     /* 0, 1, 1 */
} vc_vae_config;
typedef struct vc_vae vc_vae;
int vc_vae_create(const vc_vae_config* cfg, vc_vae** out);
int vc_vae_load_weight(vc_vae* h, const char* key, const void* dev_ptr, int ndim, const int64_t* shape);
int vc_vae_missing_weights(const vc_vae* h);
/* x: DEVICE bf16 [3][F][H][W] in [-1, 1], F = 1 + 4n, H and W multiples of 16; out: DEVICE bf16 [z_dim][1 + n][H/8][W/8] = the
 * posterior mean, normalised with the published per-channel latent mean / std (what `.mode()` returns upstream). */
int vc_vae_encode(vc_vae* h, const void* x, void* out, int F, int H, int W, void* stream);
/* z: DEVICE bf16 [z_dim][T][h][w] (normalised latents); out: DEVICE bf16 [3][1 + 4 (T - 1)][8h][8w], clamped to [-1, 1]. */
int vc_vae_decode(vc_vae* h, const void* z, void* out, int T, int h_lat, int w_lat, void* stream);
const char* vc_vae_last_error(const vc_vae* h);
int64_t vc_vae_workspace_bytes(const vc_vae* h);
int vc_vae_release_workspace(vc_vae* h);
/* The full-resolution stage of encode / decode (first conv + first residual blocks + strided conv; last upsample + last residual
 * blocks + head) can run in TIME CHUNKS with two frames of history per causal convolution -- upstream's own execution order
 * (per-conv feature caches), bit-identical to the whole-sequence walk, with a fraction of its workspace (an 81-frame 720p decode:
 * 137 GiB whole, see DESIGN.md 8 for the chunked figure).  frames: -1 (default) = automatic -- chunks of VC_VAE_CHUNK_FRAMES (8) when the
 * whole-sequence arena would exceed VC_VAE_WS_LIMIT_GB (40); 0 = never; n > 0 = always n frames.  vc_vae_last_time_chunk: what the last
 * call used (0 = whole sequence). */
int vc_vae_set_time_chunk(vc_vae* h, int frames);
int vc_vae_last_time_chunk(const vc_vae* h);
void vc_vae_destroy(vc_vae* h);

/* ---- 4D control-map renderer (second half): the per-pixel stages of inference/rendering_4D_control_maps.py.
 * Device pointers; images are [H][W] (depth float32, masks uint8 0/1) and [H][W][3] uint8; a call may cover a whole batch of frames
 * (npix = B H W).  Python mirror with the reference's function names: versecrafter_amd/rendering/control_maps.py.
 *   vc_op_render_composite     take_fg = fg_mask & (bg_depth <= 0 | (fg_depth > 0 & fg_depth < bg_depth - 1e-6)); out_rgb / out_depth take
 *                              the foreground there (composite_by_depth_batch :398-411); out_mask (optional, needs bg_mask) = 255 x
 *                              (take_fg ? 1 : !bg_mask) on three channels (merge_bg_and_fg_mask :736-763).  Any output may be NULL.
 *   vc_op_render_depth_gray    disparity 1/d (0 where d <= 0), (disp - min_disp) / denom when normalize, clamp, x 255 truncated (:520-537);
 *                              min_disp and denom are computed by the host exactly as the reference computes them.
 *   vc_op_render_gauss_density sum over n projected Gaussians of coeff exp(-mahalanobis / 2) at pixel (u, v) = (column, row) (:801-883).
 *                              records: n x 12 floats {mean_u, mean_v, inv00, inv01, inv10, inv11, coeff, valid, r, g, b, 0}.
 *   vc_op_render_gauss_frame   the same records in compositing order (far to near): each density map is divided by its maximum + 1e-8,
 *                              alpha = (d - threshold) / span above the threshold, "over" compositing of colour and alpha (:660-693);
 *                              scratch_max: n x 4 bytes.  out_rgb uint8 [H][W][3], out_alpha float32 [H][W].
 *   vc_op_render_blend         C = fg/255 alpha + bg/255 (1 - alpha) -> uint8 (:719-732); masked != 0: fg/255 alpha 255 (:1322-1326).
 * PARITY UNPINNED (PyTorch3D is not in the reference tree; restated from its published algorithms, specification = oracle/render_oracle.py):
 *   vc_op_render_points        PointsRasterizer(radius, points_per_pixel K) + AlphaCompositor(background) of :243-338 for ONE camera:
 *                              points [n][3] float32 (world), colors [n][3] uint8, w2c row-major [4][4] (OpenCV world-to-camera,
 *                              :1001-1009), K3 row-major [3][3] pixel intrinsics (HOST pointers) -> rgb uint8, depth float32, mask uint8.
 *   vc_op_render_mesh          MeshRasterizer(blur 0, 1 face per pixel) + HardPhongShader(point light) of :150-241 for ONE camera:
 *                              verts [nv][3] float32 (world), vert_colors [nv][3] float32 in [0, 1], faces [nf][3] int32.               */
const char* vc_render_last_error(void);
int vc_op_render_composite(const void* bg_rgb, const void* bg_depth, const void* fg_rgb, const void* fg_depth, const void* fg_mask,
                           const void* bg_mask, void* out_rgb, void* out_depth, void* out_mask, int64_t npix, void* stream);
int vc_op_render_depth_gray(const void* depth, void* out_rgb, int64_t npix, int normalize, float min_disp, float denom, void* stream);
int vc_op_render_gauss_density(const void* records, int n, void* out, int W, int H, void* stream);
int vc_op_render_gauss_frame(const void* records, int n, void* scratch_max, float threshold, float span, void* out_rgb, void* out_alpha, int W,
                             int H, void* stream);
int vc_op_render_blend(const void* fg_rgb, const void* alpha, const void* bg_rgb, void* out_rgb, int64_t npix, int masked, void* stream);
int64_t vc_op_render_points_scratch_bytes(int64_t npoints, int W, int H, int K);
int vc_op_render_points(const void* points, const void* colors, int64_t npoints, const float* w2c, const float* K3, int W, int H, float radius,
                        int K, float background, void* scratch, void* out_rgb, void* out_depth, void* out_mask, void* stream);
int64_t vc_op_render_mesh_scratch_bytes(int nverts, int W, int H);
int vc_op_render_mesh(const void* verts, const void* vert_colors, int nverts, const void* faces, int nfaces, const float* w2c, const float* K3,
                      const float* light_xyz, const float* eye_xyz, int W, int H, int background_u8, void* scratch, void* out_rgb,
                      void* out_depth, void* out_mask, void* stream);

/* ---- per-object 3D Gaussian fit: pre-processing step 3, upstream of the renderer (reference inference/fit_3D_gaussian.py) ----------
 * Device pointers unless marked HOST; pixel maps are row-major [H][W].  PINNED by the reference's own outputs for its demo clips
 * (tests/golden/demo_fit/).  Errors: vc_fit_last_error() (thread-local text).
 *   vc_op_fit_erode_mask   load_mask :139-159: out = (mask > 127) eroded once by cv2's ksize x ksize MORPH_ELLIPSE element (anchor at the
 *                          centre; pixels outside the image remove nothing).  mask uint8 grey, out uint8 0 / 1.  ksize <= 31.
 *   vc_op_fit_points       get_point_cloud_from_depth :35-92: world = c2w [K^-1 (x, y, 1) depth; 1] for the pixels whose mask is set
 *                          (mask NULL: depth > 0), written in row-major pixel order.  kinv HOST [3][3], c2w HOST [3][4] (row-major, the
 *                          top rows of the inverse extrinsic).  out_points float32 [H*W][3] (capacity), out_count int64 (device).
 *   vc_op_fit_moments      fit_3d_gaussian :95-136: out12 = float32 {mean[3], cov[3][3]}, cov = centred^T centred / (n - 1) + 1e-6 I;
 *                          fp64 accumulation in a fixed order (bit-reproducible).  n >= 3.
 *   vc_op_fit_project      project_gaussian_to_2d :259-285: rec11 HOST = {mean_u, mean_v, inv00, inv01, inv10, inv11, coeff, min_x, max_x,
 *                          min_y, max_y}; density = coeff exp(-m / 2), mahal = m inside [min_x, max_x) x [min_y, max_y), 0 / +inf outside
 *                          (an empty box = a culled Gaussian).  dmax (optional, device float32): the largest density.
 *   vc_op_fit_blend        visualize_gaussian_projections :381-397: mask = max(mask, mahal <= threshold), alpha = clamp(density / *dmax),
 *                          picture = rgb3 alpha + picture (1 - alpha).  picture float32 [H][W][3], mask float32 [H][W], rgb3 HOST.
 *   vc_op_fit_picture_u8   :400: uint8(clamp(x, 0, 1) 255).                                                                          */
const char* vc_fit_last_error(void);
int vc_op_fit_erode_mask(const void* mask_u8, void* out_u8, int W, int H, int ksize, void* stream);
int64_t vc_op_fit_points_scratch_bytes(int W, int H);
int vc_op_fit_points(const void* depth, const void* mask, const float* kinv, const float* c2w, int W, int H, void* scratch, void* out_points,
                     void* out_count, void* stream);
int64_t vc_op_fit_moments_scratch_bytes(void);
int vc_op_fit_moments(const void* points, int64_t n, void* scratch, void* out12, void* stream);
int vc_op_fit_project(const float* rec11, void* density, void* mahal, void* dmax, int W, int H, void* stream);
int vc_op_fit_blend(const void* density, const void* mahal, const void* dmax, float threshold, const float* rgb3, void* picture, void* mask,
                    int64_t npix, void* stream);
int vc_op_fit_picture_u8(const void* src, void* dst, int64_t n, void* stream);

/* ---- sample planes of the CLIs' .mp4 files (stands in for the ffmpeg call behind save_videos_grid, CLI.py:456, and
 * save_video_from_frames, rendering_4D_control_maps.py:455-485).  The stream is H.264 with every macroblock I_PCM (ITU-T H.264 7.3.5);
 * these entries convert uint8 RGB [F][H][W][3] <-> the macroblock layer [F][ceil(H/16) * ceil(W/16)][386] = {0x0D, 0x00, Y 16x16,
 * Cb 8x8, Cr 8x8} in BT.601 limited range (integer forms in csrc/h264pcm.hip), edge pixels replicated into the cropped border.
 * Headers and the MP4 boxes: versecrafter_amd/utils/mp4_pcm.py.  Device pointers.                                                  */
const char* vc_h264_pcm_last_error(void);
int64_t vc_op_h264_pcm_bytes(int frames, int H, int W);
int vc_op_h264_pcm_pack(const void* rgb, void* out, int frames, int H, int W, void* stream);
int vc_op_h264_pcm_unpack(const void* in, void* rgb, int frames, int H, int W, void* stream);

/* ---- fp8 GEMM operands (BASELINE config 5 names "fp8 MFMA"; a capability of this build, off by default: the reference computes in bf16
 * and its fp8 mode only STORES weights in fp8, CLI.py:292-301).  OCP e4m3, one fp32 scale per row: x ~ q * scale, scale = amax / 448.
 *   vc_op_quantize_rows_fp8   bf16 [M, K] (ldx elements) -> uint8 e4m3 [M, K] (ldq bytes) + float32 scale [M]
 *   vc_op_gemm_fp8            C bf16 [M, N] = epilogue((A W^T) a_scale[m] w_scale[n]); A [M, K], W [N, K] e4m3; epilogue kinds BIAS,
 *                             BIAS_GELU, BIAS_RESID, BIAS_GATE_RESID as vc_op_gemm_bf16.  N % 256 == 0, K % 256 == 0, M % 256 == 0 unless
 *                             a_rows_padded (rows of A up to the next multiple of 256 readable), 16-byte aligned operands.        */
/*   vc_set_fp8_linear         engine mode: on != 0 gives every nn.Linear of the main and adapter blocks (self-attention q/k/v/o, cross-attention
 *                             q/o, ffn.0/ffn.2, before_proj/after_proj) an e4m3 copy with per-output-channel scales (all weights must be
 *                             loaded); their GEMMs then quantise the activations per token and run in fp8.  Attention, norms, embeddings,
 *                             text K/V and the head stay bf16.  on == 0 frees the copies.  vc_fp8_linear: 1 when the mode is on.        */
int vc_set_fp8_linear(vc_engine* h, int on);
int vc_fp8_linear(const vc_engine* h);
/*   vc_set_fp8_attention      engine mode: on != 0 runs the SELF-attention of the main and adapter blocks through vc_op_attention_fp8's kernels
 *                             (q, k, v and the softmax weights in e4m3 under MX-style block scales; pmode as there); takes effect at the next
 *                             vc_prepare_video, which then reserves the quantised operands' workspace in the arena.  Cross-attention and the
 *                             Ulysses x ring hybrid's per-block attention stay bf16.  vc_fp8_attention: 1 when the mode is on.        */
int vc_set_fp8_attention(vc_engine* h, int on, int pmode);
int vc_fp8_attention(const vc_engine* h);
int vc_op_quantize_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, void* scale, int M, int K, void* stream);
int vc_op_gemm_fp8(const void* A, int64_t lda, const void* a_scale, const void* W, int64_t ldw, const void* w_scale, void* C, int64_t ldc,
                   const void* bias, int M, int N, int K, int epilogue, const void* resid, int64_t ldr, const void* gate,
                   int64_t gate_bstride, int rows_per_batch, int a_rows_padded, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VCENGINE_H */
