/*
 * liborv_mi355.so - C ABI of the MI355X (gfx950) kernels behind the ORV denoising hot path.
 *
 * The reference (OrangeSodahub/ORV) has no FFI seam on this path: it is nn.Module Python calling torch
 * (SURVEY.md §8b).  This header is the seam the replacement introduces *beneath* the reference's Python
 * surface; every entry point cites the reference code whose arithmetic it replaces
 * (paths relative to /root/reference).  The Python binding a maintainer adds is in INTEGRATION.md.
 *
 * Conventions
 *  - All pointers are DEVICE pointers borrowed from the caller (torch tensors); the library never
 *    allocates or frees caller memory.  bf16 tensors are passed as `const void*` to 2-byte elements.
 *  - Every call is stream-ordered and asynchronous on `stream` (a hipStream_t passed as void*;
 *    callers pass torch.cuda.current_stream().cuda_stream).  No internal synchronisation.
 *  - Return value: 0 = ok, <0 = error (ORV_E*); never throws.  `orv_last_error()` returns a
 *    thread-local message for the last failing call on this thread.
 *  - "Token group" indexing (per-frame modulation, cogvideox_control.py:99-105,133-140):
 *      row s of a [B, S, D] joint sequence (text first) belongs to group
 *      g(s) = 0                      if s <  n_text
 *             1 + (s - n_text) / P   otherwise           (P = tokens per latent frame)
 *    and modulation tables are fp32 [B, G, D] slices with explicit strides.
 */
#ifndef ORV_MI355_H
#define ORV_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORV_OK 0
#define ORV_EINVAL (-1)   /* bad argument / unsupported shape */
#define ORV_EDEVICE (-2)  /* not a gfx950 device / HIP failure */
#define ORV_ELAUNCH (-3)  /* kernel launch failed */

/* -- library ---------------------------------------------------------------------------------- */
int orv_version(void);                 /* ABI version, currently 1 */
const char* orv_last_error(void);      /* thread-local message of the last failing call */
int orv_device_check(int device);      /* 0 iff `device` is gfx950 (MI355X) */

/* Token-group descriptor shared by the modulation-aware kernels. */
typedef struct {
    int seq;        /* S: rows per batch element */
    int n_text;     /* rows [0, n_text) are text rows (group 0) */
    int per_group;  /* P: video rows per group; 0 => all video rows are group 1 */
} orv_groups_t;

/* Row remap r -> (r / rows) * bstride + off + r % rows (rows == 0: identity).  Lets a kernel read or write the video
 * (or text) rows of the joint [B, S, D] sequence in place instead of materialising torch.cat / slices
 * (cogvideox_control.py:222,267-269,437-443,792-794). */
typedef struct {
    int rows, bstride, off;
} orv_rowmap_t;

/* -- embeddings ------------------------------------------------------------------------------- */
/* diffusers Timesteps(dim, flip_sin_to_cos, shift) as called at orv/models/cogvideox_control.py:763,772:
 * out[b] = [cos(t*f) | sin(t*f)] (flip) with f_k = exp(-ln(1e4) k / (dim/2 - shift)); fp32 math, bf16 out. */
int orv_timestep_embedding(const float* t, void* out_bf16, int batch, int dim, int flip_sin_to_cos,
                           float freq_shift, void* stream);

/* Small-M linear (M <= 64): out[m, :] = act_out( W . act_in(x[m] + xb[m / xb_rep]) + bias ).
 * Replaces the tiny MLPs/linears of time_embedding (:769), ActionEmbed (components.py:37-45,65),
 * ActionRecon (components.py:86-90) and the AdaLN modulation linears (:121-130, :172).
 * x [M,K] bf16; xb optional [ceil(M/xb_rep), K] bf16 broadcast-added before act_in; W [N,K] bf16; bias [N] bf16|NULL.
 * act: 0 none, 1 SiLU, 2 GELU(tanh).  out_f32 != 0 -> fp32 output else bf16.  ldo = output row stride (elements); omap remaps output rows. */
int orv_skinny_linear(const void* x, const void* xb, int xb_rep, const void* W, const void* bias, void* out,
                      int M, int N, int K, int act_in, int act_out, int out_f32, int ldo, orv_rowmap_t omap,
                      void* stream);

/* Gather [B,T,C,H,W] latents into patch tokens (diffusers CogVideoXPatchEmbed.forward patchify, called at
 * cogvideox_control.py:788,833,842).  Two channel-concatenated sources (cogvideox_control.py:1409-1413 cat on dim 2)
 * are read in place: channels [0,c0) from src0, [c0,c0+c1) from src1 (src1 may be NULL with c1 = 0).
 * pt == 0: tokens [B, T*h*w, (C, p, p)]      (Conv2d weight.flatten(1) order)
 * pt  > 0: tokens [B, T/pt*h*w, (C, pt, p, p)]  (CogVideoX1.5 Linear order).  Index-exact (bit copies). */
int orv_patchify(const void* src0, int c0, const void* src1, int c1, void* tokens, int B, int T, int H, int W,
                 int p, int pt, void* stream);

/* Inverse scatter of cogvideox_control.py:926-936: x [B, N, p*p*(pt)*C] -> out [B,T,C,H,W].  Index-exact. */
int orv_unpatchify(const void* x, void* out, int B, int T, int C, int H, int W, int p, int pt, void* stream);

/* out[m, col_off + d] = a[amap(m), d] + b[m, d] (bf16, fp32 add, one rounding): builds the input of
 * initial_combine_linear, `hidden_states.repeat(1, 1, num_control_keys) + cat(controls, -1)` (cogvideox_control.py:849-855),
 * one control key per call, reading the video rows of the joint sequence in place. */
int orv_add_rows(const void* a, int lda, orv_rowmap_t amap, const void* b, int ldb, void* out, int ldo, int col_off, int M,
                 int D, void* stream);

/* Multiview cross-view attention (MVBlock.forward cogvideox_control.py:313-348) re-groups tokens
 * '(b v) (f s) d -> (b f) (v s) d' (einops rearrange :328,:346; text '(b v) n d -> (b f) (v n) d' :329-331):
 * orv_gather_rows:        dst[r, :] = src[idx[r], :]                      (idx = host-built permutation, int32 on device)
 * orv_scatter_gated_rows: x[idx[r], :] += gate[idx[r] / seq, :] * y[r, :]  for video target rows only - the rearrange back
 *                         fused with the gated residual `hidden + gate_msa * attn` (:347). */
int orv_gather_rows(const void* src, int ld_src, const int* idx, void* dst, int ld_dst, int R, int D, void* stream);
int orv_scatter_gated_rows(const void* y, int ldy, const int* idx, const float* gate, long gate_b, void* x, int ldx, int R,
                           int D, int seq, int n_text, void* stream);

/* -- normalisation ---------------------------------------------------------------------------- */
/* y = LN(x; gamma, beta, eps) * (1 + scale[b, g(s)]) + shift[b, g(s)]   (CogVideoXLayerNormZero.forward
 * cogvideox_control.py:117-145, AdaLayerNorm.forward :155-197, nn.LayerNorm norm_final :909-916).
 * x,y [B*S, D] bf16 (row strides ldx/ldy elements); gamma/beta bf16 [D] or NULL; scale/shift fp32 with
 * element strides (mod_b, mod_g) or NULL (plain LayerNorm).  Statistics in fp32.  Row r of the [batch*grp.seq] problem is
 * read from x row xmap(r) and written to y row r. */
int orv_layernorm_modulate(const void* x, int ldx, orv_rowmap_t xmap, void* y, int ldy, const void* gamma,
                           const void* beta, const float* scale, const float* shift, long mod_b, long mod_g,
                           orv_groups_t grp, int batch, int D, float eps, void* stream);
/* The same with y written in the packed P16 layout (orv_gemm_t: orv_packed_rows(batch * seq) x D bf16) - the A operand of the projections that
 * follow a CogVideoXLayerNormZero (cogvideox_control.py:232-234, 439) on the d8 GEMM.  Full case only (gamma, beta, scale, shift given, no row
 * map), D % 32 == 0, D <= 2048. */
int orv_layernorm_modulate_packed(const void* x, int ldx, void* y, const void* gamma, const void* beta, const float* scale,
                                  const float* shift, long mod_b, long mod_g, orv_groups_t grp, int batch, int D, float eps,
                                  void* stream);

/* All AdaLN modulation linears of one forward in a single launch (CogVideoXLayerNormZero "partially forward self.linear
 * twice" cogvideox_control.py:117-130, and AdaLayerNorm :172): for tab in [0, n_tab)
 *   out[tab][b][1+t][:] = W[tab][0:width]       . SiLU(temb[b] + action_emb[b,t]) + bias[tab][0:width]
 *   out[tab][b][0][:]   = W[tab][width:2*width] . SiLU(temb[b])                   + bias[tab][width:2*width]   (if text)
 * temb [B,E] bf16, action_emb [B,T,E] bf16 or NULL (then T must be 1 and the video rows use SiLU(temb)); W/bias are DEVICE
 * arrays of n_tab device pointers to bf16 [width*(1+text), E] / [width*(1+text)] (bias array or entries may be NULL);
 * out fp32 [n_tab, B, 1+T, width].  E in {64,128,256,512}, width % 32 == 0.  Streams every weight once (HBM-bound). */
int orv_modulation_tables(const void* temb, const void* action_emb, const void* const* W, const void* const* bias,
                          float* out, int n_tab, int B, int T, int E, int width, int text, void* stream);

/* In-place per-head LayerNorm(64, eps) on the q and k thirds of a packed qkv buffer [B*S, 3*H*64] bf16, optional
 * RoPE on rows >= n_text (pairs (2i,2i+1); cos/sin fp32 [S-n_text, 64]), and transpose of the v third into
 * vT [B, H, 64, s_pad] (zero-filled for s >= S; the key axis is stored with bits 2 and 3 of the key index exchanged,
 * the order orv_attention_fwd's PV MFMA consumes).  q is multiplied by `q_premul` before its single bf16 rounding: pass
 * softmax_scale * log2(e) and call orv_attention_fwd with scale = softmax_scale / q_premul to take its fused exp2 path
 * (1.0f = leave q as is).  Replaces cogvideox_control.py:239-254 (+ diffusers Attention.norm_q/norm_k,
 * apply_rotary_emb). */
int orv_qkv_prep(void* qkv, void* vT, const void* gq, const void* bq, const void* gk, const void* bk,
                 const float* rope_cos, const float* rope_sin, int B, int S, int H, int n_text, int s_pad,
                 float eps, float q_premul, void* stream);
/* Out-of-place form (training keeps the raw QKV GEMM output for the qk-LayerNorm adjoint): reads `src`, writes the
 * normalised q / k and the untouched v third to `qkv`.  src == qkv is the in-place call above. */
int orv_qkv_prep_from(const void* src, void* qkv, void* vT, const void* gq, const void* bq, const void* gk, const void* bk,
                      const float* rope_cos, const float* rope_sin, int B, int S, int H, int n_text, int s_pad, float eps,
                      float q_premul, void* stream);

/* -- GEMM ------------------------------------------------------------------------------------- */
/* C = epilogue(A[M,K] . W[N,K]^T + bias[N]); bf16 operands, fp32 MFMA accumulation, bf16 output.
 * Replaces nn.Linear / Conv2d(k=s=p) on the path: to_q/k/v + to_out (cogvideox_control.py:232-234,263),
 * FeedForward (:439-440), text_proj/proj (:788), proj_out (:920), initial_combine_linear (:853).
 * epilogue: 0 bias; 1 bias+GELU(tanh) (FeedForward net.0); 2 C = R[r_row] + gate[b,g(row)] * (acc + bias)
 * (gated residual :419-421,442-443; gate NULL => 1; R row r_row = r_mod ? m % r_mod : out_row, so a
 * [n,D] table broadcast over the batch - the sincos pos-embedding - uses r_mod = n).
 * epilogue 3 (backward): C = acc * GELU'(R[out_row]) - the dgrad of FeedForward net.2 fused with the GELU adjoint
 * (R = saved pre-activation).  Epilogue 2 with gate NULL and R == C accumulates gradients in place.
 * epilogue 4: the packed QKV projection with the qk LayerNorm fused (see the qn_* fields).
 * Output row remap: out_row = cmap(m) (scatter into the joint [B,S,D] sequence).  Constraints: K % 64 == 0, N % 64 == 0, 16-byte aligned rows. */
typedef struct {
    const void* A; int lda;
    const void* W; int ldw;
    const void* bias;
    void* C; int ldc;
    int M, N, K;
    int epilogue;
    const void* R; int ldr; int r_mod;
    const float* gate; long gate_b, gate_g; orv_groups_t grp;
    orv_rowmap_t cmap;
    void* Y; int ldy;   /* optional: also store acc + bias (before GELU / gate) at the same rows - saved for backward */
    /* epilogue 4 only - the packed QKV projection [M, 3*heads*64] with the per-head LayerNorm(64, eps) of the q and k
     * thirds (affine [64] bf16 or NULL) fused, q additionally multiplied by qn_premul, v stored as is; Y (if given)
     * receives the raw projection.  Equivalent to epilogue 0 followed by orv_qkv_prep without RoPE, minus the V^T copy
     * (use orv_head_transpose for that). */
    const void *qn_gamma_q, *qn_beta_q, *qn_gamma_k, *qn_beta_k; float qn_eps, qn_premul; int qn_heads;
    /* Packed operand layout "P16" (round 5, gemm_d8.hip).  A [rows, cols] bf16 matrix (cols % 32 == 0) stored as 1-KiB blocks of 16 rows x 32
     * columns, block (R, c) at byte ((R * cols / 32) + c) * 1024, element (r, k) of a block at ((k / 8) * 16 + r) * 16 + (k % 8) * 2 - the
     * order in which one wave instruction of v_mfma_f32_16x16x32_bf16 wants a 16 x 32 operand, so a GEMM reads its A operand straight into
     * registers with contiguous 1-KiB loads.  The buffer holds orv_packed_rows(rows) = rows rounded up to 256 row slots (padding rows are
     * never interpreted).  a_packed: A is in that layout (lda == K; needs K % 192 == 0); c_packed: C is written in it (ldc == N; epilogues
     * 0 / 1, no cmap / Y) - the FeedForward hidden state between cogvideox_control.py:439 and :440 never exists row-major.  Producers:
     * orv_pack_rows16, orv_layernorm_modulate (out_packed), orv_gemm_bf16 (c_packed), orv_attention_fwd* (out_packed). */
    int a_packed, c_packed;
} orv_gemm_t;
int orv_gemm_bf16(const orv_gemm_t* g, void* stream);
/* Row slots of a packed P16 buffer for `rows` rows (rounded up to the 256-row GEMM tile). */
long orv_packed_rows(long rows);
/* dst (P16 layout, orv_packed_rows(M) x K) = src [M, K] row-major (ld_src elements per row), bit copies; padding rows are zero-filled.
 * The reference has no such step (its operands stay row-major inside torch.nn.functional.linear, cogvideox_control.py:232-234,263,439-440):
 * this is the entry of tensors that no packed-writing kernel produced (tests, the first GEMM after a foreign op). */
int orv_pack_rows16(const void* src, long ld_src, void* dst, int M, int K, void* stream);
/* Inverse of orv_pack_rows16 (rows [0, M)). */
int orv_unpack_rows16(const void* src, void* dst, long ld_dst, int M, int K, void* stream);
/* Kernel symbol (as rocprofv3 prints it, e.g. "gemm_pp_kernel<192, 5, 1>") that orv_gemm_bf16 launches for this shape on this
 * device: the tile is chosen by a cost model over all candidates, so callers that label timings ask. */
int orv_gemm_kernel_name(int M, int N, int K, int epilogue, char* buf, int len);
/* The same for a call with packed operands (orv_gemm_t.a_packed / c_packed); an error when no kernel takes the combination (the caller then
 * keeps the row-major path). */
int orv_gemm_kernel_name_packed(int M, int N, int K, int epilogue, int a_packed, int c_packed, char* buf, int len);
/* Developer switch (sweeps, same-process A/B, per-instantiation tests): pin the tile candidate (kernel family `ring`: 0 simple,
 * 1 ring, 2 phased, 3 t8; tile bm x bn) for all later orv_gemm_bf16 calls of this process; bm = 0 returns to the cost model.
 * Same effect as the environment variable ORV_GEMM_TILE="ring,bm,bn" read at the first call. */
int orv_gemm_force_tile(int ring, int bm, int bn);
/* Number of orv_gemm_force_tile calls so far: a host-side plan derived from orv_gemm_kernel_name(_packed) (which GEMMs of a block take packed
 * operands) is valid for the epoch it was made in and is re-derived when this moves. */
int orv_gemm_force_epoch(void);
/* C[M, N] (+)= A[K, M]^T . W[K, N] - both operands row-major over the CONTRACTION index (K rows): the weight gradient dW = dY^T X of a linear
 * layer straight from the row-major dY [tokens, out] and X [tokens, in], without transposed copies (torch autograd of nn.Linear inside
 * `accelerator.backward(loss)`, train_cogvideox_control_to_video_sft.py:1093; the linears of cogvideox_control.py:232-234, 263, 439-440).
 * bf16 operands, fp32 accumulation, bf16 C; accumulate != 0 adds to C (gradient accumulation).  M % 8 == 0, N % 192 == 0 or N % 256 == 0. */
int orv_gemm_tn_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, int accumulate, void* stream);

/* -- backward (training) ------------------------------------------------------------------------------ */
/* dst[c, r] = src[r, c] ([R, C] bf16 -> [C, ld_dst], columns [R, ld_dst) zero-filled).  Feeds the NT GEMM with the
 * K-contiguous operands of dgrad (W^T) and wgrad (dY^T, X^T): dX = dY . W, dW = dY^T . X (torch autograd of nn.Linear). */
int orv_transpose_bf16(const void* src, int ld_src, void* dst, int ld_dst, int R, int C, void* stream);
/* orv_transpose_bf16 with colsum[c] += sum_r src[r, c] taken in the same pass (fp32 atomics): dY^T for the weight gradient and the
 * bias gradient of one nn.Linear from ONE read of dY (orv_transpose_bf16 followed by orv_colsum reads it twice). */
int orv_transpose_colsum_bf16(const void* src, int ld_src, void* dst, int ld_dst, int R, int C, float* colsum, void* stream);
/* out[c] += sum_r src[r, c] (fp32 atomics): bias gradients. */
int orv_colsum(const void* src, int ld, float* out, int R, int C, void* stream);
/* Adjoint of out = x + gate[b,g(row)] * y (cogvideox_control.py:419-421,442-443): dy = gate * dout (bf16),
 * dgate[b,g,:] += sum_rows dout * y (table laid out like `gate`; per-workgroup partials in `scratch`, then reduced). */
long orv_gated_residual_bwd_scratch(orv_groups_t grp, int batch, int D);   /* floats of per-workgroup partial sums */
int orv_gated_residual_bwd(const void* dout, const void* y, const float* gate, float* dgate, void* dy, float* scratch,
                           long mod_b, long mod_g, orv_groups_t grp, int batch, int D, void* stream);
/* Adjoint of orv_layernorm_modulate: dx[xmap(r)] = LN-path gradient (+ dres[xmap(r)] if given), and fp32 sums
 * dscale/dshift (tables like scale/shift, +=), dgamma/dbeta [D] (+=): per-workgroup partial column sums go to `scratch`
 * (orv_layernorm_modulate_bwd_scratch() floats) and are reduced by two small kernels.  Any output may be NULL. */
long orv_layernorm_modulate_bwd_scratch(orv_groups_t grp, int batch, int D);
int orv_layernorm_modulate_bwd(const void* dy, const void* x, orv_rowmap_t xmap, const void* dres, void* dx,
                               const void* gamma, const void* beta, const float* scale, float* dscale, float* dshift,
                               float* dgamma, float* dbeta, float* scratch, long mod_b, long mod_g, orv_groups_t grp,
                               int batch, int D, float eps, void* stream);
/* Adjoint of orv_modulation_tables for all n_tab AdaLN linears at once (cogvideox_control.py:117-130,:172 under autograd):
 * dtab fp32 [n_tab, B, 1+T, width] (d out); cond_v bf16 [B*T, E] = SiLU(temb + action_emb) rows, cond_t bf16 [B, E] =
 * SiLU(temb); W device array of n_tab weight pointers as in the forward.  Writes gW bf16 [n_tab, width*(1+text), E]
 * and gb fp32 [n_tab, width*(1+text)] (overwritten), accumulates d_cond_v fp32 [B*T, E] / d_cond_t fp32 [B, E].
 * Rows are processed 32 at a time (one pass over the weights up to B*T = 32). */
int orv_modulation_tables_bwd(const float* dtab, const void* cond_v, const void* cond_t, const void* const* W, void* gW,
                              float* gb, float* d_cond_v, float* d_cond_t, int n_tab, int B, int T, int E, int width,
                              int text, void* stream);
/* Small-row (R <= 4096) linear adjoint for the conditioning MLPs / AdaLN linears: dW[N,K] (+)= dy^T x (bf16),
 * db[N] (+)= colsum(dy) (fp32), dx[R,K] += dy W (fp32 atomics).  dy fp32 [R, ldy].  Any of dW/dx may be NULL. */
int orv_small_linear_bwd(const float* dy, int ldy, const void* x, int ldx, const void* W, void* dW, float* db, float* dx,
                         int lddx, int R, int N, int K, int accumulate, void* stream);
/* Fused AdamW step on bf16 params/grads with fp32 moments (torch.optim.AdamW semantics, base_train.yaml:143-153);
 * `clip_coef` (device scalar or NULL) multiplies the gradient (global-norm clipping, train...sft.py:1095-1100). */
int orv_adamw(void* p, const void* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int step, const float* clip_coef, void* stream);
/* The same AdamW update over ONE flat buffer (every parameter a segment padded to 2048 elements; seg_start[nseg] int64
 * element offsets and seg_active[nseg] bytes on the device): one launch per optimizer step instead of one per parameter.
 * Segments with seg_active == 0 (no gradient this step) are left untouched, as torch.optim.AdamW does. */
int orv_adamw_flat(void* p, const void* g, float* m, float* v, long n, const long* seg_start,
                   const unsigned char* seg_active, int nseg, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, const float* clip_coef, void* stream);
/* Same, with ONE STEP COUNT PER SEGMENT (seg_step[nseg] int32 on the device, the count INCLUDING this update; NULL = the
 * global `step`): torch.optim.AdamW keeps `state[p]["step"]` per parameter, so a parameter that was skipped in earlier
 * steps (no gradient) gets the bias correction of its own count. */
int orv_adamw_flat_steps(void* p, const void* g, float* m, float* v, long n, const long* seg_start,
                         const unsigned char* seg_active, const int* seg_step, int nseg, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int step, const float* clip_coef, void* stream);
/* dst_ptr[s][j] = bf16(src[src_off[s] + j]), j < len[s], for nseg segments (src_off / dst_ptr / len are DEVICE arrays; dst_ptr holds
 * device addresses of bf16 storage; max_len = max len[s]): the small fp32-accumulated parameter gradients go from the backward's
 * accumulator arena into the fused optimizer's flat gradient buffer in one launch (the reference leaves this to autograd's
 * per-parameter accumulation, train_cogvideox_control_to_video_sft.py:1093). */
int orv_scatter_f32_to_bf16(const float* src, const long* src_off, const long* dst_ptr, const int* len, int nseg, int max_len,
                            void* stream);
/* out[0] += sum g^2 (gradient-norm reduction, orv/utils.py:166-174). */
int orv_sumsq(const void* g, long n, float* out, void* stream);

/* src[B*S, ld] columns [col0 + h*64, +64) -> dst[b,h,d,pos(s)] per head (pos = s with bits 2<->3 exchanged, zero for
 * s >= S): the forward's V^T when the qk LayerNorm rides in the QKV GEMM (epilogue 4), and the sequence-contiguous copies
 * (q'^T, k^T, dO^T) the attention backward contracts over. */
int orv_head_transpose(const void* src, int ld, int col0, void* dst, int B, int S, int H, int s_pad, void* stream);
/* Adjoint of orv_attention_fwd on the fused path (q pre-multiplied by scale*log2 e in orv_qkv_prep): given out, dout and
 * the saved lse, writes dqkv[B*S, ld_dqkv] = (dq | dk | dv) w.r.t. the normalised, un-premultiplied q, k and v.
 * qT/kT/doT: orv_head_transpose of q', k, dout.  neg_lse2/neg_delta: fp32 scratch [B,H,s_pad]. */
int orv_attention_bwd(const void* qkv, int ld_qkv, const void* qT, const void* kT, const void* out, const void* dout,
                      int ld_out, const void* doT, const float* lse, float* neg_lse2, float* neg_delta, void* dqkv,
                      int ld_dqkv, int B, int S, int H, int s_pad, float scale, void* stream);
/* Adjoint of orv_qkv_prep for the q and k thirds, in place on dqkv: inverse RoPE, LayerNorm(64) backward (qkv_raw = the
 * QKV GEMM output before orv_qkv_prep); dgq/dbq/dgk/dbk fp32 [64] accumulate the norm_q / norm_k parameter gradients. */
int orv_qkv_prep_bwd(const void* qkv_raw, void* dqkv, const void* gq, const void* gk, const float* rope_cos,
                     const float* rope_sin, float* dgq, float* dbq, float* dgk, float* dbk, float* scratch, int B, int S,
                     int H, int n_text, float eps, void* stream);   /* scratch: fp32 [ceil(S/64)*H*B*256] */

/* -- attention -------------------------------------------------------------------------------- */
/* Non-causal, unmasked softmax(q k^T * scale) v over the joint text+video sequence, head_dim 64
 * (F.scaled_dot_product_attention at cogvideox_control.py:256-258).  q,k are read in place from the packed qkv
 * buffer [B*S, ld_qkv] (q at column h*64, k at column H*64 + h*64); vT [B,H,64,s_pad] from orv_qkv_prep / orv_head_transpose;
 * out [B*S, H*64] bf16 (heads merged as :260); lse fp32 [B,H,S] or NULL (log-sum-exp, natural log, for backward). */
int orv_attention_fwd(const void* qkv, int ld_qkv, const void* vT, void* out, int ld_out, float* lse, int B, int S,
                      int H, int s_pad, float scale, void* stream);
/* The same attention (V read in place) when the caller can BOUND the scores: |q . k| * scale * log2(e) <= score_bound for every
 * (query, key) of the call.  ORV always applies the per-head qk LayerNorm (cogvideox_control.py:243-247), so the bound follows
 * from norm_q / norm_k's affine parameters alone.  With the fused scale (q pre-multiplied) and score_bound <= 90 (16-byte aligned
 * output; <= 60 otherwise) the softmax runs WITHOUT a running max or rescale (P = exp2(s) stays a normal fp32 / bf16 number);
 * otherwise identical to orv_attention_fwd.  lse as there. */
int orv_attention_fwd_bounded(const void* qkv, int ld_qkv, void* out, int ld_out, float* lse, int B, int S, int H, float scale,
                              float score_bound, void* stream);
/* orv_attention_fwd_bounded writing `out` in the packed P16 layout (orv_gemm_t: orv_packed_rows(B S) x (H 64) bf16, 16-byte aligned) - the A
 * operand of the attention out-projection (cogvideox_control.py:263) without a row-major copy.  Needs scale * log2(e) == 1 (q pre-multiplied)
 * and 0 < score_bound <= orv_attention_static_limit(1): the caller checks and otherwise uses the row-major entry points. */
int orv_attention_fwd_packed(const void* qkv, int ld_qkv, void* out, float* lse, int B, int S, int H, float scale, float score_bound,
                             void* stream);
/* orv_attention_fwd_bounded with a workspace: at shapes whose grid ends in a small extra round of workgroups (the headline shape:
 * 1560 workgroups on 512 slots) the items of that round are cut into key ranges whose unnormalised partial results meet in `ws`
 * (fixed summation order: deterministic).  ws >= orv_attention_ws_bytes(B, S, H) bytes, 256-byte aligned, reusable across calls on one
 * stream; ws == NULL or an unsplit shape (ws_bytes 0): exactly orv_attention_fwd_bounded. */
size_t orv_attention_ws_bytes(int B, int S, int H);
int orv_attention_fwd_bounded_ws(const void* qkv, int ld_qkv, void* out, int ld_out, float* lse, int B, int S, int H, float scale,
                                 float score_bound, void* ws, size_t ws_bytes, void* stream);
/* Largest score_bound for which orv_attention_fwd_bounded(_dev) runs the fixed-shift softmax (16-byte aligned out: 90; else 60). */
float orv_attention_static_limit(int aligned_out);
/* orv_attention_fwd_bounded with the bound as ONE fp32 in device memory (the training step recomputes the per-layer bounds on the
 * device after every optimizer update, train_cogvideox_control_to_video_sft.py:1095-1104: no device -> host read per step).  Both
 * softmax forms are launched; the workgroups of the one the scalar does not select exit at once. */
int orv_attention_fwd_bounded_dev(const void* qkv, int ld_qkv, void* out, int ld_out, float* lse, int B, int S, int H, float scale,
                                  const float* score_bound_dev, void* stream);

/* -- sampler ---------------------------------------------------------------------------------- */
/* One fused scheduler update on n elements (cogvideox_control.py:1433-1459 + diffusers
 * CogVideoXDDIMScheduler.step / CogVideoXDPMScheduler.step, v-prediction):
 *   v    = guidance ? v_u + guidance_scale * (v_c - v_u) : v_c            (CFG, :1440-1443)
 *   x0   = sa * x - sb * v
 *   d    = old_x0 ? m3 * x0 - m4 * old_x0 : x0
 *   x'   = cx * x + cd * d + cn * noise                                   (noise may be NULL when cn == 0)
 * x, x' bf16 (the pipeline's latents dtype); v_c/v_u bf16 model outputs; x0_out/old_x0 fp32; noise fp32.
 * DDIM: cx = a_t, cd = b_t, cn = 0.  DPM: cx = m1, cd = -m2, cn = m_noise. Coefficients are computed on the host in
 * float64 (orv_amd/schedulers.py) exactly as the reference does. */
int orv_sched_step(const void* x, const void* v_c, const void* v_u, float guidance_scale, const float* old_x0,
                   const float* noise, void* x_out, float* x0_out, float sa, float sb, float m3, float m4, float cx,
                   float cd, float cn, long n, void* stream);

/* mean + exp(0.5*clamp(logvar,-30,20)) * eps, times `scale` (diffusers DiagonalGaussianDistribution.sample used at
 * cogvideox_control.py:1173-1177,1334-1358; train_cogvideox_control_to_video_sft.py:890-898) with the
 * [B,2C,F,H,W] -> [B,F,C,H,W] permute fused.  moments bf16, eps fp32 [B,C,F,H,W], out bf16. */
int orv_gaussian_sample(const void* moments, const float* eps, void* out, int B, int C, int F, int HW, float scale,
                        void* stream);

/* -- VAE: AutoencoderKLCogVideoX decode / encode, reference call sites
 *    orv/models/cogvideox_control.py:1476-1479 (decode_latents) and :1161-1166 (vae.encode of the reference frame); the
 *    arithmetic is diffusers' (absent: parity unpinned, oracle/vae.py).  Activations are channels-last bf16 [B,T,H,W,C]. -- */
/* Patch matrix of a causal 3-D / per-frame 2-D convolution for orv_gemm_bf16: dst[mc, Kpad] rows for voxels [m0, m0+mc) of the
 * [B*T*H*W] output grid, column = tap * C + channel (tap = (dt*kh + dy)*kw + dx), tail columns zero.  Folded into the gather:
 * replicate-first-frame temporal padding (CogVideoXCausalConv3d), zero spatial padding (pad_lo in front), spatial stride 1|2,
 * nearest x2 upsampling of the source in H,W (ups_s) and in T (ups_t: 1 = every frame doubled, 2 = first frame kept).
 * t_shift = frames of temporal context the caller prepended to src (the conv_cache diffusers carries from one frame batch to
 * the next): output frame t reads source frames t + t_shift - (kt-1) + dt; with 0 the first frame is replicated instead. */
int orv_vae_im2col(const void* src, void* dst, int B, int Ts, int Hs, int Ws, int C, int T, int H, int W, int kt, int kh, int kw,
                   int stride, int pad_lo, int ups_s, int ups_t, int t_shift, int Kpad, long m0, long mc, void* stream);
/* The same convolution as ONE launch, without the patch matrix (implicit GEMM): the GEMM's A operand is gathered by the LDS-DMA
 * from the channels-last source (geometry fields as orv_vae_im2col's), W = packed weights [N, taps*C] (column = tap*C + ci),
 * epilogue 0 (bias) or 2 (R + acc + bias: the resnet's residual add).  g->A is ignored; g->M = B*T*H*W, g->K = taps*C.
 * Requires C % 64 == 0 (one 64-deep K-tile = 64 channels of one tap). */
typedef struct {
    const void* src;
    int B, Ts, Hs, Ws, C;        /* source [B, Ts, Hs, Ws, C] bf16 */
    int T, H, W;                 /* output voxel grid */
    int kt, kh, kw, stride, pad_lo, ups_s, ups_t, t_shift;
} orv_conv_t;
int orv_conv_gemm_bf16(const orv_gemm_t* g, const orv_conv_t* c, void* stream);
/* GroupNorm statistics: sums[B, G, 2] = (sum, sum of squares) of x[B, N, C] per group, fp32, no atomics (bit-reproducible).
 * scratch: orv_vae_groupnorm_scratch(B, N, C, G) floats. */
long orv_vae_groupnorm_scratch(int B, long N, int C, int G);
int orv_vae_groupnorm_stats(const void* x, float* sums, float* scratch, int B, long N, int C, int G, void* stream);
/* out = act(GroupNorm(x) [* zy[z(v)] + zb[z(v)]]): affine GroupNorm from `sums`, CogVideoXSpatialNorm3D modulation by
 * conv_y(zq) / conv_b(zq) given at LATENT resolution [B,Tz,hz,wz,C] and looked up by F.interpolate(nearest) index rules
 * (first frame of an odd-length clip apart), optional SiLU.  zy == zb == NULL: plain GroupNorm.
 * out is [B, out_lead + T, H, W, C]: the first out_lead (0..8) frames of every clip are left untouched - the caller puts the
 * causal context of the NEXT convolution there (conv_cache frames or copies of the first frame; orv_conv_t.t_shift = out_lead). */
int orv_vae_norm_apply(const void* x, void* out, const float* sums, const void* gamma, const void* beta, const void* zy,
                       const void* zb, int B, int T, int H, int W, int C, int G, int Tz, int hz, int wz, float eps, int silu_act,
                       int out_lead, void* stream);
/* Seam blend of the tiled decode / encode (diffusers AutoencoderKLCogVideoX.blend_v / blend_h, enabled by the reference at
 * /root/reference/orv/pipeline/inference_control_to_video.py:98-99): channels-last tiles a [outer, Ha, Wa, C], b [outer, Hb, Wb, C]
 * bf16; in place on b: vertical (horizontal = 0): b[.., y, :] = a[.., Ha - extent + y, :] (1 - y / extent) + b[.., y, :] y / extent
 * for y < extent (Wa == Wb); horizontal: the same along x with a = the tile to the left (Ha == Hb). */
int orv_vae_blend(const void* a, void* b, int outer, int Ha, int Wa, int Hb, int Wb, int C, int extent, int horizontal,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ORV_MI355_H */
