/*
 * emcid_hip.h — C ABI of the MI355X (gfx950) kernels under EMCID's closed-form mass-edit path.
 *
 * The reference (SilentView/EMCID) has no FFI: its boundary is the Python call
 * apply_emcid_to_text_encoder (reference: emcid/emcid_main.py:769).  This library sits UNDER the
 * Python mirror of that call (emcid_amd/emcid_main.py) and replaces the ATen ops the reference
 * dispatches on the hot path.  Each entry point cites the reference line(s) it replaces.
 *
 * Conventions: all pointers are DEVICE pointers (HBM) unless named *_host; matrices are row-major
 * with explicit leading dimensions in ELEMENTS; `stream` is a hipStream_t passed as void*; every
 * call is asynchronous on `stream` and returns 0 (EMCID_OK) or a negative error code — no
 * exceptions cross the ABI; the caller owns every buffer.  f64 buffers and leading dimensions must
 * be 16-byte aligned / even (checked: EMCID_ERR_BAD_ARG).  The device that owns the buffers and `stream`
 * must be the CURRENT device of the calling thread (hipSetDevice) — emcid_amd/hip.py makes it so around every
 * call; internal per-device state (capture streams, cached graphs, function attributes) is keyed by it.
 */
#ifndef EMCID_HIP_H
#define EMCID_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMCID_OK 0
#define EMCID_ERR_BAD_ARG (-1)
#define EMCID_ERR_HIP (-2)
#define EMCID_ERR_WORKSPACE (-3)

#define EMCID_ABI_VERSION 14

/* ABI version of the loaded library (host-only, no GPU needed). */
int emcid_abi_version(void);
/* Human-readable text of the last error on this thread (host-only). */
const char* emcid_last_error(void);

/* Optional per-kernel-class timing with HIP events recorded on the launch stream (bench.py's live
 * roofline measurement).  class ids: 0 prep, 1 assemble(SYRK), 2 chol_leaf, 3 chol_panel, 4 chol_trail,
 * 5 trsm_diag, 6 trsm_update, 7 delta_w, 8 gram, 9 gather, 10 dgemm, 11 misc, 12 inverse build, 13 chol_inner,
 * 14 inverse apply (GEMMs against the explicit inverse factor), 15 inverse of the 512-blocks (many small products).  enable(mask) resets the
 * log; collect() synchronises the recorded events and returns summed milliseconds and launch counts. */
#define EMCID_PROF_CLASSES 16
int emcid_profile_enable(unsigned class_mask);
int emcid_profile_collect(double* ms_per_class_host, int64_t* launches_per_class_host, int n_classes);

/* ---------------------------------------------------------------------------------------------
 * Stage 0 — second moment.  Replaces `self.mom2 += a.t().mm(a)` (reference:
 * util/runningstats.py:493) for a = X[t, d] fp32.  Only the LOWER triangle (incl. diagonal) of
 * G[d, d] is accumulated (SYRK, T*d^2 flops instead of the reference's 2*T*d^2 GEMM);
 * emcid_symmetrize_lower_f32 mirrors it when the moment is read (runningstats.py:499-507).
 * ksplit > 1 splits the token dimension over workgroups that add their partial tiles with fp32
 * atomics (not bitwise reproducible run to run); ksplit == 1 is deterministic.
 * ------------------------------------------------------------------------------------------- */
int emcid_gram_accumulate_f32(const float* X, int64_t t, int64_t d, int64_t ldx,
                              float* G, int64_t ldg, int ksplit, void* stream);
int emcid_symmetrize_lower_f32(float* G, int64_t d, int64_t ldg, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K/Z assembly gather.  Replaces the N*P Python indexings + per-request `.mean(0)` of
 * get_module_input_output_at_words (reference: emcid/compute_z.py:2311-2325):
 *   out[n, :] = ( sum_{p in [seg[n], seg[n+1])} act[p, idx[p], :] ) / (seg[n+1]-seg[n])
 * summed in prompt order, fp32, true division (bit-compatible with torch-CPU mean over <= 8 rows).
 * act is [B, S, c] with row stride lds_ (elements between consecutive s) and batch stride ldb.
 * ------------------------------------------------------------------------------------------- */
int emcid_gather_mean_f32(const float* act, int64_t B, int64_t S, int64_t c, int64_t ldb, int64_t lds_,
                          const int64_t* idx, const int64_t* seg, int64_t N,
                          float* out, int64_t ldo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K/Z assembly forward — fused attention for CLIP text shapes (fp32, S <= 128, head_dim <= 128):
 *   out[b, i, h, :] = softmax_j(scale * q[b,h,i,:].k[b,h,j,:] + mask[b,i,j]) v[b,h,j,:]
 * Replaces the eager bmm/softmax/bmm of transformers' CLIPAttention inside the hooked forward the reference
 * runs (reference: emcid/compute_z.py:2296-2308).  q/k/v are [B,H,S,D] VIEWS with element strides
 * (sb, sh, ss) and unit stride over D (the layout HF produces: proj(x).view(B,S,H,D).transpose(1,2));
 * out is [B,S,H,D] contiguous.  mask_kind: 0 none; 1 uint8 keep-mask; 2 float additive; mask element
 * (b, i, j) is at b*mb + i*mi + j (broadcast over heads).  causal != 0 additionally drops j > i.
 * ------------------------------------------------------------------------------------------- */
int emcid_attention_f32(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t ss,
                        const void* mask, int mask_kind, int64_t mb, int64_t mi, int causal, float scale,
                        int64_t B, int64_t H, int64_t S, int64_t D, float* out, void* stream);

/* Attention over a token trie (prefix-deduplicated causal forward): k/v [U, H*D] fp32 rows with leading
 * dimension ld (one row per distinct prompt prefix); anc [U, anc_ld] int32 lists each node's ancestor chain
 * root..node (depth[u]+1 entries, <= 128); rows (optional, n_rows int32) selects the query nodes, NULL = all
 * U nodes (then n_rows = U); q [n_rows, H*D] (leading dimension ldq) holds the queries IN QUERY ORDER.
 * out [n_rows, H*D]:  out[i] = softmax_j(scale * q[i].k[anc[u][j]]) v[anc[u][j]],  u = rows ? rows[i] : i.
 * head_dim D <= 64, multiple of 4. */
int emcid_tree_attention_f32(const float* q, int64_t ldq, const float* k, const float* v, int64_t ld, const int* anc, int64_t anc_ld,
                             const int* depth, const int* rows, int64_t n_rows, int64_t H, int64_t D, float scale,
                             float* out, int64_t ldo, void* stream);

/* y = x * sigmoid(1.702 x), fp32, n elements (CLIP quick_gelu inside the same hooked forward; one HBM pass
 * instead of the framework's three).  x may alias y. */
int emcid_quick_gelu_f32(const float* x, float* y, int64_t n, void* stream);

/* Y[M,N] = act(X[M,K] W[N,K]^T + bias[N]) + residual[M,N]: a row-wise projection of the text-encoder forward (nn.Linear inside
 * CLIPTextModel.forward, which the reference runs for every prompt, emcid/compute_z.py:2296-2316) with its element-wise
 * neighbours fused: bias (may be NULL), act = 0 none / 1 quick_gelu x*sigmoid(1.702x) / 2 erf-gelu, residual (may be NULL,
 * added after the activation).  fp32 in, exact-f32 MFMA accumulate, fp32 out; X and W row-major with K contiguous, K % 16 == 0,
 * ldx / ldw % 4 == 0, 16-byte aligned.  cfg: -1 = pick the tile by the launch's fill of the 256 compute units; 0 = 160 x 128,
 * 1 = 128 x 128, 2 = 256 x 128, 3 = 64 x 64.  Y may alias residual (each element is read then written by the same lane). */
int emcid_linear_f32(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual,
                     int64_t ldr, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, int act, int cfg, void* stream);

/* The same projection with a workspace, for launches of few tiles on a long K (fc2 of a 100-concept edit, the mean keys of a
 * 1 000-concept one through fc2): given `workspace` (emcid_linear_workspace_bytes() bytes of HBM, 16-byte aligned, ZEROED ONCE
 * by the caller when allocated — every launch leaves its ticket counters zero —, not shared by launches that may overlap on
 * different streams) the automatic choice (cfg = -1) cuts the K range of every 128 x 128 tile over 2..8 workgroups when at most
 * 128 tiles would otherwise run and K >= 2048 ("split-K": partial tiles meet in the workspace and are summed in part order —
 * bit-reproducible from call to call, different from emcid_linear_f32's result by the summation order only).  cfg bits 7-10 =
 * parts force that form (tile 1, bit 6 set).  workspace == NULL: exactly emcid_linear_f32. */
int64_t emcid_linear_workspace_bytes(void);
int emcid_linear_ws_f32(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual,
                        int64_t ldr, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, int act, int cfg, void* workspace,
                        int64_t workspace_bytes, void* stream);

/* ---- the same projections at fp32 accuracy on the 16-bit matrix pipe ("split fp16", csrc/gemm_sp16.hip) ------------------------
 * A split matrix stands for an fp32 matrix [rows, K] (K % 8 == 0): every row r is carried under a power-of-two scale 2^e_r (the
 * row's largest magnitude lands in [2^14, 2^15)) as hi = fp16(x 2^e), lo = fp16(x 2^e - hi), i.e. x = (hi + lo) 2^-e to 22-23
 * significant bits.  `planes`: one 4-byte unit per element like the fp32 matrix (leading dimension ldp in those units, % 4 == 0);
 * inside a row, groups of 8 consecutive k as [hi k..k+7 (16 bytes)][lo k..k+7 (16 bytes)].  `inv_scale[r]` = 2^-e_r.
 * emcid_split_rows_f16 converts fp32 rows (nn.Linear weights once per weight version, activations once per producer).
 * max_row_norm (optional, one float the CALLER zeroed): raised to the largest Euclidean norm of a row (x 1.0001) — what
 * emcid_add_layernorm_sp16 needs of a weight to bound the rows of the projection's output. */
int emcid_split_rows_f16(const float* X, int64_t ldx, int64_t rows, int64_t K, void* planes, int64_t ldp, float* inv_scale,
                         float* max_row_norm, void* stream);

/* Y[M,N] = act(X W^T + bias) + residual like emcid_linear_f32 (the same nn.Linear calls of CLIPTextModel.forward,
 * emcid/compute_z.py:2296-2316), with X [M,K] and W [N,K] given as split matrices: three f16 MFMAs per k-step
 * (hi.hi + hi.lo + lo.hi, fp32 accumulate; the dropped lo.lo term is <= 2^-22 of a product), scales undone exactly in the
 * epilogue.  K % 32 == 0.  Outputs: Y fp32 (may be NULL) and / or Yp = the result as a split matrix for the next projection,
 * under the CALLER's per-row scale y_scale[m] = 2^e (a bound on the row's magnitude is enough: |Y[m][n]| y_scale[m] < 2^15
 * must hold; NULL = 1); N % 8 == 0 for Yp.  ldx, ldw < 2^20.  cfg: -1 auto (by the number of tiles each form would give); 0:
 * 128 x 128 tiles, 1: 80 x 128, 2: 64 x 64, 3: 160 x 128 — four waves, operands by LDS-DMA (buffer_load ... lds into an
 * XOR-swizzled image, no staging registers, no LDS stores), three v_mfma_f32_16x16x32_f16 per k-step and accumulator block,
 * epilogue through LDS (whole-row stores); 4: 64 x 64 with register-staged operands on v_mfma_f32_32x32x16_f16 (launches of less
 * than one tile per compute unit); + 16 / + 48 with cfg 0: timing-only builds (results wrong).  The forms sum in different
 * orders: compare, do not equate. */
int emcid_linear_sp16_f32(const void* Xp, int64_t ldx, const float* x_inv_scale, const void* Wp, int64_t ldw,
                          const float* w_inv_scale, const float* bias, const float* residual, int64_t ldr, float* Y, int64_t ldy,
                          void* Yp, int64_t ldp, const float* y_scale, int64_t M, int64_t N, int64_t K, int act, int cfg,
                          void* stream);

/* Diagnostic (ABI 13): until called again with NULL, the 128 x 128 projection launches (cfg 0) run a stamped
 * build: per workgroup {shader clock at the K loop's start, at its end, 100 MHz clock at its start, at its end, 100 MHz clock at
 * kernel entry, at kernel exit, -, -} at stamps_dev[8 * blockIdx.x] — the in-kernel clock under load and the split of a
 * workgroup's life into prologue / K loop / epilogue (scripts/mb_linear_sp16_r5.py). */
int emcid_debug_linear_sp16_stamps(long long* stamps_dev);
/* Stage 0 on the same matrix path: lower(G) += X^T X for token rows X [t, d] fp32 (reference: util/runningstats.py:469-511,
 * mom2 += a.t().mm(a)) like emcid_gram_accumulate_f32, with X^T carried as split-fp16 planes under per-FEATURE scales (the
 * contraction runs over the tokens; scales per 32 768-token chunk) and three f16 MFMAs per k-step; the lower 128 x 128 tiles, the
 * token range of a tile cut into parts whose scaled results are added into G with fp32 atomics (G is an accumulator: sums in no
 * fixed order, like the exact-f32 kernel's multi-slab mode: two runs of one job agree to rounding, not to the bit — the exact-f32
 * kernel with ksplit = 1 is the bit-reproducible form).  d % 4 == 0; workspace: emcid_gram_sp16_workspace_bytes_for(d, t).
 * row_weight (ABI 14; [t] fp32 or NULL): row r enters as fl32(row_weight[r] * X[r, :]) — the packed Stage-0 forward's
 * square-root multiplicities (emcid_amd/layer_stats.py), applied where the rows are read instead of in a pass of their own. */
int64_t emcid_gram_sp16_workspace_bytes(int64_t d);                       /* for any t */
int64_t emcid_gram_sp16_workspace_bytes_for(int64_t d, int64_t t);       /* for batches of at most t rows (ABI 14) */
int emcid_gram_accumulate_sp16_f32(const float* X, const float* row_weight, int64_t t, int64_t d, int64_t ldx, float* G, int64_t ldg,
                                   void* workspace, int64_t workspace_bytes, void* stream);

/* Producers that write their result straight as a split-fp16 matrix for the projection that consumes it (no fp32 round trip
 * through HBM, no separate split pass):
 *  - emcid_add_layernorm_sp16 / emcid_embed_layernorm_sp16: emcid_add_layernorm_f32 / emcid_embed_layernorm_f32 with z as planes
 *    [rows, cols] + inv_scale [rows] under the row's own scale (z itself optional, NULL = not written); cols % 8 == 0, <= 2048.
 *    bound = {largest Euclidean row norm of the consuming projection's weight (emcid_split_rows_f16's max_row_norm), largest
 *    |bias|} (device, optional): then out_scale[r] = 2^e (and out_scale[rows + r] = 2^-e: 2 rows floats) with
 *    (|z_r| bound[0] + bound[1]) 2^e in [2^14, 2^15) — the y_scale
 *    under which emcid_linear_sp16_f32 may write act(z W^T + b) as planes (|act(t)| <= |t| for quick_gelu, erf-gelu, none).
 *  - emcid_tree_attention_sp16: emcid_tree_attention_f32 with the [n_rows, H D] result as planes + inv_scale; chains of at most
 *    16 nodes, D % 8 == 0 (emcid_tree_attention_sp16_supported). */
int emcid_add_layernorm_sp16(const float* a, int64_t lda, const float* b, int64_t ldb, const float* gamma, const float* beta,
                             float eps, int64_t rows, int64_t cols, float* y, float* z, void* planes, int64_t ldp,
                             float* inv_scale, const float* bound, float* out_scale, void* stream);
int emcid_embed_layernorm_sp16(const float* tok, int64_t ld_tok, int64_t n_tok, const float* pos, int64_t ld_pos, int64_t n_pos,
                               const int64_t* token, const int* position, const float* gamma, const float* beta, float eps,
                               int64_t rows, int64_t cols, float* y, float* z, void* planes, int64_t ldp, float* inv_scale,
                               void* stream);
int emcid_tree_attention_sp16_supported(int64_t anc_ld, int64_t H, int64_t D);
int emcid_tree_attention_sp16(const float* q, int64_t ldq, const float* k, const float* v, int64_t ld, const int* anc,
                              int64_t anc_ld, const int* depth, const int* rows, int64_t n_rows, int64_t H, int64_t D, float scale,
                              void* planes, int64_t ldp, float* inv_scale, void* stream);

/* ---- stale-cache guard (ABI 13; csrc/clip_layers.hip) ------------------------------------------------------------------------------
 * The forward's weight-derived caches (stacked q | k | v, split-fp16 planes, native layer structs) follow torch's in-place version
 * counter, which a write through `param.data` or a raw pointer does not move.  emcid_fingerprint_store leaves {pointer, bytes,
 * fingerprint} of n weights' BYTES (up to 4 096 evenly spaced 16-byte vectors each, mixed with their index; host arrays of
 * pointers / sizes / slots, one workgroup per weight) in their slots of a device table of table_slots x 4 int64 after cache
 * entries were made; emcid_fingerprint_check recomputes the fingerprints of a
 * slot range of at most 256 slots (empty slots and those whose bit is set in the 4 host words of skip_mask are skipped) and ORs
 * 1 into *flag on a mismatch — read back with the call's one final synchronisation.
 * bytes % 16 == 0.  Every non-empty slot of a checked range must still point at live memory (the host checks tensor identity
 * and address first).  No reference counterpart: the reference reads every weight live in every forward. */
int emcid_fingerprint_store(int64_t n, const void* const* data, const int64_t* bytes, const int64_t* slots, void* table,
                            int64_t table_slots, void* stream);
int emcid_fingerprint_check(const void* table, int64_t table_slots, int64_t first_slot, int64_t n_slots, const uint64_t* skip_mask,
                            int* flag, void* stream);

/* ---- native layer runner of the trie forward (csrc/clip_layers.hip) ---------------------------------------------------------------
 * One C call issues all launches of a run of CLIP text-encoder layers on the split-fp16 projections (the forward the reference
 * gets from CLIPTextModel.forward, emcid/compute_z.py:2296-2316, inside the layer loop of emcid/emcid_main.py:981-1073): the same
 * kernels in the same order as the per-launch Python path, bit-identical results.  A layer's weights as split matrices
 * (emcid_split_rows_f16; q | k | v stacked as one [3h, h] matrix; biases may be NULL; fc1_bound = the {max row norm, max |bias|}
 * pair of fc1), LayerNorm parameters, act = 0 none / 1 quick_gelu / 2 erf-gelu. */
typedef struct emcid_clip_layer_sp16 {
    const float* ln1_gamma; const float* ln1_beta;
    const float* ln2_gamma; const float* ln2_beta;
    const void* qkv_planes; const float* qkv_inv_scale; const float* qkv_bias;
    const void* out_planes; const float* out_inv_scale; const float* out_bias;
    const void* fc1_planes; const float* fc1_inv_scale; const float* fc1_bias; const float* fc1_bound;
    const void* fc2_planes; const float* fc2_inv_scale; const float* fc2_bias;
    float ln1_eps, ln2_eps;
    int32_t act, reserved;
} emcid_clip_layer_sp16;

int64_t emcid_clip_workspace_bytes(int64_t rows, int64_t h, int64_t d);

/* Attention block + fc1 of one layer on the trie rows (anc / depth as emcid_tree_attention_f32; hs [rows, h] fp32 and its LN1 as
 * planes x_planes / x_inv_scale come in).  rows_sel == NULL: every node (n_sel = rows); else k | v for every node and q,
 * attention, out-projection, LN2, fc1 for the n_sel selected nodes only (the last edited layer's query rows).  Out: mid [n_sel, h]
 * (residual stream after the attention block), f_planes [n_sel, d] + f_scale [2, n_sel] (scale, inverse scale) = act(fc1(LN2(mid)))
 * as a split matrix, f_f32 (optional) its fp32 twin (the keys of an edited layer).  workspace: emcid_clip_workspace_bytes. */
int emcid_clip_layer_head_sp16(const emcid_clip_layer_sp16* L, int64_t rows, int64_t h, int64_t d, int64_t heads, float attn_scale,
                               const int* anc, int64_t anc_ld, const int* depth, const int* rows_sel, int64_t n_sel,
                               const float* hs, const void* x_planes, const float* x_inv_scale, float* mid, void* f_planes,
                               float* f_scale, float* f_f32, void* workspace, int64_t workspace_bytes, void* stream);

/* hs_out [n, h] = f W2^T + b2 + mid (fc2 + residual add; hs_out may not alias mid), then, if next_ln_gamma != NULL, the next
 * layer's LN1 of it as planes. */
int emcid_clip_layer_tail_sp16(const emcid_clip_layer_sp16* L, int64_t n, int64_t h, int64_t d, const void* f_planes,
                               const float* f_inv_scale, const float* mid, float* hs_out, const float* next_ln_gamma,
                               const float* next_ln_beta, float next_ln_eps, void* x_planes, float* x_inv_scale, void* stream);

/* Everything of an edited layer behind emcid_clip_layer_head_sp16 from one call — the body of the reference's layer loop
 * (emcid/emcid_main.py:981-1073) as one launch sequence, single rank, lam C' factored and in HBM (emcid_factor_cov_f64): keys
 * K_out [N, d] = per-request means of f_f32 [n_rows, d] at the prompts' lookup rows (lookup [B], seg [N + 1]); Zc_out [N, h] =
 * fc2(K_out) with the CURRENT weight: k_planes [N, d] / k_inv_scale [N] given (scratch) = on the split-fp16 kernel against the
 * layer's own fc2 planes, which must then be the planes of W as it is; both NULL = on the exact-f32 kernel against W (linear_ws
 * as emcid_linear_ws_f32, may be NULL); the dual solver's
 * apply-only form (arguments as emcid_edit_dual_apply_stage1_f64 / _stage2_f64): W = W0 + float(U), dW (may be NULL); then, if
 * hs_out != NULL, the new W split into the layer's OWN fc2 planes (in place) and fc2 + residual + the next layer's LN1 as in
 * emcid_clip_layer_tail_sp16.  hs_out == NULL (the last edited layer): nothing after the weight update. */
int emcid_clip_edit_layer_tail_sp16(const emcid_clip_layer_sp16* L, int64_t n_rows, int64_t h, int64_t d, const float* f_f32,
                                    const void* f_planes, const float* f_inv_scale, const float* mid, const int64_t* lookup,
                                    const int64_t* seg, int64_t B, int64_t N, const float* zs_t, double edit_weight,
                                    int layers_left, double lam_ratio, const void* cov_factor_ws, int64_t n_layers,
                                    int64_t layer_index, int use_inverse, const float* W0, float* W, float* dW, float* K_out,
                                    float* Zc_out, void* k_planes, float* k_inv_scale, void* dual_ws, int64_t dual_ws_bytes,
                                    int* info_dev, void* linear_ws, int64_t linear_ws_bytes, float* hs_out,
                                    const float* next_ln_gamma, const float* next_ln_beta, float next_ln_eps, void* x_planes,
                                    float* x_inv_scale, void* stream);

/* n_layers whole layers on every node, hs in place; x_planes / x_inv_scale: LN1 of hs for layers[0] in, LN1 of the result under
 * next_ln_* out (next_ln_gamma == NULL: not computed). */
int emcid_clip_layers_sp16(const emcid_clip_layer_sp16* layers, int64_t n_layers, int64_t rows, int64_t h, int64_t d, int64_t heads,
                           float attn_scale, const int* anc, int64_t anc_ld, const int* depth, float* hs, void* x_planes,
                           float* x_inv_scale, const float* next_ln_gamma, const float* next_ln_beta, float next_ln_eps,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* y = a + b ; z = LayerNorm(y) * gamma + beta over the last dimension (biased variance, eps inside the root, as
 * torch.nn.LayerNorm) — the residual add and the LayerNorm after it of every block of the same hooked forward, one
 * pass.  a/b: [rows, cols] with row strides lda/ldb (elements); y, z: [rows, cols] contiguous; cols % 4 == 0, <= 8192.
 * b == NULL: z = LayerNorm(a) (the residual add already happened in emcid_linear_f32's epilogue); y == NULL: the sum is not
 * written. */
int emcid_add_layernorm_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* gamma, const float* beta,
                            float eps, int64_t rows, int64_t cols, float* y, float* z, void* stream);

/* y[r] = tok[token[r]] + pos[position[r]] ; z = LayerNorm(y): the encoder's embedding stage (transformers CLIPTextEmbeddings,
 * called inside the hooked forward of emcid/compute_z.py:2296-2308) fused with the first layer's LN1.  tok [n_tok, cols], pos
 * [n_pos, cols] (row strides ld_tok / ld_pos); token int64 [rows] in [0, n_tok), position int32 [rows] in [0, n_pos) — the
 * caller guarantees the ranges (HF raises an index error instead); y, z [rows, cols] contiguous. */
int emcid_embed_layernorm_f32(const float* tok, int64_t ld_tok, int64_t n_tok, const float* pos, int64_t ld_pos, int64_t n_pos,
                              const int64_t* token, const int* position, const float* gamma, const float* beta, float eps,
                              int64_t rows, int64_t cols, float* y, float* z, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stage 2 — per-layer closed form (reference: emcid/emcid_main.py:1016-1061).
 *
 * emcid_edit_workspace_bytes: bytes of f64 workspace emcid_edit_layer_f64 needs for (N, d, h).
 *
 * emcid_edit_layer_f64 does, for one edited layer, entirely on `stream`:
 *   s    = sqrt(edit_weight/0.5)
 *   Kt64 = double(K) * s                                   (:1040-1043)   K   [N, d] fp32
 *   Rt   = double(zs_t - Zc) * s / layers_left             (:1016,:1049)  zs_t, Zc [N, h] fp32
 *   A    = lam * double(C * (1-edit_weight) / 0.5) + Kt64^T Kt64   (:1037,:1046)  C [d, d] fp32
 *   A    = L L^T  (blocked Cholesky, fp64 MFMA); Xt = Kt64 A^{-1}  (:1045-1048; Xt[n,:] = adj_k[:,n])
 *   U    = Rt^T Xt            [h, d] f64                     (:1050)
 *   W    = W0 + float(U)      [h, d] fp32                    (:1061)
 * Outputs (any may be NULL to skip): Xt_out [N, d] f64 (adj_k^T), Rt_out [N, h] f64 (resid^T),
 * dW_out [h, d] fp32 (= float(U)), W [h, d] fp32.  W0 may alias W.
 * info_dev: device int; set to (1 + index of the first non-positive pivot) if A is not SPD
 * (the reference's LU would still return numbers; the caller decides what to do), else left 0.
 * ------------------------------------------------------------------------------------------- */
int64_t emcid_edit_workspace_bytes(int64_t N, int64_t d, int64_t h);

int emcid_edit_layer_f64(const float* K, const float* Zc, const float* zs_t, const float* C,
                         int64_t N, int64_t d, int64_t h,
                         double lam, double edit_weight, int layers_left,
                         const float* W0, float* W,
                         double* Xt_out, double* Rt_out, float* dW_out,
                         void* workspace, int64_t workspace_bytes, int* info_dev, void* stream);
int emcid_cov_inverse_f64(void* cov_factor_ws, int64_t n_layers, int64_t d, int64_t first_layer, int64_t count, void* stream);

/* Concept-sharded variant (multi-GPU): K, Zc, zs_t hold ALL N concepts (after the all-gather),
 * A is assembled and factored from all of them, but only the rows [n_lo, n_hi) go through the triangular
 * solves and the dW contraction.  U_partial [h, d] f64 receives sum_{n in shard} Rt[n,:]^T Xt[n,:]; the caller
 * sums it over ranks (RCCL all-reduce) and finishes with emcid_apply_update_f32.  Xt_out / Rt_out (optional)
 * receive the shard's rows, [n_hi-n_lo, d] and [n_hi-n_lo, h]. */
int emcid_edit_layer_shard_f64(const float* K, const float* Zc, const float* zs_t, const float* C,
                               int64_t N, int64_t d, int64_t h, double lam, double edit_weight, int layers_left,
                               int64_t n_lo, int64_t n_hi, double* U_partial, double* Xt_out, double* Rt_out,
                               void* workspace, int64_t workspace_bytes, int* info_dev, void* stream);
/* W = W0 + float(U) (optional) ; dW = float(U) (optional), n = h*d elements.  (:1061) */
int emcid_apply_update_f32(const double* U, const float* W0, float* W, float* dW, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dual (Woodbury) form of the same layer solve, for N < d:  with M = lam * C' (independent of the concepts)
 *     Xt = Kt64 (M + Kt64^T Kt64)^-1 = (I + Pt Kt64^T)^-1 Pt,      Pt = Kt64 M^-1 .
 * emcid_factor_cov_f64 factors M for ALL edited layers in one batched pass (their serial pivot spines overlap, and
 * the pass can run on a second stream underneath the encoder forward); each layer then only factors the
 * Np x Np matrix S = I + Pt Kt64^T (Np = N rounded up to 128).  Same inputs, same scaling rules and outputs as
 * emcid_edit_layer_f64; results agree to ~1e-10 relative (fp64 rounding of a different but exact identity).
 *   emcid_factor_cov_f64      C_host_list: HOST array of n_layers device pointers to C_l [d,d] fp32.
 *   emcid_cov_inverse_f64     X_l = inv(L_l), explicit, for layers [first_layer, first_layer + count) of the factored
 *                             workspace in batched launches (~d^3/3 flops per layer, GEMMs).  Solves against those M_l
 *                             can then run as two GEMMs against X_l (use_inverse = 1 below) instead of block
 *                             substitutions with L_l (use_inverse = 0, needs no X_l).  A range, so that the first
 *                             layer's edit can start on L alone while the later layers' X are built underneath it.
 *   emcid_edit_dual_stage1    Kt64, Rt, and rows [n_lo, n_hi) of Pt = Kt64 M^-1 (= (Kt64 X^T) X with use_inverse).
 *   emcid_edit_dual_pt        address of the Pt stack [Np, dp] inside the workspace (multi-GPU: all-gather rows there).
 *   emcid_edit_dual_stage2    needs all rows of Pt: S, its Cholesky, adj_k = (S^-1 Pt)^T [d,N], U = Rt^T Xt, W = W0 + float(U).
 * lam_ratio (every stage-1 entry point of the dual forms, and emcid_edit_dual_stage2_f64) = lam of THIS edit / the lam the
 * workspace was factored with.  chol(lam C') = sqrt(lam) chol(C'), so a factored workspace serves every lam: stage 1 scales
 * Kt64 and Rt by 1/sqrt(lam_ratio), which makes Yt, S, Z and U exactly those of the call's own lam (the reference's
 * mom2_update_weight sweep, experiments/emcid_test.py:924-930, then costs no factorization); 1.0 reproduces the factored lam
 * bit for bit.  edit_weight is NOT covered: C' = fl32(fl32(C (1 - e_w)) / 0.5) is rounded in fp32 per entry (:1037), so a
 * new edit_weight needs its own emcid_factor_cov_f64.
 * ------------------------------------------------------------------------------------------- */
int64_t emcid_cov_factor_workspace_bytes(int64_t n_layers, int64_t d);
int emcid_factor_cov_f64(const float* const* C_host_list, int64_t n_layers, int64_t d, double lam, double edit_weight,
                         void* workspace, int64_t workspace_bytes, int* info_dev, void* stream);
int64_t emcid_edit_dual_workspace_bytes(int64_t N, int64_t d, int64_t h);
int emcid_edit_dual_stage1_f64(const float* K, const float* Zc, const float* zs_t, int64_t N, int64_t d, int64_t h,
                               double edit_weight, int layers_left, double lam_ratio, const void* cov_factor_ws, int64_t n_layers,
                               int64_t layer_index, int64_t n_lo, int64_t n_hi, int use_inverse, void* workspace,
                               int64_t workspace_bytes, void* stream);
double* emcid_edit_dual_pt(void* workspace, int64_t N, int64_t d, int64_t h);
int emcid_edit_dual_stage2_f64(int64_t N, int64_t d, int64_t h, double lam_ratio, const float* W0, float* W, double* adjk_out, double* Rt_out,
                               float* dW_out, void* workspace, int64_t workspace_bytes, int* info_dev, void* stream);

/* Apply-only form of the dual solver (adj_k is never formed; used when only the edited weights are wanted):
 *   Yt = Kt64 L^-T (M = L L^T),  S = I + Yt Yt^T,  Z = S^-1 Rt,  U = (Z^T Yt) L^-1,  W = W0 + float(U).
 * stage1 computes rows [n_lo, n_hi) of Yt (one forward solve), emcid_edit_dual_yt gives the Yt stack [Np, dp] inside the
 * workspace (multi-GPU all-gather target), stage2 needs all rows of Yt. */
int emcid_edit_dual_apply_stage1_f64(const float* K, const float* Zc, const float* zs_t, int64_t N, int64_t d, int64_t h,
                                     double edit_weight, int layers_left, double lam_ratio, const void* cov_factor_ws, int64_t n_layers,
                                     int64_t layer_index, int64_t n_lo, int64_t n_hi, int use_inverse, void* workspace,
                                     int64_t workspace_bytes, void* stream);
double* emcid_edit_dual_yt(void* workspace, int64_t N, int64_t d, int64_t h);
int emcid_edit_dual_apply_stage2_f64(int64_t N, int64_t d, int64_t h, const void* cov_factor_ws, int64_t n_layers,
                                     int64_t layer_index, int use_inverse, int assembled, const float* W0, float* W,
                                     float* dW_out, void* workspace, int64_t workspace_bytes, int* info_dev, void* stream);
/* S = I + Yt Yt^T alone (then pass assembled = 1 to stage 2): lets the caller hang other work on the moment the
 * latency-bound Cholesky of S starts. */
int emcid_edit_dual_apply_assemble_f64(int64_t N, int64_t d, int64_t h, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fallback with the reference's own solver semantics.  torch.linalg.solve (reference: emcid/emcid_main.py:1045-1048) is
 * LAPACK getrf + getrs: LU with partial pivoting, which returns numbers for ANY nonsingular system, whereas the Cholesky
 * paths above stop at a non-positive pivot (info_dev).  emcid_edit_layer_lu_f64 runs one edited layer through a blocked
 * right-looking LU with row pivoting (largest magnitude in the column, lowest row on ties) and two substitutions for
 * all N right-hand sides — same inputs, scaling rules and outputs as emcid_edit_layer_f64, except that adj_k is
 * returned in the reference's orientation: adjk_out [d][N].  info_dev: 1 + column of an exactly-zero pivot (singular
 * matrix: torch raises there), else left untouched.  Built for being rare, not fast (~50 ms at d = 3072).
 * emcid_lu_solve_f64: the factor + solve alone (test hook): A [n][lda] is overwritten by P L U, B [n][ldb] by the
 * solution; piv_dev: n device ints.
 * ------------------------------------------------------------------------------------------- */
int64_t emcid_edit_lu_workspace_bytes(int64_t N, int64_t d, int64_t h);
int emcid_edit_layer_lu_f64(const float* K, const float* Zc, const float* zs_t, const float* C,
                            int64_t N, int64_t d, int64_t h, double lam, double edit_weight, int layers_left,
                            const float* W0, float* W, double* adjk_out, double* Rt_out, float* dW_out,
                            void* workspace, int64_t workspace_bytes, int* info_dev, void* stream);
int emcid_lu_solve_f64(double* A, int64_t lda, int64_t n, double* B, int64_t ldb, int64_t nrhs, int* piv_dev,
                       int* info_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Apply-only dual solver, COLUMN-SHARDED over ranks (multi-GPU; SURVEY.md §8e: concept stacks are all-gathered, the
 * solve itself is split so that no rank repeats another's GEMMs).  The d columns of Yt = Kt64 X^T (X = inv(L), needs
 * emcid_cov_inverse_f64 for the layer) are dealt to the ranks in 128-wide tiles; `tiles_host`: this rank's tile indices,
 * ascending, HOST array.  With Yc = this rank's columns:
 *     stage 1:  Kt64, Rt, Yc, partial S_r = Yc Yc^T (lower 128-tiles)          -> emcid_edit_dual_s()  [Np][Np] f64
 *     caller :  all-reduce (sum) of S over the ranks                            (RCCL over xGMI: 8 MB at N = 1000)
 *     stage 2:  S + I = L_S L_S^T, Z = S^-1 Rt, V = Z^T Yc, partial U_r = V X[tiles, :] -> emcid_edit_dual_u() [hp][dp] f64
 *     caller :  all-reduce (sum) of U, then emcid_apply_update2d_f32 (W = W0 + float(U), ldu = dp)
 * Summed over the ranks this is exactly emcid_edit_dual_apply_stage1/2 (same algebra, sums in another order).
 * ------------------------------------------------------------------------------------------- */
int emcid_edit_dual_cols_stage1_f64(const float* K, const float* Zc, const float* zs_t, int64_t N, int64_t d, int64_t h,
                                    double edit_weight, int layers_left, double lam_ratio, const void* cov_factor_ws, int64_t n_layers,
                                    int64_t layer_index, const int* tiles_host, int n_tiles, void* workspace,
                                    int64_t workspace_bytes, void* stream);
double* emcid_edit_dual_s(void* workspace, int64_t N, int64_t d, int64_t h);
double* emcid_edit_dual_u(void* workspace, int64_t N, int64_t d, int64_t h);
int emcid_edit_dual_cols_stage2_f64(int64_t N, int64_t d, int64_t h, const void* cov_factor_ws, int64_t n_layers,
                                    int64_t layer_index, const int* tiles_host, int n_tiles, void* workspace,
                                    int64_t workspace_bytes, int* info_dev, void* stream);
int emcid_apply_update2d_f32(const double* U, int64_t ldu, const float* W0, float* W, float* dW, int64_t h, int64_t d,
                             void* stream);

/* The stages of emcid_edit_layer_f64 as separate calls (used by tests and micro-benchmarks). */

/* A[d,d] (f64, ld lda, LOWER triangle valid) = lam_c * double(fl32(fl32(C*cw)/0.5f)) + Kt64^T Kt64,
 * Kt64 [Np, d] f64 (rows >= N must be zero).  (:1037, :1046) */
int emcid_assemble_spd_f64(const float* C, int64_t ldc, const double* Kt64, int64_t Np, int64_t d, int64_t ldk,
                           double lam, float cw, double* A, int64_t lda, void* stream);
/* Lower Cholesky A = L L^T of A[dp,dp] (dp a multiple of 128; a caller with d < dp identity-pads rows/cols
 * d..dp).  A's lower triangle is consumed (overwritten by partial Schur complements); the factor is written
 * to L (same leading dimension).  invdiag: f64 scratch of emcid_inverse_workspace_doubles(dp) elements; its
 * first ceil(dp/512) x [512][512] slots receive the inverses of the 512 x 512 diagonal blocks of L (the later
 * triangular solves are MFMA GEMMs against them).
 * replaces the getrf half of torch.linalg.solve (:1045). */
int64_t emcid_inverse_workspace_doubles(int64_t dp);
/* Diagnostic: runs the diagonal leaf on one contiguous 128x128 SPD block and writes shader-clock stamps of its
 * phase boundaries to stamps_dev[32] (load | per panel: factor, panel solve, trailing | inverse levels | store). */
int emcid_debug_leaf_stamps(const double* A, double* L, double* inv, int* info_dev, long long* stamps_dev, void* stream);
int emcid_cholesky_f64(double* A, double* L, int64_t dp, int64_t lda, double* invdiag, int* info_dev, void* stream);
/* Bt[Np, dp] := Bt A^{-1} given the factor from emcid_cholesky_f64; Yt is [Np, dp] scratch.
 * replaces the getrs half of torch.linalg.solve (:1045-1048). */
int emcid_cholesky_solve_f64(const double* L, int64_t dp, int64_t lda, const double* invdiag,
                             double* Bt, double* Yt, int64_t Np, int64_t ldb, void* stream);
/* U = Rt^T Xt (f64, optional) ; dW = float(U) (optional) ; W = W0 + float(U) (optional).  (:1050,:1061) */
int emcid_delta_w_f64(const double* Rt, int64_t ldr, const double* Xt, int64_t ldx, int64_t Np, int64_t h, int64_t d,
                      const float* W0, float* W, int64_t ldw, float* dW, double* U, void* stream);

/* Plain fp64 MFMA GEMM, exported for tests/micro-benchmarks:
 * C[M,N] = alpha * opA(A) opB(B) + beta * C ; ta/tb: 0 = stored [rows][K] (K contiguous), 1 = stored [K][rows]. */
int emcid_dgemm_f64(int ta, int tb, int64_t M, int64_t N, int64_t K, double alpha,
                    const double* A, int64_t lda, const double* B, int64_t ldb,
                    double beta, double* C, int64_t ldc, void* stream);
/* The same GEMM with the structure hints the solver uses (tests / micro-benchmarks of those shapes):
 * flags bits 0-3 = triangular operands (1: B(k,n)=0 for k>n, 2: B(k,n)=0 for k<n, 4: A(m,k)=0 for k>m, 8: A(m,k)=0 for
 * k<m), bit 4 = compute only output tiles that touch the lower triangle, bit 5 = pair mirrored tiles of the triangular
 * dimension in one workgroup; cfg: -1 auto, 0 = 128x128, 1 = 64x64, 2 = 32x64 tiles; ksplit: 0 auto, n > 0 = even n-way split of
 * K, n < 0 = fixed runs of |n| K-tiles (16 deep) per workgroup; splits need beta == 1 (partials are added with f64 atomics). */
int emcid_dgemm_ex_f64(int ta, int tb, int64_t M, int64_t N, int64_t K, double alpha,
                       const double* A, int64_t lda, const double* B, int64_t ldb,
                       double beta, double* C, int64_t ldc, int flags, int cfg, int ksplit, void* stream);

/* Stream-K form of the same GEMM for the two shapes whose work per output tile is uneven or too scarce for the chip —
 * a triangular B operand (flags bit 0 or 1 as above) or a lower-only square output (flags bit 4: SYRK-like, M == N) —
 * WITHOUT atomics (M and N up to 16 384 tiles of 128 x 128 together): the (tile, K-step) space is cut into `wgs` equal runs; a run's partial tiles go to `workspace`, the
 * contributor of a tile that takes the last ticket sums them in run order (bit-reproducible) and writes
 * C = alpha * A op(B) (+ diag_add on the diagonal).  C needs no initial value.  A is [M][K]; tb as in emcid_dgemm_f64.
 * workspace: emcid_streamk_workspace_bytes(wgs) bytes whose LAST 65536 bytes (the ticket counters) are zero on entry;
 * they are zero again on exit, so one zero-filled allocation serves any number of stream-ordered calls. */
int64_t emcid_streamk_workspace_bytes(int wgs);
int emcid_dgemm_streamk_f64(int tb, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                            const double* B, int64_t ldb, double* C, int64_t ldc, int flags, int wgs, double diag_add,
                            void* workspace, int64_t workspace_bytes, void* stream);

/* Diagnostic: until called again with NULL, every two-phase stream-K launch writes 8 int64 shader-clock values per workgroup
 * to stamps_dev (start, end, cycles in K loops / partial publishes / last-ticket reductions / epilogues, segments, run). */
int emcid_debug_streamk_stamps(long long* stamps_dev);
/* diagnostic: per-workgroup start/end stamps of the fused Cholesky step launches (leaf + trailing tiles + shadow product);
 * layout in csrc/spd_solve.hip; scripts/step_stamps.py reads it. */
int emcid_debug_step_stamps(long long* stamps_dev);

/* `batch` independent problems of one shape: C_b = alpha * opA(A_b) opB(B_b) + beta * C_b with A_b = A + b*sA etc.
 * (element strides).  Used for the per-edit Grams sum_r k_r k_r^T of the UCE closed form (reference
 * emcid/uce_train.py:170-176, :378-404, kept per edit there as well: one batch-2 forward and one outer-product sum per
 * (edit, projection)). */
int emcid_dgemm_batched_f64(int ta, int tb, int64_t M, int64_t N, int64_t K, double alpha,
                            const double* A, int64_t lda, int64_t sA, const double* B, int64_t ldb, int64_t sB,
                            double beta, double* C, int64_t ldc, int64_t sC, int64_t batch, void* stream);

/* W[h,d] += dW[h,d]  (final insert, reference: emcid_main.py:802-809 `w[...] += upd_matrix.float()`). */
int emcid_axpy_f32(float* W, const float* dW, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EMCID_HIP_H */
