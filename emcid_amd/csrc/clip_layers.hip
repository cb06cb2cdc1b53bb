// Native layer runner of the prefix-trie forward (emcid_amd/clip_forward.py) on the split-fp16 projections: ONE C call issues all
// launches of a run of CLIP text-encoder layers — the loop the reference gets from CLIPTextModel.forward
// (emcid/compute_z.py:2296-2316 runs the encoder; emcid/emcid_main.py:981-1073 is the layer loop around it) — instead of one
// ctypes call, two tensor allocations and ~15-20 us of interpreter time per launch.  A 100-concept edit is ~150 launches whose
// device time is under 2 ms; a 1 000-concept edit's forward outruns a Python launcher as well since the projections moved to the
// 16-bit matrix pipe.  Same kernels, same order, same arguments as the Python path: the results are bit-identical to it.
//
// Per layer (rows = trie nodes, h = hidden, d = intermediate; x = LN1(hs) arrives as split-fp16 planes):
//     qkv   = x Wqkv^T + b                       emcid_linear_sp16_f32            fp32 [rows, 3h]
//     ctx   = tree attention(q, k, v)            emcid_tree_attention_sp16        planes [sel, h]
//     mid   = ctx Wo^T + bo + hs                 emcid_linear_sp16_f32            fp32 [sel, h]
//     z     = LN2(mid)                           emcid_add_layernorm_sp16         planes [sel, h] + fc1's output scale
//     f     = act(z W1^T + b1)                   emcid_linear_sp16_f32            planes [sel, d] (+ fp32 twin for the keys)
//     hs'   = f W2^T + b2 + mid                  emcid_linear_sp16_f32            fp32 [sel, h]
//     x'    = LN1_next(hs')                      emcid_add_layernorm_sp16         planes [sel, h]
// (sel = all rows, or the query rows of the last edited layer: k | v for every node, q / attention / MLP for the selected ones).
#include "common.h"

#include <algorithm>

namespace emcid {

// dst[r] = src[idx[r]] for rows of `row_bytes` bytes (a multiple of 16); one workgroup per destination row
__global__ __launch_bounds__(256) void gather_rows16_kernel(const unsigned char* __restrict__ src, int64_t src_stride,
                                                             const int* __restrict__ idx, unsigned char* __restrict__ dst,
                                                             int64_t dst_stride, int row_bytes) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const int64_t r = blockIdx.x;
    const v4u* s = reinterpret_cast<const v4u*>(src + (int64_t)idx[r] * src_stride);
    v4u* d = reinterpret_cast<v4u*>(dst + r * dst_stride);
    for (int i = threadIdx.x; i < row_bytes / 16; i += 256) d[i] = s[i];
}

__global__ __launch_bounds__(256) void gather_f32_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                          float* __restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// ---- stale-cache guard: a sampled fingerprint of a weight tensor ------------------------------------------------------------------
// Everything the forward derives from a weight (the stacked q | k | v snapshot, the split-fp16 planes, the native layer structs)
// is cached by tensor identity + torch's in-place version counter, which a write through `param.data` or a raw pointer does not
// move.  So every cached weight also leaves a fingerprint of its BYTES in a per-encoder device table when the cache entry is made,
// and every edit call recomputes the fingerprints of the entries it is about to trust (one small launch, no host
// synchronisation): a mismatch raises a device flag that the call's one final read-back (edit_engine.check_info) sees.
// Fingerprint: up to 4 096 16-byte vectors spaced evenly over the tensor (64 KB read of a 9.4 MB weight), each mixed with its
// index, XOR-combined: any dense rewrite (a restore loop, a LoRA merge, an optimizer step, `.data.copy_`) changes it; a write
// that touches only a few rows can slip between the samples.  Table entry = {pointer, bytes, fingerprint, 0} (4 x int64).
__device__ __forceinline__ unsigned long long fp_mix(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    return x;
}

__device__ __forceinline__ unsigned long long fingerprint_of(const unsigned char* data, long long bytes, unsigned long long* lds) {
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    const long long nvec = bytes / 16, ns = nvec < 4096 ? nvec : 4096;
    unsigned long long h = 0;
    v2u64 x[16];                                                            // a thread's (up to) 16 samples: every load in flight at once
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const long long i = threadIdx.x + 256 * u;
        const long long v = ns == nvec ? i : i * nvec / ns;                 // i < 4 096, nvec < 2^40
        x[u] = i < ns ? *reinterpret_cast<const v2u64*>(data + 16 * v) : (v2u64){0ull, 0ull};
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const long long i = threadIdx.x + 256 * u;
        if (i < ns)
            h ^= fp_mix(x[u][0] + 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * i + 1)) ^ fp_mix(x[u][1] + 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * i + 2));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(h & 0xffffffffu), o, 64), hi = __shfl_xor((unsigned)(h >> 32), o, 64);
        h ^= ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = h;
    __syncthreads();
    return lds[0] ^ lds[1] ^ lds[2] ^ lds[3];
}

struct FpBatch { const unsigned char* data[32]; long long bytes[32]; long long slot[32]; };      // one workgroup per entry

__global__ __launch_bounds__(256) void fingerprint_store_kernel(FpBatch b, long long* table) {
    __shared__ unsigned long long lds[4];
    const unsigned char* data = b.data[blockIdx.x];
    const long long bytes = b.bytes[blockIdx.x];
    const unsigned long long h = fingerprint_of(data, bytes, lds);
    if (threadIdx.x == 0) {
        long long* e = table + 4 * b.slot[blockIdx.x];
        e[0] = (long long)reinterpret_cast<uintptr_t>(data); e[1] = bytes; e[2] = (long long)h; e[3] = 0;
    }
}

struct FpSkip { unsigned long long m[4]; };                   // bit b of word w: slot first_slot + 64 w + b is not to be looked at

__global__ __launch_bounds__(256) void fingerprint_check_kernel(const long long* table, long long first_slot, FpSkip skip, int* flag) {
    __shared__ unsigned long long lds[4];
    if ((skip.m[blockIdx.x >> 6] >> (blockIdx.x & 63)) & 1ull) return;      // (uniform: the whole workgroup leaves)
    const long long* e = table + 4 * (first_slot + blockIdx.x);
    const long long bytes = e[1];
    if (bytes <= 0) return;                                    // an empty slot
    const unsigned long long h = fingerprint_of(reinterpret_cast<const unsigned char*>((uintptr_t)e[0]), bytes, lds);
    if (threadIdx.x == 0 && h != (unsigned long long)e[2]) atomicOr(flag, 1);
}

struct ClipWs {
    float* qkv;            // [rows, 3h]
    float* mid;            // [rows, h]
    uint32_t* ctx_p;       // [rows, h] planes
    float* ctx_s;          // [rows]
    uint32_t* z_p;         // [rows, h]
    float* z_s;            // [rows]
    float* f_scale;        // [2, rows]
    uint32_t* f_p;         // [rows, d]
    uint32_t* xq_p;        // [rows, h]  gathered LN1 rows (query-row form)
    float* xq_s;           // [rows]
    float* hs_q;           // [rows, h]  gathered residual rows (query-row form)
};

inline int64_t clip_ws_bytes(int64_t rows, int64_t h, int64_t d) {
    const int64_t r = round_up(rows, 64);
    return 4 * (r * 3 * h + r * h + r * h + r + r * h + r + 2 * r + r * d + r * h + r + r * h) + 16 * 16;
}

inline bool clip_ws_carve(void* base, int64_t bytes, int64_t rows, int64_t h, int64_t d, ClipWs& w) {
    if (base == nullptr || !aligned16(base) || bytes < clip_ws_bytes(rows, h, d)) return false;
    const int64_t r = round_up(rows, 64);
    char* p = (char*)base;
    auto take = [&](int64_t n4) { char* q = p; p += round_up(4 * n4, 16); return q; };
    w.qkv = (float*)take(r * 3 * h);
    w.mid = (float*)take(r * h);
    w.ctx_p = (uint32_t*)take(r * h);
    w.ctx_s = (float*)take(r);
    w.z_p = (uint32_t*)take(r * h);
    w.z_s = (float*)take(r);
    w.f_scale = (float*)take(2 * r);
    w.f_p = (uint32_t*)take(r * d);
    w.xq_p = (uint32_t*)take(r * h);
    w.xq_s = (float*)take(r);
    w.hs_q = (float*)take(r * h);
    return true;
}

}  // namespace emcid

using namespace emcid;

extern "C" {

/* include/emcid_hip.h, "stale-cache guard".  Stores {data, bytes, fingerprint} of n weights' bytes in their slots of the table:
 * one launch per 32 entries, one workgroup per entry (host arrays). */
int emcid_fingerprint_store(int64_t n, const void* const* data, const int64_t* bytes, const int64_t* slots, void* table,
                            int64_t table_slots, void* stream) {
    EMCID_CHECK_ARG(n >= 0 && table && (n == 0 || (data && bytes && slots)));
    for (int64_t i0 = 0; i0 < n; i0 += 32) {
        FpBatch b{};
        const int cnt = (int)std::min<int64_t>(32, n - i0);
        for (int i = 0; i < cnt; ++i) {
            EMCID_CHECK_ARG(data[i0 + i] && bytes[i0 + i] > 0 && bytes[i0 + i] % 16 == 0 && aligned16(data[i0 + i]) &&
                            slots[i0 + i] >= 0 && slots[i0 + i] < table_slots);
            b.data[i] = (const unsigned char*)data[i0 + i]; b.bytes[i] = bytes[i0 + i]; b.slot[i] = slots[i0 + i];
        }
        hipLaunchKernelGGL(fingerprint_store_kernel, dim3((unsigned)cnt), dim3(256), 0, (hipStream_t)stream, b, (long long*)table);
    }
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

/* Recomputes the fingerprints of slots [first_slot, first_slot + n_slots) (n_slots <= 256; empty slots and the slots whose bit is
 * set in skip_mask — 4 host uint64, may be NULL — are skipped) and ORs 1 into *flag where the bytes no longer match.  Every slot
 * that is looked at must still point at live memory of that size: the caller compares tensor identity, address and version on the
 * host first and masks what it will not trust anyway (clip_forward.WeightGuard). */
int emcid_fingerprint_check(const void* table, int64_t table_slots, int64_t first_slot, int64_t n_slots, const uint64_t* skip_mask,
                            int* flag, void* stream) {
    EMCID_CHECK_ARG(table && flag && first_slot >= 0 && n_slots >= 0 && first_slot + n_slots <= table_slots && n_slots <= 256);
    if (n_slots == 0) return EMCID_OK;
    FpSkip skip{};
    if (skip_mask != nullptr)
        for (int w = 0; w < 4; ++w) skip.m[w] = skip_mask[w];
    hipLaunchKernelGGL(fingerprint_check_kernel, dim3((unsigned)n_slots), dim3(256), 0, (hipStream_t)stream, (const long long*)table,
                       (long long)first_slot, skip, flag);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int64_t emcid_clip_workspace_bytes(int64_t rows, int64_t h, int64_t d) { return clip_ws_bytes(rows, h, d); }

/* attention block + fc1 of one layer.  rows_sel == NULL: every node (n_sel = rows).  Outputs: mid [n_sel, h] fp32 (the residual
 * stream after the attention block), f_planes [n_sel, d] + f_scale [2, n_sel] (scale and inverse scale of the planes' rows),
 * f_f32 [n_sel, d] (optional fp32 twin of fc2's input: the keys). */
int emcid_clip_layer_head_sp16(const emcid_clip_layer_sp16* L, int64_t rows, int64_t h, int64_t d, int64_t heads, float attn_scale,
                               const int* anc, int64_t anc_ld, const int* depth, const int* rows_sel, int64_t n_sel,
                               const float* hs, const void* x_planes, const float* x_inv_scale, float* mid, void* f_planes,
                               float* f_scale, float* f_f32, void* workspace, int64_t workspace_bytes, void* stream) {
    EMCID_CHECK_ARG(L && rows > 0 && h > 0 && d > 0 && heads > 0 && h % heads == 0 && anc && depth && hs && x_planes && x_inv_scale);
    EMCID_CHECK_ARG(mid && f_planes && f_scale && n_sel > 0 && n_sel <= rows && (rows_sel != nullptr || n_sel == rows) && h % 32 == 0 && d % 32 == 0);
    ClipWs w;
    if (!clip_ws_carve(workspace, workspace_bytes, rows, h, d, w)) return fail(EMCID_ERR_BAD_ARG, __func__, "workspace too small or misaligned");
    hipStream_t st = (hipStream_t)stream;
    const int64_t D = h / heads;
    const float* q;
    int64_t ldq;
    const float* res;
    if (rows_sel == nullptr) {
        EMCID_TRY(emcid_linear_sp16_f32(x_planes, h, x_inv_scale, L->qkv_planes, h, L->qkv_inv_scale, L->qkv_bias, nullptr, 0, w.qkv,
                                        3 * h, nullptr, 0, nullptr, rows, 3 * h, h, 0, -1, stream));
        q = w.qkv, ldq = 3 * h, res = hs;
    } else {
        // k | v for every node (columns h .. 3h of the stacked projection), q for the selected rows only
        EMCID_TRY(emcid_linear_sp16_f32(x_planes, h, x_inv_scale, (const uint32_t*)L->qkv_planes + h * h, h, L->qkv_inv_scale + h,
                                        L->qkv_bias ? L->qkv_bias + h : nullptr, nullptr, 0, w.qkv + h, 3 * h, nullptr, 0, nullptr,
                                        rows, 2 * h, h, 0, -1, stream));
        {
            ScopedProf sp(KC_MISC, st);
            hipLaunchKernelGGL(gather_rows16_kernel, dim3((unsigned)n_sel), dim3(256), 0, st, (const unsigned char*)x_planes, 4 * h,
                               rows_sel, (unsigned char*)w.xq_p, 4 * h, (int)(4 * h));
            hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)((n_sel + 255) / 256)), dim3(256), 0, st, x_inv_scale, rows_sel,
                               w.xq_s, (int)n_sel);
            hipLaunchKernelGGL(gather_rows16_kernel, dim3((unsigned)n_sel), dim3(256), 0, st, (const unsigned char*)hs, 4 * h,
                               rows_sel, (unsigned char*)w.hs_q, 4 * h, (int)(4 * h));
            EMCID_CHECK_LAUNCH();
        }
        // q into columns 0 .. h of the first n_sel rows of the same buffer (leading dimension 3h: k | v of those rows stay intact)
        EMCID_TRY(emcid_linear_sp16_f32(w.xq_p, h, w.xq_s, L->qkv_planes, h, L->qkv_inv_scale, L->qkv_bias, nullptr, 0, w.mid, h,
                                        nullptr, 0, nullptr, n_sel, h, h, 0, -1, stream));
        q = w.mid, ldq = h, res = w.hs_q;
    }
    EMCID_TRY(emcid_tree_attention_sp16(q, ldq, w.qkv + h, w.qkv + 2 * h, 3 * h, anc, anc_ld, depth, rows_sel, n_sel, heads, D,
                                        attn_scale, w.ctx_p, h, w.ctx_s, stream));
    EMCID_TRY(emcid_linear_sp16_f32(w.ctx_p, h, w.ctx_s, L->out_planes, h, L->out_inv_scale, L->out_bias, res, h, mid, h, nullptr,
                                    0, nullptr, n_sel, h, h, 0, -1, stream));
    EMCID_TRY(emcid_add_layernorm_sp16(mid, h, nullptr, 0, L->ln2_gamma, L->ln2_beta, L->ln2_eps, n_sel, h, nullptr, nullptr, w.z_p,
                                       h, w.z_s, L->fc1_bound, f_scale, stream));
    // (the LayerNorm kernel writes the scales at [0, n_sel) and their inverses at [n_sel, 2 n_sel): the caller's [2, n_sel] layout)
    EMCID_TRY(emcid_linear_sp16_f32(w.z_p, h, w.z_s, L->fc1_planes, h, L->fc1_inv_scale, L->fc1_bias, nullptr, 0, f_f32, d,
                                    f_planes, d, f_scale, n_sel, d, h, L->act, -1, stream));
    return EMCID_OK;
}

/* fc2 + residual add of one layer, then (next_ln_gamma != NULL) LN1 of the next layer as planes.
 * hs_out = f W2^T + b2 + mid;  x_planes / x_inv_scale = LayerNorm_next(hs_out). */
int emcid_clip_layer_tail_sp16(const emcid_clip_layer_sp16* L, int64_t n, int64_t h, int64_t d, const void* f_planes,
                               const float* f_inv_scale, const float* mid, float* hs_out, const float* next_ln_gamma,
                               const float* next_ln_beta, float next_ln_eps, void* x_planes, float* x_inv_scale, void* stream) {
    EMCID_CHECK_ARG(L && n > 0 && h % 32 == 0 && d % 32 == 0 && f_planes && f_inv_scale && mid && hs_out);
    EMCID_TRY(emcid_linear_sp16_f32(f_planes, d, f_inv_scale, L->fc2_planes, d, L->fc2_inv_scale, L->fc2_bias, mid, h, hs_out, h,
                                    nullptr, 0, nullptr, n, h, d, 0, -1, stream));
    if (next_ln_gamma != nullptr) {
        EMCID_CHECK_ARG(next_ln_beta && x_planes && x_inv_scale);
        EMCID_TRY(emcid_add_layernorm_sp16(hs_out, h, nullptr, 0, next_ln_gamma, next_ln_beta, next_ln_eps, n, h, nullptr, nullptr,
                                           x_planes, h, x_inv_scale, nullptr, nullptr, stream));
    }
    return EMCID_OK;
}

/* Everything of an edited layer behind its attention block + fc1 (emcid_clip_layer_head_sp16 with the fp32 twin), from ONE call:
 * keys = per-request means of the twin at the prompts' lookup rows (emcid_gather_mean_f32), Zc = fc2(keys) with the layer's
 * CURRENT weight (k_planes / k_inv_scale given: scratch for the keys as a split matrix, and the product runs on the split-fp16
 * kernel against the layer's fc2 planes; NULL: on the exact-f32 kernel against W), the dual solver's apply-only form against the cached factors
 * (emcid_edit_dual_apply_stage1_f64 / _stage2_f64: W = W0 + float(U), dW), the NEW weight split into the layer's own fc2 planes
 * (in place: the struct stays valid), and — hs_out != NULL — fc2 + residual + the next layer's LN1 (emcid_clip_layer_tail_sp16).
 * The reference's layer loop body (emcid/emcid_main.py:981-1073) as one launch sequence; single rank, factors in HBM.
 * lookup [B]: row (in f_f32's row order) of every prompt's lookup token; seg [N + 1]: prompt offsets of the requests. */
int emcid_clip_edit_layer_tail_sp16(const emcid_clip_layer_sp16* L, int64_t n_rows, int64_t h, int64_t d, const float* f_f32,
                                    const void* f_planes, const float* f_inv_scale, const float* mid, const int64_t* lookup,
                                    const int64_t* seg, int64_t B, int64_t N, const float* zs_t, double edit_weight,
                                    int layers_left, double lam_ratio, const void* cov_factor_ws, int64_t n_layers,
                                    int64_t layer_index, int use_inverse, const float* W0, float* W, float* dW, float* K_out,
                                    float* Zc_out, void* k_planes, float* k_inv_scale, void* dual_ws, int64_t dual_ws_bytes,
                                    int* info_dev, void* linear_ws, int64_t linear_ws_bytes, float* hs_out,
                                    const float* next_ln_gamma, const float* next_ln_beta, float next_ln_eps, void* x_planes,
                                    float* x_inv_scale, void* stream) {
    EMCID_CHECK_ARG(L && n_rows > 0 && h % 32 == 0 && d % 32 == 0 && f_f32 && lookup && seg && B > 0 && N > 0 && zs_t);
    EMCID_CHECK_ARG(cov_factor_ws && W0 && W && K_out && Zc_out && dual_ws && info_dev && (hs_out == nullptr || (f_planes && f_inv_scale && mid)));
    EMCID_TRY(emcid_gather_mean_f32(f_f32, B, n_rows, d, 0, d, lookup, seg, N, K_out, d, stream));
    EMCID_CHECK_ARG((k_planes == nullptr) == (k_inv_scale == nullptr));
    if (k_planes != nullptr) {
        // Zc on the split-fp16 kernel against the layer's own fc2 planes (they are the planes of the CURRENT weight: the caller
        // splits a weight once per version, and this layer's has not been touched yet in this call)
        EMCID_TRY(emcid_split_rows_f16(K_out, d, N, d, k_planes, d, k_inv_scale, nullptr, stream));
        EMCID_TRY(emcid_linear_sp16_f32(k_planes, d, k_inv_scale, L->fc2_planes, d, L->fc2_inv_scale, L->fc2_bias, nullptr, 0, Zc_out, h,
                                        nullptr, 0, nullptr, N, h, d, 0, -1, stream));
    } else {
        EMCID_TRY(emcid_linear_ws_f32(K_out, d, W, d, L->fc2_bias, nullptr, 0, Zc_out, h, N, h, d, 0, -1, linear_ws, linear_ws_bytes, stream));
    }
    EMCID_TRY(emcid_edit_dual_apply_stage1_f64(K_out, Zc_out, zs_t, N, d, h, edit_weight, layers_left, lam_ratio, cov_factor_ws,
                                               n_layers, layer_index, 0, N, use_inverse, dual_ws, dual_ws_bytes, stream));
    EMCID_TRY(emcid_edit_dual_apply_stage2_f64(N, d, h, cov_factor_ws, n_layers, layer_index, use_inverse, 0, W0, W, dW, dual_ws,
                                               dual_ws_bytes, info_dev, stream));
    if (hs_out != nullptr) {
        EMCID_TRY(emcid_split_rows_f16(W, d, h, d, const_cast<void*>(L->fc2_planes), d, const_cast<float*>(L->fc2_inv_scale), nullptr,
                                       stream));
        EMCID_TRY(emcid_clip_layer_tail_sp16(L, n_rows, h, d, f_planes, f_inv_scale, mid, hs_out, next_ln_gamma, next_ln_beta,
                                             next_ln_eps, x_planes, x_inv_scale, stream));
    }
    return EMCID_OK;
}

/* A run of whole layers on every node: hs [rows, h] fp32 (in / out, in place), x_planes / x_inv_scale = LN1 of hs for layers[0]
 * (in) and LN1 of the result under next_ln_* (out; next_ln_gamma == NULL: left as they are). */
int emcid_clip_layers_sp16(const emcid_clip_layer_sp16* layers, int64_t n_layers, int64_t rows, int64_t h, int64_t d, int64_t heads,
                           float attn_scale, const int* anc, int64_t anc_ld, const int* depth, float* hs, void* x_planes,
                           float* x_inv_scale, const float* next_ln_gamma, const float* next_ln_beta, float next_ln_eps,
                           void* workspace, int64_t workspace_bytes, void* stream) {
    EMCID_CHECK_ARG(layers && n_layers > 0 && hs && x_planes && x_inv_scale);
    ClipWs w;
    if (!clip_ws_carve(workspace, workspace_bytes, rows, h, d, w)) return fail(EMCID_ERR_BAD_ARG, __func__, "workspace too small or misaligned");
    for (int64_t i = 0; i < n_layers; ++i) {
        const emcid_clip_layer_sp16* L = layers + i;
        EMCID_TRY(emcid_clip_layer_head_sp16(L, rows, h, d, heads, attn_scale, anc, anc_ld, depth, nullptr, rows, hs, x_planes,
                                             x_inv_scale, w.mid, w.f_p, w.f_scale, nullptr, workspace, workspace_bytes, stream));
        const bool last = i + 1 == n_layers;
        const float* g = last ? next_ln_gamma : layers[i + 1].ln1_gamma;
        const float* b = last ? next_ln_beta : layers[i + 1].ln1_beta;
        const float eps = last ? next_ln_eps : layers[i + 1].ln1_eps;
        EMCID_TRY(emcid_clip_layer_tail_sp16(L, rows, h, d, w.f_p, w.f_scale + rows, w.mid, hs, g, b, eps, x_planes, x_inv_scale, stream));
    }
    return EMCID_OK;
}

}  // extern "C"
