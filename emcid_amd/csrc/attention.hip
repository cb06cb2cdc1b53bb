// Fused softmax(Q K^T * scale + mask) V for CLIP text-encoder shapes on gfx950: fp32, head_dim <= 128,
// S <= 77 tokens, tens of thousands of (prompt, head) pairs.  Part of the K/Z assembly forward
// (reference: emcid/compute_z.py:2296-2308 runs HF CLIPTextModel.forward; its attention is the eager
// bmm + softmax + bmm of transformers' CLIPAttention).  The library SDPA kernel the framework would pick
// takes ~1.35 ms per layer at B=3000, S=9 (rocprof, profiles/r01_a_*); this path is HBM-bound instead:
// one wave owns one (prompt, head) pair, stages its Q/K/V rows (S x D floats each, coalesced 256-B rows)
// into LDS once, and never writes the S x S scores to memory.
//   scores : lane <-> (i, j) pair, dot product over D from LDS (rows padded to D+1 floats)
//   softmax: lane <-> query row, exact expf
//   P V    : lane <-> output column d, P[i][j] is an LDS broadcast, V[j][d] conflict-free
// `causal` skips j > i (CLIP text is causal); an optional mask (bool keep-mask or additive float,
// broadcastable over heads) covers right-padded prompts.
#include "common.h"
#include "sp16.h"

namespace emcid {

struct AttnArgs {
    const float* q; const float* k; const float* v;
    int64_t sb, sh, ss;            // element strides of q/k/v for (batch, head, token); d-stride is 1
    const void* mask; int mask_kind;  // 0 none, 1 uint8 keep-mask, 2 float additive
    int64_t mb, mi;                // mask strides for (batch, query row); key stride is 1; head-broadcast
    int causal; float scale;
    int B, H, S, D;
    float* out;                    // [B][S][H][D] contiguous
};

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attention_f32_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = a.S, D = a.D, DP = D + 1, SP = S + 1;
    const int per_wave = 3 * S * DP + S * SP;
    float* Qs = smem + wave * per_wave;
    float* Ks = Qs + S * DP;
    float* Vs = Ks + S * DP;
    float* P = Vs + S * DP;
    const int64_t total = (int64_t)a.B * a.H;
    int64_t pair = (int64_t)blockIdx.x * WAVES + wave;
    const bool live = pair < total;
    if (!live) pair = total - 1;   // keep every wave on the same barrier schedule; results are discarded
    const int b = (int)(pair / a.H), h = (int)(pair % a.H);
    const int64_t base = b * a.sb + h * a.sh;

    for (int r = 0; r < S; ++r)
        for (int d = lane; d < D; d += 64) {
            const int64_t g = base + r * a.ss + d;
            Qs[r * DP + d] = a.q[g];
            Ks[r * DP + d] = a.k[g];
            Vs[r * DP + d] = a.v[g];
        }
    __syncthreads();

    for (int e = lane; e < S * S; e += 64) {
        const int i = e / S, j = e % S;
        float s;
        if (a.causal && j > i) {
            s = -INFINITY;
        } else {
            float acc = 0.f;
            const float* qi = Qs + i * DP;
            const float* kj = Ks + j * DP;
#pragma unroll 8
            for (int d = 0; d < D; ++d) acc = fmaf(qi[d], kj[d], acc);
            s = acc * a.scale;
            if (a.mask_kind == 1) {
                if (!reinterpret_cast<const unsigned char*>(a.mask)[b * a.mb + i * a.mi + j]) s = -INFINITY;
            } else if (a.mask_kind == 2) {
                s += reinterpret_cast<const float*>(a.mask)[b * a.mb + i * a.mi + j];
            }
        }
        P[i * SP + j] = s;
    }
    __syncthreads();

    for (int i = lane; i < S; i += 64) {
        float* row = P + i * SP;
        float m = -INFINITY;
        for (int j = 0; j < S; ++j) m = fmaxf(m, row[j]);
        float sum = 0.f;
        for (int j = 0; j < S; ++j) {
            const float p = expf(row[j] - m);
            row[j] = p;
            sum += p;
        }
        const float inv = 1.f / sum;
        for (int j = 0; j < S; ++j) row[j] *= inv;
    }
    __syncthreads();

    if (live) {
        for (int d = lane; d < D; d += 64) {
            for (int i = 0; i < S; ++i) {
                const int jend = a.causal ? i + 1 : S;
                float acc = 0.f;
                for (int j = 0; j < jend; ++j) acc = fmaf(P[i * SP + j], Vs[j * DP + d], acc);
                a.out[(((int64_t)b * S + i) * a.H + h) * D + d] = acc;
            }
        }
    }
}

// ---- attention over a token trie ------------------------------------------------------------------------------
// The prompts of a mass edit share causal prefixes ("painting by ...", BOS).  A causal encoder's state at a token
// depends only on the tokens before it, so every distinct prefix is computed ONCE: tokens are the nodes of a
// trie and a node attends to its ancestor chain.  q/k/v are [U, H*D] rows (one per node); anc[u][0..depth[u]]
// lists the chain root..u.  One workgroup per query node, one wave per head (round robin); a wave takes 4 keys
// per step: lane = (key slot, 16-byte chunk of the head), so a key row is one coalesced 256-B read.
struct TreeAttnArgs {
    const float* q; int64_t ldq; const float* k; const float* v; int64_t ld;
    const int* anc; int64_t anc_ld; const int* depth; const int* rows; int n_rows;
    int H, D; float scale; float* out; int64_t ldo;
    uint32_t* P; int64_t ldp; float* inv_scale;        // the output as a split-fp16 matrix instead of `out` (short-chain kernel)
};

__global__ __launch_bounds__(256) void tree_attention_f32_kernel(TreeAttnArgs a) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    __shared__ int pth[128];
    __shared__ float sc[4][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = lane >> 4, dl = lane & 15;
    const int qi = blockIdx.x;
    const int u = a.rows ? a.rows[qi] : qi;
    const int nk = a.depth[u] + 1;
    if (tid < nk) pth[tid] = a.anc[(int64_t)u * a.anc_ld + tid];
    __syncthreads();
    const bool dok = 4 * dl < a.D;
    const int rounds = (a.H + 3) / 4;
    for (int hh = 0; hh < rounds; ++hh) {
        const int h = hh * 4 + wave;
        const bool live = h < a.H;
        const int64_t col = (int64_t)(live ? h : 0) * a.D + 4 * dl;
        v4f q4 = {0.f, 0.f, 0.f, 0.f};
        if (dok) q4 = *reinterpret_cast<const v4f*>(a.q + (int64_t)qi * a.ldq + col);   // q rows follow the QUERY order
        // sixteen keys per step: their four row reads go out together (one dependent read per 16 keys instead of per 4 — the
        // kernel is latency-bound on long chains: Stage 0's captions, 1.19 ms per launch at 88 000 rows before)
        for (int j0 = 0; j0 < nk; j0 += 16) {
            v4f k4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = j0 + 4 * t + slot;
                k4[t] = (v4f){0.f, 0.f, 0.f, 0.f};
                if (j < nk && dok) k4[t] = *reinterpret_cast<const v4f*>(a.k + (int64_t)pth[j] * a.ld + col);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = j0 + 4 * t + slot;
                float part = q4[0] * k4[t][0] + q4[1] * k4[t][1] + q4[2] * k4[t][2] + q4[3] * k4[t][3];
                part += __shfl_xor(part, 8);
                part += __shfl_xor(part, 4);
                part += __shfl_xor(part, 2);
                part += __shfl_xor(part, 1);
                if (dl == 0 && j < nk) sc[wave][j] = part * a.scale;
            }
        }
        __syncthreads();
        float m = -INFINITY;
        for (int j = lane; j < nk; j += 64) m = fmaxf(m, sc[wave][j]);
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float l = 0.f;
        for (int j = lane; j < nk; j += 64) {
            const float p = expf(sc[wave][j] - m);
            sc[wave][j] = p;
            l += p;
        }
        for (int o = 32; o >= 1; o >>= 1) l += __shfl_xor(l, o);
        __syncthreads();
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < nk; j0 += 16) {
            v4f v4[4];
            float p[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = j0 + 4 * t + slot;
                v4[t] = (v4f){0.f, 0.f, 0.f, 0.f};
                p[t] = 0.f;
                if (j < nk && dok) {
                    p[t] = sc[wave][j];
                    v4[t] = *reinterpret_cast<const v4f*>(a.v + (int64_t)pth[j] * a.ld + col);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {          // (key order as before: j0 + slot, j0 + 4 + slot, ...)
                acc[0] += p[t] * v4[t][0]; acc[1] += p[t] * v4[t][1]; acc[2] += p[t] * v4[t][2]; acc[3] += p[t] * v4[t][3];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e] += __shfl_xor(acc[e], 16);
            acc[e] += __shfl_xor(acc[e], 32);
        }
        if (live && slot == 0 && dok) {
            const float inv = 1.f / l;
            v4f o4 = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
            *reinterpret_cast<v4f*>(a.out + (int64_t)qi * a.ldo + col) = o4;
        }
        __syncthreads();
    }
}

// y = x * sigmoid(1.702 x)  (CLIP's quick_gelu; HF runs it as mul -> sigmoid -> mul, three passes over HBM)
__global__ __launch_bounds__(256) void quick_gelu_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4,
                                                              int64_t n) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n4; i += stride) {
        const v4f a = reinterpret_cast<const v4f*>(x)[i];
        v4f r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = a[e] * (1.0f / (1.0f + expf(-1.702f * a[e])));
        reinterpret_cast<v4f*>(y)[i] = r;
    }
    if (blockIdx.x == 0)
        for (int64_t j = n4 * 4 + threadIdx.x; j < n; j += 256) y[j] = x[j] * (1.0f / (1.0f + expf(-1.702f * x[j])));
}

}  // namespace emcid

using namespace emcid;

// Chains of at most 8 nodes (the mass-edit prompts: BOS + a few template words + a few name tokens): everything a workgroup needs
// is fetched in TWO dependent rounds — its chain, then q and every k / v row of all its heads — and the softmax over <= 8 keys
// lives in registers (two keys per 16-lane group, combined with shuffles): no LDS, no barrier.  The general kernel above pays two
// dependent global reads and three barriers per round of four heads (38 us for 6 400 nodes x 12 heads; this one: see DESIGN.md).
template <int ROUNDS, int KPS, bool SP>
__global__ __launch_bounds__(256) void tree_attention_short_kernel(TreeAttnArgs a) {
    // KPS keys per 16-lane group: chains of up to 4 * KPS nodes (KPS = 2: 8, KPS = 4: 16)
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = lane >> 4, dl = lane & 15;
    const int qi = blockIdx.x;
    const int u = a.rows ? a.rows[qi] : qi;
    // the chain entries and the chain's length are fetched side by side (entry j is read whether or not j <= depth: any int
    // may stand there, it is replaced by the node itself below), so k / v rows are two dependent reads away, not three
    int av[KPS];
#pragma unroll
    for (int c = 0; c < KPS; ++c) av[c] = a.anc[(int64_t)u * a.anc_ld + min(slot + 4 * c, (int)a.anc_ld - 1)];
    const int nk = a.depth[u] + 1;
    bool has[KPS];
    int64_t pk[KPS];
#pragma unroll
    for (int c = 0; c < KPS; ++c) {
        has[c] = slot + 4 * c < nk;
        pk[c] = has[c] ? av[c] : u;
    }
    const bool dok = 4 * dl < a.D;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    v4f q4[ROUNDS], kk[ROUNDS][KPS], vv[ROUNDS][KPS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int h = r * 4 + wave;
        const int64_t col = (int64_t)(h < a.H ? h : 0) * a.D + (dok ? 4 * dl : 0);
        q4[r] = *reinterpret_cast<const v4f*>(a.q + (int64_t)qi * a.ldq + col);
#pragma unroll
        for (int c = 0; c < KPS; ++c) {
            kk[r][c] = *reinterpret_cast<const v4f*>(a.k + pk[c] * a.ld + col);
            vv[r][c] = *reinterpret_cast<const v4f*>(a.v + pk[c] * a.ld + col);
        }
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int h = r * 4 + wave;
        float sc[KPS];
#pragma unroll
        for (int c = 0; c < KPS; ++c)
            sc[c] = dok ? q4[r][0] * kk[r][c][0] + q4[r][1] * kk[r][c][1] + q4[r][2] * kk[r][c][2] + q4[r][3] * kk[r][c][3] : 0.f;
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1)
#pragma unroll
            for (int c = 0; c < KPS; ++c) sc[c] += __shfl_xor(sc[c], o);
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < KPS; ++c) {
            sc[c] = has[c] ? sc[c] * a.scale : -INFINITY;
            m = fmaxf(m, sc[c]);
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
        v4f acc = zero;
#pragma unroll
        for (int c = 0; c < KPS; ++c) {
            const float e = has[c] ? expf(sc[c] - m) : 0.f;
            l += e;
            if (dok) acc += e * vv[r][c];
        }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e] += __shfl_xor(acc[e], 16);
            acc[e] += __shfl_xor(acc[e], 32);
        }
        if constexpr (SP) {
            q4[r] = acc * (1.f / l);             // kept until the row's largest magnitude is known (q4[r] is dead by now)
        } else if (h < a.H && slot == 0 && dok) {
            const float inv = 1.f / l;
            *reinterpret_cast<v4f*>(a.out + (int64_t)qi * a.ldo + (int64_t)h * a.D + 4 * dl) = acc * inv;
        }
    }
    if constexpr (SP) {
        // the row (all heads of this node) as planes under its own scale: largest magnitude over the workgroup's four waves first
        __shared__ float wmax[4];
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            if (r * 4 + wave < a.H && dok)
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(q4[r][0]), fabsf(q4[r][1]))), fmaxf(fabsf(q4[r][2]), fabsf(q4[r][3])));
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        if (lane == 0) wmax[wave] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        float s, inv;
        sp_scale_of(amax, s, inv);
        if (tid == 0) a.inv_scale[qi] = inv;
        uint32_t* prow = a.P + (int64_t)qi * a.ldp;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int h = r * 4 + wave;
            if (h < a.H && slot == 0 && dok) sp_store4(prow, h * a.D + 4 * dl, q4[r][0], q4[r][1], q4[r][2], q4[r][3], s);
        }
    }
}

template <int KPS, bool SP>
static void launch_tree_attention_short(const TreeAttnArgs& a, int rounds, hipStream_t st) {
    const dim3 g((unsigned)a.n_rows), b(256);
    switch (rounds) {
        case 1: hipLaunchKernelGGL((tree_attention_short_kernel<1, KPS, SP>), g, b, 0, st, a); break;
        case 2: hipLaunchKernelGGL((tree_attention_short_kernel<2, KPS, SP>), g, b, 0, st, a); break;
        case 3: hipLaunchKernelGGL((tree_attention_short_kernel<3, KPS, SP>), g, b, 0, st, a); break;
        case 4: hipLaunchKernelGGL((tree_attention_short_kernel<4, KPS, SP>), g, b, 0, st, a); break;
        default: hipLaunchKernelGGL((tree_attention_short_kernel<5, KPS, SP>), g, b, 0, st, a); break;
    }
}

/* 1 when emcid_tree_attention_sp16 serves chains of up to anc_ld nodes with H heads (the short-chain kernel), else 0 */
extern "C" int emcid_tree_attention_sp16_supported(int64_t anc_ld, int64_t H, int64_t D) {
    constexpr int short_ok = 1;
    return short_ok && anc_ld >= 1 && anc_ld <= 16 && (H + 3) / 4 <= 5 && D <= 64 && D % 8 == 0 ? 1 : 0;
}

/* emcid_tree_attention_f32 with the result written as a split-fp16 matrix (planes [n_rows, H * D] + inv_scale [n_rows]) for the
 * out-projection that consumes it.  Chains of at most 16 nodes (emcid_tree_attention_sp16_supported). */
extern "C" int emcid_tree_attention_sp16(const float* q, int64_t ldq, const float* k, const float* v, int64_t ld, const int* anc,
                                         int64_t anc_ld, const int* depth, const int* rows, int64_t n_rows, int64_t H,
                                         int64_t D, float scale, void* planes, int64_t ldp, float* inv_scale, void* stream) {
    EMCID_CHECK_ARG(q && k && v && anc && depth && planes && inv_scale && n_rows > 0 && H > 0 && D > 0);
    EMCID_CHECK_ARG(emcid_tree_attention_sp16_supported(anc_ld, H, D) && ld % 4 == 0 && ldq % 4 == 0 && ldp % 4 == 0 && ldp >= H * D);
    EMCID_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(planes) && n_rows < (1LL << 31));
    TreeAttnArgs a{q, ldq, k, v, ld, anc, anc_ld, depth, rows, (int)n_rows, (int)H, (int)D, scale, nullptr, 0,
                   (uint32_t*)planes, ldp, inv_scale};
    ScopedProf sp(KC_MISC, (hipStream_t)stream);
    const int rounds = (int)((H + 3) / 4);
    if (anc_ld <= 8) launch_tree_attention_short<2, true>(a, rounds, (hipStream_t)stream);
    else launch_tree_attention_short<4, true>(a, rounds, (hipStream_t)stream);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

extern "C" int emcid_tree_attention_f32(const float* q, int64_t ldq, const float* k, const float* v, int64_t ld, const int* anc,
                                        int64_t anc_ld, const int* depth, const int* rows, int64_t n_rows, int64_t H,
                                        int64_t D, float scale, float* out, int64_t ldo, void* stream) {
    EMCID_CHECK_ARG(q && k && v && anc && depth && out && n_rows > 0 && H > 0 && D > 0);
    EMCID_CHECK_ARG(D <= 64 && D % 4 == 0 && ld % 4 == 0 && ldq % 4 == 0 && ldo % 4 == 0 && anc_ld >= 1 && anc_ld <= 128);
    EMCID_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out) && n_rows < (1LL << 31));
    TreeAttnArgs a{q, ldq, k, v, ld, anc, anc_ld, depth, rows, (int)n_rows, (int)H, (int)D, scale, out, ldo, nullptr, 0, nullptr};
    ScopedProf sp(KC_MISC, (hipStream_t)stream);
    constexpr int short_ok = 1;
    const int rounds = (int)((H + 3) / 4);
    if (short_ok && anc_ld <= 16 && rounds <= 5) {      // every chain has at most anc_ld nodes
        if (anc_ld <= 8) launch_tree_attention_short<2, false>(a, rounds, (hipStream_t)stream);
        else launch_tree_attention_short<4, false>(a, rounds, (hipStream_t)stream);
    } else {
        hipLaunchKernelGGL(tree_attention_f32_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, a);
    }
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

// y = a + b ; z = LayerNorm(y) * gamma + beta   (one row per workgroup, the row lives in registers between the passes).
// The residual add and the LayerNorm that follows it in every transformer block, as one pass over HBM instead of two
// kernels (add: 2 reads + 1 write; layer norm: 1 read + 1 write).  Two-pass mean / variance like torch's.
constexpr int LN_MAX_V4 = 8;    // float4 chunks per thread: rows up to 8 * 256 * 4 = 8192 columns
__global__ __launch_bounds__(256) void add_layernorm_f32_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b,
                                                                 int64_t ldb, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, int cols,
                                                                 float* __restrict__ y, float* __restrict__ z,
                                                                 const int64_t* __restrict__ ia, const int* __restrict__ ib) {
    // ia / ib (usually null): row r reads a[ia[r]] and b[ib[r]] — the embedding lookups of the first layer's input
    const int64_t row = blockIdx.x;
    const int nv = cols / 4;
    const float4* pa = reinterpret_cast<const float4*>(a + (ia ? ia[row] : row) * lda);
    const float4* pb = b ? reinterpret_cast<const float4*>(b + (ib ? (int64_t)ib[row] : row) * ldb) : nullptr;   // null: z = LN(a)
    float4 v[LN_MAX_V4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < nv) {
            const float4 x = pa[c], w = pb ? pb[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            v[i] = make_float4(x.x + w.x, x.y + w.y, x.z + w.z, x.w + w.w);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    __shared__ float red[8];
    auto block_sum = [&](float t) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    const float mean = block_sum(sum) / (float)cols;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < nv) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float rstd = rsqrtf(block_sum(sq) / (float)cols + eps);
    float4* py = reinterpret_cast<float4*>(y + row * (int64_t)cols);
    float4* pz = reinterpret_cast<float4*>(z + row * (int64_t)cols);
    const float4* pg = reinterpret_cast<const float4*>(gamma);
    const float4* pe = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < nv) {
            const float4 g = pg[c], e = pe[c];
            if (y) py[c] = v[i];
            pz[c] = make_float4((v[i].x - mean) * rstd * g.x + e.x, (v[i].y - mean) * rstd * g.y + e.y,
                                (v[i].z - mean) * rstd * g.z + e.z, (v[i].w - mean) * rstd * g.w + e.w);
        }
    }
}

// Rows of up to 2048 columns: one WAVE per row (four rows per workgroup), both reductions by shuffles — no LDS, no barrier; a
// workgroup's four rows are independent, so nothing in it waits for anything but its own loads (15 -> 11 us at 6400 x 768).
// The split-fp16 form of the output (include/emcid_hip.h, "split fp16"): z as planes under the row's own scale — the consumer is a
// projection (q | k | v, fc1) — and, when `bound` = {largest row norm of that projection's weight, largest |bias|} is given, the
// scale 2^e (out_scale[row]; its inverse at out_scale[rows + row]) under which the PROJECTION may write its own output as planes:
// |act(z . w_j + b_j)| <= |z| max_j |w_j| + max_j |b_j|.
struct LnPlanes {
    uint32_t* P; int64_t ldp; float* inv_scale; const float* bound; float* out_scale;
};

__global__ __launch_bounds__(256) void add_layernorm_wave_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b,
                                                                  int64_t ldb, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps, int cols, int rows,
                                                                  float* __restrict__ y, float* __restrict__ z,
                                                                  const int64_t* __restrict__ ia, const int* __restrict__ ib,
                                                                  LnPlanes sp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = cols / 4;
    const float4* pa = reinterpret_cast<const float4*>(a + (ia ? ia[row] : row) * lda);
    const float4* pb = b ? reinterpret_cast<const float4*>(b + (ib ? (int64_t)ib[row] : row) * ldb) : nullptr;   // null: z = LN(a)
    float4 v[LN_MAX_V4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const float4 x = pa[c], w = pb ? pb[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            v[i] = make_float4(x.x + w.x, x.y + w.y, x.z + w.z, x.w + w.w);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)cols;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq / (float)cols + eps);
    float4* py = reinterpret_cast<float4*>(y + row * (int64_t)cols);
    float4* pz = reinterpret_cast<float4*>(z + row * (int64_t)cols);
    const float4* pg = reinterpret_cast<const float4*>(gamma);
    const float4* pe = reinterpret_cast<const float4*>(beta);
    float amax = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const float4 g = pg[c], e = pe[c];
            if (y) py[c] = v[i];
            v[i] = make_float4((v[i].x - mean) * rstd * g.x + e.x, (v[i].y - mean) * rstd * g.y + e.y,
                               (v[i].z - mean) * rstd * g.z + e.z, (v[i].w - mean) * rstd * g.w + e.w);
            if (z) pz[c] = v[i];
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
            ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
    }
    if (sp.P == nullptr) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        ss += __shfl_xor(ss, o, 64);
    }
    float s, inv;
    sp_scale_of(amax, s, inv);
    if (lane == 0) {
        sp.inv_scale[row] = inv;
        if (sp.out_scale != nullptr) {
            float s2, inv2;
            sp_scale_of(sqrtf(ss) * 1.0001f * sp.bound[0] + sp.bound[1], s2, inv2);
            sp.out_scale[row] = s2;
            sp.out_scale[rows + row] = inv2;
        }
    }
    uint32_t* prow = sp.P + row * sp.ldp;
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) sp_store4(prow, 4 * c, v[i].x, v[i].y, v[i].z, v[i].w, s);
    }
}

static void launch_add_layernorm(const float* a, int64_t lda, const float* b, int64_t ldb, const float* gamma, const float* beta,
                                 float eps, int64_t rows, int64_t cols, float* y, float* z, const int64_t* ia, const int* ib,
                                 hipStream_t st, LnPlanes sp = LnPlanes{nullptr, 0, nullptr, nullptr, nullptr}) {
    constexpr int wave_rows = 1;
    if ((wave_rows || sp.P != nullptr) && cols <= LN_MAX_V4 * 64 * 4)
        hipLaunchKernelGGL(add_layernorm_wave_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a, lda, b, ldb, gamma, beta,
                           eps, (int)cols, (int)rows, y, z, ia, ib, sp);
    else
        hipLaunchKernelGGL(add_layernorm_f32_kernel, dim3((unsigned)rows), dim3(256), 0, st, a, lda, b, ldb, gamma, beta, eps,
                           (int)cols, y, z, ia, ib);
}

extern "C" int emcid_add_layernorm_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* gamma,
                                       const float* beta, float eps, int64_t rows, int64_t cols, float* y, float* z,
                                       void* stream) {
    EMCID_CHECK_ARG(a && gamma && beta && z && rows > 0 && cols > 0 && rows < (1LL << 31));
    EMCID_CHECK_ARG(cols % 4 == 0 && cols <= LN_MAX_V4 * 256 * 4 && lda % 4 == 0 && (b == nullptr || ldb % 4 == 0));
    EMCID_CHECK_ARG(aligned16(a) && aligned16(b) && aligned16(gamma) && aligned16(beta) && aligned16(y) && aligned16(z));
    ScopedProf sp(KC_MISC, (hipStream_t)stream);
    launch_add_layernorm(a, lda, b, ldb, gamma, beta, eps, rows, cols, y, z, nullptr, nullptr, (hipStream_t)stream);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

/* emcid_add_layernorm_f32 with z written as a split-fp16 matrix (planes + inv_scale) for the projection that consumes it; z
 * itself optional.  bound / out_scale (optional, together): see LnPlanes.  cols % 8 == 0, <= 2048. */
extern "C" int emcid_add_layernorm_sp16(const float* a, int64_t lda, const float* b, int64_t ldb, const float* gamma,
                                        const float* beta, float eps, int64_t rows, int64_t cols, float* y, float* z, void* planes,
                                        int64_t ldp, float* inv_scale, const float* bound, float* out_scale, void* stream) {
    EMCID_CHECK_ARG(a && gamma && beta && planes && inv_scale && rows > 0 && cols > 0 && rows < (1LL << 31));
    EMCID_CHECK_ARG(cols % 8 == 0 && cols <= LN_MAX_V4 * 64 * 4 && lda % 4 == 0 && (b == nullptr || ldb % 4 == 0));
    EMCID_CHECK_ARG(ldp >= cols && ldp % 4 == 0 && aligned16(planes) && ((bound == nullptr) == (out_scale == nullptr)));
    EMCID_CHECK_ARG(aligned16(a) && aligned16(b) && aligned16(gamma) && aligned16(beta) && aligned16(y) && aligned16(z));
    ScopedProf sp(KC_MISC, (hipStream_t)stream);
    launch_add_layernorm(a, lda, b, ldb, gamma, beta, eps, rows, cols, y, z, nullptr, nullptr, (hipStream_t)stream,
                         LnPlanes{(uint32_t*)planes, ldp, inv_scale, bound, out_scale});
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

/* emcid_embed_layernorm_f32 with the LayerNorm's output as a split-fp16 matrix (z optional). */
extern "C" int emcid_embed_layernorm_sp16(const float* tok, int64_t ld_tok, int64_t n_tok, const float* pos, int64_t ld_pos,
                                          int64_t n_pos, const int64_t* token, const int* position, const float* gamma,
                                          const float* beta, float eps, int64_t rows, int64_t cols, float* y, float* z, void* planes,
                                          int64_t ldp, float* inv_scale, void* stream) {
    EMCID_CHECK_ARG(tok && pos && token && position && gamma && beta && y && planes && inv_scale && rows > 0 && cols > 0);
    EMCID_CHECK_ARG(rows < (1LL << 31) && n_tok > 0 && n_pos > 0 && cols % 8 == 0 && cols <= LN_MAX_V4 * 64 * 4);
    EMCID_CHECK_ARG(ld_tok % 4 == 0 && ld_pos % 4 == 0 && ldp >= cols && ldp % 4 == 0 && aligned16(planes));
    EMCID_CHECK_ARG(aligned16(tok) && aligned16(pos) && aligned16(gamma) && aligned16(beta) && aligned16(y) && aligned16(z));
    ScopedProf sp(KC_MISC, (hipStream_t)stream);
    launch_add_layernorm(tok, ld_tok, pos, ld_pos, gamma, beta, eps, rows, cols, y, z, token, position, (hipStream_t)stream,
                         LnPlanes{(uint32_t*)planes, ldp, inv_scale, nullptr, nullptr});
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

/* y[r] = tok[token[r]] + pos[position[r]] ; z = LayerNorm(y): the embedding stage of the text encoder (HF CLIPTextEmbeddings:
 * token_embedding(ids) + position_embedding(position_ids)) and the first layer's LN1 in one launch instead of five (two gathers,
 * a cast, an add, a LayerNorm) — at the head of an edit call the GPU waits for every one of those launches. */
extern "C" int emcid_embed_layernorm_f32(const float* tok, int64_t ld_tok, int64_t n_tok, const float* pos, int64_t ld_pos,
                                         int64_t n_pos, const int64_t* token, const int* position, const float* gamma,
                                         const float* beta, float eps, int64_t rows, int64_t cols, float* y, float* z, void* stream) {
    EMCID_CHECK_ARG(tok && pos && token && position && gamma && beta && y && z && rows > 0 && cols > 0 && rows < (1LL << 31));
    EMCID_CHECK_ARG(n_tok > 0 && n_pos > 0 && cols % 4 == 0 && cols <= LN_MAX_V4 * 256 * 4 && ld_tok % 4 == 0 && ld_pos % 4 == 0);
    EMCID_CHECK_ARG(aligned16(tok) && aligned16(pos) && aligned16(gamma) && aligned16(beta) && aligned16(y) && aligned16(z));
    ScopedProf sp(KC_MISC, (hipStream_t)stream);
    launch_add_layernorm(tok, ld_tok, pos, ld_pos, gamma, beta, eps, rows, cols, y, z, token, position, (hipStream_t)stream);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

extern "C" int emcid_quick_gelu_f32(const float* x, float* y, int64_t n, void* stream) {
    EMCID_CHECK_ARG(x && y && n > 0 && aligned16(x) && aligned16(y));
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    ScopedProf sp(KC_MISC, (hipStream_t)stream);
    hipLaunchKernelGGL(quick_gelu_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n4, n);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

extern "C" int emcid_attention_f32(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t ss,
                                   const void* mask, int mask_kind, int64_t mb, int64_t mi, int causal, float scale,
                                   int64_t B, int64_t H, int64_t S, int64_t D, float* out, void* stream) {
    EMCID_CHECK_ARG(q && k && v && out && B > 0 && H > 0 && S > 0 && D > 0);
    EMCID_CHECK_ARG(S <= 128 && D <= 128 && B * H < (1LL << 31));
    EMCID_CHECK_ARG(mask_kind >= 0 && mask_kind <= 2 && ((mask_kind == 0) == (mask == nullptr)));
    AttnArgs a{q, k, v, sb, sh, ss, mask, mask_kind, mb, mi, causal, scale, (int)B, (int)H, (int)S, (int)D, out};
    const size_t per_wave = (size_t)(3 * S * (D + 1) + S * (S + 1)) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    ScopedProf sp(KC_MISC, st);
    const int64_t pairs = B * H;
    if (per_wave * 4 <= 64 * 1024) {
        hipLaunchKernelGGL(attention_f32_kernel<4>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), per_wave * 4, st, a);
    } else {
        if (per_wave > 160 * 1024) return fail(EMCID_ERR_BAD_ARG, __func__, "S x D too large for one wave's LDS image");
        static bool attr_set[64] = {};      // a function attribute is per device
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f32_kernel<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return fail(EMCID_ERR_HIP, __func__, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
        hipLaunchKernelGGL(attention_f32_kernel<1>, dim3((unsigned)pairs), dim3(64), per_wave, st, a);
    }
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}
