// Stage 0 of EMCID on gfx950: running second moment  G += X^T X  of fc2-input features, fp32.
// Replaces `self.mom2 += a.t().mm(a)` (reference: util/runningstats.py:493; driver emcid/layer_stats.py:208-219).
//
// SYRK on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32): only tiles on or below the diagonal are computed.
// X is [t][d] row-major, so both MFMA operands are d-contiguous slices of the same token rows: the LDS
// image of a tile is a straight copy [BK tokens][128 features]; a fragment read is 32 consecutive floats
// per token row (ds_read_b32, lanes 0-31 token k, lanes 32-63 token k+1) and is conflict-free.
// The token dimension is split over blockIdx.z (ksplit); partial tiles are then added with
// global_atomic_add_f32 (G is an accumulator anyway), otherwise with a plain read-modify-write.
// Global loads are kept branch-free (see ALIGNED / FAST below): with per-element tails in the loop the compiler
// serialised a stage's four loads behind `s_waitcnt vmcnt(0)` and the kernel sat at 50 % of the f32 MFMA peak; with
// that fixed and the workgroups numbered so that the 8 XCDs get equal work (see the kernel's grid comment) it runs at
// 73 % (d = 3072) to 76 % (d = 5120).
// Also here: the small byte-moving kernels of the K/Z assembly (gather + per-request mean) and the
// lower->upper mirror used when the moment is read.
#include "common.h"

namespace emcid {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int GB = 128;   // output tile edge
constexpr int GBK = 16;   // tokens per stage

// MI = 32-row MFMA blocks per wave along m: the workgroup tile is (64*MI) x 128 (2 x 2 waves, each (32*MI) x 64).
// MI = 2 (128 x 128, 4 workgroups per CU) is what is launched; MI = 4 (256 x 128: a third less data per flop through
// L2/LDS, 2 workgroups per CU) measured 2-8 % slower at d = 3072 / 5120 — the kernel is not bound by that traffic.
// ALIGNED (d % 4 == 0): a float4 of a row is entirely inside or entirely outside the matrix, so the global loads are
// unconditional (clamped address + select) and the compiler can keep all of a stage's loads in flight; the general form
// has per-element tails, whose branches serialise the loads behind `s_waitcnt vmcnt(0)`.
#ifndef GRAM_PIN_PREFETCH
#define GRAM_PIN_PREFETCH 1
#endif
template <int MI, bool ALIGNED, bool FAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void gram_f32_kernel(const float* __restrict__ X, int t, int d, int64_t ldx,
                                                        float* __restrict__ G, int64_t ldg, int kchunk, int use_atomic,
                                                        int ksplit) {
    constexpr int BMT = 64 * MI;                    // tile rows
    constexpr int STAGE = GBK * (BMT + GB);         // floats per stage: A image [GBK][BMT], then B image [GBK][GB]
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    // 1-D grid over (lower tile, token slab), slab fastest.  Workgroups go to the 8 XCDs round-robin by linear id: with a
    // (bn, bm, z) grid of 24 x 24 tiles the XCD was bn % 8 and XCD 0 owned 48 of the 300 lower tiles, XCD 7 only 27 — the
    // launch lasted as long as XCD 0 (78 % balance).  Here consecutive ids are the slabs of ONE tile, so with the slab
    // count a multiple of 8 every XCD gets the same number of (tile, slab) pairs and only ever reads its own eighth of
    // the token rows (each XCD's L2 then holds 1/8 of X instead of all of it).
    static_assert(BMT == GB, "the triangular tile numbering assumes square tiles");
    const int z = blockIdx.x % ksplit;
    const int L = blockIdx.x / ksplit;
    int bm = (int)((__builtin_sqrt(8.0 * L + 1.0) - 1.0) * 0.5);
    while ((bm + 1) * (bm + 2) / 2 <= L) ++bm;      // guard the rounding of the square root
    while (bm * (bm + 1) / 2 > L) --bm;
    const int bn = L - bm * (bm + 1) / 2;
    const int m0 = bm * BMT, n0 = bn * GB;
    const int k_begin = z * kchunk;
    const int k_end = min(t, k_begin + kchunk);
    if (k_begin >= k_end) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (32 * MI), wn0 = (wave & 1) * 64;
    const int l31 = lane & 31, l5 = lane >> 5;

    v16f acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per stage each thread moves MI float4 of the A image and 2 of the B image
    v4f ra[MI], rb[2];
    bool oka[MI], okb[2];                           // ALIGNED: the zeroing select is applied when the registers go to LDS,
    auto fetch = [&](int kr, int col0, bool& ok) {  // after the MFMAs of the stage, so that nothing waits on the loads before
        const v4f zero = {0.f, 0.f, 0.f, 0.f};
        if constexpr (ALIGNED) {
            ok = kr < k_end && col0 < d;
            const float* src = ok ? X + (int64_t)kr * ldx + col0 : X;
            return *reinterpret_cast<const v4f*>(src);
        } else {
            ok = true;
            v4f v = zero;
            if (kr < k_end) {
                const float* row = X + (int64_t)kr * ldx;
                if (col0 + 3 < d) v = *reinterpret_cast<const v4f*>(row + col0);
                else
                    for (int e = 0; e < 4; ++e) if (col0 + e < d) v[e] = row[col0 + e];
            }
            return v;
        }
    };
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int v = tid + i * 256;          // GBK * BMT / 4 float4 in the A image
            ra[i] = fetch(k0 + v / (BMT / 4), m0 + 4 * (v % (BMT / 4)), oka[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + i * 256;          // 512 float4 in the [16][128] B image
            rb[i] = fetch(k0 + v / (GB / 4), n0 + 4 * (v % (GB / 4)), okb[i]);
        }
    };
    auto store = [&](float* stage) {
        const v4f zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MI; ++i) *reinterpret_cast<v4f*>(stage + 4 * (tid + i * 256)) = oka[i] ? ra[i] : zero;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<v4f*>(stage + GBK * BMT + 4 * (tid + i * 256)) = okb[i] ? rb[i] : zero;
    };

    // FAST (every tile column inside the matrix): stages whose 16 rows all exist need no checks at all — one uniform base
    // per stage plus a loop-invariant per-thread offset, raw registers to LDS; only the last, partial stage takes `load`.
    int offa[MI], offb[2];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int v = tid + i * 256;
        offa[i] = (v / (BMT / 4)) * (int)ldx + m0 + 4 * (v % (BMT / 4));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + i * 256;
        offb[i] = (v / (GB / 4)) * (int)ldx + n0 + 4 * (v % (GB / 4));
    }
    auto load_fast = [&](int k0) {
        const float* Xs = X + (int64_t)k0 * ldx;
#pragma unroll
        for (int i = 0; i < MI; ++i) ra[i] = *reinterpret_cast<const v4f*>(Xs + offa[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const v4f*>(Xs + offb[i]);
    };
    auto store_fast = [&](float* stage) {
#pragma unroll
        for (int i = 0; i < MI; ++i) *reinterpret_cast<v4f*>(stage + 4 * (tid + i * 256)) = ra[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<v4f*>(stage + GBK * BMT + 4 * (tid + i * 256)) = rb[i];
    };
    const int T = (k_end - k_begin + GBK - 1) / GBK;
    const int T_fast = FAST ? (k_end - k_begin) / GBK : 0;      // stages [0, T_fast) are complete

    if (T_fast > 0) { load_fast(k_begin); store_fast(smem); }
    else { load(k_begin); store(smem); }
    __syncthreads();
    for (int it = 0; it < T; ++it) {
        const float* As = smem + (it & 1) * STAGE;
        const float* Bs = As + GBK * BMT;
        const int nx = it + 1;
        if (nx < T_fast) load_fast(k_begin + nx * GBK);
        else if (nx < T) load(k_begin + nx * GBK);
        // fragments of step kk+1 are read from LDS before the MFMAs of step kk issue
        float a[2][MI], b[2][2];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[0][i] = As[l5 * BMT + wm0 + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[0][j] = Bs[l5 * GB + wn0 + j * 32 + l31];
#pragma unroll
        for (int kk = 0; kk < GBK / 2; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < GBK / 2) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[nxt][i] = As[((kk + 1) * 2 + l5) * BMT + wm0 + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[nxt][j] = Bs[((kk + 1) * 2 + l5) * GB + wn0 + j * 32 + l31];
            }
            if (GRAM_PIN_PREFETCH) __builtin_amdgcn_sched_barrier(0);     // left alone, hipcc sinks those reads BELOW the four MFMAs
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            if (GRAM_PIN_PREFETCH) __builtin_amdgcn_sched_barrier(0);
        }
        if (nx < T_fast) store_fast(smem + (nx & 1) * STAGE);
        else if (nx < T) store(smem + (nx & 1) * STAGE);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * l5;
                const int n = n0 + wn0 + j * 32 + l31;
                if (m < d && n < d) {
                    float* g = G + (int64_t)m * ldg + n;
                    if (use_atomic) atomicAdd(g, acc[i][j][r]);
                    else *g += acc[i][j][r];
                }
            }
}

// G[j][i] = G[i][j] for j > i (mirror the accumulated lower triangle), 32x32 tiles through LDS.
__global__ __launch_bounds__(256) void symmetrize_lower_f32_kernel(float* __restrict__ G, int d, int64_t ldg) {
    __shared__ float tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int i = bi * 32 + r, j = bj * 32 + tx;
        tile[r][tx] = (i < d && j < d) ? G[(int64_t)i * ldg + j] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        // element (row = bj*32 + r, col = bi*32 + tx) of the upper triangle <- lower (bi*32+tx, bj*32+r)
        const int i = bj * 32 + r, j = bi * 32 + tx;
        if (i < d && j < d && j > i) G[(int64_t)i * ldg + j] = tile[tx][r];
    }
}

// out[n][:] = (sum over the request's prompts of act[p][idx[p]][:]) / count, summed in prompt order.
__global__ __launch_bounds__(256) void gather_mean_f32_kernel(const float* __restrict__ act, int64_t ldb, int64_t lds_,
                                                               int c, int64_t S, const int64_t* __restrict__ idx,
                                                               const int64_t* __restrict__ seg, float* __restrict__ out,
                                                               int64_t ldo) {
    const int n = blockIdx.y;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= c) return;
    const int64_t p0 = seg[n], p1 = seg[n + 1];
    float sum = 0.f;
    for (int64_t p = p0; p < p1; ++p) {
        int64_t s = idx[p];
        s = s < 0 ? 0 : (s >= S ? S - 1 : s);  // host validates; clamp so a bad index can never fault
        sum += act[p * ldb + s * lds_ + col];
    }
    const float cnt = (float)(p1 - p0);
    out[(int64_t)n * ldo + col] = (p1 > p0) ? sum / cnt : 0.f;
}

}  // namespace emcid

using namespace emcid;

extern "C" {

int emcid_gram_accumulate_f32(const float* X, int64_t t, int64_t d, int64_t ldx, float* G, int64_t ldg, int ksplit,
                              void* stream) {
    EMCID_CHECK_ARG(X && G && t >= 0 && d > 0 && ldx >= d && ldg >= d);
    EMCID_CHECK_ARG(t < (1LL << 31) && d < (1 << 20) && ksplit >= 0);
    EMCID_CHECK_ARG(aligned16(X) && (ldx % 4 == 0));
    if (t == 0) return EMCID_OK;  // empty batch: no-op, like SecondMoment.add (runningstats.py:485-486)
    const int tiles = (int)((d + GB - 1) / GB);
    const int64_t lower = (int64_t)tiles * (tiles + 1) / 2;
    if (ksplit == 0) {
        // auto: ~6 rounds of the chip's 1024 workgroup slots (4 per CU), so the last, partly filled round costs little
        // (measured at d 3072: 76 -> 94 SYRK-TF from 4 to 20 splits), but chunks of >= 1024 tokens, so the 64 KB of
        // atomics a workgroup ends with stay small beside the 1 KB per token it streams; short inputs: >= 256 tokens.
        const int64_t want = (6144 + lower - 1) / lower;
        int64_t maxsplit = t >= 4096 ? t / 1024 : (((t + 255) / 256) < 4 ? (t + 255) / 256 : 4);
        // few tiles (d = 768: 21): 1024-token slabs cannot even give every CU a workgroup — go down to 256-token slabs
        // until the chip has ~4 workgroups per CU; the atomics of 21 tiles are small at any split
        const int64_t fill = (1024 + lower - 1) / lower, fine = (t + 255) / 256;
        if (maxsplit < fill) maxsplit = fill < fine ? fill : fine;
        ksplit = (int)(want < maxsplit ? want : maxsplit);
        if (ksplit >= 8) ksplit = (ksplit + 4) / 8 * 8;      // a multiple of the XCD count: see the kernel's grid comment
        if (ksplit < 1) ksplit = 1;
    }
    int64_t kchunk = round_up((t + ksplit - 1) / ksplit, GBK);
    ksplit = (int)((t + kchunk - 1) / kchunk);
    EMCID_CHECK_ARG(lower * ksplit < (1LL << 31));
    dim3 grid((unsigned)(lower * ksplit));
    ScopedProf sp(KC_GRAM, (hipStream_t)stream);
    const int atomic = ksplit > 1 ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    // FAST needs every tile column inside the matrix and 32-bit element offsets inside one 16-row stage
    const bool fast = d % GB == 0 && ldx * GBK < (1LL << 30);
#define EMCID_GRAM_LAUNCH(AL_, FA_) \
    hipLaunchKernelGGL((gram_f32_kernel<2, AL_, FA_>), grid, dim3(256), 0, st, X, (int)t, (int)d, ldx, G, ldg, (int)kchunk, atomic, ksplit)
    if (d % 4 != 0) EMCID_GRAM_LAUNCH(false, false);
    else if (!fast) EMCID_GRAM_LAUNCH(true, false);
    else EMCID_GRAM_LAUNCH(true, true);
#undef EMCID_GRAM_LAUNCH
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int emcid_symmetrize_lower_f32(float* G, int64_t d, int64_t ldg, void* stream) {
    EMCID_CHECK_ARG(G && d > 0 && ldg >= d && d < (1 << 20));
    const int tiles = (int)((d + 31) / 32);
    hipLaunchKernelGGL(symmetrize_lower_f32_kernel, dim3(tiles, tiles), dim3(256), 0, (hipStream_t)stream, G, (int)d, ldg);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int emcid_gather_mean_f32(const float* act, int64_t B, int64_t S, int64_t c, int64_t ldb, int64_t lds_, const int64_t* idx,
                          const int64_t* seg, int64_t N, float* out, int64_t ldo, void* stream) {
    EMCID_CHECK_ARG(act && idx && seg && out && B > 0 && S > 0 && c > 0 && N > 0 && ldo >= c);
    EMCID_CHECK_ARG(N < 65536 * 16 && c < (1 << 24));
    dim3 grid((unsigned)((c + 255) / 256), (unsigned)N);
    EMCID_CHECK_ARG(N <= 65535);
    ScopedProf sp(KC_GATHER, (hipStream_t)stream);
    hipLaunchKernelGGL(gather_mean_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, act, ldb, lds_, (int)c, S, idx, seg,
                       out, ldo);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

}  // extern "C"
