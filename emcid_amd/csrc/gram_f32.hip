// Stage 0 of EMCID on gfx950: running second moment  G += X^T X  of fc2-input features, fp32.
// Replaces `self.mom2 += a.t().mm(a)` (reference: util/runningstats.py:493; driver emcid/layer_stats.py:208-219).
//
// SYRK on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32): only tiles on or below the diagonal are computed.
// X is [t][d] row-major, so both MFMA operands are d-contiguous slices of the same token rows: the LDS
// image of a tile is a straight copy [BK tokens][128 features]; a fragment read is 32 consecutive floats
// per token row (ds_read_b32, lanes 0-31 token k, lanes 32-63 token k+1) and is conflict-free.
// The token dimension can be split over blockIdx.z (ksplit); partial tiles are then added with
// global_atomic_add_f32 (G is an accumulator anyway), otherwise with a plain read-modify-write.
// Also here: the small byte-moving kernels of the K/Z assembly (gather + per-request mean) and the
// lower->upper mirror used when the moment is read.
#include "common.h"

namespace emcid {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int GB = 128;   // output tile edge
constexpr int GBK = 16;   // tokens per stage

__global__ __launch_bounds__(256) void gram_f32_kernel(const float* __restrict__ X, int t, int d, int64_t ldx,
                                                        float* __restrict__ G, int64_t ldg, int kchunk, int use_atomic) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * GBK * GB];
    const int bm = blockIdx.y, bn = blockIdx.x;
    if (bn > bm) return;  // lower triangle of tiles only
    const int m0 = bm * GB, n0 = bn * GB;
    const int k_begin = blockIdx.z * kchunk;
    const int k_end = min(t, k_begin + kchunk);
    if (k_begin >= k_end) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int l31 = lane & 31, l5 = lane >> 5;

    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // each thread stages 2 float4 of the A tile and 2 of the B tile per stage
    v4f ra[2], rb[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + i * 256;          // 512 float4 per [16][128] tile
            const int kr = k0 + v / (GB / 4);
            const int c = 4 * (v % (GB / 4));
            v4f a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (kr < k_end) {
                const float* row = X + (int64_t)kr * ldx;
                if (m0 + c + 3 < d) a = *reinterpret_cast<const v4f*>(row + m0 + c);
                else
                    for (int e = 0; e < 4; ++e) if (m0 + c + e < d) a[e] = row[m0 + c + e];
                if (n0 + c + 3 < d) b = *reinterpret_cast<const v4f*>(row + n0 + c);
                else
                    for (int e = 0; e < 4; ++e) if (n0 + c + e < d) b[e] = row[n0 + c + e];
            }
            ra[i] = a; rb[i] = b;
        }
    };
    auto store = [&](float* stage) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + i * 256;
            *reinterpret_cast<v4f*>(stage + 4 * v) = ra[i];
            *reinterpret_cast<v4f*>(stage + GBK * GB + 4 * v) = rb[i];
        }
    };

    load(k_begin);
    store(smem);
    __syncthreads();
    const int T = (k_end - k_begin + GBK - 1) / GBK;
    for (int it = 0; it < T; ++it) {
        const float* As = smem + (it & 1) * (2 * GBK * GB);
        const float* Bs = As + GBK * GB;
        const bool more = it + 1 < T;
        if (more) load(k_begin + (it + 1) * GBK);
#pragma unroll
        for (int kk = 0; kk < GBK / 2; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(kk * 2 + l5) * GB + wm0 + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[(kk * 2 + l5) * GB + wn0 + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) store(smem + ((it + 1) & 1) * (2 * GBK * GB));
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
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
    if (ksplit == 0) {
        // auto: enough workgroups for ~4 per CU, but at least 256 tokens per chunk
        const int64_t lower = (int64_t)tiles * (tiles + 1) / 2;
        int64_t want = (1024 + lower - 1) / lower;
        int64_t maxsplit = (t + 255) / 256;
        ksplit = (int)(want < maxsplit ? want : maxsplit);
        if (ksplit < 1) ksplit = 1;
    }
    int64_t kchunk = round_up((t + ksplit - 1) / ksplit, GBK);
    ksplit = (int)((t + kchunk - 1) / kchunk);
    dim3 grid(tiles, tiles, ksplit);
    ScopedProf sp(KC_GRAM, (hipStream_t)stream);
    hipLaunchKernelGGL(gram_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, (int)t, (int)d, ldx, G, ldg, (int)kchunk,
                       ksplit > 1 ? 1 : 0);
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
