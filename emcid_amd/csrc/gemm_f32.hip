// The row-wise projections of the CLIP text-encoder forward on gfx950, fp32:  Y = act(X W^T + bias) + residual.
// Replaces what the reference gets from nn.Linear inside CLIPTextModel.forward for every prompt token
// (emcid/compute_z.py:2296-2316 runs the encoder; q/k/v, out_proj, fc1, fc2 are its GEMMs) — here on the rows of the
// prefix trie (emcid_amd/clip_forward.py), with the element-wise neighbours fused into the epilogue: bias, the MLP's
// activation (quick_gelu for CLIP-L, erf-gelu for bigG) and the residual add.
//
// Exact-f32 MFMA (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD = the 157 TF f32 peak).  X [M][K] and W [N][K] are both
// K-contiguous (activations row-major, nn.Linear weights), so the LDS image of a tile is a straight copy of 16-float row
// pieces, rows padded to 20 floats, and a fragment is fetched 16 B per lane with ds_read_b128: lane (row r, half s) reads
// k = 8q + 4s .. 8q + 4s + 3 of its row, and MFMA k-step e of the block pairs element e of the A and B vectors, i.e. the
// two k slots of the instruction carry k = 8q + e and 8q + 4 + e — a permutation of the contraction index applied to both
// operands alike.  One b128 per 32 rows per 4 MFMAs; row stride 20 dwords puts the 16 lanes of every b128 lane group on
// 16 distinct 4-bank groups (MI355X_MICROARCH.md, LDS table).
//
// Pipeline per 16-deep K stage (two 8-deep blocks), one barrier per stage, placed in the MIDDLE of the stage's MFMAs:
//     MFMAs of block 0          | issued under them: fragment reads of block 1 (same LDS buffer)
//     wait for the global loads of stage it+1 (issued one stage ago), write them to the other LDS buffer
//     barrier                   | nobody reads the other buffer any more, everybody has written it
//     issue the global loads of stage it+2
//     MFMAs of block 1          | issued under them: fragment reads of block 0 of stage it+1 (other buffer)
// so no LDS or barrier latency sits between two MFMAs of a wave even at one wave per SIMD (a 240-tile launch gives every
// compute unit ONE workgroup), and no global load is outstanding at the barrier (hipcc drains vmcnt before s_barrier).
//
// Tiles: waves in a WM x WN grid, each (32 MI) x (32 NJ).  160 x 128 (1 x 4 waves of 160 x 32) is the workhorse: the trie
// forward's row counts are multiples of 256 (6 400 at N = 1000 x 3 templates) and N in {768, 2304, 3072}, where it gives
// 240 / 720 / 960 tiles = 0.94 / 2.81 / 3.75 rounds of the 256 compute units (128 x 128: 300 tiles = 59 % of two rounds).
#include "common.h"

namespace emcid {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int LBK = 16;      // K depth of a stage
constexpr int LLD = 20;      // LDS row stride in floats (16 + 4 padding)

enum LinearAct : int { ACT_NONE = 0, ACT_QUICK_GELU = 1, ACT_GELU_ERF = 2 };

template <int MI, int NJ, int WM, int WN>
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ W,
                                                          int64_t ldw, const float* __restrict__ bias,
                                                          const float* __restrict__ res, int64_t ldr, float* __restrict__ Y,
                                                          int64_t ldy, int M, int N, int K, int act, int tiles_n, int tiles) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
    constexpr int VA = (BM * 4 + 255) / 256, VB = (BN * 4 + 255) / 256;      // float4 per thread and stage
    constexpr int STAGE = (BM + BN) * LLD;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    // Workgroups go to the 8 XCDs round-robin by linear id.  XCD x takes the tiles [x * per, (x + 1) * per): a contiguous
    // run of row tiles with all their column tiles, so the rows of X an XCD streams are its own and every column tile of W
    // is re-read from that XCD's L2.
    const int per = (tiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= tiles || (int)(blockIdx.x >> 3) >= per) return;
    const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
    const int m0 = bm * BM, n0 = bn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, l5 = lane >> 5;
    const int wm0 = (wave / WN) * (32 * MI), wn0 = (wave % WN) * (32 * NJ);

    // global -> register staging: vector v = tid + 256 s of an image is row v / 4, floats 4 (v % 4) .. + 3 of the stage
    const float* pa[VA];
    const float* pb[VB];
    int wa[VA], wb[VB];
#pragma unroll
    for (int s = 0; s < VA; ++s) {
        const int v = tid + 256 * s, row = min(v >> 2, BM - 1);
        pa[s] = X + (int64_t)min(m0 + row, M - 1) * ldx + 4 * (v & 3);          // rows past M: a valid row, never stored
        wa[s] = row * LLD + 4 * (v & 3);
    }
#pragma unroll
    for (int s = 0; s < VB; ++s) {
        const int v = tid + 256 * s, row = min(v >> 2, BN - 1);
        pb[s] = W + (int64_t)min(n0 + row, N - 1) * ldw + 4 * (v & 3);
        wb[s] = BM * LLD + row * LLD + 4 * (v & 3);
    }
    constexpr bool TAIL_A = (BM * 4) % 256 != 0, TAIL_B = (BN * 4) % 256 != 0;
    const bool last_a = !TAIL_A || tid + 256 * (VA - 1) < BM * 4;               // wave-uniform (multiples of 64 threads)
    const bool last_b = !TAIL_B || tid + 256 * (VB - 1) < BN * 4;

    v4f ga[VA], gb[VB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int s = 0; s < VA; ++s)
            if (s + 1 < VA || last_a) ga[s] = *reinterpret_cast<const v4f*>(pa[s] + k0);
#pragma unroll
        for (int s = 0; s < VB; ++s)
            if (s + 1 < VB || last_b) gb[s] = *reinterpret_cast<const v4f*>(pb[s] + k0);
    };
    auto lstore = [&](float* stage) {
#pragma unroll
        for (int s = 0; s < VA; ++s)
            if (s + 1 < VA || last_a) *reinterpret_cast<v4f*>(stage + wa[s]) = ga[s];
#pragma unroll
        for (int s = 0; s < VB; ++s)
            if (s + 1 < VB || last_b) *reinterpret_cast<v4f*>(stage + wb[s]) = gb[s];
    };

    v16f acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fa_off = (wm0 + l31) * LLD + 4 * l5;                 // + 32 i rows, + 8 q floats
    const int fb_off = BM * LLD + (wn0 + l31) * LLD + 4 * l5;
    v4f fa[2][MI], fb[2][NJ];
    auto fread = [&](const float* stage, int q, int slot) {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[slot][i] = *reinterpret_cast<const v4f*>(stage + fa_off + i * 32 * LLD + 8 * q);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[slot][j] = *reinterpret_cast<const v4f*>(stage + fb_off + j * 32 * LLD + 8 * q);
    };
    auto mfmas = [&](int slot, int e_lo, int e_hi) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e >= e_lo && e < e_hi)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][i][e], fb[slot][j][e], acc[i][j], 0, 0, 0);
    };

    // The fragment reads of the NEXT block are issued right after the first k-step's MFMAs of the current one: the wait in front
    // of those first MFMAs then covers only reads issued a whole block ago (hipcc emits lgkmcnt(0) there, not a counted wait).
    const int T = K / LBK;
    gload(0);
    lstore(smem);
    __syncthreads();
    if (T > 1) gload(LBK);
    fread(smem, 0, 0);
    for (int it = 0; it < T; ++it) {
        float* cur = smem + (it & 1) * STAGE;
        float* oth = smem + ((it + 1) & 1) * STAGE;
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        fread(cur, 1, 1);                      // block 1 of this stage: lands under the MFMAs of block 0
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 1, 4);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < T) lstore(oth);           // stage it+1 (loaded during the previous stage) -> the other buffer
        __syncthreads();
        if (it + 2 < T) gload((it + 2) * LBK);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < T) fread(oth, 0, 0);      // block 0 of the next stage: lands under the MFMAs of block 1
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 1, 4);
    }

    // epilogue: C[row][col], row = (r & 3) + 8 (r >> 2) + 4 l5 inside a 32 x 32 block, col = l31: a wave instruction writes two
    // runs of 32 consecutive floats.  The activation is chosen once, outside the unrolled element loops.
    auto epilogue = [&](auto actfn) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wn0 + j * 32 + l31;
            const bool n_ok = n < N;
            const float bv = (bias != nullptr && n_ok) ? bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * l5;
                    if (n_ok && m < M) {
                        float v = actfn(acc[i][j][r] + bv);
                        if (res != nullptr) v += res[(int64_t)m * ldr + n];
                        Y[(int64_t)m * ldy + n] = v;
                    }
                }
        }
    };
    if (act == ACT_QUICK_GELU) epilogue([](float x) { return x / (1.0f + __expf(-1.702f * x)); });       // x * sigmoid(1.702 x)
    else if (act == ACT_GELU_ERF) epilogue([](float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); });
    else epilogue([](float x) { return x; });
}

struct LinearCfg { int bm, bn; };
static const LinearCfg kLinearCfgs[] = {{160, 128}, {128, 128}, {256, 128}, {64, 64}};

}  // namespace emcid

using namespace emcid;

extern "C" {

int emcid_linear_f32(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual,
                     int64_t ldr, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, int act, int cfg, void* stream) {
    EMCID_CHECK_ARG(X && W && Y && M > 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N);
    EMCID_CHECK_ARG(K % LBK == 0 && ldx % 4 == 0 && ldw % 4 == 0 && aligned16(X) && aligned16(W));
    EMCID_CHECK_ARG(M < (1 << 24) && N < (1 << 24) && K < (1 << 24) && (residual == nullptr || ldr >= N));
    EMCID_CHECK_ARG(act >= ACT_NONE && act <= ACT_GELU_ERF && cfg >= -1 && cfg < 4);
    if (cfg < 0) {
        // every compute unit gets ceil(tiles / 256) tiles (a second resident workgroup shares its matrix pipe): the launch
        // lasts rounds * BM * BN; the small tile pays ~25 % more per flop (one MFMA per fragment pair)
        double best = 0.0;
        for (int c = 0; c < 4; ++c) {
            const int64_t t = ((M + kLinearCfgs[c].bm - 1) / kLinearCfgs[c].bm) * ((N + kLinearCfgs[c].bn - 1) / kLinearCfgs[c].bn);
            const double cost = (double)((t + 255) / 256) * kLinearCfgs[c].bm * kLinearCfgs[c].bn * (c == 3 ? 1.25 : c == 2 ? 0.97 : 1.0);
            if (cfg < 0 || cost < best) best = cost, cfg = c;
        }
    }
    const int bm = kLinearCfgs[cfg].bm, bn = kLinearCfgs[cfg].bn;
    const int tiles_m = (int)((M + bm - 1) / bm), tiles_n = (int)((N + bn - 1) / bn);
    const int tiles = tiles_m * tiles_n;
    const int per = (tiles + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    ScopedProf sp(KC_LINEAR, st);
#define EMCID_LINEAR_LAUNCH(MI_, NJ_, WM_, WN_)                                                                              \
    hipLaunchKernelGGL((linear_f32_kernel<MI_, NJ_, WM_, WN_>), dim3((unsigned)(per * 8)), dim3(256), 0, st, X, ldx, W, ldw, \
                       bias, residual, ldr, Y, ldy, (int)M, (int)N, (int)K, act, tiles_n, tiles)
    switch (cfg) {
        case 0: EMCID_LINEAR_LAUNCH(5, 1, 1, 4); break;
        case 1: EMCID_LINEAR_LAUNCH(2, 2, 2, 2); break;
        case 2: EMCID_LINEAR_LAUNCH(4, 2, 2, 2); break;
        default: EMCID_LINEAR_LAUNCH(1, 1, 2, 2); break;
    }
#undef EMCID_LINEAR_LAUNCH
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

}  // extern "C"
