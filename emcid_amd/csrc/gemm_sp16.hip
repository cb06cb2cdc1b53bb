// The row-wise projections of the CLIP text-encoder forward on gfx950 at fp32 accuracy on the 16-bit matrix pipe:
//     Y = act(X W^T + bias) + residual          (emcid/compute_z.py:2296-2316 runs the encoder; q/k/v, out_proj, fc1, fc2)
// The exact-f32 MFMA of gemm_f32.hip runs at 1/16 of the f16 rate (MI355X_MICROARCH.md, Matrix cores).  Here every fp32 operand
// row is carried as TWO fp16 numbers per element under a per-row power-of-two scale,
//     x * 2^e = hi + lo,   hi = fp16(x 2^e),  lo = fp16(x 2^e - hi)       (e: the row's largest |x| lands in [2^14, 2^15))
// which keeps 22-23 significant bits of x (the sign of lo is the extra one; |lo| <= ulp(hi) / 2, and with the row's maximum at
// the top of the fp16 range a lo below the fp16 normal range is below 2^-28 of the row's maximum), and the product is the three
// MFMAs hi.hi + hi.lo + lo.hi accumulated in fp32 (the lo.lo term, <= 2^-22 of a product, is dropped; fp16 x fp16 products
// are exact in the fp32 accumulate): 3/16 of the f32-MFMA issue time for the same contraction.  The scales are powers of two, so
// undoing them in the epilogue is exact: Y[m][n] = acc * 2^-(e_x[m] + e_w[n]).
//
// Storage of a split matrix ("planes", one 4-byte unit per element like the fp32 matrix it stands for, so row views and row
// gathers work on it as on an int32 matrix): per row, groups of 8 consecutive k: [hi k..k+7 (16 B)][lo k..k+7 (16 B)].  A lane's
// MFMA fragment (8 consecutive k of one row, for v_mfma_f32_16x16x32_f16 and v_mfma_f32_32x32x16_f16 alike) is then ONE
// ds_read_b128 per plane, and a stage's 32 k of a row are one whole 128-byte line in global memory.
//
// The MFMA runs "transposed": A operand = W rows (n), B operand = X rows (m), so a lane of the result holds ONE row m and four
// consecutive n per register quad.
//
// Kernels: `linear_sp16_dma16_kernel` — the default: operands by LDS-DMA, 16x16x32 MFMAs, epilogue through LDS, four tile forms
// (round 5) — and `linear_sp16_kernel` — register-staged operands, 32x32x16 MFMAs (round 4): kept for launches of less than one
// 64 x 64 tile per compute unit, and its K loop (`sp_accumulate`) for the Stage-0 Gram.
#include "common.h"
#include "sp16.h"

#include <algorithm>

namespace emcid {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

#ifndef SP16_ONE_SLOT
#define SP16_ONE_SLOT 0
#endif
constexpr int SPK = 32;            // k per stage and wave group (a stage is 32 KS deep)
// LDS bytes per tile row and stage: 128 KS + 16.  KS = 1: 36 dwords, the 16 lanes of every b128 lane group fall on 16 distinct
// 4-bank groups (36 r mod 64 = 4 (9 r mod 16)); KS = 2: 68 dwords (68 r mod 64 = 4 r).

enum SpAct : int { SP_ACT_NONE = 0, SP_ACT_QUICK_GELU = 1, SP_ACT_GELU_ERF = 2 };

template <int V> struct SpIC { static constexpr int value = V; };

__device__ __forceinline__ bool sp_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct SpArgs {
    const uint32_t* X; int64_t ldx; const float* xs;      // planes of X [M][K], 2^-e per row
    const uint32_t* W; int64_t ldw; const float* ws;      // planes of W [N][K], 2^-e per row
    const float* bias;
    const float* res; int64_t ldr;
    float* Y; int64_t ldy;                                // fp32 result (may be null when only the planes are wanted)
    uint32_t* P; int64_t ldp; const float* ps;            // planes of the result under the caller's per-row scale 2^e (or null)
    int M, N, K, act, tiles_n, tiles, rb;
    long long* stamps;                                    // diagnostic builds only (emcid_debug_linear_sp16_stamps); else null
};

// Tile order as in gemm_f32.hip: super-rows of `rb` row tiles, column-major inside a super-row.
__device__ __forceinline__ void sp_tile_of(const SpArgs& a, int tile, int& bm, int& bn) {
    const int tiles_m = a.tiles / a.tiles_n, sr = tile / (a.rb * a.tiles_n), rem = tile - sr * a.rb * a.tiles_n;
    const int rows = min(a.rb, tiles_m - sr * a.rb);
    bn = rem / rows;
    bm = sr * a.rb + (rem - bn * rows);
}

// MJ x NI blocks of 32 x 32 per wave (m x n), WM x WN waves per K group, KS K groups per workgroup (KS = 2: waves 0..WM WN - 1
// contract the first 32 of a 64-deep stage, the others the second 32, both over the whole tile; the halves meet through LDS at
// the end — gemm_f32.hip's scheme: a launch whose tiles give a compute unit ONE workgroup still has two waves per SIMD), PF
// register sets of staged global loads.
template <int MJ, int NI, int WM, int WN, int PF, int KS>
struct SpGeom {
    static constexpr int NW = WM * WN;                        // waves per K group
    static constexpr int NT = 64 * NW * KS;
    static constexpr int BM = 32 * MJ * WM, BN = 32 * NI * WN;
    static constexpr int PPR = 8 * KS;                        // 16-byte pieces per row and stage
    static constexpr int SBK = SPK * KS;
    static constexpr int ROW = 128 * KS + 16;                 // LDS bytes per row and stage
    static constexpr int VA = (BM * PPR + NT - 1) / NT, VB = (BN * PPR + NT - 1) / NT;      // 16-byte pieces per thread and stage
    static constexpr int STAGE = (BM + BN) * ROW;                                          // bytes
    static constexpr int NBLK = MJ * NI, HB = KS == 2 ? (NBLK + 1) / 2 : NBLK;
    static constexpr int RED = KS == 2 ? NW * NBLK * 4 * 64 * 16 : 0;                       // bytes the final hand-over needs
    static constexpr int SMEM = 2 * STAGE > RED ? 2 * STAGE : RED;
};

template <int MJ, int NI, int WM, int WN, int PF, int DBG, int KS>
__device__ __forceinline__ void sp_accumulate(const SpArgs& a, int m0, int n0, int T, unsigned char* smem, v16f (&acc)[NI][MJ]) {
    using G = SpGeom<MJ, NI, WM, WN, PF, KS>;
    constexpr int NT = G::NT, BM = G::BM, BN = G::BN, VA = G::VA, VB = G::VB, STAGE = G::STAGE, PPR = G::PPR, SPROW = G::ROW,
                  SBK = G::SBK;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) % G::NW, grp = (tid >> 6) / G::NW;
    const int l31 = lane & 31, l5 = lane >> 5;
    const int wm0 = (wave / WN) * (32 * MJ), wn0 = (wave % WN) * (32 * NI);

    // global -> registers -> LDS: piece v = tid + NT s of an operand's stage image is 16 bytes of one row; the 8 pieces of a row
    // (one 128-byte line) go to 8 consecutive lanes, which is also one conflict-free ds_write_b128 group (32 banks).
    const uint32_t* pa[VA];
    const uint32_t* pb[VB];
    int wa[VA], wb[VB];
#pragma unroll
    for (int s = 0; s < VA; ++s) {
        const int v = tid + NT * s, row = min(v / PPR, BM - 1);
        pa[s] = a.X + (int64_t)min(m0 + row, a.M - 1) * a.ldx + 4 * (v % PPR);        // rows past M: a valid row, never stored
        wa[s] = row * SPROW + 16 * (v % PPR);
    }
#pragma unroll
    for (int s = 0; s < VB; ++s) {
        const int v = tid + NT * s, row = min(v / PPR, BN - 1);
        pb[s] = a.W + (int64_t)min(n0 + row, a.N - 1) * a.ldw + 4 * (v % PPR);
        wb[s] = (BM + row) * SPROW + 16 * (v % PPR);
    }
    constexpr bool TAIL_A = (BM * PPR) % NT != 0, TAIL_B = (BN * PPR) % NT != 0;
    const bool last_a = !TAIL_A || tid + NT * (VA - 1) < BM * PPR;
    const bool last_b = !TAIL_B || tid + NT * (VB - 1) < BN * PPR;

    v4u ga[PF][VA], gb[PF][VB];
    auto gload = [&](int it, auto rc) __attribute__((always_inline)) {
        constexpr int R = decltype(rc)::value;
        const int k0 = it * SBK;
#pragma unroll
        for (int s = 0; s < VA; ++s)
            if (s + 1 < VA || last_a) ga[R][s] = *reinterpret_cast<const v4u*>(pa[s] + k0);
#pragma unroll
        for (int s = 0; s < VB; ++s)
            if (s + 1 < VB || last_b) gb[R][s] = *reinterpret_cast<const v4u*>(pb[s] + k0);
    };
    auto lstore = [&](unsigned char* stage, auto rc) __attribute__((always_inline)) {
        constexpr int R = decltype(rc)::value;
#pragma unroll
        for (int s = 0; s < VA; ++s)
            if (s + 1 < VA || last_a) *reinterpret_cast<v4u*>(stage + wa[s]) = ga[R][s];
#pragma unroll
        for (int s = 0; s < VB; ++s)
            if (s + 1 < VB || last_b) *reinterpret_cast<v4u*>(stage + wb[s]) = gb[R][s];
    };

    // fragments: lane (row l31, half l5) of k16-step t reads group 2 t + l5 of its row: hi at +0, lo at +16
    const int fx_off = (wm0 + l31) * SPROW + 32 * l5 + 128 * grp;
    const int fw_off = (BM + wn0 + l31) * SPROW + 32 * l5 + 128 * grp;
    constexpr int NSLOT = ((MJ * NI > 4 && SP16_ONE_SLOT) || MJ * NI > 5) ? 1 : 2;
    v8h xh[NSLOT][MJ], xl[NSLOT][MJ], wh[NSLOT][NI], wl[NSLOT][NI];
    auto fread = [&](const unsigned char* stage, int t, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            xh[slot][j] = *reinterpret_cast<const v8h*>(stage + fx_off + j * 32 * SPROW + 64 * t);
            xl[slot][j] = *reinterpret_cast<const v8h*>(stage + fx_off + j * 32 * SPROW + 64 * t + 16);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            wh[slot][i] = *reinterpret_cast<const v8h*>(stage + fw_off + i * 32 * SPROW + 64 * t);
            wl[slot][i] = *reinterpret_cast<const v8h*>(stage + fw_off + i * 32 * SPROW + 64 * t + 16);
        }
    };
    // the two small products first, then hi x hi; p selects the product so that the three MFMAs of one accumulator are a
    // block apart in the issue order
    auto mfmas = [&](int slot, int p_lo, int p_hi) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            if (p >= p_lo && p < p_hi)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p == 0 ? wl[slot][i] : wh[slot][i],
                                                                           p == 1 ? xl[slot][j] : xh[slot][j], acc[i][j], 0, 0, 0);
    };

    // Everything inside the loop is unconditional — past the end the loads re-fetch the last stage and the LDS writes / fragment
    // reads touch a buffer nobody uses any more — because a load or store skipped under a branch makes hipcc's s_waitcnt for the
    // older register set fall back to vmcnt(0), i.e. an effective prefetch distance of one stage.  DBG (timing experiments only,
    // results wrong): 1 = no global loads inside the loop, 2 = also no LDS stores, 3 = also no fragment reads.
    gload(0, SpIC<0>{});
    lstore(smem, SpIC<0>{});
    gload(min(1, T - 1), SpIC<1 % PF>{});
    if constexpr (PF > 1) gload(min(2, T - 1), SpIC<2 % PF>{});
    __syncthreads();
    constexpr bool ONE_SLOT = (MJ * NI > 4 && SP16_ONE_SLOT) || MJ * NI > 5;       // five blocks per wave: a second set of fragments would not fit in 256 registers
    if constexpr (!ONE_SLOT) fread(smem, 0, 0);
    if constexpr (DBG >= 3) fread(smem, 1, 1);
    auto body = [&](auto rc, int it) __attribute__((always_inline)) {
        unsigned char* cur = smem + (it & 1) * STAGE;
        unsigned char* oth = smem + ((it + 1) & 1) * STAGE;
        if constexpr (ONE_SLOT) {
            // the partner wave of the other K group covers this wave's LDS latency
            fread(cur, 0, 0);
            mfmas(0, 0, 3);
            __builtin_amdgcn_sched_barrier(0);
            fread(cur, 1, 0);
            lstore(oth, rc);
            __syncthreads();
            gload(min(it + 1 + PF, T - 1), rc);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(0, 0, 3);
            return;
        }
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG < 3) fread(cur, 1, 1);            // second k16 step of this stage: lands under the MFMAs of the first
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG < 2) lstore(oth, rc);             // stage it+1 (loaded PF stages ago) -> the other buffer
        __syncthreads();
        if constexpr (DBG < 1) gload(min(it + 1 + PF, T - 1), rc);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG < 3) fread(oth, 0, 0);            // first k16 step of the next stage
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 1, 3);
    };
    if constexpr (PF == 1) {
        for (int it = 0; it < T; ++it) body(SpIC<0>{}, it);
    } else {
        int it = 0;
        for (; it + 1 < T; it += 2) {
            body(SpIC<1>{}, it);
            body(SpIC<0>{}, it + 1);
        }
        if (it < T) body(SpIC<1>{}, it);
    }
}

// Epilogue.  acc[i][j][4 q + e] = element (m, n): m = m0 + wm0 + 32 j + l31, n = n0 + wn0 + 32 i + 8 q + 4 l5 + e.
template <int MJ, int NI, int WM, int WN, int KS>
__device__ __forceinline__ void sp_finish(const SpArgs& a, int m0, int n0, unsigned char* smem, v16f (&acc)[NI][MJ]) {
    constexpr int BM = 32 * MJ * WM, BN = 32 * NI * WN, NW = WM * WN, NBLK = MJ * NI, HB = KS == 2 ? (NBLK + 1) / 2 : NBLK;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) % NW, grp = (tid >> 6) / NW;
    const int l31 = lane & 31, l5 = lane >> 5;
    const int wm0 = (wave / WN) * (32 * MJ), wn0 = (wave % WN) * (32 * NI);
    // KS == 2: the two wave groups hold the two halves of every sum.  They swap HALF of their accumulator blocks through LDS
    // (lane-linear 16-byte pieces: [wave][block][4][lane]) — group 0 ends up owning blocks [0, HB), group 1 blocks [HB, NBLK),
    // each complete — so that all eight waves share the epilogue.  Fixed order of the two addends: bit-reproducible.
    if constexpr (KS == 2) {
        __syncthreads();                                       // everybody is done with the stage buffers
        v4f* red = reinterpret_cast<v4f*>(smem) + wave * (NBLK * 4 * 64) + lane;
        auto put = [&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4f v = {acc[t / MJ][t % MJ][4 * q], acc[t / MJ][t % MJ][4 * q + 1], acc[t / MJ][t % MJ][4 * q + 2],
                         acc[t / MJ][t % MJ][4 * q + 3]};
                red[(t * 4 + q) * 64] = v;
            }
        };
        auto take = [&](auto tc, auto first) __attribute__((always_inline)) {      // group 0's partial first, whoever adds
            constexpr int t = decltype(tc)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4f v = red[(t * 4 + q) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = acc[t / MJ][t % MJ][4 * q + e];
                    acc[t / MJ][t % MJ][4 * q + e] = decltype(first)::value ? c + v[e] : v[e] + c;
                }
            }
        };
        static_assert(NBLK <= 8, "blocks per wave");
        if (grp == 1) {                                        // blocks the OTHER group will own
            if constexpr (0 < HB) put(SpIC<0>{});
            if constexpr (1 < HB) put(SpIC<1>{});
            if constexpr (2 < HB) put(SpIC<2>{});
            if constexpr (3 < HB) put(SpIC<3>{});
        } else {
            if constexpr (0 >= HB && 0 < NBLK) put(SpIC<0>{});
            if constexpr (1 >= HB && 1 < NBLK) put(SpIC<1>{});
            if constexpr (2 >= HB && 2 < NBLK) put(SpIC<2>{});
            if constexpr (3 >= HB && 3 < NBLK) put(SpIC<3>{});
            if constexpr (4 >= HB && 4 < NBLK) put(SpIC<4>{});
            if constexpr (5 >= HB && 5 < NBLK) put(SpIC<5>{});
            if constexpr (6 >= HB && 6 < NBLK) put(SpIC<6>{});
            if constexpr (7 >= HB && 7 < NBLK) put(SpIC<7>{});
        }
        __syncthreads();
        if (grp == 0) {
            if constexpr (0 < HB) take(SpIC<0>{}, SpIC<1>{});
            if constexpr (1 < HB) take(SpIC<1>{}, SpIC<1>{});
            if constexpr (2 < HB) take(SpIC<2>{}, SpIC<1>{});
            if constexpr (3 < HB) take(SpIC<3>{}, SpIC<1>{});
        } else {
            if constexpr (0 >= HB && 0 < NBLK) take(SpIC<0>{}, SpIC<0>{});
            if constexpr (1 >= HB && 1 < NBLK) take(SpIC<1>{}, SpIC<0>{});
            if constexpr (2 >= HB && 2 < NBLK) take(SpIC<2>{}, SpIC<0>{});
            if constexpr (3 >= HB && 3 < NBLK) take(SpIC<3>{}, SpIC<0>{});
            if constexpr (4 >= HB && 4 < NBLK) take(SpIC<4>{}, SpIC<0>{});
            if constexpr (5 >= HB && 5 < NBLK) take(SpIC<5>{}, SpIC<0>{});
            if constexpr (6 >= HB && 6 < NBLK) take(SpIC<6>{}, SpIC<0>{});
            if constexpr (7 >= HB && 7 < NBLK) take(SpIC<7>{}, SpIC<0>{});
        }
    }
    const float* __restrict__ bias = a.bias;
    const float* __restrict__ res = a.res;
    const float* __restrict__ ws = a.ws;
    float* __restrict__ Y = a.Y;
    uint32_t* __restrict__ P = a.P;
    const int M = a.M, N = a.N;
    const int64_t ldr = a.ldr, ldy = a.ldy, ldp = a.ldp;
    const bool vec_ok = (N & 3) == 0 && (Y == nullptr || ((ldy & 3) == 0 && sp_al16(Y))) &&
                        (res == nullptr || ((ldr & 3) == 0 && sp_al16(res))) && (bias == nullptr || sp_al16(bias)) && sp_al16(ws);
    const bool interior = m0 + BM <= M && n0 + BN <= N && vec_ok;
    auto epilogue = [&](auto actfn) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int m = m0 + wm0 + 32 * j + l31;
            const bool m_ok = m < M;
            const float sx = a.xs[min(m, M - 1)];
            const float sp = a.ps != nullptr ? a.ps[min(m, M - 1)] : 1.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (KS == 2 && ((i * MJ + j) < HB) != (grp == 0)) continue;        // the other group's block
                const int nb = n0 + wn0 + 32 * i + 4 * l5;
                if (interior) {
                    v4f rv[4];
                    if (res != nullptr) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const v4f*>(res + (int64_t)m * ldr + nb + 8 * q);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = nb + 8 * q;
                        const v4f w4 = *reinterpret_cast<const v4f*>(ws + n);
                        v4f b4 = {0.f, 0.f, 0.f, 0.f};
                        if (bias != nullptr) b4 = *reinterpret_cast<const v4f*>(bias + n);
                        v4f v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = actfn(acc[i][j][4 * q + e] * (sx * w4[e]) + b4[e]);
                            if (res != nullptr) v[e] += rv[q][e];
                        }
                        if (Y != nullptr) *reinterpret_cast<v4f*>(Y + (int64_t)m * ldy + n) = v;
                        if (P != nullptr) {
                            v4h h, l;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float t = v[e] * sp;
                                h[e] = (_Float16)t;
                                l[e] = (_Float16)(t - (float)h[e]);
                            }
                            unsigned char* dst = reinterpret_cast<unsigned char*>(P + (int64_t)m * ldp) + (n >> 3) * 32 + 8 * l5;
                            *reinterpret_cast<v4h*>(dst) = h;
                            *reinterpret_cast<v4h*>(dst + 16) = l;
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int n = nb + 8 * q + e;
                            if (m_ok && n < N) {
                                float v = actfn(acc[i][j][4 * q + e] * (sx * ws[n]) + (bias != nullptr ? bias[n] : 0.f));
                                if (res != nullptr) v += res[(int64_t)m * ldr + n];
                                if (Y != nullptr) Y[(int64_t)m * ldy + n] = v;
                                if (P != nullptr) {
                                    const float t = v * sp;
                                    const _Float16 h = (_Float16)t, l = (_Float16)(t - (float)h);
                                    _Float16* dst = reinterpret_cast<_Float16*>(reinterpret_cast<unsigned char*>(P + (int64_t)m * ldp) +
                                                                               (n >> 3) * 32) + (n & 7);
                                    dst[0] = h;
                                    dst[8] = l;
                                }
                            }
                        }
                }
            }
        }
    };
    if (a.act == SP_ACT_QUICK_GELU) epilogue([](float x) { return x / (1.0f + __expf(-1.702f * x)); });
    else if (a.act == SP_ACT_GELU_ERF) epilogue([](float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); });
    else epilogue([](float x) { return x; });
}

template <int MJ, int NI, int WM, int WN, int PF, int WPE, int DBG, int KS>
__global__ __launch_bounds__(64 * WM * WN * KS) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void linear_sp16_kernel(SpArgs a) {
    using G = SpGeom<MJ, NI, WM, WN, PF, KS>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::SMEM];
    // XCD x (workgroups go round-robin by linear id) takes the tiles [x per, (x + 1) per) of the super-row order
    const int per = (a.tiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= a.tiles || (int)(blockIdx.x >> 3) >= per) return;
    int bm, bn;
    sp_tile_of(a, tile, bm, bn);
    v16f acc[NI][MJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    sp_accumulate<MJ, NI, WM, WN, PF, DBG, KS>(a, bm * G::BM, bn * G::BN, a.K / G::SBK, smem, acc);
    sp_finish<MJ, NI, WM, WN, KS>(a, bm * G::BM, bn * G::BN, smem, acc);
}

// ---- operands by LDS-DMA, v_mfma_f32_16x16x32_f16 ---------------------------------------------------------------------------------
// (Round 4 built this structure on v_mfma_f32_32x32x16_f16 — 256 x 256 tiles on eight waves, 128 x 128 on four, 160 x 128 with the K
// range split inside the workgroup — after measuring that the register-staged loop spends 54 % of the LDS's time on ds_write_b128:
// a stage (32 k of the tile's rows, 128 bytes per row) written by `buffer_load ... lds`, no staging registers, no LDS stores, the
// image XOR-swizzled on the DMA's SOURCE address and on the fragment read, two stage buffers.  Round 5 measured every one of those
// forms behind the 16x16x32 forms below on every shape (profiles/r05_mb_linear_sp16_forms.txt) and removed them.)
// MI355X_MICROARCH.md, DVFS give-back item 7: at about equal cycles per FLOP the chip holds a higher clock on the 16x16x32 shape
// than on 32x32x16 (1.12-1.15x the FLOP/s in bare loops on random data).  Same planes, same stage image (32 k of a row = one
// 128-byte line, 1-KiB DMA pieces of 8 rows), but a lane now holds row l & 15 and the k group l >> 4 of a 16-row block — chunk
// 2 (l >> 4) (hi) / + 1 (lo) of its row: ONE k32 step per stage.  A ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) is then
// rows {0-3, 12-15} at chunk c and rows {4-11} at chunk c ^ 2, so the swizzle key of the 32-row kernel, (r >> 1) & 7, would put
// rows 4-11 on the slots of rows 0-3 / 12-15 (two-way conflicts); the key here flips bit 1 for rows 4-11 of every 16:
// key16(r) = ((r >> 1) & 7) ^ ((((r >> 2) ^ (r >> 3)) & 1) << 1) — conflict-free for this read (checked exhaustively).
// The MFMA runs transposed like the others (A = W rows, B = X rows): a lane of a 16 x 16 result holds row m = l & 15 and the
// four consecutive columns n = 4 (l >> 4) .. + 3.
// NSLOT = 2: both fragment sets of a stage pair in registers (wave tile 64 x 64: 64 + 2 x 64 registers), the reads of stage
// it + 1 run under the MFMAs of stage it.  NSLOT = 1 (wave tile 128 x 64: 128 accumulator + 96 fragment registers, no room for
// a second set): the fragments are reloaded PROGRESSIVELY — the barrier sits in the middle of a stage; behind it the registers
// of every m block go back to LDS for the next stage as soon as the block's last MFMA is issued, the W fragments one by one
// inside the last m block (each has >= 9 MFMAs = 144 cycles before the next stage wants it).
__device__ __forceinline__ int sp_key16(int r) { return ((r >> 1) & 7) ^ ((((r >> 2) ^ (r >> 3)) & 1) << 1); }

// Epilogue of the 16 x 16 blocks.  acc[i][j][e] = element (m, n): m = m0 + wm0 + 16 j + (l & 15), n = n0 + wn0 + 16 i + 4 (l >> 4) + e.
// Interior tiles go THROUGH LDS: straight from the accumulators a store instruction would write 16 rows x 64 bytes (the 32-row
// kernels: 32 rows x 32 bytes) — a row-per-lane store tail that is issue-bound (cdna_hip_programming.md T21; the K loop was
// HALF of a launch's duration with it, profiles/r05_mb_linear_sp16_dbg.txt).  Each wave owns an 8-KiB LDS region and moves its
// tile through it 32 rows at a time: scaled / biased / activated values in (16 rows x 16 bytes per ds_write_b128, the
// 16-byte slot c of row r at c ^ (r & 15): conflict-free both ways), whole rows out — a wave-instruction then reads, adds the
// residual to and stores 4 rows x 256 contiguous bytes (fp32) and 4 x 256 bytes of planes.  LDS operations of one wave execute
// in order and nobody else touches the region, so one workgroup barrier (everybody is out of the stage buffers) is all.
template <int MJ, int NI, int WM, int WN>
__device__ __forceinline__ void sp_finish16(const SpArgs& a, int m0, int n0, unsigned char* smem, const float* side, v4f (&acc)[NI][MJ]) {
    constexpr int BM = 16 * MJ * WM, BN = 16 * NI * WN;
    static_assert(NI == 4 || NI == 2, "the LDS pass moves 32 rows x 64 or 32 columns per wave and step");
    constexpr int SLOTS = 4 * NI, ROWBYTES = 16 * SLOTS, RPI = 64 / SLOTS;      // 16-byte slots per row, rows per wave-instruction on the way out
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int wm0 = (wave / WN) * (16 * MJ), wn0 = (wave % WN) * (16 * NI);
    const float* __restrict__ bias = a.bias;
    const float* __restrict__ res = a.res;
    const float* __restrict__ ws = a.ws;
    float* __restrict__ Y = a.Y;
    uint32_t* __restrict__ P = a.P;
    const int M = a.M, N = a.N;
    const int64_t ldr = a.ldr, ldy = a.ldy, ldp = a.ldp;
    const bool vec_ok = (N & 3) == 0 && (Y == nullptr || ((ldy & 3) == 0 && sp_al16(Y))) &&
                        (res == nullptr || ((ldr & 3) == 0 && sp_al16(res))) && (bias == nullptr || sp_al16(bias)) && sp_al16(ws);
    const bool interior = m0 + BM <= M && n0 + BN <= N && vec_ok;
    __syncthreads();                                   // uniform: every wave has left the stage buffers
    auto epilogue = [&](auto actfn) __attribute__((always_inline)) {
        if (interior) {
            unsigned char* region = smem + wave * (32 * ROWBYTES);
            v4f w4[NI], b4[NI];         // from the side area the prologue filled: [xs BM | ps BM | ws BN | bias BN]
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                w4[i] = *reinterpret_cast<const v4f*>(side + 2 * BM + wn0 + 16 * i + 4 * l4);
                b4[i] = *reinterpret_cast<const v4f*>(side + 2 * BM + BN + wn0 + 16 * i + 4 * l4);
            }
            const int oslot = lane % SLOTS, orow = lane / SLOTS;      // on the way OUT: this lane's 16-byte slot and row inside a wave-instruction
            const int ncol = n0 + wn0 + 4 * oslot;
#pragma unroll
            for (int c = 0; c < (MJ + 1) / 2; ++c) {
                constexpr int KEY = SLOTS - 1;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * c + jj;
                    if (j >= MJ) continue;
                    const float sx = side[wm0 + 16 * j + l15];
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        v4f v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = actfn(acc[i][j][e] * (sx * w4[i][e]) + b4[i][e]);
                        *reinterpret_cast<v4f*>(region + (16 * jj + l15) * ROWBYTES + 16 * ((4 * i + l4) ^ (l15 & KEY))) = v;
                    }
                }
                const int rows = (2 * c + 1 < MJ) ? 32 : 16;
#pragma unroll
                for (int t = 0; t < 32 / RPI; ++t) {
                    if (t * RPI >= rows) continue;
                    const int row = RPI * t + orow, m = m0 + wm0 + 32 * c + row;
                    v4f v = *reinterpret_cast<const v4f*>(region + row * ROWBYTES + 16 * (oslot ^ (row & KEY)));
                    if (res != nullptr) {
                        const v4f r = *reinterpret_cast<const v4f*>(res + (int64_t)m * ldr + ncol);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += r[e];
                    }
                    if (Y != nullptr) *reinterpret_cast<v4f*>(Y + (int64_t)m * ldy + ncol) = v;
                    if (P != nullptr) sp_store4(P + (int64_t)m * ldp, ncol, v[0], v[1], v[2], v[3], side[BM + wm0 + 32 * c + row]);
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int m = m0 + wm0 + 16 * j + l15;
            const bool m_ok = m < M;
            const float sx = a.xs[min(m, M - 1)];
            const float sp = a.ps != nullptr ? a.ps[min(m, M - 1)] : 1.f;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + wn0 + 16 * i + 4 * l4 + e;
                    if (m_ok && n < N) {
                        float v = actfn(acc[i][j][e] * (sx * ws[n]) + (bias != nullptr ? bias[n] : 0.f));
                        if (res != nullptr) v += res[(int64_t)m * ldr + n];
                        if (Y != nullptr) Y[(int64_t)m * ldy + n] = v;
                        if (P != nullptr) {
                            const float t = v * sp;
                            const _Float16 h = (_Float16)t, l = (_Float16)(t - (float)h);
                            _Float16* dst = reinterpret_cast<_Float16*>(reinterpret_cast<unsigned char*>(P + (int64_t)m * ldp) +
                                                                       (n >> 3) * 32) + (n & 7);
                            dst[0] = h;
                            dst[8] = l;
                        }
                    }
                }
        }
    };
    if (a.act == SP_ACT_QUICK_GELU) epilogue([](float x) { return x / (1.0f + __expf(-1.702f * x)); });
    else if (a.act == SP_ACT_GELU_ERF) epilogue([](float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); });
    else epilogue([](float x) { return x; });
}

// The K loop of a tile on v_mfma_f32_16x16x32_f16 with its operands by LDS-DMA: `Xt` / `Wt` = the tile's first row of either
// operand's planes, `mrows` / `nrows` = how many of its rows exist (>= 1; the others re-read the last one and are never stored), T
// stages of 32 k from the pointers on.  `pre.load()` runs between the first stages' DMA and the wait for them, `pre.store()` behind
// that wait and in front of the first barrier (the projection's epilogue vectors go to LDS there; SpNoPre: nothing).
// DBG (timing / diagnosis only): 1 = no DMA inside the loop (results wrong), 2 = shader clock and 100 MHz clock at the loop's start
// into clk[0..1] (results right), 3 = MFMAs only: no DMA and no fragment reads inside the loop (results wrong)
struct SpNoPre {
    __device__ __forceinline__ void load() {}
    __device__ __forceinline__ void store() {}
};

template <int MJ, int NI, int WM, int WN, int NSLOT, int DBG, bool ILV, class Pre>
__device__ __forceinline__ void sp_accumulate16(const uint32_t* Xt, int64_t ldx, int mrows, const uint32_t* Wt, int64_t ldw, int nrows, int T,
                                                unsigned char* smem, v4f (&acc)[NI][MJ], Pre& pre, long long (&clk)[2]) {
    constexpr int NWV = WM * WN, BM = 16 * MJ * WM, BN = 16 * NI * WN;
    constexpr int ROWB = 128, RPP = 8, CPR = 8;
    constexpr int STAGE = (BM + BN) * ROWB;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "whole pieces per operand");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int wm0 = (wave / WN) * (16 * MJ), wn0 = (wave % WN) * (16 * NI);

    // DMA through buffer resources (cdna_hip_programming.md T8 / T20): one descriptor per operand — base = the tile's first row,
    // made from readfirstlane'd halves so that the compiler keeps it in SGPRs — a loop-invariant 32-bit byte offset per lane and
    // piece, and the stage's advance (128 bytes per stage) as the instruction's scalar offset: no vector address arithmetic inside
    // the loop.  Wave w stages the X pieces w, w + NWV, ... and the W pieces w, w + NWV, ... of the image (X rows first); a
    // tile whose piece count is no multiple of the wave count leaves the last slot of some waves empty (a wave-uniform test).
    auto make_rsrc = [](const void* ptr) __attribute__((always_inline)) {
        const uint64_t v = reinterpret_cast<uint64_t>(ptr);
        const uint64_t lo = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
        const uint64_t hi = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(lo | (hi << 32)), 0, 0x7fffffff, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(Xt);
    const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(Wt);
    constexpr int NPX = BM / RPP, NPW = BN / RPP, PXW = (NPX + NWV - 1) / NWV, PWW = (NPW + NWV - 1) / NWV;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned off_x[PXW], off_w[PWW];
#pragma unroll
    for (int s = 0; s < PXW; ++s) {
        const int r = RPP * min(wave + NWV * s, NPX - 1) + lane / CPR;
        off_x[s] = (unsigned)(min(r, mrows - 1) * (int)(4 * ldx) + 16 * ((lane % CPR) ^ sp_key16(r)));      // rows past the end: a valid row, never stored
    }
#pragma unroll
    for (int s = 0; s < PWW; ++s) {
        const int r = RPP * min(wave + NWV * s, NPW - 1) + lane / CPR;
        off_w[s] = (unsigned)(min(r, nrows - 1) * (int)(4 * ldw) + 16 * ((lane % CPR) ^ sp_key16(r)));
    }
    auto issue = [&](int it, unsigned char* stage) __attribute__((always_inline)) {
        // ((int) casts: an argument of template-dependent type makes hipcc's HOST pass drop the kernel's instantiation without a
        // diagnostic — the launch stub stays an undefined symbol of the library)
#pragma unroll
        for (int s = 0; s < PXW; ++s)
            if (NPX % NWV == 0 || s + 1 < PXW || wave_u + NWV * s < NPX)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(stage + (wave_u + NWV * s) * 1024), 16,
                                                         (int)off_x[s], (int)(it * ROWB), 0, 0);
#pragma unroll
        for (int s = 0; s < PWW; ++s)
            if (NPW % NWV == 0 || s + 1 < PWW || wave_u + NWV * s < NPW)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(stage + (NPX + wave_u + NWV * s) * 1024), 16,
                                                         (int)off_w[s], (int)(it * ROWB), 0, 0);
    };
    const int ch = 16 * ((2 * l4) ^ sp_key16(l15));                        // hi chunk; lo = ch ^ 16
    const int fx = (wm0 + l15) * ROWB, fw = (BM + wn0 + l15) * ROWB;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    v8h xh[NSLOT][MJ], xl[NSLOT][MJ], wh[NSLOT][NI], wl[NSLOT][NI];
    auto fread_x = [&](const unsigned char* stage, int j, auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        xh[S][j] = *reinterpret_cast<const v8h*>(stage + fx + j * 16 * ROWB + ch);
        xl[S][j] = *reinterpret_cast<const v8h*>(stage + fx + j * 16 * ROWB + (ch ^ 16));
    };
    auto fread_w = [&](const unsigned char* stage, int i, auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        wh[S][i] = *reinterpret_cast<const v8h*>(stage + fw + i * 16 * ROWB + ch);
        wl[S][i] = *reinterpret_cast<const v8h*>(stage + fw + i * 16 * ROWB + (ch ^ 16));
    };
    auto fread = [&](const unsigned char* stage, auto sc) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) fread_x(stage, j, sc);
#pragma unroll
        for (int i = 0; i < NI; ++i) fread_w(stage, i, sc);
    };
    // the two small products first, then hi x hi (as in the 32-row kernels)
    auto mfma3 = [&](int i, int j, auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[S][i], xh[S][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[S][i], xl[S][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[S][i], xh[S][j], acc[i][j], 0, 0, 0);
    };
    auto mfmas = [&](auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? wl[S][i] : wh[S][i], p == 1 ? xl[S][j] : xh[S][j],
                                                                       acc[i][j], 0, 0, 0);
    };

    issue(0, smem);
    if (T > 1) issue(1, smem + STAGE);
    pre.load();
    __builtin_amdgcn_s_waitcnt(0x0f70);                               // vmcnt(0): the stages (waves stage different numbers of pieces) and pre's loads
    pre.store();
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);                               // the prologue's scalar loads (see the 32-row kernel)
    __builtin_amdgcn_sched_barrier(0);
    fread(smem, SpIC<0>{});
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG == 2) {
        clk[0] = (long long)__builtin_amdgcn_s_memtime();
        clk[1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    if constexpr (NSLOT == 2) {
        // stage `it` is in registers (slot it & 1).  Barrier: every wave's share of stage it + 1 has landed and every wave is
        // done reading stage it (its buffer takes the DMA of stage it + 2); the reads of stage it + 1 run under the MFMAs.
        auto body = [&](int it, auto sc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value;
            unsigned char* cur = smem + (it & 1) * STAGE;
            unsigned char* oth = smem + ((it + 1) & 1) * STAGE;
            __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (DBG != 1 && DBG != 3 && it + 2 < T) issue(it + 2, cur);
            if (DBG != 3 && it + 1 < T) fread(oth, SpIC<S ^ 1>{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(sc);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
        };
        // the same step for it + 2 < T, as ONE basic block whose order is prescribed: the DMA of stage it + 2 and the fragment
        // reads of stage it + 1 go out between the MFMAs of stage it (two MFMAs, one ds_read_b128, and after each of the first
        // ones one buffer_load ... lds) instead of in front of them.  Two resident workgroups hide each other's issue gaps
        // either way (33.6 cycles per MFMA and workgroup = 95 % of the pipe); a workgroup that has its compute unit to itself —
        // the tail of every round of tiles, launches of about one tile per compute unit — ran at 30 cycles per MFMA, the matrix
        // pipe idle while its one wave per SIMD issued 16 ds_reads and 8 DMAs (profiles/r05_mb_linear_sp16_rounds.txt).
        auto body_full = [&](int it, auto sc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value;
            constexpr int NREAD = 2 * (MJ + NI), NMFMA = 3 * MJ * NI, MPG = (2 * MJ * NI) / NREAD < 1 ? 1 : (2 * MJ * NI) / NREAD;
            static_assert(NMFMA >= PXW + PWW + MPG * NREAD, "MFMAs to hide the issue under");
            constexpr int NDMA = PXW + PWW;
            unsigned char* cur = smem + (it & 1) * STAGE;
            unsigned char* oth = smem + ((it + 1) & 1) * STAGE;
            __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (DBG != 1 && DBG != 3) issue(it + 2, cur);
            if (DBG != 3) fread(oth, SpIC<S ^ 1>{});
            mfmas(sc);
            // (the compiler orders the fragment reads behind the DMAs — both touch LDS —, so the DMAs go first)
#pragma unroll
            for (int g = 0; g < NDMA; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);            // VMEM read
            }
#pragma unroll
            for (int g = 0; g < NREAD; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPG, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NDMA - MPG * NREAD, 0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
        };
        int it = 0;
        if constexpr (ILV) {
            for (; it + 3 < T; it += 2) {
                body_full(it, SpIC<0>{});
                body_full(it + 1, SpIC<1>{});
            }
        }
        for (; it + 1 < T; it += 2) {
            body(it, SpIC<0>{});
            body(it + 1, SpIC<1>{});
        }
        if (it < T) body(it, SpIC<0>{});
    } else {
        for (int it = 0; it < T; ++it) {
            unsigned char* cur = smem + (it & 1) * STAGE;
            unsigned char* oth = smem + ((it + 1) & 1) * STAGE;
            const bool more = DBG != 3 && it + 1 < T;
#pragma unroll
            for (int j = 0; j < MJ / 2; ++j)
#pragma unroll
                for (int i = 0; i < NI; ++i) mfma3(i, j, SpIC<0>{});
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (DBG != 1 && DBG != 3 && it + 2 < T) issue(it + 2, cur);
            if (more) {
#pragma unroll
                for (int j = 0; j < MJ / 2; ++j) fread_x(oth, j, SpIC<0>{});
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = MJ / 2; j < MJ - 1; ++j) {
#pragma unroll
                for (int i = 0; i < NI; ++i) mfma3(i, j, SpIC<0>{});
                __builtin_amdgcn_sched_barrier(0);
                if (more) fread_x(oth, j, SpIC<0>{});
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                mfma3(i, MJ - 1, SpIC<0>{});
                __builtin_amdgcn_sched_barrier(0);
                if (more) fread_w(oth, i, SpIC<0>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) fread_x(oth, MJ - 1, SpIC<0>{});
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// the projection's epilogue vectors — 2^-e of the tile's X rows, the caller's plane scale of those rows, 2^-e and bias of its W rows
// — fetched underneath the first stages' DMA into an LDS side area [xs BM | ps BM | ws BN | bias BN], so that the epilogue starts
// from LDS instead of from three dependent global loads (a workgroup's epilogue was 4.4 us of its 27.7,
// profiles/r05_mb_linear_sp16_epi.txt)
template <int BM, int BN, int NT>
struct SpSidePre {
    static constexpr int NSV = (BM + BN + NT - 1) / NT;
    const SpArgs& a;
    int m0, n0;
    float* side;
    float sv0[NSV], sv1[NSV];
    __device__ __forceinline__ void load() {
        const int tid = threadIdx.x;
#pragma unroll
        for (int q = 0; q < NSV; ++q) {
            const int e = tid + q * NT;
            sv0[q] = sv1[q] = 0.f;
            if (e < BM) {
                sv0[q] = a.xs[min(m0 + e, a.M - 1)];
                sv1[q] = a.ps != nullptr ? a.ps[min(m0 + e, a.M - 1)] : 1.f;
            } else if (e < BM + BN) {
                sv0[q] = a.ws[min(n0 + e - BM, a.N - 1)];
                sv1[q] = a.bias != nullptr ? a.bias[min(n0 + e - BM, a.N - 1)] : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store() {
        const int tid = threadIdx.x;
#pragma unroll
        for (int q = 0; q < NSV; ++q) {
            const int e = tid + q * NT;
            if (e < BM) side[e] = sv0[q], side[BM + e] = sv1[q];
            else if (e < BM + BN) side[2 * BM + e - BM] = sv0[q], side[2 * BM + BN + e - BM] = sv1[q];
        }
    }
};

template <int MJ, int NI, int WM, int WN, int NSLOT, int DBG, int WPE = 2, bool ILV = (NSLOT == 2)>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void linear_sp16_dma16_kernel(SpArgs a) {
    constexpr int NWV = WM * WN, BM = 16 * MJ * WM, BN = 16 * NI * WN;
    constexpr int STAGE = (BM + BN) * 128;
    static_assert(2 * STAGE >= NWV * 2048 * NI, "the epilogue's LDS regions");
    constexpr int SIDE = 4 * (2 * BM + 2 * BN);
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE + SIDE];
    const int per = (a.tiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= a.tiles || (int)(blockIdx.x >> 3) >= per) return;
    int bm, bn;
    sp_tile_of(a, tile, bm, bn);
    long long t_begin = 0;
    if constexpr (DBG == 2) t_begin = (long long)__builtin_amdgcn_s_memrealtime();
    const int m0 = bm * BM, n0 = bn * BN;
    v4f acc[NI][MJ];
    long long clk[2] = {0, 0};
    SpSidePre<BM, BN, 64 * NWV> pre{a, m0, n0, reinterpret_cast<float*>(smem + 2 * STAGE)};
    sp_accumulate16<MJ, NI, WM, WN, NSLOT, DBG, ILV>(a.X + (int64_t)m0 * a.ldx, a.ldx, a.M - m0, a.W + (int64_t)n0 * a.ldw, a.ldw, a.N - n0,
                                                     a.K / SPK, smem, acc, pre, clk);
    if constexpr (DBG == 2) {
        const long long e_clk = (long long)__builtin_amdgcn_s_memtime(), e_real = (long long)__builtin_amdgcn_s_memrealtime();
        if (a.stamps != nullptr && threadIdx.x == 0) {
            long long* st = a.stamps + 8 * (int64_t)blockIdx.x;
            st[0] = clk[0]; st[1] = e_clk; st[2] = clk[1]; st[3] = e_real; st[4] = t_begin;
        }
    }
    sp_finish16<MJ, NI, WM, WN>(a, m0, n0, smem, reinterpret_cast<const float*>(smem + 2 * STAGE), acc);
    if constexpr (DBG == 2) {
        if (a.stamps != nullptr && threadIdx.x == 0) a.stamps[8 * (int64_t)blockIdx.x + 5] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

// ---- fp32 rows -> planes ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sp_split8(const v4f& a, const v4f& b, float s, v4u& hi, v4u& lo) {
    v8h h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float t = (e < 4 ? a[e] : b[e - 4]) * s;
        h[e] = (_Float16)t;
        l[e] = (_Float16)(t - (float)h[e]);
    }
    hi = *reinterpret_cast<v4u*>(&h);
    lo = *reinterpret_cast<v4u*>(&l);
}

// one wave per row, four rows per workgroup; two passes over the row (the second one hits L2)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ X, int64_t ldx, int rows, int K,
                                                          uint32_t* __restrict__ P, int64_t ldp, float* __restrict__ inv_scale,
                                                          float* __restrict__ max_row_norm) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = X + row * ldx;
    const int ng = K / 8;
    float amax = 0.f, ss = 0.f;
    for (int g = lane; g < ng; g += 64) {
        const v4f a = *reinterpret_cast<const v4f*>(x + 8 * g), b = *reinterpret_cast<const v4f*>(x + 8 * g + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            amax = fmaxf(amax, fmaxf(fabsf(a[e]), fabsf(b[e])));
            ss += a[e] * a[e] + b[e] * b[e];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        ss += __shfl_xor(ss, o, 64);
    }
    // the largest Euclidean row norm (rounded up a little): with it a consumer bounds |x . w_j| <= |x| max_j |w_j| for every
    // output column at once (the scale of a projection's split-fp16 OUTPUT, emcid_add_layernorm_sp16).  Non-negative floats
    // order like their bit patterns.
    if (max_row_norm != nullptr && lane == 0)
        atomicMax(reinterpret_cast<unsigned int*>(max_row_norm), __float_as_uint(sqrtf(ss) * 1.0001f));
    float s, inv;
    sp_scale_of(amax, s, inv);
    if (lane == 0) inv_scale[row] = inv;
    uint32_t* p = P + row * ldp;
    for (int g = lane; g < ng; g += 64) {
        const v4f a = *reinterpret_cast<const v4f*>(x + 8 * g), b = *reinterpret_cast<const v4f*>(x + 8 * g + 4);
        v4u hi, lo;
        sp_split8(a, b, s, hi, lo);
        *reinterpret_cast<v4u*>(p + 8 * g) = hi;
        *reinterpret_cast<v4u*>(p + 8 * g + 4) = lo;
    }
}

// ---- Stage-0 Gram on the same matrix path:  lower(G) += X^T X  for a chunk of token rows X [t][d] ---------------------------------
// (reference: util/runningstats.py:469-511, mom2 += a.t().mm(a) over the caption tokens.)  The contraction runs over the TOKENS,
// so the operand is X^T [d][t] and the scale is per COLUMN of X (per feature, over the chunk): gram_colmax_kernel finds the
// column maxima, gram_transpose_split_kernel writes X^T as split planes (rows = features, K = tokens, zero-padded to a multiple
// of 32), and gram_sp16_kernel is the projection kernel's K loop (sp_accumulate16) over the LOWER 128 x 128 tiles with the token range of every
// tile cut into `ks` parts (300 lower tiles at d = 3072 do not fill 512 workgroup slots): a part's tile is scaled back by
// 2^-(e_i + e_j) and added into G with fp32 atomics — G is an accumulator, like the exact-f32 SYRK's multi-slab mode.
__global__ __launch_bounds__(256) void gram_colmax_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ rw, int t,
                                                           int d, int rows_per_wg, unsigned* __restrict__ cmax) {
    const int c4 = blockIdx.x * 256 + threadIdx.x;                 // float4 column group
    if (4 * c4 >= d) return;
    const int r0 = blockIdx.y * rows_per_wg, r1 = min(t, r0 + rows_per_wg);
    v4f m = {0.f, 0.f, 0.f, 0.f};
    int r = r0;
    for (; r + 8 <= r1; r += 8) {                                  // eight rows' loads in flight per thread (round 5: one)
        v4f x[8];
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = *reinterpret_cast<const v4f*>(X + (int64_t)(r + u) * ldx + 4 * c4);
            w[u] = rw != nullptr ? rw[r + u] : 1.f;                // (a uniform load per row)
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], fabsf(x[u][e] * w[u]));
    }
    for (; r < r1; ++r) {
        const v4f x = *reinterpret_cast<const v4f*>(X + (int64_t)r * ldx + 4 * c4);
        const float w = rw != nullptr ? rw[r] : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], fabsf(x[e] * w));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicMax(cmax + 4 * c4 + e, __float_as_uint(m[e]));      // non-negative floats order like their bits
}

// one workgroup: 64 tokens x 64 features through LDS; thread -> (feature, group of 8 tokens): 32 contiguous bytes of planes
__global__ __launch_bounds__(256) void gram_transpose_split_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ rw,
                                                                    int t, int d, const unsigned* __restrict__ cmax,
                                                                    uint32_t* __restrict__ P, int64_t ldp, float* __restrict__ inv_scale) {
    __shared__ float tile[64][65];
    const int t0 = blockIdx.x * 64, d0 = blockIdx.y * 64, tid = threadIdx.x;
    for (int v = tid; v < 64 * 16; v += 256) {                    // 64 rows x 16 float4
        const int r = v >> 4, c = (v & 15) * 4;
        v4f x = {0.f, 0.f, 0.f, 0.f};
        float w = 1.f;
        if (t0 + r < t && d0 + c < d) {
            x = *reinterpret_cast<const v4f*>(X + (int64_t)(t0 + r) * ldx + d0 + c);
            if (rw != nullptr) w = rw[t0 + r];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[r][c + e] = x[e] * w;    // fl32(x w): the fp32 product the caller would have formed
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int w = tid + 256 * u, f = w >> 3, g = w & 7;       // feature 0..63, token group 0..7
        if (d0 + f >= d) continue;
        float s, inv;
        sp_scale_of(__uint_as_float(cmax[d0 + f]), s, inv);
        if (blockIdx.x == 0 && g == 0) inv_scale[d0 + f] = inv;
        v4f a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = tile[8 * g + e][f], b[e] = tile[8 * g + 4 + e][f];
        v4u hi, lo;
        sp_split8(a, b, s, hi, lo);
        uint32_t* dst = P + (int64_t)(d0 + f) * ldp + t0 + 8 * g;
        *reinterpret_cast<v4u*>(dst) = hi;
        *reinterpret_cast<v4u*>(dst + 4) = lo;
    }
}

struct GramSpArgs {
    const uint32_t* XT; int64_t ld; const float* inv; float* G; int64_t ldg; int d, K, tiles_side, ks;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gram_sp16_kernel(GramSpArgs g) {
    // the projection's 128 x 128 form (two waves by two, 64 x 64 each, v_mfma_f32_16x16x32_f16, operands by LDS-DMA)
    constexpr int MJ = 4, NI = 4, WM = 2, WN = 2, BT = 128;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (2 * BT) * 128];
    const int n_lower = g.tiles_side * (g.tiles_side + 1) / 2, items = n_lower * g.ks, per = (items + 7) / 8;
    const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= items || (int)(blockIdx.x >> 3) >= per) return;
    const int tile = item / g.ks, part = item - tile * g.ks;
    int bm = (int)((sqrtf(8.f * (float)tile + 1.f) - 1.f) * 0.5f);          // row of the lower triangle holding `tile`
    while (bm * (bm + 1) / 2 > tile) --bm;
    while ((bm + 1) * (bm + 2) / 2 <= tile) ++bm;
    const int bn = tile - bm * (bm + 1) / 2;
    const int T = g.K / SPK, it_lo = (int)((int64_t)part * T / g.ks), it_hi = (int)((int64_t)(part + 1) * T / g.ks);
    if (it_hi <= it_lo) return;
    // the K loop's "X" rows = the features of the tile's COLUMNS (block bn), its "W" rows = those of the tile's rows (block bm):
    // a register of the accumulators then holds, per wave, 16 consecutive columns in each of 4 rows of G — one atomic
    // wave-instruction = four 64-byte runs, the size an atomic request leaves the L2 with (MI355X_MICROARCH.md, global float
    // atomics; 64 lanes in 64 rows would be ~17x slower)
    const int c0 = bn * BT, r0 = bm * BT;
    const uint32_t* base = g.XT + (int64_t)it_lo * SPK;
    v4f acc[NI][MJ];
    long long clk[2];
    SpNoPre pre;
    sp_accumulate16<MJ, NI, WM, WN, 2, 0, true>(base + (int64_t)c0 * g.ld, g.ld, g.d - c0, base + (int64_t)r0 * g.ld, g.ld, g.d - r0,
                                                it_hi - it_lo, smem, acc, pre, clk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int wm0 = (wave / WN) * (16 * MJ), wn0 = (wave % WN) * (16 * NI);
    float rinv[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) rinv[i][e] = g.inv[min(r0 + wn0 + 16 * i + 4 * l4 + e, g.d - 1)];
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        const int col = c0 + wm0 + 16 * j + l15;
        const float sc = g.inv[min(col, g.d - 1)];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = r0 + wn0 + 16 * i + 4 * l4 + e;
                if (row < g.d && col < g.d) unsafeAtomicAdd(g.G + (int64_t)row * g.ldg + col, acc[i][j][e] * (sc * rinv[i][e]));
            }
    }
}

struct SpCfg { int bm, bn; };
static const SpCfg kSpCfgs[] = {{128, 128}, {80, 128}, {64, 64}, {160, 128}, {64, 64}};
inline long long* g_sp16_stamps = nullptr;      // set by emcid_debug_linear_sp16_stamps

}  // namespace emcid

using namespace emcid;

extern "C" {

int emcid_split_rows_f16(const float* X, int64_t ldx, int64_t rows, int64_t K, void* planes, int64_t ldp, float* inv_scale,
                         float* max_row_norm, void* stream) {
    EMCID_CHECK_ARG(X && planes && inv_scale && rows > 0 && K > 0 && K % 8 == 0 && ldx >= K && ldp >= K);
    EMCID_CHECK_ARG(ldx % 4 == 0 && ldp % 4 == 0 && aligned16(X) && aligned16(planes) && rows < (1LL << 31) && K < (1 << 24));
    ScopedProf sp(KC_MISC, (hipStream_t)stream);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, X, ldx, (int)rows,
                       (int)K, (uint32_t*)planes, ldp, inv_scale, max_row_norm);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int64_t emcid_gram_sp16_workspace_bytes_for(int64_t d, int64_t t) {
    // planes of one chunk of X^T [d][min(t, 32768) rounded up to 64] + the chunk's column maxima and inverse scales
    const int64_t tc = round_up(t < 1 ? 1 : (t > 32768 ? 32768 : t), 64);
    return round_up(d, 64) * tc * 4 + 2 * round_up(d, 64) * 4 + 256;
}

int64_t emcid_gram_sp16_workspace_bytes(int64_t d) { return emcid_gram_sp16_workspace_bytes_for(d, 32768); }

int emcid_gram_accumulate_sp16_f32(const float* X, const float* row_weight, int64_t t, int64_t d, int64_t ldx, float* G, int64_t ldg,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    EMCID_CHECK_ARG(X && G && workspace && t >= 0 && d > 0 && ldx >= d && ldg >= d && d % 4 == 0 && ldx % 4 == 0 && aligned16(X));
    EMCID_CHECK_ARG(workspace_bytes >= emcid_gram_sp16_workspace_bytes_for(d, t) && aligned16(workspace) && d < (1 << 20) && t < (1LL << 31));
    if (t == 0) return EMCID_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t dr = round_up(d, 64);
    uint32_t* P = (uint32_t*)workspace;
    const int64_t chunk_cap = round_up(std::min<int64_t>(t, 32768), 64);       // the workspace's plane area holds one chunk
    unsigned* cmax = (unsigned*)((char*)workspace + dr * chunk_cap * 4);
    float* inv = (float*)(cmax + dr);
    const int side = (int)((d + 127) / 128), n_lower = side * (side + 1) / 2;
    ScopedProf sp(KC_GRAM, st);
    for (int64_t c0 = 0; c0 < t; c0 += 32768) {
        const int tc = (int)std::min<int64_t>(32768, t - c0), tpad = (int)round_up(tc, 64);
        const float* Xc = X + c0 * ldx;
        const float* rwc = row_weight != nullptr ? row_weight + c0 : nullptr;
        if (hipMemsetAsync(cmax, 0, dr * sizeof(unsigned), st) != hipSuccess) return fail(EMCID_ERR_HIP, __func__, "hipMemsetAsync");
        const int rows_per_wg = 64;          // 1 536 workgroups per chunk of 32 768 tokens at d = 3072 (round 5: 256 rows, 384 workgroups: 2.9 TB/s)
        hipLaunchKernelGGL(gram_colmax_kernel, dim3((unsigned)((d / 4 + 255) / 256), (unsigned)((tc + rows_per_wg - 1) / rows_per_wg)),
                           dim3(256), 0, st, Xc, ldx, rwc, tc, (int)d, rows_per_wg, cmax);
        hipLaunchKernelGGL(gram_transpose_split_kernel, dim3((unsigned)(tpad / 64), (unsigned)(dr / 64)), dim3(256), 0, st, Xc, ldx, rwc,
                           tc, (int)d, cmax, P, (int64_t)tpad, inv);
        // parts per tile: enough work items for ~3 rounds of the 512 workgroup slots, a last round as full as possible
        const int T = tpad / SPK;
        int ks = 1;
        double best = 0.0;
        for (int k = 1; k <= 16 && T / k >= 16; ++k) {
            const int items = n_lower * k;
            const double fill = (double)items / (512.0 * ((items + 511) / 512));
            const double score = fill - 0.01 * k + (items >= 1024 ? 0.0 : -0.5);
            if (score > best) best = score, ks = k;
        }
        GramSpArgs ga{P, (int64_t)tpad, inv, G, ldg, (int)d, tpad, side, ks};
        const int items = n_lower * ks;
        hipLaunchKernelGGL(gram_sp16_kernel, dim3((unsigned)(((items + 7) / 8) * 8)), dim3(256), 0, st, ga);
    }
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

/* Diagnostic: until called again with NULL, the LDS-DMA projection launches (cfg 64 / 128 / 256 / 320) run their stamped build:
 * per workgroup {shader clock at loop start, at loop end, 100 MHz clock at loop start, at loop end, 100 MHz clock at kernel entry,
 * at kernel exit, -, -} at stamps_dev[8 * blockIdx.x] (in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz,
 * MI355X_MICROARCH.md DVFS item 6; prologue / K loop / epilogue of every workgroup in 10-ns ticks). */
int emcid_debug_linear_sp16_stamps(long long* stamps_dev) {
    g_sp16_stamps = stamps_dev;
    return EMCID_OK;
}

int emcid_linear_sp16_f32(const void* Xp, int64_t ldx, const float* x_inv_scale, const void* Wp, int64_t ldw,
                          const float* w_inv_scale, const float* bias, const float* residual, int64_t ldr, float* Y, int64_t ldy,
                          void* Yp, int64_t ldp, const float* y_scale, int64_t M, int64_t N, int64_t K, int act, int cfg,
                          void* stream) {
    EMCID_CHECK_ARG(Xp && Wp && x_inv_scale && w_inv_scale && (Y || Yp) && M > 0 && N > 0 && K > 0 && ldx >= K && ldw >= K);
    EMCID_CHECK_ARG(K % SPK == 0 && ldx % 4 == 0 && ldw % 4 == 0 && aligned16(Xp) && aligned16(Wp));
    EMCID_CHECK_ARG(M < (1 << 24) && N < (1 << 24) && K < (1 << 24) && (residual == nullptr || ldr >= N) && (Y == nullptr || ldy >= N));
    EMCID_CHECK_ARG(Yp == nullptr || (N % 8 == 0 && ldp >= N && ldp % 4 == 0 && aligned16(Yp)));
    EMCID_CHECK_ARG(act >= SP_ACT_NONE && act <= SP_ACT_GELU_ERF && cfg >= -1 && cfg < 64);
    // cfg: -1 auto; 0: 128 x 128 (two waves by two, 64 x 64 each, both fragment sets in registers); 1: 80 x 128 (four waves side
    // by side, 80 x 32 each); 2: 64 x 64 (four workgroups per compute unit); 3: 160 x 128 (80 x 64 per wave, fragments reloaded
    // progressively) — all four: operands by LDS-DMA, v_mfma_f32_16x16x32_f16, epilogue through LDS —; 4: 64 x 64 with
    // register-staged operands on v_mfma_f32_32x32x16_f16, two stages of loads in flight (latency-bound launches); + 16 / + 48
    // with cfg 0: timing-only builds (no DMA inside the loop / MFMAs only; results wrong).
    EMCID_CHECK_ARG(ldx < (1 << 20) && ldw < (1 << 20));      // 32-bit buffer offsets of a tile's rows
    const int dbg = cfg < 0 ? 0 : (cfg >> 4) & 3;
    int tile_sel = cfg < 0 ? -1 : (cfg & 15);
    EMCID_CHECK_ARG(tile_sel <= 4 && (dbg == 0 || tile_sel == 0) && dbg != 2);
    if (tile_sel < 0) {
        // profiles/r05_mb_linear_sp16_forms.txt, r05_mb_linear_sp16_small.txt (six LDS-DMA forms and the register-staged ones on
        // 27 shapes, interleaved rounds in one process).  The largest of 128 x 128, 80 x 128, 64 x 64 that still gives ~400
        // workgroups — two resident workgroups for every compute unit: q | k | v at 6 292 rows 128 x 128 (69 us; 96 on round 4's
        // 256 x 256 tile of 32x32x16 MFMAs), N = 768 at 6 292 rows 80 x 128 (300 tiles -> 474: out 30.7 against 35.1 us for round 4's
        // K-split tile, fc2 82.3 against 97.6), 3 072 x 3 072 -> 768 64 x 64 (51.8 against 60.2 / 68.9); 160 x 128 for operands far
        // beyond the L2 and many rounds of tiles (36 335 rows: 339 against 369 us).  Below one 64 x 64 tile per compute unit a launch
        // is latency-bound and the register-staged loop stays ahead on a long K (640 x 3 072 -> 768: 28 against 45 us).
        const int64_t t128 = ((M + 127) / 128) * ((N + 127) / 128), t80 = ((M + 79) / 80) * ((N + 127) / 128);
        const int64_t t64 = ((M + 63) / 64) * ((N + 63) / 64);
        if (t128 >= 1600 && M >= 16384) tile_sel = 3;
        else if (t128 >= 400) tile_sel = 0;
        else if (t80 >= 400) tile_sel = 1;
        else if (t64 >= 256) tile_sel = 2;
        else tile_sel = 4;
    }
    const int bm = kSpCfgs[tile_sel].bm, bn = kSpCfgs[tile_sel].bn;
    const int tiles_m = (int)((M + bm - 1) / bm), tiles_n = (int)((N + bn - 1) / bn);
    const int tiles = tiles_m * tiles_n;
    const int per = (tiles + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    const int rb = 4;       // row tiles per super-row of the tile order
    const SpArgs a{(const uint32_t*)Xp, ldx, x_inv_scale, (const uint32_t*)Wp, ldw, w_inv_scale, bias, residual, ldr, Y, ldy,
                   (uint32_t*)Yp, ldp, y_scale, (int)M, (int)N, (int)K, act, tiles_n, tiles, rb, g_sp16_stamps};
    ScopedProf sp(KC_LINEAR, st);
#define EMCID_SP_DMA16(MJ_, NI_, WM_, WN_, NS_, WPE_, DBG_)                                                                        \
    hipLaunchKernelGGL((linear_sp16_dma16_kernel<MJ_, NI_, WM_, WN_, NS_, DBG_, WPE_>), dim3((unsigned)(per * 8)), dim3(64 * WM_ * WN_), 0, st, a)
    switch (tile_sel) {
        case 0:
            if (dbg == 1) EMCID_SP_DMA16(4, 4, 2, 2, 2, 2, 1);
            else if (dbg == 3) EMCID_SP_DMA16(4, 4, 2, 2, 2, 2, 3);
            else if (a.stamps) EMCID_SP_DMA16(4, 4, 2, 2, 2, 2, 2);
            else EMCID_SP_DMA16(4, 4, 2, 2, 2, 2, 0);
            break;
        case 1: EMCID_SP_DMA16(5, 2, 1, 4, 2, 2, 0); break;
        case 2: EMCID_SP_DMA16(2, 2, 2, 2, 2, 4, 0); break;      // 32 KB of LDS, 128 registers: four workgroups per compute unit
        case 3: EMCID_SP_DMA16(5, 4, 2, 2, 1, 2, 0); break;      // (two fragment sets would spill: 56 registers)
        default:
            hipLaunchKernelGGL((linear_sp16_kernel<1, 1, 2, 2, 2, 4, 0, 1>), dim3((unsigned)(per * 8)), dim3(256), 0, st, a);
            break;
    }
#undef EMCID_SP_DMA16
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

}  // extern "C"
