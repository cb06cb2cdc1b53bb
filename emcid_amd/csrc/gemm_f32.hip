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
// KS = 2 (the default): EIGHT waves per workgroup, two per SIMD — waves 0-3 contract the first 16 of a 32-deep stage, waves
// 4-7 the second 16, both over the whole tile; at the end waves 4-7 hand their accumulators to waves 0-3 through LDS (a fixed
// order: bit-reproducible).  A launch of 240 tiles puts ONE workgroup on a compute unit; with four waves its SIMDs each run
// one wave and every barrier / wait is an idle matrix pipe (measured: 73 % of the MFMA rate even with no memory traffic at
// all), with eight the partner wave issues meanwhile.  Rows are then fetched 128 B at a time (whole cache lines).
//
// Tiles: waves in a WM x WN grid, each (32 MI) x (32 NJ).  160 x 128 (1 x 4 waves of 160 x 32) is the workhorse: the trie
// forward's row counts are multiples of 256 (6 400 at N = 1000 x 3 templates) and N in {768, 2304, 3072}, where it gives
// 240 / 720 / 960 tiles = 0.94 / 2.81 / 3.75 rounds of the 256 compute units (128 x 128: 300 tiles = 59 % of two rounds).
#include "common.h"

#include <algorithm>

namespace emcid {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int LBK = 16;      // K depth of a stage per wave group (a stage is 16 KS deep)

enum LinearAct : int { ACT_NONE = 0, ACT_QUICK_GELU = 1, ACT_GELU_ERF = 2 };

template <int V> struct IC { static constexpr int value = V; };

struct LinearArgs {
    const float* X; int64_t ldx;
    const float* W; int64_t ldw;
    const float* bias;
    const float* res; int64_t ldr;
    float* Y; int64_t ldy;
    int M, N, K, act, tiles_n, tiles, rb;
};

// Tile order: super-rows of `rb` row tiles, column-major inside a super-row — the ~64 tiles an XCD has in flight then share
// `rb` row tiles of X (kept in its 4 MB L2) and walk the column tiles of W together, instead of re-streaming all of W for
// every row tile (rb = 1: row-major; L2-miss traffic of the 128 x 128 launches 2.8x the algorithmic bytes).
__device__ __forceinline__ void linear_tile_of(const LinearArgs& a, int tile, int& bm, int& bn) {
    const int tiles_m = a.tiles / a.tiles_n, sr = tile / (a.rb * a.tiles_n), rem = tile - sr * a.rb * a.tiles_n;
    const int rows = min(a.rb, tiles_m - sr * a.rb);
    bn = rem / rows;
    bm = sr * a.rb + (rem - bn * rows);
}

template <int MI, int NJ, int WM, int WN, int PF, int DBG, int KS>
struct LinearGeom {
    static constexpr int NT = 256 * KS;                      // threads
    static constexpr int SBK = LBK * KS;                     // K depth of a stage
    static constexpr int LLD = SBK + 4;                      // LDS row stride in floats: 20 / 36, conflict-free for the b128 lane groups
    static constexpr int CPR = SBK / 4;                      // float4 per row and stage
    static constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
    static constexpr int VA = (BM * CPR + NT - 1) / NT, VB = (BN * CPR + NT - 1) / NT;      // float4 per thread and stage
    static constexpr int STAGE = (BM + BN) * LLD;
    static constexpr int RED = KS == 2 ? 4 * MI * NJ * 16 * 64 : 0;                         // floats the final hand-over needs
    static constexpr int SMEM = 2 * STAGE > RED ? 2 * STAGE : RED;
    static constexpr int NBLK = MI * NJ, HB = KS == 2 ? (NBLK + 1) / 2 : NBLK;
};

// The K loop of one tile over the stages [it_lo, it_hi) of its K range: acc += X[m0.., k] W[n0.., k]^T.
// PF: how many stages ahead the global loads run (= register sets of staged data).  DBG (timing experiments only, results
// wrong): 1 = no global loads inside the loop, 2 = also no LDS stores.
template <int MI, int NJ, int WM, int WN, int PF, int DBG, int KS>
__device__ __forceinline__ void linear_accumulate(const LinearArgs& a, int m0, int n0, int it_lo, int it_hi, float* smem,
                                                  v16f (&acc)[MI][NJ]) {
    using G = LinearGeom<MI, NJ, WM, WN, PF, DBG, KS>;
    constexpr int NT = G::NT, SBK = G::SBK, LLD = G::LLD, CPR = G::CPR, BM = G::BM, BN = G::BN, VA = G::VA, VB = G::VB,
                  STAGE = G::STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, grp = tid >> 8;      // grp: which 16 of a stage's K
    const int l31 = lane & 31, l5 = lane >> 5;
    const int wm0 = (wave / WN) * (32 * MI), wn0 = (wave % WN) * (32 * NJ);

    // global -> register staging: vector v = tid + NT s of an image is one 16-byte piece of one row of the stage.  A ds_write_b128
    // is served in groups of 8 consecutive lanes over 32 banks: with 36-dword rows (KS = 2) the 8 pieces of ONE row fill them
    // exactly; with 20-dword rows (KS = 1, 4 pieces per row) rows r and r + 1 would overlap on 4 banks (rocprofv3: one LDS cycle in
    // three a conflict), rows r and r + 4 (80 dwords = 16 mod 32) do not — so a group of 8 lanes takes rows r and r + 4.
    auto row_of = [](int v) {
        if (CPR == 4) return (v >> 5) * 8 + ((v >> 3) & 3) + 4 * ((v >> 2) & 1);
        return v / CPR;
    };
    const float* pa[VA];
    const float* pb[VB];
    int wa[VA], wb[VB];
    const int64_t kbase = (int64_t)it_lo * SBK;
#pragma unroll
    for (int s = 0; s < VA; ++s) {
        const int v = tid + NT * s, row = min(row_of(v), BM - 1);
        pa[s] = a.X + (int64_t)min(m0 + row, a.M - 1) * a.ldx + 4 * (v % CPR) + kbase;        // rows past M: a valid row, never stored
        wa[s] = row * LLD + 4 * (v % CPR);
    }
#pragma unroll
    for (int s = 0; s < VB; ++s) {
        const int v = tid + NT * s, row = min(row_of(v), BN - 1);
        pb[s] = a.W + (int64_t)min(n0 + row, a.N - 1) * a.ldw + 4 * (v % CPR) + kbase;
        wb[s] = BM * LLD + row * LLD + 4 * (v % CPR);
    }
    constexpr bool TAIL_A = (BM * CPR) % NT != 0, TAIL_B = (BN * CPR) % NT != 0;
    const bool last_a = !TAIL_A || tid + NT * (VA - 1) < BM * CPR;              // wave-uniform (multiples of 64 threads)
    const bool last_b = !TAIL_B || tid + NT * (VB - 1) < BN * CPR;

    v4f ga[PF][VA], gb[PF][VB];
    auto gload = [&](int k0, auto rc) {
        constexpr int R = decltype(rc)::value;
#pragma unroll
        for (int s = 0; s < VA; ++s)
            if (s + 1 < VA || last_a) ga[R][s] = *reinterpret_cast<const v4f*>(pa[s] + k0);
#pragma unroll
        for (int s = 0; s < VB; ++s)
            if (s + 1 < VB || last_b) gb[R][s] = *reinterpret_cast<const v4f*>(pb[s] + k0);
    };
    auto lstore = [&](float* stage, auto rc) {
        constexpr int R = decltype(rc)::value;
#pragma unroll
        for (int s = 0; s < VA; ++s)
            if (s + 1 < VA || last_a) *reinterpret_cast<v4f*>(stage + wa[s]) = ga[R][s];
#pragma unroll
        for (int s = 0; s < VB; ++s)
            if (s + 1 < VB || last_b) *reinterpret_cast<v4f*>(stage + wb[s]) = gb[R][s];
    };

    const int fa_off = (wm0 + l31) * LLD + 4 * l5 + LBK * grp;     // + 32 i rows, + 8 q floats
    const int fb_off = BM * LLD + (wn0 + l31) * LLD + 4 * l5 + LBK * grp;
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
    // Stage s is loaded into register set s % PF; iteration `it` stores stage it+1 and refills that set with stage it+1+PF.
    const int T = it_hi - it_lo;
    gload(0, IC<0>{});
    lstore(smem, IC<0>{});
    if (T > 1) gload(SBK, IC<1 % PF>{});
    if constexpr (PF > 1) if (T > 2) gload(2 * SBK, IC<2 % PF>{});
    if constexpr (PF > 2) if (T > 3) gload(3 * SBK, IC<3 % PF>{});
    __syncthreads();
    fread(smem, 0, 0);
    auto body = [&](auto rc, int it) {
        float* cur = smem + (it & 1) * STAGE;
        float* oth = smem + ((it + 1) & 1) * STAGE;
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        fread(cur, 1, 1);                      // block 1 of this stage: lands under the MFMAs of block 0
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 1, 4);
        __builtin_amdgcn_sched_barrier(0);
        if (DBG < 2 && it + 1 < T) lstore(oth, rc);       // stage it+1 (loaded PF stages ago) -> the other buffer
        __syncthreads();
        if (DBG < 1 && it + 1 + PF < T) gload((it + 1 + PF) * SBK, rc);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < T) fread(oth, 0, 0);      // block 0 of the next stage: lands under the MFMAs of block 1
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 1, 4);
    };
    for (int it0 = 0; it0 < T; it0 += PF) {
        body(IC<1 % PF>{}, it0);
        if constexpr (PF > 1) if (it0 + 1 < T) body(IC<2 % PF>{}, it0 + 1);
        if constexpr (PF > 2) if (it0 + 2 < T) body(IC<3 % PF>{}, it0 + 2);
    }
}

// The end of a tile: (KS = 2: the two wave groups' halves of every sum are brought together,) bias, activation, residual, store.
template <int MI, int NJ, int WM, int WN, int PF, int DBG, int KS>
__device__ __forceinline__ void linear_finish(const LinearArgs& a, int m0, int n0, float* smem, v16f (&acc)[MI][NJ]) {
    using G = LinearGeom<MI, NJ, WM, WN, PF, DBG, KS>;
    constexpr int BM = G::BM, BN = G::BN, NBLK = G::NBLK, HB = G::HB;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, grp = tid >> 8;
    const int l31 = lane & 31, l5 = lane >> 5;
    const int wm0 = (wave / WN) * (32 * MI), wn0 = (wave % WN) * (32 * NJ);
    // KS == 2: the two wave groups hold the two halves of every sum.  They swap HALF of their accumulator blocks through LDS
    // (lane-linear 16-byte pieces: [wave][block][4][lane]) — group 0 ends up owning blocks [0, HB), group 1 blocks [HB, MI NJ),
    // each complete — so that all eight waves share the epilogue.  Fixed order of the two addends: bit-reproducible.
    if constexpr (KS == 2) {
        __syncthreads();                                       // everybody is done with the stage buffers
        v4f* red = reinterpret_cast<v4f*>(smem) + wave * (NBLK * 4 * 64) + lane;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int t = i * NJ + j;
                if ((t < HB) == (grp == 1)) {                  // a block the OTHER group will own
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v4f v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        red[(t * 4 + q) * 64] = v;
                    }
                }
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int t = i * NJ + j;
                if ((t < HB) == (grp == 0)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4f v = red[(t * 4 + q) * 64];
#pragma unroll
                        for (int e = 0; e < 4; ++e)       // group 0's partial first, then group 1's, whoever adds
                            acc[i][j][4 * q + e] = grp == 0 ? acc[i][j][4 * q + e] + v[e] : v[e] + acc[i][j][4 * q + e];
                    }
                }
            }
    }

    // epilogue: C[row][col], row = (r & 3) + 8 (r >> 2) + 4 l5 inside a 32 x 32 block, col = l31: a wave instruction writes two
    // runs of 32 consecutive floats.  The activation is chosen once, outside the unrolled element loops; a tile that lies inside
    // the matrix takes a path without per-element predicates (the residual loads of a block are then issued together).
    const float* __restrict__ bias = a.bias;
    const float* __restrict__ res = a.res;
    float* __restrict__ Y = a.Y;
    const int64_t ldr = a.ldr, ldy = a.ldy;
    const int M = a.M, N = a.N;
    const bool interior = m0 + BM <= M && n0 + BN <= N;
    auto epilogue = [&](auto actfn) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wn0 + j * 32 + l31;
            const bool n_ok = n < N;
            const float bv = (bias != nullptr && n_ok) ? bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int t = i * NJ + j;
                if (KS == 2 && (t < HB) != (grp == 0)) continue;
                const int mb = m0 + wm0 + i * 32 + 4 * l5;
                if (interior) {
                    float rv[16];
                    if (res != nullptr) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) rv[r] = res[(int64_t)(mb + (r & 3) + 8 * (r >> 2)) * ldr + n];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = actfn(acc[i][j][r] + bv);
                        if (res != nullptr) v += rv[r];
                        Y[(int64_t)(mb + (r & 3) + 8 * (r >> 2)) * ldy + n] = v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mb + (r & 3) + 8 * (r >> 2);
                        if (n_ok && m < M) {
                            float v = actfn(acc[i][j][r] + bv);
                            if (res != nullptr) v += res[(int64_t)m * ldr + n];
                            Y[(int64_t)m * ldy + n] = v;
                        }
                    }
                }
            }
        }
    };
    if (a.act == ACT_QUICK_GELU) epilogue([](float x) { return x / (1.0f + __expf(-1.702f * x)); });       // x * sigmoid(1.702 x)
    else if (a.act == ACT_GELU_ERF) epilogue([](float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); });
    else epilogue([](float x) { return x; });
}

// Measured and dropped (profiles/r03_mb_linear_variants.txt): the 128 x 128 four-wave kernel compiled for THREE workgroups per
// compute unit (amdgpu_waves_per_eu(3, 8): 148 registers, no spills) runs 4-10 % slower than at two (qkv 105.9 vs 117.4 TF);
// a persistent form that walks a workgroup's tiles as one software pipeline (next tile's first stages loaded during the current
// tile's last ones, stage 0 in LDS before the epilogue) 6-8 % slower (110.6 / 112.5 vs 117.4 / 121.1 TF) — the hardware's own
// dispatch of one-tile workgroups balances the compute units better than a static walk gains from the hidden prologue.
// A v_mfma_f32_16x16x4_f32 form of the same pipeline (tile edges in steps of 16: 96 x 160, 128 x 160, 192 x 160 — the shapes
// hipBLASLt's tuned solutions take for these projections: 510 / 250 tiles for qkv / out at 6 400 rows) compiled to 254-436
// registers per lane (one wave per SIMD for the two larger tiles) and reached 95-109 TF on qkv, 61-87 on out
// (profiles/r03_mb_linear_mfma16.txt): removed.
// Staging by LDS-DMA (global_load_lds_dwordx4 into an unpadded image with 16-byte pieces XOR-swizzled by (row >> 2) & 3 — conflict
// free for the b128 fragment reads —, three stage buffers, DMAs two stages ahead behind counted s_waitcnt vmcnt and a bare
// s_barrier, no staging registers, no ds_write): correct, 111 / 119 TF on qkv / fc1 against 118 / 121 for the register-staged
// form, 124-130 against 131-134 on the SDXL shapes (profiles/r03_mb_linear_dma.txt): the LDS store path is not what holds the
// loop back; removed.
template <int MI, int NJ, int WM, int WN, int PF, int DBG, int KS>
__global__ __launch_bounds__(256 * KS) void linear_f32_kernel(LinearArgs a) {
    static_assert(WM * WN == 4 && (KS == 1 || KS == 2), "four waves per K group");
    using G = LinearGeom<MI, NJ, WM, WN, PF, DBG, KS>;
    __shared__ __attribute__((aligned(16))) float smem[G::SMEM];
    // Workgroups go to the 8 XCDs round-robin by linear id.  XCD x takes the tiles [x * per, (x + 1) * per): a contiguous
    // run of row tiles with all their column tiles, so the rows of X an XCD streams are its own and every column tile of W
    // is re-read from that XCD's L2.
    const int per = (a.tiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= a.tiles || (int)(blockIdx.x >> 3) >= per) return;
    int bm, bn;
    linear_tile_of(a, tile, bm, bn);
    v16f acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    linear_accumulate<MI, NJ, WM, WN, PF, DBG, KS>(a, bm * G::BM, bn * G::BN, 0, a.K / G::SBK, smem, acc);
    linear_finish<MI, NJ, WM, WN, PF, DBG, KS>(a, bm * G::BM, bn * G::BN, smem, acc);
}

// Measured and dropped — a "stream-K tail" for the partial last round of tiles.  A launch of t 128 x 128 tiles on the chip's 512
// workgroup slots takes ceil(t / 512) rounds whatever the fraction of the last one: 6 400 x 768 -> 2 304 (900 tiles, 1.76 rounds)
// costs what 1 008 tiles cost, 114 TF against 129; fc1 (1 200 tiles) 124 against 133 at three full rounds
// (scripts/mb_linear_rounds.py, profiles/r03_mb_linear_rounds.txt).  The form built for it: the first floor(t / 512) * 512 tiles
// as they are, the rest cut by K into 512 equal runs of stages whatever tile borders they cross, partial tiles published to a
// workspace (write-through stores, ticket per tile, the last arriver sums all parts in run order: bit-reproducible, nobody waits)
// — correct on every test shape, and worth 2 % on qkv (121.2 vs 118.7 TF), nothing on fc1 (123.1 vs 124.4), less than the
// 160 x 128 eight-wave tiles on out / fc2 (95.6 vs 111.9, 122.1 vs 122.9 TF; profiles/r03_mb_linear_tail.txt): an fp32 128 x 128
// tile is 48 us of one compute unit's matrix pipe, a partial tile 64 KB out to memory and 64 KB back in — with two to three
// parts per tile the exchange costs what the balanced last round gains.  Two independent half-size chains on two streams
// (scripts/mb_two_streams.py) fill each other's tails and end up at the one-stream rate (120 TF), not above it.
// Sixteen waves per workgroup (four K groups) for the few 64 x 64 tiles of the mean keys through fc2 (1 000 x 3072 -> 768, 192
// tiles: at most one workgroup per compute unit): 64.5 us against 66.1 with eight (profiles/r03_mb_linear_16w.txt) — that launch
// is not short of waves; removed.
// ---- few tiles, long K: the K range of every tile cut over several workgroups ("split-K") -----------------------------------
// fc2 of a 100-concept edit (640 x 3072 -> 768) or of the mean keys of a 1 000-concept one (1 000 x 3072 -> 768) is 30-48 tiles of
// 128 x 128 on a K of 3072: most compute units would get nothing, and 64 x 64 tiles (one workgroup each, 120-192 of them) stream
// 16 flop per byte.  Here tile t is computed by `parts` workgroups, part p over the stages [p T / parts, (p + 1) T / parts); every
// part publishes its accumulators to workspace slot [t][p] (lane-linear image) and takes a ticket on the tile's counter; the one
// whose ticket is the last re-reads ALL slots of the tile in part order (its own included: the sum order never depends on who came
// last — bit-reproducible), runs the epilogue and puts the counter back to zero.  Nobody waits for anybody.  Cross-CU hand-off
// as in gemm_f64.h's stream-K (cdna_hip_programming.md Guideline 16, write-through form): sc1 stores -> every wave's s_waitcnt
// vmcnt(0) -> barrier -> lane 0 relaxed agent-scope ticket; last arriver: lane 0 acquire fence (agent) -> barrier -> plain loads.
// The parts of a tile are dealt to ONE XCD (consecutive there).
struct LinearSplit {
    int parts;
    float* partials;             // [tiles][parts][256 * 64]
    unsigned* counters;          // [tiles], zero on entry, zero again on exit
};

template <int PF>
__global__ __launch_bounds__(256) void linear_f32_splitk_kernel(LinearArgs a, LinearSplit w) {
    constexpr int MI = 2, NJ = 2, WM = 2, WN = 2;
    using G = LinearGeom<MI, NJ, WM, WN, PF, 0, 1>;
    constexpr int NT = G::NT, SLOT = NT * 16 * MI * NJ;
    __shared__ __attribute__((aligned(16))) float smem[G::SMEM];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const int T = a.K / G::SBK;
    // work item u = tile * parts + part; XCD x takes the items [x * per, (x + 1) * per): all parts of a tile on one XCD
    const int items = a.tiles * w.parts, per = ((a.tiles + 7) / 8) * w.parts;
    const int u = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (u >= items || (int)(blockIdx.x >> 3) >= per) return;
    const int tile = u / w.parts, part = u - tile * w.parts;
    const int it_lo = (int)((int64_t)part * T / w.parts), it_hi = (int)((int64_t)(part + 1) * T / w.parts);
    int bm, bn;
    linear_tile_of(a, tile, bm, bn);
    const int m0 = bm * G::BM, n0 = bn * G::BN;
    v16f acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (it_hi > it_lo) linear_accumulate<MI, NJ, WM, WN, PF, 0, 1>(a, m0, n0, it_lo, it_hi, smem, acc);
    float* slots = w.partials + (int64_t)tile * w.parts * SLOT;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4f x = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                float* dst = slots + (int64_t)part * SLOT + ((int64_t)((i * NJ + j) * 4 + q) * NT + tid) * 4;
                // write-through (sc1): the bytes leave the XCD's L2 as they are stored, so publishing them needs no agent-scope
                // release (an L2 write-back of everything the workgroup's XCD has dirtied)
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(x) : "memory");
            }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(w.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old == (unsigned)(w.parts - 1)) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        w.counters[tile] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int p = 0; p < w.parts; ++p) {
        const float* src = slots + (int64_t)p * SLOT;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4f x = *reinterpret_cast<const v4f*>(src + ((int64_t)((i * NJ + j) * 4 + q) * NT + tid) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += x[e];
                }
    }
    linear_finish<MI, NJ, WM, WN, PF, 0, 1>(a, m0, n0, smem, acc);
}

constexpr int kSplitTiles = 256;                 // most tiles a split-K launch has (= ticket counters at the head of the workspace)
constexpr int kSplitItems = 512;                 // most (tile, part) work items = partial-tile slots behind the counters
constexpr int kSplitSlot = 256 * 64;             // floats of one partial 128 x 128 tile

struct LinearCfg { int bm, bn; };
static const LinearCfg kLinearCfgs[] = {{160, 128}, {128, 128}, {256, 128}, {64, 64}};

}  // namespace emcid

using namespace emcid;

extern "C" {

/* bytes of the workspace emcid_linear_ws_f32 wants (ticket counters + partial-tile slots of the split-K form); zeroed ONCE by the
 * caller when allocated, left zeroed by every launch; one per stream whose launches may overlap another's */
int64_t emcid_linear_workspace_bytes(void) { return (int64_t)kSplitTiles * 4 + (int64_t)kSplitItems * kSplitSlot * 4; }

int emcid_linear_ws_f32(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual,
                        int64_t ldr, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, int act, int cfg, void* workspace,
                        int64_t workspace_bytes, void* stream) {
    EMCID_CHECK_ARG(X && W && Y && M > 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N);
    EMCID_CHECK_ARG(K % LBK == 0 && ldx % 4 == 0 && ldw % 4 == 0 && aligned16(X) && aligned16(W));
    EMCID_CHECK_ARG(M < (1 << 24) && N < (1 << 24) && K < (1 << 24) && (residual == nullptr || ldr >= N));
    EMCID_CHECK_ARG(act >= ACT_NONE && act <= ACT_GELU_ERF && cfg >= -1 && cfg < 128 + 16 * 128);
    EMCID_CHECK_ARG(workspace == nullptr || (workspace_bytes >= emcid_linear_workspace_bytes() && aligned16(workspace)));
    // cfg: bits 0-1 tile, bits 2-3 prefetch distance - 1 (0..2), bits 4-5 DBG (tile 0, one K group only), bit 6: ONE K group
    // (4 waves) instead of two (8 waves), bits 7-10: split-K parts (2..8; tile 1, bit 6 set, needs the workspace); -1: auto
    int parts = cfg < 0 ? 0 : (cfg >> 7) & 15;
    if (cfg >= 0) cfg &= 127;
    int tile_sel = cfg < 0 ? -1 : (cfg & 3);
    constexpr int ks_env = 2;
    int pf = cfg < 0 ? 1 : ((cfg >> 2) & 3) + 1;
    const int dbg = cfg < 0 ? 0 : (cfg >> 4) & 3;
    int ks = cfg < 0 ? (ks_env == 1 ? 1 : 2) : ((cfg >> 6) & 1) ? 1 : 2;
    if (K % (2 * LBK) != 0) ks = 1;
    if (dbg) ks = 1;
    EMCID_CHECK_ARG(pf >= 1 && pf <= 3 && dbg <= 2 && (dbg == 0 || tile_sel == 0));
    const int64_t t128 = ((M + 127) / 128) * ((N + 127) / 128);
    EMCID_CHECK_ARG(parts == 0 || (parts >= 2 && parts <= 8 && tile_sel == 1 && ks == 1 && dbg == 0 && workspace != nullptr &&
                                   t128 <= kSplitTiles && t128 * parts <= kSplitItems && K / LBK >= parts));
    if (tile_sel < 0 && workspace != nullptr && t128 <= 128 && K >= 2048) {
        // at most half the compute units would get a 128 x 128 tile: cut K so that about one workgroup per compute unit runs.
        // Measured (scripts/mb_linear.py, profiles/r03_mb_linear_splitk.txt; us, 64 x 64 tiles -> split-K; hipBLASLt beside it):
        // 640 x 3072 -> 768 61.7 -> 43.2 (8 parts; 30), 1 000 x 5120 -> 1280 173.8 -> 123.0 (6) / 131.1 (3; 107), 1 000 x 3072 ->
        // 768 65.2 -> 61.6 (5; 41); a part's publish + the last arriver's re-read and epilogue cost ~15 us per launch, so K = 768
        // launches (640 x 768 -> 2304: 31.6 -> 37.3) stay on the small tiles.
        const int64_t want = std::min<int64_t>(std::min<int64_t>(8, 256 / t128), (K / LBK) / 8);
        if (want >= 2) tile_sel = 1, ks = 1, pf = 2, parts = (int)want;
    }
    if (tile_sel < 0) {
        // Pick (tile, waves) by fill x base rate.  fill: a compute unit works through ceil(tiles / 256) tiles (co-resident
        // workgroups share its matrix pipe), the average one through tiles / 256.  Base rates = fraction of the f32 MFMA rate
        // measured with scripts/mb_linear.py on MI355X at full fill (profiles/r03_mb_linear.txt):
        //   128 x 128, 4 waves, prefetch 2, two workgroups per compute unit: 0.83 (K = 768) .. 0.90 (K = 5120)
        //   160 x 128, 8 waves (K split inside the workgroup), one per compute unit: 0.765, 0.83 from K = 2048
        //   64 x 64, up to 8 workgroups per compute unit: 0.70
        auto fill = [](int64_t t) { return ((double)t / 256.0) / (double)((t + 255) / 256); };
        const int64_t t160 = ((M + 159) / 160) * ((N + 127) / 128);
        const int64_t t64 = ((M + 63) / 64) * ((N + 63) / 64);
        const bool split_ok = K % (2 * LBK) == 0 && ks_env != 1;
        const double s128 = (K < 1024 ? 0.83 : K < 4096 ? 0.875 : 0.90) * fill(t128);
        const double s160 = split_ok ? (K >= 2048 ? 0.83 : 0.765) * fill(t160) : 0.0;
        const double s64 = 0.70 * fill(t64);
        if (s128 >= s160 && s128 >= s64) tile_sel = 1, ks = 1, pf = 2;
        else if (s160 >= s64) tile_sel = 0, ks = 2, pf = 2;
        else tile_sel = 3, ks = (K >= 2048 && split_ok) ? 2 : 1, pf = ks == 2 ? 2 : 3;
    }
    const int bm = kLinearCfgs[tile_sel].bm, bn = kLinearCfgs[tile_sel].bn;
    const int tiles_m = (int)((M + bm - 1) / bm), tiles_n = (int)((N + bn - 1) / bn);
    const int tiles = tiles_m * tiles_n;
    const int per = (tiles + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    const int rb = 4;        // row tiles per super-row of the tile order (profiles/r03_ab_linear_tile_order.txt)
    const LinearArgs a{X, ldx, W, ldw, bias, residual, ldr, Y, ldy, (int)M, (int)N, (int)K, act, tiles_n, tiles, rb};
    ScopedProf sp(KC_LINEAR, st);
    if (parts >= 2) {
        LinearSplit w{parts, (float*)((char*)workspace + (int64_t)kSplitTiles * 4), (unsigned*)workspace};
        const unsigned grid = (unsigned)(((tiles + 7) / 8) * parts * 8);
        if (pf == 1) hipLaunchKernelGGL((linear_f32_splitk_kernel<1>), dim3(grid), dim3(256), 0, st, a, w);
        else if (pf == 2) hipLaunchKernelGGL((linear_f32_splitk_kernel<2>), dim3(grid), dim3(256), 0, st, a, w);
        else hipLaunchKernelGGL((linear_f32_splitk_kernel<3>), dim3(grid), dim3(256), 0, st, a, w);
        EMCID_CHECK_LAUNCH();
        return EMCID_OK;
    }
#define EMCID_LINEAR_LAUNCH(MI_, NJ_, WM_, WN_, PF_, DBG_, KS_)                                                              \
    hipLaunchKernelGGL((linear_f32_kernel<MI_, NJ_, WM_, WN_, PF_, DBG_, KS_>), dim3((unsigned)(per * 8)), dim3(256 * KS_), 0, \
                       st, a)
#define EMCID_LINEAR_PF(MI_, NJ_, WM_, WN_)                                              \
    do {                                                                                  \
        if (ks == 2) {                                                                    \
            if (pf == 1) EMCID_LINEAR_LAUNCH(MI_, NJ_, WM_, WN_, 1, 0, 2);                \
            else EMCID_LINEAR_LAUNCH(MI_, NJ_, WM_, WN_, 2, 0, 2);                        \
        } else if (pf == 1) EMCID_LINEAR_LAUNCH(MI_, NJ_, WM_, WN_, 1, 0, 1);             \
        else if (pf == 2) EMCID_LINEAR_LAUNCH(MI_, NJ_, WM_, WN_, 2, 0, 1);               \
        else EMCID_LINEAR_LAUNCH(MI_, NJ_, WM_, WN_, 3, 0, 1);                            \
    } while (0)
    if (dbg == 1) EMCID_LINEAR_LAUNCH(5, 1, 1, 4, 2, 1, 1);
    else if (dbg == 2) EMCID_LINEAR_LAUNCH(5, 1, 1, 4, 2, 2, 1);
    else switch (tile_sel) {
        case 0: EMCID_LINEAR_PF(5, 1, 1, 4); break;
        case 1: EMCID_LINEAR_PF(2, 2, 2, 2); break;
        case 2: EMCID_LINEAR_PF(4, 2, 2, 2); break;
        default: EMCID_LINEAR_PF(1, 1, 2, 2); break;
    }
#undef EMCID_LINEAR_PF
#undef EMCID_LINEAR_LAUNCH
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int emcid_linear_f32(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual,
                     int64_t ldr, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, int act, int cfg, void* stream) {
    return emcid_linear_ws_f32(X, ldx, W, ldw, bias, residual, ldr, Y, ldy, M, N, K, act, cfg, nullptr, 0, stream);
}

}  // extern "C"
