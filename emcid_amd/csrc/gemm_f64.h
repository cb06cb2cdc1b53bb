// fp64 MFMA tile GEMM for gfx950 (v_mfma_f64_16x16x4_f64), the contraction engine under the
// Stage-2 closed form: SYRK (K^T K), Cholesky panel / trailing updates, blocked TRSM, dW = R^T X.
//
//   D[m][n] = sum_k opA(m,k) * opB(k,n),   epilogue functor decides what happens to D.
//
// Operand storage is a template flag per operand:
//   KC = true  : stored [rows][K]  (K contiguous)  -> LDS image [rows][BK+4]
//   KC = false : stored [K][rows]  (rows contiguous) -> LDS image [BK][rows]
// Both images are straight 16-B copies of global memory (no transpose pass).  The f64 MFMA takes ONE
// double per lane, A[i = lane&15][slot = lane>>4], B[slot = lane>>4][j = lane&15], and sums over the 4
// slots.  Fragments are fetched 16 B per lane (ds_read_b128, 256 B/clk) — never as two 8-B reads, which
// hipcc fuses into ds_read2_b64 (half rate, 32-bank rule: measured 2-way conflicts on every read):
//   KC image : a lane reads k = 8*k8 + 2*slot + {0,1} of its row in one b128; the two halves feed two
//              MFMAs, so over a k8 step slot s covers k = 2s (first MFMA) and 2s+1 (second).  Any
//              permutation of k is legal as long as A and B use the same one.  Row stride BK+4
//              doubles (== 4 mod 8) puts the 16 lanes of every b128 lane group on 16 distinct 16-B slots.
//   !KC image: a lane reads rows {2*(lane&15), +1} of a 32-row group at one k in one b128; the halves
//              belong to two adjacent 16-row MFMA tiles, whose row r then is global row 32*p + 2*r + e.
//              Row stride = rows (a multiple of 32) keeps the lane groups conflict-free.
// C/D layout of the f64 MFMA: col = lane&15, row = (lane>>4) + 4*reg  (NOT the f32 map).
//
// Pipeline: register-staged double buffering (global_load_dwordx4 for tile t+1 issued before the
// MFMAs of tile t, ds_write_b128 after them, one barrier per K tile).  An f64 MFMA holds its SIMD
// for 64 cycles, so staging traffic hides under the matrix pipe at one or two waves per SIMD.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace emcid {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool KC, int ROWS, int BK>
struct OpTile {
    static constexpr int LD = KC ? (BK + 4) : ROWS;
    static_assert(KC || ROWS % 32 == 0, "row-contiguous tiles come in 32-row groups");
    static_assert(BK % 8 == 0, "a k8 step is 8 deep");
    // fragment values of the TILES 16-row MFMA tiles starting at tile row w0, for k8 step `k8`:
    // f[e][t] feeds the e-th of the two MFMAs of this step for tile t.
    template <int TILES>
    __device__ static __forceinline__ void frags(const double* lds, int w0, int k8, int l15, int l4, double (&f)[2][TILES]) {
        if (KC) {
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const v2d x = *reinterpret_cast<const v2d*>(lds + (w0 + t * 16 + l15) * LD + 8 * k8 + 2 * l4);
                f[0][t] = x[0]; f[1][t] = x[1];
            }
        } else {
            static_assert(KC || TILES % 2 == 0, "row-contiguous operands are fetched as tile pairs");
#pragma unroll
            for (int p = 0; p < TILES / 2; ++p)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const v2d x = *reinterpret_cast<const v2d*>(lds + (8 * k8 + 2 * l4 + e) * LD + w0 + 32 * p + 2 * l15);
                    f[e][2 * p] = x[0]; f[e][2 * p + 1] = x[1];
                }
        }
    }
    // global row (relative to the wave's origin w0) of MFMA row/col index r (0..15) of tile t
    __device__ static __forceinline__ int index_of(int t, int r) { return KC ? t * 16 + r : 32 * (t >> 1) + 2 * r + (t & 1); }
    static constexpr int SIZE = KC ? ROWS * LD : BK * LD;  // doubles
    static constexpr int NVEC = ROWS * BK / 2;             // 16-byte vectors per tile
    __device__ static __forceinline__ int off(int r, int k) { return KC ? r * LD + k : k * LD + r; }
};

// Loads this thread's share of a (ROWS x BK) operand tile into registers, zero-filling out of range.
template <bool KC, int ROWS, int BK, int NT>
__device__ __forceinline__ void load_tile(v2d (&reg)[ROWS * BK / 2 / NT], const double* __restrict__ g, int64_t ld,
                                          int row0, int rows_total, int k0, int K, int tid) {
    constexpr int NV = ROWS * BK / 2 / NT;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * NT;
        v2d x = {0.0, 0.0};
        if (KC) {
            const int r = row0 + v / (BK / 2);
            const int k = k0 + 2 * (v % (BK / 2));
            if (r < rows_total) {
                const double* p = g + (int64_t)r * ld + k;
                if (k + 1 < K) x = *reinterpret_cast<const v2d*>(p);
                else if (k < K) x[0] = p[0];
            }
        } else {
            const int k = k0 + v / (ROWS / 2);
            const int r = row0 + 2 * (v % (ROWS / 2));
            if (k < K) {
                const double* p = g + (int64_t)k * ld + r;
                if (r + 1 < rows_total) x = *reinterpret_cast<const v2d*>(p);
                else if (r < rows_total) x[0] = p[0];
            }
        }
        reg[i] = x;
    }
}

template <bool KC, int ROWS, int BK, int NT>
__device__ __forceinline__ void store_tile(const v2d (&reg)[ROWS * BK / 2 / NT], double* lds, int tid) {
    constexpr int NV = ROWS * BK / 2 / NT;
    using T = OpTile<KC, ROWS, BK>;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * NT;
        int o;
        if (KC) o = (v / (BK / 2)) * T::LD + 2 * (v % (BK / 2));
        else    o = (v / (ROWS / 2)) * T::LD + 2 * (v % (ROWS / 2));
        *reinterpret_cast<v2d*>(lds + o) = reg[i];
    }
}

struct GemmShape {
    const double* A; int64_t lda;
    const double* B; int64_t ldb;
    int M, N, K;
    int lower_only;  // skip output tiles that lie entirely above the diagonal (SYRK / Cholesky updates)
    int64_t sA = 0, sB = 0;  // element strides between the problems of a batch (blockIdx.z)
    int batch = 1;
    // an outer batch dimension on top (e.g. matrices x diagonal blocks): problem (b1, b2), b1 < batch, b2 < batch2
    int64_t sA2 = 0, sB2 = 0;
    int batch2 = 1;
    // triangular operands (inverted diagonal blocks, explicit inverse factors), bit mask:
    //   1 = B(k, n) is zero for k > n,  2 = B(k, n) zero for k < n,  4 = A(m, k) zero for k > m,  8 = A(m, k) zero for k < m.
    // The K loop of an output tile then only covers the k range that can contribute; tiles are issued heaviest first.
    int tri = 0;
    // split the K range of every output tile over ksplit workgroups (blockIdx.z = batch * ksplit + split); only for
    // epilogues that ACCUMULATE into C (EpiAxpby with beta == 1), which then add their partial with f64 atomics
    int ksplit = 1;
    // kchunk > 0: instead of an even split, every workgroup takes a fixed run of kchunk K-tiles (blockIdx.z picks the
    // run; runs past the tile's own K range exit at once) — equal work units when `tri` makes the ranges differ
    int kchunk = 0;
    // pair = 1 (with a triangular operand): one workgroup computes tile j and tile ntiles-1-j of the triangular
    // dimension, so all workgroups do equal work; the grid is halved along that dimension
    int pair = 0;
    int xcd_rows = 0;   // number tile rows fastest (set by the launcher for B-side triangles; see gemm_f64_kernel)
    int pf = 0;         // small tiles: keep PF K-tiles of global loads in flight (set by the launcher; see gemm_f64_tile)
    int lo_total = 0;   // lower-only square outputs: number of tiles that touch the lower triangle (> 0: the kernel
                        // enumerates exactly those, row by row, from the linear workgroup id; see gemm_f64_kernel)
    int lower_shift = 0;  // lower_only on a trapezoid: output row m stands for matrix row m + lower_shift (the rows of the
                          // output start lower_shift below its first column's diagonal element)
    int nofast = 0;       // 1: gemm_f64_tile_acc keeps its generic (masked) K loop for interior segments too (A/B switch)
};

// WGM x WGN waves per workgroup; each wave owns a (BM/WGM) x (BN/WGN) sub-tile.
// The same share of a tile WITHOUT branches: every 16-byte vector is either wholly inside or wholly outside the operand
// (the caller guarantees an even extent along the contiguous dimension), so the load is unconditional from a clamped
// address and `ok` says whether it counts.  The zeroing select is applied by store_tile_masked, i.e. after the MFMAs of
// the stage, so that nothing waits on the loads where they are issued and several K tiles can be in flight.
// `tile_valid` = false turns the whole call into dummy loads of g[0]: the ring issues the SAME number of loads on every
// path (a load skipped under a branch makes the compiler's s_waitcnt for an older set fall back to vmcnt(0)).
template <bool KC, int ROWS, int BK, int NT>
__device__ __forceinline__ void load_tile_nobranch(v2d (&reg)[ROWS * BK / 2 / NT], bool (&ok)[ROWS * BK / 2 / NT],
                                                   const double* __restrict__ g, int64_t ld, int row0, int rows_total, int k0,
                                                   int K, int tid, bool tile_valid) {
    constexpr int NV = ROWS * BK / 2 / NT;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * NT;
        int r, k;
        if (KC) { r = row0 + v / (BK / 2); k = k0 + 2 * (v % (BK / 2)); }
        else    { k = k0 + v / (ROWS / 2); r = row0 + 2 * (v % (ROWS / 2)); }
        ok[i] = tile_valid && r < rows_total && k < K;
        const double* p = ok[i] ? (KC ? g + (int64_t)r * ld + k : g + (int64_t)k * ld + r) : g;
        reg[i] = *reinterpret_cast<const v2d*>(p);
    }
}

template <bool KC, int ROWS, int BK, int NT>
__device__ __forceinline__ void store_tile_masked(const v2d (&reg)[ROWS * BK / 2 / NT], const bool (&ok)[ROWS * BK / 2 / NT],
                                                  double* lds, int tid) {
    constexpr int NV = ROWS * BK / 2 / NT;
    using T = OpTile<KC, ROWS, BK>;
    const v2d zero = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * NT;
        int o;
        if (KC) o = (v / (BK / 2)) * T::LD + 2 * (v % (BK / 2));
        else    o = (v / (ROWS / 2)) * T::LD + 2 * (v % (ROWS / 2));
        *reinterpret_cast<v2d*>(lds + o) = ok[i] ? reg[i] : zero;
    }
}

// One output tile (bm, bn), K tiles restricted by `tri`, split zs of the K range.
template <bool KCA, bool KCB, int BM, int BN, int BK, int WGM, int WGN, class Epi>
__device__ __forceinline__ void gemm_f64_tile(const GemmShape& p, const Epi& epi, int bm, int bn, int zs, double* smem,
                                              int kt_begin = -1, int kt_end = -1) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 16, NI = WN / 16;
    using TA = OpTile<KCA, BM, BK>;
    using TB = OpTile<KCB, BN, BK>;
    constexpr int STAGE = TA::SIZE + TB::SIZE;
    const int m0 = bm * BM, n0 = bn * BN;
    if (p.lower_only && n0 > m0 + p.lower_shift + BM - 1) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
    const int l15 = lane & 15, l4 = lane >> 4;

    v4d acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    // K tiles [t0, t1) that can contribute to this output tile
    int t0 = 0, t1 = (p.K + BK - 1) / BK;
    if (p.tri & 1) t1 = min(t1, (min(p.K, n0 + BN) + BK - 1) / BK);
    if (p.tri & 2) t0 = max(t0, min(n0, p.K) / BK);
    if (p.tri & 4) t1 = min(t1, (min(p.K, m0 + BM) + BK - 1) / BK);
    if (p.tri & 8) t0 = max(t0, min(m0, p.K) / BK);
    if (kt_begin >= 0) {              // an explicit run [kt_begin, kt_end) of this tile's own K tiles (stream-K caller)
        t1 = min(t1, t0 + kt_end);
        t0 += kt_begin;
        if (t0 >= t1) return;
    } else if (p.kchunk > 0) {
        t0 += zs * p.kchunk;
        t1 = min(t1, t0 + p.kchunk);
        if (t0 >= t1) return;
    } else if (p.ksplit > 1) {
        const int per = (t1 - t0 + p.ksplit - 1) / p.ksplit;
        t0 += zs * per;
        t1 = min(t1, t0 + per);
        if (t0 >= t1) return;
    }

    auto mfma_stage = [&](const double* As, const double* Bs) {
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            double a[2][MI], b[2][NI];
            TA::template frags<MI>(As, wm0, k8, l15, l4, a);
            TB::template frags<NI>(Bs, wn0, k8, l15, l4, b);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e][i], b[e][j], acc[i][j], 0, 0, 0);
        }
    };
    auto epilogue = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm0 + TA::index_of(i, l4 + 4 * r);
                    const int n = n0 + wn0 + TB::index_of(j, l15);
                    if (m < p.M && n < p.N) epi(m, n, acc[i][j][r]);
                }
    };

    // Small tiles (a wave has 8-16 MFMAs per K tile, ~0.2-0.4 us) at one workgroup per CU are bound by the global-load
    // latency when only the next K tile is in flight: a 32 x 64 tile took 0.9 us per K tile.  Here PF K tiles are in
    // flight in PF register sets (3-6 vectors each at these sizes); LDS stays double-buffered.  The loads are
    // branch-free (load_tile_nobranch), so the compiler can wait for the oldest set only (counted vmcnt).
    if constexpr (BM * BN <= 64 * 64) {
        if (p.pf && !p.nofast && m0 + BM <= p.M && n0 + BN <= p.N && t1 * BK <= p.K) {
            // interior tile: the same ring with precomputed per-thread pointers and no masks (see gemm_f64_tile_acc) — a wave of
            // these small tiles issues only 8-16 MFMAs per K tile, so the ~45 address / select instructions of the generic step
            // were a third of its time
            constexpr int PF = 4, NA = TA::NVEC / NT, NB = TB::NVEC / NT;
            const double* pa[NA];
            const double* pb[NB];
            int la[NA], lb[NB];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int v = tid + i * NT;
                if (KCA) { pa[i] = p.A + (int64_t)(m0 + v / (BK / 2)) * p.lda + 2 * (v % (BK / 2)); la[i] = (v / (BK / 2)) * TA::LD + 2 * (v % (BK / 2)); }
                else     { pa[i] = p.A + (int64_t)(v / (BM / 2)) * p.lda + m0 + 2 * (v % (BM / 2)); la[i] = (v / (BM / 2)) * TA::LD + 2 * (v % (BM / 2)); }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int v = tid + i * NT;
                if (KCB) { pb[i] = p.B + (int64_t)(n0 + v / (BK / 2)) * p.ldb + 2 * (v % (BK / 2)); lb[i] = (v / (BK / 2)) * TB::LD + 2 * (v % (BK / 2)); }
                else     { pb[i] = p.B + (int64_t)(v / (BN / 2)) * p.ldb + n0 + 2 * (v % (BN / 2)); lb[i] = (v / (BN / 2)) * TB::LD + 2 * (v % (BN / 2)); }
            }
            const int64_t sa = KCA ? BK : (int64_t)BK * p.lda, sb = KCB ? BK : (int64_t)BK * p.ldb;
            v2d fa[PF][NA], fb[PF][NB];
            auto fetch = [&](int s, int t) {        // past the end: the last tile again (never stored)
                const int64_t tt = t < t1 ? t : t1 - 1;
#pragma unroll
                for (int i = 0; i < NA; ++i) fa[s][i] = *reinterpret_cast<const v2d*>(pa[i] + tt * sa);
#pragma unroll
                for (int i = 0; i < NB; ++i) fb[s][i] = *reinterpret_cast<const v2d*>(pb[i] + tt * sb);
            };
            auto stash = [&](int s, double* stage) {
#pragma unroll
                for (int i = 0; i < NA; ++i) *reinterpret_cast<v2d*>(stage + la[i]) = fa[s][i];
#pragma unroll
                for (int i = 0; i < NB; ++i) *reinterpret_cast<v2d*>(stage + TA::SIZE + lb[i]) = fb[s][i];
            };
#pragma unroll
            for (int s = 0; s < PF; ++s) fetch(s, t0 + s);
            stash(0, smem);
            __syncthreads();
            for (int base = t0; base < t1; base += PF) {
#pragma unroll
                for (int s = 0; s < PF; ++s) {
                    const int t = base + s;          // tile t is in LDS stage s & 1 (base - t0 is a multiple of PF)
                    if (t < t1) {
                        const double* As = smem + (s & 1) * STAGE;
                        fetch(s, t + PF);
                        mfma_stage(As, As + TA::SIZE);
                        if (t + 1 < t1) stash((s + 1) % PF, smem + ((s + 1) & 1) * STAGE);
                        __syncthreads();
                    }
                }
            }
            epilogue();
            return;
        }
        if (p.pf) {
            constexpr int PF = 4, NA = TA::NVEC / NT, NB = TB::NVEC / NT;
            v2d qa[PF][NA], qb[PF][NB];
            bool oa[PF][NA], ob[PF][NB];
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                load_tile_nobranch<KCA, BM, BK, NT>(qa[s], oa[s], p.A, p.lda, m0, p.M, (t0 + s) * BK, p.K, tid, t0 + s < t1);
                load_tile_nobranch<KCB, BN, BK, NT>(qb[s], ob[s], p.B, p.ldb, n0, p.N, (t0 + s) * BK, p.K, tid, t0 + s < t1);
            }
            store_tile_masked<KCA, BM, BK, NT>(qa[0], oa[0], smem, tid);
            store_tile_masked<KCB, BN, BK, NT>(qb[0], ob[0], smem + TA::SIZE, tid);
            __syncthreads();
            for (int base = t0; base < t1; base += PF) {
#pragma unroll
                for (int s = 0; s < PF; ++s) {
                    const int t = base + s;          // tile t is in LDS stage s & 1 (base - t0 is a multiple of PF)
                    if (t < t1) {
                        const double* As = smem + (s & 1) * STAGE;
                        // set s is free (its tile went to LDS one step ago); past the end these are dummy loads
                        load_tile_nobranch<KCA, BM, BK, NT>(qa[s], oa[s], p.A, p.lda, m0, p.M, (t + PF) * BK, p.K, tid, t + PF < t1);
                        load_tile_nobranch<KCB, BN, BK, NT>(qb[s], ob[s], p.B, p.ldb, n0, p.N, (t + PF) * BK, p.K, tid, t + PF < t1);
                        mfma_stage(As, As + TA::SIZE);
                        {   // tile t+1 (zeros past the end: never read) into the other LDS stage
                            const int sn = (s + 1) % PF;   // a constant once the s loop is unrolled
                            double* An = smem + (sn & 1) * STAGE;
                            store_tile_masked<KCA, BM, BK, NT>(qa[sn], oa[sn], An, tid);
                            store_tile_masked<KCB, BN, BK, NT>(qb[sn], ob[sn], An + TA::SIZE, tid);
                        }
                        __syncthreads();
                    }
                }
            }
            epilogue();
            return;
        }
    }

    v2d ra[TA::NVEC / NT], rb[TB::NVEC / NT];
    load_tile<KCA, BM, BK, NT>(ra, p.A, p.lda, m0, p.M, t0 * BK, p.K, tid);
    load_tile<KCB, BN, BK, NT>(rb, p.B, p.ldb, n0, p.N, t0 * BK, p.K, tid);
    store_tile<KCA, BM, BK, NT>(ra, smem, tid);
    store_tile<KCB, BN, BK, NT>(rb, smem + TA::SIZE, tid);
    __syncthreads();

    for (int t = t0; t < t1; ++t) {
        const double* As = smem + ((t - t0) & 1) * STAGE;
        const double* Bs = As + TA::SIZE;
        const bool more = (t + 1 < t1);
        if (more) {
            load_tile<KCA, BM, BK, NT>(ra, p.A, p.lda, m0, p.M, (t + 1) * BK, p.K, tid);
            load_tile<KCB, BN, BK, NT>(rb, p.B, p.ldb, n0, p.N, (t + 1) * BK, p.K, tid);
        }
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            double a[2][MI], b[2][NI];
            TA::template frags<MI>(As, wm0, k8, l15, l4, a);
            TB::template frags<NI>(Bs, wn0, k8, l15, l4, b);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e][i], b[e][j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            double* An = smem + ((t + 1 - t0) & 1) * STAGE;
            store_tile<KCA, BM, BK, NT>(ra, An, tid);
            store_tile<KCB, BN, BK, NT>(rb, An + TA::SIZE, tid);
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm0 + TA::index_of(i, l4 + 4 * r);
                const int n = n0 + wn0 + TB::index_of(j, l15);
                if (m < p.M && n < p.N) epi(m, n, acc[i][j][r]);
            }
}


// The K loop of one output tile over its K tiles [kt_begin, kt_end) (relative to the tile's own first contributing K
// tile under `tri`), result left in the caller's accumulators: the building block of the two-phase stream-K kernel.
// PF K tiles of global loads are in flight in PF register sets (LDS stays double-buffered).  The stream-K shapes read
// operands that the XCD's L2 rarely holds (1024 x 3072 x 3072 against a triangle: ~90 % of the tile bytes come from the
// Infinity Cache); with one K tile ahead a workgroup has 32 KB in flight and needs it back within one K tile of MFMAs
// (1.7 us) — the loaded Infinity-Cache latency is above that (MI355X_MICROARCH.md, gather table: 72 KB in flight per CU for
// 33 GB/s).  Loads are branch-free (clamped address + zeroing select at the LDS store), so the waits are counted.
// Needs even extents along each operand's contiguous dimension.
template <bool KCA, bool KCB, int BM, int BN, int BK, int WGM, int WGN, int PFX>
__device__ __forceinline__ void gemm_f64_tile_acc(const GemmShape& p, int bm, int bn, double* smem, int kt_begin, int kt_end,
                                                  v4d (&acc)[BM / WGM / 16][BN / WGN / 16]) {
    constexpr int PF = PFX % 10;                  // K tiles of global loads in flight
    constexpr bool kFragPrefetch = PFX >= 10;     // LDS fragments of the next k8 step fetched ahead of the MFMAs
    constexpr int NT = WGM * WGN * 64;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 16, NI = WN / 16;
    using TA = OpTile<KCA, BM, BK>;
    using TB = OpTile<KCB, BN, BK>;
    constexpr int STAGE = TA::SIZE + TB::SIZE;
    constexpr int NA = TA::NVEC / NT, NB = TB::NVEC / NT;
    const int m0 = bm * BM, n0 = bn * BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    int t0 = 0, t1 = (p.K + BK - 1) / BK;
    if (p.tri & 1) t1 = min(t1, (min(p.K, n0 + BN) + BK - 1) / BK);
    if (p.tri & 2) t0 = max(t0, min(n0, p.K) / BK);
    if (p.tri & 4) t1 = min(t1, (min(p.K, m0 + BM) + BK - 1) / BK);
    if (p.tri & 8) t0 = max(t0, min(m0, p.K) / BK);
    t1 = min(t1, t0 + kt_end);
    t0 += kt_begin;
    if (t0 >= t1) return;
    // Interior segment (the tile lies inside the operands and so does every K tile of the run): every 16-byte vector is valid, so
    // the per-step bounds tests, clamped-address selects (64-bit shifts + 8 v_cndmask per vector) and zeroing selects of the
    // generic loop below — ~45 VALU/SALU instructions between the barrier and the first LDS fragment read of a step, during which
    // BOTH waves of a SIMD leave the matrix pipe idle — reduce to one pointer increment per vector.  EMCID_GEMM_FAST=0 (read by the
    // launchers into GemmShape.nofast) keeps the generic loop for A/B runs.
    if (!p.nofast && m0 + BM <= p.M && n0 + BN <= p.N && t1 * BK <= p.K) {
        const double* pa[NA];
        const double* pb[NB];
        int la[NA], lb[NB];                     // LDS offsets (doubles) of this thread's vectors inside a stage
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + i * NT;
            if (KCA) { pa[i] = p.A + (int64_t)(m0 + v / (BK / 2)) * p.lda + 2 * (v % (BK / 2)); la[i] = (v / (BK / 2)) * TA::LD + 2 * (v % (BK / 2)); }
            else     { pa[i] = p.A + (int64_t)(v / (BM / 2)) * p.lda + m0 + 2 * (v % (BM / 2)); la[i] = (v / (BM / 2)) * TA::LD + 2 * (v % (BM / 2)); }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = tid + i * NT;
            if (KCB) { pb[i] = p.B + (int64_t)(n0 + v / (BK / 2)) * p.ldb + 2 * (v % (BK / 2)); lb[i] = (v / (BK / 2)) * TB::LD + 2 * (v % (BK / 2)); }
            else     { pb[i] = p.B + (int64_t)(v / (BN / 2)) * p.ldb + n0 + 2 * (v % (BN / 2)); lb[i] = (v / (BN / 2)) * TB::LD + 2 * (v % (BN / 2)); }
        }
        const int64_t sa = KCA ? BK : (int64_t)BK * p.lda, sb = KCB ? BK : (int64_t)BK * p.ldb;     // per K tile
        v2d fa[PF][NA], fb[PF][NB];
        auto fetch = [&](int s, int t) {        // tile t into register set s; past the end: tile t1 - 1 again (never stored)
            const int64_t tt = t < t1 ? t : t1 - 1;
#pragma unroll
            for (int i = 0; i < NA; ++i) fa[s][i] = *reinterpret_cast<const v2d*>(pa[i] + tt * sa);
#pragma unroll
            for (int i = 0; i < NB; ++i) fb[s][i] = *reinterpret_cast<const v2d*>(pb[i] + tt * sb);
        };
        auto stash = [&](int s, double* stage) {
#pragma unroll
            for (int i = 0; i < NA; ++i) *reinterpret_cast<v2d*>(stage + la[i]) = fa[s][i];
#pragma unroll
            for (int i = 0; i < NB; ++i) *reinterpret_cast<v2d*>(stage + TA::SIZE + lb[i]) = fb[s][i];
        };
#pragma unroll
        for (int s = 0; s < PF; ++s) fetch(s, t0 + s);
        __syncthreads();          // the previous segment's last MFMA stage may still be reading LDS
        stash(0, smem);
        __syncthreads();
        for (int base = t0; base < t1; base += PF) {
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                const int t = base + s;
                if (t < t1) {
                    const double* As = smem + ((t - t0) & 1) * STAGE;
                    const double* Bs = As + TA::SIZE;
                    double a[2][2][MI], b[2][2][NI];
                    TA::template frags<MI>(As, wm0, 0, l15, l4, a[0]);        // the step's first fragments BEFORE anything else
                    TB::template frags<NI>(Bs, wn0, 0, l15, l4, b[0]);
                    fetch(s, t + PF);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k8 = 0; k8 < BK / 8; ++k8) {
                        if (k8 + 1 < BK / 8) {
                            TA::template frags<MI>(As, wm0, k8 + 1, l15, l4, a[(k8 + 1) & 1]);
                            TB::template frags<NI>(Bs, wn0, k8 + 1, l15, l4, b[(k8 + 1) & 1]);
                        }
#pragma unroll
                        for (int e = 0; e < 2; ++e)
#pragma unroll
                            for (int i = 0; i < MI; ++i)
#pragma unroll
                                for (int j = 0; j < NI; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k8 & 1][e][i], b[k8 & 1][e][j], acc[i][j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 1 < t1) stash((s + 1) % PF, smem + ((t + 1 - t0) & 1) * STAGE);
                    __syncthreads();
                }
            }
        }
        return;
    }
    v2d qa[PF][NA], qb[PF][NB];
    bool oa[PF][NA], ob[PF][NB];
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        load_tile_nobranch<KCA, BM, BK, NT>(qa[s], oa[s], p.A, p.lda, m0, p.M, (t0 + s) * BK, p.K, tid, t0 + s < t1);
        load_tile_nobranch<KCB, BN, BK, NT>(qb[s], ob[s], p.B, p.ldb, n0, p.N, (t0 + s) * BK, p.K, tid, t0 + s < t1);
    }
    __syncthreads();          // the previous segment's last MFMA stage may still be reading LDS
    store_tile_masked<KCA, BM, BK, NT>(qa[0], oa[0], smem, tid);
    store_tile_masked<KCB, BN, BK, NT>(qb[0], ob[0], smem + TA::SIZE, tid);
    __syncthreads();
    for (int base = t0; base < t1; base += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int t = base + s;          // tile t sits in LDS stage (t - t0) & 1; its register set s is free again
            if (t < t1) {
                const double* As = smem + ((t - t0) & 1) * STAGE;
                const double* Bs = As + TA::SIZE;
                load_tile_nobranch<KCA, BM, BK, NT>(qa[s], oa[s], p.A, p.lda, m0, p.M, (t + PF) * BK, p.K, tid, t + PF < t1);
                load_tile_nobranch<KCB, BN, BK, NT>(qb[s], ob[s], p.B, p.ldb, n0, p.N, (t + PF) * BK, p.K, tid, t + PF < t1);
                // keep the loads HERE: left alone, hipcc sinks them below the MFMAs into the registers the LDS store of
                // this step frees, which puts them back to one K tile ahead of their use
                if (PF > 1) __builtin_amdgcn_sched_barrier(0);
                if (kFragPrefetch) {
                    // fragments of k8 + 1 are fetched from LDS BEFORE the MFMAs of k8 (a second register set): the two
                    // waves of a SIMD leave the barrier together, so an LDS read that both wait for is a hole in the
                    // matrix pipe; only the first read of a K tile stays exposed
                    double a[2][2][MI], b[2][2][NI];
                    TA::template frags<MI>(As, wm0, 0, l15, l4, a[0]);
                    TB::template frags<NI>(Bs, wn0, 0, l15, l4, b[0]);
#pragma unroll
                    for (int k8 = 0; k8 < BK / 8; ++k8) {
                        if (k8 + 1 < BK / 8) {
                            TA::template frags<MI>(As, wm0, k8 + 1, l15, l4, a[(k8 + 1) & 1]);
                            TB::template frags<NI>(Bs, wn0, k8 + 1, l15, l4, b[(k8 + 1) & 1]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int e = 0; e < 2; ++e)
#pragma unroll
                            for (int i = 0; i < MI; ++i)
#pragma unroll
                                for (int j = 0; j < NI; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k8 & 1][e][i], b[k8 & 1][e][j], acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                for (int k8 = 0; k8 < BK / 8; ++k8) {
                    double a[2][MI], b[2][NI];
                    TA::template frags<MI>(As, wm0, k8, l15, l4, a);
                    TB::template frags<NI>(Bs, wn0, k8, l15, l4, b);
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int i = 0; i < MI; ++i)
#pragma unroll
                            for (int j = 0; j < NI; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e][i], b[e][j], acc[i][j], 0, 0, 0);
                }
                }
                if (PF > 1) __builtin_amdgcn_sched_barrier(0);   // ... and the zeroing selects of the set stored next BELOW them
                {   // tile t+1 (zeros past the end: never read) into the other LDS stage
                    const int sn = (s + 1) % PF;
                    double* An = smem + ((t + 1 - t0) & 1) * STAGE;
                    store_tile_masked<KCA, BM, BK, NT>(qa[sn], oa[sn], An, tid);
                    store_tile_masked<KCB, BN, BK, NT>(qb[sn], ob[sn], An + TA::SIZE, tid);
                }
                __syncthreads();
            }
        }
    }
}

template <bool KCA, bool KCB, int BM, int BN, int BK, int WGM, int WGN, class Epi>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_f64_kernel(GemmShape p, Epi epi) {
    const int zb = blockIdx.z / p.ksplit, zs = blockIdx.z % p.ksplit;
    const int z1 = zb % p.batch, z2 = zb / p.batch;
    p.A += (int64_t)z1 * p.sA + (int64_t)z2 * p.sA2;
    p.B += (int64_t)z1 * p.sB + (int64_t)z2 * p.sB2;
    epi.batch(z1);
    epi.batch2(z2);
    using TA = OpTile<KCA, BM, BK>;
    using TB = OpTile<KCB, BN, BK>;
    __shared__ __attribute__((aligned(16))) double smem[2 * (TA::SIZE + TB::SIZE)];
    if (p.pair) {
        // triangular K ranges: the workgroup takes column (tri & 3) or row (tri & 12) tile j and its mirror image
        // ntiles-1-j one after the other — every workgroup then contracts over the same total depth
        const bool by_n = (p.tri & 3) != 0;
        const int nt = by_n ? (p.N + BN - 1) / BN : (p.M + BM - 1) / BM;
        const int j = by_n ? blockIdx.x : blockIdx.y;
        const int first = (p.tri & (by_n ? 1 : 4)) ? nt - 1 - j : j;       // the long range first
        const int second = nt - 1 - first;
        gemm_f64_tile<KCA, KCB, BM, BN, BK, WGM, WGN>(p, epi, by_n ? (int)blockIdx.y : first, by_n ? first : (int)blockIdx.x, zs, smem);
        if (second != first)
            gemm_f64_tile<KCA, KCB, BM, BN, BK, WGM, WGN>(p, epi, by_n ? (int)blockIdx.y : second, by_n ? second : (int)blockIdx.x, zs, smem);
        return;
    }
    // Workgroups go to the 8 XCDs round-robin by linear id (x fastest).  When the K range depends on the tile COLUMN
    // (B-side triangle) and the column count is a multiple of 8, XCD = column % 8 and one XCD owns all the long columns
    // (8 columns of a 512-block: 8x the work of the XCD with the short ones) — number the tile ROWS fastest instead, so
    // that every XCD sees every column.  `xcd_rows` (set by the launcher, env EMCID_GEMM_XCD_ROWS=0 disables) selects it.
    if (p.lo_total > 0) {
        // Lower-only square output: with the (bn, bm) grid the XCD (= linear id % 8 = bn % 8 when the column count is a
        // multiple of 8) that owns tile column 0 gets 17 % more tiles than the average (80 x 40 tiles of 32 x 64) and the
        // launch lasts as long as that XCD.  Number only the tiles that exist, row by row: consecutive ids alternate
        // over the XCDs and share their A rows.  Row bm has bm / r + 1 tiles (r = BN / BM).
        constexpr int r = BN / BM;
        const int L = blockIdx.x + gridDim.x * blockIdx.y;
        if (L >= p.lo_total) return;
        int q = (int)((__builtin_sqrt(8.0 * L / r + 1.0) - 1.0) * 0.5);
        while (r * (q + 1) * (q + 2) / 2 <= L) ++q;      // guard the rounding of the square root
        while (r * q * (q + 1) / 2 > L) --q;
        const int rem = L - r * q * (q + 1) / 2;
        gemm_f64_tile<KCA, KCB, BM, BN, BK, WGM, WGN>(p, epi, r * q + rem / (q + 1), rem % (q + 1), zs, smem);
        return;
    }
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_rows) {
        const int lin = bx + gridDim.x * by;
        by = lin % gridDim.y;
        bx = lin / gridDim.y;
    }
    // tiles whose K range grows with n (tri & 1) or m (tri & 4) are numbered from the far end: long ranges start first
    const int bm = (p.tri & 4) ? gridDim.y - 1 - by : by;
    const int bn = (p.tri & 1) ? gridDim.x - 1 - bx : bx;
    gemm_f64_tile<KCA, KCB, BM, BN, BK, WGM, WGN>(p, epi, bm, bn, zs, smem);
}

// ---- epilogues -------------------------------------------------------------------------------

// C = alpha * D + beta * C
struct EpiAxpby {
    double* C; int64_t ldc; double alpha, beta; int64_t sC = 0; int atomic = 0; int64_t sC2 = 0;
    __device__ __forceinline__ void batch(int z) { C += (int64_t)z * sC; }
    __device__ __forceinline__ void batch2(int z) { C += (int64_t)z * sC2; }
    __device__ __forceinline__ void operator()(int m, int n, double v) const {
        double* c = C + (int64_t)m * ldc + n;
        if (atomic) unsafeAtomicAdd(c, alpha * v);   // split-K partial of C += alpha * D (global_atomic_add_f64)
        else *c = (beta != 0.0) ? alpha * v + beta * *c : alpha * v;
    }
};

// A = lam * double(fl32(fl32(C * cw) / 0.5f)) + D inside [0,d)^2 ; identity outside (padding to dp).
// reference: emcid/emcid_main.py:1037 (`cov * (1 - edit_weight) / 0.5`, fp32) and :1046 (`lam * cov.double() + K K^T`).
struct EpiAssemble {
    const float* Cf; int64_t ldcf; double lam; float cw; double* A; int64_t lda; int d;
    __device__ __forceinline__ void batch(int) {}
    __device__ __forceinline__ void batch2(int) {}
    __device__ __forceinline__ void operator()(int m, int n, double v) const {
        double out;
        if (m < d && n < d) {
            const float c1 = Cf[(int64_t)m * ldcf + n] * cw;
            const float c2 = c1 / 0.5f;
            out = lam * (double)c2 + v;
        } else {
            out = (m == n) ? 1.0 : 0.0;
        }
        A[(int64_t)m * lda + n] = out;
    }
};

// U = D (f64, optional); dW = float(D) (optional); W = W0 + float(D) (optional).
// reference: emcid/emcid_main.py:1050 (`resid @ adj_k.T`) and :1061 (`weights_copy + upd_matrix.float()`).
struct EpiDeltaW {
    const float* W0; float* W; int64_t ldw; float* dW; int64_t lddw; double* U; int64_t ldu;
    __device__ __forceinline__ void batch(int) {}
    __device__ __forceinline__ void batch2(int) {}
    __device__ __forceinline__ void operator()(int m, int n, double v) const {
        const float f = (float)v;
        if (U) U[(int64_t)m * ldu + n] = v;
        if (dW) dW[(int64_t)m * lddw + n] = f;
        if (W) W[(int64_t)m * ldw + n] = W0[(int64_t)m * ldw + n] + f;
    }
};

// ---- stream-K for triangular contractions -------------------------------------------------------------------------
// With a triangular operand the K depth of an output tile grows linearly along one tile dimension, so "one tile per
// workgroup" leaves the chip waiting for the deepest tiles, and small tiles (for balance) give up the efficiency of the
// 128x128 configuration.  Here the (tile, K-step) space is linearised — column tile major, 16-deep K steps — and cut into
// equal runs, one per workgroup: a run covers the tail of one tile, some whole tiles, the head of another.  (Round 1 added
// the partial tiles into a zeroed C with f64 atomics; that form is gone: the two-phase form below is faster and
// bit-reproducible.)  Only B-side triangles (tri = 1 or 2) and the f64 EpiAxpby epilogue.
__device__ __forceinline__ int streamk_depth(const GemmShape& p, int bn, int BN, int BK) {
    const int KT = (p.K + BK - 1) / BK;
    if (p.tri & 1) return min(KT, (min(p.K, (bn + 1) * BN) + BK - 1) / BK);
    if (p.tri & 2) return KT - min(bn * BN, p.K) / BK;
    return KT;
}

// ---- stream-K: no atomics, no zero fill, bit-reproducible ---------------------------------------------------
// Same partition as above (the (tile, K-step) space cut into equal runs), but a run's partial tiles go to a workspace slot
// (lane-linear image of the accumulators, 16-byte write-through stores) instead of being added into C with f64 atomics:
//   * a run holds at most two partial segments: its first (the tail of a tile an earlier run started) and its last (the
//     head of a tile a later run continues) -> slots 2r and 2r+1 of `partials` (BM*BN doubles each);
//   * every contributor of a tile publishes its slot, then takes a ticket on the tile's counter; the one whose ticket is
//     the last re-reads ALL slots of the tile in run order — its own included, so the sum order never depends on who came
//     last — applies the epilogue and puts the counter back to zero (the workspace is ready for the next launch);
//   * nobody waits for anybody: no spinning, no co-residency assumption.
// Atomics execute at the memory side at ~1.3 TB/s chip-wide (MI355X_MICROARCH.md): the 256 runs of the first form all end at
// the same time and leave ~33 MB of them behind as a tail; write-through stores move the same bytes at ~6 TB/s, and the
// output needs no zero-filling launch.  Cross-CU hand-off follows cdna_hip_programming.md Guideline 16 (R1), write-through form: sc1 stores ->
// every wave's s_waitcnt vmcnt(0) -> barrier -> lane 0 relaxed agent-scope ticket; the last arriver: ticket -> lane 0
// acquire fence (agent) -> s_waitcnt -> barrier -> plain loads.
// Runs are dealt to workgroups so that the 32 workgroups of an XCD (blockIdx % 8) hold 32 CONSECUTIVE runs: they work
// on the same one or two column tiles at the same time, whose B panel (<= 3 MB) then stays in that XCD's 4 MB L2.
struct StreamKWork {
    double* partials;        // [2 * runs][BM * BN]
    unsigned* counters;      // [tiles], zero on entry, zero again on exit
    double diag_add;         // added to C(i, i) by the epilogue (the identity of S = I + Yt Yt^T); 0 otherwise
    long long* stamps;       // diagnostic (usually null): 8 shader-clock values per workgroup, see scripts/streamk_stamps.py
};

template <int MI, int NI>
__device__ __forceinline__ void streamk_store_partial(double* slot, const v4d (&acc)[MI][NI], int tid, int nthreads) {
    // lane-linear: vector v = (i * NI + j) * 2 + half holds registers {2 half, 2 half + 1} of tile (i, j)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                v2d x = {acc[i][j][2 * hf], acc[i][j][2 * hf + 1]};
                double* q = slot + ((int64_t)((i * NI + j) * 2 + hf) * nthreads + tid) * 2;
                // write-through (sc1): the bytes leave the XCD's L2 as they are stored, so publishing them needs no
                // agent-scope release (an L2 write-back of everything the workgroup's XCD has dirtied: 2-6 us)
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(q), "v"(x) : "memory");
            }
}

template <int MI, int NI>
__device__ __forceinline__ void streamk_add_partial(const double* slot, v4d (&acc)[MI][NI], int tid, int nthreads) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const v2d x = *reinterpret_cast<const v2d*>(slot + ((int64_t)((i * NI + j) * 2 + hf) * nthreads + tid) * 2);
                acc[i][j][2 * hf] += x[0];
                acc[i][j][2 * hf + 1] += x[1];
            }
}

template <bool KCA, bool KCB, int BM, int BN, int BK, int WGM, int WGN, int PF>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_f64_streamk2_kernel(GemmShape p, EpiAxpby epi, StreamKWork w, long long total_units,
                                                                            long long units_per_wg, int xcd_map) {
    using TA = OpTile<KCA, BM, BK>;
    using TB = OpTile<KCB, BN, BK>;
    constexpr int NT = WGM * WGN * 64;
    constexpr int MI = BM / WGM / 16, NI = BN / WGN / 16;
    __shared__ __attribute__((aligned(16))) double smem[2 * (TA::SIZE + TB::SIZE) + (KCB ? 0 : 2)];
    // "I took the last ticket" flag: with a K-contiguous B image the last double of the array is row padding nobody reads
    // (a separate __shared__ word would push 2 x 80 KB over the CU's 160 KB); otherwise two spare doubles behind it
    volatile int& s_last = *reinterpret_cast<volatile int*>(smem + (KCB ? 2 * (TA::SIZE + TB::SIZE) - 1 : 2 * (TA::SIZE + TB::SIZE)));
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WGN) * (BM / WGM), wn0 = (wave % WGN) * (BN / WGN);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int MT = (p.M + BM - 1) / BM, NTL = (p.N + BN - 1) / BN;
    const int KT = (p.K + BK - 1) / BK;
    const int G = gridDim.x;
    long long run = blockIdx.x;
    if (xcd_map && G % 8 == 0) run = (long long)(blockIdx.x % 8) * (G / 8) + blockIdx.x / 8;
    long long u = run * units_per_wg;
    const long long u0 = u;
    const long long u1 = min(total_units, u + units_per_wg);
    GemmShape q = p;
    q.lower_only = 0;
    int bn = 0, bm = 0;
    long long base = 0;      // first unit of the current tile (lower_only) or column tile (triangular)
    long long st_begin = 0, st_loop = 0, st_publish = 0, st_reduce = 0, st_epi = 0, st_segments = 0, st_mark = 0;
    if (w.stamps) st_begin = st_mark = __builtin_amdgcn_s_memtime();
#define EMCID_SK_LAP(acc_var)                                                \
    if (w.stamps) {                                                          \
        const long long now_ = __builtin_amdgcn_s_memtime();                 \
        acc_var += now_ - st_mark;                                           \
        st_mark = now_;                                                      \
    }
    if (!p.lower_only) {
        while (bn < NTL) {
            const long long span = (long long)MT * streamk_depth(p, bn, BN, BK);
            if (u < base + span) break;
            base += span;
            ++bn;
        }
    }
    while (u < u1) {
        int depth, tile_id;
        long long tile_begin;
        if (p.lower_only) {
            // SYRK-like: the lower tiles (bn <= bm) of a square output, every tile KT deep, numbered row by row
            const long long t = u / KT;
            bm = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
            while ((long long)(bm + 1) * (bm + 2) / 2 <= t) ++bm;
            while ((long long)bm * (bm + 1) / 2 > t) --bm;
            bn = (int)(t - (long long)bm * (bm + 1) / 2);
            depth = KT;
            tile_begin = t * KT;
            tile_id = (int)t;
        } else {
            if (bn >= NTL) break;
            depth = streamk_depth(p, bn, BN, BK);
            const long long span = (long long)MT * depth;
            if (depth == 0 || u >= base + span) { base += span; ++bn; continue; }
            bm = (int)((u - base) / depth);
            tile_begin = base + (long long)bm * depth;
            tile_id = bn * MT + bm;
        }
        const int k_begin = (int)(u - tile_begin);
        const int k_end = (int)min((long long)depth, k_begin + (u1 - u));
        v4d acc[MI][NI];
        gemm_f64_tile_acc<KCA, KCB, BM, BN, BK, WGM, WGN, PF>(q, bm, bn, smem, k_begin, k_end, acc);
        EMCID_SK_LAP(st_loop)
        ++st_segments;
        bool finish = (k_begin == 0 && k_end == depth);
        if (!finish) {
            const long long r_first = tile_begin / units_per_wg, r_last = (tile_begin + depth - 1) / units_per_wg;
            const int slot = (int)(2 * run + (u == u0 ? 0 : 1));
            streamk_store_partial<MI, NI>(w.partials + (int64_t)slot * BM * BN, acc, tid, NT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned old = __hip_atomic_fetch_add(w.counters + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = (old == (unsigned)(r_last - r_first)) ? 1 : 0;
            }
            __syncthreads();
            EMCID_SK_LAP(st_publish)
            if (s_last) {
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
                for (long long r = r_first; r <= r_last; ++r) {
                    // run r's segment of this tile starts at the run's own start unless the tile begins inside the run
                    const int sl = (int)(2 * r + ((r == r_first && tile_begin != r * units_per_wg) ? 1 : 0));
                    streamk_add_partial<MI, NI>(w.partials + (int64_t)sl * BM * BN, acc, tid, NT);
                }
                if (tid == 0) w.counters[tile_id] = 0;
                finish = true;
            }
            __syncthreads();       // s_last is rewritten by the next segment
            EMCID_SK_LAP(st_reduce)
        }
        if (finish) {
            const int m0 = bm * BM, n0 = bn * BN;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + wm0 + TA::index_of(i, l4 + 4 * r);
                        const int n = n0 + wn0 + TB::index_of(j, l15);
                        if (m < p.M && n < p.N) {
                            double v = epi.alpha * acc[i][j][r];
                            if (m == n) v += w.diag_add;
                            epi.C[(int64_t)m * epi.ldc + n] = v;
                        }
                    }
            EMCID_SK_LAP(st_epi)
        }
        u += k_end - k_begin;
    }
    if (w.stamps && tid == 0) {
        long long* o = w.stamps + (int64_t)blockIdx.x * 8;
        o[0] = st_begin; o[1] = __builtin_amdgcn_s_memtime(); o[2] = st_loop; o[3] = st_publish; o[4] = st_reduce; o[5] = st_epi;
        o[6] = st_segments; o[7] = (long long)run;
    }
#undef EMCID_SK_LAP
}

inline long long* g_streamk_stamps = nullptr;      // diagnostic target, set by emcid_debug_streamk_stamps

inline int64_t streamk_workspace_doubles(int wgs) { return (int64_t)2 * wgs * 128 * 128 + 8192; }   // slots + counters (as doubles)

// C = alpha * A B (+ diag_add on the diagonal); C needs no initial value.  `work`: streamk_workspace_doubles(wgs) doubles whose
// counter part (the last 8192 doubles) is zero (it is left zero).
template <bool KCA, bool KCB, int BK, int BM, int BN, int WGM, int WGN>
inline void launch_gemm_f64_streamk2_tile(GemmShape p, EpiAxpby epi, hipStream_t stream, int wgs_alloc, int runs_per_cu, double* work,
                                          double diag_add) {
    const int MT = (p.M + BM - 1) / BM, NTL = (p.N + BN - 1) / BN, KT = (p.K + BK - 1) / BK;
    long long total = 0;
    if (p.lower_only) total = (long long)MT * (MT + 1) / 2 * KT;
    else for (int bn = 0; bn < NTL; ++bn) {
        int depth = KT;
        if (p.tri & 1) { const int ke = p.K < (bn + 1) * BN ? p.K : (bn + 1) * BN; depth = (ke + BK - 1) / BK; if (depth > KT) depth = KT; }
        else if (p.tri & 2) { const int kb = bn * BN < p.K ? bn * BN : p.K; depth = KT - kb / BK; }
        total += (long long)MT * depth;
    }
    if (total <= 0) return;
    // runs: wgs_alloc * runs_per_cu, as long as their 2 partial slots each fit what the caller allocated for 128 x 128 tiles
    int wgs = wgs_alloc * runs_per_cu;
    while (wgs > wgs_alloc && (long long)2 * wgs * BM * BN > (long long)2 * wgs_alloc * 128 * 128) wgs -= wgs_alloc;
    long long per = (total + wgs - 1) / wgs;
    if (per < 128 / BK) per = 128 / BK;                      // never less than a 128-deep run per workgroup
    unsigned grid = (unsigned)((total + per - 1) / per);
    int map = 1;
    if (grid % 8) {                    // keep the XCD dealing exact: round the grid up (trailing runs are empty)
        const unsigned g8 = (grid + 7) / 8 * 8;
        if ((int)g8 <= wgs) grid = g8; else map = 0;
    }
    // the ticket counters sit behind the slot area of the ALLOCATION (fixed place, whatever tile shape a launch uses)
    StreamKWork w{work, reinterpret_cast<unsigned*>(work + (int64_t)2 * wgs_alloc * 128 * 128), diag_add, g_streamk_stamps};
    p.nofast = 0;
    // 11: one K tile of global loads in flight + LDS fragments of the next k8 step fetched ahead.  With the interior fast path
    // this instantiation runs the SYRK 20 % faster than the plain one (80 vs 100 us) and the triangular GEMMs 1-2 % faster
    constexpr int NTH = WGM * WGN * 64;
    hipLaunchKernelGGL((gemm_f64_streamk2_kernel<KCA, KCB, BM, BN, BK, WGM, WGN, 11>), dim3(grid), dim3(NTH), 0, stream, p, epi, w, total, per, map);
}

template <bool KCA, bool KCB, int BK>
inline void launch_gemm_f64_streamk2_bk(GemmShape p, EpiAxpby epi, hipStream_t stream, int wgs, double* work, double diag_add) {
    // Measured (scripts/mb_tri.py, us: Yt = Kt X^T / U = G X / S = I + Yt Yt^T): 128 x 128 214 / 168 / 81, 64 x 128 270 / 222 / -,
    // 64 x 64 243 / 217 / 74: the smaller tiles' independent barriers do not make up for their operand traffic, except on the
    // SYRK's 36 big tiles and on products with at most 128 rows (a 100-concept edit: ONE row of 128 x 128 tiles, every tile cut
    // over ~10 workgroups: inv_apply 0.575 -> 0.509 ms per 100-concept call): those take 64 x 64.
    if constexpr (BK == 16) {
        if (p.lower_only || p.M <= 128) {
            launch_gemm_f64_streamk2_tile<KCA, KCB, BK, 64, 64, 2, 2>(p, epi, stream, wgs, 3, work, diag_add);
            return;
        }
    }
    launch_gemm_f64_streamk2_tile<KCA, KCB, BK, 128, 128, 2, 4>(p, epi, stream, wgs, 1, work, diag_add);
}

// C = alpha * A B (+ diag_add on the diagonal); C needs no initial value.  `work`: streamk_workspace_doubles(wgs) doubles whose
// counter part (the last 8192 doubles) is zero (it is left zero).
// one ticket counter per 128 x 128 output tile (the 64 x 64 form is only taken at M <= 128 or on lower tiles, never past this)
inline bool streamk2_fits(int64_t M, int64_t N) { return ((M + 127) / 128) * ((N + 127) / 128) <= 16384; }

// Returns false (nothing launched) when the product has more tiles than ticket counters: the caller takes another form.
template <bool KCA, bool KCB>
[[nodiscard]] inline bool launch_gemm_f64_streamk2(GemmShape p, EpiAxpby epi, hipStream_t stream, int wgs, double* work,
                                                   double diag_add = 0.0) {
    if (!streamk2_fits(p.M, p.N)) return false;
    launch_gemm_f64_streamk2_bk<KCA, KCB, 16>(p, epi, stream, wgs, work, diag_add);
    return true;
}

// ---- launcher ----------------------------------------------------------------------------------

template <class Epi> inline bool epi_accumulates(const Epi&) { return false; }
inline bool epi_accumulates(const EpiAxpby& e) { return e.beta == 1.0; }
template <class Epi> inline void epi_set_atomic(Epi&) {}
inline void epi_set_atomic(EpiAxpby& e) { e.atomic = 1; }

template <bool KCA, bool KCB, class Epi>
inline void launch_gemm_f64(GemmShape p, Epi epi, hipStream_t stream, int force_cfg = -1) {
    // cfg 0: 128x128 tile, 8 waves  — when that many tiles still fill the chip
    // cfg 1: 64x64 tile, 4 waves    — 4x the workgroups
    // cfg 2: 32x64 tile, 4 waves    — skinny problems (M ~ number of concepts); needs a K-contiguous A operand
    auto tiles = [&](int64_t bm, int64_t bn) {   // output tiles that are actually computed (estimate for lower_only)
        const int64_t tm = (p.M + bm - 1) / bm, tn = (p.N + bn - 1) / bn;
        const int64_t sq = tn * bn < tm * bm ? tn * bn : tm * bm;          // edge of the square part on the diagonal
        const int64_t skipped = p.lower_only ? (sq / bm) * (sq / bn) / 2 : 0;
        return (tm * tn - skipped) * p.batch * p.batch2;
    };
    const int64_t big_tiles = tiles(128, 128), mid_tiles = tiles(64, 64);
    // measured on the M ~ 1000 solve shapes (scripts/mb_shapes.py): 32x64 beats 64x64 whenever 128x128 cannot fill the chip
    // (re-measured with the prefetch ring in the small-tile kernels: 32x64 also wins between 224 and 512 big tiles, e.g.
    // the batched Cholesky trailing updates: chol_trail 0.93 -> 0.86 ms per step)
    int cfg = (big_tiles >= 512) ? 0 : (KCA ? 2 : 1);
    if (force_cfg >= 0) cfg = force_cfg;
    if (cfg == 2 && !KCA) cfg = 1;
    // too few workgroups to hide the global->LDS latency of a shallow tile: split K (accumulating epilogues only)
    const int64_t wgs = cfg == 0 ? big_tiles : cfg == 1 ? mid_tiles : tiles(32, 64);
    const int ktiles = (p.K + 15) / 16;
    if (p.ksplit == 1 && epi_accumulates(epi) && cfg != 0 && wgs < 512 && ktiles >= 16) {
        const int64_t want = (768 + wgs - 1) / wgs, cap = ktiles / 8;
        p.ksplit = (int)(want < cap ? want : cap);
        if (p.ksplit < 1) p.ksplit = 1;
    }
    if (p.kchunk > 0) {   // fixed-length runs: as many z-slices as the longest K range needs
        p.ksplit = epi_accumulates(epi) ? (ktiles + p.kchunk - 1) / p.kchunk : 1;
        if (p.ksplit == 1) p.kchunk = 0;
    }
    if (p.ksplit > 1) epi_set_atomic(epi);
    const unsigned gz = (unsigned)(p.batch * p.batch2 * p.ksplit);
    if (p.pair && (p.tri == 0 || p.lower_only)) p.pair = 0;
    p.xcd_rows = ((p.tri & 3) && !(p.tri & 12) && !p.lower_only && !p.pair) ? 1 : 0;
    // prefetch ring for the small-tile configurations: needs an even extent along each operand's contiguous dimension
    p.pf = (cfg != 0 && ((KCA ? p.K : p.M) % 2 == 0) && ((KCB ? p.K : p.N) % 2 == 0)) ? 1 : 0;
    p.nofast = 0;
    if (p.lower_only && !p.pair && p.tri == 0 && p.M == p.N && p.lower_shift == 0) {
        const int bm_ = cfg == 0 ? 128 : cfg == 1 ? 64 : 32, r_ = (cfg == 0 ? 128 : 64) / bm_;
        const int gy_ = (p.M + bm_ - 1) / bm_, q_ = gy_ / r_, s_ = gy_ % r_;
        p.lo_total = r_ * q_ * (q_ + 1) / 2 + s_ * (q_ + 1);
    }
    const bool pair_n = p.pair && (p.tri & 3), pair_m = p.pair && !(p.tri & 3);
    auto half = [](unsigned n, bool h) { return h ? (n + 1) / 2 : n; };
    if (cfg == 0) {
        dim3 grid(half((p.N + 127) / 128, pair_n), half((p.M + 127) / 128, pair_m), gz);
        hipLaunchKernelGGL((gemm_f64_kernel<KCA, KCB, 128, 128, 16, 2, 4, Epi>), grid, dim3(512), 0, stream, p, epi);
    } else if (cfg == 1) {
        dim3 grid(half((p.N + 63) / 64, pair_n), half((p.M + 63) / 64, pair_m), gz);
        hipLaunchKernelGGL((gemm_f64_kernel<KCA, KCB, 64, 64, 16, 2, 2, Epi>), grid, dim3(256), 0, stream, p, epi);
    } else {
        if constexpr (KCA) {
            dim3 grid(half((p.N + 63) / 64, pair_n), half((p.M + 31) / 32, pair_m), gz);
            hipLaunchKernelGGL((gemm_f64_kernel<KCA, KCB, 32, 64, 16, 2, 2, Epi>), grid, dim3(256), 0, stream, p, epi);
        }
    }
}

}  // namespace emcid
