// Stage 2 of EMCID on gfx950: assemble A = lam*C' + K K^T, blocked fp64 Cholesky, blocked TRSM with the
// N concept columns as right-hand sides, dW = R X^T and the in-place fp32 weight update.
// Replaces emcid/emcid_main.py:1016-1061 of the reference (torch.linalg.solve + `@` in fp64).
//
// Everything dense runs through gemm_f64.h (v_mfma_f64_16x16x4_f64).  The Cholesky is right-looking
// with NB = 128: a single-workgroup LDS leaf factors the diagonal block and inverts it, so the panel
// solve and every later triangular solve are MFMA GEMMs against the inverted diagonal blocks.
#include <functional>
#include <mutex>

#include "common.h"
#include "gemm_f64.h"

namespace emcid {

thread_local char g_last_error[512] = "";

// ---- profiling state ------------------------------------------------------------------------------------
namespace {
constexpr int PROF_MAX = 16384;
unsigned g_prof_mask = 0;
int g_prof_n = 0;
hipEvent_t g_prof_ev[PROF_MAX][2];
int g_prof_cls[PROF_MAX];
bool g_prof_init = false;
}  // namespace

void prof_begin(int cls, hipStream_t st) {
    if (!(g_prof_mask & (1u << cls)) || g_prof_n >= PROF_MAX) return;
    g_prof_cls[g_prof_n] = cls;
    (void)hipEventRecord(g_prof_ev[g_prof_n][0], st);
}
void prof_end(int cls, hipStream_t st) {
    if (!(g_prof_mask & (1u << cls)) || g_prof_n >= PROF_MAX) return;
    (void)hipEventRecord(g_prof_ev[g_prof_n][1], st);
    ++g_prof_n;
}

// ---- element-wise preparation ---------------------------------------------------------------------

// Kt64[n][j] = double(K[n][j]) * s * g (zero padded to [Np][dp]);
// Rt[n][i]   = double(zs_t[n][i] - Zc[n][i]) * s / layers_left * g  (fp32 subtract first, like the reference).
// g = 1 for the direct solver.  The dual solver passes g = sqrt(lam_factored / lam): it then works with a factor of
// lam_factored * C' whatever the call's lam is (chol(lam C') = sqrt(lam) chol(C')), see emcid_factor_cov_f64.
__global__ __launch_bounds__(256) void prep_kr_kernel(const float* __restrict__ K, const float* __restrict__ Zc,
                                                       const float* __restrict__ zs_t, int N, int d, int h, double s,
                                                       double layers_left, double* __restrict__ Kt64, int Np, int dp,
                                                       double* __restrict__ Rt, int hp, double g = 1.0) {
    const int n = blockIdx.x;
    for (int j = threadIdx.x; j < dp; j += 256) {
        double v = 0.0;
        if (n < N && j < d) v = (double)K[(int64_t)n * d + j] * s * g;
        Kt64[(int64_t)n * dp + j] = v;
    }
    if (Rt) {
        for (int i = threadIdx.x; i < hp; i += 256) {
            double v = 0.0;
            if (n < N && i < h) {
                const float src = zs_t[(int64_t)n * h + i] - Zc[(int64_t)n * h + i];
                v = ((double)src * s) / layers_left * g;
            }
            Rt[(int64_t)n * hp + i] = v;
        }
    }
}

__global__ __launch_bounds__(256) void copy2d_f64_kernel(const double* __restrict__ src, int64_t lds_, double* __restrict__ dst,
                                                          int64_t ldd, int rows, int cols, double scale = 1.0) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += 256) dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds_ + c] * scale;
}

// zero fill by kernel: a hipMemset node inside a captured graph binds the allocation object of capture time, which
// goes stale when the caller's allocator recycles the address range; a kernel only carries the raw pointer
__global__ __launch_bounds__(256) void zero_f64_kernel(double* __restrict__ p, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) p[i] = 0.0;
}

__global__ __launch_bounds__(256) void axpy_f32_kernel(float* __restrict__ W, const float* __restrict__ dW, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) W[i] += dW[i];
}

// W = W0 + float(U) ; dW = float(U)   (after the partial U of the concept shards were summed)
__global__ __launch_bounds__(256) void apply_u_kernel(const double* __restrict__ U, const float* __restrict__ W0,
                                                       float* __restrict__ W, float* __restrict__ dW, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const float f = (float)U[i];
        if (dW) dW[i] = f;
        if (W) W[i] = W0[i] + f;
    }
}

// W = W0 + float(U), dW = float(U) with U[h][ldu] (the apply-only dual path leaves U with the padded leading dimension)
__global__ __launch_bounds__(256) void apply_u2d_kernel(const double* __restrict__ U, int64_t ldu, const float* __restrict__ W0,
                                                         float* __restrict__ W, float* __restrict__ dW, int d) {
    const int i = blockIdx.x;
    for (int j = threadIdx.x; j < d; j += 256) {
        const float f = (float)U[(int64_t)i * ldu + j];
        if (dW) dW[(int64_t)i * d + j] = f;
        if (W) W[(int64_t)i * d + j] = W0[(int64_t)i * d + j] + f;
    }
}

// ---- diagonal leaf: Cholesky of one NB x NB block + its inverse, one workgroup ---------------------------
//
// The leaf is the serial spine of the factorization (d sequential pivots), so it is built for latency:
//  * the 128-block is processed as four 32-column panels;
//  * a panel's 32x32 diagonal block is factored by ONE wave entirely in registers: lane i < 32 holds row i,
//    and lanes 32..63 hold the rows of an identity appended below it.  Running the same right-looking
//    elimination on the augmented [A; I] leaves L in the top lanes and Z = L^-T in the bottom lanes
//    ([A; I] = [L; Z] L^T), so the block's inverse costs no extra instructions.  Pivot rows are broadcast
//    with v_readlane (lane index is a compile-time constant after unrolling); 1/sqrt(pivot) is v_rsq_f64
//    plus two Newton steps;
//  * the rows below the diagonal block (panel solve P = A_below * Z) and the in-leaf trailing update
//    (T -= P P^T) are f64 MFMAs straight out of the LDS image, all four waves;
//  * the inverse of the whole 128-block is assembled from the four 32x32 inverses by two levels of
//    inv([[A,0],[C,B]]) = [[A^-1,0],[-B^-1 C A^-1, B^-1]], again MFMA out of LDS.
// LDS image S[128][130]: row stride 130 doubles (= 4*65 dwords) makes the k-contiguous MFMA fragment
// reads (16 rows x 2 k) hit 64 distinct banks.

constexpr int LEAF_T = 512;   // 8 waves: wave 0 factors the diagonal sub-blocks, all 8 run the MFMA phases
constexpr int LEAF_LOAD_BATCH = 16;   // all of a thread's 16 block vectors in flight at once (8: two dependent round trips)
constexpr int SB = 32;        // register-factored diagonal sub-block
constexpr int SLD = NB + 2;   // 130
constexpr int ZLD = 48;       // temp tiles [32][48]: k-row stride == 32 (mod 64) dwords

__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rsqrt_f64(double d) {
    double r = __builtin_amdgcn_rsq(d);
    // two Newton steps: r <- r * (1.5 - 0.5 * d * r^2)
    const double hd = 0.5 * d;
    r = r * fma(-hd * r, r, 1.5);
    r = r * fma(-hd * r, r, 1.5);
    return r;
}

// one 16x16 output tile, K deep, operands fetched through address functors (doubles in LDS)
template <int K, class FA, class FB>
__device__ __forceinline__ v4d leaf_tile(const double* lds, FA fa, FB fb, int l15, int l4) {
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};   // two chains: dependent f64 MFMAs do not issue back to back
#pragma unroll
    for (int kk = 0; kk < K / 4; kk += 2) {
        const double a0 = lds[fa(l15, kk * 4 + l4)];
        const double b0 = lds[fb(kk * 4 + l4, l15)];
        const double a1 = lds[fa(l15, kk * 4 + 4 + l4)];
        const double b1 = lds[fb(kk * 4 + 4 + l4, l15)];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}

// ---- phase A on the matrix pipe ---------------------------------------------------------------------------------------
// The 32x32 diagonal block W = [T | E] (T symmetric, E = I) is held by ONE wave as 16x16 f64 MFMA accumulator tiles
// (element (row, col) of a tile sits in lane (col, row & 3), register row >> 2) and eliminated four pivot rows at a
// time:   R = inv(chol(T[p,p])) * W[p, :]   (the pivot rows become rows of [L^T | L^-1]),
//         W[i, :] -= R[:, i]^T * R           (rows below)
// In that layout a 4 x 16 slab of pivot rows is at once the MFMA B operand (k = pivot row, n = column) and — read as
// "column tile i" — the A operand of the update of row tile i (m = row, k = pivot row): no data movement between
// steps.  Scaling the slab by the inverse 4x4 factor is one more MFMA whose A operand carries that inverse in its
// first four rows; only the 4x4 factorization itself (10 values fetched with v_readlane, ~40 scalar-uniform flops)
// is outside the matrix pipe.  The lower-left T tile is never needed (its rows' pivots come later and read only
// columns right of the diagonal), so three T tiles and three E tiles are live.
struct Inv44 { double w00, w10, w11, w20, w21, w22, w30, w31, w32, w33; };

__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// 1/sqrt(p) on the pivot chain: hardware estimate + ONE third-order correction (4 dependent fp64 ops instead of the
// 6 of two Newton steps):  e = 1 - p r^2,  r <- r + r e (1/2 + 3/8 e)   (error ~ e^3)
__device__ __forceinline__ double rsqrt_chain(double p) {
    const double r = __builtin_amdgcn_rsq(p);
    const double e = fma(-p * r, r, 1.0);
    return fma(r * e, fma(e, 0.375, 0.5), r);
}

// src holds the 4x4 SPD block D at lanes 16*a + lane0 + b (a = row, b = col); returns inv(chol(D)); bad = first
// non-positive pivot (0..3) or 4.  Written for dependency depth: everything that does not need the newest
// reciprocal root is formed before it arrives.
__device__ __forceinline__ Inv44 inv_chol44(double src, int lane0, int& bad) {
    double d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) d[a][b] = readlane_f64(src, 16 * a + lane0 + b);
    // Pivots are taken in PAIRS: for [[a, b], [b, c]] the second pivot is det/a with det = a c - b^2 (the same
    // cancellation as c - (b/sqrt a)^2), so 1/sqrt(a) and 1/sqrt(det) are refined side by side and
    // r1 = 1/sqrt(det/a) = (a r0) rsqrt(det): two reciprocal roots for the latency of one.
    const double a = d[0][0], b = d[1][0];
    const double det01 = fma(a, d[1][1], -(b * b));
    const double r0 = rsqrt_chain(a), rd01 = rsqrt_chain(det01);
    const double r1 = (a * r0) * rd01;
    const double l10 = b * r0, l20 = d[2][0] * r0, l30 = d[3][0] * r0;
    const double t21 = fma(-l20, l10, d[2][1]), t31 = fma(-l30, l10, d[3][1]);
    const double q22a = fma(-l20, l20, d[2][2]), q32a = fma(-l30, l20, d[3][2]), q33a = fma(-l30, l30, d[3][3]);
    const double l21 = t21 * r1, l31 = t31 * r1;
    const double p2 = fma(-l21, l21, q22a), t32 = fma(-l31, l21, q32a), q33b = fma(-l31, l31, q33a);
    const double det23 = fma(p2, q33b, -(t32 * t32));
    const double r2 = rsqrt_chain(p2), rd23 = rsqrt_chain(det23);
    const double r3 = (p2 * r2) * rd23;
    const double l32 = t32 * r2;
    // first non-positive (or NaN) pivot, 4 = none (pivot 1 = det01/a, pivot 3 = det23/p2); independent selects + min
    // instead of an early-exit chain, which the compiler turns into scalar branches in the middle of the chain
    const int b0 = a > 0.0 ? 4 : 0, b1 = det01 > 0.0 ? 4 : 1, b2 = p2 > 0.0 ? 4 : 2, b3 = det23 > 0.0 ? 4 : 3;
    bad = min(min(b0, b1), min(b2, b3));
    // inverse of the factor, row by row; each row's last multiply is by its own reciprocal root
    Inv44 w;
    w.w00 = r0;
    const double u10 = -l10 * r0;                     // w10 = u10 * r1
    w.w10 = u10 * r1; w.w11 = r1;
    const double l20r0 = l20 * r0;
    w.w20 = -fma(l21, w.w10, l20r0) * r2; w.w21 = -(l21 * r1) * r2; w.w22 = r2;
    const double s30 = fma(l31, w.w10, l30 * r0), s31 = l31 * r1;
    w.w30 = -fma(l32, w.w20, s30) * r3; w.w31 = -fma(l32, w.w21, s31) * r3; w.w32 = -(l32 * r2) * r3; w.w33 = r3;
    return w;
}

// One wave.  In: the lower triangle of the block at S[c0.., c0..].  Out: L^-1 -> the same block of S (the panel solve
// multiplies by its transpose), L^T tiles -> 12 staging rows of 64 doubles (flush_diag_block), first bad pivot -> badcol.
__device__ __forceinline__ void factor32_mfma(double* S, double* stage, int c0, int lane, int& badcol, long long* dbg = nullptr) {
    const int l15 = lane & 15, l4 = lane >> 4;
    int nst = 0;
    auto stamp = [&]() {
        if (dbg && lane == 0) dbg[nst] = (long long)__builtin_amdgcn_s_memtime();
        ++nst;
    };
    stamp();
    v4d T00, T01, T11, E00, E10, E11;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = l4 + 4 * r, col = l15;
        const int hi = row > col ? row : col, lo = row > col ? col : row;
        T00[r] = S[(c0 + hi) * SLD + c0 + lo];
        T01[r] = S[(c0 + 16 + col) * SLD + c0 + row];
        T11[r] = S[(c0 + 16 + hi) * SLD + c0 + 16 + lo];
        E00[r] = row == col ? 1.0 : 0.0;
        E11[r] = E00[r];
        E10[r] = 0.0;
    }
    stamp();
    const v4d zero = {0.0, 0.0, 0.0, 0.0};
    // lane (m, k) of the scaling MFMA's A operand carries inv[m][k] (m < 4, k <= m), zero elsewhere.  Bitwise select
    // with lane-constant masks: plain AND/OR on the pivot chain (a ternary chain here gets compiled into a
    // dynamically indexed private array, i.e. a scratch-memory round trip per step)
    auto lane_mask = [&](int m, int k) { return (l15 == m && l4 == k) ? ~0ull : 0ull; };
    const unsigned long long M00 = lane_mask(0, 0), M10 = lane_mask(1, 0), M11 = lane_mask(1, 1), M20 = lane_mask(2, 0),
                             M21 = lane_mask(2, 1), M22 = lane_mask(2, 2), M30 = lane_mask(3, 0), M31 = lane_mask(3, 1),
                             M32 = lane_mask(3, 2), M33 = lane_mask(3, 3);
    auto bits = [](double x) { return (unsigned long long)__double_as_longlong(x); };
    auto a_operand = [&](const Inv44& w) {
        const unsigned long long v = (bits(w.w00) & M00) | (bits(w.w10) & M10) | (bits(w.w11) & M11) | (bits(w.w20) & M20) |
                                     (bits(w.w21) & M21) | (bits(w.w22) & M22) | (bits(w.w30) & M30) | (bits(w.w31) & M31) |
                                     (bits(w.w32) & M32) | (bits(w.w33) & M33);
        return __longlong_as_double((long long)v);
    };
    // the scaled pivot rows (final rows of [L^T | L^-1]) are kept apart from the accumulators: nothing reads an
    // accumulator's pivot rows again (the update's A operand is zero there)
    double fT00[4], fT01[4], fE00[4], fT11[4], fE10[4], fE11[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {               // pivot rows 4q .. 4q+3 (row tile 0)
        int bad;
        const Inv44 w = inv_chol44(T00[q], 4 * q, bad);
        badcol = (badcol < 0 && bad < 4) ? c0 + 4 * q + bad : badcol;
        const double aw = a_operand(w);
        // the next pivot block lives in T00 (q < 3) or T11 (q == 3): its slab and its update go down the matrix pipe first
        const double sT0 = mfma_f64(aw, T00[q], zero)[0];
        const double a0 = l15 >= 4 * q + 4 ? -sT0 : 0.0;         // rows of tile 0 below the pivot rows
        if (q < 3) T00 = mfma_f64(a0, sT0, T00);
        __builtin_amdgcn_sched_barrier(0);                       // keep this pair ahead of the independent slabs
        const double sT1 = mfma_f64(aw, T01[q], zero)[0];
        const double a1 = -sT1;                                  // every row of tile 1
        if (q == 3) { T11 = mfma_f64(a1, sT1, T11); __builtin_amdgcn_sched_barrier(0); }
        const double sE0 = mfma_f64(aw, E00[q], zero)[0];
        fT00[q] = sT0; fT01[q] = sT1; fE00[q] = sE0;
        if (q < 3) {
            T01 = mfma_f64(a0, sT1, T01);
            E00 = mfma_f64(a0, sE0, E00);
            T11 = mfma_f64(a1, sT1, T11);
        }
        E10 = mfma_f64(a1, sE0, E10);
        stamp();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {               // pivot rows 16 + 4q .. (row tile 1)
        int bad;
        const Inv44 w = inv_chol44(T11[q], 4 * q, bad);
        badcol = (badcol < 0 && bad < 4) ? c0 + 16 + 4 * q + bad : badcol;
        const double aw = a_operand(w);
#ifdef EMCID_LEAF_MIDSTAMP
        stamp();
#endif
        const double sT1 = mfma_f64(aw, T11[q], zero)[0];
        const double a1 = l15 >= 4 * q + 4 ? -sT1 : 0.0;
        if (q < 3) T11 = mfma_f64(a1, sT1, T11);
        __builtin_amdgcn_sched_barrier(0);
        const double sE0 = mfma_f64(aw, E10[q], zero)[0];
        const double sE1 = mfma_f64(aw, E11[q], zero)[0];
        fT11[q] = sT1; fE10[q] = sE0; fE11[q] = sE1;
        if (q < 3) {
            E10 = mfma_f64(a1, sE0, E10);
            E11 = mfma_f64(a1, sE1, E11);
        }
        stamp();
    }
    // ---- outputs.  fT* hold L^T (valid for row <= col), fE* hold L^-1 (upper right tile = 0).
    stamp();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = l4 + 4 * r, col = l15;
        // L^-1 replaces the block in S (the panel solve reads it from there, transposed)
        S[(c0 + row) * SLD + c0 + col] = fE00[r];
        S[(c0 + row) * SLD + c0 + 16 + col] = 0.0;
        S[(c0 + 16 + row) * SLD + c0 + col] = fE10[r];
        S[(c0 + 16 + row) * SLD + c0 + 16 + col] = fE11[r];
    }
    stamp();
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // L^T tiles -> staging rows (lane-linear, conflict free); another wave takes them to global
        stage[(0 + r) * SLD + lane] = fT00[r];
        stage[(4 + r) * SLD + lane] = fT01[r];
        stage[(8 + r) * SLD + lane] = fT11[r];
    }
    stamp();
}

// second half of phase A's output, by a wave that is not on the pivot chain: L_pp -> global from the staged L^T tiles
// (L[i][j] = L^T[j][i]; a lane's four rows are adjacent columns of L; zeros above the diagonal)
__device__ __forceinline__ void flush_diag_block(const double* stage, int c0, double* __restrict__ L, int64_t ldl, int lane) {
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = l4 + 4 * r, col = l15;
        double* Lc = L + (int64_t)(c0 + col) * ldl + c0;
        Lc[row] = row <= col ? stage[(0 + r) * SLD + lane] : 0.0;
        Lc[16 + row] = 0.0;
        double* Lc1 = Lc + 16 * ldl;
        Lc1[row] = stage[(4 + r) * SLD + lane];
        Lc1[16 + row] = row <= col ? stage[(8 + r) * SLD + lane] : 0.0;
    }
}

constexpr int LEAF_LDS = NB * SLD + 2 * SB * ZLD;      // doubles

__device__ __forceinline__ void chol_leaf_body(const double* __restrict__ A, int64_t lda, double* __restrict__ L, int64_t ldl,
                                               double* __restrict__ inv, int64_t ldinv, int* info, int col0, long long* dbg,
                                               double* lds) {
    double* S = lds;
    constexpr int ZOFF = NB * SLD;          // two [32][48] temporaries behind S
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    int nstamp = 0;
    auto stamp = [&]() {   // diagnostic only (dbg != nullptr): shader-clock stamps of wave 0 at phase boundaries
        if (dbg && tid == 0) dbg[nstamp] = (long long)__builtin_amdgcn_s_memtime();
        ++nstamp;
    };
    stamp();

    {
        // 128 x 128 block = 8192 16-byte vectors, 16 per thread, fetched in batches of LEAF_LOAD_BATCH independent loads: the block
        // was written by other compute units in the launch before, so every batch pays a cold-read latency of its own
        constexpr int VPR = NB / 2;   // vectors per row
        constexpr int LB = LEAF_LOAD_BATCH;
#pragma unroll 1
        for (int b0 = 0; b0 < NB * VPR / LEAF_T; b0 += LB) {
            v2d x[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int v = tid + (b0 + u) * LEAF_T;
                const int i = v / VPR, j = 2 * (v % VPR);
                x[u] = (j <= i) ? *reinterpret_cast<const v2d*>(A + (int64_t)i * lda + j) : (v2d){0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int v = tid + (b0 + u) * LEAF_T;
                const int i = v / VPR, j = 2 * (v % VPR);
                if (j + 1 > i) x[u][1] = 0.0;   // strictly-upper element of a pair straddling the diagonal
                *reinterpret_cast<v2d*>(S + i * SLD + j) = x[u];
            }
        }
    }
    __syncthreads();
    stamp();

    int badcol = -1;   // first non-positive pivot seen by wave 0 (uniform)
#pragma unroll 1
    for (int p = 0; p < NB / SB; ++p) {
        const int c0 = p * SB;
        const int R0 = c0 + SB;               // first row below the diagonal block
        const int mb = (NB - R0) / 16;        // 16-row blocks below

        // ---- phase A: wave 0 factors the diagonal block and its inverse on the matrix pipe (factor32_mfma) ------------
        // staging rows: the never-touched upper right quadrant S[0:64][64:128] (12 rows per panel; lvl2 reuses it later)
        double* stage = S + (12 * p) * SLD + 64;
        if (wave == 0) factor32_mfma(S, stage, c0, lane, badcol, (dbg && p == 1) ? dbg + 16 : nullptr);
        __syncthreads();
        stamp();
        if (wave == LEAF_T / 64 - 1) flush_diag_block(stage, c0, L, ldl, lane);   // the last wave has the fewest panel rows
        if (mb == 0) break;

        // ---- phase B: P = A_below * Z (rows R0.., 32 columns), in place; one wave owns a 16-row block ---------
        for (int ib = wave; ib < mb; ib += LEAF_T / 64) {
            const int r0 = R0 + ib * 16;
            v4d acc[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                acc[cb] = leaf_tile<SB>(lds, [&](int i, int k) { return (r0 + i) * SLD + c0 + k; },
                                        [&](int k, int j) { return (c0 + cb * 16 + j) * SLD + c0 + k; }, l15, l4);   // Z[k][c] = inv[c][k]
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = r0 + l4 + 4 * q, col = c0 + cb * 16 + l15;
                    S[row * SLD + col] = acc[cb][q];
                    L[(int64_t)row * ldl + col] = acc[cb][q];
                }
        }
        __syncthreads();
        stamp();

        // ---- phase C: trailing update T -= P P^T on the lower 16x16 tiles ---------------------------------------
        const int ntiles = mb * (mb + 1) / 2;
        for (int t = wave; t < ntiles; t += LEAF_T / 64) {
            int ib = 0, rem = t;
            while (rem > ib) { rem -= ib + 1; ++ib; }
            const int jb = rem;
            const int ri = R0 + ib * 16, rj = R0 + jb * 16;
            const v4d acc = leaf_tile<SB>(lds, [&](int i, int k) { return (ri + i) * SLD + c0 + k; },
                                          [&](int k, int j) { return (rj + j) * SLD + c0 + k; }, l15, l4);
#pragma unroll
            for (int q = 0; q < 4; ++q) S[(ri + l4 + 4 * q) * SLD + rj + l15] -= acc[q];
        }
        __syncthreads();
        stamp();
    }
    if (wave == 0 && lane == 0 && badcol >= 0) atomicCAS(info, 0, col0 + badcol + 1);  // not SPD (or NaN pivot)

    // ---- inverse assembly, level 1: the two 64-blocks.  X = -B (C A), A/B = 32x32 inverses, C = L block ------
    {
        constexpr int NW = LEAF_T / 64;
        // 8 tiles: (block q, tile ti, tj)
        for (int t = wave; t < 8; t += NW) {     // T1 = C A
            const int q = t >> 2, ti = (t >> 1) & 1, tj = t & 1, b = 64 * q, toff = ZOFF + q * SB * ZLD;
            const v4d acc = leaf_tile<SB>(lds, [&](int i, int k) { return (b + 32 + ti * 16 + i) * SLD + b + k; },
                                          [&](int k, int j) { return (b + k) * SLD + b + tj * 16 + j; }, l15, l4);
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[toff + (ti * 16 + l4 + 4 * r) * ZLD + tj * 16 + l15] = acc[r];
        }
        __syncthreads();
        for (int t = wave; t < 8; t += NW) {     // X = -B T1, written over C
            const int q = t >> 2, ti = (t >> 1) & 1, tj = t & 1, b = 64 * q, toff = ZOFF + q * SB * ZLD;
            const v4d acc = leaf_tile<SB>(lds, [&](int i, int k) { return (b + 32 + ti * 16 + i) * SLD + b + 32 + k; },
                                          [&](int k, int j) { return toff + k * ZLD + tj * 16 + j; }, l15, l4);
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(b + 32 + ti * 16 + l4 + 4 * r) * SLD + b + tj * 16 + l15] = -acc[r];
        }
        __syncthreads();
    }
    stamp();
    // ---- level 2: X64 = -B64 (C64 A64); T lives in the unused upper-right block S[0:64][64:128] ----------------
    for (int t = wave; t < 16; t += LEAF_T / 64) {
        const int ti = t >> 2, tj = t & 3;
        const v4d acc = leaf_tile<64>(lds, [&](int i, int k) { return (64 + ti * 16 + i) * SLD + k; },
                                      [&](int k, int j) { return k * SLD + tj * 16 + j; }, l15, l4);
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(ti * 16 + l4 + 4 * r) * SLD + 64 + tj * 16 + l15] = acc[r];
    }
    __syncthreads();
    {
        v4d acc[(16 + LEAF_T / 64 - 1) / (LEAF_T / 64)];
        int n = 0;
        for (int t = wave; t < 16; t += LEAF_T / 64, ++n) {
            const int ti = t >> 2, tj = t & 3;
            acc[n] = leaf_tile<64>(lds, [&](int i, int k) { return (64 + ti * 16 + i) * SLD + 64 + k; },
                                   [&](int k, int j) { return k * SLD + 64 + tj * 16 + j; }, l15, l4);
        }
        __syncthreads();   // all reads of B64 / T done before C64's place is overwritten
        n = 0;
        for (int t = wave; t < 16; t += LEAF_T / 64, ++n) {
            const int ti = t >> 2, tj = t & 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(64 + ti * 16 + l4 + 4 * r) * SLD + tj * 16 + l15] = -acc[n][r];
        }
    }
    __syncthreads();
    stamp();
    for (int v = tid; v < NB * NB / 2; v += LEAF_T) {
        const int i = v / (NB / 2), j = 2 * (v % (NB / 2));
        v2d x = *reinterpret_cast<const v2d*>(S + i * SLD + j);
        if (j > i) x[0] = 0.0;
        if (j + 1 > i) x[1] = 0.0;
        *reinterpret_cast<v2d*>(inv + (int64_t)i * ldinv + j) = x;
    }
    __syncthreads();
    stamp();
}

__global__ __launch_bounds__(LEAF_T) void chol_leaf_kernel(const double* __restrict__ A, int64_t lda, double* __restrict__ L,
                                                            int64_t ldl, double* __restrict__ inv, int64_t ldinv, int* info,
                                                            int col0, long long* dbg = nullptr, int64_t s_mat = 0,
                                                            int64_t s_inv = 0) {
    // blockIdx.x = which matrix of a batch (independent factorizations share the launch: their serial spines overlap)
    __shared__ __attribute__((aligned(16))) double lds[LEAF_LDS];
    chol_leaf_body(A + blockIdx.x * s_mat, lda, L + blockIdx.x * s_mat, ldl, inv + blockIdx.x * s_inv, ldinv, info, col0, dbg, lds);
}

// A product that does NOT depend on the matrix being factored, cut into `nslices` K slices, one per leaf launch: the leaf
// occupies ONE compute unit for ~36 us while the rest of the chip has (almost) nothing to do, so an independent GEMM of the
// caller's rides along in the launches' other workgroups ("shadow").  Used by the dual solver for P = Yt X (X = inv(L) of
// lam*C', lower triangular, stored [k][n]), which turns U = (Z^T Yt) X — a GEMM against the triangle AFTER the N x N solve, on
// the critical path — into U = Z^T P.  C = A B with B(k, n) = 0 for k < n; 64 x 128 output tiles; a workgroup takes column tile
// j and its mirror image NT-1-j (equal total depth for every workgroup) and contracts slice `slice` of each tile's own K range;
// slice 0 stores, later slices add — launches of one stream are ordered and a tile belongs to one workgroup per launch, so the
// sum order is fixed (bit-reproducible) and nothing is atomic.
struct ShadowJob {
    const double* A; int64_t lda;     // [M][K], K contiguous
    const double* B; int64_t ldb;     // [K][N], N contiguous, zero for k < n
    double* C; int64_t ldc;           // [M][N]
    int M, N, K;
    int wgs;                          // workgroups per launch: ceil(M / 64) * ceil(ceil(N / 128) / 2); 0 = no job
    long long* stamps;                // diagnostic (usually null): per launch slice and workgroup [start, mid, end, kind]
    int xcd_gx;                       // > 0: XCD-blocked tile assignment, the 8 XCDs as a xcd_gx x (8 / xcd_gx) grid (set by the launcher)
    int nofast;                       // GemmShape.nofast for the shadow tiles (EMCID_GEMM_FAST=0)
    int fuse_pair;                    // both tiles of a pair through one software pipeline (EMCID_SHADOW_FUSE, default 1)
};

// The explicit inverse of the factor, built ROW BLOCK BY ROW BLOCK inside the factorization's own launches (dual solver: XS =
// inv(LS) is what turns Z = S^-1 R into two GEMMs).  Stored transposed, Xt[n][k] = X[k][n], so that both products of a step are
// K-contiguous on both sides:
//     leaf launch j  (j >= 1):  Tt [128 j, 128] = Xt[0:128j, 0:128j] L[j, 0:j]^T        (X of the leading j blocks is complete)
//     spine launch j (j >= 0):  Xt[0:128j, j]  = -Tt inv(L_jj)^T,   Xt[j, j] = inv(L_jj)^T
// i.e. X[j, 0:j] = -inv(L_jj) L[j, 0:j] X[0:j, 0:j].  Behind the last leaf only that block row's second product is left (one
// small launch) where the recursive-halving build took six dependent ones (~70 us per layer at N = 1000).
struct XrowJob {
    double* Xt; int64_t ldx;      // [n, n] transposed inverse (every entry the consumers read is written here)
    double* Tt;                   // [n, 128] scratch, leading dimension 128
};
inline long long* g_step_stamps = nullptr;      // set by emcid_debug_step_stamps
constexpr int SH_BM = 64, SH_BN = 128;

template <int SH_BK, int SH_PF>
__device__ __forceinline__ void shadow_tile(const ShadowJob& sh, int bm, int bn, int slice, int nslices, double* smem) {
    using TA = OpTile<true, SH_BM, SH_BK>;
    using TB = OpTile<false, SH_BN, SH_BK>;
    GemmShape q{sh.A, sh.lda, sh.B, sh.ldb, sh.M, sh.N, sh.K, 0};
    q.tri = 2;
    q.nofast = sh.nofast;
    const int depth = streamk_depth(q, bn, SH_BN, SH_BK);
    const int per = (depth + nslices - 1) / nslices;
    const int kb = slice * per, ke = min(depth, kb + per);
    if (kb >= ke) return;
    constexpr int MI = SH_BM / 2 / 16, NI = SH_BN / 4 / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / 4) * (SH_BM / 2), wn0 = (wave % 4) * (SH_BN / 4);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int m0 = bm * SH_BM, n0 = bn * SH_BN;
    // The earlier slices' sum is fetched BEFORE the K loop (the launch before this one wrote it from another compute unit: a
    // cold read costs ~2.7 us here, scripts/step_stamps.py) and added after it; clamped addresses, so the loads are unconditional
    v4d prev[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = min(m0 + wm0 + TA::index_of(i, l4 + 4 * r), sh.M - 1);
                const int n = min(n0 + wn0 + TB::index_of(j, l15), sh.N - 1);
                prev[i][j][r] = slice ? sh.C[(int64_t)m * sh.ldc + n] : 0.0;
            }
    __builtin_amdgcn_sched_barrier(0);      // keep those loads up here
    v4d acc[MI][NI];
    gemm_f64_tile_acc<true, false, SH_BM, SH_BN, SH_BK, 2, 4, SH_PF>(q, bm, bn, smem, kb, ke, acc);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm0 + TA::index_of(i, l4 + 4 * r);
                const int n = n0 + wn0 + TB::index_of(j, l15);
                if (m < sh.M && n < sh.N) sh.C[(int64_t)m * sh.ldc + n] = prev[i][j][r] + acc[i][j][r];
            }
    __syncthreads();      // the next tile's first LDS store must not overtake this tile's last MFMA stage
}

// Both tiles of a pair in ONE software pipeline (interior shapes: M % 64 == N % 128 == K % 16 == 0): the K tiles of tile bn0's
// slice, then those of bn1's, flow through the same register ring and LDS double buffer; at the seam tile bn0's sum goes out
// while bn1's first K tiles are already in flight, so the second tile pays no load-latency prologue (~2.5 us of a 38 us launch).
__device__ __forceinline__ void shadow_pair_fused(const ShadowJob& sh, int bm, int bn0, int bn1, int slice, int nslices, double* smem) {
    constexpr int BK = 16, PF = 3, NT = LEAF_T;
    using TA = OpTile<true, SH_BM, BK>;
    using TB = OpTile<false, SH_BN, BK>;
    constexpr int STAGE = TA::SIZE + TB::SIZE;
    constexpr int MI = SH_BM / 2 / 16, NI = SH_BN / 4 / 16;
    constexpr int NA = TA::NVEC / NT, NB = TB::NVEC / NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / 4) * (SH_BM / 2), wn0 = (wave % 4) * (SH_BN / 4);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int m0 = bm * SH_BM;
    const int KT = sh.K / BK;
    // segment g: column tile bn[g], absolute K tiles [kb[g], ke[g])  (B(k, n) = 0 for k < n: the tile's own range starts at its column)
    int bn[2] = {bn0, bn1}, kb[2], ke[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int first = bn[g] * SH_BN / BK, depth = KT - first;
        const int per = (depth + nslices - 1) / nslices;
        kb[g] = first + min(depth, slice * per);
        ke[g] = first + min(depth, slice * per + per);
    }
    const int n0len = ke[0] - kb[0], total = n0len + (ke[1] - kb[1]);
    if (total <= 0) return;
    const double* pa[NA];
    const double* pb[NB];
    int la[NA], lb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int v = tid + i * NT;
        pa[i] = sh.A + (int64_t)(m0 + v / (BK / 2)) * sh.lda + 2 * (v % (BK / 2));
        la[i] = (v / (BK / 2)) * TA::LD + 2 * (v % (BK / 2));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int v = tid + i * NT;
        pb[i] = sh.B + (int64_t)(v / (SH_BN / 2)) * sh.ldb + 2 * (v % (SH_BN / 2));
        lb[i] = (v / (SH_BN / 2)) * TB::LD + 2 * (v % (SH_BN / 2));
    }
    v2d fa[PF][NA], fb[PF][NB];
    auto fetch = [&](int s, int u) {            // unified step u -> (segment, absolute K tile); past the end: the last one again
        const int uu = u < total ? u : total - 1;
        const int g = uu < n0len ? 0 : 1;
        const int64_t kt = g ? kb[1] + (uu - n0len) : kb[0] + uu;
        const int64_t ncol = (int64_t)bn[g] * SH_BN;
#pragma unroll
        for (int i = 0; i < NA; ++i) fa[s][i] = *reinterpret_cast<const v2d*>(pa[i] + kt * BK);
#pragma unroll
        for (int i = 0; i < NB; ++i) fb[s][i] = *reinterpret_cast<const v2d*>(pb[i] + kt * BK * sh.ldb + ncol);
    };
    auto stash = [&](int s, double* stage) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<v2d*>(stage + la[i]) = fa[s][i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<v2d*>(stage + TA::SIZE + lb[i]) = fb[s][i];
    };
    v4d prev[MI][NI], acc[MI][NI];
    auto c_ptr = [&](int g, int i, int j, int r) {
        return sh.C + (int64_t)(m0 + wm0 + TA::index_of(i, l4 + 4 * r)) * sh.ldc + bn[g] * SH_BN + wn0 + TB::index_of(j, l15);
    };
    auto load_prev = [&](int g) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) prev[i][j][r] = slice ? *c_ptr(g, i, j, r) : 0.0;
    };
    auto flush = [&](int g) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    *c_ptr(g, i, j, r) = prev[i][j][r] + acc[i][j][r];
                    acc[i][j][r] = 0.0;
                }
    };
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    load_prev(n0len > 0 ? 0 : 1);
#pragma unroll
    for (int s = 0; s < PF; ++s) fetch(s, s);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    stash(0, smem);
    __syncthreads();
    for (int base = 0; base < total; base += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int u = base + s;
            if (u < total) {
                const double* As = smem + (u & 1) * STAGE;
                const double* Bs = As + TA::SIZE;
                fetch(s, u + PF);
                __builtin_amdgcn_sched_barrier(0);
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
                __builtin_amdgcn_sched_barrier(0);
                if (u + 1 < total) stash((s + 1) % PF, smem + ((u + 1) & 1) * STAGE);
                if (u + 1 == n0len && n0len < total) {       // the seam: tile bn0 is complete, bn1's K tiles are already in flight
                    flush(0);
                    load_prev(1);
                }
                __syncthreads();
            }
        }
    }
    flush(n0len < total ? 1 : 0);
    __syncthreads();
}

template <int SH_BK, int SH_PF>
__device__ __forceinline__ void shadow_pair(const ShadowJob& sh, int s_, int slice, int nslices, double* lds) {
    const int ntl = (sh.N + SH_BN - 1) / SH_BN, npairs = (ntl + 1) / 2;
    int bm = s_ / npairs, pr = s_ % npairs;
    if (sh.xcd_gx > 0) {
        // The workgroups of a launch go to the 8 XCDs round-robin by id, so shadow workgroups s_ = g (mod 8) share an L2.  Give
        // each XCD a compact gx x gy block of the (row tile, column pair) grid: its 24 workgroups then re-read each other's A and
        // B tiles out of their own L2 (B: half of the X slice, 2.4 MB at d = 3072) instead of all 192 streaming everything from
        // the Infinity Cache — the K loop here is bound by that traffic, not by the matrix pipe (scripts/step_stamps.py).
        const int mb = (sh.M + SH_BM - 1) / SH_BM;
        const int g = s_ & 7, r = s_ >> 3;
        const int gy = 8 / sh.xcd_gx, rows = mb / sh.xcd_gx, cols = npairs / gy;
        bm = (g / gy) * rows + r / cols;
        pr = (g % gy) * cols + r % cols;
    }
    if (SH_BK == 16 && SH_PF == 3 && sh.fuse_pair && ntl - 1 - pr != pr && sh.M % SH_BM == 0 && sh.N % SH_BN == 0 &&
        sh.K % SH_BK == 0) {
        shadow_pair_fused(sh, bm, pr, ntl - 1 - pr, slice, nslices, lds);
        return;
    }
    shadow_tile<SH_BK, SH_PF>(sh, bm, pr, slice, nslices, lds);                            // the long K range first
    if (ntl - 1 - pr != pr) shadow_tile<SH_BK, SH_PF>(sh, bm, ntl - 1 - pr, slice, nslices, lds);
}

// One launch = the leaf of block j (workgroup 0) AND the trailing update of step j-1 that the leaf does not depend on (the other
// workgroups, one 128 x 128 tile each): on one in-order stream a leaf and the previous step's bulk cannot overlap as two kernels,
// and as parallel graph branches they cost more than they save (see cholesky_lookahead) — as one grid they simply run side by side.
// Behind those, sh.wgs workgroups of the caller's shadow product (slice `slice` of `nslices`).
__global__ __launch_bounds__(LEAF_T) void chol_step_leaf_kernel(const double* __restrict__ A, int64_t lda, double* __restrict__ L,
                                                                 int64_t ldl, double* __restrict__ inv, int64_t ldinv, int* info,
                                                                 int col0, GemmShape trail, EpiAxpby trail_epi, int ntrail,
                                                                 ShadowJob sh, int slice, int nslices, GemmShape xr, EpiAxpby xr_epi,
                                                                 int nxr) {
    __shared__ __attribute__((aligned(16))) double lds[LEAF_LDS];
    long long* stamp = (sh.stamps && blockIdx.x < 512) ? sh.stamps + ((int64_t)slice * 512 + blockIdx.x) * 4 : nullptr;
    if (stamp && threadIdx.x == 0) stamp[0] = (long long)__builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0) {
        chol_leaf_body(A, lda, L, ldl, inv, ldinv, info, col0, nullptr, lds);
        if (stamp && threadIdx.x == 0) { stamp[2] = (long long)__builtin_amdgcn_s_memrealtime(); stamp[3] = 1; }
        return;
    }
    if ((int)blockIdx.x <= ntrail) {
        // lower-trapezoid tile (bm, bn), bn <= bm + 1, from the linear index: row bm holds bm + 2 tiles
        int bm = 0, rem = (int)blockIdx.x - 1;
        while (rem >= bm + 2) { rem -= bm + 2; ++bm; }
        gemm_f64_tile<true, true, 128, 128, 16, 2, 4, EpiAxpby>(trail, trail_epi, bm, rem, 0, lds);
        if (stamp && threadIdx.x == 0) { stamp[2] = (long long)__builtin_amdgcn_s_memrealtime(); stamp[3] = 2; }
        return;
    }
    if ((int)blockIdx.x <= ntrail + nxr) {      // a 32 x 64 tile of Tt = Xt L[j, 0:j]^T (XrowJob) on the workgroup's first four waves
        // (a 128 x 128 tile of it contracts up to 896 deep on ONE compute unit: ~100 us, three leaf launches long; the small tiles
        // finish under the leaf, and 8 j of them still leave every shadow workgroup a compute unit of its own)
        if (threadIdx.x >= 256) return;
        const int t = (int)blockIdx.x - 1 - ntrail;
        gemm_f64_tile<true, true, 32, 64, 16, 2, 2, EpiAxpby>(xr, xr_epi, t / 2, t % 2, 0, lds);
        if (stamp && threadIdx.x == 0) { stamp[2] = (long long)__builtin_amdgcn_s_memrealtime(); stamp[3] = 6; }
        return;
    }
    ntrail += nxr;
    int s_ = (int)blockIdx.x - 1 - ntrail;
    // (K-tile depth 32, 6 or 8 K tiles of loads in flight and LDS fragment prefetch were measured here as template variants:
    //  none moved the launch time by more than 1 us)
    shadow_pair<16, 3>(sh, s_, slice, nslices, lds);
    if (stamp && threadIdx.x == 0) { stamp[2] = (long long)__builtin_amdgcn_s_memrealtime(); stamp[3] = 3; }
}


// ---- the step between two leaves of a small Cholesky: L[j+1, j] and the next diagonal block -------------------------------------
// After leaf j the next leaf needs exactly one 128 x 128 block: D = A[j+1, j+1] - P P^T with P = L[j+1, j] = A[j+1, j] inv(L_jj)^T.
// As two launches of the tile GEMM this costs ~17 us of latency on the serial spine of the factorization; here it is one
// launch of 36 workgroups, one per lower 16 x 16 tile (bi, bj) of D: the workgroup forms the two 16-row strips P_i and P_j it
// needs itself (redundantly — no workgroup waits for another), updates its tile, and the bj == 0 column also stores its P_i
// strip into L.  Operands through LDS with the leaf's row stride; the triangular inverse is staged in two 64-row halves.
constexpr int SP_T = 256;
constexpr int SP_R = 16;                    // strip height
constexpr int SP_LDS = (64 + 4 * SP_R) * SLD;

template <class FA, class FB>
__device__ __forceinline__ v4d leaf_tile_k(const double* lds, FA fa, FB fb, int l15, int l4, int K) {
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk < K / 4; kk += 2) {
        const double a0 = lds[fa(l15, kk * 4 + l4)];
        const double b0 = lds[fb(kk * 4 + l4, l15)];
        const double a1 = lds[fa(l15, kk * 4 + 4 + l4)];
        const double b1 = lds[fb(kk * 4 + 4 + l4, l15)];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}

__device__ __forceinline__ void chol_spine_body(const double* __restrict__ Ablk, int64_t lda, const double* __restrict__ inv, int64_t ldi,
                                                double* __restrict__ Lout, int64_t ldl, double* __restrict__ D, int tile, double* lds) {
    double* sInv = lds;                     // [64][SLD]   half of inv(L_jj): rows n of the half, columns k
    double* sAi = lds + 64 * SLD;           // [16][SLD]   A strips
    double* sAj = sAi + SP_R * SLD;
    double* sPi = sAj + SP_R * SLD;
    double* sPj = sPi + SP_R * SLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // lower tile (bi, bj) of the 8 x 8 tiles of D from the linear tile index
    int bi = 0, rem = tile;
    while (rem > bi) { rem -= bi + 1; ++bi; }
    const int bj = rem;
    // every global vector this thread will stage, fetched up front (32 independent 16-byte loads: the two A strips, the
    // 64 x 64 and the 64 x 128 halves of the triangular inverse), so the kernel pays ONE memory latency, not one per vector
    v2d ra[8], r0[8], r1[16];
    double dold[4];         // the D tile as it is (wave 0 subtracts from it at the very end: fetched NOW, not then — a cold read is ~2.5 us)
#pragma unroll
    for (int q = 0; q < 4; ++q) dold[q] = D[(int64_t)(bi * SP_R + l4 + 4 * q) * lda + bj * SP_R + l15];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int v = tid + u * SP_T;                               // 2 strips x 16 rows x 64 vectors
        const int which = v / (SP_R * (NB / 2)), w = v % (SP_R * (NB / 2));
        const int r = w / (NB / 2), c = 2 * (w % (NB / 2));
        ra[u] = *reinterpret_cast<const v2d*>(Ablk + (int64_t)((which ? bj : bi) * SP_R + r) * lda + c);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int v = tid + u * SP_T;                               // 64 rows x 32 vectors
        r0[u] = *reinterpret_cast<const v2d*>(inv + (int64_t)(v / 32) * ldi + 2 * (v % 32));
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int v = tid + u * SP_T;                               // 64 rows x 64 vectors
        r1[u] = *reinterpret_cast<const v2d*>(inv + (int64_t)(64 + v / 64) * ldi + 2 * (v % 64));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int v = tid + u * SP_T;
        const int which = v / (SP_R * (NB / 2)), w = v % (SP_R * (NB / 2));
        *reinterpret_cast<v2d*>((which ? sAj : sAi) + (w / (NB / 2)) * SLD + 2 * (w % (NB / 2))) = ra[u];
    }
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                    // previous half's reads done
        if (half == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = tid + u * SP_T;
                *reinterpret_cast<v2d*>(sInv + (v / 32) * SLD + 2 * (v % 32)) = r0[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int v = tid + u * SP_T;
                *reinterpret_cast<v2d*>(sInv + (v / 64) * SLD + 2 * (v % 64)) = r1[u];
            }
        }
        __syncthreads();
        // P[:, 16 t .. 16 t + 16] = A[:, 0 : K] inv[16 t .., 0 : K]^T with K = 16 (t + 1): four column tiles per half, one per
        // wave — the deeper tiles of the second half go to the waves that had the shallow ones in the first
        const int tl = half ? 3 - wave : wave;          // tile inside the half
        const int t = 4 * half + tl, K = SP_R * (t + 1);
        // P_i and P_j tiles in ONE loop: they share the inverse fragment (B operand), and four independent accumulator chains
        // cover the LDS latency that two chains per call left exposed (stamps: a spine workgroup was 12.5 us, 5 of them here)
        v4d pi0 = {0.0, 0.0, 0.0, 0.0}, pi1 = pi0, pj0 = pi0, pj1 = pi0;
        {
            const double* ai = sAi + l15 * SLD + l4;
            const double* aj = sAj + l15 * SLD + l4;
            const double* bb = sInv + (tl * SP_R + l15) * SLD + l4;
            for (int k = 0; k < K; k += 16) {       // K is a multiple of 16: two 8-deep steps per trip, all twelve reads up front
                const double b0 = bb[k], b1 = bb[k + 4], b2 = bb[k + 8], b3 = bb[k + 12];
                const double x0 = ai[k], x1 = ai[k + 4], x2 = ai[k + 8], x3 = ai[k + 12];
                const double y0 = aj[k], y1 = aj[k + 4], y2 = aj[k + 8], y3 = aj[k + 12];
                pi0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, b0, pi0, 0, 0, 0);
                pj0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, b0, pj0, 0, 0, 0);
                pi1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, b1, pi1, 0, 0, 0);
                pj1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, b1, pj1, 0, 0, 0);
                pi0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, b2, pi0, 0, 0, 0);
                pj0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y2, b2, pj0, 0, 0, 0);
                pi1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, b3, pi1, 0, 0, 0);
                pj1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y3, b3, pj1, 0, 0, 0);
            }
        }
        const v4d pi = pi0 + pi1, pj = pj0 + pj1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sPi[(l4 + 4 * q) * SLD + t * SP_R + l15] = pi[q];
            sPj[(l4 + 4 * q) * SLD + t * SP_R + l15] = pj[q];
        }
    }
    __syncthreads();
    // ---- D tile -= P_i P_j^T, the 128-deep contraction cut over the four waves and summed through LDS (sInv is free now)
    {
        const int k0 = 32 * wave;
        const v4d part = leaf_tile_k(lds, [&](int i, int k) { return (int)(sPi - lds) + i * SLD + k0 + k; },
                                     [&](int k, int j) { return (int)(sPj - lds) + j * SLD + k0 + k; }, l15, l4, 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) sInv[(wave * SP_R + l4 + 4 * q) * SLD + l15] = part[q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = l4 + 4 * q;
            const double sum = (sInv[r * SLD + l15] + sInv[(SP_R + r) * SLD + l15]) + (sInv[(2 * SP_R + r) * SLD + l15] + sInv[(3 * SP_R + r) * SLD + l15]);
            D[(int64_t)(bi * SP_R + r) * lda + bj * SP_R + l15] = dold[q] - sum;
        }
    }
    if (bj == 0) {      // this column of workgroups covers every strip once: P_i -> L[j+1, j]
        for (int v = tid; v < SP_R * (NB / 2); v += SP_T) {
            const int r = v / (NB / 2), c = 2 * (v % (NB / 2));
            *reinterpret_cast<v2d*>(Lout + (int64_t)(bi * SP_R + r) * ldl + c) = *reinterpret_cast<const v2d*>(sPi + r * SLD + c);
        }
    }
}

// One launch = the spine step of block j (workgroups 0..35) AND the rest of panel j (32 x 64 tiles of L[j+2.., j] = A[j+2.., j]
// inv(L_jj)^T): both need leaf j only, neither needs the other.
__global__ __launch_bounds__(SP_T) void chol_step_spine_kernel(const double* __restrict__ Ablk, int64_t lda, const double* __restrict__ inv,
                                                                int64_t ldi, double* __restrict__ Lout, int64_t ldl, double* __restrict__ D,
                                                                GemmShape panel, EpiAxpby panel_epi, long long* stamps, int step,
                                                                int npanel, GemmShape xs, EpiAxpby xs_epi, int nxs, double* xdiag,
                                                                int64_t ldxd) {
    __shared__ __attribute__((aligned(16))) double lds[SP_LDS];
    // diagnostic (stamps usually null): [start, -, end, kind] per workgroup behind the leaf launches' table (kind 4 spine, 5 panel)
    long long* stamp = (stamps && blockIdx.x < 512) ? stamps + ((int64_t)(16 + step) * 512 + blockIdx.x) * 4 : nullptr;
    if (stamp && threadIdx.x == 0) stamp[0] = (long long)__builtin_amdgcn_s_memrealtime();
    if (blockIdx.x < 36) {
        chol_spine_body(Ablk, lda, inv, ldi, Lout, ldl, D, (int)blockIdx.x, lds);
        if (stamp && threadIdx.x == 0) { stamp[2] = (long long)__builtin_amdgcn_s_memrealtime(); stamp[3] = 4; }
        return;
    }
    int t = (int)blockIdx.x - 36;
    if (t < npanel) {
        gemm_f64_tile<true, true, 32, 64, 16, 2, 2, EpiAxpby>(panel, panel_epi, t / 2, t % 2, 0, lds);
        if (stamp && threadIdx.x == 0) { stamp[2] = (long long)__builtin_amdgcn_s_memrealtime(); stamp[3] = 5; }
        return;
    }
    t -= npanel;
    if (t < nxs) {                              // XrowJob: Xt[0:128j, j] = -Tt inv(L_jj)^T
        gemm_f64_tile<true, true, 32, 64, 16, 2, 2, EpiAxpby>(xs, xs_epi, t / 2, t % 2, 0, lds);
        if (stamp && threadIdx.x == 0) { stamp[2] = (long long)__builtin_amdgcn_s_memrealtime(); stamp[3] = 7; }
        return;
    }
    if (xdiag != nullptr)                       // XrowJob: Xt[j, j] = inv(L_jj)^T, zeros below its diagonal
        for (int v = threadIdx.x; v < NB * NB; v += SP_T) {
            const int r = v / NB, c = v % NB;
            xdiag[(int64_t)r * ldxd + c] = (r <= c) ? inv[(int64_t)c * ldi + r] : 0.0;
        }
}

// the same transposed copy as a launch of its own (behind the last leaf there is no spine launch)
__global__ __launch_bounds__(256) void copy_inverse_transposed_kernel(const double* __restrict__ inv, int64_t ldi, double* __restrict__ xdiag,
                                                                       int64_t ldxd) {
    for (int v = blockIdx.x * 256 + threadIdx.x; v < NB * NB; v += gridDim.x * 256) {
        const int r = v / NB, c = v % NB;
        xdiag[(int64_t)r * ldxd + c] = (r <= c) ? inv[(int64_t)c * ldi + r] : 0.0;
    }
}

// ---- host orchestration -------------------------------------------------------------------------------

// invw: [ceil(dp/OB)] x (OB*OB inverse block, ld OB) followed by ceil(dp/OB) x (TB*TB scratch)
static inline double* inv_block(double* invw, int64_t J) { return invw + J * (int64_t)OB * OB; }
static inline const double* inv_block(const double* invw, int64_t J) { return invw + J * (int64_t)OB * OB; }

static int env_flag(const char* name, int dflt);

// Completes inv(L_JJ) for every OB x OB diagonal block from the leaf's 128 x 128 inverses, two levels of
//   inv([[A,0],[C,B]]) = [[A^-1, 0], [-B^-1 C A^-1, B^-1]]      (batched over the blocks)
static void build_block_inverses(const double* L, int64_t dp, int64_t lda, double* invw, hipStream_t st, int nmat = 1,
                                 int64_t s_mat = 0, int64_t s_inv = 0) {
    // nmat matrices (strides s_mat for L, s_inv for their inverse workspaces) share every launch: the products are
    // batched over the diagonal blocks AND over the matrices (GemmShape.batch2)
    const int64_t nob = (dp + OB - 1) / OB, nfull = dp / OB, rem = dp % OB;
    double* tmp = invw + nob * (int64_t)OB * OB;
    auto product = [&](const double* A, int64_t ldA, int64_t sA, int64_t sA2, bool b_lower, const double* B, int64_t ldB,
                       int64_t sB, int64_t sB2, double* C, int64_t ldC, int64_t sC, int64_t sC2, int M, int N, int K,
                       double alpha, int cnt) {
        GemmShape p{A, ldA, B, ldB, M, N, K, 0, sA, sB, cnt};
        p.sA2 = sA2; p.sB2 = sB2; p.batch2 = nmat;
        p.tri = b_lower ? 2 : 0;   // B stored [k][n] and lower triangular: zero for k < n
        EpiAxpby e{C, ldC, alpha, 0.0, sC};
        e.sC2 = sC2;
        launch_gemm_f64<true, false>(p, e, st, 2);
    };
    ScopedProf sp(KC_INV_BLOCK, st);
    const int64_t sI = (int64_t)OB * OB, sT = (int64_t)TB * TB, sL = (int64_t)OB * lda + OB;
    // level a: 256-blocks from pairs of 128-inverses.  u selects the pair inside an OB block.  ONE matrix with whole OB blocks
    // only (the N x N system of the dual solver at N = 1000): its second batch dimension is free, so the two pairs of every
    // block share a launch (two launches instead of four on the tail behind the last leaf, ~8 us each).
    if (nmat == 1 && rem == 0 && nfull > 0) {
        const int64_t hopL = 2 * NB * (lda + 1), hopI = 2 * NB * (int64_t)(OB + 1), hopT = (int64_t)NB * TB;
        const double* Cb = L + (int64_t)NB * lda;
        double* Ai = inv_block(invw, 0);
        double* Bi = inv_block(invw, 0) + NB * (int64_t)(OB + 1);
        double* X = inv_block(invw, 0) + NB * (int64_t)OB;
        GemmShape p1{Cb, lda, Ai, OB, NB, NB, NB, 0, sL, sI, (int)nfull};
        p1.sA2 = hopL; p1.sB2 = hopI; p1.batch2 = 2;
        p1.tri = 2;
        EpiAxpby e1{tmp, TB, 1.0, 0.0, sT};
        e1.sC2 = hopT;
        launch_gemm_f64<true, false>(p1, e1, st, 2);                                                 // T = C A^-1
        GemmShape p2{Bi, OB, tmp, TB, NB, NB, NB, 0, sI, sT, (int)nfull};
        p2.sA2 = hopI; p2.sB2 = hopT; p2.batch2 = 2;
        EpiAxpby e2{X, OB, -1.0, 0.0, sI};
        e2.sC2 = hopI;
        launch_gemm_f64<true, false>(p2, e2, st, 2);                                                 // X = -B^-1 T
    } else
    for (int u = 0; u < 2; ++u) {
        auto level_a = [&](int64_t J0, int cnt) {
            const double* Cb = L + ((J0 * 4 + 2 * u + 1) * NB) * lda + (J0 * 4 + 2 * u) * NB;
            double* Ai = inv_block(invw, J0) + (2 * u * NB) * (int64_t)(OB + 1);
            double* Bi = inv_block(invw, J0) + ((2 * u + 1) * NB) * (int64_t)(OB + 1);
            double* X = inv_block(invw, J0) + ((2 * u + 1) * NB) * (int64_t)OB + 2 * u * NB;
            double* T = tmp + J0 * sT;
            product(Cb, lda, sL, s_mat, true, Ai, OB, sI, s_inv, T, TB, sT, s_inv, NB, NB, NB, 1.0, cnt);      // T = C A^-1
            product(Bi, OB, sI, s_inv, false, T, TB, sT, s_inv, X, OB, sI, s_inv, NB, NB, NB, -1.0, cnt);      // X = -B^-1 T
        };
        if (nfull > 0) level_a(0, (int)nfull);
        if (rem >= (u + 1) * 2 * NB) level_a(nfull, 1);
    }
    // level b: the OB block from its two 256-halves (the lower half may be short in the last block)
    auto level_b = [&](int64_t J0, int cnt, int mrows) {
        const double* Cb = L + ((J0 * 4 + 2) * NB) * lda + (J0 * 4) * NB;
        double* Ai = inv_block(invw, J0);
        double* Bi = inv_block(invw, J0) + (2 * NB) * (int64_t)(OB + 1);
        double* X = inv_block(invw, J0) + (2 * NB) * (int64_t)OB;
        double* T = tmp + J0 * sT;
        product(Cb, lda, sL, s_mat, true, Ai, OB, sI, s_inv, T, TB, sT, s_inv, mrows, 2 * NB, 2 * NB, 1.0, cnt);
        product(Bi, OB, sI, s_inv, false, T, TB, sT, s_inv, X, OB, sI, s_inv, mrows, 2 * NB, mrows, -1.0, cnt);
    };
    if (nfull > 0) level_b(0, (int)nfull, 2 * NB);
    if (rem > 2 * NB) level_b(nfull, 1, (int)(rem - 2 * NB));
}

// Graphs are captured on an internal stream: the caller's stream may be the legacy default stream, which cannot capture.
namespace {
constexpr int MAX_DEVICES = 64;
hipStream_t g_capture_stream[MAX_DEVICES] = {};      // one per device: a stream belongs to the device current at its creation
std::mutex g_state_mutex;                           // graph cache + capture streams (entry points may be called from threads)
int current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev;
}
int capture_stream_init(int dev, hipStream_t* out) {
    if (dev < 0 || dev >= MAX_DEVICES) return fail(EMCID_ERR_BAD_ARG, "emcid graph", "device ordinal out of range");
    if (!g_capture_stream[dev] && hipStreamCreateWithFlags(&g_capture_stream[dev], hipStreamNonBlocking) != hipSuccess)
        return fail(EMCID_ERR_HIP, "emcid graph", "hipStreamCreateWithFlags");
    *out = g_capture_stream[dev];
    return EMCID_OK;
}
}  // namespace

static int env_flag(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// Two-level blocked Cholesky on one stream.  Outer blocks of OB = 512 columns are brought up to date LEFT-looking
// (one GEMM against all previous columns, contraction depth = their count, so the Schur complement is read and
// written once per 512 columns instead of once per 128); inside an outer block the classic right-looking
// leaf -> panel -> trailing-update runs with NB = 128, its updates confined to the block's own <= 384 remaining columns.
static int cholesky_serial(double* A, double* L, int64_t dp, int64_t lda, double* invw, int* info, hipStream_t st,
                           int nbatch = 1, int64_t s_mat = 0, int64_t s_inv = 0) {
    // nbatch independent matrices (A + b*s_mat, L + b*s_mat, invw + b*s_inv; s_inv == inv_doubles(dp)) are factored by
    // the same launches: every GEMM is batched over blockIdx.z and the leaf runs one workgroup per matrix.
    hipLaunchKernelGGL(zero_f64_kernel, dim3(1024), dim3(256), 0, st, invw, inv_doubles(dp) * nbatch);
    const int nob = (int)((dp + OB - 1) / OB);
    for (int J = 0; J < nob; ++J) {
        const int64_t c0 = (int64_t)J * OB;
        const int w = (int)((dp - c0) < OB ? (dp - c0) : OB);
        if (J > 0) {
            // A[c0:, c0:c0+w] -= L[c0:, 0:c0] L[c0:c0+w, 0:c0]^T   (its top-left corner is on the diagonal)
            const double* Lr = L + c0 * lda;
            GemmShape g{Lr, lda, Lr, lda, (int)(dp - c0), w, (int)c0, 1, s_mat, s_mat, nbatch};
            ScopedProf sp(KC_CHOL_TRAIL, st);
            launch_gemm_f64<true, true>(g, EpiAxpby{A + c0 * lda + c0, lda, -1.0, 1.0, s_mat}, st);
        }
        for (int jj = 0; jj < w / NB; ++jj) {
            const int64_t o = c0 + (int64_t)jj * NB;
            double* inv = inv_block(invw, J) + (jj * NB) * (int64_t)(OB + 1);
            {
                ScopedProf sp(KC_CHOL_LEAF, st);
                hipLaunchKernelGGL(chol_leaf_kernel, dim3(nbatch), dim3(LEAF_T), 0, st, A + o * lda + o, lda, L + o * lda + o,
                                   lda, inv, (int64_t)OB, info, (int)o, (long long*)nullptr, s_mat, s_inv);
            }
            const int m = (int)(dp - o - NB);
            if (m == 0) break;
            GemmShape ps{A + (o + NB) * lda + o, lda, inv, OB, m, NB, NB, 0, s_mat, s_inv, nbatch};
            ps.tri = 1;   // inv(L11) is lower triangular: B(k, n) = inv[n][k] vanishes for k > n
            {
                ScopedProf sp(KC_CHOL_PANEL, st);
                launch_gemm_f64<true, true>(ps, EpiAxpby{L + (o + NB) * lda + o, lda, 1.0, 0.0, s_mat}, st);
            }
            const int wi = (int)(c0 + w - (o + NB));   // columns of this outer block still to the right
            if (wi > 0) {
                const double* L21 = L + (o + NB) * lda + o;
                GemmShape ts{L21, lda, L21, lda, m, wi, NB, 1, s_mat, s_mat, nbatch};
                ScopedProf sp(KC_CHOL_INNER, st);
                launch_gemm_f64<true, true>(ts, EpiAxpby{A + (o + NB) * lda + (o + NB), lda, -1.0, 1.0, s_mat}, st);
            }
        }
    }
    build_block_inverses(L, dp, lda, invw, st, nbatch, s_mat, s_inv);
    return check_launch("emcid_cholesky_f64");
}


// ---- small Cholesky, two heterogeneous launches per 128 columns (n <= 2048) -------------------------------------------------------
// Step j:   launch A_j = { leaf j }  +  { trailing update of step j-1 minus the block the leaf needs }
//           launch B_j = { spine step j: L[j+1, j] and the next diagonal block }  +  { the rest of panel j }
// Everything a launch contains depends on earlier LAUNCHES only, so the parts run side by side on the chip and the serial chain
// per step is leaf + spine step + two kernel boundaries (~48 us) instead of leaf + panel + trailing update + three (~65 us) — with
// one stream, i.e. without the parallel graph branches that sank cholesky_lookahead.
static int cholesky_fused_steps(double* A, double* L, int64_t n, int64_t lda, double* invw, int* info, hipStream_t st,
                                const ShadowJob* shadow = nullptr, const XrowJob* xrow = nullptr) {
    const int nb = (int)(n / NB);
    hipLaunchKernelGGL(zero_f64_kernel, dim3(1024), dim3(256), 0, st, invw, inv_doubles(n));
    for (int j = 0; j < nb; ++j) {
        const int64_t o = (int64_t)j * NB;
        double* inv = inv_block(invw, j / 4) + ((j % 4) * NB) * (int64_t)(OB + 1);
        {   // A_j
            const int M = (j >= 1) ? (int)(n - o - NB) : 0;           // rows j+1.. still to be updated by column j-1
            GemmShape tr{L, lda, L, lda, 0, 0, NB, 1};
            EpiAxpby te{A, lda, -1.0, 1.0};
            int ntiles = 0;
            if (M > 0) {
                tr = GemmShape{L + (o + NB) * lda + (o - NB), lda, L + o * lda + (o - NB), lda, M, M + NB, NB, 1};
                tr.lower_shift = NB;
                te.C = A + (o + NB) * lda + o;
                const int Mb = M / NB;
                ntiles = Mb * (Mb + 3) / 2;
            }
            ShadowJob sh{};
            if (shadow) sh = *shadow;
            sh.stamps = g_step_stamps;
            sh.xcd_gx = 0;
            sh.nofast = 0;
            sh.fuse_pair = 1;
            if (sh.wgs && sh.wgs % 8 == 0) {
                const int mb = (sh.M + SH_BM - 1) / SH_BM, np_ = ((sh.N + SH_BN - 1) / SH_BN + 1) / 2;
                // (shadow ids with the same s_ % 8 land on the same XCD whatever the id of the first one is)
                for (int gx : {4, 2, 8, 1})
                    if (mb % gx == 0 && np_ % (8 / gx) == 0) { sh.xcd_gx = gx; break; }
            }
            const int shadow_ids = sh.wgs;
            // XrowJob: Tt = Xt[0:o, 0:o] L[j, 0:j]^T on 32 x 64 tiles (K range from the tile's own first row on: X is lower
            // triangular, Xt upper)
            GemmShape xr{L, lda, L, lda, 0, 0, NB, 0};
            EpiAxpby xe{A, lda, 1.0, 0.0};
            int nxr = 0;
            if (xrow && j >= 1) {
                xr = GemmShape{xrow->Xt, xrow->ldx, L + o * lda, lda, (int)o, NB, (int)o, 0};
                xr.tri = 8;
                xr.pf = 1;
                xe = EpiAxpby{xrow->Tt, NB, 1.0, 0.0};
                nxr = (int)(o / 32) * 2;
            }
            ScopedProf sp(KC_CHOL_LEAF, st);
            hipLaunchKernelGGL(chol_step_leaf_kernel, dim3(1 + ntiles + nxr + shadow_ids), dim3(LEAF_T), 0, st, A + o * lda + o, lda,
                               L + o * lda + o, lda, inv, (int64_t)OB, info, (int)o, tr, te, ntiles, sh, j, nb, xr, xe, nxr);
        }
        if (j == nb - 1) {
            if (xrow) {     // the last block row of X: its second product and its diagonal block, no spine launch to ride in
                ScopedProf sp(KC_INV_BLOCK, st);
                GemmShape g{xrow->Tt, NB, inv, OB, (int)o, NB, NB, 0};
                g.tri = 1;      // B(k, n) = inv[n][k], zero for k > n
                launch_gemm_f64<true, true>(g, EpiAxpby{xrow->Xt + o, xrow->ldx, -1.0, 0.0}, st, 2);
                hipLaunchKernelGGL(copy_inverse_transposed_kernel, dim3(16), dim3(256), 0, st, inv, (int64_t)OB,
                                   xrow->Xt + o * xrow->ldx + o, xrow->ldx);
            }
            break;
        }
        {   // B_j
            const int m2 = (int)(n - o - 2 * NB);
            GemmShape ps{A, lda, inv, OB, 0, NB, NB, 0};
            EpiAxpby pe{L, lda, 1.0, 0.0};
            int npanel = 0;
            if (m2 > 0) {
                ps = GemmShape{A + (o + 2 * NB) * lda + o, lda, inv, OB, m2, NB, NB, 0};
                ps.tri = 1;
                ps.pf = 1;
                pe.C = L + (o + 2 * NB) * lda + o;
                npanel = (m2 / 32) * 2;
            }
            // XrowJob: Xt[0:o, j] = -Tt inv(L_jj)^T on 32 x 64 tiles, and one workgroup for Xt[j, j] = inv(L_jj)^T
            GemmShape xs{A, lda, inv, OB, 0, NB, NB, 0};
            EpiAxpby xse{L, lda, -1.0, 0.0};
            int nxs = 0;
            double* xdiag = nullptr;
            if (xrow) {
                if (j >= 1) {
                    xs = GemmShape{xrow->Tt, NB, inv, OB, (int)o, NB, NB, 0};
                    xs.tri = 1;      // B(k, n) = inv[n][k], zero for k > n
                    xs.pf = 1;
                    xse.C = xrow->Xt + o;
                    xse.ldc = xrow->ldx;
                    nxs = (int)(o / 32) * 2;
                }
                xdiag = xrow->Xt + o * xrow->ldx + o;
            }
            ScopedProf sp(KC_CHOL_PANEL, st);
            hipLaunchKernelGGL(chol_step_spine_kernel, dim3(36 + npanel + nxs + (xdiag ? 1 : 0)), dim3(SP_T), 0, st,
                               A + (o + NB) * lda + o, lda, inv, (int64_t)OB, L + (o + NB) * lda + o, lda, A + (o + NB) * lda + o + NB,
                               ps, pe, g_step_stamps, j, npanel, xs, xse, nxs, xdiag, xrow ? xrow->ldx : 0);
        }
    }
    if (!xrow) build_block_inverses(L, n, lda, invw, st);      // (with an XrowJob the consumers read Xt, not the 512-block inverses)
    return check_launch("emcid_cholesky_f64");
}

// (A two-stream look-ahead schedule — spine leaf -> one panel block -> diagonal update on the caller's stream, bulk
// panel/trailing on a side stream — was built and measured 6-11 % SLOWER, eager and as a graph: the leaf needs a
// whole CU's LDS, so it cannot start while the bulk GEMM keeps every CU populated.  Kept serial.)
static inline bool cholesky_takes_shadow(int64_t dp) {      // the schedule whose leaf launches can carry a ShadowJob
    return dp <= 2048 && dp >= 2 * NB;
}

static int cholesky_impl(double* A, double* L, int64_t dp, int64_t lda, double* invw, int* info, hipStream_t st,
                         const ShadowJob* shadow = nullptr, const XrowJob* xrow = nullptr) {
    // (A look-ahead schedule on two streams — spine on the caller's, bulk on a side stream — was built in rounds 1-2 and lost:
    // 13.3 -> 22.3 ms per device step inside the captured graph, every fork / join between capture streams costing ~70 us;
    // removed in round 5, DESIGN.md "measured and dropped".)
    if (cholesky_takes_shadow(dp)) return cholesky_fused_steps(A, L, dp, lda, invw, info, st, shadow, xrow);
    if (shadow || xrow) return fail(EMCID_ERR_BAD_ARG, "cholesky_impl", "shadow / inverse job without the fused schedule");
    return cholesky_serial(A, L, dp, lda, invw, info, st);
}

// Bt[M, dp] := Bt (L L^T)^-1, right-looking over OB-wide column blocks: multiply by the inverted diagonal
// block, then one rank-OB GEMM update of every remaining column (forward), the same backward.
// forward half:  Yt L^T = Bt   (Bt is consumed as scratch)
static void trsm_forward(const double* L, int64_t dp, int64_t lda, const double* invw, double* Bt, double* Yt, int M,
                         int64_t ldb, hipStream_t st) {
    const int nob = (int)((dp + OB - 1) / OB);
    for (int J = 0; J < nob; ++J) {
        const int64_t c = (int64_t)J * OB;
        const int w = (int)((dp - c) < OB ? (dp - c) : OB);
        const double* inv = inv_block(invw, J);
        GemmShape a{Bt + c, ldb, inv, OB, M, w, w, 0};
        a.tri = 1;   // B(k, n) = inv[n][k], zero for k > n
        {
            ScopedProf sp(KC_TRSM_DIAG, st);
            launch_gemm_f64<true, true>(a, EpiAxpby{Yt + c, ldb, 1.0, 0.0}, st);
        }
        const int m = (int)(dp - c - w);
        if (m > 0) {
            GemmShape b{Yt + c, ldb, L + (c + w) * lda + c, lda, M, m, w, 0};
            ScopedProf sp(KC_TRSM_UPDATE, st);
            launch_gemm_f64<true, true>(b, EpiAxpby{Bt + c + w, ldb, -1.0, 1.0}, st);
        }
    }
}

// backward half:  Xt L = Yt   (Yt is consumed as scratch, Xt may be any buffer of the same shape)
static void trsm_backward(const double* L, int64_t dp, int64_t lda, const double* invw, double* Yt, double* Xt, int M,
                          int64_t ldb, hipStream_t st) {
    const int nob = (int)((dp + OB - 1) / OB);
    for (int J = nob - 1; J >= 0; --J) {
        const int64_t c = (int64_t)J * OB;
        const int w = (int)((dp - c) < OB ? (dp - c) : OB);
        const double* inv = inv_block(invw, J);
        GemmShape a{Yt + c, ldb, inv, OB, M, w, w, 0};
        a.tri = 2;   // B(k, n) = inv[k][n], zero for k < n
        {
            ScopedProf sp(KC_TRSM_DIAG, st);
            launch_gemm_f64<true, false>(a, EpiAxpby{Xt + c, ldb, 1.0, 0.0}, st);
        }
        if (c > 0) {
            GemmShape b{Xt + c, ldb, L + c * lda, lda, M, (int)c, w, 0};
            ScopedProf sp(KC_TRSM_UPDATE, st);
            launch_gemm_f64<true, false>(b, EpiAxpby{Yt, ldb, -1.0, 1.0}, st);
        }
    }
}

static int cholesky_solve_impl(const double* L, int64_t dp, int64_t lda, const double* invw, double* Bt, double* Yt,
                               int64_t Mrows, int64_t ldb, hipStream_t st) {
    trsm_forward(L, dp, lda, invw, Bt, Yt, (int)Mrows, ldb, st);
    trsm_backward(L, dp, lda, invw, Yt, Bt, (int)Mrows, ldb, st);   // Xt written over Bt
    return check_launch("emcid_cholesky_solve_f64");
}

// X = inv(L), explicit, for `nbatch` factors at once (the dual solver's M = lam*C' factors: they do not depend on the
// concepts, so a layer's two triangular solves become two GEMMs against X).  Recursive halving over the OB-blocks:
//   inv([[L11, 0], [L21, L22]]) = [[X11, 0], [-X22 (L21 X11), X22]]
// with the diagonal OB-blocks taken from build_block_inverses.  Three quarters of the ~dp^3/3 flops sit in the two
// products of the top split (dp/2 cubed, one operand triangular: its zero part is skipped), which is what makes this
// form run at GEMM rates; a block-row recurrence has the same flops in 512-row slivers.
// T: scratch, nbatch x (>= (dp/2)^2 with leading dimension lda) doubles, stride s_mat.  Only the lower triangle of X
// (and the zeros inside its diagonal 128-blocks) is written; consumers never read anything else.
__global__ __launch_bounds__(256) void copy_diag_inverse_kernel(const double* __restrict__ invw, int64_t s_inv, double* __restrict__ X,
                                                                 int64_t ldx, int64_t s_x, int64_t dp) {
    const int64_t r = blockIdx.x;                 // global row
    const int64_t J = r / OB, i = r % OB;
    const int w = (int)((dp - J * OB) < OB ? (dp - J * OB) : OB);
    const double* src = invw + blockIdx.y * s_inv + J * (int64_t)OB * OB + i * OB;
    double* dst = X + blockIdx.y * s_x + r * ldx + J * OB;
    // zeros only where a diagonal 128-block could be read above the diagonal
    const int hi = (int)((i / NB + 1) * NB);
    for (int j = threadIdx.x; j < (hi < w ? hi : w); j += 256) dst[j] = (j <= i) ? src[j] : 0.0;
}

__global__ __launch_bounds__(256) void zero2d_f64_kernel(double* __restrict__ p, int64_t ld, int64_t s_batch, int cols) {
    double* row = p + blockIdx.y * s_batch + (int64_t)blockIdx.x * ld;
    for (int j = threadIdx.x; j < cols; j += 256) row[j] = 0.0;
}

static int build_full_inverse(const double* L, int64_t dp, int64_t lda, const double* invw, double* X, double* T, int nbatch,
                              int64_t s_mat, int64_t s_inv, hipStream_t st) {
    ScopedProf sp(KC_INV_BUILD, st);
    hipLaunchKernelGGL(copy_diag_inverse_kernel, dim3((unsigned)dp, (unsigned)nbatch), dim3(256), 0, st, invw, s_inv, X, lda, s_mat, dp);
    const int nob = (int)((dp + OB - 1) / OB);
    // 32x64 tiles in mirrored pairs: 1.06 ms for 4 x 3072^2 vs 1.30 ms with the launcher's own choice
    // (scripts/inverse_alone.py sweeps these)
    constexpr int cfg_a = 2, cfg_b = 2, pair_ = 1;
    // Recursion over block ranges [b0, b1).  When the two halves of a range are the same problem (equal block counts, no
    // short last block) they are solved ONCE with the copy count doubled: `reps` copies of the range sit `stride` blocks
    // apart on the diagonal and share every launch through the GEMM's second batch dimension.
    const bool regular = dp % OB == 0;
    std::function<void(int, int, int, int)> rec = [&](int b0, int b1, int reps, int stride) {
        const int n = b1 - b0;
        if (n <= 1) return;
        const int mid = b0 + n / 2, left = mid - b0, right = b1 - mid;
        if (regular && left == right && (reps == 1 || stride == 2 * left)) {
            rec(b0, mid, reps * 2, left);
        } else {
            rec(b0, mid, reps, stride);
            rec(mid, b1, reps, stride);
        }
        const int64_t r0 = (int64_t)b0 * OB, rm = (int64_t)mid * OB, r1 = (int64_t)b1 * OB < dp ? (int64_t)b1 * OB : dp;
        const int m = (int)(r1 - rm), nn = (int)(rm - r0);
        const int64_t hop = (int64_t)stride * OB * (lda + 1), thop = (int64_t)m * lda;      // between copies: diagonal / scratch
        GemmShape a{L + rm * lda + r0, lda, X + r0 * lda + r0, lda, m, nn, nn, 0, s_mat, s_mat, nbatch};
        a.sA2 = hop; a.sB2 = hop; a.batch2 = reps;
        a.tri = 2;   // B(k, n) = X11[k][n], zero for k < n
        a.pair = pair_;
        EpiAxpby ea{T, lda, 1.0, 0.0, s_mat};
        ea.sC2 = thop;
        launch_gemm_f64<true, false>(a, ea, st, cfg_a);
        GemmShape b{X + rm * lda + rm, lda, T, lda, m, nn, m, 0, s_mat, s_mat, nbatch};
        b.sA2 = hop; b.sB2 = thop; b.batch2 = reps;
        b.tri = 4;   // A(m, k) = X22[m][k], zero for k > m
        b.pair = pair_;
        EpiAxpby eb{X + rm * lda + r0, lda, -1.0, 0.0, s_mat};
        eb.sC2 = hop;
        launch_gemm_f64<true, false>(b, eb, st, cfg_b);
    };
    rec(0, nob, 1, 0);
    return check_launch("build_full_inverse");
}

// Both GEMMs against X contract over a triangular K range; they run as stream-K over 128x128 tiles (gemm_f64.h): the
// (tile, K-step) space cut into 256 equal runs, whole tiles stored, the two partial tiles of a run added atomically
// into a zeroed output.  Measured for 1024 x 3072 x 3072: 218 us (zeroing included) against 253 us for mirrored 32x64
// tile pairs and 306 us for plain 64x64 tiles; 768 rows: 188 / 238 / 299 us (scripts/mb_tri.py).
static const int kStreamKWgs = 256;

// With the interior fast path of the small-tile ring, mirrored PAIRS of 32 x 64 tiles (every workgroup contracts over the same
// total depth, three 4-wave workgroups per compute unit, no split, no fix-up) beat the stream-K form whenever their count fills
// the chip's 768 slots evenly (scripts/mb_tri_small.py, d = 3072, us stream-K / pairs: 1024 rows 187 / 172 and 195 / 177; 768
// rows 146 / 161; 512 rows 104 / 118), and for very few rows on the backward product (128 rows: 114 / 80; 256: 98 / 81), where a
// stream-K run is mostly fix-up.  EMCID_TRI_PAIRS=0 keeps stream-K everywhere.
static inline bool pairs_fill_the_chip(int rows, int64_t dp, bool backward) {
    const int64_t wgs = (int64_t)((rows + 31) / 32) * (((dp + 63) / 64 + 1) / 2);
    // (at most 128 rows: the stream-K form on 64 x 64 tiles is ahead — P = Yt X of a 100-concept edit 75 -> 54 us, inv_apply 0.51 -> 0.42 ms per call)
    if (backward && rows <= 256 && rows > 128) return true;
    if (wgs <= 768) return wgs >= 614;          // one round at >= 80 % of the slots (d = 5120, 1024 rows: 640 -> 515 vs 542 us)
    const int64_t tail = wgs % 768;
    return tail == 0 || tail >= 700;
}

static inline bool few_rows_ksplit(int rows, int64_t dp) {
    static const int on = env_flag("EMCID_FEW_ROWS_KSPLIT", 1);
    return on && rows <= 128 && dp >= 1024;
}

// Yt[rows, dp] = Kt[rows, dp] * X^T  (= Kt L^-T: the forward substitution as one GEMM)
static void apply_inverse_forward(const double* X, int64_t dp, const double* Kt, double* Yt, int rows, hipStream_t st,
                                  double* sk_work = nullptr) {
    ScopedProf sp(KC_INV_APPLY, st);
    GemmShape g{Kt, dp, X, dp, rows, (int)dp, (int)dp, 0};
    g.tri = 1;       // B(k, n) = X[n][k], zero for k > n
    if (pairs_fill_the_chip(rows, dp, false)) {
        g.pair = 1;
        launch_gemm_f64<true, true>(g, EpiAxpby{Yt, dp, 1.0, 0.0}, st, 2);
        return;
    }
    if (few_rows_ksplit(rows, dp)) {
        // one row of 128 x 128 tiles (a 100-concept edit): 32 x 64 tiles with the contraction cut in four, partial sums added with
        // f64 atomics into a zeroed result — 36 us against 50 on the two-phase stream-K form (profiles/r06_mb_tri_small.txt)
        hipLaunchKernelGGL(zero2d_f64_kernel, dim3((unsigned)rows, 1u), dim3(256), 0, st, Yt, dp, (int64_t)0, (int)dp);
        g.ksplit = 4;
        launch_gemm_f64<true, true>(g, EpiAxpby{Yt, dp, 1.0, 1.0}, st, 2);
        return;
    }
    if (sk_work && launch_gemm_f64_streamk2<true, true>(g, EpiAxpby{Yt, dp, 1.0, 0.0}, st, kStreamKWgs, sk_work)) return;
    g.pair = 1;      // (no stream-K workspace, or more tiles than its ticket counters: mirrored tile pairs)
    launch_gemm_f64<true, true>(g, EpiAxpby{Yt, dp, 1.0, 0.0}, st, 2);
}

// C[rows, ncols] (f64, leading dimension ldc) = V[rows, dp] * X  (= V L^-1: the backward substitution as one GEMM)
static void apply_inverse_backward(const double* X, int64_t dp, const double* V, int rows, int ncols, double* C, int64_t ldc,
                                   hipStream_t st, double* sk_work = nullptr) {
    ScopedProf sp(KC_INV_APPLY, st);
    GemmShape g{V, dp, X, dp, rows, ncols, (int)dp, 0};
    g.tri = 2;       // B(k, n) = X[k][n], zero for k < n
    if (ncols == (int)dp && pairs_fill_the_chip(rows, dp, true)) {
        g.pair = 1;
        launch_gemm_f64<true, false>(g, EpiAxpby{C, ldc, 1.0, 0.0}, st, 2);
        return;
    }
    if (ncols == (int)dp && few_rows_ksplit(rows, dp)) {
        hipLaunchKernelGGL(zero2d_f64_kernel, dim3((unsigned)rows, 1u), dim3(256), 0, st, C, ldc, (int64_t)0, ncols);
        g.ksplit = 4;
        launch_gemm_f64<true, false>(g, EpiAxpby{C, ldc, 1.0, 1.0}, st, 2);
        return;
    }
    if (sk_work && launch_gemm_f64_streamk2<true, false>(g, EpiAxpby{C, ldc, 1.0, 0.0}, st, kStreamKWgs, sk_work)) return;
    g.pair = ncols == (int)dp ? 1 : 0;      // (no stream-K workspace / too many tiles: mirrored tile pairs where the output is the whole width)
    launch_gemm_f64<true, false>(g, EpiAxpby{C, ldc, 1.0, 0.0}, st, 2);
}

// layout of the covariance-factor workspace (emcid_factor_cov_f64): [M | L | 512-block inverses | X = inv(L)] x n_layers
static inline const double* cov_inverse(const void* cov_factor_ws, int64_t n_layers, int64_t dp, int64_t layer) {
    return (const double*)cov_factor_ws + n_layers * (2 * dp * dp + inv_doubles(dp)) + layer * dp * dp;
}

// ---- factor + solve as one cached hipGraph -----------------------------------------------------------------------
// The ~100 launches of one layer's Cholesky + block-inverse build + triangular solves take only workspace
// pointers and sizes, so the whole chain (is captured once
// per (workspace, shape) and replayed: dependent-kernel boundaries inside a graph cost ~1.2 us instead of a host
// launch each.  Not used while per-kernel event timing is on (the events would be recorded at capture time).
namespace {
struct GraphKey {
    const void* ptr[8];
    int64_t num[6];
    int64_t dev;          // device the launches were captured for (filled in by with_graph)
    bool operator==(const GraphKey& o) const { return memcmp(this, &o, sizeof(GraphKey)) == 0; }
};
struct GraphSlot { GraphKey key; hipGraphExec_t exec; hipGraph_t graph; uint64_t used; };
constexpr int GRAPH_SLOTS = 64;     // (32 was one bench process short: its SDXL record re-captured graphs in every call once the other records had filled the cache)
GraphSlot g_graphs[GRAPH_SLOTS];
int g_graph_n = 0;
uint64_t g_graph_clock = 0;

GraphKey make_key(int tag, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> nums) {
    GraphKey k;
    memset(&k, 0, sizeof(k));
    int i = 0;
    for (const void* p : ptrs) k.ptr[i++] = p;
    i = 0;
    k.num[5] = tag;
    for (int64_t n : nums) k.num[i++] = n;
    return k;
}
}  // namespace

// Runs `body(stream)` — a chain of launches whose arguments are fully determined by `key` — as a cached hipGraph.
template <class F>
static int with_graph(const GraphKey& key_in, hipStream_t st, F&& body) {
    static const int use_graph = env_flag("EMCID_GRAPH", 1);
    if (!use_graph || g_prof_mask != 0) return body(st);
    std::lock_guard<std::mutex> lock(g_state_mutex);
    GraphKey key = key_in;
    key.dev = current_device();     // the caller made the buffers' device current (emcid_amd/hip.py does; see emcid_hip.h)
    GraphSlot* slot = nullptr;
    for (int i = 0; i < g_graph_n; ++i)
        if (g_graphs[i].key == key) { slot = &g_graphs[i]; break; }
    if (!slot) {
        hipStream_t cap = nullptr;
        EMCID_TRY(capture_stream_init((int)key.dev, &cap));
        if (hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) != hipSuccess)
            return fail(EMCID_ERR_HIP, "emcid graph", "hipStreamBeginCapture");
        const int rc = body(cap);
        hipGraph_t graph = nullptr;
        const hipError_t ec = hipStreamEndCapture(cap, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ec != hipSuccess || !graph) return fail(EMCID_ERR_HIP, "emcid graph", "hipStreamEndCapture");
        hipGraphExec_t exec = nullptr;
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGraphDestroy(graph);
            return fail(EMCID_ERR_HIP, "emcid graph", "hipGraphInstantiate");
        }
        if (g_graph_n < GRAPH_SLOTS) {
            slot = &g_graphs[g_graph_n++];
        } else {   // evict the least recently used graph
            slot = &g_graphs[0];
            for (int i = 1; i < GRAPH_SLOTS; ++i)
                if (g_graphs[i].used < slot->used) slot = &g_graphs[i];
            (void)hipDeviceSynchronize();   // the evicted graph may still be executing
            (void)hipGraphExecDestroy(slot->exec);
            (void)hipGraphDestroy(slot->graph);
        }
        slot->key = key; slot->exec = exec; slot->graph = graph;
    }
    slot->used = ++g_graph_clock;
    if (hipGraphLaunch(slot->exec, st) != hipSuccess) return fail(EMCID_ERR_HIP, "emcid graph", "hipGraphLaunch");
    return EMCID_OK;
}

static int factor_and_solve(double* A, double* L, int64_t dp, int64_t lda, double* invw, int* info, double* B, double* Y,
                            int64_t rows, int64_t ldb, hipStream_t st) {
    return with_graph(make_key(1, {A, L, invw, info, B, Y}, {dp, lda, rows, ldb}), st, [&](hipStream_t s) {
        EMCID_TRY(cholesky_impl(A, L, dp, lda, invw, info, s));
        return cholesky_solve_impl(L, dp, lda, invw, B, Y, rows, ldb, s);
    });
}

// ---- dual (Woodbury) solver -----------------------------------------------------------------------------------------
// A = M + Kt^T Kt with M = lam*C' independent of the concepts.  Then  Xt = Kt A^-1 = (I + Pt Kt^T)^-1 Pt,  Pt = Kt M^-1:
// the d x d factorization is of M only — done for ALL edited layers at once, batched, before (and concurrently with)
// the forward pass — and each layer factors just the Np x Np matrix S = I + Pt Kt^T.

// M[l] = lam * double(fl32(fl32(C[l]*cw)/0.5f)) on the lower triangle, identity on the padding (as EpiAssemble)
struct CovPtrs { const float* c[32]; };
__global__ __launch_bounds__(256) void scale_cov_kernel(CovPtrs cov, int d, int dp, double lam, float cw, double* __restrict__ M,
                                                         int64_t s_mat) {
    const int l = blockIdx.y;
    const int i = blockIdx.x;
    const float* C = cov.c[l];
    double* row = M + l * s_mat + (int64_t)i * dp;
    for (int j = threadIdx.x; j <= i; j += 256) {
        double v;
        if (i < d) {
            const float c1 = C[(int64_t)i * d + j] * cw;
            v = lam * (double)(c1 / 0.5f);
        } else {
            v = (i == j) ? 1.0 : 0.0;
        }
        row[j] = v;
    }
}


// S = I (full square): start value of the split-K accumulation S += Yt Yt^T
__global__ __launch_bounds__(256) void eye_f64_kernel(double* __restrict__ S, int n) {
    const int i = blockIdx.x;
    for (int j = threadIdx.x; j < n; j += 256) S[(int64_t)i * n + j] = (i == j) ? 1.0 : 0.0;
}

// S[Np, Np] = I + P Q^T on the lower tiles, K = dp deep.  Np x Np is too few output tiles for the chip, so the
// contraction is split over workgroups that add their partials into the identity with f64 atomics.
static void assemble_dual_system(const double* P, const double* Q, int64_t dp, double* S, int Np, hipStream_t st,
                                 double* sk_work = nullptr) {
    ScopedProf sp(KC_ASSEMBLE, st);
    GemmShape g{P, dp, Q, dp, Np, Np, (int)dp, 1};
    // S = I + P Q^T written once per tile, no identity pass (unless there are more tiles than ticket counters)
    if (Np >= 512 && sk_work && launch_gemm_f64_streamk2<true, true>(g, EpiAxpby{S, Np, 1.0, 0.0}, st, kStreamKWgs, sk_work, 1.0))
        return;
    hipLaunchKernelGGL(eye_f64_kernel, dim3((unsigned)Np), dim3(256), 0, st, S, Np);
    const int kt = (int)(dp / 16);
    g.ksplit = kt >= 64 ? 4 : kt >= 32 ? 2 : 1;
    launch_gemm_f64<true, true>(g, EpiAxpby{S, Np, 1.0, 1.0}, st, Np >= 512 ? 1 : 2);
}

__global__ __launch_bounds__(256) void transpose_f64_kernel(const double* __restrict__ src, int64_t lds_, double* __restrict__ dst,
                                                             int64_t ldd, int rows, int cols) {
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = (by + r < rows && bx + tx < cols) ? src[(int64_t)(by + r) * lds_ + bx + tx] : 0.0;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (bx + r < cols && by + tx < rows) dst[(int64_t)(bx + r) * ldd + by + tx] = tile[tx][r];
}

struct DualWorkspace {
    int64_t Np, dp, hp;
    int64_t off_K, off_P, off_Y, off_R, off_S, off_LS, off_invS, off_PT, off_Y2, off_V, off_U, off_SK, total;   // doubles
    DualWorkspace(int64_t N, int64_t d, int64_t h) {
        Np = round_up(N, NB);
        dp = round_up(d, NB);
        hp = round_up(h, 2);
        int64_t o = 0;
        off_K = o; o += Np * dp;
        off_P = o; o += Np * dp;
        off_Y = o; o += Np * dp;
        off_R = o; o += Np * hp;
        off_S = o; o += Np * Np;
        off_LS = o; o += Np * Np;
        off_invS = o; o += inv_doubles(Np);
        off_PT = o; o += dp * Np;
        off_Y2 = o; o += dp * Np;
        off_V = o; o += hp * dp;
        off_U = o; o += hp * dp;
        off_SK = o; o += streamk_workspace_doubles(kStreamKWgs);      // partial-tile slots + ticket counters (zero between launches)
        off_XT = o; o += Np * Np;                                     // XrowJob: inv(LS)^T ...
        off_TT = o; o += Np * NB;                                     // ... and its per-step scratch
        total = o;
    }
    int64_t off_XT, off_TT;
};

struct EditWorkspace {
    int64_t Np, dp, hp;
    int64_t off_A, off_L, off_inv, off_B, off_Y, off_R, total;  // in doubles
    EditWorkspace(int64_t N, int64_t d, int64_t h) {
        Np = round_up(N, NPAD);
        dp = round_up(d, NB);
        hp = round_up(h, 2);
        int64_t o = 0;
        off_A = o; o += dp * dp;
        off_L = o; o += dp * dp;
        off_inv = o; o += inv_doubles(dp);
        off_B = o; o += Np * dp;
        off_Y = o; o += Np * dp;
        off_R = o; o += Np * hp;
        total = o;
    }
};

}  // namespace emcid

using namespace emcid;

extern "C" {

int emcid_abi_version(void) { return EMCID_ABI_VERSION; }

int emcid_profile_enable(unsigned class_mask) {
    if (class_mask && !g_prof_init) {
        for (int i = 0; i < PROF_MAX; ++i)
            for (int j = 0; j < 2; ++j)
                if (hipEventCreate(&g_prof_ev[i][j]) != hipSuccess) return fail(EMCID_ERR_HIP, __func__, "hipEventCreate");
        g_prof_init = true;
    }
    g_prof_mask = class_mask;
    g_prof_n = 0;
    return EMCID_OK;
}

int emcid_profile_collect(double* ms_per_class, int64_t* launches_per_class, int n_classes) {
    EMCID_CHECK_ARG(ms_per_class && launches_per_class && n_classes >= KC_COUNT);
    for (int c = 0; c < n_classes; ++c) { ms_per_class[c] = 0.0; launches_per_class[c] = 0; }
    for (int i = 0; i < g_prof_n; ++i) {
        if (hipEventSynchronize(g_prof_ev[i][1]) != hipSuccess) return fail(EMCID_ERR_HIP, __func__, "hipEventSynchronize");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof_ev[i][0], g_prof_ev[i][1]) != hipSuccess)
            return fail(EMCID_ERR_HIP, __func__, "hipEventElapsedTime");
        ms_per_class[g_prof_cls[i]] += ms;
        launches_per_class[g_prof_cls[i]] += 1;
    }
    const int dropped = (g_prof_n >= PROF_MAX) ? 1 : 0;
    g_prof_n = 0;
    return dropped ? fail(EMCID_ERR_WORKSPACE, __func__, "event pool exhausted; enable fewer classes") : EMCID_OK;
}
const char* emcid_last_error(void) { return g_last_error; }

int emcid_dgemm_f64(int ta, int tb, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                    const double* B, int64_t ldb, double beta, double* C, int64_t ldc, void* stream) {
    EMCID_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C);
    EMCID_CHECK_ARG(aligned16(A) && aligned16(B) && (lda % 2 == 0) && (ldb % 2 == 0));
    EMCID_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30));
    hipStream_t st = (hipStream_t)stream;
    GemmShape p{A, lda, B, ldb, (int)M, (int)N, (int)K, 0};
    EpiAxpby e{C, ldc, alpha, beta};
    ScopedProf sp(KC_DGEMM, st);
    // ta/tb == 0: K contiguous ([rows][K]); 1: rows contiguous ([K][rows])
    if (ta == 0 && tb == 0) launch_gemm_f64<true, true>(p, e, st);
    else if (ta == 0 && tb == 1) launch_gemm_f64<true, false>(p, e, st);
    else if (ta == 1 && tb == 0) launch_gemm_f64<false, true>(p, e, st);
    else launch_gemm_f64<false, false>(p, e, st);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int emcid_dgemm_ex_f64(int ta, int tb, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                       const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int flags, int cfg, int ksplit,
                       void* stream) {
    EMCID_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C);
    EMCID_CHECK_ARG(aligned16(A) && aligned16(B) && (lda % 2 == 0) && (ldb % 2 == 0));
    EMCID_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30) && cfg >= -1 && cfg <= 2 && (flags & ~63) == 0);
    EMCID_CHECK_ARG(ksplit == 0 || beta == 1.0);
    hipStream_t st = (hipStream_t)stream;
    GemmShape p{A, lda, B, ldb, (int)M, (int)N, (int)K, (flags >> 4) & 1};
    p.tri = flags & 15;
    p.pair = (flags >> 5) & 1;
    if (ksplit > 0) p.ksplit = ksplit;
    if (ksplit < 0) p.kchunk = -ksplit;
    EpiAxpby e{C, ldc, alpha, beta};
    ScopedProf sp(KC_DGEMM, st);
    if (ta == 0 && tb == 0) launch_gemm_f64<true, true>(p, e, st, cfg);
    else if (ta == 0 && tb == 1) launch_gemm_f64<true, false>(p, e, st, cfg);
    else if (ta == 1 && tb == 0) launch_gemm_f64<false, true>(p, e, st, cfg);
    else launch_gemm_f64<false, false>(p, e, st, cfg);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int64_t emcid_streamk_workspace_bytes(int wgs) { return wgs > 0 ? streamk_workspace_doubles(wgs) * (int64_t)sizeof(double) : 0; }

int emcid_dgemm_streamk_f64(int tb, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda, const double* B,
                            int64_t ldb, double* C, int64_t ldc, int flags, int wgs, double diag_add, void* workspace,
                            int64_t workspace_bytes, void* stream) {
    EMCID_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C && workspace && wgs > 0 && wgs <= 4096);
    EMCID_CHECK_ARG(aligned16(A) && aligned16(B) && aligned16(workspace) && (lda % 2 == 0) && (ldb % 2 == 0));
    EMCID_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30) && (flags & ~19) == 0);
    const int tri = flags & 3, lower = (flags >> 4) & 1;
    EMCID_CHECK_ARG((lower && M == N && tri == 0) || (!lower && (tri == 1 || tri == 2)));
    EMCID_CHECK_ARG(((M + 127) / 128) * ((N + 127) / 128) <= 16384);      // one ticket counter per 128 x 128 tile
    if (workspace_bytes < emcid_streamk_workspace_bytes(wgs)) return fail(EMCID_ERR_WORKSPACE, __func__, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    GemmShape p{A, lda, B, ldb, (int)M, (int)N, (int)K, lower};
    p.tri = tri;
    ScopedProf sp(KC_DGEMM, st);
    const bool launched = tb == 0 ? launch_gemm_f64_streamk2<true, true>(p, EpiAxpby{C, ldc, alpha, 0.0}, st, wgs, (double*)workspace, diag_add)
                                  : launch_gemm_f64_streamk2<true, false>(p, EpiAxpby{C, ldc, alpha, 0.0}, st, wgs, (double*)workspace, diag_add);
    if (!launched) return fail(EMCID_ERR_BAD_ARG, __func__, "more 128 x 128 output tiles than ticket counters (16384)");
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

/* diagnostic: later two-phase stream-K launches write 8 shader-clock values per workgroup to stamps_dev (NULL: stop) —
 * [0] start, [1] end, cycles in [2] K loops, [3] partial-tile publishes, [4] last-ticket reductions, [5] epilogues,
 * [6] segments, [7] run index */
/* diagnostic: chol_step_leaf_kernel writes, per launch slice s < 16 and workgroup b < 512, [start, -, end, kind] (constant
 * 100 MHz clock; kind 1 leaf, 2 trailing tile, 3 shadow tile pair) at stamps_dev[(s * 512 + b) * 4], and the same for the spine
 * launches (kind 4 spine tile, 5 panel tile) at stamps_dev[((16 + s) * 512 + b) * 4]: 32 * 512 * 4 values; null switches it off.
 * Set it before the first edit of the process (captured graphs keep the pointer they were captured with). */
int emcid_debug_step_stamps(long long* stamps_dev) {
    g_step_stamps = stamps_dev;
    return EMCID_OK;
}

int emcid_debug_streamk_stamps(long long* stamps_dev) {
    g_streamk_stamps = stamps_dev;
    return EMCID_OK;
}

int emcid_dgemm_batched_f64(int ta, int tb, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                            int64_t sA, const double* B, int64_t ldb, int64_t sB, double beta, double* C, int64_t ldc, int64_t sC,
                            int64_t batch, void* stream) {
    EMCID_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C && batch > 0 && batch <= 65535);
    EMCID_CHECK_ARG(aligned16(A) && aligned16(B) && (lda % 2 == 0) && (ldb % 2 == 0) && (sA % 2 == 0) && (sB % 2 == 0));
    EMCID_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30) && sA >= 0 && sB >= 0 && sC >= 0);
    hipStream_t st = (hipStream_t)stream;
    GemmShape p{A, lda, B, ldb, (int)M, (int)N, (int)K, 0};
    p.sA = sA; p.sB = sB; p.batch = (int)batch;
    EpiAxpby e{C, ldc, alpha, beta};
    e.sC = sC;
    ScopedProf sp(KC_DGEMM, st);
    // no K split here: with beta == 1 the launcher would add partials atomically, which is pointless for short K
    if (ta == 0 && tb == 0) launch_gemm_f64<true, true>(p, e, st);
    else if (ta == 0 && tb == 1) launch_gemm_f64<true, false>(p, e, st);
    else if (ta == 1 && tb == 0) launch_gemm_f64<false, true>(p, e, st);
    else launch_gemm_f64<false, false>(p, e, st);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int emcid_assemble_spd_f64(const float* C, int64_t ldc, const double* Kt64, int64_t Np, int64_t d, int64_t ldk, double lam,
                           float cw, double* A, int64_t lda, void* stream) {
    EMCID_CHECK_ARG(C && Kt64 && A && Np > 0 && d > 0);
    const int64_t dp = round_up(d, NB);
    EMCID_CHECK_ARG(lda >= dp && ldk >= dp && ldc >= d && (ldk % 2 == 0) && aligned16(Kt64));
    GemmShape p{Kt64, ldk, Kt64, ldk, (int)dp, (int)dp, (int)Np, 1};
    {
        ScopedProf sp(KC_ASSEMBLE, (hipStream_t)stream);
        launch_gemm_f64<false, false>(p, EpiAssemble{C, ldc, lam, cw, A, lda, (int)d}, (hipStream_t)stream);
    }
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

/* diagnostic: one leaf on a 128x128 SPD block with phase stamps (shader clock ticks) written to stamps_dev[32] */
int emcid_debug_leaf_stamps(const double* A, double* L, double* inv, int* info_dev, long long* stamps_dev, void* stream) {
    EMCID_CHECK_ARG(A && L && inv && info_dev && stamps_dev);
    hipLaunchKernelGGL(chol_leaf_kernel, dim3(1), dim3(LEAF_T), 0, (hipStream_t)stream, A, (int64_t)NB, L, (int64_t)NB, inv,
                       (int64_t)NB, info_dev, 0, stamps_dev);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int64_t emcid_inverse_workspace_doubles(int64_t dp) { return dp > 0 ? inv_doubles(round_up(dp, NB)) : 0; }

int emcid_cholesky_f64(double* A, double* L, int64_t dp, int64_t lda, double* invdiag, int* info_dev, void* stream) {
    EMCID_CHECK_ARG(A && L && invdiag && info_dev && dp > 0 && dp % NB == 0 && lda >= dp && lda % 2 == 0);
    EMCID_CHECK_ARG(aligned16(A) && aligned16(L) && aligned16(invdiag));
    return cholesky_impl(A, L, dp, lda, invdiag, info_dev, (hipStream_t)stream);
}

int emcid_cholesky_solve_f64(const double* L, int64_t dp, int64_t lda, const double* invdiag, double* Bt, double* Yt,
                             int64_t Np, int64_t ldb, void* stream) {
    EMCID_CHECK_ARG(L && invdiag && Bt && Yt && dp > 0 && dp % NB == 0 && Np > 0 && ldb >= dp && lda >= dp);
    EMCID_CHECK_ARG(aligned16(L) && aligned16(Bt) && aligned16(Yt) && lda % 2 == 0 && ldb % 2 == 0);
    return cholesky_solve_impl(L, dp, lda, invdiag, Bt, Yt, Np, ldb, (hipStream_t)stream);
}

int emcid_delta_w_f64(const double* Rt, int64_t ldr, const double* Xt, int64_t ldx, int64_t Np, int64_t h, int64_t d,
                      const float* W0, float* W, int64_t ldw, float* dW, double* U, void* stream) {
    EMCID_CHECK_ARG(Rt && Xt && Np > 0 && h > 0 && d > 0 && ldr % 2 == 0 && ldx % 2 == 0 && aligned16(Rt) && aligned16(Xt));
    EMCID_CHECK_ARG((W == nullptr) || (W0 != nullptr));
    GemmShape p{Rt, ldr, Xt, ldx, (int)h, (int)d, (int)Np, 0};
    {
        ScopedProf sp(KC_DELTA_W, (hipStream_t)stream);
        launch_gemm_f64<false, false>(p, EpiDeltaW{W0, W, ldw, dW, d, U, d}, (hipStream_t)stream);
    }
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int emcid_axpy_f32(float* W, const float* dW, int64_t n, void* stream) {
    EMCID_CHECK_ARG(W && dW && n > 0);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(axpy_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, W, dW, n);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int64_t emcid_edit_workspace_bytes(int64_t N, int64_t d, int64_t h) {
    if (N <= 0 || d <= 0 || h <= 0) return 0;
    return EditWorkspace(N, d, h).total * (int64_t)sizeof(double);
}

static int edit_layer_impl(const float* K, const float* Zc, const float* zs_t, const float* C, int64_t N, int64_t d, int64_t h,
                           double lam, double edit_weight, int layers_left, int64_t n_lo, int64_t n_hi, const float* W0,
                           float* W, double* Xt_out, double* Rt_out, float* dW_out, double* U_out, void* workspace,
                           int64_t workspace_bytes, int* info_dev, void* stream) {
    EMCID_CHECK_ARG(K && Zc && zs_t && C && N > 0 && d > 0 && h > 0 && layers_left > 0 && workspace && info_dev);
    EMCID_CHECK_ARG(N < (1 << 24) && d <= 32768 && h <= 32768);
    EMCID_CHECK_ARG(0 <= n_lo && n_lo < n_hi && n_hi <= N);
    EMCID_CHECK_ARG((W == nullptr) || (W0 != nullptr));
    EMCID_CHECK_ARG(aligned16(workspace));
    EditWorkspace ws(N, d, h);
    if (workspace_bytes < ws.total * (int64_t)sizeof(double))
        return fail(EMCID_ERR_WORKSPACE, __func__, "workspace too small (see emcid_edit_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    double* base = (double*)workspace;
    double *A = base + ws.off_A, *L = base + ws.off_L, *inv = base + ws.off_inv;
    double *B = base + ws.off_B, *Y = base + ws.off_Y, *R = base + ws.off_R;
    const double s = sqrt(edit_weight / 0.5);
    const float cw = (float)(1.0 - edit_weight);  // torch multiplies the fp32 tensor by the scalar rounded to fp32
    const int64_t rows = n_hi - n_lo;

    {
        ScopedProf sp(KC_PREP, st);
        hipLaunchKernelGGL(prep_kr_kernel, dim3((unsigned)ws.Np), dim3(256), 0, st, K, Zc, zs_t, (int)N, (int)d, (int)h, s,
                           (double)layers_left, B, (int)ws.Np, (int)ws.dp, R, (int)ws.hp);
    }
    EMCID_CHECK_LAUNCH();
    EMCID_TRY(emcid_assemble_spd_f64(C, d, B, ws.Np, d, ws.dp, lam, cw, A, ws.dp, stream));
    // only this shard's concept rows go through the triangular solves and the dW contraction
    double* Bs = B + n_lo * ws.dp;
    EMCID_TRY(factor_and_solve(A, L, ws.dp, ws.dp, inv, info_dev, Bs, Y + n_lo * ws.dp, rows, ws.dp, st));
    if (W || dW_out || U_out)
        EMCID_TRY(emcid_delta_w_f64(R + n_lo * ws.hp, ws.hp, Bs, ws.dp, rows, h, d, W0, W, d, dW_out, U_out, stream));
    if (Xt_out)
        hipLaunchKernelGGL(copy2d_f64_kernel, dim3((unsigned)rows), dim3(256), 0, st, Bs, ws.dp, Xt_out, d, (int)rows, (int)d);
    if (Rt_out)
        hipLaunchKernelGGL(copy2d_f64_kernel, dim3((unsigned)rows), dim3(256), 0, st, R + n_lo * ws.hp, ws.hp, Rt_out, h,
                           (int)rows, (int)h);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int emcid_edit_layer_f64(const float* K, const float* Zc, const float* zs_t, const float* C, int64_t N, int64_t d, int64_t h,
                         double lam, double edit_weight, int layers_left, const float* W0, float* W, double* Xt_out,
                         double* Rt_out, float* dW_out, void* workspace, int64_t workspace_bytes, int* info_dev,
                         void* stream) {
    return edit_layer_impl(K, Zc, zs_t, C, N, d, h, lam, edit_weight, layers_left, 0, N, W0, W, Xt_out, Rt_out, dW_out,
                           nullptr, workspace, workspace_bytes, info_dev, stream);
}

int emcid_edit_layer_shard_f64(const float* K, const float* Zc, const float* zs_t, const float* C, int64_t N, int64_t d,
                               int64_t h, double lam, double edit_weight, int layers_left, int64_t n_lo, int64_t n_hi,
                               double* U_partial, double* Xt_out, double* Rt_out, void* workspace, int64_t workspace_bytes,
                               int* info_dev, void* stream) {
    EMCID_CHECK_ARG(U_partial != nullptr);
    return edit_layer_impl(K, Zc, zs_t, C, N, d, h, lam, edit_weight, layers_left, n_lo, n_hi, nullptr, nullptr, Xt_out,
                           Rt_out, nullptr, U_partial, workspace, workspace_bytes, info_dev, stream);
}

int emcid_apply_update_f32(const double* U, const float* W0, float* W, float* dW, int64_t n, void* stream) {
    EMCID_CHECK_ARG(U && n > 0 && (W || dW) && ((W == nullptr) || (W0 != nullptr)));
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(apply_u_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, U, W0, W, dW, n);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

/* ---- dual (Woodbury) solver ------------------------------------------------------------------------------------- */

int64_t emcid_cov_factor_workspace_bytes(int64_t n_layers, int64_t d) {
    if (n_layers <= 0 || d <= 0) return 0;
    const int64_t dp = round_up(d, NB);
    return n_layers * (3 * dp * dp + inv_doubles(dp)) * (int64_t)sizeof(double);
}

int emcid_factor_cov_f64(const float* const* C_host_list, int64_t n_layers, int64_t d, double lam, double edit_weight,
                         void* workspace, int64_t workspace_bytes, int* info_dev, void* stream) {
    EMCID_CHECK_ARG(C_host_list && n_layers > 0 && n_layers <= 32 && d > 0 && d <= 32768 && workspace && info_dev);
    EMCID_CHECK_ARG(aligned16(workspace));
    if (workspace_bytes < emcid_cov_factor_workspace_bytes(n_layers, d))
        return fail(EMCID_ERR_WORKSPACE, __func__, "workspace too small (see emcid_cov_factor_workspace_bytes)");
    const int64_t dp = round_up(d, NB), s_mat = dp * dp, s_inv = inv_doubles(dp);
    double* Mb = (double*)workspace;                 // [n_layers][dp*dp]  lam*C' (consumed by the factorization)
    double* Lb = Mb + n_layers * s_mat;              // [n_layers][dp*dp]  factors
    double* Ib = Lb + n_layers * s_mat;              // [n_layers][inv_doubles]
    // [n_layers][dp*dp] after Ib: X = inv(L), explicit, built per layer by emcid_cov_inverse_f64
    CovPtrs cp;
    for (int i = 0; i < 32; ++i) cp.c[i] = i < n_layers ? C_host_list[i] : nullptr;
    for (int i = 0; i < n_layers; ++i) EMCID_CHECK_ARG(cp.c[i] != nullptr);
    const float cw = (float)(1.0 - edit_weight);
    hipStream_t st = (hipStream_t)stream;
    GraphKey key = make_key(2, {Mb, info_dev}, {n_layers, d, 0, 0, 0});
    memcpy(&key.num[2], &lam, sizeof(double));
    memcpy(&key.num[3], &cw, sizeof(float));
    for (int i = 0; i < n_layers; ++i)   // every C pointer takes part in the key (6 slots, then folded)
        if (i < 6) key.ptr[2 + i] = cp.c[i]; else key.num[4] = key.num[4] * 1000003 + (int64_t)(uintptr_t)cp.c[i];
    return with_graph(key, st, [&](hipStream_t s) {
        hipLaunchKernelGGL(scale_cov_kernel, dim3((unsigned)dp, (unsigned)n_layers), dim3(256), 0, s, cp, (int)d, (int)dp, lam, cw,
                           Mb, s_mat);
        return cholesky_serial(Mb, Lb, dp, dp, Ib, info_dev, s, (int)n_layers, s_mat, s_inv);
    });
}

/* X_l = inv(L_l) for ONE layer of a factored workspace (needs emcid_factor_cov_f64 earlier on the same stream, or an
 * event dependency on it).  Per layer so that the first edited layer's solve can start while the later layers' inverse
 * factors are still being built underneath it. */
int emcid_cov_inverse_f64(void* cov_factor_ws, int64_t n_layers, int64_t d, int64_t first_layer, int64_t count, void* stream) {
    EMCID_CHECK_ARG(cov_factor_ws && n_layers > 0 && n_layers <= 32 && d > 0 && d <= 32768 && aligned16(cov_factor_ws));
    EMCID_CHECK_ARG(0 <= first_layer && count > 0 && first_layer + count <= n_layers);
    const int64_t dp = round_up(d, NB), s_mat = dp * dp, s_inv = inv_doubles(dp);
    double* Mb = (double*)cov_factor_ws + first_layer * s_mat;          // consumed by the factorization: scratch now
    const double* Lb = (const double*)cov_factor_ws + (n_layers + first_layer) * s_mat;
    const double* Ib = (const double*)cov_factor_ws + 2 * n_layers * s_mat + first_layer * s_inv;
    double* Xb = (double*)cov_factor_ws + n_layers * (2 * s_mat + s_inv) + first_layer * s_mat;
    return with_graph(make_key(7, {Mb, Lb, Ib, Xb}, {dp, count}), (hipStream_t)stream, [&](hipStream_t s) {
        return build_full_inverse(Lb, dp, dp, Ib, Xb, Mb, (int)count, s_mat, s_inv, s);   // batched over the range
    });
}

int64_t emcid_edit_dual_workspace_bytes(int64_t N, int64_t d, int64_t h) {
    if (N <= 0 || d <= 0 || h <= 0) return 0;
    return DualWorkspace(N, d, h).total * (int64_t)sizeof(double);
}

/* stage 1: Kt64 = s*K, Rt, and the shard's rows of Pt = Kt64 M^-1 (into Pt_rows_out if given, else only the workspace) */
int emcid_edit_dual_stage1_f64(const float* K, const float* Zc, const float* zs_t, int64_t N, int64_t d, int64_t h,
                               double edit_weight, int layers_left, double lam_ratio, const void* cov_factor_ws, int64_t n_layers,
                               int64_t layer_index, int64_t n_lo, int64_t n_hi, int use_inverse, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    EMCID_CHECK_ARG(K && Zc && zs_t && N > 0 && d > 0 && h > 0 && layers_left > 0 && cov_factor_ws && workspace);
    EMCID_CHECK_ARG(0 <= layer_index && layer_index < n_layers && 0 <= n_lo && n_lo < n_hi && n_hi <= N);
    EMCID_CHECK_ARG(lam_ratio > 0.0 && lam_ratio < 1e300);
    DualWorkspace ws(N, d, h);
    if (workspace_bytes < ws.total * (int64_t)sizeof(double)) return fail(EMCID_ERR_WORKSPACE, __func__, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* base = (double*)workspace;
    double *Kt = base + ws.off_K, *Pt = base + ws.off_P, *Y = base + ws.off_Y, *R = base + ws.off_R;
    const int64_t dp = ws.dp, s_mat = dp * dp;
    const double* Lb = (const double*)cov_factor_ws + n_layers * s_mat + layer_index * s_mat;
    const double* Ib = (const double*)cov_factor_ws + 2 * n_layers * s_mat + layer_index * inv_doubles(dp);
    const double s = sqrt(edit_weight / 0.5);
    {
        ScopedProf sp(KC_PREP, st);
        hipLaunchKernelGGL(prep_kr_kernel, dim3((unsigned)ws.Np), dim3(256), 0, st, K, Zc, zs_t, (int)N, (int)d, (int)h, s,
                           (double)layers_left, Kt, (int)ws.Np, (int)dp, R, (int)ws.hp, 1.0 / sqrt(lam_ratio));
    }
    const int64_t rows = n_hi - n_lo;
    if (use_inverse) {
        // Pt = (Kt X^T) X : both triangular solves against M = L L^T are GEMMs against the explicit X = inv(L)
        const double* Xb = cov_inverse(cov_factor_ws, n_layers, dp, layer_index);
        apply_inverse_forward(Xb, dp, Kt + n_lo * dp, Y + n_lo * dp, (int)rows, st, base + ws.off_SK);
        apply_inverse_backward(Xb, dp, Y + n_lo * dp, (int)rows, (int)dp, Pt + n_lo * dp, dp, st, base + ws.off_SK);
        EMCID_CHECK_LAUNCH();
        return EMCID_OK;
    }
    if (hipMemcpyAsync(Pt + n_lo * dp, Kt + n_lo * dp, rows * dp * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(EMCID_ERR_HIP, __func__, "hipMemcpyAsync");
    return with_graph(make_key(3, {Lb, Ib, Pt + n_lo * dp, Y + n_lo * dp}, {dp, rows}), st, [&](hipStream_t q) {
        return cholesky_solve_impl(Lb, dp, dp, Ib, Pt + n_lo * dp, Y + n_lo * dp, rows, dp, q);
    });
}

/* pointer to the Pt stack [Np, dp] inside a dual workspace (multi-GPU: ranks all-gather their row blocks in place) */
double* emcid_edit_dual_pt(void* workspace, int64_t N, int64_t d, int64_t h) {
    if (!workspace || N <= 0 || d <= 0 || h <= 0) return nullptr;
    return (double*)workspace + DualWorkspace(N, d, h).off_P;
}

/* stage 2 (needs ALL rows of Pt): S = I + Pt Kt^T, S = L_S L_S^T, adj_k = (S^-1 Pt)^T  [d, Np], U = Rt^T Xt, W = W0 + float(U) */
int emcid_edit_dual_stage2_f64(int64_t N, int64_t d, int64_t h, double lam_ratio, const float* W0, float* W, double* adjk_out, double* Rt_out,
                               float* dW_out, void* workspace, int64_t workspace_bytes, int* info_dev, void* stream) {
    EMCID_CHECK_ARG(N > 0 && d > 0 && h > 0 && workspace && info_dev && ((W == nullptr) || (W0 != nullptr)));
    EMCID_CHECK_ARG(lam_ratio > 0.0 && lam_ratio < 1e300);
    DualWorkspace ws(N, d, h);
    if (workspace_bytes < ws.total * (int64_t)sizeof(double)) return fail(EMCID_ERR_WORKSPACE, __func__, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* base = (double*)workspace;
    double *Kt = base + ws.off_K, *Pt = base + ws.off_P, *R = base + ws.off_R, *S = base + ws.off_S, *LS = base + ws.off_LS;
    double *invS = base + ws.off_invS, *PT = base + ws.off_PT, *Y2 = base + ws.off_Y2;
    const int64_t dp = ws.dp, Np = ws.Np;
    EMCID_TRY(with_graph(make_key(4, {Kt, Pt, S, LS, invS, PT, Y2, info_dev}, {dp, Np, N}), st, [&](hipStream_t q) {
        if (Np > N)   // rows of the padding concepts: zero (their Kt rows are zero, so S gets identity rows there)
            hipLaunchKernelGGL(zero_f64_kernel, dim3(256), dim3(256), 0, q, Pt + N * dp, (Np - N) * dp);
        assemble_dual_system(Pt, Kt, dp, S, (int)Np, q, base + ws.off_SK);
        EMCID_TRY(cholesky_impl(S, LS, Np, Np, invS, info_dev, q));
        hipLaunchKernelGGL(transpose_f64_kernel, dim3((unsigned)(dp / 32), (unsigned)(Np / 32)), dim3(256), 0, q, Pt, dp, PT, Np,
                           (int)Np, (int)dp);
        return cholesky_solve_impl(LS, Np, Np, invS, PT, Y2, dp, Np, q);   // PT := PT S^-1  ->  adj_k padded [dp, Np]
    }));
    if (W || dW_out) {
        ScopedProf sp(KC_DELTA_W, st);
        GemmShape g{R, ws.hp, PT, Np, (int)h, (int)d, (int)Np, 0};
        launch_gemm_f64<false, true>(g, EpiDeltaW{W0, W, d, dW_out, d, nullptr, d}, st);
    }
    // the workspace holds sqrt(lam_ratio) * adj_k and Rt / sqrt(lam_ratio) (stage 1's gain): the caller gets both in its own scale
    const double root = sqrt(lam_ratio);
    if (adjk_out)
        hipLaunchKernelGGL(copy2d_f64_kernel, dim3((unsigned)d), dim3(256), 0, st, PT, Np, adjk_out, N, (int)d, (int)N, 1.0 / root);
    if (Rt_out) hipLaunchKernelGGL(copy2d_f64_kernel, dim3((unsigned)N), dim3(256), 0, st, R, ws.hp, Rt_out, h, (int)N, (int)h, root);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

/* ---- dual solver, apply-only form: adj_k is never formed -----------------------------------------------------------
 * With M = L L^T:  Yt = Kt64 L^-T,  S = I + Yt Yt^T,  Z = S^-1 Rt,  U = Rt^T Xt = (Z^T Yt) L^-1,  W = W0 + float(U).
 * One forward solve on the N concept rows, a true SYRK, the N x N Cholesky, two solves with only h right-hand sides,
 * one GEMM and one backward solve on h rows.  Same algebra as stage1 + stage2 by associativity. */
int emcid_edit_dual_apply_stage1_f64(const float* K, const float* Zc, const float* zs_t, int64_t N, int64_t d, int64_t h,
                                     double edit_weight, int layers_left, double lam_ratio, const void* cov_factor_ws, int64_t n_layers,
                                     int64_t layer_index, int64_t n_lo, int64_t n_hi, int use_inverse, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    EMCID_CHECK_ARG(K && Zc && zs_t && N > 0 && d > 0 && h > 0 && layers_left > 0 && cov_factor_ws && workspace);
    EMCID_CHECK_ARG(0 <= layer_index && layer_index < n_layers && 0 <= n_lo && n_lo < n_hi && n_hi <= N);
    EMCID_CHECK_ARG(lam_ratio > 0.0 && lam_ratio < 1e300);
    DualWorkspace ws(N, d, h);
    if (workspace_bytes < ws.total * (int64_t)sizeof(double)) return fail(EMCID_ERR_WORKSPACE, __func__, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* base = (double*)workspace;
    double *Kt = base + ws.off_K, *Bs = base + ws.off_P, *Yt = base + ws.off_Y, *R = base + ws.off_R;
    const int64_t dp = ws.dp, s_mat = dp * dp;
    const double* Lb = (const double*)cov_factor_ws + n_layers * s_mat + layer_index * s_mat;
    const double* Ib = (const double*)cov_factor_ws + 2 * n_layers * s_mat + layer_index * inv_doubles(dp);
    const double s = sqrt(edit_weight / 0.5);
    {
        ScopedProf sp(KC_PREP, st);
        hipLaunchKernelGGL(prep_kr_kernel, dim3((unsigned)ws.Np), dim3(256), 0, st, K, Zc, zs_t, (int)N, (int)d, (int)h, s,
                           (double)layers_left, Kt, (int)ws.Np, (int)dp, R, (int)ws.hp, 1.0 / sqrt(lam_ratio));
    }
    const int64_t rows = n_hi - n_lo;
    if (use_inverse) {
        // the whole concept range: run over the Np padded rows (Kt's padding rows are zero, so are the products) — every
        // 128-row tile then lies inside the operand and takes the interior fast path of the stream-K K loop
        const int64_t gemm_rows = (n_lo == 0 && n_hi == N) ? ws.Np : rows;
        apply_inverse_forward(cov_inverse(cov_factor_ws, n_layers, dp, layer_index), dp, Kt + n_lo * dp, Yt + n_lo * dp, (int)gemm_rows,
                              st, base + ws.off_SK);
        // stage 1 leaves the padding rows [N, Np) of Yt zero (the later stages rely on it): the full-range GEMM has just produced
        // them; a partial range (row-sharded callers) zeroes them here
        if (gemm_rows != ws.Np && ws.Np > N)
            hipLaunchKernelGGL(zero_f64_kernel, dim3(256), dim3(256), 0, st, Yt + N * dp, (ws.Np - N) * dp);
        EMCID_CHECK_LAUNCH();
        return EMCID_OK;
    }
    if (ws.Np > N) hipLaunchKernelGGL(zero_f64_kernel, dim3(256), dim3(256), 0, st, Yt + N * dp, (ws.Np - N) * dp);
    if (hipMemcpyAsync(Bs + n_lo * dp, Kt + n_lo * dp, rows * dp * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(EMCID_ERR_HIP, __func__, "hipMemcpyAsync");
    return with_graph(make_key(5, {Lb, Ib, Bs + n_lo * dp, Yt + n_lo * dp}, {dp, rows}), st, [&](hipStream_t q) {
        trsm_forward(Lb, dp, dp, Ib, Bs + n_lo * dp, Yt + n_lo * dp, (int)rows, dp, q);
        return check_launch("emcid_edit_dual_apply_stage1_f64");
    });
}

/* address of the Yt stack [Np, dp] inside a dual workspace (multi-GPU: ranks all-gather their row blocks there) */
double* emcid_edit_dual_yt(void* workspace, int64_t N, int64_t d, int64_t h) {
    if (!workspace || N <= 0 || d <= 0 || h <= 0) return nullptr;
    return (double*)workspace + DualWorkspace(N, d, h).off_Y;
}

/* The first part of stage 2 on its own: S = I + Yt Yt^T.  A caller that wants to start other work exactly when the
 * latency-bound Cholesky of S begins (the engine builds the next layer's inverse factor on a second stream then) calls
 * this, records its event, and passes assembled = 1 to stage 2. */
int emcid_edit_dual_apply_assemble_f64(int64_t N, int64_t d, int64_t h, void* workspace, int64_t workspace_bytes, void* stream) {
    EMCID_CHECK_ARG(N > 0 && d > 0 && h > 0 && workspace);
    DualWorkspace ws(N, d, h);
    if (workspace_bytes < ws.total * (int64_t)sizeof(double)) return fail(EMCID_ERR_WORKSPACE, __func__, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* base = (double*)workspace;
    double *Yt = base + ws.off_Y, *S = base + ws.off_S;
    const int64_t dp = ws.dp, Np = ws.Np;
    assemble_dual_system(Yt, Yt, dp, S, (int)Np, st, base + ws.off_SK);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

int emcid_edit_dual_apply_stage2_f64(int64_t N, int64_t d, int64_t h, const void* cov_factor_ws, int64_t n_layers,
                                     int64_t layer_index, int use_inverse, int assembled, const float* W0, float* W,
                                     float* dW_out, void* workspace, int64_t workspace_bytes, int* info_dev, void* stream) {
    EMCID_CHECK_ARG(N > 0 && d > 0 && h > 0 && workspace && info_dev && cov_factor_ws && ((W == nullptr) || (W0 != nullptr)));
    EMCID_CHECK_ARG(0 <= layer_index && layer_index < n_layers && (W || dW_out));
    DualWorkspace ws(N, d, h);
    if (workspace_bytes < ws.total * (int64_t)sizeof(double)) return fail(EMCID_ERR_WORKSPACE, __func__, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* base = (double*)workspace;
    double *Yt = base + ws.off_Y, *R = base + ws.off_R, *S = base + ws.off_S, *LS = base + ws.off_LS, *invS = base + ws.off_invS;
    double *RT = base + ws.off_PT, *Y2 = base + ws.off_Y2, *V = base + ws.off_V, *U = base + ws.off_U;
    const int64_t dp = ws.dp, Np = ws.Np, hp = ws.hp, s_mat = dp * dp;
    const double* Lb = (const double*)cov_factor_ws + n_layers * s_mat + layer_index * s_mat;
    const double* Ib = (const double*)cov_factor_ws + 2 * n_layers * s_mat + layer_index * inv_doubles(dp);
    // P = Yt X rides in the Cholesky's leaf launches (ShadowJob) when X is explicit and the fused schedule runs: then
    // U = Z^T P is one GEMM and the GEMM against the triangle after the N x N solve (U = (Z^T Yt) X) disappears from the chain
    static const int shadow_env = env_flag("EMCID_SHADOW_P", 1);      // 0: never, 1: when it fits under the leaves, 2: always
    bool shadow = shadow_env && use_inverse && cholesky_takes_shadow(Np);
    if (shadow && shadow_env == 1) {
        // The product only pays while a launch's shadow tiles finish about when its leaf does (~36 us).  Measured on MI355X
        // (scripts/step_stamps.py): a tile pair's slice costs ~9 us + 1.5 us per 16-deep K step, one workgroup per compute unit.
        // SD dims, N = 1000: 25 steps, 192 workgroups -> 46 us; SDXL TE2 (d = 5120) at N = 1000: 41 steps in two rounds -> no.
        const int64_t ntl = (dp + SH_BN - 1) / SH_BN, nb = Np / NB;
        const int64_t steps = ((ntl + 1) * (SH_BN / 16) + nb - 1) / nb;
        const int64_t wgs = ((Np + SH_BM - 1) / SH_BM) * ((ntl + 1) / 2), rounds = (wgs + 239) / 240;
        shadow = rounds * (9.0 + 1.5 * (double)steps) <= 50.0;
    }
    double* P = base + ws.off_P;
    const double* X = use_inverse ? cov_inverse(cov_factor_ws, n_layers, dp, layer_index) : nullptr;
    // Few concepts (a 100-concept edit: Np = 128, no chain of leaves to ride in): U = Z^T (Yt X) with the triangle multiplied on the
    // Np-row side as a launch of its own, instead of U = (Z^T Yt) X on the h-row side — 128 rows against 768 at SD dims
    // (61 + ~15 us instead of ~20 + 155 per layer).  EMCID_P_FIRST=0: the h-side form.
    const bool p_first = !shadow && use_inverse && Np < h;
    const bool xrow = cholesky_takes_shadow(Np);       // (= the fused leaf / spine schedule runs)
    double *XT = base + ws.off_XT, *TT = base + ws.off_TT;
    EMCID_TRY(with_graph(make_key(6, {Yt, R, S, LS, RT, V, U, info_dev},
                                  {dp, Np, N, hp, (int64_t)(uintptr_t)Lb,
                                   use_inverse + 2 * (assembled != 0) + 4 * (int)shadow + 8 * h + ((int64_t)p_first << 31) + ((int64_t)xrow << 30) + (d << 32)}),
                         st,
                         [&](hipStream_t q) {
        if (!assembled) {
            assemble_dual_system(Yt, Yt, dp, S, (int)Np, q, base + ws.off_SK);      // S = I + Yt Yt^T (lower tiles)
        }
        if (p_first) apply_inverse_backward(X, dp, Yt, (int)Np, (int)dp, P, dp, q, base + ws.off_SK);       // P = Yt X
        ShadowJob job{Yt, dp, X, dp, P, dp, (int)Np, (int)dp, (int)dp, 0, nullptr, 0, 0, 0};
        job.wgs = (int)((Np + SH_BM - 1) / SH_BM) * (int)(((dp + SH_BN - 1) / SH_BN + 1) / 2);
        // XS = inv(LS) rides in the factorization's launches (XrowJob), transposed
        const XrowJob xj{XT, Np, TT};
        EMCID_TRY(cholesky_impl(S, LS, Np, Np, invS, info_dev, q, shadow ? &job : nullptr, xrow ? &xj : nullptr));
        // RT[h, Np] = Rt^T ; Z^T = RT S^-1 (two solves with h rows)
        hipLaunchKernelGGL(transpose_f64_kernel, dim3((unsigned)((hp + 31) / 32), (unsigned)(Np / 32)), dim3(256), 0, q, R, hp, RT,
                           Np, (int)Np, (int)hp);
        // Z^T = RT S^-1.  As block substitution this is 6 dependent launches on h rows (~140 us at N = 1000, latency-bound); with
        // XS = inv(LS) made explicit (one more halving level on top of the 512-block inverses, into S, which the factorization has
        // consumed) it is two GEMMs against a triangle:  Z^T = (RT XS^T) XS (the substitution stays for N the fused schedule does not take).
        if (xrow) {
            ScopedProf sp(KC_TRSM_DIAG, q);
            GemmShape f{RT, Np, XT, Np, (int)h, (int)Np, (int)Np, 0};
            f.tri = 1;       // B(k, n) = XS[n][k] = Xt[k][n], zero for k > n
            f.pair = 1;
            launch_gemm_f64<true, false>(f, EpiAxpby{Y2, Np, 1.0, 0.0}, q);
            GemmShape b{Y2, Np, XT, Np, (int)h, (int)Np, (int)Np, 0};
            b.tri = 2;       // B(k, n) = XS[k][n] = Xt[n][k], zero for k < n
            b.pair = 1;
            launch_gemm_f64<true, true>(b, EpiAxpby{RT, Np, 1.0, 0.0}, q);
        } else if (Np <= 4096) {
            EMCID_TRY(build_full_inverse(LS, Np, Np, invS, S, Y2, 1, 0, 0, q));
            ScopedProf sp(KC_TRSM_DIAG, q);
            GemmShape f{RT, Np, S, Np, (int)h, (int)Np, (int)Np, 0};
            f.tri = 1;       // B(k, n) = XS[n][k], zero for k > n
            f.pair = 1;
            launch_gemm_f64<true, true>(f, EpiAxpby{Y2, Np, 1.0, 0.0}, q);
            GemmShape b{Y2, Np, S, Np, (int)h, (int)Np, (int)Np, 0};
            b.tri = 2;       // B(k, n) = XS[k][n], zero for k < n
            b.pair = 1;
            launch_gemm_f64<true, false>(b, EpiAxpby{RT, Np, 1.0, 0.0}, q);
        } else {
            EMCID_TRY(cholesky_solve_impl(LS, Np, Np, invS, RT, Y2, h, Np, q));
        }
        if (!shadow && !p_first) {
            ScopedProf sp(KC_DELTA_W, q);       // V[h, dp] = Z^T Yt
            GemmShape g{RT, Np, Yt, dp, (int)h, (int)dp, (int)Np, 0};
            launch_gemm_f64<true, false>(g, EpiAxpby{V, dp, 1.0, 0.0}, q);
        }
        if (!use_inverse) trsm_backward(Lb, dp, dp, Ib, V, U, (int)h, dp, q);   // U L = V by block substitution
        return check_launch("emcid_edit_dual_apply_stage2_f64");
    }));
    if (shadow || p_first) {
        // U = Z^T P straight into the weights: W = W0 + float(U), dW = float(U) in the GEMM's epilogue (outside the cached graph:
        // W0 / W / dW are the caller's tensors and change from layer to layer and call to call)
        ScopedProf sp(KC_DELTA_W, st);
        GemmShape g{RT, Np, P, dp, (int)h, (int)d, (int)Np, 0};
        launch_gemm_f64<true, false>(g, EpiDeltaW{W0, W, d, dW_out, d, nullptr, 0}, st, -1);      // the launcher's choice: 32 x 64 tiles
        EMCID_CHECK_LAUNCH();
        return EMCID_OK;
    }
    if (use_inverse)   // U = V inv(L)  (V's padding columns are zero: Kt's are, X is the identity there)
        apply_inverse_backward(X, dp, V, (int)h, (int)dp, U, dp, st, base + ws.off_SK);
    hipLaunchKernelGGL(apply_u2d_kernel, dim3((unsigned)h), dim3(256), 0, st, U, dp, W0, W, dW_out, (int)d);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

/* ---- dual solver, apply-only form, COLUMN-SHARDED over ranks (multi-GPU) -----------------------------------
 * The d columns of Yt = Kt64 X^T are dealt to the ranks in 128-wide tiles (`tiles`: this rank's tile indices, ascending).
 * With Yc = the rank's columns of Yt:
 *     S = I + sum_ranks Yc Yc^T        (one all-reduce of the N x N partial sums — the only coupling of the concepts)
 *     V[:, mine] = Z^T Yc,  Z = S^-1 Rt (every rank factors S itself: d^3-free, latency-bound, 0.5 ms)
 *     U = V X = sum_ranks V[:, mine] X[mine, :]      (one all-reduce of the h x d partial sums)
 * so a rank's GEMM work is 1/world of the layer's and nothing but S and U crosses the links.  Needs X = inv(L) of the layer
 * (emcid_cov_inverse_f64).  stage 1 leaves the partial S (no identity) at emcid_edit_dual_s(); stage 2 expects the SUMMED S
 * there and leaves the partial U (leading dimension dp = d rounded up to 128) at emcid_edit_dual_u(). */
__global__ __launch_bounds__(256) void add_identity_f64_kernel(double* __restrict__ S, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) S[(int64_t)i * n + i] += 1.0;
}

static int check_tiles(const int* tiles, int n_tiles, int64_t dp) {
    if (!tiles || n_tiles <= 0 || n_tiles > 256) return 0;
    for (int i = 0; i < n_tiles; ++i)
        if (tiles[i] < 0 || (int64_t)tiles[i] * NB >= dp || (i > 0 && tiles[i] <= tiles[i - 1])) return 0;
    return 1;
}

int emcid_edit_dual_cols_stage1_f64(const float* K, const float* Zc, const float* zs_t, int64_t N, int64_t d, int64_t h,
                                    double edit_weight, int layers_left, double lam_ratio, const void* cov_factor_ws, int64_t n_layers,
                                    int64_t layer_index, const int* tiles_host, int n_tiles, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
    EMCID_CHECK_ARG(K && Zc && zs_t && N > 0 && d > 0 && h > 0 && layers_left > 0 && cov_factor_ws && workspace);
    EMCID_CHECK_ARG(0 <= layer_index && layer_index < n_layers && lam_ratio > 0.0 && lam_ratio < 1e300);
    DualWorkspace ws(N, d, h);
    EMCID_CHECK_ARG(check_tiles(tiles_host, n_tiles, ws.dp));
    if (workspace_bytes < ws.total * (int64_t)sizeof(double)) return fail(EMCID_ERR_WORKSPACE, __func__, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* base = (double*)workspace;
    double *Kt = base + ws.off_K, *Yc = base + ws.off_Y, *R = base + ws.off_R, *S = base + ws.off_S, *sk = base + ws.off_SK;
    const int64_t dp = ws.dp, Np = ws.Np;
    const double* X = cov_inverse(cov_factor_ws, n_layers, dp, layer_index);
    const double s = sqrt(edit_weight / 0.5);
    {
        ScopedProf sp(KC_PREP, st);
        hipLaunchKernelGGL(prep_kr_kernel, dim3((unsigned)Np), dim3(256), 0, st, K, Zc, zs_t, (int)N, (int)d, (int)h, s,
                           (double)layers_left, Kt, (int)Np, (int)dp, R, (int)ws.hp, 1.0 / sqrt(lam_ratio));
    }
    for (int i = 0; i < n_tiles; ++i) {        // Yc[:, 128 i : 128 i + 128] = Kt[:, 0 : kd] X[128 t : 128 t + 128, 0 : kd]^T,  kd = 128 (t + 1)
        const int64_t t = tiles_host[i], kd = (t + 1) * NB;
        ScopedProf sp(KC_INV_APPLY, st);
        GemmShape g{Kt, dp, X + t * NB * dp, dp, (int)Np, NB, (int)kd, 0};
        if (!launch_gemm_f64_streamk2<true, true>(g, EpiAxpby{Yc + (int64_t)i * NB, dp, 1.0, 0.0}, st, kStreamKWgs, sk))
            return fail(EMCID_ERR_BAD_ARG, __func__, "more output tiles than stream-K ticket counters");
    }
    {   // partial S = Yc Yc^T on the lower 128-tiles, K = 128 n_tiles deep
        ScopedProf sp(KC_ASSEMBLE, st);
        GemmShape g{Yc, dp, Yc, dp, (int)Np, (int)Np, n_tiles * NB, 1};
        if (!launch_gemm_f64_streamk2<true, true>(g, EpiAxpby{S, Np, 1.0, 0.0}, st, kStreamKWgs, sk, 0.0))
            return fail(EMCID_ERR_BAD_ARG, __func__, "more output tiles than stream-K ticket counters");
    }
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

double* emcid_edit_dual_s(void* workspace, int64_t N, int64_t d, int64_t h) {
    if (!workspace || N <= 0 || d <= 0 || h <= 0) return nullptr;
    return (double*)workspace + DualWorkspace(N, d, h).off_S;
}

double* emcid_edit_dual_u(void* workspace, int64_t N, int64_t d, int64_t h) {
    if (!workspace || N <= 0 || d <= 0 || h <= 0) return nullptr;
    return (double*)workspace + DualWorkspace(N, d, h).off_U;
}

int emcid_edit_dual_cols_stage2_f64(int64_t N, int64_t d, int64_t h, const void* cov_factor_ws, int64_t n_layers,
                                    int64_t layer_index, const int* tiles_host, int n_tiles, void* workspace,
                                    int64_t workspace_bytes, int* info_dev, void* stream) {
    EMCID_CHECK_ARG(N > 0 && d > 0 && h > 0 && workspace && info_dev && cov_factor_ws);
    EMCID_CHECK_ARG(0 <= layer_index && layer_index < n_layers);
    DualWorkspace ws(N, d, h);
    EMCID_CHECK_ARG(check_tiles(tiles_host, n_tiles, ws.dp));
    if (workspace_bytes < ws.total * (int64_t)sizeof(double)) return fail(EMCID_ERR_WORKSPACE, __func__, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* base = (double*)workspace;
    double *Yc = base + ws.off_Y, *R = base + ws.off_R, *S = base + ws.off_S, *LS = base + ws.off_LS, *invS = base + ws.off_invS;
    double *RT = base + ws.off_PT, *Y2 = base + ws.off_Y2, *V = base + ws.off_V, *U = base + ws.off_U;
    const int64_t dp = ws.dp, Np = ws.Np, hp = ws.hp;
    const double* X = cov_inverse(cov_factor_ws, n_layers, dp, layer_index);
    const int w = n_tiles * NB;
    const bool xrow = cholesky_takes_shadow(Np);
    EMCID_TRY(with_graph(make_key(8, {Yc, R, S, LS, RT, V, U, info_dev}, {dp, Np, N, hp, (int64_t)w + ((int64_t)xrow << 40)}), st, [&](hipStream_t q) {
        hipLaunchKernelGGL(add_identity_f64_kernel, dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, q, S, (int)Np);
        const XrowJob xj{base + ws.off_XT, Np, base + ws.off_TT};
        EMCID_TRY(cholesky_impl(S, LS, Np, Np, invS, info_dev, q, nullptr, xrow ? &xj : nullptr));
        hipLaunchKernelGGL(transpose_f64_kernel, dim3((unsigned)((hp + 31) / 32), (unsigned)(Np / 32)), dim3(256), 0, q, R, hp, RT,
                           Np, (int)Np, (int)hp);
        if (xrow) {     // Z^T = (RT XS^T) XS against the inverse that rode in the factorization (as emcid_edit_dual_apply_stage2_f64)
            ScopedProf sp(KC_TRSM_DIAG, q);
            GemmShape f{RT, Np, xj.Xt, Np, (int)h, (int)Np, (int)Np, 0};
            f.tri = 1;
            f.pair = 1;
            launch_gemm_f64<true, false>(f, EpiAxpby{Y2, Np, 1.0, 0.0}, q);
            GemmShape b{Y2, Np, xj.Xt, Np, (int)h, (int)Np, (int)Np, 0};
            b.tri = 2;
            b.pair = 1;
            launch_gemm_f64<true, true>(b, EpiAxpby{RT, Np, 1.0, 0.0}, q);
        } else {
            EMCID_TRY(cholesky_solve_impl(LS, Np, Np, invS, RT, Y2, h, Np, q));      // RT := Z^T
        }
        {
            ScopedProf sp(KC_DELTA_W, q);       // V[h, w] = Z^T Yc
            GemmShape g{RT, Np, Yc, dp, (int)h, w, (int)Np, 0};
            launch_gemm_f64<true, false>(g, EpiAxpby{V, dp, 1.0, 0.0}, q);
        }
        hipLaunchKernelGGL(zero2d_f64_kernel, dim3((unsigned)h, 1u), dim3(256), 0, q, U, dp, (int64_t)0, (int)dp);
        return check_launch("emcid_edit_dual_cols_stage2_f64");
    }));
    for (int i = 0; i < n_tiles; ++i) {        // U[:, 0 : kd] += V[:, 128 i : 128 i + 128] X[128 t : 128 t + 128, 0 : kd]
        const int64_t t = tiles_host[i], kd = (t + 1) * NB;
        ScopedProf sp(KC_INV_APPLY, st);
        GemmShape g{V + (int64_t)i * NB, dp, X + t * NB * dp, dp, (int)h, (int)kd, NB, 0};
        launch_gemm_f64<true, false>(g, EpiAxpby{U, dp, 1.0, 1.0}, st);
    }
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

/* W = W0 + float(U) (optional), dW = float(U) (optional) for U [h][ldu] f64 with ldu >= d (the padded partial sums above) */
int emcid_apply_update2d_f32(const double* U, int64_t ldu, const float* W0, float* W, float* dW, int64_t h, int64_t d, void* stream) {
    EMCID_CHECK_ARG(U && h > 0 && d > 0 && ldu >= d && (W || dW) && ((W == nullptr) || (W0 != nullptr)));
    hipLaunchKernelGGL(apply_u2d_kernel, dim3((unsigned)h), dim3(256), 0, (hipStream_t)stream, U, ldu, W0, W, dW, (int)d);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

}  // extern "C"
