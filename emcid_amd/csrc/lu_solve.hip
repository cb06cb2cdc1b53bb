// Pivoted LU fallback of the Stage-2 layer solve on gfx950.
//
// The reference solves (lam*C' + K K^T) X = K with torch.linalg.solve (LAPACK getrf + getrs, partial pivoting;
// emcid/emcid_main.py:1045-1048): it returns numbers for ANY nonsingular system.  The fast paths here factor with
// Cholesky and report a non-positive pivot instead; when that happens the host reruns the layer through this file,
// which reproduces the reference's algorithm: right-looking blocked LU with row pivoting (largest magnitude in the
// column, lowest row on ties), unit-lower L, then the two triangular substitutions on all N right-hand sides.
//
// Built for correctness and for being rare, not for speed: the panel is factored by ONE workgroup straight out of
// global memory (every column step is a pivot search + a rank-1 update of the 32-column panel), the trailing updates
// and the substitution updates are the library's fp64 MFMA GEMM (gemm_f64.h).  ~50 ms for d = 3072.
#include "common.h"
#include "gemm_f64.h"

namespace emcid {

constexpr int LU_NB = 32;       // panel width
constexpr int LU_T = 1024;      // threads of the panel workgroup

// One panel: columns [j0, j0 + nb) of the n x n matrix A (row-major, lda), rows j0..n-1.  piv[j0 + c] = row swapped
// into position j0 + c (swaps are applied to the panel columns only; lu_swap_rows_kernel does the rest of the rows).
__global__ __launch_bounds__(LU_T) void lu_panel_kernel(double* __restrict__ A, int64_t lda, int n, int j0, int nb,
                                                         int* __restrict__ piv, int* __restrict__ info) {
    __shared__ double s_prow[LU_NB];
    __shared__ double s_val[LU_T / 64];
    __shared__ int s_idx[LU_T / 64];
    __shared__ int s_piv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double best = -1.0;
    int best_r = n;
    for (int r = j0 + tid; r < n; r += LU_T) {          // pivot candidates of the first column
        const double v = fabs(A[(int64_t)r * lda + j0]);
        if (v > best || (v == best && r < best_r) || v != v) { best = v != v ? __builtin_inf() : v; best_r = r; }
    }
    for (int c = 0; c < nb; ++c) {
        const int col = j0 + c;
        // ---- block arg-max (value, then lowest row) -------------------------------------------------------------
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_down(best, off);
            const int orow = __shfl_down(best_r, off);
            if (ov > best || (ov == best && orow < best_r)) { best = ov; best_r = orow; }
        }
        if (lane == 0) { s_val[wave] = best; s_idx[wave] = best_r; }
        __syncthreads();
        if (tid == 0) {
            double b = s_val[0];
            int br = s_idx[0];
            for (int w = 1; w < LU_T / 64; ++w)
                if (s_val[w] > b || (s_val[w] == b && s_idx[w] < br)) { b = s_val[w]; br = s_idx[w]; }
            if (br >= n) br = col;                       // no candidate (cannot happen for col < n)
            s_piv = br;
            piv[col] = br;
        }
        __syncthreads();
        const int pr = s_piv;
        if (pr != col && tid < nb) {                     // swap the panel segments of rows col and pr
            double* a = A + (int64_t)col * lda + j0 + tid;
            double* b = A + (int64_t)pr * lda + j0 + tid;
            const double t = *a; *a = *b; *b = t;
        }
        __syncthreads();
        if (tid < nb) s_prow[tid] = A[(int64_t)col * lda + j0 + tid];
        __syncthreads();
        const double p = s_prow[c];
        if (p == 0.0 || p != p) {                        // singular to working precision (torch.linalg.solve raises here)
            if (tid == 0) atomicCAS(info, 0, col + 1);
        }
        const double rinv = 1.0 / p;                     // LAPACK dgetf2 scales by the reciprocal
        best = -1.0;
        best_r = n;
        for (int r = col + 1 + tid; r < n; r += LU_T) {
            double* row = A + (int64_t)r * lda + j0;
            const double l = row[c] * rinv;
            row[c] = l;
            for (int t = c + 1; t < nb; ++t) row[t] = fma(-l, s_prow[t], row[t]);
            if (c + 1 < nb) {
                const double v = fabs(row[c + 1]);
                if (v > best || (v == best && r < best_r) || v != v) { best = v != v ? __builtin_inf() : v; best_r = r; }
            }
        }
        __syncthreads();
    }
}

// Apply the panel's row swaps, in order, to columns [c_lo, c_hi) of a row-major matrix (one thread per column).
__global__ __launch_bounds__(256) void lu_swap_rows_kernel(double* __restrict__ M, int64_t ldm, int c_lo, int c_hi, int j0, int nb,
                                                            const int* __restrict__ piv) {
    const int j = c_lo + blockIdx.x * 256 + threadIdx.x;
    if (j >= c_hi) return;
    for (int c = 0; c < nb; ++c) {
        const int r0 = j0 + c, r1 = piv[r0];
        if (r1 != r0) {
            const double t = M[(int64_t)r0 * ldm + j];
            M[(int64_t)r0 * ldm + j] = M[(int64_t)r1 * ldm + j];
            M[(int64_t)r1 * ldm + j] = t;
        }
    }
}

// B[j0 : j0+nb, c_lo : c_hi) := T^-1 B for the nb x nb triangle T = A[j0.., j0..]:
//   upper == 0: unit lower (forward substitution, L11),  upper == 1: upper with its diagonal (back substitution, U11).
// One thread per column, the column's nb values in registers, T in LDS.
__global__ __launch_bounds__(256) void lu_trsm_block_kernel(const double* __restrict__ A, int64_t lda, int j0, int nb, double* __restrict__ B,
                                                             int64_t ldb, int c_lo, int c_hi, int upper) {
    __shared__ double s_t[LU_NB][LU_NB + 1];
    for (int v = threadIdx.x; v < nb * nb; v += 256) s_t[v / nb][v % nb] = A[(int64_t)(j0 + v / nb) * lda + j0 + v % nb];
    __syncthreads();
    const int j = c_lo + blockIdx.x * 256 + threadIdx.x;
    if (j >= c_hi) return;
    double x[LU_NB];
#pragma unroll
    for (int r = 0; r < LU_NB; ++r) x[r] = r < nb ? B[(int64_t)(j0 + r) * ldb + j] : 0.0;
    if (!upper) {
#pragma unroll
        for (int c = 0; c < LU_NB; ++c)
#pragma unroll
            for (int r = c + 1; r < LU_NB; ++r)
                if (r < nb) x[r] = fma(-s_t[r][c], x[c], x[r]);
    } else {
#pragma unroll
        for (int c = LU_NB - 1; c >= 0; --c) {
            if (c < nb) {
                x[c] = x[c] / s_t[c][c];
#pragma unroll
                for (int r = 0; r < c; ++r) x[r] = fma(-s_t[r][c], x[c], x[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < LU_NB; ++r)
        if (r < nb) B[(int64_t)(j0 + r) * ldb + j] = x[r];
}

// A (upper triangle) := mirror of the lower triangle (the assembly only writes the lower 128-tiles)
__global__ __launch_bounds__(256) void mirror_lower_f64_kernel(double* __restrict__ A, int64_t lda, int n) {
    const int i = blockIdx.x;
    for (int j = i + 1 + threadIdx.x; j < n; j += 256) A[(int64_t)i * lda + j] = A[(int64_t)j * lda + i];
}

// In place: A = P L U (row-major n x n, lda), piv[n] device ints; info: 1 + first exactly-zero pivot column, else untouched.
int lu_factor(double* A, int64_t lda, int n, int* piv, int* info, double* B, int64_t ldb, int nrhs, hipStream_t st) {
    for (int j0 = 0; j0 < n; j0 += LU_NB) {
        const int nb = n - j0 < LU_NB ? n - j0 : LU_NB;
        hipLaunchKernelGGL(lu_panel_kernel, dim3(1), dim3(LU_T), 0, st, A, lda, n, j0, nb, piv, info);
        // the rest of the swapped rows: left of the panel, right of it, and the right-hand sides
        if (j0 > 0)
            hipLaunchKernelGGL(lu_swap_rows_kernel, dim3((j0 + 255) / 256), dim3(256), 0, st, A, lda, 0, j0, j0, nb, piv);
        const int right = n - j0 - nb;
        if (right > 0)
            hipLaunchKernelGGL(lu_swap_rows_kernel, dim3((right + 255) / 256), dim3(256), 0, st, A, lda, j0 + nb, n, j0, nb, piv);
        if (B && nrhs > 0)
            hipLaunchKernelGGL(lu_swap_rows_kernel, dim3((nrhs + 255) / 256), dim3(256), 0, st, B, ldb, 0, nrhs, j0, nb, piv);
        if (right > 0) {
            // U12 = L11^-1 A12, then A22 -= L21 U12
            hipLaunchKernelGGL(lu_trsm_block_kernel, dim3((right + 255) / 256), dim3(256), 0, st, A, lda, j0, nb, A, lda, j0 + nb, n, 0);
            GemmShape g{A + (int64_t)(j0 + nb) * lda + j0, lda, A + (int64_t)j0 * lda + j0 + nb, lda, right, right, nb, 0};
            launch_gemm_f64<true, false>(g, EpiAxpby{A + (int64_t)(j0 + nb) * lda + j0 + nb, lda, -1.0, 1.0}, st);
        }
    }
    return check_launch("lu_factor");
}

// B (n x nrhs, already row-permuted by lu_factor) := U^-1 L^-1 B
int lu_substitute(const double* A, int64_t lda, int n, double* B, int64_t ldb, int nrhs, hipStream_t st) {
    const unsigned gx = (unsigned)((nrhs + 255) / 256);
    for (int j0 = 0; j0 < n; j0 += LU_NB) {             // forward: L y = b
        const int nb = n - j0 < LU_NB ? n - j0 : LU_NB;
        hipLaunchKernelGGL(lu_trsm_block_kernel, dim3(gx), dim3(256), 0, st, A, lda, j0, nb, B, ldb, 0, nrhs, 0);
        const int below = n - j0 - nb;
        if (below > 0) {
            GemmShape g{A + (int64_t)(j0 + nb) * lda + j0, lda, B + (int64_t)j0 * ldb, ldb, below, nrhs, nb, 0};
            launch_gemm_f64<true, false>(g, EpiAxpby{B + (int64_t)(j0 + nb) * ldb, ldb, -1.0, 1.0}, st);
        }
    }
    for (int j0 = ((n - 1) / LU_NB) * LU_NB; j0 >= 0; j0 -= LU_NB) {     // backward: U x = y
        const int nb = n - j0 < LU_NB ? n - j0 : LU_NB;
        hipLaunchKernelGGL(lu_trsm_block_kernel, dim3(gx), dim3(256), 0, st, A, lda, j0, nb, B, ldb, 0, nrhs, 1);
        if (j0 > 0) {
            GemmShape g{A + j0, lda, B + (int64_t)j0 * ldb, ldb, j0, nrhs, nb, 0};
            launch_gemm_f64<true, false>(g, EpiAxpby{B, ldb, -1.0, 1.0}, st);
        }
    }
    return check_launch("lu_substitute");
}

__global__ __launch_bounds__(256) void lu_prep_kernel(const float* __restrict__ K, const float* __restrict__ Zc, const float* __restrict__ zs_t,
                                                       int N, int d, int h, double s, double layers_left, double* __restrict__ Kt64,
                                                       int Np, int dp, double* __restrict__ Rt, int hp) {
    const int n = blockIdx.x;
    for (int j = threadIdx.x; j < dp; j += 256) Kt64[(int64_t)n * dp + j] = (n < N && j < d) ? (double)K[(int64_t)n * d + j] * s : 0.0;
    for (int i = threadIdx.x; i < hp; i += 256) {
        double v = 0.0;
        if (n < N && i < h) {
            const float src = zs_t[(int64_t)n * h + i] - Zc[(int64_t)n * h + i];
            v = ((double)src * s) / layers_left;
        }
        Rt[(int64_t)n * hp + i] = v;
    }
}

// B[d][ldb] = Kt64[Np][dp]^T (first N columns), 32 x 32 LDS tiles
__global__ __launch_bounds__(256) void lu_transpose_kernel(const double* __restrict__ src, int64_t lds_, double* __restrict__ dst, int64_t ldd,
                                                            int rows, int cols) {
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = (by + r < rows && bx + tx < cols) ? src[(int64_t)(by + r) * lds_ + bx + tx] : 0.0;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (bx + r < cols && by + tx < rows) dst[(int64_t)(bx + r) * ldd + by + tx] = tile[tx][r];
}

struct LuWorkspace {
    int64_t Np, dp, hp, ldn;
    int64_t off_A, off_K, off_R, off_X, off_piv, total;      // doubles
    LuWorkspace(int64_t N, int64_t d, int64_t h) {
        Np = round_up(N, NPAD);
        dp = round_up(d, NB);
        hp = round_up(h, 2);
        ldn = round_up(N, 2);
        int64_t o = 0;
        off_A = o; o += dp * dp;
        off_K = o; o += Np * dp;
        off_R = o; o += Np * hp;
        off_X = o; o += dp * ldn;
        off_piv = o; o += (dp + 1) / 2 + 2;
        total = o;
    }
};

}  // namespace emcid

using namespace emcid;

extern "C" {

int64_t emcid_edit_lu_workspace_bytes(int64_t N, int64_t d, int64_t h) {
    if (N <= 0 || d <= 0 || h <= 0) return 0;
    return LuWorkspace(N, d, h).total * (int64_t)sizeof(double);
}

/* Plain pivoted LU solve A X = B (test hook): A [n][lda] is overwritten by its factors, B [n][ldb] by X. */
int emcid_lu_solve_f64(double* A, int64_t lda, int64_t n, double* B, int64_t ldb, int64_t nrhs, int* piv_dev, int* info_dev, void* stream) {
    EMCID_CHECK_ARG(A && B && piv_dev && info_dev && n > 0 && nrhs > 0 && n < (1 << 24) && nrhs < (1 << 24));
    EMCID_CHECK_ARG(lda >= n && ldb >= nrhs && lda % 2 == 0 && ldb % 2 == 0 && aligned16(A) && aligned16(B));
    hipStream_t st = (hipStream_t)stream;
    EMCID_TRY(lu_factor(A, lda, (int)n, piv_dev, info_dev, B, ldb, (int)nrhs, st));
    return lu_substitute(A, lda, (int)n, B, ldb, (int)nrhs, st);
}

/* One edited layer with the reference's own solver semantics (see the header): same inputs/outputs as emcid_edit_layer_f64
 * except that adj_k comes out in the reference's orientation, adjk_out [d][N]. */
int emcid_edit_layer_lu_f64(const float* K, const float* Zc, const float* zs_t, const float* C, int64_t N, int64_t d, int64_t h,
                            double lam, double edit_weight, int layers_left, const float* W0, float* W, double* adjk_out,
                            double* Rt_out, float* dW_out, void* workspace, int64_t workspace_bytes, int* info_dev, void* stream) {
    EMCID_CHECK_ARG(K && Zc && zs_t && C && N > 0 && d > 0 && h > 0 && layers_left > 0 && workspace && info_dev);
    EMCID_CHECK_ARG(N < (1 << 24) && d <= 32768 && h <= 32768 && aligned16(workspace));
    EMCID_CHECK_ARG((W == nullptr) || (W0 != nullptr));
    LuWorkspace ws(N, d, h);
    if (workspace_bytes < ws.total * (int64_t)sizeof(double)) return fail(EMCID_ERR_WORKSPACE, __func__, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* base = (double*)workspace;
    double *A = base + ws.off_A, *Kt = base + ws.off_K, *R = base + ws.off_R, *X = base + ws.off_X;
    int* piv = (int*)(base + ws.off_piv);
    const double s = sqrt(edit_weight / 0.5);
    const float cw = (float)(1.0 - edit_weight);
    hipLaunchKernelGGL(lu_prep_kernel, dim3((unsigned)ws.Np), dim3(256), 0, st, K, Zc, zs_t, (int)N, (int)d, (int)h, s, (double)layers_left,
                       Kt, (int)ws.Np, (int)ws.dp, R, (int)ws.hp);
    EMCID_CHECK_LAUNCH();
    // A = lam * double(fl32(fl32(C*cw)/0.5f)) + Kt^T Kt on the lower tiles (the Cholesky path's assembly), mirrored to a full
    // matrix; only its leading d x d part is factored (no identity padding: the pivot search must not see it)
    EMCID_TRY(emcid_assemble_spd_f64(C, d, Kt, ws.Np, d, ws.dp, lam, cw, A, ws.dp, stream));
    hipLaunchKernelGGL(mirror_lower_f64_kernel, dim3((unsigned)d), dim3(256), 0, st, A, ws.dp, (int)d);
    hipLaunchKernelGGL(lu_transpose_kernel, dim3((unsigned)((d + 31) / 32), (unsigned)((N + 31) / 32)), dim3(256), 0, st, Kt, ws.dp, X,
                       ws.ldn, (int)N, (int)d);
    EMCID_TRY(lu_factor(A, ws.dp, (int)d, piv, info_dev, X, ws.ldn, (int)N, st));
    EMCID_TRY(lu_substitute(A, ws.dp, (int)d, X, ws.ldn, (int)N, st));          // X = adj_k  [d][N]
    if (W || dW_out) {                                                           // U = R^T adj_k^T  [h][d]
        GemmShape g{R, ws.hp, X, ws.ldn, (int)h, (int)d, (int)N, 0};
        launch_gemm_f64<false, true>(g, EpiDeltaW{W0, W, d, dW_out, d, nullptr, d}, st);
    }
    if (adjk_out && hipMemcpy2DAsync(adjk_out, N * sizeof(double), X, ws.ldn * sizeof(double), N * sizeof(double), d,
                                     hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(EMCID_ERR_HIP, __func__, "hipMemcpy2DAsync");
    if (Rt_out && hipMemcpy2DAsync(Rt_out, h * sizeof(double), R, ws.hp * sizeof(double), h * sizeof(double), N,
                                   hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(EMCID_ERR_HIP, __func__, "hipMemcpy2DAsync");
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

}  // extern "C"
