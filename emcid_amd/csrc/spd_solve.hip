// Stage 2 of EMCID on gfx950: assemble A = lam*C' + K K^T, blocked fp64 Cholesky, blocked TRSM with the
// N concept columns as right-hand sides, dW = R X^T and the in-place fp32 weight update.
// Replaces emcid/emcid_main.py:1016-1061 of the reference (torch.linalg.solve + `@` in fp64).
//
// Everything dense runs through gemm_f64.h (v_mfma_f64_16x16x4_f64).  The Cholesky is right-looking
// with NB = 128: a single-workgroup LDS leaf factors the diagonal block and inverts it, so the panel
// solve and every later triangular solve are MFMA GEMMs against the inverted diagonal blocks.
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
    hipEventRecord(g_prof_ev[g_prof_n][0], st);
}
void prof_end(int cls, hipStream_t st) {
    if (!(g_prof_mask & (1u << cls)) || g_prof_n >= PROF_MAX) return;
    hipEventRecord(g_prof_ev[g_prof_n][1], st);
    ++g_prof_n;
}

// ---- element-wise preparation ---------------------------------------------------------------------

// Kt64[n][j] = double(K[n][j]) * s (zero padded to [Np][dp]);
// Rt[n][i]   = double(zs_t[n][i] - Zc[n][i]) * s / layers_left  (fp32 subtract first, like the reference).
__global__ __launch_bounds__(256) void prep_kr_kernel(const float* __restrict__ K, const float* __restrict__ Zc,
                                                       const float* __restrict__ zs_t, int N, int d, int h, double s,
                                                       double layers_left, double* __restrict__ Kt64, int Np, int dp,
                                                       double* __restrict__ Rt, int hp) {
    const int n = blockIdx.x;
    for (int j = threadIdx.x; j < dp; j += 256) {
        double v = 0.0;
        if (n < N && j < d) v = (double)K[(int64_t)n * d + j] * s;
        Kt64[(int64_t)n * dp + j] = v;
    }
    if (Rt) {
        for (int i = threadIdx.x; i < hp; i += 256) {
            double v = 0.0;
            if (n < N && i < h) {
                const float src = zs_t[(int64_t)n * h + i] - Zc[(int64_t)n * h + i];
                v = ((double)src * s) / layers_left;
            }
            Rt[(int64_t)n * hp + i] = v;
        }
    }
}

__global__ __launch_bounds__(256) void copy2d_f64_kernel(const double* __restrict__ src, int64_t lds_, double* __restrict__ dst,
                                                          int64_t ldd, int rows, int cols) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += 256) dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds_ + c];
}

__global__ __launch_bounds__(256) void axpy_f32_kernel(float* __restrict__ W, const float* __restrict__ dW, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) W[i] += dW[i];
}

// ---- diagonal leaf: Cholesky of one NB x NB block + its inverse, one workgroup, all in LDS ------------

constexpr int LEAF_T = 512;
constexpr int SLD = NB + 1;

__global__ __launch_bounds__(LEAF_T) void chol_leaf_kernel(const double* __restrict__ A, int64_t lda, double* __restrict__ L,
                                                            int64_t ldl, double* __restrict__ inv, int* info, int col0) {
    __shared__ double S[NB * SLD];
    __shared__ double sq[NB];
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += LEAF_T) {
        const int i = e / NB, j = e % NB;
        S[i * SLD + j] = (j <= i) ? A[(int64_t)i * lda + j] : 0.0;
    }
    __syncthreads();

    // Square-root-free right-looking elimination: column k keeps its UNSCALED values (= L[i][k]*sqrt(d_k)),
    // so a step only writes columns > k and needs one barrier; the scaling is a parallel pass afterwards.
    const int ty = tid >> 5, tx = tid & 31;
    for (int k = 0; k < NB - 1; ++k) {
        const double dkk = S[k * SLD + k];
        const double invd = 1.0 / dkk;
        for (int i = k + 1 + ty; i < NB; i += LEAF_T / 32) {
            const double lik = S[i * SLD + k] * invd;
            for (int j = k + 1 + tx; j <= i; j += 32) S[i * SLD + j] -= lik * S[j * SLD + k];
        }
        __syncthreads();
    }
    if (tid < NB) {
        const double dkk = S[tid * SLD + tid];
        if (!(dkk > 0.0)) atomicCAS(info, 0, col0 + tid + 1);  // not SPD (or NaN): report first seen pivot
        sq[tid] = sqrt(dkk);
    }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += LEAF_T) {
        const int i = e / NB, j = e % NB;
        double v = 0.0;
        if (j < i) v = S[i * SLD + j] / sq[j];
        else if (j == i) v = sq[j];
        S[i * SLD + j] = v;
        L[(int64_t)i * ldl + j] = v;
    }
    __syncthreads();

    // In-place inverse of the lower-triangular block, last column first:
    //   X[j][j] = 1/L[j][j];  X[i][j] = -X[j][j] * sum_{k=j+1..i} X[i][k] * L[k][j]   (i > j)
    // row i is owned by 4 consecutive lanes that split the k range and reduce with DPP shuffles.
    const int row = tid >> 2, part = tid & 3;
    for (int j = NB - 1; j >= 0; --j) {
        const double xjj = 1.0 / S[j * SLD + j];
        double t = 0.0;
        if (row > j) {
            for (int k = j + 1 + part; k <= row; k += 4) t += S[row * SLD + k] * S[k * SLD + j];
        }
        t += __shfl_xor(t, 1);
        t += __shfl_xor(t, 2);
        __syncthreads();  // every read of column j (rows > j) is done
        if (part == 0) {
            if (row > j) S[row * SLD + j] = -xjj * t;
            else if (row == j) S[j * SLD + j] = xjj;
        }
        __syncthreads();
    }
    for (int e = tid; e < NB * NB; e += LEAF_T) {
        const int i = e / NB, j = e % NB;
        inv[e] = (j <= i) ? S[i * SLD + j] : 0.0;
    }
}

// ---- host orchestration -------------------------------------------------------------------------------

static int cholesky_impl(double* A, double* L, int64_t dp, int64_t lda, double* invdiag, int* info, hipStream_t st) {
    const int nb = (int)(dp / NB);
    for (int j = 0; j < nb; ++j) {
        const int64_t o = (int64_t)j * NB;
        double* inv = invdiag + (int64_t)j * NB * NB;
        {
            ScopedProf sp(KC_CHOL_LEAF, st);
            hipLaunchKernelGGL(chol_leaf_kernel, dim3(1), dim3(LEAF_T), 0, st, A + o * lda + o, lda, L + o * lda + o, lda,
                               inv, info, (int)o);
        }
        const int m = (int)(dp - o - NB);
        if (m == 0) break;
        // panel: L21 = A21 * inv(L11)^T
        GemmShape ps{A + (o + NB) * lda + o, lda, inv, NB, m, NB, NB, 0};
        {
            ScopedProf sp(KC_CHOL_PANEL, st);
            launch_gemm_f64<true, true>(ps, EpiAxpby{L + (o + NB) * lda + o, lda, 1.0, 0.0}, st, 1);
        }
        // trailing: A22 -= L21 * L21^T (lower tiles only)
        const double* L21 = L + (o + NB) * lda + o;
        GemmShape ts{L21, lda, L21, lda, m, m, NB, 1};
        {
            ScopedProf sp(KC_CHOL_TRAIL, st);
            launch_gemm_f64<true, true>(ts, EpiAxpby{A + (o + NB) * lda + (o + NB), lda, -1.0, 1.0}, st, 1);
        }
    }
    return check_launch("emcid_cholesky_f64");
}

static int cholesky_solve_impl(const double* L, int64_t dp, int64_t lda, const double* invdiag, double* Bt, double* Yt,
                               int64_t Np, int64_t ldb, hipStream_t st) {
    const int nb = (int)(dp / NB);
    const int M = (int)Np;
    for (int j = 0; j < nb; ++j) {  // forward: Yt L^T = Bt
        const int64_t o = (int64_t)j * NB;
        const double* inv = invdiag + (int64_t)j * NB * NB;
        GemmShape a{Bt + o, ldb, inv, NB, M, NB, NB, 0};
        {
            ScopedProf sp(KC_TRSM_DIAG, st);
            launch_gemm_f64<true, true>(a, EpiAxpby{Yt + o, ldb, 1.0, 0.0}, st, 1);
        }
        const int m = (int)(dp - o - NB);
        if (m > 0) {
            GemmShape b{Yt + o, ldb, L + (o + NB) * lda + o, lda, M, m, NB, 0};
            ScopedProf sp(KC_TRSM_UPDATE, st);
            launch_gemm_f64<true, true>(b, EpiAxpby{Bt + o + NB, ldb, -1.0, 1.0}, st, 1);
        }
    }
    for (int j = nb - 1; j >= 0; --j) {  // backward: Xt L = Yt, Xt written over Bt
        const int64_t o = (int64_t)j * NB;
        const double* inv = invdiag + (int64_t)j * NB * NB;
        GemmShape a{Yt + o, ldb, inv, NB, M, NB, NB, 0};
        {
            ScopedProf sp(KC_TRSM_DIAG, st);
            launch_gemm_f64<true, false>(a, EpiAxpby{Bt + o, ldb, 1.0, 0.0}, st, 1);
        }
        if (o > 0) {
            GemmShape b{Bt + o, ldb, L + o * lda, lda, M, (int)o, NB, 0};
            ScopedProf sp(KC_TRSM_UPDATE, st);
            launch_gemm_f64<true, false>(b, EpiAxpby{Yt, ldb, -1.0, 1.0}, st, 1);
        }
    }
    return check_launch("emcid_cholesky_solve_f64");
}

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
        off_inv = o; o += (dp / NB) * NB * NB;
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

int emcid_edit_layer_f64(const float* K, const float* Zc, const float* zs_t, const float* C, int64_t N, int64_t d, int64_t h,
                         double lam, double edit_weight, int layers_left, const float* W0, float* W, double* Xt_out,
                         double* Rt_out, float* dW_out, void* workspace, int64_t workspace_bytes, int* info_dev,
                         void* stream) {
    EMCID_CHECK_ARG(K && Zc && zs_t && C && N > 0 && d > 0 && h > 0 && layers_left > 0 && workspace && info_dev);
    EMCID_CHECK_ARG(N < (1 << 24) && d <= 32768 && h <= 32768);
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

    {
        ScopedProf sp(KC_PREP, st);
        hipLaunchKernelGGL(prep_kr_kernel, dim3((unsigned)ws.Np), dim3(256), 0, st, K, Zc, zs_t, (int)N, (int)d, (int)h, s,
                           (double)layers_left, B, (int)ws.Np, (int)ws.dp, R, (int)ws.hp);
    }
    EMCID_CHECK_LAUNCH();
    EMCID_TRY(emcid_assemble_spd_f64(C, d, B, ws.Np, d, ws.dp, lam, cw, A, ws.dp, stream));
    EMCID_TRY(cholesky_impl(A, L, ws.dp, ws.dp, inv, info_dev, st));
    EMCID_TRY(cholesky_solve_impl(L, ws.dp, ws.dp, inv, B, Y, ws.Np, ws.dp, st));
    if (W || dW_out) EMCID_TRY(emcid_delta_w_f64(R, ws.hp, B, ws.dp, ws.Np, h, d, W0, W, d, dW_out, nullptr, stream));
    if (Xt_out) hipLaunchKernelGGL(copy2d_f64_kernel, dim3((unsigned)N), dim3(256), 0, st, B, ws.dp, Xt_out, d, (int)N, (int)d);
    if (Rt_out) hipLaunchKernelGGL(copy2d_f64_kernel, dim3((unsigned)N), dim3(256), 0, st, R, ws.hp, Rt_out, h, (int)N, (int)h);
    EMCID_CHECK_LAUNCH();
    return EMCID_OK;
}

}  // extern "C"
