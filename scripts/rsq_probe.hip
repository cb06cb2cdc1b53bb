// Accuracy of v_rsq_f64 (and of one Newton / one third-order correction on top of it) on gfx950.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* x, double* raw, double* n1, double* h1, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double p = x[i];
    const double r = __builtin_amdgcn_rsq(p);
    raw[i] = r;
    const double e = fma(-p * r, r, 1.0);
    n1[i] = fma(r * 0.5, e, r);
    h1[i] = fma(r * e, fma(e, 0.375, 0.5), r);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-30.0, 30.0);
    for (auto& v : x) v = std::exp(u(g));
    double *dx, *da, *db, *dc;
    hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dc, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, da, db, dc, n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
    double ea = 0, eb = 0, ec = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / sqrtl((long double)x[i]);
        ea = fmax(ea, (double)fabsl((a[i] - t) / t));
        eb = fmax(eb, (double)fabsl((b[i] - t) / t));
        ec = fmax(ec, (double)fabsl((c[i] - t) / t));
    }
    printf("max rel err: v_rsq_f64 %.3e (2^%.1f)   +1 Newton %.3e   +1 third-order %.3e   (eps = 1.1e-16)\n", ea, log2(ea), eb, ec);
    return 0;
}
