// ft_atan (csrc/common.h) against ocml's atan and, on the host, against long double atanl:
//   hipcc --offload-arch=gfx950 -O3 -I fthmc_amd/csrc -I include tools/atan_check.hip -o tools/atan_check && tools/atan_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "fthmc_hip.h"
#include "common.h"
__global__ void k(const double* x, double* a, double* b, double* w, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = ft_atan(x[i]); b[i] = atan(x[i]); w[i] = ft_wrap_pm_pi(2 * a[i]) - ft_wrap(2 * a[i]); }
}
int main() {
    std::vector<double> x;
    for (int e = -320; e <= 320; ++e) for (int m = 0; m < 64; ++m) { const double v = ldexp(1.0 + m / 64.0, e); x.push_back(v); x.push_back(-v); }
    for (int i = 0; i <= 200000; ++i) { const double v = 1e-3 * i * 0.05; x.push_back(v); x.push_back(-v); x.push_back(1.0 / (v + 1e-9)); }
    const double sp[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, NAN, 1e308, -1e308, 1e300, 1e301, 5e-324, 1.0000000000000002, 0.9999999999999999};
    for (double v : sp) x.push_back(v);
    const int n = (int)x.size();
    double *dx, *da, *db, *dw;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&da, n * 8); (void)hipMalloc(&db, n * 8); (void)hipMalloc(&dw, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(dx, da, db, dw, n);
    std::vector<double> a(n), b(n), w(n);
    (void)hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(w.data(), dw, n * 8, hipMemcpyDeviceToHost);
    double ea = 0, eb = 0, ew = 0; int bad = 0;
    for (int i = 0; i < n; ++i) {
        if (std::isnan(x[i])) { if (!std::isnan(a[i])) { ++bad; printf("NaN lost\n"); } continue; }
        const long double t = atanl((long double)x[i]);
        const double ulp = t == 0 ? 1 : fabs((double)t) * 0x1p-52;
        const double da_ = fabs((double)((long double)a[i] - t)) / ulp, db_ = fabs((double)((long double)b[i] - t)) / ulp;
        if (da_ > ea) ea = da_;
        if (db_ > eb) eb = db_;
        if (std::signbit(a[i]) != std::signbit(x[i])) { ++bad; printf("sign at %g\n", x[i]); }
        if (fabs(w[i]) > ew) ew = fabs(w[i]);
    }
    printf("%d arguments: max error ft_atan %.3f ulp, ocml atan %.3f ulp; ft_wrap_pm_pi - ft_wrap max %.3g; bad %d\n", n, ea, eb, ew, bad);
    return bad != 0 || ea > 2.0 || ew != 0.0;
}
