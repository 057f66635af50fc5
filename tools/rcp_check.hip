// Accuracy of v_rcp_f64 and of rcp + k Newton steps on t in [1, 2] (the sigmoid's denominators).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const double* t, double* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = t[i];
    double y0 = __builtin_amdgcn_rcp(x);
    double y1 = fma(fma(-x, y0, 1.0), y0, y0);
    double y2 = fma(fma(-x, y1, 1.0), y1, y1);
    o[3 * i] = y0; o[3 * i + 1] = y1; o[3 * i + 2] = y2;
}
int main() {
    const int n = 1 << 20;
    double* h = (double*)malloc(n * sizeof(double));
    for (int i = 0; i < n; ++i) h[i] = 1.0 + (double)((i * 2654435761u) & 0xfffff) / 1048576.0 + 1e-9 * i / n;
    double *d, *o; hipMalloc(&d, n * 8); hipMalloc(&o, 3 * n * 8);
    hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, o, n);
    double* r = (double*)malloc(3 * n * 8);
    hipMemcpy(r, o, 3 * n * 8, hipMemcpyDeviceToHost);
    double e[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {
        long double ex = 1.0L / (long double)h[i];
        for (int j = 0; j < 3; ++j) { double er = fabs((double)(((long double)r[3 * i + j] - ex) / ex)); if (er > e[j]) e[j] = er; }
    }
    printf("max rel err: rcp %.3e  +1 NR %.3e  +2 NR %.3e  (eps = %.3e)\n", e[0], e[1], e[2], 2.220446049250313e-16);
    return 0;
}
