// Empirical lane layout of v_mfma_f64_4x4x4_4b_f64 (gfx950): one-hot A at lane p, one-hot B at lane q.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(double* d) {
    const int p = blockIdx.x >> 6, q = blockIdx.x & 63, l = threadIdx.x;
    d[blockIdx.x * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(l == p ? 1.0 : 0.0, l == q ? 1.0 : 0.0, 0.0, 0, 0, 0);
}
int main() {
    double* dd; hipMalloc(&dd, 4096 * 64 * 8);
    hipLaunchKernelGGL(k, dim3(4096), dim3(64), 0, 0, dd);
    std::vector<double> d(4096 * 64);
    hipMemcpy(d.data(), dd, d.size() * 8, hipMemcpyDeviceToHost);
    // out[p][q] = D lane that receives A[p] * B[q], or -1
    printf("rows: A lane p; columns: B lane q; entry: D lane receiving the product (.. = none)\n");
    for (int p = 0; p < 64; ++p) {
        printf("p=%2d:", p);
        for (int q = 0; q < 64; ++q) {
            int hit = -1, n = 0;
            for (int l = 0; l < 64; ++l) if (d[(p * 64 + q) * 64 + l] != 0.0) { hit = l; ++n; }
            if (n == 0) printf(" .."); else if (n == 1) printf(" %2d", hit); else printf(" **");
        }
        printf("\n");
    }
    return 0;
}
