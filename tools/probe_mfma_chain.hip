// Issue interval of v_mfma_f64_16x16x4_f64 from ONE wave: NCH independent accumulator chains, operands in registers.
// hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_chain.hip -o tools/probe_mfma_chain && tools/probe_mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int REPS = 2000;
template <int NCH, int WAVES>
__global__ void k(double* out, long long* cyc) {
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-3;
    double4_t acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = double4_t{0, 0, 0, 0};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll
        for (int it = 0; it < 64; ++it) acc[it % NCH] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[it % NCH], 0, 0, 0);
    }
    asm volatile("" :: "v"(acc[0][0]));
    double s = 0; for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NCH, int WAVES> void run(const char* name) {
    double* out; long long* cyc; hipMalloc(&out, 256 * 64 * WAVES * 8); hipMalloc(&cyc, 256 * 8);
    hipLaunchKernelGGL((k<NCH, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NCH, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
    const double nm = 64.0 * REPS;
    printf("%s: %d chain(s), %d wave(s) per workgroup, one workgroup per CU: %.1f counter ticks per MFMA of a wave; kernel %.3f ms -> counter %.2f GHz, %.1f TFLOP/s, %.1f ns per MFMA and SIMD\n",
           name, NCH, WAVES, m / nm, ms, m / (ms * 1e6), 256.0 * WAVES * nm * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / (nm * (WAVES > 4 ? WAVES / 4.0 : 1.0)));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1, 1>("mfma_f64_16x16x4"); run<2, 1>("mfma_f64_16x16x4"); run<4, 1>("mfma_f64_16x16x4");
    run<1, 4>("mfma_f64_16x16x4"); run<2, 4>("mfma_f64_16x16x4"); run<1, 8>("mfma_f64_16x16x4"); run<2, 8>("mfma_f64_16x16x4");
    return 0;
}
