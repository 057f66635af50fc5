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
int main2();
int main() {
    main2();
    run<1, 1>("mfma_f64_16x16x4"); run<2, 1>("mfma_f64_16x16x4"); run<4, 1>("mfma_f64_16x16x4");
    run<1, 4>("mfma_f64_16x16x4"); run<2, 4>("mfma_f64_16x16x4"); run<1, 8>("mfma_f64_16x16x4"); run<2, 8>("mfma_f64_16x16x4");
    return 0;
}
// ---- LDS-fed variant: both operands of every MFMA come from LDS (one ds_read_b64 each), PFD steps ahead
template <int NCH, int WAVES, int PFD>
__global__ void kl(double* out, long long* cyc) {
    __shared__ double sa[64 * 24 + 64], sb[64 * 24 + 64];
    for (int i = threadIdx.x; i < 64 * 24 + 64; i += blockDim.x) { sa[i] = 1.0 + i * 1e-4; sb[i] = 0.5 + i * 1e-4; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double4_t acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = double4_t{0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < REPS / 4; ++rep) {
        const double* pa = sa + lane; const double* pb = sb + lane + (rep & 1);
#pragma unroll
        for (int it = 0; it < 24; ++it) acc[it % NCH] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[it * 64], pb[it * 64], acc[it % NCH], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * PFD, 0);
#pragma unroll
        for (int q = 0; q < 24 - PFD; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
#pragma unroll
        for (int q = 0; q < PFD; ++q) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    double s = 0; for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NCH, int WAVES, int PFD> void runl() {
    double* out; long long* cyc; hipMalloc(&out, 256 * 64 * WAVES * 8); hipMalloc(&cyc, 256 * 8);
    hipLaunchKernelGGL((kl<NCH, WAVES, PFD>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((kl<NCH, WAVES, PFD>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
    printf("LDS-fed: %d chain(s), %d wave(s) per workgroup, reads %d step(s) ahead: %.1f cycles per MFMA of wave 0\n", NCH, WAVES, PFD, m / (24.0 * (REPS / 4)));
    hipFree(out); hipFree(cyc);
}
int main2() {
    runl<1, 1, 1>(); runl<1, 1, 2>(); runl<1, 1, 4>(); runl<2, 1, 4>(); runl<1, 4, 1>(); runl<1, 4, 4>(); runl<1, 8, 1>(); runl<1, 8, 4>(); runl<2, 8, 4>();
    return 0;
}
