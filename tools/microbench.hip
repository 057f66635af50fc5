// fp64 pipe microbenchmarks on gfx950: VALU FMA, MFMA f64 16x16x4 / 4x4x4, both at once, HBM copy.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

__global__ void k_fma(double* out, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mfma16(double* out, double a, double b) {
    double4_t c[4];
    for (int i = 0; i < 4; ++i) c[i] = double4_t{0, 0, 0, 0};
    double av = a + threadIdx.x * 1e-9, bv = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c[i], 0, 0, 0);
    }
    double s = 0; for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mfma4(double* out, double a, double b) {
    double c[8];
    for (int i = 0; i < 8; ++i) c[i] = 0;
    double av = a + threadIdx.x * 1e-9, bv = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, c[i], 0, 0, 0);
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// even waves MFMA, odd waves VALU FMA (co-issue test); same trip count
__global__ void k_both(double* out, double a, double b) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double s = 0;
    if (wave & 1) {
        double x[8];
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-9 + i;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
        }
        for (int i = 0; i < 8; ++i) s += x[i];
    } else {
        double4_t c[4];
        for (int i = 0; i < 4; ++i) c[i] = double4_t{0, 0, 0, 0};
        double av = a + threadIdx.x * 1e-9, bv = b;
        for (int it = 0; it < ITERS / 8; ++it) {     // 16 FMA-instr-equivalents per MFMA
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_exp(double* out, double a) {
    double x[4];
    for (int i = 0; i < 4; ++i) x[i] = threadIdx.x * 1e-3 + i * 0.1 + a;
    for (int it = 0; it < ITERS / 16; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = 1.0 / (1.0 + exp(-x[i]));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}

__global__ void k_copy(const double2* __restrict__ in, double2* __restrict__ o, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) o[i] = in[i];
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int CU = p.multiProcessorCount;
    double* out; CK(hipMalloc(&out, sizeof(double) * CU * 8 * 1024));
    for (int wpb : {4, 8, 16}) {
        const int blocks = CU * (wpb == 16 ? 1 : 2), thr = wpb * 64;
        const double nthreads = (double)blocks * thr;
        float t = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(thr), 0, 0, out, 1.000001, 1e-9); }, 5);
        printf("valu fma f64      waves/blk %2d blocks %4d: %8.3f ms  %7.2f TFLOP/s\n", wpb, blocks, t, nthreads * ITERS * 8 * 2 / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL(k_mfma16, dim3(blocks), dim3(thr), 0, 0, out, 1.000001, 1e-9); }, 5);
        printf("mfma f64 16x16x4  waves/blk %2d blocks %4d: %8.3f ms  %7.2f TFLOP/s\n", wpb, blocks, t, nthreads / 64 * ITERS * 4 * 2048.0 / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL(k_mfma4, dim3(blocks), dim3(thr), 0, 0, out, 1.000001, 1e-9); }, 5);
        printf("mfma f64 4x4x4x4b waves/blk %2d blocks %4d: %8.3f ms  %7.2f TFLOP/s\n", wpb, blocks, t, nthreads / 64 * ITERS * 8 * 512.0 / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL(k_both, dim3(blocks), dim3(thr), 0, 0, out, 1.000001, 1e-9); }, 5);
        printf("both (half/half)  waves/blk %2d blocks %4d: %8.3f ms  valu %7.2f + mfma %7.2f TFLOP/s\n", wpb, blocks, t,
               nthreads / 2 * ITERS * 8 * 2 / t / 1e9, nthreads / 128 * (ITERS / 8) * 4 * 2048.0 / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL(k_exp, dim3(blocks), dim3(thr), 0, 0, out, 0.1); }, 5);
        printf("sigmoid f64       waves/blk %2d blocks %4d: %8.3f ms  %7.2f Gsigmoid/s  (%.1f cycles/wave-op at 2.4GHz/SIMD)\n", wpb, blocks, t,
               nthreads * (ITERS / 16) * 4 / t / 1e6, t * 1e-3 * 2.4e9 / ((double)(ITERS / 16) * 4 * (nthreads / 64) / (CU * 4)));
    }
    const size_t n = (size_t)1 << 27;  // 2 GiB of double2
    double2 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 1, n * 16));
    float t = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(CU * 8), dim3(256), 0, 0, a, b, n); }, 5);
    printf("hbm copy 2x%.1f GiB: %.3f ms  %.2f TB/s (read+write)\n", n * 16.0 / (1 << 30), t, 2.0 * n * 16 / t / 1e9);
    return 0;
}
