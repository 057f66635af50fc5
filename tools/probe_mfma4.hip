// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: (1) lane layout of A, B, D; (2) issue rate of a dependent /
// independent accumulator chain with the A operand streamed from LDS (ds_read_b64 with immediate offsets).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_mfma4.hip -o tools/probe_mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

__global__ void k_layout(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

template <int NACC>
__global__ __launch_bounds__(512) void k_rate(double* out, long long* cyc, int iters) {
    __shared__ double sm[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = 1e-3 * i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const double* base = sm + (lane & 15) + 18 * ((lane >> 2) & 3);
    double w[18];
#pragma unroll
    for (int t = 0; t < 18; ++t) w[t] = 1.0 + 1e-6 * (t + lane);
    double acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 18; ++t) {
#pragma unroll
            for (int q = 0; q < NACC; ++q)
                acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(base[q * 64 + t * 20 + (it & 7) * 400], w[t], acc[q], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    // ---- layout
    std::vector<double> a(64), b(64), d(64);
    srand(1);
    for (int l = 0; l < 64; ++l) { a[l] = (rand() % 1000) / 100.0; b[l] = (rand() % 1000) / 100.0; }
    double *da, *db, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
    // hypotheses: index of (x, y) inside a block's 16 lanes: x + 4 y  (mode 0) or 4 x + y (mode 1)
    auto idx = [](int mode, int x, int y) { return mode == 0 ? x + 4 * y : 4 * x + y; };
    for (int fa = 0; fa < 2; ++fa) for (int fb = 0; fb < 2; ++fb) for (int fd = 0; fd < 2; ++fd) {
        double err = 0;
        for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += a[16 * blk + idx(fa, i, k)] * b[16 * blk + idx(fb, j, k)];
            err = fmax(err, fabs(s - d[16 * blk + idx(fd, j, i)]));
        }
        printf("A lane = blk*16 + %s, B lane = blk*16 + %s, D lane = blk*16 + %s : max err %.3e%s\n",
               fa ? "4 i + k" : "i + 4 k", fb ? "4 j + k" : "j + 4 k", fd ? "4 j + i" : "j + 4 i", err, err < 1e-9 ? "   <== MATCH" : "");
    }
    // ---- rate
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int CU = p.multiProcessorCount;
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * CU * 2 * 512); hipMalloc(&cyc, sizeof(long long) * CU * 2);
    std::vector<long long> hc(CU * 2);
    const int iters = 64;
    for (int wpb : {4, 8}) for (int bpc : {1, 2}) {
        auto run = [&](auto kern, int nacc) {
            hipLaunchKernelGGL(kern, dim3(CU * bpc), dim3(wpb * 64), 0, 0, out, cyc, iters);
            hipLaunchKernelGGL(kern, dim3(CU * bpc), dim3(wpb * 64), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
            hipMemcpy(hc.data(), cyc, sizeof(long long) * CU * bpc, hipMemcpyDeviceToHost);
            double m = 0; for (int i = 0; i < CU * bpc; ++i) m += hc[i]; m /= CU * bpc;
            const double wps = wpb * bpc / 4.0;   // waves per SIMD
            printf("waves/SIMD %.0f, %d accumulators: %.1f cycles per MFMA per wave, %.1f cycles per MFMA per SIMD\n",
                   wps, nacc, m / (iters * 18.0 * nacc), m / (iters * 18.0 * nacc * wps));
        };
        run(k_rate<1>, 1); run(k_rate<2>, 2); run(k_rate<4>, 4);
    }
    return 0;
}
