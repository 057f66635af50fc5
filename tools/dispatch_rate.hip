// How fast does the chip START workgroups?  An (almost) empty kernel with the forward's / the backward's launch shape (256 threads +
// 53.7 KB of LDS, 512 threads + 78.9 KB) at the grid sizes of the bench: the launch duration is what the dispatcher needs to hand out
// the workgroups (plus one workgroup's few hundred cycles).   hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate.hip -o tools/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NT, int LDS_DOUBLES>
__global__ __launch_bounds__(NT) void k_touch(double* out, int spin) {
    __shared__ double sm[LDS_DOUBLES];
    sm[threadIdx.x] = (double)threadIdx.x;
    __syncthreads();
    double a = sm[(threadIdx.x * 7) % NT];
    for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;          // spin > 0: a workgroup that lives for a while
    if (a == -1.0) out[blockIdx.x] = a;
}

template <int NT, int LDS_DOUBLES>
static void run(const char* name, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spin : {0, 2000}) {
        for (int wgs : {256, 512, 768, 1024, 2048, 4096}) {
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_touch<NT, LDS_DOUBLES>), dim3(wgs), dim3(NT), 0, 0, out, spin);
            hipEventRecord(e0, 0);
            const int reps = 200;
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_touch<NT, LDS_DOUBLES>), dim3(wgs), dim3(NT), 0, 0, out, spin);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s spin %4d: %5d workgroups: %7.2f us per launch (back to back on one stream) = %6.1f ns per workgroup\n", name, spin, wgs,
                   ms * 1e3 / reps, ms * 1e6 / reps / wgs);
        }
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    double* out;
    hipMalloc(&out, 1 << 20);
    run<256, 6712>("forward shape  (256 threads, 53.7 KB LDS)", out);
    run<512, 9864>("backward shape (512 threads, 78.9 KB LDS)", out);
    run<256, 64>("256 threads, 0.5 KB LDS              ", out);
    hipFree(out);
    return 0;
}
