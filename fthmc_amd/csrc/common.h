// Shared device helpers for the gfx950 ftHMC kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/fthmc_hip.h"

#define FT_PI      3.14159265358979323846
#define FT_TWO_PI  6.28318530717958647692

#define FT_WAVE 64

namespace fthmc { void note_hip_error(hipError_t e, const char* file, int line); }
#define FT_LAUNCH_CHECK()                                                              \
    do { hipError_t e_ = hipGetLastError();                                            \
         if (e_ != hipSuccess) { fthmc::note_hip_error(e_, __FILE__, __LINE__); return FTHMC_ERR_LAUNCH; } } while (0)

// torch.remainder(x + pi, 2 pi) - pi   (fmod is exact; sign fix as ATen does)
// Within three periods of the principal range the remainder is one exact subtraction (Sterbenz) or
// the same rounded addition ATen performs, so the short path is bit-identical to fmod's.
__device__ __forceinline__ double ft_wrap(double x) {
    double r = x + FT_PI;
    if (r >= -FT_TWO_PI && r < 2.0 * FT_TWO_PI) {
        r = r >= FT_TWO_PI ? r - FT_TWO_PI : (r < 0.0 ? r + FT_TWO_PI : r);
    } else {
        r = fmod(r, FT_TWO_PI);
        if (r < 0.0) r += FT_TWO_PI;
    }
    return r - FT_PI;
}

// qed_helpers.regularize: 2 pi (f_ - floor(f_) - 0.5), f_ = (f - pi) / 2 pi
__device__ __forceinline__ double ft_regularize(double f) {
    double f_ = (f - FT_PI) / FT_TWO_PI;
    return FT_TWO_PI * (f_ - floor(f_) - 0.5);
}

__device__ __forceinline__ int ft_modL(int v, int L) {   // v >= -L
    int r = (v + L) % L;
    return r;
}

// wave64 sum, result valid in every lane
__device__ __forceinline__ double ft_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, FT_WAVE);
    return v;
}

// Block sum for blockDim.x <= 1024 (multiple of 64); `red` holds >= 16 doubles.
// Result valid in every thread.  Fixed order => deterministic.
__device__ __forceinline__ double ft_block_sum(double v, double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    v = ft_wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// stripe class of a lattice site for layer (mu, off): 0 active, 1|2 frozen, 3 passive
// (fthmc/utils/layers.py:213-292)
__device__ __forceinline__ int ft_stripe(int i, int j, int mu, int off) {
    const int s = mu == 0 ? j : i;
    return (s - off) & 3;            // L % 4 == 0 and s >= 0
}

// Also drops any stale, non-sticky error a previous runtime call of the host framework left
// behind, so that FT_LAUNCH_CHECK only reports our own launches.
static inline hipStream_t ft_stream(void* s) { (void)hipGetLastError(); return reinterpret_cast<hipStream_t>(s); }
