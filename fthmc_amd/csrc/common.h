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

// fp64 FMAs with a CONSTANT operand held in an SGPR pair.  Left to itself hipcc (kernels here run at ~100 SGPRs) builds
// such constants in a VGPR pair with two v_mov_b32 and issues v_fmac_f64: three VALU instructions per polynomial
// step instead of one (s_mov_b32 runs on the scalar unit, off the fp64 pipe).
__device__ __forceinline__ double ft_fma_vvs(double a, double b, double C) {       // a * b + C
    double d; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(C)); return d;
}
__device__ __forceinline__ double ft_fma_nvsv(double a, double C, double b) {      // -a * C + b
    double d; asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(d) : "v"(a), "s"(C), "v"(b)); return d;
}
__device__ __forceinline__ double ft_mul_vs(double a, double C) {
    double d; asm("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(C)); return d;
}

// sin and cos for moderate arguments (|x| up to ~1e5; plaquette angles are sums of four links):
// Cody-Waite reduction by pi/2 in three exact-product pieces + the fdlibm kernel polynomials on
// |r| <= pi/4.  ~35 DP ops, no slow path, < 1 ulp each (ocml's sincos carries a Payne-Hanek
// branch and is ~3x longer on the critical path of the serial stages).
__device__ __forceinline__ void ft_sincos(double x, double* sn, double* cs) {
    const double fn = rint(ft_mul_vs(x, 6.36619772367581382433e-01));   // x * 2/pi
    double r = ft_fma_nvsv(fn, 1.57079632673412561417e+00, x);         // pi/2, first 33 bits (exact product)
    r = ft_fma_nvsv(fn, 6.07710050630396597660e-11, r);                // next 33 bits
    r = ft_fma_nvsv(fn, 2.02226624879595063154e-21, r);                // tail
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = ft_fma_vvs(ps, z, -2.50507602534068634195e-08);
    ps = ft_fma_vvs(ps, z, 2.75573137070700676789e-06);
    ps = ft_fma_vvs(ps, z, -1.98412698298579493134e-04);
    ps = ft_fma_vvs(ps, z, 8.33333333332248946124e-03);
    ps = ft_fma_vvs(ps, z, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = ft_fma_vvs(pc, z, 2.08757232129817482790e-09);
    pc = ft_fma_vvs(pc, z, -2.75573143513906633035e-07);
    pc = ft_fma_vvs(pc, z, 2.48015872894767294178e-05);
    pc = ft_fma_vvs(pc, z, -1.38888888888741095749e-03);
    pc = ft_fma_vvs(pc, z, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)fn & 3;
    const double s_ = (q & 1) ? c : s, c_ = (q & 1) ? s : c;
    *sn = (q & 2) ? -s_ : s_;
    *cs = ((q + 1) & 2) ? -c_ : c_;
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
