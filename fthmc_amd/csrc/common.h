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

// -DFT_DRYRUN (the sanitizer build, `make san`): the HOST side only -- every launch, copy and memset is a no-op that succeeds,
// so that tests/test_sanitizer.py can take every entry point through ALL of its host sequencing (argument checks, workspace
// carving, launch geometry) under AddressSanitizer + UBSan on a box without a GPU.  Never part of the product library.
#ifdef FT_DRYRUN
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(...) do { } while (0)
#define hipMemcpyAsync(...) hipSuccess
#define hipMemsetAsync(...) hipSuccess
#undef FT_LAUNCH_CHECK
#define FT_LAUNCH_CHECK() do { } while (0)
#endif

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

// atan for the tan-mixture transform.  |x| > 1 folds to 1 / |x| (v_rcp_f64 + one third-order step; |x| clamped to 1e300 first, so
// that an infinite tangent gives pi/2 and not 0 * inf), then atan(a) = a + a s p(s), s = a^2, with the 20-coefficient near-minimax
// p of tools/minimax_atan.py (7.7e-17 relative, exact arithmetic) and its constants as SGPR operands.  ~40 DP operations, no
// branch; ocml's atan is ~80: an IEEE division for the fold and its 19 coefficients built in VGPR pairs (38 v_mov_b32).  A NaN
// argument comes back as NaN.  tools/atan_check.hip compares it with ocml's over the range.
__device__ __forceinline__ double ft_atan(double x) {
    double am;
    asm("v_min_f64 %0, |%1|, %2" : "=v"(am) : "v"(x), "s"(1e300));       // min(|x|, 1e300) (a NaN becomes 1e300: unused below then)
    double y = __builtin_amdgcn_rcp(am);
    const double u = fma(-am, y, 1.0);
    y = fma(fma(u, u, u), y, y);
    const double ax = fabs(x);
    const bool big = ax > 1.0;
    const double a = big ? y : ax;                                       // NaN: not big, a = NaN
    const double z = a * a;
    double p = 1.806195461861215e-05;
    p = ft_fma_vvs(p, z, -0.00019996189377901382);
    p = ft_fma_vvs(p, z, 0.0010496035084968515);
    p = ft_fma_vvs(p, z, -0.0034958859739163094);
    p = ft_fma_vvs(p, z, 0.008368931178450162);
    p = ft_fma_vvs(p, z, -0.015535152475414177);
    p = ft_fma_vvs(p, z, 0.023696731580048622);
    p = ft_fma_vvs(p, z, -0.031277189066996385);
    p = ft_fma_vvs(p, z, 0.03749486812535247);
    p = ft_fma_vvs(p, z, -0.04260356632601652);
    p = ft_fma_vvs(p, z, 0.04737749579527779);
    p = ft_fma_vvs(p, z, -0.052579733342841106);
    p = ft_fma_vvs(p, z, 0.05881506877793656);
    p = ft_fma_vvs(p, z, -0.06666564699289104);
    p = ft_fma_vvs(p, z, 0.07692298971033217);
    p = ft_fma_vvs(p, z, -0.09090908590891934);
    p = ft_fma_vvs(p, z, 0.11111111093490827);
    p = ft_fma_vvs(p, z, -0.14285714285384132);
    p = ft_fma_vvs(p, z, 0.1999999999999753);
    p = ft_fma_vvs(p, z, -0.3333333333333333);
    double r = fma(a * z, p, a);
    const double rb = (1.5707963267948966 - r) + 6.123233995736766e-17;  // pi/2 - r, pi/2 in two pieces
    r = big ? rb : r;
    return copysign(r, x);
}
// torch.remainder(x + pi, 2 pi) - pi for -pi <= x <= pi (2 atan of anything): the one case that moves is x = pi -> -pi; the same
// additions as ft_wrap's short path, bit for bit
__device__ __forceinline__ double ft_wrap_pm_pi(double x) {
    double r = x + FT_PI;
    r = r >= FT_TWO_PI ? r - FT_TWO_PI : r;
    return r - FT_PI;
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
