// Shared constants / helpers of the coupling-layer kernels (flow.hip: VALU variant,
// flow_fwd.hip: MFMA variant).
#pragma once
#include "common.h"
#include "kernels.h"

namespace fthmc_flow {

using namespace fthmc;

constexpr int FT = FLOW_TILE;
constexpr int R0 = FT + 6, R1 = FT + 4, R2 = FT + 2;
constexpr int N0 = R0 * R0, N1 = R1 * R1, N2 = R2 * R2, N3 = FT * FT;
constexpr int NACT = N3 / 4;                  // active sites per tile (64 = one wave)
constexpr int NMIX = 2;

// canonical per-layer offsets (PyTorch [Cout][Cin][3][3])
constexpr int CW0 = 0, CB0 = 144, CW1 = 152, CB1 = 728, CW2 = 736, CB2 = 952;
// kernel layout offsets
constexpr int W1F = 0;      // [ci 2][tap 9][co 8]
constexpr int B1 = 144;     // [8]
constexpr int W2F = 152;    // [ci 8][tap 9][co 8]
constexpr int B2 = 728;     // [8]
constexpr int W3F = 736;    // [ci 8][tap 9][co 4] (co 3 = 0)
constexpr int B3 = 1024;    // [4]
constexpr int W3B = 1028;   // [co 3][tap 9][ci 8]
constexpr int W2B = 1244;   // [co 8][tap 9][ci 8]
constexpr int W1B = 1820;   // [co 8][tap 9][ci 2]
static_assert(W1B + 144 <= FLOW_WINT, "weight layout");

// verbatim copy of the layer's canonical weights (955 doubles, PyTorch order) + zeros: the MFMA
// kernels keep it in LDS and read both conv operands straight from it
constexpr int WCAN = 1968;            // [960]
constexpr int WCAN_SIZE = 960;
constexpr int WZERO = CB2 + 4;        // a 0.0 inside the canonical copy (index 956)
static_assert(WCAN + WCAN_SIZE <= FLOW_WINT, "weight layout");

// Weight blocks of the MFMA kernels, copied verbatim into LDS.  The 3x3 tables of the implicit-GEMM stages are
// padded with a zero line on either side along the direction in which two output sites are paired, so the
// weight of (window tap, pair member dd) is table[line + 1 - dd]: every operand address of an MFMA is then
// (a lane part) + (a compile-time constant of the K step) and costs no VALU instruction inside the stage.
// Table order: [3 = tap along the pair-free direction][g 4][h 2][5 = padded line][cN 8], input channel (K) = 4 h + g,
// output ROW cN of the MFMA = output channel ft_chan(cN): the 32 lanes of a half-wave (cN, two values of g, dd) then
// read 32 different LDS banks.  ft_chan: a lane holds MFMA rows g and g + 4; with row r carrying channel
// 2 (r & 3) + (r >> 2) those are the ADJACENT channels 2 g, 2 g + 1, which the channel-minor activation stash
// (flow_mfma_common.h: struct Stash) stores and loads as one 16-byte access.
//   forward block (one per mu):  b0[8] b1[8] b2[3]+0  P2[960]  |  P1[3][2 ci: stride 48][5 padded lines: stride 8][8 co]  BC[4 s][2 dd][8 co]  |  w2[3][8][9]
//       conv1 pairs its output sites ACROSS the stripe lines (flow_fwd.hip), conv2 along them:
//       mu = 0 (conv1 pairs = columns):  P1[ky][ci][c5][co] = w0[co][ci][ky][c5 - 1]
//       mu = 1 (conv1 pairs = rows):     P1[kx][ci][r5][co] = w0[co][ci][r5 - 1][kx]
//       BC[s][dd][co] = b0[co] + the contribution of the constant net input (cos, sin) = (1, 0) on the two non-frozen
//       lines of the pair's four-line window, s = stripe class of the window's first line, dd = site of the pair
//       mu = 0 (pairs = rows r, r + 1):  P2[kx][g][h][r5][co] = w1[co][4 h + g][r5 - 1][kx]
//       mu = 1 (pairs = columns):        P2[ky][g][h][c5][co] = w1[co][4 h + g][ky][c5 - 1]
//   backward block (one per mu; conv2^T pairs its output sites ACROSS the stripe lines, see flow_bwd_gather.hip):
//       w0[8][2][9]  w2[3][8][9]  T2[960]
//       mu = 1 (pairs = rows):     T2[kx][g][h][r5][ci] = w1[4 h + g][ci][2 - (r5 - 1)][2 - kx]
//       mu = 0 (pairs = columns):  T2[ky][g][h][c5][ci] = w1[4 h + g][ci][2 - ky][2 - (c5 - 1)]
__host__ __device__ constexpr int ft_chan(int row) { return 2 * (row & 3) + (row >> 2); }
constexpr int LF_B0 = 0, LF_B1 = 8, LF_B2 = 16, LF_P2 = 20, LF_SIZE = LF_P2 + 960;   // resident part of the LDS copy
constexpr int LF_P1 = LF_SIZE, LF_BC = 288, LF_P1_SIZE = LF_BC + 64, LF_LDS = LF_P1 + LF_P1_SIZE;   // conv1 tables: conv1 stage only
constexpr int LF_W2 = LF_LDS, LF_BLOCK = LF_W2 + 216;   // conv3's weights: never copied to LDS (wave-uniform: scalar loads into SGPRs)
constexpr int LB_W0 = 0, LB_W2 = 144, LB_T2 = 360, LB_SIZE = LB_T2 + 960;
constexpr int WFWD0 = 2944, WFWD1 = WFWD0 + LF_BLOCK, WBWD = WFWD1 + LF_BLOCK, WBWD1 = WBWD + LB_SIZE;   // WBWD: rows, WBWD1: columns
// conv3^T on the matrix cores (flow_mfma_common.h conv3t_mfma; the tiled backward kernels): the upstream gradient g_out lives on
// the ACTIVE stripe lines only, so -- as for conv1 in the forward -- output sites are paired ACROSS the lines: a pair's window is
// four consecutive lines of which exactly one is active, K = that line's 3 taps along it x 3 channels (padded to 4) = 3 MFMA
// steps.  Table (one per pairing direction): T3[t 3: tap along the line][co 4: stride 48][l5 5: stride 8][cN 8] =
// w2[co][ft_chan(cN)][tap across = l5 - 1][tap along = t] (zero for l5 = 0, 4 and co = 3): the weight of (active window line u,
// pair member dd) is the entry l5 = dd + 3 - u, so the operand address is (lane part) + t * 192 -- no VALU in the K loop.
constexpr int LT3_CO = 48, LT3_T = 4 * LT3_CO, LT3_SIZE = 3 * LT3_T;            // 576
constexpr int WT3R = WBWD1 + LB_SIZE, WT3C = WT3R + LT3_SIZE;                    // pairs = rows (mu = 1) / columns (mu = 0)
static_assert(WT3C + LT3_SIZE <= FLOW_WSTAMP0, "weight layout: the stamps of k_pack_weights sit behind the last block");

// exp(x) for either sign (|x| clamped to the finite range): range reduction by ln2 (hi/lo split) + degree-13
// Taylor (|r| <= ln2/2: truncation 4e-18) + v_ldexp.  ~20 dependent DP ops instead of ocml exp's ~60, < 1.5 ulp.
__device__ __forceinline__ double ft_exp(double x) {
    x = fmin(fmax(x, -745.0), 709.0);
    const double n = rint(ft_mul_vs(x, 1.4426950408889634074));
    double r = ft_fma_nvsv(n, 6.93147180369123816490e-01, x);
    r = ft_fma_nvsv(n, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = ft_fma_vvs(p, r, 2.0876756987868100e-09);
    p = ft_fma_vvs(p, r, 2.5052108385441720e-08);
    p = ft_fma_vvs(p, r, 2.7557319223985888e-07);
    p = ft_fma_vvs(p, r, 2.7557319223985893e-06);
    p = ft_fma_vvs(p, r, 2.4801587301587302e-05);
    p = ft_fma_vvs(p, r, 1.9841269841269841e-04);
    p = ft_fma_vvs(p, r, 1.3888888888888889e-03);
    p = ft_fma_vvs(p, r, 8.3333333333333332e-03);
    p = ft_fma_vvs(p, r, 4.1666666666666664e-02);
    p = ft_fma_vvs(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// 1 / t for t well inside the normal range: v_rcp_f64 + one third-order correction (see ft_sigmoid); 4 operations
// instead of the ~18 of an IEEE division
__device__ __forceinline__ double ft_rcp(double t) {
    const double y = __builtin_amdgcn_rcp(t);
    const double u = fma(-t, y, 1.0);
    return fma(fma(u, u, u), y, y);
}

// N independent ft_exp, written step-interleaved (the compiler keeps the source order of independent
// instructions, and one exp is a ~22-deep dependent DP chain).  Same arithmetic as ft_exp.
template <int N>
__device__ __forceinline__ void ft_expN(const double (&xin)[N], double (&e)[N]) {
    double x[N], n[N], r[N], p[N];
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] = fmin(fmax(xin[q], -745.0), 709.0);
#pragma unroll
    for (int q = 0; q < N; ++q) n[q] = rint(x[q] * 1.4426950408889634074);
#pragma unroll
    for (int q = 0; q < N; ++q) r[q] = fma(-n[q], 6.93147180369123816490e-01, x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) r[q] = fma(-n[q], 1.90821492927058770002e-10, r[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) p[q] = fma(1.6059043836821613e-10, r[q], 2.0876756987868100e-09);
    constexpr double C[12] = {2.5052108385441720e-08, 2.7557319223985888e-07, 2.7557319223985893e-06,
                              2.4801587301587302e-05, 1.9841269841269841e-04, 1.3888888888888889e-03,
                              8.3333333333333332e-03, 4.1666666666666664e-02, 1.6666666666666666e-01,
                              0.5, 1.0, 1.0};
#pragma unroll
    for (int c = 0; c < 12; ++c)
#pragma unroll
        for (int q = 0; q < N; ++q) p[q] = fma(p[q], r[q], C[c]);
#pragma unroll
    for (int q = 0; q < N; ++q) e[q] = ldexp(p[q], (int)n[q]);
}

// exp(r) on |r| <= ln2/2: degree-11 polynomial with c0 = c1 = 1 pinned (near-minimax: Chebyshev-node fit of
// (exp(r) - 1 - r) / r^2, tools/minimax_exp.py): relative error 1.6e-17 in exact arithmetic (degree-13 Taylor: 4e-18,
// two FMAs more per value).
#define FT_EXP11(p, r)                                   \
    p = 2.5100375832561234e-08;                          \
    p = ft_fma_vvs(p, r, 2.7620075879983367e-07);        \
    p = ft_fma_vvs(p, r, 2.7557268480310024e-06);        \
    p = ft_fma_vvs(p, r, 2.4801521322368692e-05);        \
    p = ft_fma_vvs(p, r, 0.00019841269863040545);        \
    p = ft_fma_vvs(p, r, 0.0013888888917196719);         \
    p = ft_fma_vvs(p, r, 0.008333333333330065);          \
    p = ft_fma_vvs(p, r, 0.041666666666624164);          \
    p = ft_fma_vvs(p, r, 0.16666666666666669);           \
    p = ft_fma_vvs(p, r, 0.5000000000000001);            \
    p = fma(p, r, 1.0);                                  \
    p = fma(p, r, 1.0)

// sigmoid(z) = 1 / (1 + e), e = exp(-z), for either sign of z with ONE code path: -z is clamped from above only
// (e <= exp(700) stays finite, so 1 / (1 + e) is the correctly scaled tiny number; a large positive z underflows e to
// 0 through v_ldexp).  Reciprocal: v_rcp_f64 (4.6e-8 on this chip, tools/rcp_check.hip) + ONE third-order step
// y (1 + u + u^2), u = 1 - t y (u^3 ~ 1e-22).  26 DP operations per value together with h and act' below (was 35).
// min(-z, 700) as ONE v_min_f64 with a source negation (fmin() compiles to a canonicalising v_max_f64 + v_min_f64)
__device__ __forceinline__ double ft_min_neg(double z, double hi) {
    double a;
    asm("v_min_f64 %0, -%1, %2" : "=v"(a) : "v"(z), "s"(hi));
    return a;
}

// FT_SIG_RATIONAL (default 1): exp(r) in fdlibm's rational form folded into the sigmoid's own division,
//     c = r - r^2 P(r^2) (degree 4 in r^2),  exp(r) = (m + 2 r) / m  with  m = 2 - c,
//     sigmoid = m / (m + 2^n (m + 2 r)):
// 24 DP operations per value with h and act' instead of 26 (six polynomial steps instead of eleven, one multiply more at
// the end).  Against quad precision over |z| <= 40 (tools/sigmoid_check.c): max 2.8 ulp / mean 0.50 (polynomial form: 2.3 /
// 0.42).  0: the degree-11 polynomial of exp(r).
#ifndef FT_SIG_RATIONAL
#define FT_SIG_RATIONAL 1
#endif

// Range reduction n = round(a log2 e) without v_rndne_f64 / v_cvt_i32_f64 (FT_SIG_MAGIC, default 1): t = a log2 e + 1.5 * 2^52
// is rounded to an integer by the FMA itself, n = t - 1.5 * 2^52 is exact, and the low dword of t IS n in two's complement
// (2^52 + 2^51 = 0 mod 2^32), which v_ldexp_f64 takes as its exponent operand: {mul, rndne, cvt, ldexp} -> {fma, add, ldexp},
// one fp64-pipe slot less per value.  Valid for |a log2 e| < 2^31 (a <= 700 by the clamp; z < 1.4e9 on the other side).
#ifndef FT_SIG_MAGIC
#define FT_SIG_MAGIC 1
#endif
#define FT_MAGIC52 6755399441055744.0
// r = a - n ln 2 in ONE FMA with the correctly rounded ln 2 (FT_SIG_LN2_ONE, default 1) instead of the hi / lo pair: the
// constant's rounding error (2.3e-17) enters r as |n| 2.3e-17, i.e. sigma picks up a RELATIVE error of |n| 2.3e-17 where it is
// tiny (z << 0) and an ABSOLUTE error sigma (1 - sigma) |n| 2.3e-17 <= 1.3e-17 everywhere (|n| ~ 1.44 |z| grows as
// sigma (1 - sigma) ~ e^-|z| falls): an eighth of the rounding error of a sigma near 1/2.  One fp64-pipe slot less per value.
#ifndef FT_SIG_LN2_ONE
#define FT_SIG_LN2_ONE 1
#endif
#define FT_LN2 6.93147180559945309417e-01
__device__ __forceinline__ double ft_round_magic(double a, int& ni) {
    double t;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t) : "v"(a), "s"(1.4426950408889634074), "v"(FT_MAGIC52));   // one SGPR operand per VOP3 on gfx9
    ni = (int)__double2loint(t);
    return t - FT_MAGIC52;
}

__device__ __forceinline__ double ft_sigmoid(double z) {
    const double a = ft_min_neg(z, 700.0);
#if FT_SIG_MAGIC
    int ni;
    const double n = ft_round_magic(a, ni);
#else
    const double n = rint(ft_mul_vs(a, 1.4426950408889634074));
    const int ni = (int)n;
#endif
#if FT_SIG_LN2_ONE
    const double r = ft_fma_nvsv(n, FT_LN2, a);
#else
    double r = ft_fma_nvsv(n, 6.93147180369123816490e-01, a);
    r = ft_fma_nvsv(n, 1.90821492927058770002e-10, r);
#endif
#if FT_SIG_RATIONAL
    const double s = r * r;
    double P = 4.13813679705723846039e-08;
    P = ft_fma_vvs(P, s, -1.65339022054652515390e-06);
    P = ft_fma_vvs(P, s, 6.61375632143793436117e-05);
    P = ft_fma_vvs(P, s, -2.77777777770155933842e-03);
    P = ft_fma_vvs(P, s, 1.66666666666666019037e-01);
    const double m = 2.0 - fma(-s, P, r);
    const double t = m + ldexp(fma(2.0, r, m), ni);
    double y = __builtin_amdgcn_rcp(t);
    const double u = fma(-t, y, 1.0);
    return m * fma(fma(u, u, u), y, y);
#else
    double p;
    FT_EXP11(p, r);
    const double t = 1.0 + ldexp(p, ni);
    double y = __builtin_amdgcn_rcp(t);
    const double u = fma(-t, y, 1.0);
    return fma(fma(u, u, u), y, y);
#endif
}

__device__ __forceinline__ void act_eval(double z, int act, double& h, double& d) {
    if (act == FTHMC_ACT_SILU) {
        const double sg = ft_sigmoid(z);
        h = z * sg;
        d = fma(h, 1.0 - sg, sg);                          // silu' = sg (1 + z (1 - sg))
    } else if (act == FTHMC_ACT_RELU) {
        h = z > 0.0 ? z : 0.0;  d = z > 0.0 ? 1.0 : 0.0;
    } else {
        h = z > 0.0 ? z : 0.01 * z;  d = z > 0.0 ? 1.0 : 0.01;
    }
}

// Four independent activations at once, written step-interleaved: the compiler keeps the
// source order of independent instructions, and a single sigmoid is a ~25-deep dependent DP
// chain (each link waits ~4 issue slots), so four chains side by side run ~3x faster than
// four sigmoids back to back.  Same arithmetic as ft_sigmoid.
__device__ __forceinline__ void sigmoid4(const double (&z)[4], double (&sg)[4]) {
    double a[4], n[4], r[4], p[4], t[4], y[4], u[4];
    int ni[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = ft_min_neg(z[q], 700.0);
#if FT_SIG_MAGIC
#pragma unroll
    for (int q = 0; q < 4; ++q) n[q] = ft_round_magic(a[q], ni[q]);
#else
#pragma unroll
    for (int q = 0; q < 4; ++q) { n[q] = rint(ft_mul_vs(a[q], 1.4426950408889634074)); ni[q] = (int)n[q]; }
#endif
#if FT_SIG_LN2_ONE
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = ft_fma_nvsv(n[q], FT_LN2, a[q]);
#else
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = ft_fma_nvsv(n[q], 6.93147180369123816490e-01, a[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = ft_fma_nvsv(n[q], 1.90821492927058770002e-10, r[q]);
#endif
#if FT_SIG_RATIONAL
    double s[4], m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = r[q] * r[q];
    {
        const double c5 = 4.13813679705723846039e-08;                  // one VGPR pair for the leading coefficient, shared by the four chains
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = ft_fma_vvs(c5, s[q], -1.65339022054652515390e-06);
    }
    constexpr double C[3] = {6.61375632143793436117e-05, -2.77777777770155933842e-03, 1.66666666666666019037e-01};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = ft_fma_vvs(p[q], s[q], C[c]);
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = fma(-s[q], p[q], r[q]);          // c
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = 2.0 - p[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = fma(2.0, r[q], m[q]);            // m exp(r)
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = m[q] + ldexp(p[q], ni[q]);
#else
    {
        const double c11 = 2.5100375832561234e-08;                     // one VGPR pair for the leading coefficient, shared by the four chains
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = ft_fma_vvs(c11, r[q], 2.7620075879983367e-07);
    }
    constexpr double C[8] = {2.7557268480310024e-06, 2.4801521322368692e-05, 0.00019841269863040545,
                             0.0013888888917196719, 0.008333333333330065, 0.041666666666624164,
                             0.16666666666666669, 0.5000000000000001};
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = ft_fma_vvs(p[q], r[q], C[c]);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = fma(p[q], r[q], 1.0);
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = 1.0 + ldexp(p[q], ni[q]);
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) y[q] = __builtin_amdgcn_rcp(t[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) u[q] = fma(-t[q], y[q], 1.0);
#pragma unroll
    for (int q = 0; q < 4; ++q) u[q] = fma(u[q], u[q], u[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) sg[q] = fma(u[q], y[q], y[q]);
#if FT_SIG_RATIONAL
#pragma unroll
    for (int q = 0; q < 4; ++q) sg[q] *= m[q];
#endif
}

__device__ __forceinline__ void act_eval4(const double (&z)[4], int act, double (&h)[4], double (&d)[4]) {
    if (act == FTHMC_ACT_SILU) {
        double sg[4];
        sigmoid4(z, sg);
#pragma unroll
        for (int q = 0; q < 4; ++q) { h[q] = z[q] * sg[q]; d[q] = fma(h[q], 1.0 - sg[q], sg[q]); }
    } else if (act == FTHMC_ACT_RELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { h[q] = z[q] > 0.0 ? z[q] : 0.0; d[q] = z[q] > 0.0 ? 1.0 : 0.0; }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { h[q] = z[q] > 0.0 ? z[q] : 0.01 * z[q]; d[q] = z[q] > 0.0 ? 1.0 : 0.01; }
    }
}

}  // namespace fthmc_flow
