// MFMA variant of the coupling-layer forward / backward-wrt-x kernels.
//
// Same tiling, LDS staging and scatter-form adjoint as flow.hip, but the four big
// convolutions (conv1, conv2, conv3^T, conv2^T) run as implicit GEMMs on
// v_mfma_f64_16x16x4_f64:
//   M = 16 "pair sites" (a site column c and the two rows 2q, 2q+1 it stands for),
//   N = 16 = 8 output channels x 2 rows of the pair,
//   K = (4 x 3 input window that covers both rows) x input channels.
// Packing two output rows into N fills the 16-wide tile that 8 channels alone
// would leave half empty (75 % of the issued MACs are useful instead of 50 %).
// The weights are the B operand and stay in VGPRs for a whole stage (one double
// per lane per k-step, pre-swizzled by k_pack_weights); the A operand is one
// ds_read_b64 per MFMA straight out of the activation planes, whose strides are
// = 16 (mod 32) doubles so the four k-lanes hit disjoint banks.
// Measured on MI355X: fp64 MFMA and fp64 VALU share the DP pipe (tools/microbench
// "both"), so MFMA buys issue efficiency and register-resident weights, not flops.
#include "flow_common.h"

namespace {

using namespace fthmc;
using namespace fthmc_flow;

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int NT = 512;                 // threads per workgroup (8 waves, 2 per SIMD)
constexpr int NW = NT / 64;
// plane strides (doubles), all = 16 (mod 32)
constexpr int PS0 = 496;                // 22x22 planes (net input, padded gz2)
constexpr int PS1 = 400;                // 20x20 planes (h1, d1, padded g_out)
constexpr int PS2 = 336;                // 18x18 planes (h2, d2)
static_assert(PS0 % 32 == 16 && PS1 % 32 == 16 && PS2 % 32 == 16, "bank layout");
static_assert(PS0 >= N0 && PS1 >= N1 && PS2 >= N2, "plane size");

// LDS copy of the weights the VALU stages use (scalar loads in an LDS-heavy loop serialise on
// lgkmcnt(0)): [W3F | B3](292) [W1B](144) [B1](8) [B2](8)
constexpr int SW_W3F = 0, SW_B3 = 288, SW_W1B = 292, SW_B1 = 436, SW_B2 = 444, SW_SIZE = 452 + 12;

template <int MODE> struct SmemM {
    static constexpr bool BWD = (MODE == 1);
    static constexpr int P = 0;                          // [N0]
    static constexpr int IN = P + N0 + 12;               // [2][PS0]      (+12: keep 32-double alignment)
    static constexpr int H1 = IN + 2 * PS0;              // [8][PS1]  | bwd: padded gz2 [8][PS0] over H1|H2
    static constexpr int H2 = H1 + 8 * PS1;              // [8][PS2]
    static constexpr int ST = H2 + 8 * PS2;              // [8 channels][3][64]
    static constexpr int T2 = ST + 8 * 3 * NACT;        // [NMIX][4][64]
    static constexpr int DL = T2 + NMIX * 4 * NACT;      // [N3]
    static constexpr int SW = DL + N3;                   // [SW_SIZE] small weights read by VALU stages
    static constexpr int D1 = SW + SW_SIZE;              // [8][PS1]   (bwd)
    static constexpr int D2 = D1 + (BWD ? 8 * PS1 : 0);  // [8][PS2]
    static constexpr int GO = D2 + (BWD ? 8 * PS2 : 0);  // padded g_out [3][PS1] (20x20, ring 2)
    static constexpr int GP = GO + (BWD ? 3 * PS1 : 0);  // [N0]
    static constexpr int SIZE = GP + (BWD ? N0 : 0);
    static_assert(8 * PS1 + 8 * PS2 >= 8 * PS0, "padded gz2 must fit over h1|h2");
};

// One implicit-GEMM stage.  Output region HOUT x WOUT (HOUT even) whose input planes are one
// site larger on every side (forward conv) or ring-2 padded (transposed conv): input index of
// output (r, c) and window tap (ky4, kx) is (r + ky4, c + kx) for the pair's upper row r = 2q.
// B operand of a stage: issued before the barrier that precedes the stage so that the global
// (L2) latency overlaps the tail of the previous stage.
template <int NSTEP>
__device__ __forceinline__ void stage_prefetch(const double* __restrict__ wB, int lane, double (&breg)[NSTEP]) {
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) breg[t] = wB[t * 64 + lane];
}

template <int NSTEP, int KC, int HOUT, int WOUT, int RSA, int PSA, class Epi>
__device__ __forceinline__ void mfma_stage(const double* __restrict__ A, const double (&breg)[NSTEP],
                                           double cinit, int wave, int lane, Epi epi, long long* dbg = nullptr) {
#define DSTAMP(k) do { if (dbg && wave == 0 && lane == 0) dbg[k] = (long long)__builtin_readcyclecounter(); } while (0)
    constexpr int NPAIR = (HOUT / 2) * WOUT;
    constexpr int NTILE = (NPAIR + 15) / 16;
    const int g = lane >> 4, i = lane & 15;
    // A-operand offset of k = 4 t + g: k -> (tap = k / KC, channel = k % KC).  For KC = 8 the lane
    // part is just g * PSA and the rest is a compile-time immediate; otherwise a small table.
    int koff[KC == 8 ? 1 : NSTEP];
    if (KC != 8) {
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) {
            const int k = 4 * t + g, tap = k / KC, cK = k - tap * KC;
            koff[t] = cK * PSA + (tap / 3) * RSA + (tap % 3);
        }
    }
    const int cN = i & 7, dd = i >> 3;
    DSTAMP(11);
    for (int tile = wave; tile < NTILE; tile += NW) {
        int p = tile * 16 + i;
        if (p >= NPAIR) p = NPAIR - 1;                       // padding lanes: any valid address
        const int pr = p / WOUT, pc = p - pr * WOUT;
        const double* a0 = A + (2 * pr) * RSA + pc + (KC == 8 ? g * PSA : 0);
        // NCH independent accumulator chains: a dependent f64 MFMA waits ~4 issue slots for its
        // predecessor, so one chain per tile leaves the matrix pipe 3/4 idle at 2 waves per SIMD.
        constexpr int NCH = NSTEP >= 8 ? 4 : 3;
        double4_t accs[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) accs[ch] = double4_t{0.0, 0.0, 0.0, 0.0};
        accs[0] = double4_t{cinit, cinit, cinit, cinit};
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) {
            double av;
            if (KC == 8) av = a0[(t & 1) * 4 * PSA + ((t >> 1) / 3) * RSA + ((t >> 1) % 3)];
            else av = a0[koff[t]];
            accs[t % NCH] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, breg[t], accs[t % NCH], 0, 0, 0);
        }
        double4_t acc = accs[0];
#pragma unroll
        for (int ch = 1; ch < NCH; ++ch) acc += accs[ch];
        if (dbg && wave == 0 && lane == 0) { dbg[tile == 0 ? 12 : 14] = (long long)__builtin_readcyclecounter() + (long long)(acc[0] == 12345.678); }
        // D[row = g + 4 q][col = i]: row = pair site, col = (channel, row of the pair).
        // All four values go to the epilogue together so their activation chains interleave.
        int off4[4]; bool ok4[4]; double z4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int pp = tile * 16 + g + 4 * q;
            ok4[q] = pp < NPAIR;
            if (!ok4[q]) pp = NPAIR - 1;
            const int qr = pp / WOUT, qc = pp - qr * WOUT;
            off4[q] = (2 * qr + dd) * WOUT + qc;             // index inside the HOUT x WOUT plane
            z4[q] = acc[q];
        }
        epi(cN, off4, ok4, z4);
        if (dbg && wave == 0 && lane == 0) dbg[tile == 0 ? 13 : 15] = (long long)__builtin_readcyclecounter();
    }
}

// MODE 0 forward, 1 backward wrt x
template <int MODE>
__global__ __launch_bounds__(NT) void k_flow_mfma(FlowLayerArgs A) {
    using S = SmemM<MODE>;
    constexpr bool BWD = S::BWD;
    __shared__ __attribute__((aligned(16))) double sm[S::SIZE];
    double* sP = sm + S::P;   double* sIn = sm + S::IN;
    double* sH1 = sm + S::H1; double* sH2 = sm + S::H2;
    double* sST = sm + S::ST; double* sT2 = sm + S::T2;
    double* sDL = sm + S::DL; double* sW = sm + S::SW;
    double* sD1 = sm + S::D1; double* sD2 = sm + S::D2;
    double* sGO = sm + S::GO; double* sGP = sm + S::GP;
    double* sGZ2 = sm + S::H1;                     // padded gz2 [8][PS0] (22x22, ring 2), bwd only

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = A.L, mu = A.mu, off = A.off, act = A.act;
    const int n = L * L;
    const int b = blockIdx.z;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    const int ntiles = gridDim.x * gridDim.y;
    const int i0 = blockIdx.y * FT, j0 = blockIdx.x * FT;
    const double* __restrict__ x0 = A.x + (size_t)b * 2 * n;
    const double* __restrict__ x1 = x0 + n;
    const double* __restrict__ w = A.wint;
    long long* dbg = A.dbg ? A.dbg + ((size_t)b * ntiles + tile) * 16 : nullptr;
#define STAMP(k) do { if (dbg && tid == 0) dbg[k] = (long long)__builtin_readcyclecounter(); } while (0)
    STAMP(0);

    // ---- plaquette window + net input ------------------------------------
    for (int t = tid; t < N0; t += NT) {
        const int r = t / R0, c = t - r * R0;
        const int i = ft_modL(i0 - 3 + r, L), j = ft_modL(j0 - 3 + c, L);
        const int ip = i + 1 == L ? 0 : i + 1, jp = j + 1 == L ? 0 : j + 1;
        const double p = x0[i * L + j] - x1[i * L + j] - x0[i * L + jp] + x1[ip * L + j];
        const int sel = ft_stripe(i, j, mu, off);
        const bool frozen = (sel == 1 || sel == 2);
        sP[t] = p;
        sIn[t] = frozen ? cos(p) : 1.0;
        sIn[PS0 + t] = frozen ? sin(p) : 0.0;
        if (BWD) sGP[t] = 0.0;
    }
    if (MODE == 0) { if (tid < N3) sDL[tid] = 0.0; }
    if (BWD) { for (int t = tid; t < 3 * PS1; t += NT) sGO[t] = 0.0; }
    if (tid < 292) sW[SW_W3F + tid] = w[W3F + tid];
    else if (tid < 292 + 144) sW[SW_W1B + tid - 292] = w[W1B + tid - 292];
    else if (tid < 292 + 144 + 8) sW[SW_B1 + tid - 436] = w[B1 + tid - 436];
    else if (tid < 292 + 144 + 16) sW[SW_B2 + tid - 444] = w[B2 + tid - 444];
    double breg1[6];
    stage_prefetch<6>(w + WB1, lane, breg1);
    __syncthreads();
    STAMP(1);

    // ---- conv1 (2 -> 8) + act on the tile+2 window (20x20) ------------------
    mfma_stage<6, 2, R1, R1, R0, PS0>(sIn, breg1, 0.0, wave, lane,
        [&](int co, const int (&o)[4], const bool (&ok)[4], double (&z)[4]) {
            const double bias = sW[SW_B1 + co];
            double h[4], d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] += bias;
            act_eval4(z, act, h, d);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ok[q]) { sH1[co * PS1 + o[q]] = h[q]; if (BWD) sD1[co * PS1 + o[q]] = d[q]; }
        });
    double breg2[24];
    stage_prefetch<24>(w + WB2, lane, breg2);
    __syncthreads();
    STAMP(2);

    // ---- conv2 (8 -> 8) + act on the tile+1 window (18x18) ------------------
    mfma_stage<24, 8, R2, R2, R1, PS1>(sH1, breg2, 0.0, wave, lane,
        [&](int co, const int (&o)[4], const bool (&ok)[4], double (&z)[4]) {
            const double bias = sW[SW_B2 + co];
            double h[4], d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] += bias;
            act_eval4(z, act, h, d);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ok[q]) { sH2[co * PS2 + o[q]] = h[q]; if (BWD) sD2[co * PS2 + o[q]] = d[q]; }
        }, dbg);
    __syncthreads();
    STAMP(3);

    // ---- conv3 (8 -> 3) at the 64 active sites; one input channel per wave ---
    const int ar = mu == 0 ? (lane >> 2) : off + 4 * (lane >> 4);
    const int ac = mu == 0 ? off + 4 * (lane & 3) : (lane & 15);
    const int ai = i0 + ar, aj = j0 + ac;
    const bool avalid = (ai < L) && (aj < L);
    if (wave < 8) {
        double acc[3] = {0.0, 0.0, 0.0};
        const int ci = wave;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const double v = sH2[ci * PS2 + (ar + ky) * R2 + ac + kx];
                const double* wp = sW + SW_W3F + (ci * 9 + ky * 3 + kx) * 4;
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[k] = fma(v, wp[k], acc[k]);
            }
#pragma unroll
        for (int k = 0; k < 3; ++k) sST[(wave * 3 + k) * NACT + lane] = acc[k];
    }
    __syncthreads();
    STAMP(4);

    // ---- tan-mixture transform: wave k evaluates mixture component k -------
    double Pa = 0.0, tval = 0.0, es = 0.0, ems = 0.0, cs2 = 0.0, sn2 = 0.0, Dk = 1.0, yk = 0.0, ljk = 0.0;
    if (wave < NMIX) {
        Pa = sP[(ar + 3) * R0 + ac + 3];
        double sk = sW[SW_B3 + wave];
#pragma unroll
        for (int q = 0; q < 8; ++q) sk += sST[(q * 3 + wave) * NACT + lane];
        const double hx = Pa / 2;
        es = exp(sk); ems = exp(-sk);
        const double cs = cos(hx), sn = sin(hx);
        cs2 = cs * cs; sn2 = sn * sn;
        yk = ft_wrap(2 * atan(es * tan(hx)));
        Dk = ems * cs2 + es * sn2;
        ljk = -log(Dk);
        sT2[(wave * 4 + 0) * NACT + lane] = yk;
        sT2[(wave * 4 + 1) * NACT + lane] = ljk;
    }
    __syncthreads();
    STAMP(5);

    if (MODE == 0) {
        if (wave == 0) {
            tval = sW[SW_B3 + 2];
#pragma unroll
            for (int q = 0; q < 8; ++q) tval += sST[(q * 3 + 2) * NACT + lane];
            double ysum = 0.0, m = -INFINITY;
#pragma unroll
            for (int k = 0; k < NMIX; ++k) { ysum += sT2[(k * 4) * NACT + lane]; m = fmax(m, sT2[(k * 4 + 1) * NACT + lane]); }
            double se = 0.0;
#pragma unroll
            for (int k = 0; k < NMIX; ++k) se += exp(sT2[(k * 4 + 1) * NACT + lane] - m);
            const double lj = m + log(se) - log((double)NMIX);
            const double newP = ft_wrap(ysum / NMIX + tval);
            if (avalid) sDL[ar * FT + ac] = newP - Pa;
            const double tot = ft_wave_sum(avalid ? lj : 0.0);
            if (lane == 0 && A.logj_part) A.logj_part[(size_t)b * ntiles + tile] = tot;
        }
        __syncthreads();
        if (A.y && tid < N3) {
            const int r = tid / FT, c = tid - r * FT;
            const int i = i0 + r, j = j0 + c;
            if (i < L && j < L) {
                double v0 = x0[i * L + j], v1 = x1[i * L + j];
                if (ft_stripe(i, j, mu, off) == 0) {
                    const double d = sDL[tid];
                    if (mu == 0) v0 = ft_wrap(d + v0); else v1 = ft_wrap(-d + v1);
                }
                double* y0 = A.y + (size_t)b * 2 * n;
                y0[i * L + j] = v0; y0[n + i * L + j] = v1;
            }
        }
        STAMP(6);
        return;
    }

    if (BWD) {
        // ---- adjoint of the transform at the tile's own active sites -------
        double gdelta = 0.0, cb = 0.0;
        if (wave < NMIX) {
            cb = A.glogj ? A.glogj[b] : A.glogj_const;
            if (avalid) {
                if (A.up_link) {
                    const double gl = A.up_link[(size_t)b * 2 * n + (size_t)mu * n + ai * L + aj];
                    gdelta = mu == 0 ? gl : -gl;
                } else {
                    const double* gp = A.up_gp + (size_t)b * n;
                    const int im = ai == 0 ? L - 1 : ai - 1, jm = aj == 0 ? L - 1 : aj - 1;
                    gdelta = gp[ai * L + aj] - (mu == 0 ? gp[ai * L + jm] : gp[im * L + aj]);
                }
            }
            double m = -INFINITY;
#pragma unroll
            for (int k = 0; k < NMIX; ++k) m = fmax(m, sT2[(k * 4 + 1) * NACT + lane]);
            double se = 0.0;
#pragma unroll
            for (int k = 0; k < NMIX; ++k) se += exp(sT2[(k * 4 + 1) * NACT + lane] - m);
            const double wk = exp(ljk - m) / se;
            const double sinP = sin(Pa);
            const double elj = 1.0 / Dk;
            const double gs = avalid ? gdelta * (sinP * elj / NMIX) + cb * wk * (ems * cs2 - es * sn2) * elj : 0.0;
            const double gpk = avalid ? gdelta * (elj / NMIX) - cb * wk * sinP * 0.5 * (es - ems) * elj : 0.0;
            if (avalid) sGO[wave * PS1 + (ar + 2) * R1 + ac + 2] = gs;
            sT2[(wave * 4 + 2) * NACT + lane] = gpk;
        }
        // h1 / h2 are dead from here on: clear the padded gz2 planes that alias them
        // (conv3 above is complete: every wave passed the barrier after the sST writes)
        for (int t = tid; t < 8 * PS0; t += NT) sGZ2[t] = 0.0;
        __syncthreads();
        if (wave == 0 && avalid) {
            double gsum = -gdelta;
#pragma unroll
            for (int k = 0; k < NMIX; ++k) gsum += sT2[(k * 4 + 2) * NACT + lane];
            sGP[(ar + 3) * R0 + ac + 3] = gsum;
            sGO[NMIX * PS1 + (ar + 2) * R1 + ac + 2] = gdelta;      // dL/dt
        }
        double breg3[9];
        stage_prefetch<9>(w + WB3T, lane, breg3);
        __syncthreads();
        STAMP(6);

        // ---- conv3^T, times act'(z2) -> gz2 into the ring-2 padded 22x22 planes ----
        mfma_stage<9, 3, R2, R2, R1, PS1>(sGO, breg3, 0.0, wave, lane,
            [&](int ci, const int (&o)[4], const bool (&ok)[4], double (&gh)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (ok[q]) {
                        const int r = o[q] / R2, c = o[q] - r * R2;
                        sGZ2[ci * PS0 + (r + 2) * R0 + c + 2] = gh[q] * sD2[ci * PS2 + o[q]];
                    }
            });
        double breg4[24];
        stage_prefetch<24>(w + WB2T, lane, breg4);
        __syncthreads();
        STAMP(7);

        // ---- conv2^T, times act'(z1) -> gz1 in place over d1 -------------------------
        mfma_stage<24, 8, R1, R1, R0, PS0>(sGZ2, breg4, 0.0, wave, lane,
            [&](int ci, const int (&o)[4], const bool (&ok)[4], double (&gh)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (ok[q]) sD1[ci * PS1 + o[q]] *= gh[q];
            });
        __syncthreads();
        STAMP(8);

        // ---- conv1^T and the (cos, sin) adjoint at frozen plaquettes ----------------
        for (int t = tid; t < N0; t += NT) {
            const int r = t / R0, c = t - r * R0;
            const int i = ft_modL(i0 - 3 + r, L), j = ft_modL(j0 - 3 + c, L);
            const int sel = ft_stripe(i, j, mu, off);
            if (sel == 1 || sel == 2) {
                double gc = 0.0, gs = 0.0;
                // clamped addresses + zero mask: no per-tap branches, nothing to hoist across co
                int aoff[9]; double msk[9];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int rr = r - ky, cc = c - kx;
                        const bool ok = (rr >= 0) && (rr < R1) && (cc >= 0) && (cc < R1);
                        aoff[ky * 3 + kx] = ok ? rr * R1 + cc : 0;
                        msk[ky * 3 + kx] = ok ? 1.0 : 0.0;
                    }
#pragma unroll 1
                for (int co = 0; co < 8; ++co) {
                    const double* gz = sD1 + co * PS1;
                    const double* wp = sW + SW_W1B + co * 18;
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) {
                        const double gv = gz[aoff[tp]] * msk[tp];
                        gc = fma(gv, wp[tp * 2], gc); gs = fma(gv, wp[tp * 2 + 1], gs);
                    }
                }
                sGP[t] = -sIn[PS0 + t] * gc + sIn[t] * gs;
            }
        }
        __syncthreads();
        STAMP(9);
        double* out = A.gp_part + ((size_t)b * ntiles + tile) * N0;
        for (int t = tid; t < N0; t += NT) out[t] = sGP[t];
        STAMP(10);
    }
}

inline dim3 flow_grid(int B, int L) { int t = (L + FT - 1) / FT; return dim3(t, t, B); }
int g_variant = 1;

}  // namespace

namespace fthmc {

void set_flow_variant(int v) { g_variant = v; }
int get_flow_variant() { return g_variant; }

int launch_flow_fwd_mfma(const FlowLayerArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_flow_mfma<0>, flow_grid(a.B, a.L), dim3(NT), 0, s, a);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_flow_bwd_mfma(const FlowLayerArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_flow_mfma<1>, flow_grid(a.B, a.L), dim3(NT), 0, s, a);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}

}  // namespace fthmc
