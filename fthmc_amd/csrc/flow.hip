// Gauge-equivariant coupling layer for 2D U(1) (fthmc/utils/layers.py:188-396)
// as one fused gfx950 kernel per layer and direction.
//
// A workgroup owns a 16x16 tile of one chain.  Everything between the link
// field and the link update lives in LDS: the plaquette window (tile + 3 halo),
// the (cos, sin) net input, both hidden activations (tile + 2 / + 1 halo) and
// the (s, t) output at the tile's active stripe.  The backward pass re-runs the
// forward in LDS and then walks the adjoint in *scatter* form: only the tile's
// own active sites seed the gradient, so no forward halo has to be widened; the
// resulting partial plaquette-gradient window (22x22) is written out and summed
// across tiles in a fixed order by k_gather_gp (deterministic, no atomics).
//
// Convs run on the fp64 VALU with weights streamed through SGPRs (wave-uniform
// addresses => s_load): each lane owns one site and half of the output channels.
#include "flow_common.h"

namespace {

using namespace fthmc;
using namespace fthmc_flow;

// token != 0: the call states a weight version.  A workgroup whose stamp (kernels.h: FLOW_WSTAMPS) already carries this call's
// token for its layer leaves at once -- the slice behind the stamp is the expansion of exactly these weights --, any other
// workgroup expands its slice and stamps it.  token = 0: expand, and clear the stamp.
__global__ void k_pack_weights(const double* __restrict__ w, int n_layers, double* __restrict__ o, unsigned long long token) {
    const int l = blockIdx.x;                                    // gridDim.y = FLOW_WSTAMPS workgroups per layer
    const double* c = w + (size_t)l * FTHMC_W_PER_LAYER;
    double* d = o + (size_t)l * FLOW_WINT;
    unsigned long long* stamp = reinterpret_cast<unsigned long long*>(d + FLOW_WSTAMP0) + blockIdx.y;
    const bool guarded = l < FLOW_WHEAD_LAYERS;                  // layers beyond the workspace head: always expanded, never stamped
    unsigned long long mine = 0ull;
    if (token != 0ull && guarded) {
        mine = (token + 0x9E3779B97F4A7C15ull * (unsigned long long)(l + 1)) | 1ull;
        const unsigned long long seen = *stamp;
        __syncthreads();                                         // every thread has read the stamp before thread 0 may rewrite it
        if (seen == mine) return;
    }
    for (int t = blockIdx.y * blockDim.x + threadIdx.x; t < FLOW_WSTAMP0; t += gridDim.y * blockDim.x) {
        double v = 0.0;
        if (t < B1) { int co = t % 8, tap = (t / 8) % 9, ci = t / 72; v = c[CW0 + (co * 2 + ci) * 9 + tap]; }
        else if (t < W2F) v = c[CB0 + t - B1];
        else if (t < B2) { int u = t - W2F; int co = u % 8, tap = (u / 8) % 9, ci = u / 72; v = c[CW1 + (co * 8 + ci) * 9 + tap]; }
        else if (t < W3F) v = c[CB1 + t - B2];
        else if (t < B3) { int u = t - W3F; int co = u % 4, tap = (u / 4) % 9, ci = u / 36; v = co < 3 ? c[CW2 + (co * 8 + ci) * 9 + tap] : 0.0; }
        else if (t < W3B) { int co = t - B3; v = co < 3 ? c[CB2 + co] : 0.0; }
        else if (t < W2B) { int u = t - W3B; int ci = u % 8, tap = (u / 8) % 9, co = u / 72; v = c[CW2 + (co * 8 + ci) * 9 + tap]; }
        else if (t < W1B) { int u = t - W2B; int ci = u % 8, tap = (u / 8) % 9, co = u / 72; v = c[CW1 + (co * 8 + ci) * 9 + tap]; }
        else if (t < W1B + 144) { int u = t - W1B; int ci = u % 2, tap = (u / 2) % 9, co = u / 18; v = c[CW0 + (co * 2 + ci) * 9 + tap]; }
        else if (t >= WCAN && t < WCAN + WCAN_SIZE) { const int u = t - WCAN; v = u < FTHMC_W_PER_LAYER ? c[u] : 0.0; }
        else if (t >= WFWD0 && t < WFWD1 + LF_BLOCK) {                   // forward blocks of the MFMA kernels (flow_common.h)
            const int col = t >= WFWD1, u = t - (col ? WFWD1 : WFWD0);
            if (u < LF_B1) v = c[CB0 + u];
            else if (u < LF_B2) v = c[CB1 + u - LF_B1];
            else if (u < LF_P2) v = u - LF_B2 < 3 ? c[CB2 + u - LF_B2] : 0.0;
            else if (u >= LF_W2) v = c[CW2 + u - LF_W2];
            else if (u < LF_P1) {
                const int e = u - LF_P2, a3 = e / 320, ci = 4 * ((e / 40) % 2) + (e / 80) % 4, l5 = (e / 8) % 5 - 1, co = ft_chan(e % 8);
                const int ky = col ? a3 : l5, kx = col ? l5 : a3;
                v = (l5 >= 0 && l5 <= 2) ? c[CW1 + (co * 8 + ci) * 9 + ky * 3 + kx] : 0.0;
            } else if (u < LF_P1 + LF_BC) {                              // conv1 pairs across the lines: columns for mu = 0
                const int e = u - LF_P1, a3 = e / 96, ci = (e / 48) % 2, l5 = (e / 8) % 6 - 1, co = ft_chan(e % 8);
                const int ky = col ? l5 : a3, kx = col ? a3 : l5;
                v = (l5 >= 0 && l5 <= 2) ? c[CW0 + (co * 2 + ci) * 9 + ky * 3 + kx] : 0.0;
            } else {
                // bias + the constant part of conv1: window lines k = 0..3 of stripe class (s + k) & 3, frozen = classes 1, 2;
                // a non-frozen line carries (cos, sin) = (1, 0): site dd sees it through tap line k - dd, all three taps along
                const int e = u - LF_P1 - LF_BC, s4 = e / 16, dd = (e / 8) % 2, co = e % 8;
                v = c[CB0 + co];
                for (int k = 0; k < 4; ++k) {
                    const int cls = (s4 + k) & 3, tl = k - dd;
                    if (cls == 1 || cls == 2 || tl < 0 || tl > 2) continue;
                    for (int a = 0; a < 3; ++a) v += c[CW0 + (co * 2 + 0) * 9 + (col ? tl * 3 + a : a * 3 + tl)];
                }
            }
        } else if (t >= WBWD && t < WBWD1 + LB_SIZE) {                     // backward blocks (rows, columns)
            const int col = t >= WBWD1, u = t - (col ? WBWD1 : WBWD);
            if (u < LB_W2) v = c[CW0 + u];
            else if (u < LB_T2) v = c[CW2 + u - LB_W2];
            else {
                const int e = u - LB_T2, a3 = e / 320, co = 4 * ((e / 40) % 2) + (e / 80) % 4, l5 = (e / 8) % 5 - 1, ci = ft_chan(e % 8);
                const int ky = col ? a3 : l5, kx = col ? l5 : a3;
                v = (l5 >= 0 && l5 <= 2) ? c[CW1 + (co * 8 + ci) * 9 + (2 - ky) * 3 + (2 - kx)] : 0.0;
            }
        } else if (t >= WT3R && t < WT3C + LT3_SIZE) {                    // conv3^T tables (flow_common.h): pairs = rows, pairs = columns
            const int col = t >= WT3C, u = t - (col ? WT3C : WT3R);
            const int tt = u / LT3_T, co = (u % LT3_T) / LT3_CO, e = u % LT3_CO, l5 = e / 8 - 1, ci = ft_chan(e % 8);
            // col (mu = 0): across the lines = kx, along = ky;  rows (mu = 1): across = ky, along = kx
            const int ky = col ? tt : l5, kx = col ? l5 : tt;
            v = (co < 3 && e < 40 && l5 >= 0 && l5 <= 2) ? c[CW2 + (co * 8 + ci) * 9 + ky * 3 + kx] : 0.0;
        }
        d[t] = v;
    }
    if (guarded) {
        __threadfence();
        __syncthreads();                                         // the slice is written before its stamp says so
        if (threadIdx.x == 0) *stamp = mine;
    }
}


// LDS plan (doubles)
template <int MODE> struct Smem {
    static constexpr bool BWD = (MODE == 1 || MODE == 2);
    static constexpr int P = 0;                       // [N0] plaquette window
    static constexpr int IN = P + N0;                 // [2][N0] cos, sin
    static constexpr int H1 = IN + 2 * N0;            // [8][N1]
    static constexpr int H2 = H1 + 8 * N1;            // [8][N2]
    static constexpr int ST = H2 + 8 * N2;            // [4 waves][3][64] conv3 partials
    static constexpr int T2 = ST + 4 * 3 * NACT;      // [NMIX][4][64] y, lj, (part gP, spare)
    static constexpr int DL = T2 + NMIX * 4 * NACT;   // [N3] delta at tile sites
    static constexpr int RED = DL + N3;               // [16]
    static constexpr int D1 = RED + 16;               // [8][N1] act' -> gz1      (bwd)
    static constexpr int D2 = D1 + (BWD ? 8 * N1 : 0);   // [8][N2] act' -> gz2
    static constexpr int GO = D2 + (BWD ? 8 * N2 : 0);   // [3][N3] g(s0, s1, t)
    static constexpr int GP = GO + (BWD ? 3 * N3 : 0);   // [N0] partial plaquette gradient
    static constexpr int SIZE = GP + (BWD ? N0 : 0);
};

// MODE 0 forward, 1 backward wrt x, 2 backward wrt x and weights, 3 reverse
template <int MODE>
__global__ __launch_bounds__(256) void k_flow_layer(FlowLayerArgs A) {
    using S = Smem<MODE>;
    constexpr bool BWD = S::BWD;
    __shared__ double sm[S::SIZE];
    double* sP = sm + S::P;   double* sIn = sm + S::IN;
    double* sH1 = sm + S::H1; double* sH2 = sm + S::H2;
    double* sST = sm + S::ST; double* sT2 = sm + S::T2;
    double* sDL = sm + S::DL;
    double* sD1 = sm + S::D1; double* sD2 = sm + S::D2;
    double* sGO = sm + S::GO; double* sGP = sm + S::GP;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave & 1;                       // which 4 output channels this wave owns
    const int lane128 = lane | ((wave >> 1) << 6);   // site slot among the 2 waves of a half
    const int L = A.L, mu = A.mu, off = A.off, act = A.act;
    const int n = L * L;
    const int b = blockIdx.z;
    const int ntj = gridDim.x;
    const int tile = blockIdx.y * ntj + blockIdx.x;
    const int ntiles = gridDim.x * gridDim.y;
    const int i0 = blockIdx.y * FT, j0 = blockIdx.x * FT;
    const double* __restrict__ x0 = A.x + (size_t)b * 2 * n;
    const double* __restrict__ x1 = x0 + n;
    const double* __restrict__ w = A.wint;

    // ---- plaquette window + net input ------------------------------------
    for (int t = tid; t < N0; t += 256) {
        const int r = t / R0, c = t - r * R0;
        const int i = ft_modL(i0 - 3 + r, L), j = ft_modL(j0 - 3 + c, L);
        const int ip = i + 1 == L ? 0 : i + 1, jp = j + 1 == L ? 0 : j + 1;
        const double p = x0[i * L + j] - x1[i * L + j] - x0[i * L + jp] + x1[ip * L + j];
        const int sel = ft_stripe(i, j, mu, off);
        const bool frozen = (sel == 1 || sel == 2);
        sP[t] = p;
        sIn[t] = frozen ? cos(p) : 1.0;
        sIn[N0 + t] = frozen ? sin(p) : 0.0;
        if (BWD) sGP[t] = 0.0;
    }
    if (MODE == 0 || MODE == 3) { if (tid < N3) sDL[tid] = 0.0; }
    if (BWD) { for (int t = tid; t < 3 * N3; t += 256) sGO[t] = 0.0; }
    __syncthreads();

    // ---- conv1 (2 -> 8) + act on the tile+2 window ------------------------
    for (int s = lane128; s < N1; s += 128) {
        const int r = s / R1, c = s - r * R1;
        double acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = w[B1 + half * 4 + k];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const double v = sIn[ci * N0 + (r + ky) * R0 + c + kx];
                    const double* wp = w + W1F + ((ci * 9 + ky * 3 + kx) * 8 + half * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = fma(v, wp[k], acc[k]);
                }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double h, d; act_eval(acc[k], act, h, d);
            sH1[(half * 4 + k) * N1 + s] = h;
            if (BWD) sD1[(half * 4 + k) * N1 + s] = d;
        }
    }
    __syncthreads();

    // ---- conv2 (8 -> 8) + act on the tile+1 window -------------------------
    for (int s = lane128; s < N2; s += 128) {
        const int r = s / R2, c = s - r * R2;
        double acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = w[B2 + half * 4 + k];
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const double v = sH1[ci * N1 + (r + ky) * R1 + c + kx];
                    const double* wp = w + W2F + ((ci * 9 + ky * 3 + kx) * 8 + half * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = fma(v, wp[k], acc[k]);
                }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double h, d; act_eval(acc[k], act, h, d);
            sH2[(half * 4 + k) * N2 + s] = h;
            if (BWD) sD2[(half * 4 + k) * N2 + s] = d;
        }
    }
    __syncthreads();

    // ---- conv3 (8 -> 3) at the 64 active sites; K split over the 4 waves ---
    // active site `lane`: mu=0 columns off+4m, mu=1 rows off+4m (tile origin % 4 == 0)
    const int ar = mu == 0 ? (lane >> 2) : off + 4 * (lane >> 4);
    const int ac = mu == 0 ? off + 4 * (lane & 3) : (lane & 15);
    const int ai = i0 + ar, aj = j0 + ac;
    const bool avalid = (ai < L) && (aj < L);
    {
        double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int ci = wave * 2 + cc;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const double v = sH2[ci * N2 + (ar + ky) * R2 + ac + kx];
                    const double* wp = w + W3F + (ci * 9 + ky * 3 + kx) * 4;
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc[k] = fma(v, wp[k], acc[k]);
                }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) sST[(wave * 3 + k) * NACT + lane] = acc[k];
    }
    __syncthreads();

    // ---- tan-mixture transform: wave k evaluates mixture component k -------
    // registers that survive the barriers below (same lane = same active site)
    double Pa = 0.0, tval = 0.0, es = 0.0, ems = 0.0, cs2 = 0.0, sn2 = 0.0, Dk = 1.0, yk = 0.0, ljk = 0.0;
    if (MODE != 3 && wave < NMIX) {
        Pa = sP[(ar + 3) * R0 + ac + 3];
        double sk = w[B3 + wave];
#pragma unroll
        for (int q = 0; q < 4; ++q) sk += sST[(q * 3 + wave) * NACT + lane];
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
    if (MODE != 3) __syncthreads();

    if (MODE == 0) {
        if (wave == 0) {
            tval = w[B3 + 2];
#pragma unroll
            for (int q = 0; q < 4; ++q) tval += sST[(q * 3 + 2) * NACT + lane];
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
    }

    if (MODE == 3) {
        // reverse: solve mean_k y_k(x) = wrap(P' - t) per active site (layers.py:373-396)
        if (wave == 0) {
            const double Pn = sP[(ar + 3) * R0 + ac + 3];
            double sk[NMIX], ek[NMIX], emk[NMIX];
#pragma unroll
            for (int k = 0; k < NMIX; ++k) {
                sk[k] = w[B3 + k];
#pragma unroll
                for (int q = 0; q < 4; ++q) sk[k] += sST[(q * 3 + k) * NACT + lane];
                ek[k] = exp(sk[k]); emk[k] = exp(-sk[k]);
            }
            tval = w[B3 + 2];
#pragma unroll
            for (int q = 0; q < 4; ++q) tval += sST[(q * 3 + 2) * NACT + lane];
            const double target = ft_wrap(Pn - tval);
            double lo = -FT_PI, hi = FT_PI, xs = 0.0;
            bool done = false;
            for (int it = 0; it < 200; ++it) {
                const double hx = xs / 2, cs = cos(hx), sn = sin(hx), th = tan(hx);
                double f = 0.0, fp = 0.0;
#pragma unroll
                for (int k = 0; k < NMIX; ++k) {
                    f += ft_wrap(2 * atan(ek[k] * th));
                    fp += 1.0 / (emk[k] * cs * cs + ek[k] * sn * sn);
                }
                f /= NMIX; fp /= NMIX;
                const double err = target - f;
                if (!done) {
                    if (fabs(err) <= A.tol) done = true;
                    else {
                        if (err > 0) lo = xs; else hi = xs;
                        double xn = xs + err / fp;
                        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
                        if (xn == xs) done = true;
                        xs = xn;
                    }
                }
                if (__all(done)) break;
            }
            double m = -INFINITY, ljv[NMIX];
            const double hx = xs / 2, cs = cos(hx), sn = sin(hx);
#pragma unroll
            for (int k = 0; k < NMIX; ++k) { ljv[k] = -log(emk[k] * cs * cs + ek[k] * sn * sn); m = fmax(m, ljv[k]); }
            double se = 0.0;
#pragma unroll
            for (int k = 0; k < NMIX; ++k) se += exp(ljv[k] - m);
            const double lj = m + log(se) - log((double)NMIX);
            if (avalid) sDL[ar * FT + ac] = xs - Pn;
            const double tot = ft_wave_sum(avalid ? -lj : 0.0);
            if (lane == 0 && A.logj_part) A.logj_part[(size_t)b * ntiles + tile] = tot;
        }
    }

    if (MODE == 0 || MODE == 3) {
        __syncthreads();
        // ---- link update x' = wrap(x +- delta) on the active stripe -------
        if (A.y) {
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
        return;
    }

    if (BWD) {
        // ---- adjoint of the transform at the tile's own active sites -------
        double gdelta = 0.0, cb = 0.0;
        if (wave < NMIX) {
            cb = A.glogj ? A.glogj[b] : A.glogj_const;
            if (avalid) {
                if (A.up_link) {
                    const double g = A.up_link[(size_t)b * 2 * n + (size_t)mu * n + ai * L + aj];
                    gdelta = mu == 0 ? g : -g;
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
            const double wk = exp(ljk - m) / se;                    // softmax weight of component k
            const double sinP = sin(Pa);
            const double elj = 1.0 / Dk;                            // dy_k/dx
            const double gs = avalid ? gdelta * (sinP * elj / NMIX) + cb * wk * (ems * cs2 - es * sn2) * elj : 0.0;
            const double gpk = avalid ? gdelta * (elj / NMIX) - cb * wk * sinP * 0.5 * (es - ems) * elj : 0.0;
            if (avalid) sGO[wave * N3 + ar * FT + ac] = gs;
            sT2[(wave * 4 + 2) * NACT + lane] = gpk;
        }
        __syncthreads();
        if (wave == 0 && avalid) {
            double g = -gdelta;
#pragma unroll
            for (int k = 0; k < NMIX; ++k) g += sT2[(k * 4 + 2) * NACT + lane];
            sGP[(ar + 3) * R0 + ac + 3] = g;
            sGO[NMIX * N3 + ar * FT + ac] = gdelta;                // dL/dt
        }
        __syncthreads();

        // ---- weight gradient of conv3 (needs g_out and h2) ------------------
        if (MODE == 2) {
            double* gwp = A.gw_part + ((size_t)b * ntiles + tile) * FLOW_GW_STRIDE;
            for (int t = tid; t < 216 + 3; t += 256) {
                double acc = 0.0;
                if (t < 216) {
                    const int co = t / 72, ci = (t / 9) % 8, tap = t % 9, ky = tap / 3, kx = tap % 3;
                    for (int a = 0; a < NACT; ++a) {
                        const int r = mu == 0 ? (a >> 2) : off + 4 * (a >> 4);
                        const int c = mu == 0 ? off + 4 * (a & 3) : (a & 15);
                        acc = fma(sGO[co * N3 + r * FT + c], sH2[ci * N2 + (r + ky) * R2 + c + kx], acc);
                    }
                    gwp[CW2 + t] = acc;
                } else {
                    const int co = t - 216;
                    for (int a = 0; a < N3; ++a) acc += sGO[co * N3 + a];
                    gwp[CB2 + co] = acc;
                }
            }
        }

        // ---- conv3^T, times act'(z2)  -> gz2 (in place over D2) -------------
        for (int s = lane128; s < N2; s += 128) {
            const int r = s / R2, c = s - r * R2;
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int co = 0; co < 3; ++co)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int rr = r - ky, cc = c - kx;
                        const bool ok = (rr >= 0) && (rr < FT) && (cc >= 0) && (cc < FT);
                        const double g = ok ? sGO[co * N3 + rr * FT + cc] : 0.0;
                        const double* wp = w + W3B + ((co * 9 + ky * 3 + kx) * 8 + half * 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] = fma(g, wp[k], acc[k]);
                    }
#pragma unroll
            for (int k = 0; k < 4; ++k) sD2[(half * 4 + k) * N2 + s] *= acc[k];
        }
        __syncthreads();

        if (MODE == 2) {
            double* gwp = A.gw_part + ((size_t)b * ntiles + tile) * FLOW_GW_STRIDE;
            for (int t = tid; t < 576 + 8; t += 256) {
                double acc = 0.0;
                if (t < 576) {
                    const int co = t / 72, ci = (t / 9) % 8, tap = t % 9, ky = tap / 3, kx = tap % 3;
                    for (int r = 0; r < R2; ++r)
                        for (int c = 0; c < R2; ++c)
                            acc = fma(sD2[co * N2 + r * R2 + c], sH1[ci * N1 + (r + ky) * R1 + c + kx], acc);
                    gwp[CW1 + t] = acc;
                } else {
                    const int co = t - 576;
                    for (int a = 0; a < N2; ++a) acc += sD2[co * N2 + a];
                    gwp[CB1 + co] = acc;
                }
            }
        }

        // ---- conv2^T, times act'(z1) -> gz1 (in place over D1) --------------
        for (int s = lane128; s < N1; s += 128) {
            const int r = s / R1, c = s - r * R1;
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            for (int co = 0; co < 8; ++co)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int rr = r - ky, cc = c - kx;
                        const bool ok = (rr >= 0) && (rr < R2) && (cc >= 0) && (cc < R2);
                        const double g = ok ? sD2[co * N2 + rr * R2 + cc] : 0.0;
                        const double* wp = w + W2B + ((co * 9 + ky * 3 + kx) * 8 + half * 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] = fma(g, wp[k], acc[k]);
                    }
#pragma unroll
            for (int k = 0; k < 4; ++k) sD1[(half * 4 + k) * N1 + s] *= acc[k];
        }
        __syncthreads();

        if (MODE == 2) {
            double* gwp = A.gw_part + ((size_t)b * ntiles + tile) * FLOW_GW_STRIDE;
            for (int t = tid; t < 144 + 8; t += 256) {
                double acc = 0.0;
                if (t < 144) {
                    const int co = t / 18, ci = (t / 9) % 2, tap = t % 9, ky = tap / 3, kx = tap % 3;
                    for (int r = 0; r < R1; ++r)
                        for (int c = 0; c < R1; ++c)
                            acc = fma(sD1[co * N1 + r * R1 + c], sIn[ci * N0 + (r + ky) * R0 + c + kx], acc);
                    gwp[CW0 + t] = acc;
                } else {
                    const int co = t - 144;
                    for (int a = 0; a < N1; ++a) acc += sD1[co * N1 + a];
                    gwp[CB0 + co] = acc;
                }
            }
        }

        // ---- conv1^T and the (cos, sin) adjoint at frozen plaquettes --------
        for (int t = tid; t < N0; t += 256) {
            const int r = t / R0, c = t - r * R0;
            const int i = ft_modL(i0 - 3 + r, L), j = ft_modL(j0 - 3 + c, L);
            const int sel = ft_stripe(i, j, mu, off);
            if (sel == 1 || sel == 2) {
                double gc = 0.0, gs = 0.0;
                for (int co = 0; co < 8; ++co)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int rr = r - ky, cc = c - kx;
                            const bool ok = (rr >= 0) && (rr < R1) && (cc >= 0) && (cc < R1);
                            const double g = ok ? sD1[co * N1 + rr * R1 + cc] : 0.0;
                            const double* wp = w + W1B + (co * 9 + ky * 3 + kx) * 2;
                            gc = fma(g, wp[0], gc); gs = fma(g, wp[1], gs);
                        }
                sGP[t] = -sIn[N0 + t] * gc + sIn[t] * gs;
            }
        }
        __syncthreads();
        double* out = A.gp_part + ((size_t)b * ntiles + tile) * N0;
        for (int t = tid; t < N0; t += 256) out[t] = sGP[t];
    }
}

// out[b] (+)= sign * sum_t part[b][t]
__global__ void k_sum_parts(const double* __restrict__ part, int B, int np, int nsets, double sign, int accumulate,
                            double* __restrict__ out) {
    // one wave per chain: lane l sums the partials t = l, l + 64, ... in order, then a fixed xor tree
    // (deterministic; one thread per chain read L^2 / 256 partials serially: 124 us per call at L = 256).
    // nsets partial sets [set][B][np] (the layers of a sweep) are summed one after the other, in order: the same
    // arithmetic as one accumulating call per set, in one launch.
    // The launch has one wave per set, up to sixteen: wave k sums set q0 + k, thread 0 adds the sets' sums in order behind a barrier
    // (one wave running 16 layers x 256 tiles by itself: 29 us of a training step at L = 256).
    __shared__ double sa[16];
    const int b = blockIdx.x, lane = threadIdx.x & (FT_WAVE - 1), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    double tot = (threadIdx.x == 0 && accumulate) ? out[b] : 0.0;
    for (int q0 = 0; q0 < nsets; q0 += nw) {
        const int q = q0 + wave;
        if (q < nsets) {
            double a = 0.0;
            for (int t = lane; t < np; t += FT_WAVE) a += part[((size_t)q * B + b) * np + t];
            a = ft_wave_sum(a);
            if (lane == 0) sa[wave] = a;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int k = 0; k < nw && q0 + k < nsets; ++k) tot += sign * sa[k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[b] = tot;
}

// gp[b][i][j] (+)= sum over tiles and over every window position that wraps onto (i, j).
// Tiles are tr x tc sites, their windows (tr+6) x (tc+6) at offset -3; fixed summation order.
__global__ void k_gather_gp(const double* __restrict__ part, int L, int tr, int tc, int nti, int ntj,
                            int accumulate, double* __restrict__ gp) {
    const int b = blockIdx.y;
    const int n = L * L;
    const int ntiles = nti * ntj;
    const int wr = tr + 6, wc = tc + 6, n0 = wr * wc;
    const bool fast = nti >= 4 && ntj >= 4 && nti * tr == L && ntj * tc == L;   // windows never wrap onto themselves
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const int i = s / L, j = s - i * L;
        double acc = 0.0;
        if (fast) {
            // a site lies in its own tile's window and in at most one neighbour's per dimension
            const int ci = i / tr, cj = j / tc, li = i - ci * tr, lj = j - cj * tc;
            int ti[2], ri[2], tj[2], rj[2];
            int ni = 1, nj = 1;
            ti[0] = ci; ri[0] = li + 3;
            if (li < 3) { ti[1] = ci == 0 ? nti - 1 : ci - 1; ri[1] = li + tr + 3; ni = 2; }
            else if (li >= tr - 3) { ti[1] = ci + 1 == nti ? 0 : ci + 1; ri[1] = li - tr + 3; ni = 2; }
            tj[0] = cj; rj[0] = lj + 3;
            if (lj < 3) { tj[1] = cj == 0 ? ntj - 1 : cj - 1; rj[1] = lj + tc + 3; nj = 2; }
            else if (lj >= tc - 3) { tj[1] = cj + 1 == ntj ? 0 : cj + 1; rj[1] = lj - tc + 3; nj = 2; }
            // same fixed order as the general path: tiles ascending in (di, dj) of [c-1, c, c+1]
            double v[4] = {0.0, 0.0, 0.0, 0.0};
            for (int a = 0; a < ni; ++a)
                for (int c2 = 0; c2 < nj; ++c2)
                    v[a * 2 + c2] = part[((size_t)b * ntiles + ti[a] * ntj + tj[c2]) * n0 + ri[a] * wc + rj[c2]];
            acc = ((v[0] + v[1]) + v[2]) + v[3];
        } else {
        // candidate tiles per dimension: all of them when there are <= 3, else own and both neighbours
        const int ni = nti <= 3 ? nti : 3, bi = nti <= 3 ? 0 : i / tr - 1;
        const int nj = ntj <= 3 ? ntj : 3, bj = ntj <= 3 ? 0 : j / tc - 1;
        for (int di = 0; di < ni; ++di) {
            const int ti = (bi + di + nti) % nti;
            const int r0 = ft_modL(i - ti * tr + 3, L);
            for (int r = r0; r < wr; r += L)
                for (int dj = 0; dj < nj; ++dj) {
                    const int tj = (bj + dj + ntj) % ntj;
                    const int c0 = ft_modL(j - tj * tc + 3, L);
                    const double* p = part + ((size_t)b * ntiles + ti * ntj + tj) * n0 + r * wc;
                    for (int c = c0; c < wc; c += L) acc += p[c];
                }
        }
        }
        const size_t o = (size_t)b * n + s;
        gp[o] = (accumulate ? gp[o] : 0.0) + acc;
    }
}

// gx = gy + adj(gp):  gx0[i][j] = gy0 + gp[i][j] - gp[i][j-1];  gx1 = gy1 - gp[i][j] + gp[i-1][j]
__global__ void k_adj_add(const double* __restrict__ gp, const double* __restrict__ gy, int L,
                          double* __restrict__ gx) {
    const int b = blockIdx.y;
    const int n = L * L;
    const double* g = gp + (size_t)b * n;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const int i = s / L, j = s - i * L;
        const int im = i == 0 ? L - 1 : i - 1, jm = j == 0 ? L - 1 : j - 1;
        const size_t s0 = (size_t)b * 2 * n + s, s1 = s0 + n;
        const double gc = g[s];
        gx[s0] = (gy ? gy[s0] : 0.0) + (gc - g[i * L + jm]);
        gx[s1] = (gy ? gy[s1] : 0.0) + (g[im * L + j] - gc);
    }
}

// out[g][idx] = scale * sum over the g-th chunk of partials part[p][idx] (p in [g*chunk, (g+1)*chunk));
// 64 idx x RG_SL slices per block, fixed order.  Run twice (partials -> groups of <= 32 rows -> 1 row) so that a thread
// sums a handful of rows: each load is a dependent 7.7 KB-strided access, and with one level and four slices a training
// step at L = 16, batch 512 spent half of its GPU time here (128 back-to-back loads per thread, 39 us per layer).
constexpr int RG_SL = 16;
__global__ __launch_bounds__(64 * RG_SL) void k_reduce_gw(const double* __restrict__ part, int np, int chunk, double scale,
                                                          int accumulate, double* __restrict__ out, size_t part_lstride,
                                                          size_t out_lstride) {
    __shared__ double red[64 * RG_SL];
    const int li = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + li;
    const int g = blockIdx.y;
    part += (size_t)blockIdx.z * part_lstride;               // several layers in one launch: blockIdx.z = layer
    out += (size_t)blockIdx.z * out_lstride;
    const int p0 = g * chunk, p1 = min(np, p0 + chunk);
    double a = 0.0;
    if (idx < FTHMC_W_PER_LAYER)
        for (int p = p0 + sl; p < p1; p += RG_SL) a += part[(size_t)p * FLOW_GW_STRIDE + idx];
    red[threadIdx.x] = a;
    __syncthreads();
    if (sl == 0 && idx < FTHMC_W_PER_LAYER) {
        double t = red[li];
#pragma unroll
        for (int k = 1; k < RG_SL; ++k) t += red[64 * k + li];
        double* o = out + (size_t)g * FLOW_GW_STRIDE + idx;
        *o = (accumulate ? *o : 0.0) + scale * t;
    }
}

inline dim3 flow_grid(int B, int L) { int t = (L + FT - 1) / FT; return dim3(t, t, B); }

}  // namespace

namespace fthmc {

int launch_pack_weights(const double* w, int n_layers, double* wint, hipStream_t s, unsigned long long token) {
    if (n_layers <= 0) return FTHMC_OK;
    static_assert(WT3C + LT3_SIZE <= FLOW_WSTAMP0, "the stamps sit behind the last weight block");
    hipLaunchKernelGGL(k_pack_weights, dim3(n_layers, FLOW_WSTAMPS), dim3(256), 0, s, w, n_layers, wint, token);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_flow_fwd(const FlowLayerArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_flow_layer<0>, flow_grid(a.B, a.L), dim3(256), 0, s, a);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_flow_rev(const FlowLayerArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_flow_layer<3>, flow_grid(a.B, a.L), dim3(256), 0, s, a);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_flow_bwd(const FlowLayerArgs& a, bool wgrad, hipStream_t s) {
    if (wgrad) hipLaunchKernelGGL(k_flow_layer<2>, flow_grid(a.B, a.L), dim3(256), 0, s, a);
    else       hipLaunchKernelGGL(k_flow_layer<1>, flow_grid(a.B, a.L), dim3(256), 0, s, a);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_sum_parts(const double* part, int B, int nparts, double sign, int accumulate, double* out,
                     hipStream_t s, int nsets) {
    const int nw = nsets < 1 ? 1 : nsets > 16 ? 16 : nsets;
    hipLaunchKernelGGL(k_sum_parts, dim3(B), dim3(nw * FT_WAVE), 0, s, part, B, nparts, nsets, sign, accumulate, out);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_gather_gp(const double* gp_part, int B, int L, FlowGeom g, int accumulate, double* gp, hipStream_t s) {
    int gx = (L * L + 255) / 256; if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_gather_gp, dim3(gx, B), dim3(256), 0, s, gp_part, L, g.tr, g.tc, g.nti(L), g.ntj(L), accumulate, gp);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_adj_add(const double* gp, const double* gy, int B, int L, double* gx, hipStream_t s) {
    int g = (L * L + 255) / 256; if (g > 64) g = 64;
    hipLaunchKernelGGL(k_adj_add, dim3(g, B), dim3(256), 0, s, gp, gy, L, gx);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_reduce_gw(const double* gw_part, int nparts, double scale, int accumulate, double* gw,
                     double* tmp, hipStream_t s, int nlayers, size_t part_lstride) {
    const int nb = (FTHMC_W_PER_LAYER + 63) / 64;
    const unsigned nz = nlayers > 1 ? nlayers : 1;
    static_assert(4 * RG_SL == 64, "flow_reduce_groups (kernels.h) states the same threshold");
    if (nparts <= 4 * RG_SL || !tmp) {                     // a thread sums at most four rows: one level
        hipLaunchKernelGGL(k_reduce_gw, dim3(nb, 1, nz), dim3(64 * RG_SL), 0, s, gw_part, nparts, nparts, scale, accumulate, gw,
                           part_lstride, (size_t)FTHMC_W_PER_LAYER);
        FT_LAUNCH_CHECK(); return FTHMC_OK;
    }
    // groups of ~32 rows (two per thread), at most FLOW_REDUCE_GROUPS of them (the size of tmp)
    const int groups = flow_reduce_groups(nparts), g0 = (nparts + 31) / 32 > FLOW_REDUCE_GROUPS ? FLOW_REDUCE_GROUPS : (nparts + 31) / 32;
    const int chunk = (nparts + g0 - 1) / g0;
    const size_t tmp_l = (size_t)groups * FLOW_GW_STRIDE;   // the layers' rows side by side
    hipLaunchKernelGGL(k_reduce_gw, dim3(nb, groups, nz), dim3(64 * RG_SL), 0, s, gw_part, nparts, chunk, 1.0, 0, tmp, part_lstride, tmp_l);
    FT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_reduce_gw, dim3(nb, 1, nz), dim3(64 * RG_SL), 0, s, tmp, groups, groups, scale, accumulate, gw, tmp_l,
                       (size_t)FTHMC_W_PER_LAYER);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}

}  // namespace fthmc
