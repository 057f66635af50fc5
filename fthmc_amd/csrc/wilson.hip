// Wilson-action stencil kernels for 2D U(1): plaquette, action, topological
// charge, gauge force, fused leapfrog step, Metropolis select.
// HBM-bound: every kernel reads each link once per pass (coalesced, rows of the
// [B][2][L][L] field are contiguous in j) and shares plaquettes through LDS.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TS = 16;          // stencil tile (TS x TS sites, 256 threads)

// ---------------------------------------------------------------- elementwise
__global__ void k_wrap(const double* __restrict__ x, double* __restrict__ o, size_t n, int reg) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) o[i] = reg ? ft_regularize(x[i]) : ft_wrap(x[i]);
}

__global__ void k_axpy(const double* __restrict__ x, const double* __restrict__ p, double a,
                       double* __restrict__ o, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) o[i] = x[i] + a * p[i];
}

// ---------------------------------------------------------------- plaquettes
__global__ void k_plaq(const double* __restrict__ x, double* __restrict__ P, int L) {
    const int b = blockIdx.y;
    const int n = L * L;
    const double* x0 = x + (size_t)b * 2 * n;
    const double* x1 = x0 + n;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const int i = s / L, j = s - i * L;
        const int ip = i + 1 == L ? 0 : i + 1, jp = j + 1 == L ? 0 : j + 1;
        P[(size_t)b * n + s] = x0[s] - x1[s] - x0[i * L + jp] + x1[ip * L + j];
    }
}

// One workgroup per chain: S = -beta sum cos P, Q = sum wrap(P) / 2pi.
// xform: 0 none, 1 regularize links first (end of an HMC trajectory).
// (tid of nthr: the thread's place in the chain's workgroup -- k_action_charge_waves runs the workgroup's waves as workgroups of
// their own)
template <int XFORM>
__device__ __forceinline__ void chain_action_charge(const double* __restrict__ x0, int L, int tid, int nthr,
                                                    double& csum, double& qsum) {
    const int n = L * L;
    const double* x1 = x0 + n;
    double c = 0.0, q = 0.0;
    for (int s = tid; s < n; s += nthr) {
        const int i = s / L, j = s - i * L;
        const int ip = i + 1 == L ? 0 : i + 1, jp = j + 1 == L ? 0 : j + 1;
        double a = x0[s], bb = x1[s], cc = x0[i * L + jp], d = x1[ip * L + j];
        if (XFORM) { a = ft_regularize(a); bb = ft_regularize(bb); cc = ft_regularize(cc); d = ft_regularize(d); }
        c += cos(a + d - cc - bb);           // summation order of BatchAction._u1_plaq
        q += ft_wrap(a - bb - cc + d);       // summation order of batch_plaqs
    }
    csum = c; qsum = q;
}

template <int XFORM>
__device__ __forceinline__ void chain_action_charge(const double* __restrict__ x0, int L, double& csum, double& qsum) {
    chain_action_charge<XFORM>(x0, L, (int)threadIdx.x, (int)blockDim.x, csum, qsum);
}

__global__ void k_action_charge(const double* __restrict__ x, int L, double beta,
                                double* __restrict__ S, double* __restrict__ Q,
                                double* __restrict__ plaq) {
    __shared__ double red[16];
    const int b = blockIdx.x;
    double c, q;
    chain_action_charge<0>(x + (size_t)b * 2 * L * L, L, c, q);
    c = ft_block_sum(c, red);
    q = ft_block_sum(q, red);
    if (threadIdx.x == 0) {
        const double s = (-beta) * c;
        if (S) S[b] = s;
        if (Q) Q[b] = q / FT_TWO_PI;
        if (plaq) plaq[b] = (-s) / (beta * (double)(L * L));
    }
}

// The same sums by the same threads in the same order, for a FEW chains of a LARGE lattice (a training shard: 32 chains of
// L = 256 kept 32 of 256 CUs busy for 87 us): wave w of chain b's workgroup runs as a workgroup of its own (grid nw x B), leaves
// its two wave sums in part[b][w][2], and k_action_charge_fin adds them as ft_block_sum does (0.0 + wave 0 + wave 1 + ...):
// bit-identical to k_action_charge with nw waves.
__global__ __launch_bounds__(FT_WAVE) void k_action_charge_waves(const double* __restrict__ x, int L, double* __restrict__ part) {
    const int b = blockIdx.y, w = blockIdx.x, nw = gridDim.x;
    double c, q;
    chain_action_charge<0>(x + (size_t)b * 2 * L * L, L, w * FT_WAVE + (int)threadIdx.x, nw * FT_WAVE, c, q);
    c = ft_wave_sum(c);
    q = ft_wave_sum(q);
    if (threadIdx.x == 0) { part[((size_t)b * nw + w) * 2] = c; part[((size_t)b * nw + w) * 2 + 1] = q; }
}
__global__ void k_action_charge_fin(const double* __restrict__ part, int nw, int B, int L, double beta, double* __restrict__ S,
                                    double* __restrict__ Q, double* __restrict__ plaq) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double c = 0.0, q = 0.0;
    for (int w = 0; w < nw; ++w) { c += part[((size_t)b * nw + w) * 2]; q += part[((size_t)b * nw + w) * 2 + 1]; }
    const double s = (-beta) * c;
    if (S) S[b] = s;
    if (Q) Q[b] = q / FT_TWO_PI;
    if (plaq) plaq[b] = (-s) / (beta * (double)(L * L));
}

__global__ void k_kinetic(const double* __restrict__ v, int n, double* __restrict__ K) {
    __shared__ double red[16];
    const int b = blockIdx.x;
    const double* vb = v + (size_t)b * n;
    double a = 0.0;
    for (int s = threadIdx.x; s < n; s += blockDim.x) a += vb[s] * vb[s];
    a = ft_block_sum(a, red);
    if (threadIdx.x == 0) K[b] = a;
}

// The scalars of one end of a flowed trajectory in ONE launch, one workgroup per chain (was five: k_sum_parts,
// k_action_charge, k_lincomb, k_kinetic, k_lincomb -- with two chain groups in lockstep the chip sat nearly idle through each):
//   log det J = sum over the sweep's layers and tiles of the log J partials   (k_sum_parts' order: wave 0, lane-strided, xor tree)
//   S_W, Q, plaq of the flowed field                                          (k_action_charge's order)
//   trip = (S_eff = S_W - log det J, plaq, Q)   -- or copied from state_in when the caller carries it over
//   H = S_eff + K / 2,  K = sum v^2                                           (k_kinetic's order)
// Launched with the block size k_action_charge / k_kinetic use for this L, so every sum is bit-identical to theirs.
__global__ void k_traj_energy(const double* __restrict__ xphys, int L, double beta, const double* __restrict__ lj_part, int np,
                              int nsets, const double* __restrict__ state_in, const double* __restrict__ v,
                              double* __restrict__ trip, double* __restrict__ H, int B) {
    __shared__ double red[16];
    __shared__ double sld;
    const int b = blockIdx.x;
    double seff, pq, qq;
    if (state_in) { seff = state_in[b]; pq = state_in[B + b]; qq = state_in[2 * B + b]; }
    else {
        if (threadIdx.x < FT_WAVE) {
            double tot = 0.0;
            for (int q = 0; q < nsets; ++q) {
                double a = 0.0;
                for (int t = threadIdx.x; t < np; t += FT_WAVE) a += lj_part[((size_t)q * B + b) * np + t];
                a = ft_wave_sum(a);
                tot += 1.0 * a;
            }
            if (threadIdx.x == 0) sld = tot;
        }
        double c, q;
        chain_action_charge<0>(xphys + (size_t)b * 2 * L * L, L, c, q);
        c = ft_block_sum(c, red);
        q = ft_block_sum(q, red);
        const double s = (-beta) * c;
        seff = 1.0 * s + (nsets > 0 ? -1.0 * sld : 0.0) + 0.0;           // k_lincomb(S, 1, logdet, -1, 0)
        pq = (-s) / (beta * (double)(L * L));
        qq = q / FT_TWO_PI;
    }
    const int n = 2 * L * L;
    const double* vb = v + (size_t)b * n;
    double a = 0.0;
    for (int s = threadIdx.x; s < n; s += blockDim.x) a += vb[s] * vb[s];
    a = ft_block_sum(a, red);
    if (threadIdx.x == 0) {
        trip[b] = seff; trip[B + b] = pq; trip[2 * B + b] = qq;
        H[b] = 1.0 * seff + 0.5 * a + 0.0;                                // k_lincomb(S_eff, 1, K, 0.5, 0)
    }
}

// out[b][mu][s] = sign * g[b][s]: a plaquette-level upstream gradient dressed as the link gradient that produces it in the
// coupling layer's adjoint (gdelta = +-gy[mu] at the active sites): lets the link-level backward kernels serve the
// plaquette-level map (NCPPlaqCouplingLayer.forward under autograd)
__global__ void k_plane_from(const double* __restrict__ g, int n, int mu, double sign, double* __restrict__ out) {
    const int b = blockIdx.y;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x)
        out[((size_t)b * 2 + mu) * n + s] = sign * g[(size_t)b * n + s];
}

// xo = x + a v and vo = v in one pass (the first half drift of a flowed trajectory and the working copy of the momenta)
__global__ void k_axpy_copy(const double* __restrict__ x, const double* __restrict__ p, double a, double* __restrict__ xo,
                            double* __restrict__ po, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { const double pv = p[i]; xo[i] = x[i] + a * pv; po[i] = pv; }
}

// ---------------------------------------------------------------- force
// Tile of TS x TS sites; sin P is evaluated once per plaquette on a
// (TS+1) x (TS+1) region (one extra row above / column to the left) in LDS.
// MODE 0: F = dS/dx of x.
// MODE 1: fused leapfrog step  x' = x + a p ; p' = p - dt F(x')  (ping-pong buffers)
// MODE 2: gP = beta sin P (plaquette-gradient field that seeds the flow backward sweep)
template <int MODE>
__global__ __launch_bounds__(256) void k_force(const double* __restrict__ x,
                                               const double* __restrict__ p,
                                               double* __restrict__ o0,   // F | x' | gP
                                               double* __restrict__ o1,   // - | p' | -
                                               int L, double beta, double a, double dt) {
    // Links are staged through LDS once (x0 on a (TS+1) x (TS+2) window, x1 on (TS+2) x (TS+1), MODE 1:
    // already drifted, x + a p, with p of the window kept for the kick), beta sin P is evaluated once per
    // plaquette of the (TS+1) x (TS+1) window (one extra row above / column to the left), the force of a site
    // is two differences of it.  Per site and launch: ~1.2 x 4 doubles read, 4 written.
    constexpr int W0C = TS + 2, W1C = TS + 1, NW0 = (TS + 1) * W0C, NW1 = (TS + 2) * W1C, WP = TS + 1;
    __shared__ double sx0[NW0], sx1[NW1], sq0[MODE == 1 ? NW0 : 1], sq1[MODE == 1 ? NW1 : 1], sp[WP * WP];
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
    const int n = L * L;
    const double* x0 = x + (size_t)b * 2 * n;
    const double* x1 = x0 + n;
    const double* p0 = MODE == 1 ? p + (size_t)b * 2 * n : nullptr;
    const double* p1 = MODE == 1 ? p0 + n : nullptr;
    const bool fastw = L >= TS + 2;                               // window lines wrap at most once (uniform)
    for (int t = threadIdx.x; t < NW0 + NW1; t += blockDim.x) {
        const bool second = t >= NW0;
        const int u = second ? t - NW0 : t;
        const int r = second ? (int)(__umul24((unsigned)u, ((1u << 20) + W1C - 1) / W1C) >> 20)
                             : (int)(__umul24((unsigned)u, ((1u << 20) + W0C - 1) / W0C) >> 20);
        const int c = u - r * (second ? W1C : W0C);
        int wi = i0 - 1 + r, wj = j0 - 1 + c;
        if (fastw) {
            wi = (int)min(min((unsigned)wi, (unsigned)(wi - L)), (unsigned)(wi + L));
            wj = (int)min(min((unsigned)wj, (unsigned)(wj - L)), (unsigned)(wj + L));
        } else { wi = ft_modL(wi, L); wj = ft_modL(wj, L); }
        const int at = __mul24(wi, L) + wj;
        double v = second ? x1[at] : x0[at];
        if (MODE == 1) {
            const double q = second ? p1[at] : p0[at];
            v += a * q;
            (second ? sq1 : sq0)[u] = q;
        }
        (second ? sx1 : sx0)[u] = v;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < WP * WP; t += blockDim.x) {
        const int r = (int)(__umul24((unsigned)t, ((1u << 20) + WP - 1) / WP) >> 20), c = t - r * WP;
        double sn, cs;
        ft_sincos(sx0[r * W0C + c] - sx1[r * W1C + c] - sx0[r * W0C + c + 1] + sx1[(r + 1) * W1C + c], &sn, &cs);
        sp[t] = beta * sn;
    }
    __syncthreads();
    const int r = threadIdx.x / TS, c = threadIdx.x - r * TS;
    const int i = i0 + r, j = j0 + c;
    if (i < L && j < L) {
        const double s = sp[(r + 1) * WP + c + 1];
        const size_t s0 = (size_t)b * 2 * n + (size_t)i * L + j, s1 = s0 + n;
        if (MODE == 2) {
            o0[(size_t)b * n + (size_t)i * L + j] = s;
        } else {
            const double f0 = s - sp[(r + 1) * WP + c];
            const double f1 = sp[r * WP + c + 1] - s;
            if (MODE == 0) { o0[s0] = f0; o0[s1] = f1; }
            else {
                const int u0 = (r + 1) * W0C + c + 1, u1 = (r + 1) * W1C + c + 1;
                o0[s0] = sx0[u0]; o0[s1] = sx1[u1];
                o1[s0] = sq0[u0] - dt * f0; o1[s1] = sq1[u1] - dt * f1;
            }
        }
    }
}

// Fused leapfrog step for lattices whose rows are whole multiples of 64 sites: x' = x + a p ; p' = p - dt F(x').
// A workgroup owns TR rows x 64 columns of one chain; a thread owns TWO adjacent sites of one row, so every access to the
// four planes (x0, x1, p0, p1) is one 16-byte load / store per lane and a wave covers two whole 512-byte row segments.
// The drifted links of the (TR + 2) x 66 window go through LDS once (the momenta of the halo are folded in at the load and
// never stored; the own momenta stay in registers), beta sin P is evaluated once per plaquette of the (TR + 1) x 65
// window, the force of a site is two differences of it.  HBM-bound: 64 B per site and launch algorithmically; the halo
// rows (2.5 of TR + ... per plane pair) are re-reads that the neighbouring workgroup's own loads leave in L2.
template <int TR>
__global__ __launch_bounds__(TR * 32) void k_leap_rows(const double* __restrict__ x, const double* __restrict__ p,
                                                       double* __restrict__ xo, double* __restrict__ po,
                                                       int L, double beta, double a, double dt) {
    constexpr int TC = 64, WC = TC + 2, NTH = TR * 32;
    typedef double double2_t __attribute__((ext_vector_type(2)));
    // window row r <-> lattice row i0 - 1 + r, window column c <-> lattice column j0 - 1 + c
    __shared__ __attribute__((aligned(16))) double sx0[(TR + 1) * WC], sx1[(TR + 2) * WC], ss[(TR + 1) * WC];
    const int b = blockIdx.z, i0 = blockIdx.y * TR, j0 = blockIdx.x * TC;
    const int n = L * L, t = threadIdx.x;
    const double* x0 = x + (size_t)b * 2 * n;
    const double* p0 = p + (size_t)b * 2 * n;
    auto wrapi = [&](int v) { return v < 0 ? v + L : (v >= L ? v - L : v); };
    // own sites: row i0 + tr, columns j0 + 2 q, j0 + 2 q + 1
    const int tr = t >> 5, q = t & 31;
    const int own = (i0 + tr) * L + j0 + 2 * q;
    const double2_t vx0 = *reinterpret_cast<const double2_t*>(x0 + own), vx1 = *reinterpret_cast<const double2_t*>(x0 + n + own);
    const double2_t vp0 = *reinterpret_cast<const double2_t*>(p0 + own), vp1 = *reinterpret_cast<const double2_t*>(p0 + n + own);
    // halo rows: plane 0 at i0 - 1, plane 1 at i0 - 1 and at i0 + TR (16 bytes per lane), then the halo columns
    // (plane 0 at j0 + 64 and j0 - 1, plane 1 at j0 - 1: one site per lane)
    double2_t hx = {0.0, 0.0}, hp = {0.0, 0.0};
    double cx = 0.0, cp = 0.0;
    int hrow = -1, hplane = 0, crow = -1, ccol = 0, cplane = 0;
    if (t < 96) {
        const int which = t >> 5;
        hplane = which == 0 ? 0 : 1;
        hrow = which == 2 ? TR + 1 : 0;
        const int at = hplane * n + wrapi(i0 - 1 + hrow) * L + j0 + 2 * q;
        hx = *reinterpret_cast<const double2_t*>(x0 + at); hp = *reinterpret_cast<const double2_t*>(p0 + at);
    } else if (t < 96 + 3 * TR + 2) {
        const int k = t - 96;
        if (k < TR + 1) { cplane = 0; ccol = TC + 1; crow = k; }                  // x0' at column j0 + 64, rows 0 .. TR
        else if (k < 2 * TR + 1) { cplane = 0; ccol = 0; crow = k - TR; }          // x0' at column j0 - 1, rows 1 .. TR
        else { cplane = 1; ccol = 0; crow = k - 2 * TR; }                          // x1' at column j0 - 1, rows 1 .. TR + 1
        const int at = cplane * n + wrapi(i0 - 1 + crow) * L + wrapi(j0 - 1 + ccol);
        cx = x0[at]; cp = p0[at];
    }
    // drift (own sites in registers, the window through LDS)
    const double2_t nx0 = {vx0.x + a * vp0.x, vx0.y + a * vp0.y}, nx1 = {vx1.x + a * vp1.x, vx1.y + a * vp1.y};
    {
        const int at = (tr + 1) * WC + 1 + 2 * q;
        sx0[at] = nx0.x; sx0[at + 1] = nx0.y; sx1[at] = nx1.x; sx1[at + 1] = nx1.y;
    }
    if (hrow >= 0) {
        double* d = (hplane == 0 ? sx0 : sx1) + hrow * WC + 1 + 2 * q;
        d[0] = hx.x + a * hp.x; d[1] = hx.y + a * hp.y;
    }
    if (crow >= 0) (cplane == 0 ? sx0 : sx1)[crow * WC + ccol] = cx + a * cp;
    __syncthreads();
    // beta sin P on window rows 0 .. TR, columns 0 .. 64 (the corner (0, 0) is never used)
    for (int u = t; u < (TR + 1) * (TC + 1); u += NTH) {
        const int r = u / (TC + 1), c = u - r * (TC + 1);
        double sn, cs;
        ft_sincos(sx0[r * WC + c] - sx1[r * WC + c] - sx0[r * WC + c + 1] + sx1[(r + 1) * WC + c], &sn, &cs);
        ss[r * WC + c] = beta * sn;
    }
    __syncthreads();
    {
        const int at = (tr + 1) * WC + 1 + 2 * q;
        const double s_l = ss[at - 1], s_a = ss[at], s_b = ss[at + 1], u_a = ss[at - WC], u_b = ss[at - WC + 1];
        const double2_t np0 = {vp0.x - dt * (s_a - s_l), vp0.y - dt * (s_b - s_a)};
        const double2_t np1 = {vp1.x - dt * (u_a - s_a), vp1.y - dt * (u_b - s_b)};
        double* xb = xo + (size_t)b * 2 * n;
        double* pb = po + (size_t)b * 2 * n;
        *reinterpret_cast<double2_t*>(xb + own) = nx0; *reinterpret_cast<double2_t*>(xb + n + own) = nx1;
        *reinterpret_cast<double2_t*>(pb + own) = np0; *reinterpret_cast<double2_t*>(pb + n + own) = np1;
    }
}

// v' = v - dt * adj(gP)   (adjoint of the plaquette stencil applied to a plaquette-
// gradient field; closes one flowed leapfrog kick), optionally followed by the
// drift x' = x + a v'.  In place on v (and x): every thread touches its own site only.
__global__ void k_kick_from_gp(const double* __restrict__ gp, double* __restrict__ v,
                               double* __restrict__ xq, double* __restrict__ Fout,
                               int L, double dt, double a, double* __restrict__ xreg) {
    const int b = blockIdx.y;
    const int n = L * L;
    const double* g = gp + (size_t)b * n;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const int i = s / L, j = s - i * L;
        const int im = i == 0 ? L - 1 : i - 1, jm = j == 0 ? L - 1 : j - 1;
        const double gc = g[s];
        const double f0 = gc - g[i * L + jm];
        const double f1 = g[im * L + j] - gc;
        const size_t s0 = (size_t)b * 2 * n + s, s1 = s0 + n;
        if (Fout) { Fout[s0] = f0; Fout[s1] = f1; }
        if (v) {
            const double v0 = v[s0] - dt * f0, v1 = v[s1] - dt * f1;
            v[s0] = v0; v[s1] = v1;
            if (xq) {
                const double y0 = xq[s0] + a * v0, y1 = xq[s1] + a * v1;
                xq[s0] = y0; xq[s1] = y1;
                if (xreg) { xreg[s0] = ft_regularize(y0); xreg[s1] = ft_regularize(y1); }
            }
        }
    }
}

// The two stencil kernels of a FLOWED leapfrog step for lattices whose rows are whole multiples of 64 sites (k_leap_rows'
// treatment): a workgroup owns TR rows x 64 columns of one chain, a thread TWO adjacent sites of one row, every access to a
// plane is one 16-byte load / store per lane; the left / right neighbour of a lane's pair comes from the neighbouring lane
// (the 32 lanes of a row segment are one half-wave), only the segment's edge lanes load it.  No LDS, no barrier.
// Same arithmetic per site as k_force<2> / k_kick_from_gp: results are bit-identical.
//
// gP = beta sin P(x): the seed of the flow's backward sweep (qed_helpers.py:226-242: d S_W / d P)
template <int TR>
__global__ __launch_bounds__(TR * 32) void k_gp_rows(const double* __restrict__ x, double* __restrict__ gp, int L, double beta) {
    typedef double double2_t __attribute__((ext_vector_type(2)));
    const int b = blockIdx.z, i = blockIdx.y * TR + (threadIdx.x >> 5), q = threadIdx.x & 31, j = blockIdx.x * 64 + 2 * q;
    const int n = L * L;
    const double* x0 = x + (size_t)b * 2 * n;
    const double* x1 = x0 + n;
    const int ip = i + 1 == L ? 0 : i + 1;
    const double2_t a0 = *reinterpret_cast<const double2_t*>(x0 + i * L + j), a1 = *reinterpret_cast<const double2_t*>(x1 + i * L + j);
    const double2_t d1 = *reinterpret_cast<const double2_t*>(x1 + ip * L + j);
    double r0 = __shfl_down(a0.x, 1, 32);                                     // x0[i][j + 2]: the next lane's first site
    if (q == 31) r0 = x0[i * L + (j + 2 == L ? 0 : j + 2)];
    double sa, sb, cs;
    ft_sincos(a0.x - a1.x - a0.y + d1.x, &sa, &cs);
    ft_sincos(a0.y - a1.y - r0 + d1.y, &sb, &cs);
    *reinterpret_cast<double2_t*>(gp + (size_t)b * n + i * L + j) = double2_t{beta * sa, beta * sb};
}

// v' = v - dt adj(gP), optionally x' = x + a v' and / or F = adj(gP)  (k_kick_from_gp on row strips)
template <int TR>
__global__ __launch_bounds__(TR * 32) void k_kick_rows(const double* __restrict__ gp, double* __restrict__ v, double* __restrict__ xq,
                                                       double* __restrict__ Fout, int L, double dt, double a, double* __restrict__ xreg) {
    typedef double double2_t __attribute__((ext_vector_type(2)));
    const int b = blockIdx.z, i = blockIdx.y * TR + (threadIdx.x >> 5), q = threadIdx.x & 31, j = blockIdx.x * 64 + 2 * q;
    const int n = L * L;
    const double* g = gp + (size_t)b * n;
    const int im = i == 0 ? L - 1 : i - 1;
    const double2_t gc = *reinterpret_cast<const double2_t*>(g + i * L + j), gu = *reinterpret_cast<const double2_t*>(g + im * L + j);
    double gl = __shfl_up(gc.y, 1, 32);                                       // gP[i][j - 1]: the previous lane's second site
    if (q == 0) gl = g[i * L + (j == 0 ? L - 1 : j - 1)];
    const double2_t f0 = {gc.x - gl, gc.y - gc.x}, f1 = {gu.x - gc.x, gu.y - gc.y};
    const size_t s0 = (size_t)b * 2 * n + (size_t)i * L + j, s1 = s0 + n;
    if (Fout) { *reinterpret_cast<double2_t*>(Fout + s0) = f0; *reinterpret_cast<double2_t*>(Fout + s1) = f1; }
    if (v) {
        const double2_t w0 = *reinterpret_cast<const double2_t*>(v + s0), w1 = *reinterpret_cast<const double2_t*>(v + s1);
        const double2_t v0 = {w0.x - dt * f0.x, w0.y - dt * f0.y}, v1 = {w1.x - dt * f1.x, w1.y - dt * f1.y};
        *reinterpret_cast<double2_t*>(v + s0) = v0; *reinterpret_cast<double2_t*>(v + s1) = v1;
        if (xq) {
            double2_t y0 = *reinterpret_cast<const double2_t*>(xq + s0), y1 = *reinterpret_cast<const double2_t*>(xq + s1);
            y0.x += a * v0.x; y0.y += a * v0.y; y1.x += a * v1.x; y1.y += a * v1.y;
            *reinterpret_cast<double2_t*>(xq + s0) = y0; *reinterpret_cast<double2_t*>(xq + s1) = y1;
            if (xreg) {                                                   // end of the MD: xr = regularize(x_) (ipynb/ft_hmc.py:426)
                *reinterpret_cast<double2_t*>(xreg + s0) = double2_t{ft_regularize(y0.x), ft_regularize(y0.y)};
                *reinterpret_cast<double2_t*>(xreg + s1) = double2_t{ft_regularize(y1.x), ft_regularize(y1.y)};
            }
        }
    }
}

// ---------------------------------------------------------------- Metropolis
// Per chain: H1 = S1 + K1/2, dH = H1 - H0, acc = u < exp(-dH),
// x_new = acc ? xform(x_prop) : x_old.   xform: 0 none, 1 regularize, 2 wrap.
__global__ void k_metropolis(const double* __restrict__ x_old, const double* __restrict__ x_prop,
                             const double* __restrict__ u, const double* __restrict__ H0,
                             const double* __restrict__ H1, int n2, int xform,
                             double* __restrict__ x_new, double* __restrict__ dH,
                             double* __restrict__ acc,
                             const double* __restrict__ obs_old, const double* __restrict__ obs_new,
                             double* __restrict__ obs_out, int n_obs, int B, double* __restrict__ o1, double* __restrict__ o2) {
    const int b = blockIdx.x;                                    // gridDim.y workgroups share the copy of a chain
    const double d = H1[b] - H0[b];
    const bool a = u[b] < exp(-d);
    if (threadIdx.x == 0 && blockIdx.y == 0) {
        if (dH) dH[b] = d;
        if (acc) acc[b] = a ? 1.0 : 0.0;
        // all selections are read before any is written: obs_out may be obs_old (the carried state updated in place)
        double sel[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < n_obs && k < 4; ++k) sel[k] = a ? obs_new[k * B + b] : obs_old[k * B + b];
        for (int k = 0; k < n_obs && k < 4; ++k) if (obs_out) obs_out[k * B + b] = sel[k];
        if (o1) o1[b] = sel[1];                                  // observables 1, 2 of the selected state (plaq, Q) where the caller wants them
        if (o2) o2[b] = sel[2];
    }
    const size_t o = (size_t)b * n2;
    for (int s = blockIdx.y * blockDim.x + threadIdx.x; s < n2; s += gridDim.y * blockDim.x) {
        double v;
        if (a) {
            v = x_prop[o + s];
            if (xform == 1) v = ft_regularize(v); else if (xform == 2) v = ft_wrap(v);
        } else v = x_old[o + s];
        x_new[o + s] = v;
    }
}

// out = ca * a + cb * b + c0 for every chain (tiny; H = S + K/2, S_eff = S_W - logdet, ...)
__global__ void k_lincomb(const double* __restrict__ a, double ca, const double* __restrict__ bv,
                          double cb, double c0, double* __restrict__ out, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[b] = ca * a[b] + (bv ? cb * bv[b] : 0.0) + c0;
}

// Run statistics of one trajectory folded into the running sums (one workgroup, fixed order):
// vec[8] += (n, acc, plaq, Q, Q^2, |Q - Qold|, dH, exp(-dH)) summed over the chains; Qold <- Q.
__global__ __launch_bounds__(256) void k_stats_accumulate(const double* __restrict__ acc, const double* __restrict__ plaq,
                                                          const double* __restrict__ Q, double* __restrict__ qold,
                                                          const double* __restrict__ dH, int B, double* __restrict__ vec) {
    __shared__ double red[16];
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const double q = Q[b], d = dH[b];
        s[0] += 1.0; s[1] += acc[b]; s[2] += plaq[b]; s[3] += q; s[4] += q * q; s[5] += fabs(q - qold[b]); s[6] += d; s[7] += exp(-d);
        qold[b] = q;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double t = ft_block_sum(s[k], red);
        if (threadIdx.x == 0) vec[k] += t;
    }
}

// ---------------------------------------------------------------- whole plain-HMC trajectory, one launch
// One workgroup per chain, links in LDS, momenta in registers (every thread owns NSITE sites of
// its chain for the whole trajectory): between H0 and the Metropolis select nothing touches HBM.
// For L <= 64 (x: 64 KB + beta sin P plane: 32 KB of LDS).  Same arithmetic as the step kernels:
//   x' = x + dt/2 v;  v' = v - dt F(x');  (nstep-1) x {x' += dt v'; v' -= dt F(x')};  x' += dt/2 v'
//   xr = regularize(x');  dH = S(xr) + v'^2/2 - S(x) - v^2/2;  acc = u < exp(-dH)
constexpr int TJ_NT = 1024, TJ_MAXL = 64, TJ_NSITE = TJ_MAXL * TJ_MAXL / TJ_NT;

// Metrics of one reverse-KL training step (train_step, fthmc/train.py:206-228; calc_dkl / calc_ess,
// fthmc/utils/distributions.py:23-37) in one workgroup:
//   row = [loss_dkl, ess, logp[B], logq[B], q[B], dq[B], plaq[B]]
//   loss_dkl = dkl_factor * mean(logq - logp);  ess = exp(2 logsumexp(logw) - logsumexp(2 logw)) / B,  logw = logp - logq
//   dq = sqrt((q - qi)^2);  plaq = logp / (beta V)    (train.py:219-221)
__global__ __launch_bounds__(256) void k_train_metrics(const double* __restrict__ logq, const double* __restrict__ logp,
                                                       const double* __restrict__ q, const double* __restrict__ qi, int B,
                                                       double inv_beta_vol, double dkl_factor, double* __restrict__ row) {
    __shared__ double red[16];
    __shared__ double smax;
    double m = -INFINITY, sd = 0.0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const double lw = logp[b] - logq[b];
        m = fmax(m, lw); sd += logq[b] - logp[b];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, FT_WAVE));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { double t = red[0]; for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = fmax(t, red[i]); smax = t; }
    __syncthreads();
    m = smax;
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const double e = exp(logp[b] - logq[b] - m);
        s1 += e; s2 += e * e;
        const double qq = q[b];
        row[2 + b] = logp[b]; row[2 + B + b] = logq[b]; row[2 + 2 * B + b] = qq;
        row[2 + 3 * B + b] = fabs(qq - qi[b]); row[2 + 4 * B + b] = logp[b] * inv_beta_vol;
    }
    const double t1 = ft_block_sum(s1, red), t2 = ft_block_sum(s2, red), td = ft_block_sum(sd, red);
    if (threadIdx.x == 0) {
        row[0] = dkl_factor * td / B;
        row[1] = t1 * t1 / t2 / B;                 // exp(2 (m + log s1) - (2 m + log s2)) / B
    }
}

// Adam / AdamW on ONE flat parameter buffer (optim.Adam(model.layers.parameters(), lr), fthmc/train.py:297; AdamW in
// restore_model_from_checkpoint, train.py:86): torch's single-tensor update, element by element --
//   g += wd p (Adam) | p *= 1 - lr wd (AdamW);  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// hp = (t, lr) lives on the DEVICE: the launch reads the step count and the learning rate from there and the last
// workgroup to finish moves t on, so the same launch can be replayed from a captured graph step after step.
__global__ __launch_bounds__(256) void k_adam(double* __restrict__ p, const double* __restrict__ g, double* __restrict__ m,
                                              double* __restrict__ v, double* __restrict__ hp, unsigned* __restrict__ ticket,
                                              size_t n, double b1, double b2, double eps, double wd, int decoupled) {
    const double t = hp[0] + 1.0, lr = hp[1];
    const double bc1 = 1.0 - pow(b1, t), bc2s = sqrt(1.0 - pow(b2, t));
    const double step_size = lr / bc1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double pi = p[i], gi = g[i];
        if (wd != 0.0) { if (decoupled) pi *= 1.0 - lr * wd; else gi += wd * pi; }
        const double mi = m[i] + (gi - m[i]) * (1.0 - b1);
        const double vi = v[i] * b2 + (1.0 - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step_size * (mi / (sqrt(vi) / bc2s + eps));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {          // every other workgroup has read hp[0] and finished
            *ticket = 0u;
            hp[0] = t;
        }
    }
}

__global__ __launch_bounds__(TJ_NT) void k_hmc_trajectory(const double* __restrict__ x, const double* __restrict__ v,
                                                          const double* __restrict__ u, int L, double beta, double dt,
                                                          int nstep, double* __restrict__ x_new, double* __restrict__ dH,
                                                          double* __restrict__ acc, double* __restrict__ H0o,
                                                          double* __restrict__ H1o) {
    __shared__ double sx[2 * TJ_MAXL * TJ_MAXL];        // links
    __shared__ double sp[TJ_MAXL * TJ_MAXL];            // beta sin P
    __shared__ double red[16];
    const int b = blockIdx.x, tid = threadIdx.x, n = L * L;
    const double* xb = x + (size_t)b * 2 * n;
    const double* vb = v + (size_t)b * 2 * n;
    double v0[TJ_NSITE], v1[TJ_NSITE];
    int st[TJ_NSITE], sjm[TJ_NSITE], sim[TJ_NSITE], sjp[TJ_NSITE], sip[TJ_NSITE];
    double kin = 0.0;
#pragma unroll
    for (int k = 0; k < TJ_NSITE; ++k) {
        const int s = tid + k * TJ_NT;
        st[k] = s < n ? s : -1;
        v0[k] = v1[k] = 0.0; sjm[k] = sim[k] = sjp[k] = sip[k] = 0;
        if (s < n) {
            const int i = s / L, j = s - i * L;
            sjp[k] = i * L + (j + 1 == L ? 0 : j + 1);  sip[k] = (i + 1 == L ? 0 : i + 1) * L + j;
            sjm[k] = i * L + (j == 0 ? L - 1 : j - 1);  sim[k] = (i == 0 ? L - 1 : i - 1) * L + j;
            sx[s] = xb[s]; sx[n + s] = xb[n + s];
            v0[k] = vb[s]; v1[k] = vb[n + s];
            kin += v0[k] * v0[k] + v1[k] * v1[k];
        }
    }
    __syncthreads();
    auto action = [&]() {                                // -beta sum cos P over the chain (summation order of BatchAction)
        double c = 0.0;
#pragma unroll
        for (int k = 0; k < TJ_NSITE; ++k)
            if (st[k] >= 0) { double sn_, cs_; ft_sincos(sx[st[k]] + sx[n + sip[k]] - sx[sjp[k]] - sx[n + st[k]], &sn_, &cs_); c += cs_; }
        return (-beta) * ft_block_sum(c, red);
    };
    const double h0 = action() + 0.5 * ft_block_sum(kin, red);
    for (int step = 0; step <= nstep; ++step) {
        const double a = (step == 0 || step == nstep) ? 0.5 * dt : dt;
#pragma unroll
        for (int k = 0; k < TJ_NSITE; ++k)
            if (st[k] >= 0) { sx[st[k]] += a * v0[k]; sx[n + st[k]] += a * v1[k]; }
        if (step == nstep) break;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TJ_NSITE; ++k)
            if (st[k] >= 0) {                       // own sincos (common.h): ~3x shorter than ocml's sin, same accuracy class
                double sn_, cs_;
                ft_sincos(sx[st[k]] - sx[n + st[k]] - sx[sjp[k]] + sx[n + sip[k]], &sn_, &cs_);
                sp[st[k]] = beta * sn_;
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TJ_NSITE; ++k)
            if (st[k] >= 0) {
                const double sc = sp[st[k]];
                v0[k] -= dt * (sc - sp[sjm[k]]);
                v1[k] -= dt * (sp[sim[k]] - sc);
            }
    }
    kin = 0.0;
#pragma unroll
    for (int k = 0; k < TJ_NSITE; ++k)
        if (st[k] >= 0) {
            sx[st[k]] = ft_regularize(sx[st[k]]); sx[n + st[k]] = ft_regularize(sx[n + st[k]]);
            kin += v0[k] * v0[k] + v1[k] * v1[k];
        }
    __syncthreads();
    const double h1 = action() + 0.5 * ft_block_sum(kin, red);
    const double d = h1 - h0;
    const bool ok = u[b] < exp(-d);
    if (tid == 0) {
        if (dH) dH[b] = d;
        if (acc) acc[b] = ok ? 1.0 : 0.0;
        if (H0o) H0o[b] = h0;
        if (H1o) H1o[b] = h1;
    }
    double* xo = x_new + (size_t)b * 2 * n;
#pragma unroll
    for (int k = 0; k < TJ_NSITE; ++k)
        if (st[k] >= 0) {
            xo[st[k]] = ok ? sx[st[k]] : xb[st[k]];
            xo[n + st[k]] = ok ? sx[n + st[k]] : xb[n + st[k]];
        }
}

int g_leap_rows = 1;     // FTHMC_LEAP_ROWS=0 in the environment: the 16 x 16-tile kernel for every L (A/B runs)

inline int ew_grid(size_t n) { size_t g = (n + 255) / 256; return (int)(g > 2048 ? 2048 : (g ? g : 1)); }
inline dim3 tile_grid(int B, int L) { return dim3((L + TS - 1) / TS, (L + TS - 1) / TS, B); }

}  // namespace

namespace fthmc {

void set_leap_rows(int v) { g_leap_rows = v; }

int launch_wrap(const double* x, double* o, size_t n, int reg, hipStream_t s) {
    if (n == 0) return FTHMC_OK;
    hipLaunchKernelGGL(k_wrap, dim3(ew_grid(n)), dim3(256), 0, s, x, o, n, reg);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_axpy(const double* x, const double* p, double a, double* o, size_t n, hipStream_t s) {
    if (n == 0) return FTHMC_OK;
    hipLaunchKernelGGL(k_axpy, dim3(ew_grid(n)), dim3(256), 0, s, x, p, a, o, n);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_plaq(const double* x, double* P, int B, int L, hipStream_t s) {
    int gx = (L * L + 255) / 256; if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_plaq, dim3(gx, B), dim3(256), 0, s, x, P, L);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_action_charge(const double* x, int B, int L, double beta, double* S, double* Q,
                         double* plaq, hipStream_t s, double* wave_part) {
    const int nt = L * L >= 4096 ? 1024 : (L * L >= 1024 ? 512 : 256);
    if (wave_part && B < 128 && L >= 128) {                 // fewer workgroups than half the CUs, each busy for tens of us
        const int nw = nt / FT_WAVE;
        hipLaunchKernelGGL(k_action_charge_waves, dim3(nw, B), dim3(FT_WAVE), 0, s, x, L, wave_part);
        FT_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_action_charge_fin, dim3((B + 63) / 64), dim3(64), 0, s, wave_part, nw, B, L, beta, S, Q, plaq);
        FT_LAUNCH_CHECK(); return FTHMC_OK;
    }
    hipLaunchKernelGGL(k_action_charge, dim3(B), dim3(nt), 0, s, x, L, beta, S, Q, plaq);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_kinetic(const double* v, int B, int L, double* K, hipStream_t s) {
    const int n = 2 * L * L;
    const int nt = n >= 8192 ? 1024 : (n >= 2048 ? 512 : 256);
    hipLaunchKernelGGL(k_kinetic, dim3(B), dim3(nt), 0, s, v, n, K);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_lincomb(const double* a, double ca, const double* b, double cb, double c0, double* out,
                   int B, hipStream_t s) {
    hipLaunchKernelGGL(k_lincomb, dim3((B + 255) / 256), dim3(256), 0, s, a, ca, b, cb, c0, out, B);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_stats_accumulate(const double* acc, const double* plaq, const double* Q, double* qold, const double* dH, int B,
                            double* vec, hipStream_t s) {
    hipLaunchKernelGGL(k_stats_accumulate, dim3(1), dim3(256), 0, s, acc, plaq, Q, qold, dH, B, vec);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_train_metrics(const double* logq, const double* logp, const double* q, const double* qi, int B,
                         double inv_beta_vol, double dkl_factor, double* row, hipStream_t s) {
    hipLaunchKernelGGL(k_train_metrics, dim3(1), dim3(256), 0, s, logq, logp, q, qi, B, inv_beta_vol, dkl_factor, row);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_adam(double* p, const double* g, double* m, double* v, double* hp, size_t n, double b1, double b2, double eps,
                double wd, int decoupled, hipStream_t s) {
    size_t gx = (n + 1023) / 1024; if (gx > 256) gx = 256; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)gx), dim3(256), 0, s, p, g, m, v, hp, reinterpret_cast<unsigned*>(hp + 2), n, b1, b2,
                       eps, wd, decoupled);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_wilson_force(const double* x, int B, int L, double beta, double* F, hipStream_t s) {
    hipLaunchKernelGGL(k_force<0>, tile_grid(B, L), dim3(256), 0, s, x, nullptr, F, nullptr, L, beta, 0.0, 0.0);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_leap_step(const double* x, const double* p, double* xo, double* po, int B, int L,
                     double beta, double a, double dt, hipStream_t s) {
    if (L % 64 == 0 && g_leap_rows) {            // whole 64-site row segments: 16-byte accesses, two sites per thread
        constexpr int TR = 8;
        hipLaunchKernelGGL(k_leap_rows<TR>, dim3(L / 64, L / TR, B), dim3(TR * 32), 0, s, x, p, xo, po, L, beta, a, dt);
        FT_LAUNCH_CHECK(); return FTHMC_OK;
    }
    hipLaunchKernelGGL(k_force<1>, tile_grid(B, L), dim3(256), 0, s, x, p, xo, po, L, beta, a, dt);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_hmc_trajectory_fused(const double* x, const double* v, const double* u, int B, int L, double beta,
                                double dt, int nstep, double* x_new, double* dH, double* acc, double* H0,
                                double* H1, hipStream_t s) {
    if (L > TJ_MAXL) return FTHMC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_hmc_trajectory, dim3(B), dim3(TJ_NT), 0, s, x, v, u, L, beta, dt, nstep, x_new, dH, acc, H0, H1);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_wilson_gp(const double* x, int B, int L, double beta, double* gp, hipStream_t s) {
    if (L % 64 == 0 && g_leap_rows) {
        constexpr int TR = 8;
        hipLaunchKernelGGL(k_gp_rows<TR>, dim3(L / 64, L / TR, B), dim3(TR * 32), 0, s, x, gp, L, beta);
        FT_LAUNCH_CHECK(); return FTHMC_OK;
    }
    hipLaunchKernelGGL(k_force<2>, tile_grid(B, L), dim3(256), 0, s, x, nullptr, gp, nullptr, L, beta, 0.0, 0.0);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_kick_from_gp(const double* gp, double* v, double* xq, double* Fout, int B, int L,
                        double dt, double a, hipStream_t s, double* xreg) {
    if (L % 64 == 0 && g_leap_rows) {
        constexpr int TR = 8;
        hipLaunchKernelGGL(k_kick_rows<TR>, dim3(L / 64, L / TR, B), dim3(TR * 32), 0, s, gp, v, xq, Fout, L, dt, a, xreg);
        FT_LAUNCH_CHECK(); return FTHMC_OK;
    }
    int gx = (L * L + 255) / 256; if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_kick_from_gp, dim3(gx, B), dim3(256), 0, s, gp, v, xq, Fout, L, dt, a, xreg);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_traj_energy(const double* xphys, int B, int L, double beta, const double* lj_part, int np, int nsets,
                       const double* state_in, const double* v, double* trip, double* H, hipStream_t s) {
    const int nt = L * L >= 4096 ? 1024 : (L * L >= 1024 ? 512 : 256);           // = launch_action_charge = launch_kinetic
    hipLaunchKernelGGL(k_traj_energy, dim3(B), dim3(nt), 0, s, xphys, L, beta, lj_part, np, nsets, state_in, v, trip, H, B);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_plane_from(const double* g, int B, int L, int mu, double sign, double* out, hipStream_t s) {
    int gx = (L * L + 255) / 256; if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_plane_from, dim3(gx, B), dim3(256), 0, s, g, L * L, mu, sign, out);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_axpy_copy(const double* x, const double* p, double a, double* xo, double* po, size_t n, hipStream_t s) {
    if (n == 0) return FTHMC_OK;
    hipLaunchKernelGGL(k_axpy_copy, dim3(ew_grid(n)), dim3(256), 0, s, x, p, a, xo, po, n);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_metropolis(const double* x_old, const double* x_prop, const double* u, const double* H0,
                      const double* H1, int B, int L, int xform, double* x_new, double* dH,
                      double* acc, const double* obs_old, const double* obs_new, double* obs_out,
                      int n_obs, hipStream_t s, double* o1, double* o2) {
    int gy = (2 * L * L + 2047) / 2048;                          // ~8 sites per thread; one workgroup per chain left the copy latency-bound
    if (gy > 16) gy = 16;
    hipLaunchKernelGGL(k_metropolis, dim3(B, gy), dim3(256), 0, s, x_old, x_prop, u, H0, H1,
                       2 * L * L, xform, x_new, dH, acc, obs_old, obs_new, obs_out, n_obs, B, o1, o2);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}

}  // namespace fthmc
