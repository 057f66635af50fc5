// Weight gradients of one coupling layer (training: fthmc/train.py:191-210, loss.backward() wrt the conv weights).
//
// The backward kernel (flow_bwd_gather.hip, A.gz) has written the gradients wrt the layer's pre-activations at every
// tile's own sites: gz2, gz1 (channel-minor) and g_out = (dL/ds_0, dL/ds_1, dL/dt) at the active sites; the forward
// kernel has stashed h1, h2 (channel-minor) and cos / sin of the frozen plaquettes.  A 3x3 conv's weight gradient
//     gw[co][ci][ky][kx] = sum_s gz[co][s] * hin[ci][s + (ky - 1, kx - 1)]
// is a GEMM with K = sites.  One 512-thread workgroup owns a 16 x 16 tile (hin with a halo of one site) and runs it on
// v_mfma_f64_16x16x4_f64: M = 16 = 8 co x (dy = 0, 1) with A[(co, dy)][s] = gz[co][s - dy rows], N = (ci, kx, kyb) with
// ky = 2 kyb + dy, so D[(co, dy)][(ci, kx, kyb)] = gw[co][ci][2 kyb + dy][kx] (the ky = 3 row is discarded: 75 % useful).
// conv2 (8 -> 8): 48 columns = 3 N tiles, conv1 (2 -> 8): 12 columns = 1 N tile; each of the four tiles is split over two
// waves by halves of the site rows, so all eight waves run 32..36 MFMAs with both operands addressed as lane part +
// compile-time constant (gz planes carry a zero row above and below: no bounds logic in the K loop).  conv3 (8 -> 3,
// active sites only) runs on the VALU in a second phase that reuses the h1 planes for h2.
// Every workgroup writes TWO complete 955-entry partials (site halves) to A.gw_part; k_reduce_gw sums them in a fixed
// order.  Replaces the in-kernel weight-gradient stages of round 1 (8 x 16 tiles, most of them on 1..3 waves:
// 49 k cycles per 128 sites; this kernel: see DESIGN.md section 4).
#include "flow_mfma_common.h"

namespace {

using namespace fthmc;
using namespace fthmc_flow;

typedef double double2_t __attribute__((ext_vector_type(2)));

template <int TR, int TC> struct SmemW {
    static constexpr int GR = TR + 2, NGZ = GR * TC, PSG = ps_round(NGZ);      // gz planes: rows -1 .. TR (zero rings)
    static constexpr int W1R = TR + 2, W1C = TC + 2, NH = W1R * W1C, PSH = ps_round(NH + W1C);   // hin planes: tile+1 (+ one row of slack)
    static constexpr int N3 = TR * TC;
    static constexpr int GZ2 = 0;                       // [8][PSG]   phase 2: g_out [3][N3]
    static constexpr int HA = GZ2 + 8 * PSG;            // [8][PSH] h1   phase 2: h2
    static constexpr int GZ1 = HA + 8 * PSH;            // [8][PSG]
    static constexpr int IN = GZ1 + 8 * PSG;            // [2][PSH] cos, sin (1, 0 off the frozen sites)
    static constexpr int SIZE = IN + 2 * PSH;
    static_assert(TC % 4 == 0 && 3 * N3 <= 8 * PSG, "K steps of four sites; g_out over gz2");
    static_assert(2 * SIZE * 8 <= 160 * 1024, "two workgroups per CU");
};

// one N tile x one half of the site rows: rows [R0, R1) of the K walk (row r pairs gz rows r, r - 1 with hin rows r, r + 2)
template <int TC, int PSG, int W1C, int PSH, int CIN, int R0, int R1, class Store>
__device__ __forceinline__ void wgrad_half(const double* __restrict__ gz, const double* __restrict__ hin, int nt, int lane, Store store) {
    const int g = lane >> 4, i = lane & 15;
    const int co = i & 7, dy = i >> 3;                      // A row m = (co, dy)
    const int ncol = nt * 16 + i;                           // B column n = (ci, kx, kyb)
    constexpr int NCOL = CIN * 6;
    const bool ncol_ok = ncol < NCOL;
    const int nc = ncol_ok ? ncol : 0;
    const int ci = nc / 6, kx = (nc % 6) >> 1, kyb = nc & 1;
    const double* pa = gz + co * PSG + (1 - dy) * TC + g;              // + r * TC + 4 cs
    const double* pb = hin + ci * PSH + 2 * kyb * W1C + kx + g;        // + r * W1C + 4 cs
    double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = R0; r < R1; ++r)
#pragma unroll
        for (int cs = 0; cs < TC / 4; ++cs) {
            const double av = pa[r * TC + 4 * cs], bv = pb[r * W1C + 4 * cs];
            if ((r * (TC / 4) + cs) & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc1, 0, 0, 0);
            else                         acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc0, 0, 0, 0);
        }
    const double4_t acc = acc0 + acc1;
    if (ncol_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                       // D[row = g + 4 q][col = i]
            const int m = g + 4 * q, co2 = m & 7, dy2 = m >> 3, ky = 2 * kyb + dy2;
            if (ky <= 2) store(co2, ci, ky, kx, acc[q]);
        }
    }
}

template <int TR, int TC, bool FASTW>
__global__ __launch_bounds__(NT, 4) void k_flow_wgrad(FlowLayerArgs A) {
    using S = SmemW<TR, TC>;
    constexpr int PSG = S::PSG, PSH = S::PSH, W1C = S::W1C, NH = S::NH, N3 = S::N3;
    __shared__ __attribute__((aligned(16))) double sm[S::SIZE];
    double* sGZ2 = sm + S::GZ2; double* sHA = sm + S::HA; double* sGZ1 = sm + S::GZ1; double* sIn = sm + S::IN;
    double* sGO = sm + S::GZ2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = A.L, n = L * L;
    const int nti_ = (L + TR - 1) / TR, ntj_ = (L + TC - 1) / TC;
    // several layers in one launch (A.nlb): blockIdx.z = layer * ceil(B / 8) + chain group
    const int nzb = (A.B + 7) / 8, lz = A.nlb > 0 ? (int)blockIdx.z / nzb : 0;
    const int mu = A.nlb > 0 ? (lz & 1) : A.mu, off = A.nlb > 0 ? ((lz >> 1) & 3) : A.off;
    BlockTile bt;
    if (!block_tile(A.B, nti_, ntj_, bt, A.nlb > 0 ? (int)blockIdx.z - lz * nzb : (int)blockIdx.z)) return;
    const int b = bt.b, tile = bt.tile, ntiles = nti_ * ntj_;
    const int i0 = bt.ti * TR, j0 = bt.tj * TC;
    const int rmax = min(TR, L - i0), cmax = min(TC, L - j0);
    const unsigned wmagic = FASTW ? 0u : wrap_magic(L);
    auto WI = [&](int k) { return mul24(wrap_line<FASTW>(i0 + k, L, wmagic), L); };
    auto WJ = [&](int k) { return wrap_line<FASTW>(j0 + k, L, wmagic); };
    auto ldu2 = [](const double* base, unsigned idx) {
        return *reinterpret_cast<const double2_t*>(reinterpret_cast<const char*>(base) + idx * 8u);
    };
    const double* __restrict__ gz2g = uniform_ptr((const double*)A.gz, (size_t)lz * A.gz_lstride + (size_t)b * 17 * n);
    const double* __restrict__ gz1g = gz2g + (size_t)8 * n;
    const double* __restrict__ gog = gz2g + (size_t)16 * n;
    const double* __restrict__ scs = uniform_ptr((const double*)A.stash, (size_t)lz * A.stash_lstride + ((size_t)A.B * 18 + b) * n);
    const double* __restrict__ sh1 = uniform_ptr((const double*)A.stash, (size_t)lz * A.stash_lstride + ((size_t)A.B * 19 + (size_t)b * 8) * n);
    const double* __restrict__ sh2 = uniform_ptr((const double*)A.stash, (size_t)lz * A.stash_lstride + ((size_t)A.B * 27 + (size_t)b * 8) * n);
    double* gw0 = A.gw_part + (size_t)lz * A.gwp_lstride + ((size_t)b * ntiles + tile) * 2 * FLOW_GW_STRIDE;   // the tile's two partials (site halves)

    // ---- phase 1 loads (unconditional, clamped: straight-line code keeps the waits counted)
    // own sites: thread = (site, channel quad): 32 bytes of gz2 and of gz1
    const int os = tid >> 1, oq = tid & 1, orr = fdiv<TC>(os), occ = os - orr * TC;
    const bool ovalid = orr < rmax && occ < cmax;
    const unsigned oat = ovalid ? (unsigned)(mul24(i0 + orr, L) + j0 + occ) * 8u + 4u * oq : 0u;
    const double2_t z2a = ldu2(gz2g, oat), z2b = ldu2(gz2g, oat + 2), z1a = ldu2(gz1g, oat), z1b = ldu2(gz1g, oat + 2);
    // tile+1 window of h1 (and, in phase 2, of h2): items (window site, channel quad), NH * 2 of them in two rounds
    constexpr int NIT = 2 * NH, NRH = (NIT + NT - 1) / NT;
    unsigned hat[NRH]; int hls[NRH];
    double2_t hv[NRH][2];
#pragma unroll
    for (int k = 0; k < NRH; ++k) {
        const int it = min(tid + k * NT, NIT - 1), ws = it >> 1, wq = it & 1, wr = fdiv<W1C>(ws), wc = ws - wr * W1C;
        hat[k] = (unsigned)(WI(wr - 1) + WJ(wc - 1)) * 8u + 4u * wq;
        hls[k] = tid + k * NT < NIT ? (4 * wq) * PSH + wr * W1C + wc : -1;
        hv[k][0] = ldu2(sh1, hat[k]); hv[k][1] = ldu2(sh1, hat[k] + 2);
    }
    // net input on the tile+1 window: cos / sin where the plaquette is frozen, (1, 0) elsewhere
    double fcs = 1.0, fsn = 0.0;
    int fls = -1;
    if (tid < NH) {
        const int wr = fdiv<W1C>(tid), wc = tid - wr * W1C;
        const int cls = ((mu == 0 ? j0 - 1 + wc : i0 - 1 + wr) - off) & 3;
        fls = tid;
        if (cls == 1 || cls == 2) {
            const unsigned ic = (unsigned)stash_frozen_idx(wrap_line<FASTW>(i0 + wr - 1, L, wmagic), WJ(wc - 1), L, mu, off);
            fcs = ldu(scs, ic); fsn = ldu(scs + (n >> 1), ic);
        }
    }
    // ---- phase 1 LDS fill: gz planes with zero rings (rows -1 and TR), zeros at own sites beyond the lattice
    if (tid < 2 * TC) {                                   // ring rows of all 16 planes
        const int rr = tid < TC ? 0 : TR + 1, cc = tid < TC ? tid : tid - TC;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) { sGZ2[ch * PSG + rr * TC + cc] = 0.0; sGZ1[ch * PSG + rr * TC + cc] = 0.0; }
    }
    {
        double* p2 = sGZ2 + (4 * oq) * PSG + (orr + 1) * TC + occ;
        double* p1 = sGZ1 + (4 * oq) * PSG + (orr + 1) * TC + occ;
        p2[0] = ovalid ? z2a.x : 0.0; p2[PSG] = ovalid ? z2a.y : 0.0; p2[2 * PSG] = ovalid ? z2b.x : 0.0; p2[3 * PSG] = ovalid ? z2b.y : 0.0;
        p1[0] = ovalid ? z1a.x : 0.0; p1[PSG] = ovalid ? z1a.y : 0.0; p1[2 * PSG] = ovalid ? z1b.x : 0.0; p1[3 * PSG] = ovalid ? z1b.y : 0.0;
    }
#pragma unroll
    for (int k = 0; k < NRH; ++k)
        if (hls[k] >= 0) { double* p = sHA + hls[k]; p[0] = hv[k][0].x; p[PSH] = hv[k][0].y; p[2 * PSH] = hv[k][1].x; p[3 * PSH] = hv[k][1].y; }
    if (fls >= 0) { sIn[fls] = fcs; sIn[PSH + fls] = fsn; }
    if (tid < 2 * (PSH - NH)) {                           // slack behind the windows (read by discarded ky = 3 columns only)
        const int pl = tid / (PSH - NH), e = NH + tid % (PSH - NH);
        sIn[pl * PSH + e] = 0.0;
#pragma unroll
        for (int ch = pl; ch < 8; ch += 2) sHA[ch * PSH + e] = 0.0;
    }
    // phase 2 operands: issued now, they land under the MFMA phase
#pragma unroll
    for (int k = 0; k < NRH; ++k) { hv[k][0] = ldu2(sh2, hat[k]); hv[k][1] = ldu2(sh2, hat[k] + 2); }
    // own active site `tid` (tid < N3 / 4): mu = 0 columns off + 4 m, mu = 1 rows off + 4 q
    const int ar = mu == 0 ? tid / (TC / 4) : off + 4 * (tid / TC), ac = mu == 0 ? off + 4 * (tid % (TC / 4)) : tid % TC;
    const bool atask = tid < N3 / 4, avalid = atask && ar < rmax && ac < cmax;
    double2_t gva = {0.0, 0.0}, gvb = {0.0, 0.0};
    {
        const unsigned ia = avalid ? (unsigned)stash_active_idx(i0 + ar, j0 + ac, L, mu) * 4u : 0u;
        gva = ldu2(gog, ia); gvb = ldu2(gog, ia + 2);
        if (!avalid) { gva = double2_t{0.0, 0.0}; gvb = double2_t{0.0, 0.0}; }
    }
    lds_barrier();

    // ---- phase 1 GEMMs: wave = (N tile nt = wave >> 1: 0..2 conv2, 3 conv1; half of the site rows = wave & 1)
    {
        const int nt = wave >> 1, kh = wave & 1;
        double* gw = gw0 + (size_t)kh * FLOW_GW_STRIDE;
        constexpr int RH = (TR + 2) / 2;                  // rows 0 .. TR of the K walk: [0, RH) and [RH, TR + 1)
        if (nt < 3) {
            auto st = [&](int co, int ci, int ky, int kx, double v) { gw[CW1 + (co * 8 + ci) * 9 + ky * 3 + kx] = v; };
            if (kh == 0) wgrad_half<TC, PSG, W1C, PSH, 8, 0, RH>(sGZ2, sHA, nt, lane, st);
            else         wgrad_half<TC, PSG, W1C, PSH, 8, RH, TR + 1>(sGZ2, sHA, nt, lane, st);
        } else {
            auto st = [&](int co, int ci, int ky, int kx, double v) { gw[CW0 + (co * 2 + ci) * 9 + ky * 3 + kx] = v; };
            if (kh == 0) wgrad_half<TC, PSG, W1C, PSH, 2, 0, RH>(sGZ1, sIn, 0, lane, st);
            else         wgrad_half<TC, PSG, W1C, PSH, 2, RH, TR + 1>(sGZ1, sIn, 0, lane, st);
        }
        // biases b2, b1: wave w sums channel w of gz2 and of gz1 (the rings are zeros)
        double a2 = 0.0, a1 = 0.0;
        for (int e = lane; e < S::NGZ; e += 64) { a2 += sGZ2[wave * PSG + e]; a1 += sGZ1[wave * PSG + e]; }
        a2 = ft_wave_sum(a2); a1 = ft_wave_sum(a1);
        if (lane == 0) { gw0[CB1 + wave] = a2; gw0[CB0 + wave] = a1; gw0[FLOW_GW_STRIDE + CB1 + wave] = 0.0; gw0[FLOW_GW_STRIDE + CB0 + wave] = 0.0; }
    }
    lds_barrier();

    // ---- phase 2: h2 over h1, g_out over gz2; conv3 weight gradient on the VALU
#pragma unroll
    for (int k = 0; k < NRH; ++k)
        if (hls[k] >= 0) { double* p = sHA + hls[k]; p[0] = hv[k][0].x; p[PSH] = hv[k][0].y; p[2 * PSH] = hv[k][1].x; p[3 * PSH] = hv[k][1].y; }
    if (atask) { sGO[tid] = gva.x; sGO[N3 / 4 + tid] = gva.y; sGO[2 * (N3 / 4) + tid] = gvb.x; }    // [3][N3 / 4], own active sites in task order
    lds_barrier();
    {
        // thread = (output (co, ci, tap), half of the active sites); threads beyond 2 * 216 idle
        constexpr int NA = N3 / 4, NAH = NA / 2;
        const int hf = tid >= 216 ? 1 : 0, t = tid - 216 * hf;
        if (tid < 432) {
            const int co = t / 72, ci = (t / 9) % 8, tap = t % 9, ky = tap / 3, kx = tap % 3;
            const double* ph = sHA + ci * PSH + ky * W1C + kx;            // h2 at own (r, c) + (ky - 1, kx - 1): ph[r * W1C + c]
            const double* pg = sGO + co * NA + hf * NAH;
            double acc = 0.0;
#pragma unroll 8
            for (int a = 0; a < NAH; ++a) {
                const int aa = hf * NAH + a;
                const int r = mu == 0 ? aa / (TC / 4) : off + 4 * (aa / TC), c = mu == 0 ? off + 4 * (aa % (TC / 4)) : aa % TC;
                acc = fma(pg[a], ph[r * W1C + c], acc);                  // g_out is 0 at active sites beyond the lattice
            }
            gw0[(size_t)hf * FLOW_GW_STRIDE + CW2 + t] = acc;
        } else if (tid < 432 + 3) {                                      // b3
            const int k = tid - 432;
            double a_ = 0.0;
            for (int a = 0; a < NA; ++a) a_ += sGO[k * NA + a];
            gw0[CB2 + k] = a_; gw0[FLOW_GW_STRIDE + CB2 + k] = 0.0;
        }
    }
}

}  // namespace

namespace fthmc {

int launch_flow_wgrad(const FlowLayerArgs& a, hipStream_t s) {
    dim3 grid = xcd_grid(a.B, (a.L + MG_TR - 1) / MG_TR, (a.L + MG_TC - 1) / MG_TC);
    if (a.nlb > 0) grid.z *= a.nlb;
    if (wrap_fast_ok(a.L, MG_TR, MG_TC)) hipLaunchKernelGGL((k_flow_wgrad<MG_TR, MG_TC, true>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((k_flow_wgrad<MG_TR, MG_TC, false>), grid, dim3(NT), 0, s, a);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}

}  // namespace fthmc
