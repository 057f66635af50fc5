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
// A workgroup WALKS A.tpw (chain, tile) items of its layer, A.wg_ns apart (the sum over sites simply runs on: accumulators
// stay in registers) with the next item's operands prefetched into registers while the current one is in the MFMA phase:
// one tile per workgroup spent 8 k of its 21 k cycles issuing loads and waiting for them (profiles/r04_workgroup_lifetime.txt).
// At the end it writes TWO complete 955-entry partials (site halves) to A.gw_part; k_reduce_gw sums them in a fixed
// order.  Replaces the in-kernel weight-gradient stages of round 1 (8 x 16 tiles, most of them on 1..3 waves:
// 49 k cycles per 128 sites; this kernel: see DESIGN.md section 4).
#include "flow_mfma_common.h"

#ifndef FT_WGRAD_TEST
#define FT_WGRAD_TEST 0
#endif

namespace {

using namespace fthmc;
using namespace fthmc_flow;

typedef double double2_t __attribute__((ext_vector_type(2)));

template <int TR, int TC> struct SmemW {
    static constexpr int GR = TR + 2, NGZ = GR * TC, PSG = ps_round(NGZ);      // gz planes: rows -1 .. TR (zero rings)
    // hin planes: tile+1 (+ one row of slack).  Plane stride = 12 (mod 32): the B operand reads of every N tile (lanes = (ci, kx + g,
    // kyb): 18 ci + 4 kyb + kx + g with the stride 18 of ps_round put ci = 2 on the banks of ci = 0, kyb = 1) and the fill's
    // writes (channel quads 4 planes apart) are both free of bank conflicts
    static constexpr int W1R = TR + 2, W1C = TC + 2, NH = W1R * W1C, PSH = ((NH + W1C - 12 + 31) / 32) * 32 + 12;
    static constexpr int N3 = TR * TC, NA = N3 / 4;
    static constexpr int GZ2 = 0;                       // [8][PSG]
    static constexpr int HA = GZ2 + 8 * PSG;            // [8][PSH] h1   phase 2: h2
    static constexpr int GZ1 = HA + 8 * PSH;            // [8][PSG]
    static constexpr int IN = GZ1 + 8 * PSG;            // [2][PSH] cos, sin (1, 0 off the frozen sites)
    static constexpr int GO = IN + 2 * PSH;             // [3][NA] g_out at the own active sites, task order
    static constexpr int SIZE = GO + 3 * NA;
    static_assert(TC % 4 == 0 && TR % 4 == 0 && NA == 64 && GO % 2 == 0, "K steps of four sites; one wave sums a g_out plane; 16-byte reads of g_out");
    static_assert(2 * SIZE * 8 <= 160 * 1024, "two workgroups per CU");
};

// one N tile x one half of the site rows: rows [R0, R1) of the K walk (row r pairs gz rows r, r - 1 with hin rows r, r + 2)
template <int TC, int PSG, int W1C, int PSH, int CIN, int R0, int R1, bool BIAS>
__device__ __forceinline__ void wgrad_half(const double* __restrict__ gz, const double* __restrict__ hin, int nt, int lane,
                                           double4_t& acc0, double4_t& acc1, double& asum) {
    const int g = lane >> 4, i = lane & 15;
    const int co = i & 7, dy = i >> 3;                      // A row m = (co, dy)
    const int ncol = nt * 16 + i;                           // B column n = (ci, kx, kyb)
    constexpr int NCOL = CIN * 6;
    const int nc = ncol < NCOL ? ncol : 0;
    const int ci = nc / 6, kx = (nc % 6) >> 1, kyb = nc & 1;
    const double* pa = gz + co * PSG + (1 - dy) * TC + g;              // + r * TC + 4 cs
    const double* pb = hin + ci * PSH + 2 * kyb * W1C + kx + g;        // + r * W1C + 4 cs
#pragma unroll
    for (int r = R0; r < R1; ++r)
#pragma unroll
        for (int cs = 0; cs < TC / 4; ++cs) {
#if FT_WGRAD_TEST == 1                                                  // timing only: no B operand reads
            const double av = pa[r * TC + 4 * cs], bv = av;
#else
            const double av = pa[r * TC + 4 * cs], bv = pb[r * W1C + 4 * cs];
#endif
#if FT_WGRAD_TEST == 2                                                  // timing only: the reads without the MFMAs
            if ((r * (TC / 4) + cs) & 1) acc1[0] += av * bv; else acc0[0] += av + bv;
#else
            if ((r * (TC / 4) + cs) & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc1, 0, 0, 0);
            else                         acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc0, 0, 0, 0);
#endif
            if (BIAS) asum += av;                           // lanes dy = 0: the A operand walks every own site of channel co once
        }
}
// the lane's four results of an N tile: D[row = g + 4 q][col = i] = gw[co][ci][2 kyb + dy][kx]
template <int CIN, class Store>
__device__ __forceinline__ void wgrad_store(const double4_t& acc, int nt, int lane, Store store) {
    const int g = lane >> 4, i = lane & 15, ncol = nt * 16 + i;
    if (ncol < CIN * 6) {
        const int ci = ncol / 6, kx = (ncol % 6) >> 1, kyb = ncol & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = g + 4 * q, co2 = m & 7, dy2 = m >> 3, ky = 2 * kyb + dy2;
            if (ky <= 2) store(co2, ci, ky, kx, acc[q]);
        }
    }
}
// conv3 (8 -> 3, active sites only): thread = (output (co, ci, tap), half of the active sites); the 32 sites of the half
// at compile-time offsets from the thread's base (the stripe offset `off` and the half are folded into `ph`)
template <int MU, int TC, int W1C>
__device__ __forceinline__ void conv3_acc(const double* __restrict__ pg, const double* __restrict__ ph, double (&acc)[4]) {
#pragma unroll
    for (int a = 0; a < 32; a += 2) {
        const double2_t g2 = *reinterpret_cast<const double2_t*>(pg + a);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int aa = a + e;
            const int o = MU == 0 ? (aa / (TC / 4)) * W1C + 4 * (aa % (TC / 4)) : 4 * (aa / TC) * W1C + aa % TC;
            acc[aa & 3] = fma(e ? g2.y : g2.x, ph[o], acc[aa & 3]);
        }
    }
}

template <int TR, int TC, bool FASTW>
__global__ __launch_bounds__(NT, 4) void k_flow_wgrad(FlowLayerArgs A) {
    using S = SmemW<TR, TC>;
    constexpr int PSG = S::PSG, PSH = S::PSH, W1C = S::W1C, NH = S::NH, N3 = S::N3, NA = S::NA;
    __shared__ __attribute__((aligned(16))) double sm[S::SIZE];
    double* sGZ2 = sm + S::GZ2; double* sHA = sm + S::HA; double* sGZ1 = sm + S::GZ1; double* sIn = sm + S::IN;
    double* sGO = sm + S::GO;

    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int L = A.L, n = L * L;
    const int nti_ = (L + TR - 1) / TR, ntj_ = (L + TC - 1) / TC, ntiles = nti_ * ntj_;
    // The walk is STRIDED: the ns workgroups that are resident together on an XCD (slot s of a round) stand on ns consecutive
    // tiles at every step and move on by ns tiles -- at L = 256 four whole tile rows of a chain per step, so that the halo lines a
    // tile shares with its neighbours are fetched once into the XCD's L2 (walking CONSECUTIVE tiles, every halo came from HBM
    // again: 781 MB per launch at the config-5 shard against 573 MB for one tile per workgroup).
    // grid: x = 8 XCDs (blockIdx.x % 8) x rounds x ns; round kr = xcd * R + r covers items [kr * tpw * ns, (kr + 1) * tpw * ns)
    const int items = A.B * ntiles, tpw = A.tpw, ns = A.wg_ns;
    const int KR = (items + tpw * ns - 1) / (tpw * ns), R = (KR + 7) >> 3;
    const int idx = (int)blockIdx.x >> 3, r_ = idx / ns, s_ = idx - r_ * ns, kr = ((int)blockIdx.x & 7) * R + r_;
    const int first = kr * tpw * ns + s_;
    if (r_ >= R || first >= items) return;
    const int grp = kr * ns + s_;                                         // valid groups are a prefix of this numbering
    const int nwalk = min(tpw, (items - first + ns - 1) / ns);
    const int lz = (int)blockIdx.y;
    const int mu = A.nlb > 0 ? (lz & 1) : A.mu, off = A.nlb > 0 ? ((lz >> 1) & 3) : A.off;
    const unsigned wmagic = FASTW ? 0u : wrap_magic(L);
    auto ldu2 = [](const double* base, unsigned idx) {
        return *reinterpret_cast<const double2_t*>(reinterpret_cast<const char*>(base) + idx * 8u);
    };
    const double* __restrict__ gzl = uniform_ptr((const double*)A.gz, (size_t)lz * A.gz_lstride);
    const double* __restrict__ stl = uniform_ptr((const double*)A.stash, (size_t)lz * A.stash_lstride);
    double* gw0 = A.gw_part + (size_t)lz * A.gwp_lstride + (size_t)grp * 2 * FLOW_GW_STRIDE;     // the group's two partials (site halves)
    long long* dbg = A.dbg ? A.dbg + (size_t)grp * 16 : nullptr;
#define STAMP(k) do { if (dbg && tid0 == 0) dbg[k] = (long long)__builtin_readcyclecounter(); } while (0)
    STAMP(8);                                                              // 8 -> 9 prologue, 9 -> 10 the walk, 10 -> 11 epilogue

    // ---- per-thread coordinates that do not depend on the item: recomputed from an opaque copy of the thread index where they
    //      are used (twice per item, ~40 VALU) -- kept in registers across the walk they are 13 of the 128 a wave has, and spill
    constexpr int NIT = 2 * NH, NRH = (NIT + NT - 1) / NT;
    struct Coord {
        int oq, orr, occ;                 // own sites: thread = (site, channel quad): 32 bytes of gz2 and of gz1
        int hwr[NRH], hwc[NRH], hwq[NRH], hls[NRH];   // tile+1 window of h1 / h2: tasks (window site, channel quad) in two rounds; LDS slot or -1
        int fwr, fwc; bool ftask, ffrozen;            // net input on the tile+1 window (thread = window site): frozen stripe classes 1, 2
        int ar, ac; bool atask;           // own active site `tid` (tid < NA): mu = 0 columns off + 4 m, mu = 1 rows off + 4 q
    };
    auto coords = [&]() {
        int t0 = tid0;
        asm volatile("" : "+v"(t0));
        Coord c;
        const int os = t0 >> 1;
        c.oq = t0 & 1; c.orr = fdiv<TC>(os); c.occ = os - c.orr * TC;
#pragma unroll
        for (int k = 0; k < NRH; ++k) {
            const int t = min(t0 + k * NT, NIT - 1), ws = t >> 1;
            c.hwq[k] = t & 1; c.hwr[k] = fdiv<W1C>(ws); c.hwc[k] = ws - c.hwr[k] * W1C;
            c.hls[k] = t0 + k * NT < NIT ? (4 * c.hwq[k]) * PSH + c.hwr[k] * W1C + c.hwc[k] : -1;
        }
        c.fwr = fdiv<W1C>(min(t0, NH - 1)); c.fwc = min(t0, NH - 1) - c.fwr * W1C;
        const int fl = ((mu == 0 ? c.fwc : c.fwr) - 1 - off) & 3;      // tile origins are multiples of 4
        c.ftask = t0 < NH; c.ffrozen = c.ftask && (fl == 1 || fl == 2);
        c.ar = mu == 0 ? t0 / (TC / 4) : off + 4 * (t0 / TC); c.ac = mu == 0 ? off + 4 * (t0 % (TC / 4)) : t0 % TC;
        c.atask = t0 < NA;
        return c;
    };

    // ---- zeros that stay: ring rows of the gz planes (rows -1 and TR), slack behind the hin windows
    if (tid0 < 2 * TC) {
        const int rr = tid0 < TC ? 0 : TR + 1, cc = tid0 < TC ? tid0 : tid0 - TC;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) { sGZ2[ch * PSG + rr * TC + cc] = 0.0; sGZ1[ch * PSG + rr * TC + cc] = 0.0; }
    }
    if (tid0 < 2 * (PSH - NH)) {                          // read by discarded ky = 3 columns only
        const int pl = tid0 / (PSH - NH), e = NH + tid0 % (PSH - NH);
        sIn[pl * PSH + e] = 0.0;
#pragma unroll
        for (int ch = pl; ch < 8; ch += 2) sHA[ch * PSH + e] = 0.0;
    }

    // ---- the item's operands in registers (every load unconditional from a clamped address: straight-line code keeps the
    //      waits counted)
    double2_t z2a, z2b, z1a, z1b, hv[NRH][2], gva;
    double fcs, fsn, gvc;
    unsigned hat[NRH];
    int cb, cti, ctj;                                     // the item the registers hold: chain, tile row, tile column
    auto issue = [&](int b, int ti, int tj) {
        const Coord c = coords();
        const int i0 = ti * TR, j0 = tj * TC;
        const int rmax = min(TR, L - i0), cmax = min(TC, L - j0);
        const double* __restrict__ gz2g = uniform_ptr(gzl, (size_t)b * 17 * n);
        const double* __restrict__ gz1g = gz2g + (size_t)8 * n;
        const double* __restrict__ gog = gz2g + (size_t)16 * n;
        const double* __restrict__ scs = uniform_ptr(stl, ((size_t)A.B * 18 + b) * n);
        const double* __restrict__ sh1 = uniform_ptr(stl, ((size_t)A.B * 19 + (size_t)b * 8) * n);
        const bool ovalid = c.orr < rmax && c.occ < cmax;
        const unsigned oat = ovalid ? (unsigned)(mul24(i0 + c.orr, L) + j0 + c.occ) * 8u + 4u * c.oq : 0u;
        z2a = ldu2(gz2g, oat); z2b = ldu2(gz2g, oat + 2); z1a = ldu2(gz1g, oat); z1b = ldu2(gz1g, oat + 2);
#pragma unroll
        for (int k = 0; k < NRH; ++k) {
            hat[k] = (unsigned)(mul24(wrap_line<FASTW>(i0 + c.hwr[k] - 1, L, wmagic), L) + wrap_line<FASTW>(j0 + c.hwc[k] - 1, L, wmagic)) * 8u
                     + 4u * (unsigned)c.hwq[k];
            hv[k][0] = ldu2(sh1, hat[k]); hv[k][1] = ldu2(sh1, hat[k] + 2);
        }
        {
            const unsigned ic = c.ffrozen ? (unsigned)stash_frozen_idx(wrap_line<FASTW>(i0 + c.fwr - 1, L, wmagic), wrap_line<FASTW>(j0 + c.fwc - 1, L, wmagic), L, mu, off) : 0u;
            fcs = ldu(scs, ic); fsn = ldu(scs + (n >> 1), ic);
        }
        {
            const bool avalid = c.atask && c.ar < rmax && c.ac < cmax;
            const unsigned ia = avalid ? (unsigned)stash_active_idx(i0 + c.ar, j0 + c.ac, L, mu) * 4u : 0u;
            gva = ldu2(gog, ia); gvc = ldu(gog, ia + 2);
        }
        cb = b; cti = ti; ctj = tj;
    };
    auto item_coords = [&](int item, int& b, int& ti, int& tj) {        // uniform: scalar divisions, once per item
        b = item / ntiles;
        const int t = item - b * ntiles;
        ti = t / ntj_; tj = t - ti * ntj_;
    };
    {
        int ib, iti, itj;
        item_coords(first, ib, iti, itj);
        issue(ib, iti, itj);
    }

    // accumulators of the whole walk
    double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};    // this wave's N tile x site half
    double asum = 0.0;                                                     // bias sums: lanes dy = 0 of the waves nt = 0 (b2) and nt = 3 (b1)
    double acc3[3] = {0.0, 0.0, 0.0};                                      // conv3: thread = (output, site half) in [0]; wave 7: b3 lane partials

    STAMP(9);
    for (int it = 0; it < nwalk; ++it) {
        STAMP(0);
        // opaque copies of the thread coordinates: what the stages derive from them is recomputed per item instead of living
        // in registers across the walk
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        // ---- phase 1 LDS fill from the registers; own sites and active sites beyond the lattice are zeros
        const Coord c = coords();
        {
            const int rmax = min(TR, L - cti * TR), cmax = min(TC, L - ctj * TC);
            const bool ovalid = c.orr < rmax && c.occ < cmax;
            double* p2 = sGZ2 + (4 * c.oq) * PSG + (c.orr + 1) * TC + c.occ;
            double* p1 = sGZ1 + (4 * c.oq) * PSG + (c.orr + 1) * TC + c.occ;
            p2[0] = ovalid ? z2a.x : 0.0; p2[PSG] = ovalid ? z2a.y : 0.0; p2[2 * PSG] = ovalid ? z2b.x : 0.0; p2[3 * PSG] = ovalid ? z2b.y : 0.0;
            p1[0] = ovalid ? z1a.x : 0.0; p1[PSG] = ovalid ? z1a.y : 0.0; p1[2 * PSG] = ovalid ? z1b.x : 0.0; p1[3 * PSG] = ovalid ? z1b.y : 0.0;
#pragma unroll
            for (int k = 0; k < NRH; ++k)
                if (c.hls[k] >= 0) { double* p = sHA + c.hls[k]; p[0] = hv[k][0].x; p[PSH] = hv[k][0].y; p[2 * PSH] = hv[k][1].x; p[3 * PSH] = hv[k][1].y; }
            if (c.ftask) { sIn[tid0] = c.ffrozen ? fcs : 1.0; sIn[PSH + tid0] = c.ffrozen ? fsn : 0.0; }
            if (c.atask) {
                const bool avalid = c.ar < rmax && c.ac < cmax;
                sGO[tid0] = avalid ? gva.x : 0.0; sGO[NA + tid0] = avalid ? gva.y : 0.0; sGO[2 * NA + tid0] = avalid ? gvc : 0.0;
            }
        }
        // phase 2 operands of this item (h2 on the same window): issued now, they land under the MFMA phase
        {
            const double* __restrict__ sh2 = uniform_ptr(stl, ((size_t)A.B * 27 + (size_t)cb * 8) * n);
#pragma unroll
            for (int k = 0; k < NRH; ++k) { hv[k][0] = ldu2(sh2, hat[k]); hv[k][1] = ldu2(sh2, hat[k] + 2); }
        }
        STAMP(1);
        lds_barrier();
        STAMP(2);
        double2_t h2v[NRH][2];
#pragma unroll
        for (int k = 0; k < NRH; ++k) { h2v[k][0] = hv[k][0]; h2v[k][1] = hv[k][1]; }
        // ---- the next item's phase 1 operands (the last item loads itself again: no branch around the loads)
        {
            int nb, nti, ntj;
            item_coords(first + (it + 1 < nwalk ? it + 1 : it) * ns, nb, nti, ntj);
            issue(nb, nti, ntj);
        }

        // ---- phase 1 GEMMs: wave = (N tile nt = wave >> 1: 0..2 conv2, 3 conv1; half of the site rows = wave & 1)
        {
            const int nt = wave >> 1, kh = wave & 1;
            constexpr int RH = (TR + 2) / 2;              // rows 0 .. TR of the K walk: [0, RH) and [RH, TR + 1)
            if (nt < 3) {
                if (nt == 0) {
                    if (kh == 0) wgrad_half<TC, PSG, W1C, PSH, 8, 0, RH, true>(sGZ2, sHA, 0, lane, acc0, acc1, asum);
                    else         wgrad_half<TC, PSG, W1C, PSH, 8, RH, TR + 1, true>(sGZ2, sHA, 0, lane, acc0, acc1, asum);
                } else {
                    if (kh == 0) wgrad_half<TC, PSG, W1C, PSH, 8, 0, RH, false>(sGZ2, sHA, nt, lane, acc0, acc1, asum);
                    else         wgrad_half<TC, PSG, W1C, PSH, 8, RH, TR + 1, false>(sGZ2, sHA, nt, lane, acc0, acc1, asum);
                }
            } else {
                if (kh == 0) wgrad_half<TC, PSG, W1C, PSH, 2, 0, RH, true>(sGZ1, sIn, 0, lane, acc0, acc1, asum);
                else         wgrad_half<TC, PSG, W1C, PSH, 2, RH, TR + 1, true>(sGZ1, sIn, 0, lane, acc0, acc1, asum);
            }
        }
        STAMP(3);
        lds_barrier();
        STAMP(4);

        // ---- phase 2: h2 over h1; conv3 weight gradient on the VALU
        const Coord c2 = coords();
#pragma unroll
        for (int k = 0; k < NRH; ++k)
            if (c2.hls[k] >= 0) { double* p = sHA + c2.hls[k]; p[0] = h2v[k][0].x; p[PSH] = h2v[k][0].y; p[2 * PSH] = h2v[k][1].x; p[3 * PSH] = h2v[k][1].y; }
        lds_barrier();
        STAMP(5);
        if (tid < 432) {
            const int hf = tid >= 216 ? 1 : 0, t = tid - 216 * hf;
            const int co = fdiv<9>(fdiv<8>(t)), ci = fdiv<9>(t) & 7, tap = t - fdiv<9>(t) * 9, ky = fdiv<3>(tap), kx = tap - 3 * ky;
            const double* pg = sGO + co * NA + hf * (NA / 2);
            // h2 at own (r, c) + (ky - 1, kx - 1): window index (r + ky) * W1C + c + kx
            const double* ph = sHA + ci * PSH + ky * W1C + kx;
            double c3[4] = {0.0, 0.0, 0.0, 0.0};
            if (mu == 0) conv3_acc<0, TC, W1C>(pg, ph + hf * (NA / 2 / (TC / 4)) * W1C + off, c3);
            else         conv3_acc<1, TC, W1C>(pg, ph + (off + 4 * hf * (NA / 2 / TC)) * W1C, c3);
            acc3[0] += (c3[0] + c3[1]) + (c3[2] + c3[3]);
        } else if (tid >= 448) {                                         // b3: the idle wave sums the three g_out planes
#pragma unroll
            for (int k = 0; k < 3; ++k) acc3[k] += sGO[k * NA + lane];
        }
        STAMP(6);
        if (it + 1 < nwalk) lds_barrier();                               // the next item refills the planes
    }

    STAMP(10);
    // ---- the group's partials
    {
        const int lane = tid0 & 63, nt = wave >> 1, kh = wave & 1;
        double* gw = gw0 + (size_t)kh * FLOW_GW_STRIDE;
        const double4_t acc = acc0 + acc1;
        if (nt < 3) wgrad_store<8>(acc, nt, lane, [&](int co, int ci, int ky, int kx, double v) { gw[CW1 + (co * 8 + ci) * 9 + ky * 3 + kx] = v; });
        else        wgrad_store<2>(acc, 0, lane, [&](int co, int ci, int ky, int kx, double v) { gw[CW0 + (co * 2 + ci) * 9 + ky * 3 + kx] = v; });
        // biases b2 (waves nt = 0), b1 (waves nt = 3): the lane partials of A rows (co, dy = 0) over the four K lane groups
        asum += __shfl_xor(asum, 16); asum += __shfl_xor(asum, 32);
        if ((nt == 0 || nt == 3) && lane < 8) gw[(nt == 0 ? CB1 : CB0) + lane] = asum;
        if (tid0 < 432) gw0[(size_t)(tid0 >= 216 ? 1 : 0) * FLOW_GW_STRIDE + CW2 + (tid0 >= 216 ? tid0 - 216 : tid0)] = acc3[0];
        else if (tid0 >= 448) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double v = ft_wave_sum(acc3[k]);
                if (lane == 0) { gw0[CB2 + k] = v; gw0[FLOW_GW_STRIDE + CB2 + k] = 0.0; }
            }
        }
    }
    STAMP(11);
#undef STAMP
}

}  // namespace

namespace fthmc {

int launch_flow_wgrad(const FlowLayerArgs& a, hipStream_t s) {
    FlowLayerArgs b = a;
    b.tpw = a.tpw > 0 ? a.tpw : 1;
    b.wg_ns = flow_wgrad_ns(a.B, a.L, b.tpw);
    const int items = a.B * FlowGeom{MG_TR, MG_TC}.ntiles(a.L);
    const int KR = (items + b.tpw * b.wg_ns - 1) / (b.tpw * b.wg_ns), R = (KR + 7) / 8;
    const dim3 grid(8 * R * b.wg_ns, a.nlb > 0 ? a.nlb : 1, 1);
    if (wrap_fast_ok(a.L, MG_TR, MG_TC)) hipLaunchKernelGGL((k_flow_wgrad<MG_TR, MG_TC, true>), grid, dim3(NT), 0, s, b);
    else hipLaunchKernelGGL((k_flow_wgrad<MG_TR, MG_TC, false>), grid, dim3(NT), 0, s, b);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}

}  // namespace fthmc
