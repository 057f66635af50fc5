// Weight gradients of one coupling layer (training: fthmc/train.py:191-210, loss.backward() wrt the conv weights).
//
// The backward kernel (flow_bwd_gather.hip, A.gz) has written the gradients wrt the layer's pre-activations at every
// tile's own sites: gz2, gz1 (channel-minor) and g_out = (dL/ds_0, dL/ds_1, dL/dt) at the active sites; the forward
// kernel has stashed h1, h2 (channel-minor) and cos / sin of the frozen plaquettes.  A 3x3 conv's weight gradient
//     gw[co][ci][ky][kx] = sum_s gz[co][s] * hin[ci][s + (ky - 1, kx - 1)]
// is a GEMM with K = sites.  One 512-thread workgroup owns a 16 x 16 tile (hin with a halo of one site) and runs it on
// v_mfma_f64_16x16x4_f64: M = 16 = 8 co x (dy = 0, 1) with A[(co, dy)][s] = gz[co][s - dy rows], N = (ci, kx, kyb) with
// ky = 2 kyb + dy, so D[(co, dy)][(ci, kx, kyb)] = gw[co][ci][2 kyb + dy][kx] (the ky = 3 row is discarded: 75 % useful).
// conv2 (8 -> 8): 48 columns = 3 N tiles, conv1 (2 -> 8): 12 columns = 1 N tile.  The K walk (17 window rows x 4 steps) is
// split over the eight waves (two rows each, the last row's four steps on waves 0..3) and every wave runs ALL FOUR tiles on its
// slice: 32..36 MFMAs on 1.5 LDS reads each (the three conv2 tiles share their A operand), both operands addressed as lane
// part + wave-uniform row offset + compile-time constant (gz planes carry a zero row above and below: no bounds logic in the
// K loop).  conv3 (8 -> 3, active sites only) runs on the VALU in a second phase that reuses the h1 planes for h2.
// A workgroup WALKS A.tpw (chain, tile) items of its layer, A.wg_ns apart (the sum over sites simply runs on: accumulators
// stay in registers); the next item's operands are prefetched into registers in two halves, its h1 window ahead of the
// current item's MFMA phase, its gz / cos, sin / g_out behind it (all of it plus the 32 accumulator registers does not fit
// the 128 a wave has at two workgroups per CU, and a spill inside the walk costs more than anything else here: docs/history.md
// section 4).  At the end the waves' slices are summed through LDS in a fixed order and the workgroup writes ONE complete
// 955-entry partial to A.gw_part; k_reduce_gw sums the partials in a fixed order.  Replaces the in-kernel weight-gradient
// stages of round 1 (8 x 16 tiles, most of them on 1..3 waves: 49 k cycles per 128 sites).
#include "flow_mfma_common.h"


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
    // after the walk, over the planes: the waves' accumulators [8][4 tiles][4][64], bias lane sums [8][2][8], conv3 sums [432]
    static constexpr int RED = 0, RBS = RED + 8 * 4 * 4 * 64, RC3 = RBS + 8 * 2 * 8, RSIZE = RC3 + 432;
    static constexpr int SIZE = GO + 3 * NA > RSIZE ? GO + 3 * NA : RSIZE;
    static_assert(TR == 16 && TC == 16, "K split of the MFMA phase: 8 waves x 2 rows + row 16 on four of them");
    static_assert(TC % 4 == 0 && TR % 4 == 0 && NA == 64 && GO % 2 == 0, "K steps of four sites; one wave sums a g_out plane; 16-byte reads of g_out");
    static_assert(2 * SIZE * 8 <= 160 * 1024, "two workgroups per CU");
};

// One wave's slice of the K walk for ALL FOUR N tiles: the conv2 tiles nt = 0, 1, 2 share their A operand (gz2: one LDS read
// instead of three), conv1 (nt = 3) reads gz1 at the same offset.  Row r of the walk pairs gz rows r, r - 1 with hin rows
// r, r + 2.  `bsum`: lanes dy = 0 see every own site of channel co once.
struct WgradLane { int pa, pb0, pb1, pb2, pb3; };            // LDS element offsets of the lane's operands at row 0, cs = 0
template <int TC, int PSG, int W1C, int PSH>
__device__ __forceinline__ WgradLane wgrad_lane(int lane) {
    const int g = lane >> 4, i = lane & 15, co = i & 7, dy = i >> 3;   // A row m = (co, dy); B column n = (ci, kx, kyb)
    WgradLane w;
    w.pa = co * PSG + (1 - dy) * TC + g;
    auto pb = [&](int ncol, int ncols) { const int nc = ncol < ncols ? ncol : 0, ci = nc / 6, kx = (nc % 6) >> 1, kyb = nc & 1; return ci * PSH + 2 * kyb * W1C + kx + g; };
    w.pb0 = pb(i, 48); w.pb1 = pb(16 + i, 48); w.pb2 = pb(32 + i, 48); w.pb3 = pb(i, 12);
    return w;
}
__device__ __forceinline__ void wgrad_step(const double* __restrict__ gz2, const double* __restrict__ gz1, const double* __restrict__ h,
                                           const double* __restrict__ in, const WgradLane& w, int oa, int ob,
                                           double4_t (&acc)[4], double (&bsum)[2]) {
    const double a2 = gz2[w.pa + oa], a1 = gz1[w.pa + oa];
    const double b0 = h[w.pb0 + ob], b1 = h[w.pb1 + ob], b2 = h[w.pb2 + ob], b3 = in[w.pb3 + ob];
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b1, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b3, acc[3], 0, 0, 0);
    bsum[0] += a2; bsum[1] += a1;
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

// DBG: the cycle stamps of the diagnostic launches (fthmc_profile_stages kind 3) exist in their own instance only
template <int TR, int TC, bool FASTW, bool DBG>
__global__ FT_LDS_B64 __launch_bounds__(NT, 4) void k_flow_wgrad(FlowLayerArgs A) {
    using S = SmemW<TR, TC>;
    constexpr int PSG = S::PSG, PSH = S::PSH, W1C = S::W1C, NH = S::NH, NA = S::NA;
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
    double* gw0 = A.gw_part + (size_t)lz * A.gwp_lstride + (size_t)grp * FLOW_GW_STRIDE;         // the group's partial
    long long* dbg = (DBG && A.dbg) ? A.dbg + (size_t)grp * 16 : nullptr;
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
    // two halves of an item's prefetch: the h1 window (with the window addresses `hat`, which the item's h2 loads reuse) goes
    // out BEFORE the MFMA phase of the item ahead of it, the rest (gz2, gz1, cos / sin, g_out: 26 registers) behind that phase
    // -- the eight waves hold all four N tiles' accumulators there (32 registers)
    auto issue_h1 = [&](int b, int ti, int tj) {
        const Coord c = coords();
        const int i0 = ti * TR, j0 = tj * TC;
        const double* __restrict__ sh1 = uniform_ptr(stl, ((size_t)A.B * 19 + (size_t)b * 8) * n);
#pragma unroll
        for (int k = 0; k < NRH; ++k) {
            hat[k] = (unsigned)(mul24(wrap_line<FASTW>(i0 + c.hwr[k] - 1, L, wmagic), L) + wrap_line<FASTW>(j0 + c.hwc[k] - 1, L, wmagic)) * 8u
                     + 4u * (unsigned)c.hwq[k];
            hv[k][0] = ldu2(sh1, hat[k]); hv[k][1] = ldu2(sh1, hat[k] + 2);
        }
        cb = b; cti = ti; ctj = tj;
    };
    auto issue_rest = [&]() {                             // of item (cb, cti, ctj)
        const Coord c = coords();
        const int b = cb, i0 = cti * TR, j0 = ctj * TC;
        const int rmax = min(TR, L - i0), cmax = min(TC, L - j0);
        const double* __restrict__ gz2g = uniform_ptr(gzl, (size_t)b * 17 * n);
        const double* __restrict__ gz1g = gz2g + (size_t)8 * n;
        const double* __restrict__ gog = gz2g + (size_t)16 * n;
        const double* __restrict__ scs = uniform_ptr(stl, ((size_t)A.B * 18 + b) * n);
        const bool ovalid = c.orr < rmax && c.occ < cmax;
        const unsigned oat = ovalid ? (unsigned)(mul24(i0 + c.orr, L) + j0 + c.occ) * 8u + 4u * c.oq : 0u;
        z2a = ldu2(gz2g, oat); z2b = ldu2(gz2g, oat + 2); z1a = ldu2(gz1g, oat); z1b = ldu2(gz1g, oat + 2);
        {
            const unsigned ic = c.ffrozen ? (unsigned)stash_frozen_idx(wrap_line<FASTW>(i0 + c.fwr - 1, L, wmagic), wrap_line<FASTW>(j0 + c.fwc - 1, L, wmagic), L, mu, off) : 0u;
            fcs = ldu(scs, ic); fsn = ldu(scs + (n >> 1), ic);
        }
        {
            const bool avalid = c.atask && c.ar < rmax && c.ac < cmax;
            const unsigned ia = avalid ? (unsigned)stash_active_idx(i0 + c.ar, j0 + c.ac, L, mu) * 4u : 0u;
            gva = ldu2(gog, ia); gvc = ldu(gog, ia + 2);
        }
    };
    auto item_coords = [&](int item, int& b, int& ti, int& tj) {        // uniform: scalar divisions, once per item
        b = item / ntiles;
        const int t = item - b * ntiles;
        ti = t / ntj_; tj = t - ti * ntj_;
    };
    {
        int ib, iti, itj;
        item_coords(first, ib, iti, itj);
        issue_h1(ib, iti, itj);
        issue_rest();
    }

    // accumulators of the whole walk
    double4_t acc[4];                                                      // this wave's K slice of the four N tiles
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = double4_t{0.0, 0.0, 0.0, 0.0};
    double bsum[2] = {0.0, 0.0};                                           // bias sums b2, b1 of the slice: lanes dy = 0
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
            issue_h1(nb, nti, ntj);
        }

        // ---- phase 1 GEMMs: wave = K slice (window rows 2 wave, 2 wave + 1; waves 0..3 also the step (row TR, cs = wave)) of all
        //      four N tiles (0..2 conv2, 3 conv1)
        {
            const WgradLane wl = wgrad_lane<TC, PSG, W1C, PSH>(lane);
            const int oa0 = 2 * wave * TC, ob0 = 2 * wave * W1C;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int cs = 0; cs < TC / 4; ++cs)
                    wgrad_step(sGZ2, sGZ1, sHA, sIn, wl, oa0 + r * TC + 4 * cs, ob0 + r * W1C + 4 * cs, acc, bsum);
            if (wave < 4) wgrad_step(sGZ2, sGZ1, sHA, sIn, wl, TR * TC + 4 * wave, TR * W1C + 4 * wave, acc, bsum);
        }
        STAMP(3);
        issue_rest();                                                    // the next item's gz, cos / sin, g_out: they land under the phases below
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
    // ---- the group's partial: the waves' K slices summed through LDS in a fixed order
    lds_barrier();
    {
        const int lane = tid0 & 63;
        double* R = sm + S::RED; double* BS = sm + S::RBS; double* C3 = sm + S::RC3;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) R[((wave * 4 + nt) * 4 + q) * 64 + lane] = acc[nt][q];
#pragma unroll
        for (int k = 0; k < 2; ++k) {                                    // A rows (co, dy = 0): lanes co + 16 g
            double v = bsum[k];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            if (lane < 8) BS[(wave * 2 + k) * 8 + lane] = v;
        }
        if (tid0 < 432) C3[tid0] = acc3[0];
        else if (tid0 >= 448) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double v = ft_wave_sum(acc3[k]);
                if (lane == 0) gw0[CB2 + k] = v;
            }
        }
        lds_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                    // element e = (nt, q, lane) = D_nt[row g + 4 q][col i]
            const int e = tid0 + NT * h, nt = e >> 8, q = (e >> 6) & 3, ln = e & 63;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += R[w * 1024 + e];
            const int g = ln >> 4, i = ln & 15, m = g + 4 * q, co = m & 7, dy = m >> 3;
            const int ncol = (nt < 3 ? nt * 16 : 0) + i;
            if (ncol < (nt < 3 ? 48 : 12)) {
                const int ci = ncol / 6, kx = (ncol % 6) >> 1, ky = 2 * (ncol & 1) + dy;
                if (ky <= 2) gw0[(nt < 3 ? CW1 + (co * 8 + ci) * 9 : CW0 + (co * 2 + ci) * 9) + ky * 3 + kx] = v;
            }
        }
        if (tid0 < 16) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += BS[w * 16 + tid0];
            gw0[(tid0 < 8 ? CB1 : CB0) + (tid0 & 7)] = v;
        }
        if (tid0 < 216) gw0[CW2 + tid0] = C3[tid0] + C3[216 + tid0];
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
    if (wrap_fast_ok(a.L, MG_TR, MG_TC)) {
        if (b.dbg) hipLaunchKernelGGL((k_flow_wgrad<MG_TR, MG_TC, true, true>), grid, dim3(NT), 0, s, b);
        else hipLaunchKernelGGL((k_flow_wgrad<MG_TR, MG_TC, true, false>), grid, dim3(NT), 0, s, b);
    } else hipLaunchKernelGGL((k_flow_wgrad<MG_TR, MG_TC, false, false>), grid, dim3(NT), 0, s, b);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}

}  // namespace fthmc
