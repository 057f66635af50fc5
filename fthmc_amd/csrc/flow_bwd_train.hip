// Training backward of one coupling layer WITH its weight gradients (round 6): k_flow_bwd_gather's adjoint walk and
// k_flow_wgrad's GEMMs in one kernel.
//
// The two-kernel form (flow_bwd_gather.hip writing gz2, gz1, g_out -- 17 doubles per site -- for flow_wgrad.hip to read back
// beside the stashed h1, h2 windows) moved 109 doubles per site and layer through HBM in a training step (forward 35 written,
// backward 19 read + 17 written, weight gradients 38 read): at the config-5 shard 29 GB per step.  The pre-activation
// gradients are BORN in the backward's LDS planes -- gz2 on the tile+2 window after conv3^T, gz1 on tile+1 after conv2^T's
// epilogue, g_out on tile+3 after the transform adjoint -- and the weight gradient
//     gw[co][ci][ky][kx] = sum_s gz[co][s] * hin[ci][s + (ky - 1, kx - 1)]
// needs them at the tile's OWN sites only: here the GEMMs read them where they lie.  What is left of the traffic is the h1, h2
// windows (stashed by the forward; 20 doubles per site with the halo): 35 + 39 instead of 35 + 74.
//
// One 512-thread workgroup per CU (141 KB of LDS, 241 VGPRs) WALKS (chain, tile) items of its layer -- the weight gradient's
// sum over sites simply runs on, its accumulators (four 16 x 16 MFMA tiles per wave) stay in registers -- and writes ONE
// 955-entry partial at the end (k_reduce_gw sums the partials in a fixed order: bit-deterministic).  Per item the stages are
// k_flow_bwd_gather's, each sharing its barrier interval with the weight-gradient work that reads the planes the stage has
// just consumed or produced:
//     1  transform adjoint -> g_out, the (cos, sin) window into LDS
//     2  conv3^T (MFMA, 3 steps per tile) -> gz2 | the h1 window into LDS
//     3  conv2^T (MFMA) -> gz1                 | conv2 weight gradient (MFMA: M = 8 co x 2 shifts along the walk from gz2, N = (ci, tap across
//                                              |   the walk, tap along it in {0, 2}) from the h1 window; K = the tile's twelve LIVE stripe
//                                              |   lines -- gz2 is an exact 0 on every fourth), the h2 window into LDS
//     4  conv1^T (VALU)                        | conv1 weight gradient (MFMA from gz1 and the net-input window)
//     (1 of the next item)                     | conv3 weight gradient (VALU from g_out, compact per item parity, and the h2 window)
// so a tile costs FOUR barriers (the backward's five less the one between conv1^T's channel partials and the store: the
// thread of an own frozen site sums its partials itself).  The GEMM maps are k_flow_wgrad's (flow_wgrad.hip: 75 % useful
// MACs, the K walk of 17 window rows split over the eight waves, bias sums from the A operand); their A operand comes from
// the backward's planes, whose rows above and below the tile hold the neighbours' values where k_flow_wgrad had zero rings:
// the two K steps that would pair them (walk row 0 for the lower row shift, walk row 16 for the upper) select 0 instead.
// One workgroup per CU has no second workgroup to hide its loads behind: every group of an item's operands is issued for
// item i + 1 right behind the stage of item i that consumed the group (registers; barriers wait for LDS only).
// Measured (round 6, config-5 shard, 32 chains of L = 256 per launch; profiles/r06_ab_train_fused_backward.txt,
// r06_ab_train_fused_vs_two_kernels.txt): fthmc_train_grad 7.11 ms against 8.41-8.43 ms of the two-kernel form on the same
// device, the kernel 257 us per launch against 175 + 159; then 7.00 with conv2's GEMM on the live lines (stage 3: 8.5 k ->
// 7.7-7.9 k cycles per item for 92 MFMAs per SIMD instead of 105) and 6.9 with the layers' partials reduced in one go (api.hip
// force_gp) -- per item 17.3 k cycles of which the matrix pipe is busy 7.4 k: the VALU / LDS stages of a lone workgroup run
// beside an idle matrix pipe.  What
// did NOT take that (same file): pieces of the weight-gradient GEMMs moved into the VALU stages (both waves of a SIMD run the
// same stage: nothing overlaps, +2 %), the two waves of a SIMD in opposite order (the kernel outgrows the instruction cache:
// +12 %), gz2 planes at the conflict-free stride for the GEMM's A operand, MFMA operand reads pipelined ahead in registers of
// their own, the store's partial reads batched (all +-0).
// -DFT_BT_STAMPS prints the stage cycles.
//
// Built for the tiled-exactly shapes (L a power of two >= 32: every BASELINE training shape); anything else keeps the
// two-kernel form.  Reference: loss.backward() of fthmc/train.py:191-210 through GaugeEquivCouplingLayer.forward
// (fthmc/utils/layers.py:196-202,348-371) and make_conv_net (:138-167).
#include "flow_mfma_common.h"

#if !FT_RECOMP_D1       // the act'(z1)-recompute build (an A/B switch of the force path) keeps the two-kernel form
namespace {

using namespace fthmc;
using namespace fthmc_flow;

typedef double double2_t __attribute__((ext_vector_type(2)));
constexpr int cmax_(int a, int b) { return a > b ? a : b; }

template <int TR, int TC> struct SmemT {
    // the backward's windows (flow_bwd_gather.hip SmemG)
    static constexpr int W3R = TR + 6, W3C = TC + 6, N3W = W3R * W3C;   // g_out window (active sites only)
    static constexpr int W2R = TR + 4, W2C = TC + 4, N2W = W2R * W2C;   // act'(z2) -> gz2
    static constexpr int W1R = TR + 2, W1C = TC + 2, N1W = W1R * W1C;   // act'(z1) -> gz1; h1, h2, (cos, sin)
    static constexpr int N3 = TR * TC, NA = N3 / 4;
    static constexpr int RS2 = W2C + 1;
    static constexpr int PS2 = ps_round16(W2R * RS2), PS1 = ps_round(N1W);
    // h1 / h2 / net-input planes on tile+1 (+ one row of slack): stride = 12 (mod 32) as in k_flow_wgrad (its B operand reads
    // and the fill's writes are free of bank conflicts)
    static constexpr int NH = N1W, PSH = ((NH + W1C - 12 + 31) / 32) * 32 + 12;
    static constexpr int NLC = (W3C + 3) / 4, NLR = (W3R + 3) / 4;
    static constexpr int NSLOT = cmax_(W3R * NLC, NLR * W3C);           // transform tasks
    static constexpr int NTT = (NSLOT + 63) / 64 * 64;                  // threads that run them (last waves)
    static constexpr int GO = 0;                                        // [3][N3W] g(s0, s1, t) on the tile+3 window
    static constexpr int GOC = GO + 3 * N3W;                            // [2][3][NA] the same at the own active sites, task order; one buffer per item parity
    static constexpr int GZ2 = GOC + 2 * 3 * NA;                        // [8][PS2] gz2; later conv1^T's channel partials
    static constexpr int D1 = GZ2 + 8 * PS2;                            // [8][PS1] gz1
    static constexpr int IN = D1 + 8 * PS1;                             // [2][2][PSH] cos, sin on tile+1 ((1, 0) off the frozen sites), per item parity
    static constexpr int DIR = IN + 4 * PSH;                            // [2][N3]  transform's contribution at the own active sites, per item parity
    static constexpr int SW = DIR + 2 * N3;                             // [LB_SIZE] backward weight block
    static constexpr int T3 = SW + LB_SIZE;                             // [LT3_SIZE] conv3^T's weight table (flow_common.h)
    static constexpr int HA1 = T3 + LT3_SIZE;                           // [8][PSH] h1 window
    static constexpr int HA2 = HA1 + 8 * PSH;                           // [8][PSH] h2 window
    static constexpr int WALK = HA2 + 8 * PSH;
    // after the walk, over the planes: the waves' accumulators [8][4 tiles][4][64], bias lane sums [8][2][8], conv3 sums [432]
    static constexpr int RED = 0, RBS = RED + 8 * 4 * 4 * 64, RC3 = RBS + 8 * 2 * 8, RSIZE = RC3 + 432;
    static constexpr int SIZE = WALK > RSIZE ? WALK : RSIZE;
    static_assert(TR == 16 && TC == 16 && NA == 64, "thread maps: conv2^T tile map, K split of the weight-gradient GEMMs, one wave per g_out plane");
    static_assert(W1R % 2 == 0 && NTT <= NT && 2 * N3 <= NT && N1W <= NT && GOC % 2 == 0, "thread maps");
    static_assert(SIZE * 8 <= 160 * 1024, "one workgroup per CU (160 KB of LDS on gfx950)");
};

// conv3 (8 -> 3, active sites only) weight gradient: thread = (output (co, ci, tap), half of the active sites); the 32 sites of
// the half at compile-time offsets from the thread's base (k_flow_wgrad's loop)
template <int MU, int TC, int W1C>
__device__ __forceinline__ void conv3_acc(const double* __restrict__ pg, const double* __restrict__ ph, double (&acc)[4]) {
    // in batches of eight sites: the eight h2 reads and the four 16-byte g_out reads of a batch are issued together, then its
    // eight FMAs (left to itself the scheduler of the mu = 0 instance put a full LDS wait behind every single read: 2.2 k cycles
    // for this loop against 1.0 k in the mu = 1 instance with the same instructions)
#pragma unroll
    for (int a0 = 0; a0 < 32; a0 += 8) {
        double2_t g2[4];
        double hv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) g2[e] = *reinterpret_cast<const double2_t*>(pg + a0 + 2 * e);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int aa = a0 + e;
            hv[e] = ph[MU == 0 ? (aa / (TC / 4)) * W1C + 4 * (aa % (TC / 4)) : 4 * (aa / TC) * W1C + aa % TC];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e & 3] = fma((e & 1) ? g2[e >> 1].y : g2[e >> 1].x, hv[e], acc[e & 3]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int TR, int TC, int MU>
__global__ FT_LDS_B64 __launch_bounds__(NT, 2) void k_flow_bwd_train(FlowLayerArgs A) {
    using S = SmemT<TR, TC>;
    constexpr int W3C = S::W3C, N3W = S::N3W, W2R = S::W2R, W2C = S::W2C, N2W = S::N2W;
    constexpr int W1R = S::W1R, W1C = S::W1C, N3 = S::N3, NA = S::NA, PS1 = S::PS1, PS2 = S::PS2, RS2 = S::RS2;
    constexpr int NH = S::NH, PSH = S::PSH;
    __shared__ __attribute__((aligned(16))) double sm[S::SIZE];
    double* sGO = sm + S::GO;   double* sGZ2 = sm + S::GZ2;  double* sD1 = sm + S::D1;
    double* sW = sm + S::SW;    double* sHA1 = sm + S::HA1;  double* sHA2 = sm + S::HA2;

    const int tid = threadIdx.x;
    __builtin_assume(tid >= 0 && tid < NT);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < NW);
    constexpr int mu = MU;
    const int L = A.L, off = A.off;
    // what the launcher guarantees (flow_bwd_train_shape)
    __builtin_assume(off >= 0 && off < 4 && L >= 32 && L <= 8192 && (L & (L - 1)) == 0);
    const int n = L * L;
    const int nti_ = L / TR, ntj_ = L / TC, ntiles = nti_ * ntj_;
    (void)nti_;

    // ---- the walk (k_flow_wgrad's): the ns workgroups resident together on an XCD stand on ns consecutive items at every step
    //      and move on by ns, so that the halo lines neighbouring tiles share are fetched once into the XCD's L2.
    //      grid: x = 8 XCDs (blockIdx.x % 8) x rounds x ns; round kr = xcd * R + r covers items [kr * tpw * ns, (kr + 1) * tpw * ns)
    const int items = A.B * ntiles, tpw = A.tpw, ns = A.wg_ns;
    const int KR = (items + tpw * ns - 1) / (tpw * ns), R = (KR + 7) >> 3;
    const int idx = (int)blockIdx.x >> 3, r_ = idx / ns, s_ = idx - r_ * ns, kr = ((int)blockIdx.x & 7) * R + r_;
    const int first = kr * tpw * ns + s_;
    if (r_ >= R || first >= items) return;
    const int grp = kr * ns + s_;                                         // valid groups are a prefix of this numbering
    const int nwalk = min(tpw, (items - first + ns - 1) / ns);
    double* gw0 = A.gw_part + (size_t)grp * FLOW_GW_STRIDE;              // the group's partial
    const double* __restrict__ w = A.wint;
    const double cb = A.glogj_const;
    const unsigned Bn = (unsigned)A.B * (unsigned)n;
    auto ldu2 = [](const double* base, unsigned idx_) {                  // 16-byte load, scalar base + 32-bit element offset
        const double2_t* p = reinterpret_cast<const double2_t*>(reinterpret_cast<const char*>(base) + idx_ * 8u);
        return FT_NT_LOAD >= 1 ? __builtin_nontemporal_load(p) : *p;
    };

    // ---- once per walk: the layer's backward weight block, the zeros behind the windows (read by discarded ky = 3 columns only)
    {
        constexpr int NWC = (LB_SIZE + NT - 1) / NT;
#pragma unroll
        for (int k = 0; k < NWC; ++k)
            if (tid + k * NT < LB_SIZE) sW[tid + k * NT] = ldu(w + (mu == 0 ? WBWD1 : WBWD), (unsigned)(tid + k * NT));
#pragma unroll
        for (int k = 0; k < (LT3_SIZE + NT - 1) / NT; ++k)
            if (tid + k * NT < LT3_SIZE) sm[S::T3 + tid + k * NT] = ldu(w + (mu == 0 ? WT3C : WT3R), (unsigned)(tid + k * NT));
        if (tid < 2 * (PSH - NH)) {
            const int pl = tid / (PSH - NH), e = NH + tid % (PSH - NH);
            sm[S::IN + pl * PSH + e] = 0.0; sm[S::IN + (2 + pl) * PSH + e] = 0.0;
#pragma unroll
            for (int ch = pl; ch < 8; ch += 2) { sHA1[ch * PSH + e] = 0.0; sHA2[ch * PSH + e] = 0.0; }
        }
    }

    // ---- item-independent thread maps
    // transform tasks on the last waves: slot `ta` of the tile+3 window's active lines
    const int ta = tid - (NT - S::NTT);
    // conv3^T on the matrix cores (flow_mfma_common.h Conv3T: here the matrix pipe idles through the VALU stages of the lone
    // workgroup, so the 42 MFMAs cost less than the VALU stage's 72 FMAs on 62 LDS instructions per thread did -- in
    // k_flow_bwd_gather, two workgroups per CU on a saturated DP pipe, the same change LOST 1 %: profiles/r06_ab_bwd_conv3t_mfma.txt)
    using C3 = Conv3T<mu, W2R>;
    static_assert(W2R == W2C && C3::NIT == 2 && N2W <= NT, "square window, two rounds of conv3^T tiles");
    // conv2^T tile map (flow_bwd_gather.hip): tiles 0 .. NU-1: u = tile, v = 0 .. 15; tiles NU, NU+1: v = 16, 17 of the even / odd u
    constexpr int NU = W1C / 2, NTILE1 = NU + 2, NIT1 = (NTILE1 + NW - 1) / NW;
    static_assert(NIT1 == 2 && W1R == W1C, "two rounds of conv2^T tiles");
    int pu[NIT1], pv[NIT1];
    bool pok[NIT1];
#pragma unroll
    for (int it = 0; it < NIT1; ++it) {
        const int T = wave + NW * it, i = lane & 15;
        if (T < NU) { pu[it] = T; pv[it] = i; pok[it] = true; }
        else { pu[it] = 2 * (i >> 1) + (T - NU); pv[it] = 16 + (i & 1); pok[it] = T < NTILE1 && pu[it] < NU; if (!pok[it]) { pu[it] = 0; pv[it] = 0; } }
    }
    // h1 / h2 windows: tasks (window site, channel quad) in two rounds; the (cos, sin) window: thread = window site
    constexpr int NITH = 2 * NH, NRH = (NITH + NT - 1) / NT;
    int hwr[NRH], hwc[NRH], hwq[NRH], hls[NRH];
#pragma unroll
    for (int k = 0; k < NRH; ++k) {
        const int t = min(tid + k * NT, NITH - 1), ws = t >> 1;
        hwq[k] = t & 1; hwr[k] = fdiv<W1C>(ws); hwc[k] = ws - hwr[k] * W1C;
        hls[k] = tid + k * NT < NITH ? (4 * hwq[k]) * PSH + hwr[k] * W1C + hwc[k] : -1;
    }
    const int fwr = fdiv<W1C>(min(tid, NH - 1)), fwc = min(tid, NH - 1) - fwr * W1C;
    const bool fwtask = tid < NH;
    const bool fwfrozen = fwtask && ((((mu == 0 ? fwc : fwr) - 1 - off) & 3) == 1 || (((mu == 0 ? fwc : fwr) - 1 - off) & 3) == 2);   // tile origins are multiples of 4
    // own sites (final store)
    const int orr = fdiv<TC>(tid), occ = tid - orr * TC;
    const bool ovalid = tid < N3;
    // weight-gradient GEMM lane maps (k_flow_wgrad: A row m = (co, dy), B column n = (ci, kx, kyb), ky = 2 kyb + dy)
    const int wg_ = lane >> 4, wi_ = lane & 15, wco = wi_ & 7, wdy = wi_ >> 3;
    // conv1 (row pairs, all sixteen columns of a walk row in four K steps):
    const int pa1 = wco * PS1 + (1 - wdy) * W1C + 1 + wg_;                // gz1 of the tile's site (walk row - dy, 4 cs + g) in the tile+1 plane
    auto pbf = [&](int ncol, int ncols) { const int nc = ncol < ncols ? ncol : 0, ci = nc / 6, kx = (nc % 6) >> 1, kyb = nc & 1; return ci * PSH + 2 * kyb * W1C + kx + wg_; };
    const int pb3 = pbf(wi_, 12);
    // conv2: gz2 is an exact 0 on every fourth stripe line (no active site within reach), so its GEMM walks the LIVE lines
    // only -- twelve of a tile's sixteen, three K steps where the dense walk takes four (153 MFMAs per item instead of 204) --
    // and pairs the shifted copies of gz2 ALONG the lines, so that the K lanes run across them: mu = 0 (lines = columns) as
    // conv1 above (M = (co, dy), N = (ci, kx, kyb), the walk goes down the rows), mu = 1 (lines = rows) transposed (M = (co,
    // dx), N = (ci, ky, kxb), kx = 2 kxb + dx, the walk goes along the columns).  K lane g of K step cs = live line 4 cs + g.
    auto live_line_of = [&](int cs, int kl) {                             // live line of K step cs, K lane kl (tile coordinate across the stripe lines)
        const int l = 4 * cs + kl, q = fdiv<3>(l);
        return 4 * q + ((off + 3 + (l - 3 * q)) & 3);                     // stripe classes 3, 0, 1 of quad q
    };
    // A: gz2 at (walk - d along the walk, live line across), tile+2 plane; B: h1 at (walk + 2 kb, line + ka), tile+1 plane
    const int pa2 = wco * PS2 + (mu == 0 ? (2 - wdy) * RS2 + 2 : 2 * RS2 + (2 - wdy));
    auto pb2f = [&](int ncol) { const int ci = ncol / 6, ka = (ncol % 6) >> 1, kb = ncol & 1; return ci * PSH + (mu == 0 ? 2 * kb * W1C + ka : ka * W1C + 2 * kb); };
    const int pb0 = pb2f(wi_), pb1 = pb2f(16 + wi_), pb2 = pb2f(32 + wi_);

    // accumulators of the whole walk
    double4_t acc[4];                                                      // this wave's K slice of the four N tiles (0..2 conv2, 3 conv1)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = double4_t{0.0, 0.0, 0.0, 0.0};
    double bsum[2] = {0.0, 0.0};                                           // bias sums b2, b1 of the slice: lanes dy = 0
    double acc3[3] = {0.0, 0.0, 0.0};                                      // conv3: thread = (output, site half) in [0]; wave 7: b3 lane partials

    // ---- what does not depend on the item: tile origins are multiples of 16, so the stripe phase of every window is the layer's
    const int c0 = (off + 3) & 3, r0 = c0;                                 // first active column / row of the tile+3 window
    int tr3 = 3, tc3 = 3;                                                  // transform task: active site (tr3, tc3) of the tile+3 window
    bool ttask = false;
    if (ta >= 0) {
        if (mu == 0) { tr3 = fdiv<S::NLC>(ta); tc3 = c0 + 4 * (ta - tr3 * S::NLC); ttask = tr3 < S::W3R && tc3 < W3C; }
        else { const int m = fdiv<W3C>(ta); tc3 = ta - m * W3C; tr3 = r0 + 4 * m; ttask = tr3 < S::W3R; }
        if (!ttask) { tr3 = 3; tc3 = 3; }                                    // any valid site
    }
    const unsigned wmagic = (unsigned)(L - 1);

    // ---- an item's operands, in registers.  Every load is unconditional, from a clamped address; each GROUP of them is issued for
    //      item i + 1 right behind the stage of item i that consumed the group (the last item loads itself again: no branch around
    //      the loads), so nothing of an item's load latency is left in front of its stages -- one workgroup per CU has no second
    //      workgroup to hide it behind.  Barriers wait for LDS traffic only (lds_barrier): the loads stay in flight across them.
    double tcv[4 * NMIX], ag[2], fcs, fsn;                                 // group A: transform adjoint, net-input window
    double d2v[C3::NIT][4];                                                // group D2: act'(z2) of the lane's conv3^T pairs (channels 2 g, 2 g + 1 of both members)
    double d1v[NIT1][4];                                                   // group D1: act'(z1) of the conv2^T epilogue
    double gpin;                                                           // group G: upstream gradient of the own site
    double2_t hv1[NRH][2], hv2[NRH][2];                                    // groups H1, H2: the h1 / h2 windows
    struct Item { int b, i0, j0; };
    auto item_at = [&](int k) {                                            // uniform: scalar divisions
        const int item = first + (k < nwalk ? k : nwalk - 1) * ns;
        Item q;
        q.b = item / ntiles;
        const int tl = item - q.b * ntiles, ti = tl / ntj_;
        q.i0 = ti * TR; q.j0 = (tl - ti * ntj_) * TC;
        return q;
    };
#define WI_(q, k) mul24(wrap_line<true, true>((q).i0 + (k), L, wmagic), L)
#define wi_(q, k) wrap_line<true, true>((q).i0 + (k), L, wmagic)
#define WJ_(q, k) wrap_line<true, true>((q).j0 + (k), L, wmagic)
    auto issue_A = [&](const Item& q) {
        const unsigned bn = (unsigned)q.b * (unsigned)n;
        const double* __restrict__ stc = uniform_at(A.stash, 16u * Bn + 2u * bn);
        const double* __restrict__ scs = uniform_at(A.stash, 18u * Bn + bn);
        const int i = wi_(q, tr3 - 3), j = WJ_(q, tc3 - 3);
        const unsigned ia = (unsigned)stash_active_idx(i, j, L, mu);
#pragma unroll
        for (int e = 0; e < 4 * NMIX; e += 2) {                              // [k][n/4][A B C E] (struct Stash): 16 bytes per load
            const double2_t t2 = ldu2(stc + (size_t)(e >> 2) * n, ia * 4u + (e & 3));
            tcv[e] = t2.x; tcv[e + 1] = t2.y;
        }
        const double* gsrc = uniform_at(A.up_gp, bn);                        // upstream gradient: the plaquette-gradient field
        const int iL = mul24(i, L);
        ag[0] = ldu(gsrc, (unsigned)(iL + j));
        ag[1] = ldu(gsrc, (unsigned)(mu == 0 ? iL + WJ_(q, tc3 - 4) : WI_(q, tr3 - 4) + j));
        // (cos, sin) of the frozen plaquettes on the tile+1 window (the net input: conv1's weight gradient reads the window,
        // conv1^T's adjoint the own sites), one window site per thread
        const unsigned ic = fwfrozen ? (unsigned)stash_frozen_idx(wi_(q, fwr - 1), WJ_(q, fwc - 1), L, mu, off) : 0u;
        fcs = ldu_j(scs, ic); fsn = ldu_j(scs + (n >> 1), ic);
    };
    auto issue_G = [&](const Item& q) {
        gpin = ldu_j(uniform_at(A.up_gp, (unsigned)q.b * (unsigned)n), ovalid ? (unsigned)(mul24(q.i0 + orr, L) + q.j0 + occ) : 0u);
    };
    auto issue_H = [&](const Item& q, int plane, double2_t (&hv)[NRH][2]) {  // plane 19: h1, 27: h2 (struct Stash)
        const double* __restrict__ sh = uniform_at(A.stash, (unsigned)plane * Bn + 8u * (unsigned)q.b * (unsigned)n);
#pragma unroll
        for (int k = 0; k < NRH; ++k) {
            const unsigned hat = (unsigned)(WI_(q, hwr[k] - 1) + WJ_(q, hwc[k] - 1)) * 8u + 4u * (unsigned)hwq[k];
            hv[k][0] = ldu2(sh, hat); hv[k][1] = ldu2(sh, hat + 2);
        }
    };
    auto issue_D2 = [&](const Item& q) {
        const double* pl = uniform_at(A.stash, 8u * (Bn + (unsigned)q.b * (unsigned)n));
        auto ldu2o = [](const double* base, unsigned o) {
            const double2_t* p = reinterpret_cast<const double2_t*>(reinterpret_cast<const char*>(base) + o);
            return FT_NT_LOAD >= 1 ? __builtin_nontemporal_load(p) : *p;
        };
#pragma unroll
        for (int e = 0; e < C3::NIT; ++e) {
            const C3 P(wave + NW * e, lane, c0);
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) {
                // on a dead line (no active site within reach: an exact 0 there) the stash holds nothing: record 0 instead
                const bool live = P.ok && !P.dead(dd);
                const int i = wi_(q, P.row(dd) - 2), j = WJ_(q, P.col(dd) - 2);
#if FT_D2_C
                const int go = mu == 0 ? mul24(i, 3 * (L >> 2)) + stash_live_line<true>(j, L, off) : mul24(stash_live_line<true>(i, L, off), L) + j;
#else
                const int go = mul24(i, L) + j;
#endif
                const double2_t vd = ldu2o(pl, ft_off32((unsigned)(live ? go : 0) * 8u) + 16u * (unsigned)(lane >> 4));
                d2v[e][2 * dd] = vd.x; d2v[e][2 * dd + 1] = vd.y;
            }
        }
    };
    auto issue_D1 = [&](const Item& q) {
        const double* __restrict__ st1 = uniform_at(A.stash, 8u * (unsigned)q.b * (unsigned)n);
#pragma unroll
        for (int e = 0; e < NIT1; ++e) {
            const int ra = mu == 0 ? pv[e] : 2 * pu[e], ca = mu == 0 ? 2 * pu[e] : pv[e];
            // mu = 0: the act'(z1) plane is stored transposed (FT_D1_T, flow_mfma_common.h): site index j L + i
            const int ga = (FT_D1_T && mu == 0) ? mul24(WJ_(q, ca - 1), L) + wi_(q, ra - 1) : WI_(q, ra - 1) + WJ_(q, ca - 1);
            const int gb = mu == 0 ? (FT_D1_T ? mul24(WJ_(q, ca), L) + wi_(q, ra - 1) : WI_(q, ra - 1) + WJ_(q, ca)) : WI_(q, ra) + WJ_(q, ca - 1);
            const unsigned og = 2u * (unsigned)(lane >> 4);                  // channels 2 g, 2 g + 1: one 16-byte load per site
            const double2_t va = ldu2(st1, (unsigned)ga * 8u + og), vb = ldu2(st1, (unsigned)gb * 8u + og);
            d1v[e][0] = va.x; d1v[e][1] = va.y; d1v[e][2] = vb.x; d1v[e][3] = vb.y;
        }
    };
    {
        const Item q0 = item_at(0);                                          // in the order the stages consume them (loads return in order)
        issue_A(q0); issue_D2(q0); issue_H(q0, 19, hv1); issue_D1(q0); issue_H(q0, 27, hv2); issue_G(q0);
    }

    // conv3's weight gradient of an item (VALU; g_out at the own active sites and the h2 window): runs in the FIRST stage of
    // the next item, beside the transform adjoint that keeps three of the eight waves busy there (its planes outlive the item:
    // g_out compact per item parity, the h2 window until the next item's third stage)
    auto conv3_wgrad = [&](const double* sGOCi) {
        if (tid < 432) {                                                 // gw2[co][ci][tap] += sum over the own active sites of g_out[co] h2[ci][site + tap]
            const int hf = tid >= 216 ? 1 : 0, t = tid - 216 * hf;
            const int co = fdiv<9>(fdiv<8>(t)), ci = fdiv<9>(t) & 7, tap = t - fdiv<9>(t) * 9, ky = fdiv<3>(tap), kx = tap - 3 * ky;
            const double* pg = sGOCi + co * NA + hf * (NA / 2);
            const double* ph = sHA2 + ci * PSH + ky * W1C + kx;          // h2 at own (r, c) + (ky - 1, kx - 1): window index (r + ky) W1C + c + kx
            double c3[4] = {0.0, 0.0, 0.0, 0.0};
            if (mu == 0) conv3_acc<0, TC, W1C>(pg, ph + hf * (NA / 2 / (TC / 4)) * W1C + off, c3);
            else         conv3_acc<1, TC, W1C>(pg, ph + (off + 4 * hf * (NA / 2 / TC)) * W1C, c3);
            acc3[0] += (c3[0] + c3[1]) + (c3[2] + c3[3]);
        } else if (tid >= 448) {                                         // b3: the eighth wave sums the three g_out planes
#pragma unroll
            for (int k = 0; k < 3; ++k) acc3[k] += sGOCi[k * NA + lane];
        }
    };
#ifdef FT_BT_STAMPS       // measurement builds only: cycles per stage, summed over the walk, printed by two workgroups
    long long stc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stl_ = (long long)__builtin_readcyclecounter();
#define BT_STAMP(k) do { const long long t_ = (long long)__builtin_readcyclecounter(); stc_[k] += t_ - stl_; stl_ = t_; } while (0)
#else
#define BT_STAMP(k) do { } while (0)
#endif
#pragma unroll 1
    for (int it = 0; it < nwalk; ++it) {
        const Item cur = item_at(it), nxt = item_at(it + 1);
        BT_STAMP(7);
        const int i0 = cur.i0, j0 = cur.j0;
        __builtin_assume(i0 >= 0 && i0 < L && j0 >= 0 && j0 < L && cur.b >= 0 && cur.b < (1 << 20));
        double* sDir = sm + S::DIR + (it & 1) * N3;
        double* sGOC = sm + S::GOC + (it & 1) * 3 * NA;
        double* sIn = sm + S::IN + (it & 1) * 2 * PSH;
        const unsigned bn = (unsigned)cur.b * (unsigned)n;

        // ---- stage 1: transform adjoint -> g_out; the net-input window into LDS -----------------------------------
        if (ttask) {
            // adjoint of the tan-mixture transform (layers.py:66-90) from the forward's coefficients
            const double gdelta = ag[0] - ag[1];
            const int at = tr3 * W3C + tc3;
            double csum = 0.0, esum = 0.0;
#pragma unroll
            for (int k = 0; k < NMIX; ++k) { csum += tcv[4 * k + 2]; esum += tcv[4 * k + 3]; }
            const double tsum = NMIX * csum;                                 // sum_k 1 / D_k
            double rs = __builtin_amdgcn_rcp(tsum);
            rs = fma(fma(-tsum, rs, 1.0), rs, rs);
            rs = fma(fma(-tsum, rs, 1.0), rs, rs);
            const double cbr = cb * rs;
            static_assert(NMIX == 2, "g_out record: dL/ds_0, dL/ds_1, dL/dt");
            const double gs0 = gdelta * tcv[0] + cbr * tcv[1], gs1 = gdelta * tcv[4] + cbr * tcv[5];
            sGO[at] = gs0; sGO[N3W + at] = gs1;                              // dL/ds_k
            sGO[2 * N3W + at] = gdelta;                                      // dL/dt
            const int r = tr3 - 3, c = tc3 - 3;
            if ((unsigned)r < (unsigned)TR && (unsigned)c < (unsigned)TC) {
                sDir[r * TC + c] = gdelta * (csum - 1.0) - cbr * esum;
                // the own active sites once more, compact, in k_flow_wgrad's task order: conv3's weight gradient reads them in pairs
                const int a = mu == 0 ? r * (TC / 4) + ((c - off) >> 2) : ((r - off) >> 2) * TC + c;
                sGOC[a] = gs0; sGOC[NA + a] = gs1; sGOC[2 * NA + a] = gdelta;
            }
        }
        if (fwtask) { sIn[tid] = fwfrozen ? fcs : 1.0; sIn[PSH + tid] = fwfrozen ? fsn : 0.0; }
        __builtin_amdgcn_sched_barrier(0);
        issue_A(nxt);
        BT_STAMP(8);
        if (it > 0) conv3_wgrad(sm + S::GOC + ((it - 1) & 1) * 3 * NA);      // of the item before
        BT_STAMP(0);
        lds_barrier();
        BT_STAMP(1);

        // ---- stage 2: conv3^T on the matrix cores -> gz2; the h1 window into LDS -------------------------------------
        {
            const int g = lane >> 4;
#pragma unroll
            for (int e = 0; e < C3::NIT; ++e) {
                const int T = wave + NW * e;
                if (T >= C3::NTILE) break;
                const C3 P(T, lane, c0);
                const double4_t z = conv3t_tile<mu, W2R, W3C, N3W>(sGO, sm + S::T3, P, lane);
                if (P.ok) {
                    // z[q]: channel 2 g + (q & 1) at member q >> 1; dead lines get an exact 0, whatever the stash holds there
                    const int s0 = P.row(0) * RS2 + P.col(0), ds = mu == 0 ? 1 : RS2;
                    double* pz = sGZ2 + 2 * g * PS2 + s0;
                    pz[0] = P.dead(0) ? 0.0 : z[0] * d2v[e][0]; pz[PS2] = P.dead(0) ? 0.0 : z[1] * d2v[e][1];
                    pz[ds] = P.dead(1) ? 0.0 : z[2] * d2v[e][2]; pz[PS2 + ds] = P.dead(1) ? 0.0 : z[3] * d2v[e][3];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NRH; ++k)
            if (hls[k] >= 0) { double* p1 = sHA1 + hls[k]; p1[0] = hv1[k][0].x; p1[PSH] = hv1[k][0].y; p1[2 * PSH] = hv1[k][1].x; p1[3 * PSH] = hv1[k][1].y; }
        __builtin_amdgcn_sched_barrier(0);
        issue_D2(nxt); issue_H(nxt, 19, hv1);
        lds_barrier();
        BT_STAMP(2);

        // ---- stage 3: conv2^T (MFMA) times act'(z1) -> gz1; conv2's weight gradient from gz2 and the h1 window; the h2 window into LDS
        {
            const int g = lane >> 4, i = lane & 15;
            const double* wp = sW + LB_T2 + KConv2Row::wlane(g, i & 7, i >> 3);        // the lane part is the same for both K orders
            const int kd0 = ((mu == 0 ? c0 : r0) + 1) & 3;                             // dead window line of the even pairs (odd: + 2)
#pragma unroll
            for (int q = 0; q < NIT1; ++q) {
                const int T = wave + NW * q;
                if (T >= NTILE1) break;
                const int kd = (kd0 + 2 * (T < NU ? T & 1 : T - NU)) & 3;              // wave-uniform
                const double* a0 = sGZ2 + g * PS2 + (mu == 0 ? pv[q] * RS2 + 2 * pu[q] : 2 * pu[q] * RS2 + pv[q]);
                double4_t ac;
                if (mu == 0) {
                    switch (kd) {
                        case 0: ac = conv2t_tile<KConv2Col, 4, 0, RS2, PS2>(wp, a0); break;
                        case 1: ac = conv2t_tile<KConv2Col, 4, 1, RS2, PS2>(wp, a0); break;
                        case 2: ac = conv2t_tile<KConv2Col, 4, 2, RS2, PS2>(wp, a0); break;
                        default: ac = conv2t_tile<KConv2Col, 4, 3, RS2, PS2>(wp, a0); break;
                    }
                } else {
                    switch (kd) {
                        case 0: ac = conv2t_tile<KConv2Row, 3, 0, RS2, PS2>(wp, a0); break;
                        case 1: ac = conv2t_tile<KConv2Row, 3, 1, RS2, PS2>(wp, a0); break;
                        case 2: ac = conv2t_tile<KConv2Row, 3, 2, RS2, PS2>(wp, a0); break;
                        default: ac = conv2t_tile<KConv2Row, 3, 3, RS2, PS2>(wp, a0); break;
                    }
                }
                if (pok[q]) {
                    const int ra = mu == 0 ? pv[q] : 2 * pu[q], ca = mu == 0 ? 2 * pu[q] : pv[q];   // site 0; site 1 = next column / row
                    const int ds = mu == 0 ? 1 : W1C;
                    double* pd = sD1 + 2 * g * PS1 + ra * W1C + ca;          // MFMA rows g, g + 4 = channels 2 g, 2 g + 1 (ft_chan)
                    pd[0] = ac[0] * d1v[q][0]; pd[PS1] = ac[1] * d1v[q][1]; pd[ds] = ac[2] * d1v[q][2]; pd[PS1 + ds] = ac[3] * d1v[q][3];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        issue_D1(nxt);
        {
            // K walk of the weight-gradient GEMMs: wave = window rows 2 wave, 2 wave + 1 (waves 0..3 also the step (row TR, cs = wave)).
            // Walk row rho pairs gz rows rho - dy with hin rows rho + 2 kyb; the tile's rows -1 (rho = 0, dy = 1) and TR (rho = TR,
            // dy = 0) are other tiles' sites: 0 instead.
            constexpr int SA = mu == 0 ? RS2 : 1, SB = mu == 0 ? W1C : 1;      // plane steps of one walk position
            auto step = [&](int oa, int ob, int zero_d) {                   // zero_d: the shift d whose source line lies outside the tile (-1: none)
                double a2 = sGZ2[pa2 + oa];
                if (zero_d >= 0) a2 = (wdy == zero_d) ? 0.0 : a2;
                const double b0 = sHA1[pb0 + ob], b1 = sHA1[pb1 + ob], b2 = sHA1[pb2 + ob];
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc[2], 0, 0, 0);
                bsum[0] += a2;
            };
            // wave = walk positions 2 wave, 2 wave + 1; position TR's three K steps on waves 3 and 7 (their SIMD has the fewest
            // conv2^T tiles)
            constexpr int LA = mu == 0 ? 1 : RS2, LB = mu == 0 ? 1 : W1C;      // plane steps of one line across
            int wgv = wg_;
            asm volatile("" : "+v"(wgv));                                    // the lines' addresses are worked out HERE, per item: hoisted out of the walk they cost twelve registers for all of it
#pragma unroll
            for (int cs = 0; cs < 3; ++cs) {
                const int line = live_line_of(cs, wgv), oa = line * LA, ob = line * LB;
#pragma unroll
                for (int r = 0; r < 2; ++r) step((2 * wave + r) * SA + oa, (2 * wave + r) * SB + ob, (r == 0 && wave == 0) ? 1 : -1);
                if ((wave == 3 && cs < 2) || (wave == 7 && cs == 2)) step(TR * SA + oa, TR * SB + ob, 0);
                __builtin_amdgcn_sched_barrier(0);                           // bounds how many operand reads are hoisted ahead (registers)
            }
        }
#pragma unroll
        for (int k = 0; k < NRH; ++k)
            if (hls[k] >= 0) { double* p2 = sHA2 + hls[k]; p2[0] = hv2[k][0].x; p2[PSH] = hv2[k][0].y; p2[2 * PSH] = hv2[k][1].x; p2[3 * PSH] = hv2[k][1].y; }
        __builtin_amdgcn_sched_barrier(0);
        issue_H(nxt, 27, hv2);
        // conv1^T's 18 weights of this wave's hidden channel: scalar loads from the weight block (constant address space)
        typedef const double __attribute__((address_space(4))) * cdptr;
        double w0s[18];
        {
            cdptr wq = (cdptr)(size_t)(w + (mu == 0 ? WBWD1 : WBWD) + LB_W0 + wave * 18);
            typedef double double8c_t __attribute__((ext_vector_type(8)));
            typedef const double8c_t __attribute__((address_space(4))) * cd8ptr;
            const double8c_t va = *(cd8ptr)(wq), vb = *(cd8ptr)(wq + 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) { w0s[k] = va[k]; w0s[8 + k] = vb[k]; }
            w0s[16] = wq[16]; w0s[17] = wq[17];
        }
        lds_barrier();
        BT_STAMP(3);

        // ---- stage 4: conv1^T at the tile's own frozen plaquettes (wave = hidden channel, two sites per lane; channel partials
        //      over the gz2 planes, free by now); conv1's weight gradient (MFMA) from gz1 and the net-input window; conv3's
        //      (VALU) from g_out and the h2 window ----------------------------------------------------------------------
        static_assert(NW == 8 && N3 / 2 == 2 * 64 && 8 * 2 * (N3 / 2) <= 8 * PS2, "one wave per hidden channel, two sites per lane");
        double* sPart = sGZ2;                                                // [8 co][2: cos, sin][N3 / 2]
        auto frozen_site = [&](int f, int& r, int& c) {
            const int h = f >> 4, q = f & 15;
            if (mu == 0) { r = q; c = 4 * (h >> 1) + ((off + 1 + (h & 1)) & 3); }
            else { c = q; r = 4 * (h >> 1) + ((off + 1 + (h & 1)) & 3); }
        };
        {
            const int co = wave;
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                const int f = lane + 64 * sx;
                int r, c;
                frozen_site(f, r, c);
                const double* gz = sD1 + co * PS1 + r * W1C + c;            // window coordinates (r + 2 - ky, c + 2 - kx)
                double gv[9], gc = 0.0, gs = 0.0;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) gv[tp] = gz[(2 - tp / 3) * W1C + 2 - tp % 3];
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) { gc = fma(gv[tp], w0s[tp], gc); gs = fma(gv[tp], w0s[9 + tp], gs); }
                sPart[(co * 2 + 0) * (N3 / 2) + f] = gc;
                sPart[(co * 2 + 1) * (N3 / 2) + f] = gs;
            }
        }
        {
            const int oa0 = 2 * wave * W1C, ob0 = 2 * wave * W1C;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int cs = 0; cs < TC / 4; ++cs) {
                    double a1 = sD1[pa1 + oa0 + r * W1C + 4 * cs];
                    if (r == 0) a1 = (wave == 0 && wdy) ? 0.0 : a1;
                    const double b3 = sIn[pb3 + ob0 + r * W1C + 4 * cs];
                    acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b3, acc[3], 0, 0, 0);
                    bsum[1] += a1;
                }
            if (wave < 4) {
                double a1 = sD1[pa1 + TR * W1C + 4 * wave];
                a1 = wdy ? a1 : 0.0;
                const double b3 = sIn[pb3 + TR * W1C + 4 * wave];
                acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b3, acc[3], 0, 0, 0);
                bsum[1] += a1;
            }
        }
        lds_barrier();
        BT_STAMP(4);

        // ---- gP_out = gP_in + this layer's contribution at the own sites: the transform's at the active ones (sDir), conv1^T's
        //      through (cos, sin) at the frozen ones -- the sum over the eight channel partials by the site's own thread: no
        //      barrier between the partials and the store (what it reads is not written before the next item's barriers: sDir,
        //      the net-input window and compact g_out alternate between two buffers) ------------------------------------------
        if (ovalid) {
            const int cls = ((mu == 0 ? occ : orr) - off) & 3;               // 0 active, 1|2 frozen, 3 passive (tile origins: multiples of 4)
            double contrib = 0.0;
            if (cls == 0) contrib = sDir[tid];
            else if (cls != 3) {
                // the site's index in frozen_site's enumeration: f = position along the lines + 16 (2 (line quad) + class - 1)
                const int f = (mu == 0 ? orr + 16 * (2 * (occ >> 2) + cls - 1) : occ + 16 * (2 * (orr >> 2) + cls - 1));
                double gct = 0.0, gst = 0.0;
#pragma unroll
                for (int co = 0; co < 8; ++co) { gct += sPart[(co * 2 + 0) * (N3 / 2) + f]; gst += sPart[(co * 2 + 1) * (N3 / 2) + f]; }
                const int at = (orr + 1) * W1C + occ + 1;
                contrib = -sIn[PSH + at] * gct + sIn[at] * gst;
            }
            uniform_at(A.gp_out, bn)[mul24(i0 + orr, L) + j0 + occ] = gpin + contrib;
        }
        __builtin_amdgcn_sched_barrier(0);
        issue_G(nxt);
        BT_STAMP(5);
    }
    conv3_wgrad(sm + S::GOC + ((nwalk - 1) & 1) * 3 * NA);               // of the last item
#ifdef FT_BT_STAMPS
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 101))
        printf("bwd_train wg %d mu %d: %d items; cycles per item: stage1 %lld [of it: fill + issue_A %lld] (+barrier %lld) stage2 %lld stage3 %lld stage4 %lld tail %lld top %lld\n", (int)blockIdx.x, mu, nwalk,
               (stc_[0] + stc_[8]) / nwalk, stc_[8] / nwalk, stc_[1] / nwalk, stc_[2] / nwalk, stc_[3] / nwalk, stc_[4] / nwalk, stc_[5] / nwalk, stc_[7] / nwalk);
#endif
#undef BT_STAMP
#undef WI_
#undef wi_
#undef WJ_

    // ---- the group's partial: the waves' K slices summed through LDS in a fixed order (k_flow_wgrad's epilogue)
    lds_barrier();
    {
        double* Rr = sm + S::RED; double* BS = sm + S::RBS; double* C3 = sm + S::RC3;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) Rr[((wave * 4 + nt) * 4 + q) * 64 + lane] = acc[nt][q];
#pragma unroll
        for (int k = 0; k < 2; ++k) {                                    // A rows (co, dy = 0): lanes co + 16 g
            double v = bsum[k];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            if (lane < 8) BS[(wave * 2 + k) * 8 + lane] = v;
        }
        if (tid < 432) C3[tid] = acc3[0];
        else if (tid >= 448) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double v = ft_wave_sum(acc3[k]);
                if (lane == 0) gw0[CB2 + k] = v;
            }
        }
        lds_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                    // element e = (nt, q, lane) = D_nt[row g + 4 q][col i]
            const int e = tid + NT * h, nt = e >> 8, q = (e >> 6) & 3, ln = e & 63;
            double v = 0.0;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) v += Rr[wv * 1024 + e];
            const int g = ln >> 4, i = ln & 15, m = g + 4 * q, co = m & 7, dy = m >> 3;
            const int ncol = (nt < 3 ? nt * 16 : 0) + i;
            if (ncol < (nt < 3 ? 48 : 12)) {
                // conv1 (nt = 3) and conv2 of a mu = 0 layer: columns (ci, kx, kyb), ky = 2 kyb + d; conv2 of a mu = 1 layer: (ci, ky, kxb), kx = 2 kxb + d
                const bool tr = nt < 3 && mu == 1;
                const int ci = ncol / 6, ka = (ncol % 6) >> 1, kb2 = 2 * (ncol & 1) + dy;
                const int ky = tr ? ka : kb2, kx = tr ? kb2 : ka;
                if (kb2 <= 2) gw0[(nt < 3 ? CW1 + (co * 8 + ci) * 9 : CW0 + (co * 2 + ci) * 9) + ky * 3 + kx] = v;
            }
        }
        if (tid < 16) {
            double v = 0.0;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) v += BS[wv * 16 + tid];
            gw0[(tid < 8 ? CB1 : CB0) + (tid & 7)] = v;
        }
        if (tid < 216) gw0[CW2 + tid] = C3[tid] + C3[216 + tid];
    }
}

}  // namespace
#endif

namespace fthmc {

bool flow_bwd_train_built() { return !FT_RECOMP_D1; }

int launch_flow_bwd_train(const FlowLayerArgs& a, hipStream_t s) {
#if FT_RECOMP_D1
    (void)a; (void)s;
    return FTHMC_ERR_UNSUPPORTED;
#else
    if (!flow_shape_ok(a.B, a.L, a.off)) return FTHMC_ERR_ARG;
    if (!flow_bwd_train_shape(a.L) || !a.up_gp || a.up_link || a.glogj || !a.stash || !a.gp_out || !a.gw_part) return FTHMC_ERR_UNSUPPORTED;
    if (!flow_stash_fits32(a.B, a.L, true)) return FTHMC_ERR_UNSUPPORTED;                   // 32-bit plane offsets (uniform_at)
    FlowLayerArgs b = a;
    b.tpw = flow_bwd_train_tpw(a.B, a.L);
    b.wg_ns = flow_bwd_train_ns(a.B, a.L, b.tpw);
    const int items = a.B * FlowGeom{MG_TR, MG_TC}.ntiles(a.L);
    const int KR = (items + b.tpw * b.wg_ns - 1) / (b.tpw * b.wg_ns), R = (KR + 7) / 8;
    const dim3 grid(8 * R * b.wg_ns, 1, 1);
    if (a.mu == 0) hipLaunchKernelGGL((k_flow_bwd_train<MG_TR, MG_TC, 0>), grid, dim3(NT), 0, s, b);
    else hipLaunchKernelGGL((k_flow_bwd_train<MG_TR, MG_TC, 1>), grid, dim3(NT), 0, s, b);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
#endif
}

}  // namespace fthmc
