// Coupling-layer forward on the matrix cores (GaugeEquivCouplingLayer.forward, fthmc/utils/layers.py:196-202,
// with NCPPlaqCouplingLayer.forward :348-371, the conv net :138-167 and the tan-mixture transform :58-90).
//
// A workgroup (512 threads) owns a 16 x 16 tile of one chain, three workgroups share a CU; everything
// between the link loads and the link update lives in LDS:
//   plaquettes + net input (cos P, sin P) on tile+3  ->  conv1 (2 -> 8) + act on tile+2  ->  conv2 (8 -> 8) + act
//   on the live lines of tile+1  ->  conv3 (8 -> 3) at the tile's active sites  ->  tan-mixture transform,
//   log J  ->  link update x' = wrap(x +- (P' - P)) at the active links.
// conv1 and conv2 are implicit GEMMs on v_mfma_f64_16x16x4_f64 (conv2: flow_mfma_common.h mfma_stage; conv1: the frozen
// taps only, below), both operands one ds_read_b64 per MFMA out of LDS; conv3 (N = 3) stays on the fp64 VALU with its
// wave-uniform weights in SGPRs.
// With a.stash the kernel also writes what the backward needs (struct Stash: act'(z1), act'(z2), the
// transform's adjoint coefficients, cos/sin of the frozen plaquettes, and h1, h2 for the weight gradients),
// so that flow_bwd_gather.hip never re-runs the network.
// Measured on MI355X (tools/microbench): fp64 MFMA and fp64 VALU share one DP pipe, so the matrix path
// buys issue efficiency and register-free operands, not extra flops.
#include "flow_mfma_common.h"

namespace {

using namespace fthmc;
using namespace fthmc_flow;

constexpr unsigned FWD_HAS_POUT = 1u << 16, FWD_HAS_DBG = 1u << 17, FWD_HAS_PIN = 1u << 18;     // flags in the forward kernel's hoa word

// LDS plan: three workgroups share a CU (3 x 51.7 KB), so planes whose lifetimes do not overlap share memory:
//   region A: h1 (conv1 -> conv2's MFMA reads), then h2 (conv2's epilogue, after a barrier -> conv3)
//   region B: the net input (stage 0 -> conv1), then the conv3 partials / delta and the transform scratch
template <int TR, int TC> struct SmemF {
    using G = Geom<TR, TC>;
    static constexpr int cmax2(int a, int b) { return a > b ? a : b; }
    static constexpr int H1 = 0;                              // [8][PS1]
    static constexpr int H2 = 0;                              // [8][PS2]   (over h1)
    static constexpr int IN = cmax2(8 * G::PS1, 8 * G::PS2);  // [2][PS0] cos, sin
    static constexpr int ST = IN;                             // [8][3][NAS] conv3 partials, then delta [N3]   (over the net input)
    static constexpr int T2 = ST + 8 * 3 * G::NAS;            // [NMIX][2][NAS] y_k, 1/D_k
    static constexpr int P1 = IN + 2 * G::PS0;                // [LF_P1_SIZE] conv1 weight table (until conv1 is done)
    static constexpr int PG = IN + cmax2(2 * G::PS0 + LF_P1_SIZE, 8 * 3 * G::NAS + NMIX * 2 * G::NAS);   // [N0] plaquettes
    static constexpr int SW = PG + G::N0;                     // [LF_SIZE] resident weight block
    static constexpr int SIZE = SW + LF_SIZE;
    static_assert(G::N3 <= 8 * 3 * G::NAS, "delta must fit over the conv3 partials");
    static_assert(G::N0 % 2 == 0 && 3 * ((SIZE * 8 + 1279) / 1280 * 1280) <= 160 * 1024, "three workgroups per CU (LDS is granted in 1280-byte units)");
};

// REV: the inverse layer (GaugeEquivCouplingLayer.reverse, layers.py:204-210, 373-396): same net on the same
// frozen plaquettes, then the scalar map is inverted per active site instead of applied.
// MU: the layer's stripe direction (A.mu) as a compile-time constant: every `mu == 0 ? a : b` below folds, the LDS steps
// across / along the lines become immediate offsets of the operand reads, and each kernel carries one of the two conv2
// code paths instead of both (selects on per-lane values by a uniform mu were ~5 % of the VALU instructions).
// EXACT: the tiles divide the lattice and L is a power of two (L = 64, 128, 256: BASELINE configs 3, 4, 5), so every tile site
// is a lattice site -- the lattice-edge halves of the bounds tests of the stash stores, the link update and the active sites
// fold away -- and a window line wraps by one v_and.
// Hot arguments: the explicit scalars ahead of the argument block arrive in SGPRs WITH the wave (kernarg preload: csrc/Makefile
// -amdgpu-kernarg-preload-count=16; 15 dwords here), so the first loads of a workgroup do not wait for a scalar load of a cold
// argument segment; the block itself (A0) serves what is needed later or rarely.  hoa = off | act << 8 | flags << 16
// (13 dwords: the preload takes 14).
// SILU: the activation (A.act: silu / relu / leaky_relu, layers.py:124-135) is the reference's default and known at compile time:
// the other two activations' code and the tests on `act` around every sigmoid-of-four leave the kernel (trajectory -0.9 %).
// SWEEP = 1: the launch is a layer of a FORCE sweep (the forward's hot case: 160 of the 176 launches of a trajectory) -- link
// field in and out, activation stash without h1 / h2, no log J, no plaquette-level map -- so every test on those (uniform)
// conditions and the code behind the other outcome leave the kernel, and so do the cycle stamps of the diagnostic launches
// (tools/lifetime.py, fthmc_profile_stages: those run SWEEP = 2, the same specialization WITH the stamps).  SWEEP = 3: a layer
// of an ACTION sweep (the H1 sweep of a trajectory, ft_action: link field in and out, log J, no stash).  SWEEP = 4: a layer of a
// TRAINING sweep (fthmc_train_grad: stash with h1 / h2, log J).  5, 6: SWEEP = 1, 4 with NON-TEMPORAL stash stores, for a stash
// nobody finds in a cache again (launch_fwd: a layer's stash beyond FT_NT_MIN_BYTES).  0: whatever the argument block says.
template <int TR, int TC, bool FASTW, bool REV, int MU, bool EXACT, bool SILU, int SWEEP>
__global__ FT_LDS_B64 __launch_bounds__(NT, REV ? 4 : 6) void k_flow_fwd(const double* hx, const double* hw, double* hy, double* hstash, double* hlogj,
                                                                          int hB, int hL, unsigned hoa, FlowLayerArgs A0) {
    FlowLayerArgs A = A0;
    A.x = hx; A.wint = hw; A.y = hy; A.stash = hstash; A.logj_part = hlogj; A.B = hB; A.L = hL;
    A.off = (int)(hoa & 0xffu); A.act = (int)((hoa >> 8) & 0xffu);
    constexpr bool FS = SWEEP == 1 || SWEEP == 2 || SWEEP == 5, ES = SWEEP == 3, TS = SWEEP == 4 || SWEEP == 6, SW = FS || ES || TS;
    constexpr bool NTS = SWEEP == 5 || SWEEP == 6;                    // stash stores with the non-temporal hint (launch_fwd: big stashes)
    const bool has_pout = !SW && (hoa & FWD_HAS_POUT) != 0, has_pin = !SW && (hoa & FWD_HAS_PIN) != 0;
    const bool has_dbg = (SWEEP == 0 || SWEEP == 2) && (hoa & FWD_HAS_DBG) != 0;
    const bool has_stash = FS || TS || (!ES && A.stash != nullptr), has_y = SW || A.y != nullptr;
    const bool want_logj = ES || TS || (!FS && A.logj_part != nullptr);
    const bool stash_h = TS || (!SW && has_stash && A0.stash_h != 0);
    using S = SmemF<TR, TC>;
    using G = Geom<TR, TC>;
    constexpr int R0C = G::R0C, R1R = G::R1R, R1C = G::R1C, R2R = G::R2R, R2C = G::R2C;
    constexpr int N3 = G::N3, NA = G::NA, NAS = G::NAS, TQ = 2, RS1 = G::RS1;
    constexpr int PS0 = G::PS0, PS1 = G::PS1, PS2 = G::PS2;
    __shared__ __attribute__((aligned(16))) double sm[S::SIZE];
    double* sIn = sm + S::IN;  double* sP = sm + S::PG;   double* sH1 = sm + S::H1;  double* sH2 = sm + S::H2;
    double* sST = sm + S::ST;  double* sDL = sm + S::ST;  double* sT2 = sm + S::T2;  double* sW = sm + S::SW;
    double* sP1 = sm + S::P1;

    const int tid = threadIdx.x;
    __builtin_assume(tid >= 0 && tid < NT);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < NW);
    constexpr int mu = MU;
    const int L = A.L, off = A.off, act = SILU ? (int)FTHMC_ACT_SILU : A.act;
    // what the launchers guarantee, said to the compiler (ranges decide between 24-bit and 64-bit index arithmetic)
    __builtin_assume(off >= 0 && off < 4 && L >= 4 && L <= 8192 && (L & 3) == 0);
    if (EXACT) __builtin_assume(L >= 16 && (L & (L - 1)) == 0);
    const int n = L * L;
    const int nti_ = (A.L + TR - 1) / TR, ntj_ = (A.L + TC - 1) / TC;
    BlockTile bt;
    if (!block_tile(A.B, nti_, ntj_, bt)) return;               // padding blocks when B % 8 != 0 (whole block exits)
    const int b = bt.b, tile = bt.tile, ntiles = nti_ * ntj_;
    const int i0 = bt.ti * TR, j0 = bt.tj * TC;
    __builtin_assume(i0 >= 0 && i0 < L && j0 >= 0 && j0 < L && b >= 0 && b < (1 << 20) && A.B > 0 && A.B <= (1 << 20));
    const unsigned bn = (unsigned)b * (unsigned)n;                    // 32-bit plane offsets: uniform_at()
    const double* __restrict__ x0 = uniform_at(A.x, 2u * bn);
    const double* __restrict__ x1 = x0 + n;
    // plaquette-level map (NCPPlaqCouplingLayer.forward / .reverse, layers.py:348-396): the plaquette field
    // is the input (A.pin) and the output (A.pout) instead of being derived from / folded back into links
    const double* __restrict__ pin = has_pin ? uniform_at(A0.pin, bn) : nullptr;
    const double* __restrict__ w = A.wint;
    // the stamp record is addressed inside the flag's branch: the pointer is not waited for in a production launch
#define DBG_REC (A0.dbg + ((size_t)b * ntiles + tile) * 16)
#define STAMP(k) do { if (has_dbg && tid == 0) DBG_REC[k] = (long long)__builtin_readcyclecounter(); } while (0)
    STAMP(0);
    // slots 14, 15: the constant 100 MHz counter at both ends of the workgroup -- lifetime in cycles / lifetime in ticks = the
    // shader clock the kernel actually ran at (tools/lifetime.py)
    if (has_dbg && tid == 0) DBG_REC[14] = (long long)__builtin_amdgcn_s_memrealtime();
#ifdef FT_DIAG
    long long* dbg = has_dbg ? DBG_REC : nullptr;
#endif

    const unsigned fastw = EXACT ? (unsigned)(L - 1) : (FASTW ? 0u : wrap_magic(L));
    // this layer's forward weight block (the conv2 table padded along the pair direction of this mu): the loads are
    // issued FIRST and land under the plaquette loads and the sincos; issued behind them (where they are consumed) the
    // stage pays two memory latencies in a row.  Unconditional, clamped: straight-line code keeps the waits counted.
    constexpr int NWC = (LF_LDS + NT - 1) / NT;
    double wv[NWC];
    {
        const double* wb = w + (mu == 0 ? WFWD0 : WFWD1);
#pragma unroll
        for (int k = 0; k < NWC; ++k) wv[k] = ldu(wb, (unsigned)min(tid + k * NT, LF_LDS - 1));
    }
    // the tile's own links for the final link update: issued now, consumed in the last stage
    double xv0 = 0.0, xv1 = 0.0;
    if ((has_y || has_pout) && tid < N3) {
        const int r = fdiv<TC>(tid), c = tid - r * TC;
        if (EXACT || (i0 + r < L && j0 + c < L)) {
            const unsigned at = (unsigned)(mul24(i0 + r, L) + j0 + c);
            if (pin) xv0 = ldu(pin, at); else { xv0 = ldu(x0, at); xv1 = ldu(x1, at); }
        }
    }

    // ---- plaquettes and net input, only where somebody reads them; small weights -> LDS ------------------
    // Of the 22 x 22 window the net reads (cos P, sin P) on the FROZEN lines only (conv1 runs on the frozen taps, the
    // constant (1, 0) of the other lines is in its bias table) and the transform reads P at the tile's own active sites:
    // 264 + 64 of 484 plaquettes.  One task per thread:
    //   tid < NFT:          frozen site: position a = tid / NFL along the lines, k-th frozen line across them (the frozen
    //                       lines come in pairs, classes 1 and 2, every four lines: x = ps + 4 (k >> 1) + (k & 1))
    //   NFT <= tid < +NA:   own active site (the map of the conv3 / transform stages)
    //   the next 36:        the constant (1, 0) at the non-frozen sites of the window corner that conv1's leftover pairs
    //                       read with all 18 taps (rows and columns 16 .. 21)
    // Five waves instead of eight run the loads, the wrapped addresses and the sincos; passive and foreign active
    // plaquettes are never formed.
    constexpr int NFL = (R0C + 3) / 4 * 2, NFT = R0C * NFL;
    static_assert(G::R0R == R0C && NFT + NA + 36 <= NT && R0C == 22, "stage-0 task list");
    {
        const int ps = off == 3 ? -1 : off;                                // first line of the first frozen pair (class 1: x = off mod 4)
        int r = 0, c = 0;
        bool fz = false, ao = false;
        if (tid < NFT) {
            const int a = fdiv<NFL>(tid), k = tid - a * NFL;
            const int x = ps + 4 * (k >> 1) + (k & 1);
            fz = (unsigned)x < (unsigned)R0C;
            r = mu == 0 ? a : x; c = mu == 0 ? x : a;
        } else if (tid < NFT + NA) {
            const int a = tid - NFT;
            r = 3 + (mu == 0 ? a / (TC / 4) : off + 4 * (a / TC));
            c = 3 + (mu == 0 ? off + 4 * (a % (TC / 4)) : a % TC);
            ao = true;
        } else if (tid < NFT + NA + 36) {
            const int k = tid - (NFT + NA), rr = 16 + fdiv<6>(k), cc = 16 + (k - 6 * fdiv<6>(k));
            const int sel = ((mu == 0 ? cc : rr) - 3 - off) & 3;
            if (sel != 1 && sel != 2) { sIn[rr * R0C + cc] = 1.0; sIn[PS0 + rr * R0C + cc] = 0.0; }
        }
        if (fz || ao) {
            const int iL = mul24(wrap_line<FASTW, EXACT>(i0 - 3 + r, L, fastw), L), ipL = mul24(wrap_line<FASTW, EXACT>(i0 - 2 + r, L, fastw), L);
            const int j = wrap_line<FASTW, EXACT>(j0 - 3 + c, L, fastw), jp = wrap_line<FASTW, EXACT>(j0 - 2 + c, L, fastw);
            const double p = pin ? ldu(pin, (unsigned)(iL + j))
                                 : ldu(x0, (unsigned)(iL + j)) - ldu(x1, (unsigned)(iL + j)) - ldu(x0, (unsigned)(iL + jp)) + ldu(x1, (unsigned)(ipL + j));
            const int at = r * R0C + c;
            // one sincos per task: of P where the plaquette is frozen (the net input), of P / 2 at an own active site (the
            // transform needs it) -- left in the plaquette plane next to P, in the two slots of the following lines
            // (stripe classes 1, 2), whose own plaquettes nobody reads
            double sn = 0.0, cs = 1.0;
            if (fz || !REV) ft_sincos(fz ? p : 0.5 * p, &sn, &cs);
            if (ao) {
                sP[at] = p;
                if (!REV) { const int st = mu == 0 ? 1 : R0C; sP[at + st] = cs; sP[at + 2 * st] = sn; }
            } else {
                sIn[at] = cs;
                sIn[PS0 + at] = sn;
                if (has_stash && (unsigned)(r - 3) < (unsigned)(EXACT ? TR : min(TR, L - i0)) &&
                    (unsigned)(c - 3) < (unsigned)(EXACT ? TC : min(TC, L - j0))) {         // the net input of the tile's own frozen sites
                    double* cs_ = uniform_at(A.stash, 18u * (unsigned)A.B * (unsigned)n + bn);
                    const unsigned fi = (unsigned)stash_frozen_idx(i0 + r - 3, j0 + c - 3, L, mu, off);
                    sts<NTS>(cs_, fi, cs); sts<NTS>(cs_, fi + (unsigned)(n >> 1), sn);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NWC; ++k) {
        const int t = tid + k * NT;
        if (t < LF_SIZE) sW[t] = wv[k]; else if (t < LF_LDS) sP1[t - LF_SIZE] = wv[k];
    }
    lds_barrier();
    STAMP(1);
#ifdef FT_DIAG
    if (A.dbg_stop == 1) return;
#endif

    // stash of this lane's output channels 2 g, 2 g + 1, g = lane >> 4 (fixed for the kernel): act' channel-minor
    // (one 16-byte store per site), and so is h (training)
    const int rmax = EXACT ? TR : min(TR, L - i0), cmax = EXACT ? TC : min(TC, L - j0);    // tile sites inside the lattice
    const Stash sv = has_stash ? stash_view(A.stash, A.B, b, n) : Stash{};
    typedef double double2_t __attribute__((ext_vector_type(2)));
    // uniform plane bases (SGPRs) + this lane's channel pair as part of the 32-bit element index
    double* const st_d1 = has_stash ? uniform_ptr(sv.d1, 0) : nullptr;
    double* const st_d2 = has_stash ? uniform_ptr(sv.d2, 0) : nullptr;
    double* const st_h1 = has_stash ? uniform_ptr(sv.h1, 0) : nullptr;
    double* const st_h2 = has_stash ? uniform_ptr(sv.h2, 0) : nullptr;
    const unsigned stg = 2u * (unsigned)(lane >> 4);

    // ---- conv1 (2 -> 8) + act on the tile+2 window ---------------------------
    // The net input is (cos P, sin P) on the frozen lines and the constant (1, 0) elsewhere, so only HALF of conv1's
    // taps carry data.  Output sites are paired ACROSS the stripe lines (columns c, c + 1 of one row for mu = 0, rows for
    // mu = 1): a pair's input window is then four consecutive lines, exactly two of them frozen -- which two depends
    // only on the parity of the pair's position u across the lines.  K = 2 frozen lines x 3 taps along x 2 channels =
    // 3 MFMA steps instead of 6 (lane group g: channel g & 1, frozen line g >> 1; step t: tap along the lines); the
    // constant lines are folded into the bias (table BC, flow_common.h).  MFMA tiles hold pairs of ONE parity:
    //     pair index inside a parity class = uu * R1 + v  (u = 2 uu + parity across, v along the lines),
    // 6 whole tiles per class; the last few pairs (v >= R1 - 4 of the last uu: 12.5 tiles in all, a 13th would land
    // on one SIMD as a 4th) run on the VALU, all 18 taps, on the waves that own one tile only.
    static_assert(R1R == R1C && R1R % 2 == 0, "square window, pairs across either direction");
    constexpr int R1 = R1C, NU1 = R1 / 2, NPC = (NU1 / 2) * R1;          // pairs per parity class (NU1 even)
    static_assert(NU1 % 2 == 0, "as many even as odd pair positions");
    constexpr int NTC = NPC / 16, NREMP = NPC - NTC * 16, NREM1 = 2 * (2 * NREMP) * 8;   // whole tiles per class; leftover outputs
    static_assert(2 * NTC <= 2 * NW && NREM1 <= NT / 2 && NREMP <= R1, "tile rounds; leftover outputs go to the upper waves");
    // Tile -> pairs: tiles 0 .. NU1 - 1 hold ONE pair position u = tile across the lines and v = lane & 15 along them, so
    // that everything that depends on u -- the window lines, the stripe class, the LDS row / column of the outputs, the
    // stash row and its bounds -- is wave-uniform (scalar instructions) and what depends on the lane is the same for both
    // tiles of a wave; tiles NU1, NU1 + 1 take v = 16 .. 19 of the even / odd u < 8 (four u per tile), and u = 8, 9 there are
    // the leftover pairs below.  (Tiles of 16 consecutive pairs of a (u, v) enumeration cost ~50 VALU instructions of index
    // arithmetic per tile and wave: division by the window width, per-lane parity, per-lane bounds.)
    static_assert(R1 == 20 && NU1 == 10 && NTC == 6 && NREMP == 4, "conv1 tile map: 16 + 4 positions along the lines, 10 across");
    {
        const int g = lane >> 4, i = lane & 15, cN = i & 7, dd = i >> 3;
        const int sbase = ((mu == 0 ? j0 : i0) - 3 - off) & 3;           // stripe class of the input window's first line
        const int lstep = mu == 0 ? 1 : R0C;                             // LDS step across the lines / along them
        const int astep = mu == 0 ? R0C : 1;
        auto conv1_tile = [&](int u, int v, int par) {
            const int s4 = (sbase + 2 * par) & 3;                        // class of the pair window's first line (wave-uniform)
            const int fl = ((g >> 1) + 1 - s4) & 3;                      // this lane group's frozen line of the window: classes 1, 2
            const double* a0 = sIn + (g & 1) * PS0 + (2 * u + fl) * lstep + v * astep;
            const double* wp = sP1 + cN + (g & 1) * 48 + (fl + 1 - dd) * 8;
            // z[q]: channel 2 g + (q & 1) at site q >> 1 of the pair; bias + constant lines from BC[s4][site][channel] are the
            // accumulator's start value
            const double* bc = sP1 + LF_BC + s4 * 16 + 2 * g;
            double4_t acc = {bc[0], bc[1], bc[8], bc[9]};
#pragma unroll
            for (int t = 0; t < 3; ++t)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wp[t * 96], a0[t * astep], acc, 0, 0, 0);
            double z[4] = {acc[0], acc[1], acc[2], acc[3]};
            double h[4], d[4];
            act_eval4(z, act, h, d);
            const int r = mu == 0 ? v : 2 * u, c = mu == 0 ? 2 * u : v;  // site 0 in h1-window coordinates; site 1 = next column / row
            const int ds = mu == 0 ? 1 : RS1;
            double* ph = sH1 + 2 * g * PS1 + r * RS1 + c;
            ph[0] = h[0]; ph[PS1] = h[1]; ph[ds] = h[2]; ph[PS1 + ds] = h[3];
            if (FT_RECOMP_D1 ? stash_h : has_stash) {   // act'(z1) (and h1) of the tile's own sites
                const int r0 = r - 2, c0 = c - 2, r1 = mu == 0 ? r0 : r0 + 1, c1 = mu == 0 ? c0 + 1 : c0;
                const int at = mul24(i0 + r0, L) + j0 + c0, dat = mu == 0 ? 1 : L;
                // act'(z1) of a mu = 0 layer: transposed site index (FT_D1_T): the lanes of a tile run down a column
                const int atd = (FT_D1_T && mu == 0) ? mul24(j0 + c0, L) + i0 + r0 : at, datd = (FT_D1_T && mu == 0) ? L : dat;
                if ((unsigned)r0 < (unsigned)rmax && (unsigned)c0 < (unsigned)cmax) {
                    if (!FT_RECOMP_D1) sts2<NTS>(st_d1, 8u * (unsigned)atd + stg, double2_t{d[0], d[1]});
                    if (stash_h) sts2<NTS>(st_h1, 8u * (unsigned)at + stg, double2_t{h[0], h[1]});
                }
                if ((unsigned)r1 < (unsigned)rmax && (unsigned)c1 < (unsigned)cmax) {
                    if (!FT_RECOMP_D1) sts2<NTS>(st_d1, 8u * (unsigned)(atd + datd) + stg, double2_t{d[2], d[3]});
                    if (stash_h) sts2<NTS>(st_h1, 8u * (unsigned)(at + dat) + stg, double2_t{h[2], h[3]});
                }
            }
        };
        static_assert(NW <= NU1 && 2 * NW >= NU1 + 2, "first round: whole-u tiles only; second round: the rest");
        conv1_tile(wave, i, wave & 1);
        const int T = wave + NW;
        if (T < NU1) conv1_tile(T, i, T & 1);
        else if (T < NU1 + 2) conv1_tile(2 * (i >> 2) + (T - NU1), 16 + (i & 3), T - NU1);
    }
    if (NREM1 > 0 && tid >= NT / 2 && tid < NT / 2 + NREM1) {
        // leftover pairs of either class: uu = NU1 / 2 - 1, v = R1 - NREMP ..; thread = (site, output channel), all 18 taps
        const int idx = tid - NT / 2, co = idx & 7, site = idx >> 3, pair = site >> 1, sd = site & 1;
        const int par = pair >= NREMP, v = R1 - NREMP + (pair - par * NREMP), u = NU1 - 2 + par;
        const int r = mu == 0 ? v : 2 * u + sd, c = mu == 0 ? 2 * u + sd : v;
        const double* in = sIn + r * R0C + c;
        const double* wp = sP1 + (co >> 1) + 4 * (co & 1) + 8;         // P1[a][ci][line + 1][row], ft_chan(row) = co
        double z = sW[LF_B0 + co];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)                              // mu = 0: P1[ky][ci][kx + 1], mu = 1: P1[kx][ci][ky + 1]
                z = fma(in[ci * PS0 + (tp / 3) * R0C + tp % 3], wp[(mu == 0 ? tp / 3 : tp % 3) * 96 + ci * 48 + (mu == 0 ? tp % 3 : tp / 3) * 8], z);
        double h, d;
        act_eval(z, act, h, d);
        sH1[co * PS1 + r * RS1 + c] = h;
        if (has_stash) {
            const int rr = r - 2, cc = c - 2;
            if ((unsigned)rr < (unsigned)rmax && (unsigned)cc < (unsigned)cmax) {
                const int at = mul24(i0 + rr, L) + j0 + cc;
                if (!FT_RECOMP_D1) sts<NTS>(st_d1, 8u * (unsigned)((FT_D1_T && mu == 0) ? mul24(j0 + cc, L) + i0 + rr : at) + (unsigned)co, d);
                if (stash_h) sts<NTS>(st_h1, 8u * (unsigned)at + (unsigned)co, h);
            }
        }
    }
    lds_barrier();
    STAMP(2);
#ifdef FT_DIAG
    if (A.dbg_stop == 2) return;
#endif

    // ---- conv2 (8 -> 8) + act on the live lines of the tile+1 window -----------------------------
    // conv3 reads h2 only within one site of an active line, so every 4th line of the window (stripe
    // class 2, first one d0) is never used: the pair sites enumerate the (at most 14 of 18) live lines only and pair
    // ALONG them (rows for mu = 0, columns for mu = 1): 126 pairs = 8 M tiles, one per wave, instead
    // of 11 over the full window (16 x 16 tiles).  Dead lines of h2 / act'(z2) (LDS and stash) stay unwritten; the
    // backward kernels write an exact 0 there instead of multiplying.
    constexpr int NLC = R2C - R2C / 4, NLR = R2R - R2R / 4;            // live columns / rows at most
    static_assert(((R2R / 2) * NLC + 15) / 16 == NW && (NLR * (R2C / 2) + 15) / 16 == NW,
                  "one conv2 tile per wave: its epilogue holds a workgroup barrier");
    const int d0 = ((off + 3) - (mu == 0 ? j0 : i0)) & 3;               // first dead line of the window
    auto conv2_epi = [&](int g, bool ok, int r, int c, int dr, int dc, double (&z)[4]) {
        // sites (r, c) and (r + dr, c + dc) in window coordinates; z[q]: channel 2 g + (q & 1), site q >> 1
        double h[4], d[4];                                  // the bias came in through the accumulator (bias2 below)
        act_eval4(z, act, h, d);
        lds_barrier();                                      // h2 overwrites h1: every wave has finished its MFMA reads
        const int so = dr * R2C + dc;
        if (ok) {
            double* ph = sH2 + 2 * g * PS2 + r * R2C + c;
            ph[0] = h[0]; ph[PS2] = h[1]; ph[so] = h[2]; ph[PS2 + so] = h[3];
        }
        if (has_stash && ok) {                                // act'(z2) (and h2) of the tile's own sites
            const int at = mul24(i0 + r - 1, L) + j0 + c - 1;
            // act'(z2): live lines only (FT_D2_C, stash_live_idx); the pair's second site is one line-step along the stripe lines
            const int a2 = FT_D2_C ? stash_live_idx<EXACT>(i0 + r - 1, j0 + c - 1, L, mu, off) : at;
            const int da2 = FT_D2_C ? (mu == 0 ? dr * 3 * (L >> 2) + dc : dr * L + dc) : dr * L + dc;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if ((unsigned)(r - 1 + q * dr) < (unsigned)rmax && (unsigned)(c - 1 + q * dc) < (unsigned)cmax) {
                    const int aq = at + q * (dr * L + dc);
                    sts2<NTS>(st_d2, 8u * (unsigned)(a2 + q * da2) + stg, double2_t{d[2 * q], d[2 * q + 1]});
                    if (stash_h) sts2<NTS>(st_h2, 8u * (unsigned)aq + stg, double2_t{h[2 * q], h[2 * q + 1]});
                }
        }
    };
    double4_t bias2;
    { const double b0 = sW[LF_B1 + 2 * (lane >> 4)], b1 = sW[LF_B1 + 2 * (lane >> 4) + 1]; bias2 = double4_t{b0, b1, b0, b1}; }
    // Tile -> pairs: tile T (one per wave) holds the pair line T -- pair row T for mu = 0, pair column T for mu = 1 -- with the
    // live lines l = lane & 15 < NLC across it; the ninth pair line is spread over the two spare lanes of tiles 0 .. 6
    // (l = 2 T + lane - NLC).  The pair line is then wave-uniform for 14 of 16 lanes and the live-line index a per-lane
    // constant, instead of a division of the pair number by the line count per lane and tile.
    constexpr int NPL = R2R / 2;                                        // pair lines (rows or columns of pairs)
    static_assert(NLC == 14 && NLR == 14 && NPL == 9 && NPL - 1 == NW && 2 * (NW - 1) == NLC, "conv2 tile map");
    // one tile per wave: this lane's pair (pair line pl, live-line index -> window line wl) once, for the operand address and
    // for the epilogue alike
    int pl, wl;
    bool pok;
    {
        const int i = lane & 15;
        const bool own = i < NLC;
        pl = own ? wave : NPL - 1;
        wl = live_line(own ? i : 2 * wave + i - NLC, d0);
        pok = (own || wave < NW - 1) && wl < R2C;
    }
    if (mu == 0) {
        // B[k = (tap = ky4 * 3 + kx, ci)][n = (co, dd)] = W1[co][ci][ky4 - dd][kx]; pairs = rows (2 pr, 2 pr + 1)
        mfma_stage<KConv2Row, 16 * NW, RS1, PS1, false, false, FT_NCH ? FT_NCH : 1>(sH1, sW + LF_P2, wave, lane,
            [&](int) { return 2 * pl * RS1 + min(wl, R2C - 1); },
            [&](int g, int, bool, double (&z)[4], int) { conv2_epi(g, pok, 2 * pl, wl, 1, 0, z); },
#ifdef FT_DIAG
            dbg ? dbg + 11 : nullptr,
#else
            nullptr,
#endif
            bias2);
    } else {
        // B[k = (tap = ky * 4 + kx4, ci)][n = (co, dd)] = W1[co][ci][ky][kx4 - dd]; pairs = columns (2 pc, 2 pc + 1)
        mfma_stage<KConv2Col, 16 * NW, RS1, PS1, false, false, FT_NCH ? FT_NCH : 1>(sH1, sW + LF_P2, wave, lane,
            [&](int) { return min(wl, R2R - 1) * RS1 + 2 * pl; },
            [&](int g, int, bool, double (&z)[4], int) { conv2_epi(g, pok, wl, 2 * pl, 0, 1, z); },
            nullptr, bias2);
    }
    // conv3's 27 weights of this wave's input channel are wave-uniform: scalar loads straight from the weight block
    // (constant address space -> s_load, SGPR operands of the FMAs) instead of 27 LDS reads per wave; issued ahead of
    // the barrier so that they land while the slower waves finish conv2
    typedef const double __attribute__((address_space(4))) * cdptr;
    double w3[27];
    {
        cdptr w3p = (cdptr)(size_t)(w + (mu == 0 ? WFWD0 : WFWD1) + LF_W2 + wave * 9);
        // wide scalar loads written out (8 + 1 doubles per output channel: s_load_dwordx16 + s_load_dwordx2): the pass that would
        // merge 27 s_load_dwordx2 is off for this kernel (FT_LDS_B64), and 27 scalar-memory instructions per wave are not free
        typedef double double8c_t __attribute__((ext_vector_type(8)));
        typedef const double8c_t __attribute__((address_space(4))) * cd8ptr;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double8c_t v8 = *(cd8ptr)(w3p + k * 72);
#pragma unroll
            for (int tp = 0; tp < 8; ++tp) w3[k * 9 + tp] = v8[tp];
            w3[k * 9 + 8] = w3p[k * 72 + 8];
        }
    }
    lds_barrier();
    STAMP(3);
#ifdef FT_DIAG
    if (A.dbg_stop == 3) return;
#endif

    // ---- conv3 (8 -> 3) at the NA active sites; one input channel per wave ------
    // active site `lane`: mu=0 columns off+4m, mu=1 rows off+4m (tile origin % 4 == 0)
    const int ar = mu == 0 ? lane / (TC / 4) : off + 4 * (lane / TC);
    const int ac = mu == 0 ? off + 4 * (lane % (TC / 4)) : lane % TC;
    const int ai = i0 + ar, aj = j0 + ac;
    const bool alane = lane < NA;
    const bool avalid = alane && (EXACT || ((ai < L) && (aj < L)));
    if (alane) {
        double acc[3] = {0.0, 0.0, 0.0};
        const int ci = wave;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const double v = sH2[ci * PS2 + (ar + ky) * R2C + ac + kx];
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[k] = fma(v, w3[k * 9 + ky * 3 + kx], acc[k]);
            }
#pragma unroll
        for (int k = 0; k < 3; ++k) sST[(wave * 3 + k) * NAS + lane] = acc[k];
    }
    lds_barrier();
    STAMP(4);
#ifdef FT_DIAG
    if (A.dbg_stop == 4) return;
#endif

    if (REV) {
        // ---- inverse of the tan-mixture transform: solve mean_k y_k(P) = wrap(P' - t) per active site by
        //      safeguarded Newton (the map is monotone with derivative mean_k 1/D_k; the reference bisects
        //      to a global 1e-6, layers.py:294-320), started from the target (s ~ 0: identity) ------------
        if (wave == 0) {
            double dl = 0.0, lj = 0.0, xsol = 0.0;
            if (alane) {
                const double Pn = sP[(ar + 3) * R0C + ac + 3];
                double sk[NMIX], tval = sW[LF_B2 + NMIX];
#pragma unroll
                for (int k = 0; k < NMIX; ++k) sk[k] = sW[LF_B2 + k];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
#pragma unroll
                    for (int k = 0; k < NMIX; ++k) sk[k] += sST[(q * 3 + k) * NAS + lane];
                    tval += sST[(q * 3 + NMIX) * NAS + lane];
                }
                double ea[2 * NMIX], eo[2 * NMIX];
#pragma unroll
                for (int k = 0; k < NMIX; ++k) { ea[2 * k] = sk[k]; ea[2 * k + 1] = -sk[k]; }
                ft_expN<2 * NMIX>(ea, eo);
                const double target = ft_wrap(Pn - tval);
                double lo = -FT_PI, hi = FT_PI, xs = target, fp = 1.0;
                bool done = false;
                for (int it = 0; it < 200 && !done; ++it) {
                    double sn, cs;
                    ft_sincos(xs / 2, &sn, &cs);
                    const double th = sn / cs;
                    double f = 0.0;
                    fp = 0.0;
#pragma unroll
                    for (int k = 0; k < NMIX; ++k) {
                        f += ft_wrap_pm_pi(2 * ft_atan(eo[2 * k] * th));
                        fp += 1.0 / (eo[2 * k + 1] * cs * cs + eo[2 * k] * sn * sn);
                    }
                    f /= NMIX; fp /= NMIX;
                    const double err = target - f;
                    if (fabs(err) <= A.tol) { done = true; break; }
                    if (err > 0) lo = xs; else hi = xs;
                    double xn = xs + err / fp;
                    if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
                    if (xn == xs) done = true;
                    xs = xn;
                }
                dl = xs - Pn;
                xsol = xs;
                lj = -(log(fp));                                     // log J of the inverse = -log mean_k 1/D_k at the root
            }
            if (avalid) sDL[ar * TC + ac] = has_pout ? xsol : dl;
            if (want_logj) {
                const double tot = ft_wave_sum(avalid ? lj : 0.0);
                if (lane == 0) A.logj_part[(size_t)b * ntiles + tile] = tot;
            }
        }
        lds_barrier();
        if (has_y && tid < N3) {
            const int r = fdiv<TC>(tid), c = tid - r * TC;
            const int i = i0 + r, j = j0 + c;
            if (EXACT || (i < L && j < L)) {
                double v0 = xv0, v1 = xv1;
                if (ft_stripe(i, j, mu, off) == 0) {
                    const double d = sDL[tid];
                    if (mu == 0) v0 = ft_wrap(d + v0); else v1 = ft_wrap(-d + v1);
                }
                double* y0 = uniform_at(A.y, 2u * bn);
                const unsigned at = (unsigned)(mul24(i, L) + j);
                stu(y0, at, v0); stu(y0, (unsigned)n + at, v1);
            }
        }
        if (has_pout && tid < N3) {                                    // plaquette-level inverse: x1 at the active sites, fx elsewhere
            const int r = fdiv<TC>(tid), c = tid - r * TC;
            const int i = i0 + r, j = j0 + c;
            if (EXACT || (i < L && j < L)) A.pout[(size_t)b * n + mul24(i, L) + j] = ft_stripe(i, j, mu, off) == 0 ? sDL[tid] : xv0;
        }
        return;
    }

    // ---- tan-mixture transform: wave k evaluates mixture component k -------------
    //   y_k = wrap(2 atan(e^{s_k} tan(P/2))),  D_k = e^{-s_k} cos^2(P/2) + e^{s_k} sin^2(P/2),
    //   log J = log(sum_k 1/D_k) - log K   (= logsumexp_k(-log D_k) - log K of layers.py:85-90)
    double Pa = 0.0;
    if (wave < NMIX && alane) {
        Pa = sP[(ar + 3) * R0C + ac + 3];
        double sk = sW[LF_B2 + wave];
#pragma unroll
        for (int q = 0; q < 8; ++q) sk += sST[(q * 3 + wave) * NAS + lane];
        const int pst = mu == 0 ? 1 : R0C;
        const double cs = sP[(ar + 3) * R0C + ac + 3 + pst], sn = sP[(ar + 3) * R0C + ac + 3 + 2 * pst];   // of P / 2
        // e^{-s} = 1 / e^{s} and the two quotients by reciprocal + one correction step (ft_rcp): |s| is O(1) for any
        // usable flow and clamped to +-700 by ft_exp, so every denominator is far from the ends of the range
        const double es = ft_exp(sk), ems = ft_rcp(es);
        const double cs2 = cs * cs, sn2 = sn * sn, sincs = sn * cs;
        const double invD = ft_rcp(ems * cs2 + es * sn2);
        sT2[(wave * TQ + 1) * NAS + lane] = invD;
        // tan(P/2) = sn / cs: |cs| can be tiny (P near +-pi) -- a true division keeps the correctly rounded quotient there
        sT2[(wave * TQ + 0) * NAS + lane] = ft_wrap_pm_pi(2 * ft_atan(es * (sn / cs)));
        if (has_stash && avalid) {
            // coefficients of the transform's adjoint (struct Stash): the backward kernel then needs no
            // plaquettes, sincos or exp at the active sites of its tile+3 window
            const double sinP = 2.0 * sincs, invD2 = invD * invD;
            // component-major [k][n/4][A B C E]: the wave of component k writes whole cache lines (site-major, two waves
            // wrote the two 32-byte halves of every 64-byte record at different times)
            double* tc = uniform_ptr(sv.tc, (size_t)wave * n);
            const unsigned ti = 4u * (unsigned)stash_active_idx(ai, aj, L, mu);
            sts2<NTS>(tc, ti, double2_t{sinP * invD / NMIX,                                             // A_k
                                   (ems * cs2 - es * sn2) * invD2});                               // B_k
            sts2<NTS>(tc, ti + 2u, double2_t{invD / NMIX,                                               // C_k
                                        sinP * 0.5 * (es - ems) * invD2});                         // E_k
        }
    }
    if (wave == NMIX && alane) {                                 // t on an otherwise idle wave
        double tv = sW[LF_B2 + NMIX];
#pragma unroll
        for (int q = 0; q < 8; ++q) tv += sST[(q * 3 + NMIX) * NAS + lane];
        sP[(ar + 3) * R0C + ac + 3 + 3 * (mu == 0 ? 1 : R0C)] = tv;
    }
    lds_barrier();
    STAMP(5);
#ifdef FT_DIAG
    if (A.dbg_stop == 5) return;
#endif

    {
        if (wave == 0) {
            double ysum = 0.0, si = 0.0;
#pragma unroll
            for (int k = 0; k < NMIX; ++k) { ysum += sT2[(k * TQ) * NAS + (lane & (NAS - 1))]; si += sT2[(k * TQ + 1) * NAS + (lane & (NAS - 1))]; }
            const double tval = sP[(ar + 3) * R0C + ac + 3 + 3 * (mu == 0 ? 1 : R0C)];
            const double newP = ft_wrap(ysum / NMIX + tval);
            if (avalid) sDL[ar * TC + ac] = has_pout ? newP : newP - Pa;
            if (want_logj) {                                       // force sweeps do not ask for log J
                const double lj = avalid ? log(si) - log((double)NMIX) : 0.0;
                const double tot = ft_wave_sum(lj);
                if (lane == 0) A.logj_part[(size_t)b * ntiles + tile] = tot;
            }
        }
        lds_barrier();
        if (has_y && tid < N3) {
            const int r = fdiv<TC>(tid), c = tid - r * TC;
            const int i = i0 + r, j = j0 + c;
            if (EXACT || (i < L && j < L)) {
                double v0 = xv0, v1 = xv1;
                if (ft_stripe(i, j, mu, off) == 0) {
                    const double d = sDL[tid];
                    if (mu == 0) v0 = ft_wrap(d + v0); else v1 = ft_wrap(-d + v1);
                }
                double* y0 = uniform_at(A.y, 2u * bn);
                const unsigned at = (unsigned)(mul24(i, L) + j);
                stu(y0, at, v0); stu(y0, (unsigned)n + at, v1);
            }
        }
        if (has_pout && tid < N3) {                                    // plaquette-level map: P' at the active sites, P elsewhere
            const int r = fdiv<TC>(tid), c = tid - r * TC;
            const int i = i0 + r, j = j0 + c;
            if (EXACT || (i < L && j < L)) A.pout[(size_t)b * n + mul24(i, L) + j] = ft_stripe(i, j, mu, off) == 0 ? sDL[tid] : xv0;
        }
        STAMP(6);
        if (has_dbg && tid == 0) DBG_REC[15] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

int g_variant = 1;

}  // namespace

namespace {
#define FWD_LAUNCH_(...) hipLaunchKernelGGL((k_flow_fwd<__VA_ARGS__>), grid, dim3(NT), 0, s, a.x, a.wint, a.y, a.stash, a.logj_part, a.B, a.L, hoa, a)
// the silu / other-activation instances of a shape, generic in everything else
#define FWD_LAUNCH(...) do { if (a.act != FTHMC_ACT_SILU) FWD_LAUNCH_(__VA_ARGS__, false, 0); else FWD_LAUNCH_(__VA_ARGS__, true, 0); } while (0)
// ... and with the sweep specializations (SWEEP template parameter): the forward map on the tiled-exactly shapes, silu
#define FWD_LAUNCH_SWEEPS(...) do { if (a.act != FTHMC_ACT_SILU) FWD_LAUNCH_(__VA_ARGS__, false, 0); \
                             else if (force_sweep && !a.dbg && nt_stash) FWD_LAUNCH_(__VA_ARGS__, true, 5); \
                             else if (force_sweep && !a.dbg) FWD_LAUNCH_(__VA_ARGS__, true, 1); \
                             else if (force_sweep) FWD_LAUNCH_(__VA_ARGS__, true, 2); else if (action_sweep) FWD_LAUNCH_(__VA_ARGS__, true, 3); \
                             else if (train_sweep && nt_stash) FWD_LAUNCH_(__VA_ARGS__, true, 6); \
                             else if (train_sweep) FWD_LAUNCH_(__VA_ARGS__, true, 4); \
                             else FWD_LAUNCH_(__VA_ARGS__, true, 0); } while (0)
template <bool REV> void launch_fwd(const fthmc::FlowLayerArgs& a, dim3 grid, hipStream_t s) {
    constexpr int TR = fthmc::MF_FWD_TR, TC = fthmc::MF_FWD_TC;
    const unsigned hoa = (unsigned)a.off | (unsigned)a.act << 8 | (a.pout ? FWD_HAS_POUT : 0u) | (a.dbg ? FWD_HAS_DBG : 0u) | (a.pin ? FWD_HAS_PIN : 0u);
    // the SWEEP = 1 instances serve exactly this combination (a layer of a force sweep)
    const bool force_sweep = !REV && a.y && a.stash && !a.stash_h && !a.logj_part && !a.pin && !a.pout;
    const bool action_sweep = !REV && a.y && !a.stash && a.logj_part && !a.pin && !a.pout && !a.dbg;
    const bool train_sweep = !REV && a.y && a.stash && a.stash_h && a.logj_part && !a.pin && !a.pout && !a.dbg;
    // A stash the backward will not find in a cache again is stored past the caches: one layer's stash of a training shard (32
    // chains of L = 256: 587 MB) is more than the 256 MB Infinity Cache, and its stores with the hint took 4.5 % off the whole
    // training step (6.92 -> 6.6 ms).  At the headline shape (40 MB per layer and chain group) the backward finds the LAST layers'
    // stash in the caches -- the hint on every layer cost 0.8 % there, on all but the last two (stash_far) it gains 0.3-1.1 %
    // (profiles/r06_ab_nontemporal_stash.txt).
    const bool nt_stash = FT_NT_STASH && a.stash && ((FT_NT_STASH <= 2 && fthmc::flow_stash_doubles(a.B, a.L, a.stash_h != 0) * sizeof(double) >= FT_NT_MIN_BYTES) ||
                                                      (FT_NT_STASH >= 2 && a.stash_far));
    const bool fast = wrap_fast_ok(a.L, TR, TC);
    const bool exact = fast && a.L % TR == 0 && a.L % TC == 0 && (a.L & (a.L - 1)) == 0;
    if (a.mu == 0) {
        if (exact) { if constexpr (REV) FWD_LAUNCH(TR, TC, true, REV, 0, true); else FWD_LAUNCH_SWEEPS(TR, TC, true, REV, 0, true); }
        else if (fast) FWD_LAUNCH(TR, TC, true, REV, 0, false);
        else FWD_LAUNCH(TR, TC, false, REV, 0, false);
    } else {
        if (exact) { if constexpr (REV) FWD_LAUNCH(TR, TC, true, REV, 1, true); else FWD_LAUNCH_SWEEPS(TR, TC, true, REV, 1, true); }
        else if (fast) FWD_LAUNCH(TR, TC, true, REV, 1, false);
        else FWD_LAUNCH(TR, TC, false, REV, 1, false);
    }
    (void)force_sweep; (void)action_sweep; (void)train_sweep; (void)nt_stash;
}
}  // namespace

namespace fthmc {

void set_flow_variant(int v) { g_variant = v; }
int get_flow_variant() { return g_variant; }

int launch_flow_fwd_mfma(const FlowLayerArgs& a, hipStream_t s) {
    if (!flow_shape_ok(a.B, a.L, a.off)) return FTHMC_ERR_ARG;
    if (!flow_stash_fits32(a.B, a.L, a.stash_h != 0)) return FTHMC_ERR_UNSUPPORTED;                  // 32-bit plane offsets (uniform_at)
    // 16 x 16 tiles, three workgroups per CU (SmemF)
    const dim3 grid = xcd_grid(a.B, (a.L + MF_FWD_TR - 1) / MF_FWD_TR, (a.L + MF_FWD_TC - 1) / MF_FWD_TC);
    launch_fwd<false>(a, grid, s);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_flow_rev_mfma(const FlowLayerArgs& a, hipStream_t s) {
    if (!flow_shape_ok(a.B, a.L, a.off)) return FTHMC_ERR_ARG;
    if (!flow_stash_fits32(a.B, a.L, a.stash_h != 0)) return FTHMC_ERR_UNSUPPORTED;                  // 32-bit plane offsets (uniform_at)
    const dim3 grid = xcd_grid(a.B, (a.L + MF_FWD_TR - 1) / MF_FWD_TR, (a.L + MF_FWD_TC - 1) / MF_FWD_TC);
    launch_fwd<true>(a, grid, s);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}

}  // namespace fthmc
