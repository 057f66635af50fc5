// Coupling-layer backward from stashed activations, gather form.
//
// A workgroup (512 threads) owns a TR x TC tile of one chain and produces the COMPLETE
// plaquette-gradient of its own sites:  gP_out[p] = gP_in[p] + (layer's contribution at p).
// Walking the adjoint backwards from the tile, each stage needs its input on a window one site
// larger:  conv1^T at the tile <- gz1 on tile+1 <- conv2^T <- gz2 on tile+2 <- conv3^T <- g_out at
// the active sites of tile+3, which is the tan-mixture adjoint there: four FMAs per site on the
// coefficients the forward kernel stashed (struct Stash), so this kernel never touches the links.
// Compared with the scatter form (k_flow_bwd_stash: each tile pushes the gradient of its own
// active sites out to a tile+3 window of partial sums that k_gather_gp adds up afterwards):
//   * the expensive stages shrink (conv2^T runs on tile+1 instead of tile+2, conv1^T on the
//     tile instead of tile+3), no zero-padded planes, no bounds masks in any conv stage,
//   * no partial windows in HBM and no gather launch,
//   * every window position is a real (wrapped) lattice site, so lattices smaller than a tile
//     need no special case; only the final store and the weight-gradient sums look at validity,
//   * LDS drops enough for 16 x 16 tiles at two workgroups per CU (halo factors 1.27 / 1.56
//     instead of 1.9 / 1.4 on the two stashed windows).
// act'(z2) and act'(z1) never pass through LDS: each thread loads the values it will multiply by
// (conv3^T task layout, conv2^T epilogue layout) at kernel start, and they land while the transform
// adjoint, conv3^T's accumulation and the MFMA loop run: the barriers in between wait for LDS traffic
// only (s_waitcnt lgkmcnt), not for global loads in flight.
//
// Reference: autograd of GaugeEquivCouplingLayer.forward (fthmc/utils/layers.py:196-202,348-371)
// as used by ft_force (qed_helpers.py:226-242) and train_step (train.py:162-228).
#include "flow_mfma_common.h"

namespace {

using namespace fthmc;
using namespace fthmc_flow;

constexpr int cmax_(int a, int b) { return a > b ? a : b; }
constexpr unsigned BWD_HAS_UPLINK = 1u << 16, BWD_HAS_GLOGJ = 1u << 17, BWD_HAS_GZ = 1u << 18, BWD_HAS_DBG = 1u << 19;   // flags in the hoa word

template <int TR, int TC> struct SmemG {
    static constexpr int W3R = TR + 6, W3C = TC + 6, N3W = W3R * W3C;   // g_out window (active sites only)
    static constexpr int W2R = TR + 4, W2C = TC + 4, N2W = W2R * W2C;   // act'(z2) -> gz2
    static constexpr int W1R = TR + 2, W1C = TC + 2, N1W = W1R * W1C;   // act'(z1) -> gz1; cos/sin
    static constexpr int N3 = TR * TC;
    // gz2 rows are RS2 apart in LDS: odd, so that the 16 lanes of a conv2^T operand read (one per window row, below)
    // fall into 16 different banks
    static constexpr int RS2 = W2C + 1;
    static constexpr int PS2 = ps_round16(W2R * RS2), PS1 = ps_round(N1W);   // gz2: MFMA operand; gz1: read by four channel lanes per site
    // (cos, sin) of the frozen plaquettes: on the tile+2 window when conv1 is recomputed here (its MFMA operand), else tile+1
    static constexpr int WIR = FT_RECOMP_D1 ? TR + 4 : W1R, WIC = FT_RECOMP_D1 ? TC + 4 : W1C, NIW = WIR * WIC;
    static constexpr int PSI = FT_RECOMP_D1 ? ps_round16(NIW) : PS1, IOFF = FT_RECOMP_D1 ? 2 : 1;   // IOFF: window origin = tile - IOFF
    // active lines of the g_out window: every 4th column (mu = 0) or row (mu = 1)
    static constexpr int NLC = (W3C + 3) / 4, NLR = (W3R + 3) / 4;
    static constexpr int NSLOT = cmax_(W3R * NLC, NLR * W3C);           // transform tasks
    static constexpr int NTT = (NSLOT + 63) / 64 * 64;                  // threads that run them (last waves)
    static constexpr int GO = 0;                                        // [3][N3W] g(s0, s1, t)
    static constexpr int GZ2 = GO + 3 * N3W;                            // [8][PS2] gz2
    static constexpr int D1 = GZ2 + 8 * PS2;                            // [8][PS1] gz1
    static constexpr int IN = D1 + 8 * PS1;                             // [2][PSI] cos, sin
    static constexpr int DIR = IN + 2 * PSI;                            // [N3] layer's contribution at own sites
    static constexpr int P1 = D1;                                       // [LF_P1_SIZE] conv1 tables, until conv2^T's epilogues write gz1 there
    static constexpr int SW = DIR + N3;                                 // [LB_SIZE] backward weight block (flow_common.h)
    static constexpr int SIZE = SW + LB_SIZE;
    static_assert(W1R % 2 == 0, "row pairs");
    static_assert(NTT <= NT && 2 * N3 <= NT && N1W <= NT && NIW <= NT && LF_P1_SIZE <= NT && LF_P1_SIZE <= 8 * PS1, "thread maps");
    static_assert(2 * SIZE * 8 <= 160 * 1024, "two workgroups per CU (160 KB of LDS on gfx950)");
};

// A.gz (training): the kernel additionally writes the gradients wrt the layer's pre-activations at the tile's own
// sites -- gz2, gz1 (channel-minor) and the transform adjoint g_out at the active sites -- for k_flow_wgrad
// (flow_wgrad.hip), which turns them into weight gradients; nothing else changes.
// MU: the layer's stripe direction as a compile-time constant (as in k_flow_fwd: the selects on it fold, each kernel carries
// one of the two conv2^T code paths).
// EXACT: the tiles divide the lattice and L is a power of two (64, 128, 256): every own site is a lattice site, the lattice-edge
// tests fold away, a window line wraps by one v_and.
// SWEEP >= 1 (FS): the launch is a layer of a FORCE sweep (all 160 backward launches of a trajectory): upstream gradient = the plaquette-gradient
// field, dL/dlogJ a constant, no pre-activation gradients written -- the tests on those (uniform) conditions and the code behind
// the other outcomes leave the kernel; SWEEP = 1 also drops the cycle stamps of the diagnostic launches (those run SWEEP = 2).
// 0: whatever the argument block says (standalone VJPs, training).
template <int TR, int TC, bool FASTW, int MU, bool EXACT, int SWEEP>
// Hot arguments as explicit scalars ahead of the argument block (kernarg preload, as k_flow_fwd): 13 dwords.
// hoa = off | act << 8 | flags << 16.
__global__ FT_LDS_B64 __launch_bounds__(NT, 4) void k_flow_bwd_gather(const double* hw, double* hstash, const double* hup_gp, double* hgp_out, double hglogj_const,
                                                                       int hB, int hL, unsigned hoa, FlowLayerArgs A0) {
    FlowLayerArgs A = A0;
    A.wint = hw; A.stash = hstash; A.up_gp = hup_gp; A.gp_out = hgp_out; A.glogj_const = hglogj_const; A.B = hB; A.L = hL;
    A.off = (int)(hoa & 0xffu); A.act = (int)((hoa >> 8) & 0xffu);
    constexpr bool FS = SWEEP >= 1, TS = SWEEP == 3;        // TS: a layer of a TRAINING sweep = a force sweep that also writes A.gz
    const bool has_uplink = !FS && (hoa & BWD_HAS_UPLINK) != 0, has_glogj = !FS && (hoa & BWD_HAS_GLOGJ) != 0,
               has_gz = TS || (!FS && (hoa & BWD_HAS_GZ) != 0), has_dbg = (SWEEP == 0 || SWEEP == 2) && (hoa & BWD_HAS_DBG) != 0;
    const bool has_upgp = FS || A.up_gp != nullptr;
    using S = SmemG<TR, TC>;
    constexpr int W3C = S::W3C, N3W = S::N3W, W2R = S::W2R, W2C = S::W2C, N2W = S::N2W;
    constexpr int W1R = S::W1R, W1C = S::W1C, N3 = S::N3, PS1 = S::PS1, PS2 = S::PS2, RS2 = S::RS2;
    constexpr int WIC = S::WIC, NIW = S::NIW, PSI = S::PSI, IOFF = S::IOFF;
    __shared__ __attribute__((aligned(16))) double sm[S::SIZE];
    double* sGO = sm + S::GO;   double* sGZ2 = sm + S::GZ2;  double* sD1 = sm + S::D1;
    double* sIn = sm + S::IN;   double* sDir = sm + S::DIR;  double* sW = sm + S::SW;

    const int tid = threadIdx.x;
    __builtin_assume(tid >= 0 && tid < NT);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < NW);                          // lets the tile maps below fold their wave-uniform cases (T = wave + NW it)
    constexpr int mu = MU;
    const int L = A.L, off = A.off;
    // what the launchers guarantee, said to the compiler (ranges decide between 24-bit and 64-bit index arithmetic)
    __builtin_assume(off >= 0 && off < 4 && L >= 4 && L <= 8192 && (L & 3) == 0);
    if (EXACT) __builtin_assume(L >= 16 && (L & (L - 1)) == 0);
    const int n = L * L;
    const int nti_ = (L + TR - 1) / TR, ntj_ = (L + TC - 1) / TC;
    BlockTile bt;
    if (!block_tile(A.B, nti_, ntj_, bt)) return;               // padding blocks when B % 8 != 0 (whole block exits)
    const int b = bt.b, tile = bt.tile, ntiles = nti_ * ntj_;
    const int i0 = bt.ti * TR, j0 = bt.tj * TC;
    __builtin_assume(i0 >= 0 && i0 < L && j0 >= 0 && j0 < L && b >= 0 && b < (1 << 20) && A.B > 0 && A.B <= (1 << 20));
    const int rmax = EXACT ? TR : min(TR, L - i0), cmax = EXACT ? TC : min(TC, L - j0);    // own sites inside the lattice
    const double* __restrict__ w = A.wint;
    const Stash sv = stash_view(A.stash, A.B, b, n);
    // stash planes of this chain (struct Stash), as kernel-argument base + uniform offset
    const unsigned bn = (unsigned)b * (unsigned)n, Bn = (unsigned)A.B * (unsigned)n;       // 32-bit plane offsets: uniform_at()
    const double* __restrict__ st1 = uniform_at(A.stash, 8u * bn);
    const double* __restrict__ stc = uniform_at(A.stash, 16u * Bn + 2u * bn);
    const double* __restrict__ scs = uniform_at(A.stash, 18u * Bn + bn);
    (void)sv;
    // training outputs of this chain (kernels.h: FlowLayerArgs::gz)
    double* const gz2o = has_gz ? A0.gz + (size_t)b * 17 * n : nullptr;
    double* const gz1o = has_gz ? gz2o + (size_t)8 * n : nullptr;
    double* const goo = has_gz ? gz2o + (size_t)16 * n : nullptr;
    long long* dbg = has_dbg ? A0.dbg + ((size_t)b * ntiles + tile) * 16 : nullptr;
#define STAMP(k) do { if (has_dbg && tid == 0) dbg[k] = (long long)__builtin_readcyclecounter(); } while (0)
    STAMP(0);
    if (has_dbg && tid == 0) dbg[14] = (long long)__builtin_amdgcn_s_memrealtime();   // 14, 15: the 100 MHz counter (see k_flow_fwd)

    // wrapped lattice coordinates of window lines, relative to the tile origin (rows premultiplied by L)
    const unsigned wmagic = EXACT ? (unsigned)(L - 1) : (FASTW ? 0u : wrap_magic(L));
    auto wi = [&](int k) { return wrap_line<FASTW, EXACT>(i0 + k, L, wmagic); };
    auto WI = [&](int k) { return mul24(wi(k), L); };
    auto WJ = [&](int k) { return wrap_line<FASTW, EXACT>(j0 + k, L, wmagic); };

    // ---- load phase.  Every load is unconditional, from a clamped address (idle lanes read element 0
    //      and drop it): straight-line code lets the compiler count outstanding loads (s_waitcnt
    //      vmcnt(N)) instead of draining them all at the first use after a branch.  Loads return in
    //      issue order, so what the first stage consumes is issued first and the big act' operands last.
    constexpr int NWC = (LB_SIZE + NT - 1) / NT;
    double wsw[NWC];
#pragma unroll
    for (int k = 0; k < NWC; ++k) wsw[k] = ldu(w + (mu == 0 ? WBWD1 : WBWD), (unsigned)min(tid + k * NT, LB_SIZE - 1));
    // (1) transform tasks on the last waves: active site `a` of the tile+3 window, both mixture components
    const int ta = tid - (NT - S::NTT);
    const int c0 = (off - (j0 - 3)) & 3, r0 = (off - (i0 - 3)) & 3;    // first active column / row of the window
    int tr3 = 0, tc3 = 0;
    bool ttask = false;
    if (ta >= 0) {
        if (mu == 0) { tr3 = fdiv<S::NLC>(ta); tc3 = c0 + 4 * (ta - tr3 * S::NLC); ttask = tr3 < S::W3R && tc3 < W3C; }
        else { const int m = fdiv<W3C>(ta); tc3 = ta - m * W3C; tr3 = r0 + 4 * m; ttask = tr3 < S::W3R; }
        if (!ttask) { tr3 = 3; tc3 = 3; }                                // any valid site
    } else { tr3 = 3; tc3 = 3; }
    typedef double double2_t __attribute__((ext_vector_type(2)));
    auto ldu2 = [](const double* base, unsigned idx) {               // 16-byte load, scalar base + 32-bit element offset
        const double2_t* p = reinterpret_cast<const double2_t*>(reinterpret_cast<const char*>(base) + idx * 8u);
        return FT_NT_LOAD >= 2 ? __builtin_nontemporal_load(p) : *p;
    };
    double tcv[4 * NMIX], ag[2];
    const double cb = has_glogj ? A0.glogj[b] : A.glogj_const;
    {
        const int i = wi(tr3 - 3), j = WJ(tc3 - 3);
        // wave-uniform base + 32-bit per-lane offset everywhere: the address costs no VALU op per load
        const unsigned ia = (unsigned)stash_active_idx(i, j, L, mu);
#pragma unroll
        for (int q = 0; q < 4 * NMIX; q += 2) {                             // [k][n/4][A B C E] (struct Stash): 16 bytes per load
            const double2_t t2 = ldu2(stc + (size_t)(q >> 2) * n, ia * 4u + (q & 3));
            tcv[q] = t2.x; tcv[q + 1] = t2.y;
        }
        // upstream gradient: a link field (first layer of a standalone call) or the plaquette-gradient field
        const double* gsrc = uniform_at(A.up_gp, bn);
        if (has_uplink) {                      // a real branch (the empty asm keeps it one): the rarely used pointer is fetched from the
            asm volatile("" ::: "memory");     // argument block only here, and nobody waits for that fetch in a force sweep
            gsrc = uniform_at(A0.up_link, 2u * bn + (unsigned)(mu * n));
        }
        const int iL = mul24(i, L);
        ag[0] = ldu(gsrc, (unsigned)(iL + j));
        ag[1] = ldu(gsrc, (unsigned)(mu == 0 ? iL + WJ(tc3 - 4) : WI(tr3 - 4) + j));   // unused with up_link
    }
    // (2) cos / sin of the frozen plaquettes: of the tile's own sites (the adjoint of the net input), and with FT_RECOMP_D1 of
    //     the whole tile+2 window (the input of the recomputed conv1), one window site per thread; a non-frozen site feeds the
    //     net the constant (1, 0).  Idle and non-frozen lanes read element 0 and drop it.
    int fwin = 0;                                                        // this thread's slot in the (cos, sin) planes
    bool ftask = false;
    double fcs, fsn;
    if (FT_RECOMP_D1) {
        const int r = fdiv<WIC>(tid), c = tid - r * WIC;
        ftask = tid < NIW;
        fwin = ftask ? tid : 0;
        const int cls = ((mu == 0 ? j0 + c : i0 + r) - IOFF - off) & 3;
        const bool frozen = ftask && (cls == 1 || cls == 2);
        const unsigned ic = frozen ? (unsigned)stash_frozen_idx(wi(r - IOFF), WJ(c - IOFF), L, mu, off) : 0u;
        fcs = ldu(scs, ic); fsn = ldu(scs + (n >> 1), ic);
        if (!frozen) { fcs = 1.0; fsn = 0.0; }
    } else {
        int fr1 = 1, fc1 = 1;                                            // tile+1 coordinates of this thread's site
        if (tid < N3 / 2) {
            int r, c;
            if (mu == 0) { r = fdiv<TC / 2>(tid); const int h = tid - r * (TC / 2); c = 4 * (h >> 1) + ((off + 1 + (h & 1)) & 3); }
            else { const int hh = fdiv<TC>(tid); c = tid - hh * TC; r = 4 * (hh >> 1) + ((off + 1 + (hh & 1)) & 3); }
            fr1 = r + 1; fc1 = c + 1; ftask = true;
        }
        fwin = fr1 * W1C + fc1;
        const unsigned ic = ftask ? (unsigned)stash_frozen_idx(wi(fr1 - 1), WJ(fc1 - 1), L, mu, off) : 0u;
        fcs = ldu_j(scs, ic); fsn = ldu_j(scs + (n >> 1), ic);
        if (!ftask) { fcs = 1.0; fsn = 0.0; }
    }
    // the conv1 tables of the forward block (P1, BC: flow_common.h), one entry per thread
    double wp1 = 0.0;
    if (FT_RECOMP_D1) wp1 = ldu(w + (mu == 0 ? WFWD0 : WFWD1) + LF_P1, (unsigned)min(tid, LF_P1_SIZE - 1));
    // (3) upstream gradient of the own sites (pass-through term)
    const int orr = fdiv<TC>(tid), occ = tid - orr * TC;
    const bool ovalid = tid < N3 && (EXACT || (orr < rmax && occ < cmax));
    double gpin;
    {
        const double* gsrc = has_upgp ? uniform_at(A.up_gp, bn) : scs;   // no pass-through without up_gp
        gpin = ldu_j(gsrc, ovalid ? (unsigned)(mul24(i0 + orr, L) + j0 + occ) : 0u);
        if (!ovalid || !has_upgp) gpin = 0.0;
    }
    // (4) act'(z2) and act'(z1) go straight into the registers of the thread that multiplies by them
    //     (conv3^T task layout / conv2^T epilogue layout), never through LDS: they are issued here with
    //     everything else and land while the transform adjoint, conv3^T and the MFMA loop run.
    // conv3^T task = (two sites of the same line class, half of the 8 channels)
    static_assert(W2R % 2 == 0 && W2C % 2 == 0 && N2W <= NT, "site pairs, one round");
    constexpr int NPR = N2W / 2;
    static_assert(NPR <= NT / 2, "one half of the channels per half of the workgroup");
    const int c3half = wave >= NW / 2;                                   // wave-uniform: plane bases stay in SGPRs
    const bool c3task = (tid & (NT / 2 - 1)) < NPR;
    const int c3u = c3task ? (tid & (NT / 2 - 1)) : 0;
    int c3r, c3c;
    if (mu == 0) { c3r = fdiv<W2C>(c3u); c3c = c3u - c3r * W2C; }             // (r, c), (r + W2R/2, c)
    else { c3r = fdiv<W2C / 2>(c3u); c3c = c3u - c3r * (W2C / 2); }         // (r, c), (r, c + W2C/2)
    double d2v[2][4];
    {
        // both sites of a task sit on the same line class; on a dead line (no active site within reach, conv3^T writes an
        // exact 0 there) the stash holds nothing: those lanes all read element 0 (one cache line) instead of a window row
        const bool c3live = ((mu == 0 ? c3c + 2 - c0 : c3r + 2 - r0) & 3) <= 2;
#if FT_D2_C
        // live lines only (stash_live_idx): the two sites share their stripe line, i.e. its compact index
        const int lx = stash_live_line<EXACT>(mu == 0 ? WJ(c3c - 2) : wi(c3r - 2), L, off);
        const int goA = !c3live ? 0 : mu == 0 ? mul24(wi(c3r - 2), 3 * (L >> 2)) + lx : mul24(lx, L) + WJ(c3c - 2);
        const int goB = !c3live ? 0 : mu == 0 ? mul24(wi(c3r + W2R / 2 - 2), 3 * (L >> 2)) + lx : mul24(lx, L) + WJ(c3c + W2C / 2 - 2);
#else
        const int goA = c3live ? WI(c3r - 2) + WJ(c3c - 2) : 0;
        const int goB = !c3live ? 0 : mu == 0 ? WI(c3r + W2R / 2 - 2) + WJ(c3c - 2) : WI(c3r - 2) + WJ(c3c + W2C / 2 - 2);
#endif
        // channel-minor stash (struct Stash): the task's four channels of a site are 32 contiguous bytes
        const double* pl = uniform_at(A.stash, 8u * (Bn + bn) + (unsigned)(c3half * 4));
        const unsigned oA = ft_off32((unsigned)goA * 8u), oB = ft_off32((unsigned)goB * 8u);       // byte offsets of the two records
        auto ldu2o = [](const double* base, unsigned o) {
            const double2_t* p = reinterpret_cast<const double2_t*>(reinterpret_cast<const char*>(base) + o);
            return FT_NT_LOAD >= 2 ? __builtin_nontemporal_load(p) : *p;
        };
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
            const double2_t va = ldu2o(pl + k, oA), vb = ldu2o(pl + k, oB);
            d2v[0][k] = va.x; d2v[0][k + 1] = va.y; d2v[1][k] = vb.x; d2v[1][k + 1] = vb.y;
        }
    }
    // conv2^T pairs its output sites ACROSS the stripe lines -- columns (c, c + 1) of one row for mu = 0, rows (r, r + 1)
    // of one column for mu = 1 -- so the pair's input window is four consecutive lines of gz2, exactly one of which is
    // dead (conv3^T wrote zeros there): its six K steps are skipped, 18 of 24 remain.  Which of the four it is depends on
    // the parity of the pair's position u across the lines only, so an MFMA tile holds pairs of ONE parity:
    //     tiles 0 .. NU-1:  u = tile, positions v = 0 .. 15 along the lines        (NU = 9 pairs across, NV = 18 along)
    //     tiles NU, NU+1:   the remaining v = 16, 17 of the even / of the odd u
    // The epilogue's lane (g = lane >> 4, i = lane & 15) of tile T = wave + 8 it owns pair i of the tile, channels
    // 2 g and 2 g + 1, both sites of the pair.
    static_assert(TR == 16 && TC == 16, "conv2^T tile map: 16 positions along the lines + 2");
    constexpr int NU = W1C / 2, NTILE1 = NU + 2, NIT1 = (NTILE1 + NW - 1) / NW;
    static_assert(NIT1 == 2 && W1R == W1C, "two rounds of conv2^T tiles");
    int pu[NIT1], pv[NIT1];                                              // pair position (across, along); rows / columns of site 0:
    bool pok[NIT1];                                                      //   mu = 0: (pv, 2 pu)   mu = 1: (2 pu, pv)   in tile+1 coordinates
    double d1v[NIT1][4];
#pragma unroll
    for (int it = 0; it < NIT1; ++it) {
        const int T = wave + NW * it, i = lane & 15;
        if (T < NU) { pu[it] = T; pv[it] = i; pok[it] = true; }
        else { pu[it] = 2 * (i >> 1) + (T - NU); pv[it] = 16 + (i & 1); pok[it] = T < NTILE1 && pu[it] < NU; if (!pok[it]) { pu[it] = 0; pv[it] = 0; } }
        const int ra = mu == 0 ? pv[it] : 2 * pu[it], ca = mu == 0 ? 2 * pu[it] : pv[it];
        // mu = 0: the act'(z1) plane is stored transposed (FT_D1_T, flow_mfma_common.h): site index j L + i
        const int ga = (FT_D1_T && mu == 0) ? mul24(WJ(ca - 1), L) + wi(ra - 1) : WI(ra - 1) + WJ(ca - 1);
        const int gb = mu == 0 ? (FT_D1_T ? mul24(WJ(ca), L) + wi(ra - 1) : WI(ra - 1) + WJ(ca)) : WI(ra) + WJ(ca - 1);
        if (!FT_RECOMP_D1) {
            const unsigned og = 2u * (unsigned)(lane >> 4);                  // channels 2 g, 2 g + 1: one 16-byte load per site
            const double2_t va = ldu2(st1, (unsigned)ga * 8u + og), vb = ldu2(st1, (unsigned)gb * 8u + og);
            d1v[it][0] = va.x; d1v[it][1] = va.y; d1v[it][2] = vb.x; d1v[it][3] = vb.y;
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- consume ---------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < NWC; ++k) if (tid + k * NT < LB_SIZE) sW[tid + k * NT] = wsw[k];
    if (ttask) {
        // adjoint of the tan-mixture transform (layers.py:66-90) from the forward's coefficients
        const double gdelta = has_uplink ? (mu == 0 ? ag[0] : -ag[0]) : ag[0] - ag[1];
        const int at = tr3 * W3C + tc3;
        double csum = 0.0, esum = 0.0;
#pragma unroll
        for (int k = 0; k < NMIX; ++k) { csum += tcv[4 * k + 2]; esum += tcv[4 * k + 3]; }
        const double tsum = NMIX * csum;                                 // sum_k 1 / D_k
        double rs = __builtin_amdgcn_rcp(tsum);
        rs = fma(fma(-tsum, rs, 1.0), rs, rs);
        rs = fma(fma(-tsum, rs, 1.0), rs, rs);
        const double cbr = cb * rs;
#pragma unroll
        for (int k = 0; k < NMIX; ++k) sGO[k * N3W + at] = gdelta * tcv[4 * k] + cbr * tcv[4 * k + 1];   // dL/ds_k
        sGO[NMIX * N3W + at] = gdelta;                                   // dL/dt
        const int r = tr3 - 3, c = tc3 - 3;
        if ((unsigned)r < (unsigned)TR && (unsigned)c < (unsigned)TC) {
            sDir[r * TC + c] = gdelta * (csum - 1.0) - cbr * esum;
            if (goo && r < rmax && c < cmax) {                           // training: g_out of the own active sites
                static_assert(NMIX == 2, "g_out record: dL/ds_0, dL/ds_1, dL/dt, 0");
                double* po = goo + 4 * (size_t)stash_active_idx(i0 + r, j0 + c, L, mu);
                *reinterpret_cast<double2_t*>(po) = double2_t{gdelta * tcv[0] + cbr * tcv[1], gdelta * tcv[4] + cbr * tcv[5]};
                *reinterpret_cast<double2_t*>(po + 2) = double2_t{gdelta, 0.0};
            }
        }
    }
    if (ftask) { sIn[fwin] = fcs; sIn[PSI + fwin] = fsn; }
    if (FT_RECOMP_D1 && tid < LF_P1_SIZE) sm[S::P1 + tid] = wp1;
#ifndef FT_DIAG
    if (has_dbg && lane == 0) dbg[6 + wave] = (long long)__builtin_readcyclecounter();   // arrival at the first barrier
#endif
    lds_barrier();
    STAMP(1);
#ifdef FT_DIAG
    if (A.dbg_stop == 1) return;
#endif

    // ---- conv3^T on the VALU: g_out lives on the active lines, so of the 9 taps of a site at most 3
    //      (one line) contribute; times act'(z2) -> gz2 in place ------------------------------------
    // the 36 weights of a task are read once and serve both sites (the stage is bound by LDS reads per FMA)
    if (c3task) {
        const int half = c3half, r = c3r, c = c3c;
        const int s2off = mu == 0 ? (W2R / 2) * RS2 : W2C / 2;
        const int s = r * RS2 + c;
        // source = site - (ky - 1, kx - 1): window coordinates (r + 2 - ky, c + 2 - kx) of tile+3
        const int ksel = mu == 0 ? (c + 2 - c0) & 3 : (r + 2 - r0) & 3;   // the one kx (mu=0) / ky (mu=1)
        const int s3off = mu == 0 ? (W2R / 2) * W3C : W2C / 2;            // second site in tile+3 coordinates
        double acc0[4] = {0.0, 0.0, 0.0, 0.0}, acc1[4] = {0.0, 0.0, 0.0, 0.0};
        if (ksel <= 2) {
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                double wv[3][4], g0[3], g1[3];
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    const int ky = mu == 0 ? kk : ksel, kx = mu == 0 ? ksel : kk;
                    const int at = (r + 2 - ky) * W3C + c + 2 - kx;
                    g0[kk] = sGO[co * N3W + at]; g1[kk] = sGO[co * N3W + at + s3off];
#pragma unroll
                    for (int k = 0; k < 4; ++k) wv[kk][k] = sW[LB_W2 + (co * 8 + half * 4 + k) * 9 + ky * 3 + kx];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 3; ++kk)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc0[k] = fma(g0[kk], wv[kk][k], acc0[k]);
                        acc1[k] = fma(g1[kk], wv[kk][k], acc1[k]);
                    }
            }
        }
        // dead lines (no active site within reach) get an exact 0, whatever the stash holds there
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double* pz = sGZ2 + (half * 4 + k) * PS2 + s;
            acc0[k] = ksel <= 2 ? d2v[0][k] * acc0[k] : 0.0;
            acc1[k] = ksel <= 2 ? d2v[1][k] * acc1[k] : 0.0;
            pz[0] = acc0[k];
            pz[s2off] = acc1[k];
        }
        if (gz2o) {                                                      // training: gz2 of the tile's own sites, 32 bytes per task and site
            const int ra = r - 2, ca = c - 2, rb = mu == 0 ? ra + W2R / 2 : ra, cb_ = mu == 0 ? ca : ca + W2C / 2;
            if ((unsigned)ra < (unsigned)rmax && (unsigned)ca < (unsigned)cmax) {
                double* po = gz2o + 8 * (size_t)(mul24(i0 + ra, L) + j0 + ca) + 4 * half;
                *reinterpret_cast<double2_t*>(po) = double2_t{acc0[0], acc0[1]}; *reinterpret_cast<double2_t*>(po + 2) = double2_t{acc0[2], acc0[3]};
            }
            if ((unsigned)rb < (unsigned)rmax && (unsigned)cb_ < (unsigned)cmax) {
                double* po = gz2o + 8 * (size_t)(mul24(i0 + rb, L) + j0 + cb_) + 4 * half;
                *reinterpret_cast<double2_t*>(po) = double2_t{acc1[0], acc1[1]}; *reinterpret_cast<double2_t*>(po + 2) = double2_t{acc1[2], acc1[3]};
            }
        }
    }
    if (FT_RECOMP_D1) {
        // ---- act'(z1) of this lane's conv2^T pairs, recomputed: conv1 (2 -> 8) on the frozen taps exactly as the forward
        //      kernel runs it (flow_fwd.hip: pairs across the stripe lines, K = 2 frozen lines x 3 taps x 2 channels = 3 MFMA
        //      steps, the constant lines in the bias table BC), on the pair map of conv2^T, so that the result lands in the
        //      registers of the lane that multiplies by it.  Operands: the (cos, sin) planes of the tile+2 window.
        const int g = lane >> 4, i = lane & 15, cN = i & 7, dd = i >> 3;
        const double* sP1 = sm + S::P1;
        const int sbase = ((mu == 0 ? j0 : i0) - IOFF - off) & 3;        // stripe class of the input window's first line
        const int lstep = mu == 0 ? 1 : WIC, astep = mu == 0 ? WIC : 1;  // LDS step across the lines / along them
#pragma unroll
        for (int it = 0; it < NIT1; ++it) {
            const int T = wave + NW * it;
            if (T >= NTILE1) break;
            const int par = (T < NU ? T : T - NU) & 1;                   // parity of the tile's pair positions across the lines
            const int s4 = (sbase + 2 * par) & 3;                        // class of the pair window's first line (wave-uniform)
            const int fl = ((g >> 1) + 1 - s4) & 3;                      // this lane group's frozen line of the four
            const double* a0 = sIn + (g & 1) * PSI + (2 * pu[it] + fl) * lstep + pv[it] * astep;
            const double* wp = sP1 + cN + (g & 1) * 48 + (fl + 1 - dd) * 8;
            const double* bc = sP1 + LF_BC + s4 * 16 + 2 * g;            // as the forward: the bias is the accumulator's start value
            double4_t acc = {bc[0], bc[1], bc[8], bc[9]};
#pragma unroll
            for (int t = 0; t < 3; ++t)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wp[t * 96], a0[t * astep], acc, 0, 0, 0);
            double z[4] = {acc[0], acc[1], acc[2], acc[3]};
            double h_[4];
            act_eval4(z, A.act, h_, d1v[it]);
        }
    }
    lds_barrier();
    STAMP(2);
#ifdef FT_DIAG
    if (A.dbg_stop == 2) return;
#endif

    // ---- conv2^T (MFMA), times act'(z1) -> gz1 -------------------------------------------------
    // W[k = (tap, co)][n = (ci, dd)]: the flipped, transposed, padded table T2 of this mu (flow_common.h)
    {
        const int g = lane >> 4, i = lane & 15;
        const double* wp = sW + LB_T2 + KConv2Row::wlane(g, i & 7, i >> 3);        // the lane part is the same for both K orders
        const int kd0 = ((mu == 0 ? c0 : r0) + 1) & 3;                             // dead window line of the even pairs (odd: + 2)
#pragma unroll
        for (int it = 0; it < NIT1; ++it) {
            const int T = wave + NW * it;
            if (T >= NTILE1) break;
            const int kd = (kd0 + 2 * (T < NU ? T & 1 : T - NU)) & 3;              // wave-uniform
            const double* a0 = sGZ2 + g * PS2 + (mu == 0 ? pv[it] * RS2 + 2 * pu[it] : 2 * pu[it] * RS2 + pv[it]);
            double4_t acc;
            if (mu == 0) {
                switch (kd) {
                    case 0: acc = conv2t_tile<KConv2Col, 4, 0, RS2, PS2>(wp, a0); break;
                    case 1: acc = conv2t_tile<KConv2Col, 4, 1, RS2, PS2>(wp, a0); break;
                    case 2: acc = conv2t_tile<KConv2Col, 4, 2, RS2, PS2>(wp, a0); break;
                    default: acc = conv2t_tile<KConv2Col, 4, 3, RS2, PS2>(wp, a0); break;
                }
            } else {
                switch (kd) {
                    case 0: acc = conv2t_tile<KConv2Row, 3, 0, RS2, PS2>(wp, a0); break;
                    case 1: acc = conv2t_tile<KConv2Row, 3, 1, RS2, PS2>(wp, a0); break;
                    case 2: acc = conv2t_tile<KConv2Row, 3, 2, RS2, PS2>(wp, a0); break;
                    default: acc = conv2t_tile<KConv2Row, 3, 3, RS2, PS2>(wp, a0); break;
                }
            }
#ifdef FT_DIAG      // slots 6..9: wave 0 after the MFMAs / the epilogue of its two tiles; 10, 11: wave 4 (one tile)
            if (dbg && (wave == 0 || wave == 4)) { asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3])); stampx(dbg + (wave == 0 ? 6 + 2 * it : 10)); }
#endif
            if (pok[it]) {
                const int ra = mu == 0 ? pv[it] : 2 * pu[it], ca = mu == 0 ? 2 * pu[it] : pv[it];   // site 0; site 1 = next column / row
                const int ds = mu == 0 ? 1 : W1C;
                double* pd = sD1 + 2 * g * PS1 + ra * W1C + ca;              // MFMA rows g, g + 4 = channels 2 g, 2 g + 1 (ft_chan)
                const double v0 = acc[0] * d1v[it][0], v1 = acc[1] * d1v[it][1], v2 = acc[2] * d1v[it][2], v3 = acc[3] * d1v[it][3];
                pd[0] = v0; pd[PS1] = v1; pd[ds] = v2; pd[PS1 + ds] = v3;
                if (gz1o) {                                                  // training: gz1 of the tile's own sites, 16 bytes per lane and site
                    const int r1 = ra - 1, c1 = ca - 1, r2 = mu == 0 ? r1 : r1 + 1, c2 = mu == 0 ? c1 + 1 : c1;
                    if ((unsigned)r1 < (unsigned)rmax && (unsigned)c1 < (unsigned)cmax)
                        *reinterpret_cast<double2_t*>(gz1o + 8 * (size_t)(mul24(i0 + r1, L) + j0 + c1) + 2 * g) = double2_t{v0, v1};
                    if ((unsigned)r2 < (unsigned)rmax && (unsigned)c2 < (unsigned)cmax)
                        *reinterpret_cast<double2_t*>(gz1o + 8 * (size_t)(mul24(i0 + r2, L) + j0 + c2) + 2 * g) = double2_t{v2, v3};
                }
            }
#ifdef FT_DIAG
            if (dbg && (wave == 0 || wave == 4)) stampx(dbg + (wave == 0 ? 7 + 2 * it : 11));
#endif
        }
    }
    // conv1^T's 18 weights of this wave's hidden channel: scalar loads from the weight block (constant address space)
    typedef const double __attribute__((address_space(4))) * cdptr;
    double w0s[18];
    {
        cdptr wq = (cdptr)(size_t)(w + (mu == 0 ? WBWD1 : WBWD) + LB_W0 + wave * 18);
        // wide scalar loads written out (8 + 8 + 2 doubles): the merging pass is off for this kernel (FT_LDS_B64)
        typedef double double8c_t __attribute__((ext_vector_type(8)));
        typedef const double8c_t __attribute__((address_space(4))) * cd8ptr;
        const double8c_t va = *(cd8ptr)(wq), vb = *(cd8ptr)(wq + 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) { w0s[k] = va[k]; w0s[8 + k] = vb[k]; }
        w0s[16] = wq[16]; w0s[17] = wq[17];
    }
    lds_barrier();
    STAMP(3);
#ifdef FT_DIAG
    if (A.dbg_stop == 3) return;
#endif

    // ---- conv1^T and the (cos, sin) adjoint at the tile's own frozen plaquettes ------------------
    // wave = hidden channel co, lane = two of the tile's N3 / 2 frozen sites: the 18 weights of a channel are wave-uniform
    // (scalar loads, requested ahead of the barrier above) and the gz1 reads of a wave stay inside ONE plane; the sum
    // over the channels goes through LDS (the gz2 planes are free by now)
    static_assert(NW == 8 && N3 / 2 == 2 * 64 && 8 * 2 * (N3 / 2) <= 8 * PS2, "one wave per hidden channel, two sites per lane");
    double* sPart = sGZ2;                                                // [8 co][2: cos, sin][N3 / 2]
    auto frozen_site = [&](int f, int& r, int& c) {
        // mu = 0: 16 rows x 8 frozen columns, f = row + 16 h (a 32-lane group then holds an odd and an even column of 16
        // rows: 32 different banks at row stride 18); mu = 1: 8 frozen rows x 16 columns, f = column + 16 hh
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
    lds_barrier();
    if (tid < N3 / 2) {
        int r, c;
        frozen_site(tid, r, c);
        double gct = 0.0, gst = 0.0;
#pragma unroll
        for (int co = 0; co < 8; ++co) { gct += sPart[(co * 2 + 0) * (N3 / 2) + tid]; gst += sPart[(co * 2 + 1) * (N3 / 2) + tid]; }
        const int at = (r + IOFF) * WIC + c + IOFF;
        sDir[r * TC + c] = -sIn[PSI + at] * gct + sIn[at] * gst;
    }
    lds_barrier();
    STAMP(4);
#ifdef FT_DIAG
    if (A.dbg_stop == 4) return;
#endif

    // ---- gP_out = gP_in + this layer's contribution at the own sites ---------------------------
    if (ovalid) {
        const int cls = ((mu == 0 ? j0 + occ : i0 + orr) - off) & 3;  // 0 active, 1|2 frozen, 3 passive
        uniform_at(A.gp_out, bn)[mul24(i0 + orr, L) + j0 + occ] = gpin + (cls != 3 ? sDir[tid] : 0.0);
    }
    STAMP(5);
    if (has_dbg && tid == 0) dbg[15] = (long long)__builtin_amdgcn_s_memrealtime();
#undef STAMP
}

}  // namespace

namespace fthmc {

int launch_flow_bwd_gather(const FlowLayerArgs& a, hipStream_t s) {
    if (!flow_shape_ok(a.B, a.L, a.off)) return FTHMC_ERR_ARG;
    if (!flow_stash_fits32(a.B, a.L, false)) return FTHMC_ERR_UNSUPPORTED;                  // 32-bit plane offsets (uniform_at)
    const dim3 grid = xcd_grid(a.B, (a.L + MG_TR - 1) / MG_TR, (a.L + MG_TC - 1) / MG_TC);
    const bool fast = wrap_fast_ok(a.L, MG_TR, MG_TC);
    const bool exact = fast && a.L % MG_TR == 0 && a.L % MG_TC == 0 && (a.L & (a.L - 1)) == 0;
    const unsigned hoa = (unsigned)a.off | (unsigned)a.act << 8 | (a.up_link ? BWD_HAS_UPLINK : 0u) | (a.glogj ? BWD_HAS_GLOGJ : 0u) |
                         (a.gz ? BWD_HAS_GZ : 0u) | (a.dbg ? BWD_HAS_DBG : 0u);
    const bool force_sweep = a.up_gp && !a.up_link && !a.glogj && !a.gz, train_sweep = a.up_gp && !a.up_link && !a.glogj && a.gz && !a.dbg;       // what the FS instances serve
#define BWD_LAUNCH_(...) hipLaunchKernelGGL((k_flow_bwd_gather<__VA_ARGS__>), grid, dim3(NT), 0, s, a.wint, a.stash, a.up_gp, a.gp_out, a.glogj_const, a.B, a.L, hoa, a)
#define BWD_LAUNCH(...) do { if (force_sweep && !a.dbg) BWD_LAUNCH_(__VA_ARGS__, 1); else if (force_sweep) BWD_LAUNCH_(__VA_ARGS__, 2); \
                             else if (train_sweep) BWD_LAUNCH_(__VA_ARGS__, 3); \
                             else BWD_LAUNCH_(__VA_ARGS__, 0); } while (0)
    // the sweep specializations exist for the tiled-exactly shapes (L = 64, 128, 256); everything else runs the generic instance
    if (a.mu == 0) {
        if (exact) BWD_LAUNCH(MG_TR, MG_TC, true, 0, true);
        else if (fast) BWD_LAUNCH_(MG_TR, MG_TC, true, 0, false, 0);
        else BWD_LAUNCH_(MG_TR, MG_TC, false, 0, false, 0);
    } else {
        if (exact) BWD_LAUNCH(MG_TR, MG_TC, true, 1, true);
        else if (fast) BWD_LAUNCH_(MG_TR, MG_TC, true, 1, false, 0);
        else BWD_LAUNCH_(MG_TR, MG_TC, false, 1, false, 0);
    }
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}

}  // namespace fthmc
