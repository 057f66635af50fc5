// Small lattices (L <= 16): a whole chain lives in ONE workgroup, and whole sequences of the flowed path run in ONE launch.
//
// The tiled kernels (flow_fwd.hip, flow_bwd_gather.hip) give every 16 x 16 tile of a chain its own workgroup and one
// launch per layer.  At L <= 16 a chain IS one tile: a launch is B workgroups on 256 CUs, each alone on its CU, and a
// trajectory is ~100 dependent launches whose length is the latency of one lone workgroup, most of it halo work the
// periodic lattice does not need (a 22 x 22 plaquette window, 20 x 20 conv1 outputs for 16 x 16 sites).  Here
//   * one 512-thread workgroup per chain holds the links (latent and flowed), the plaquette gradient and every activation
//     plane in LDS as PERIODIC planes: (L + 2) x (L + 2) with a one-site border that the producing stage fills with the
//     wrapped duplicates, so the implicit-GEMM stages keep their constant tap offsets and nothing is computed twice;
//   * a force evaluation (all layers forward with the activation stash, Wilson seed, all layers backward, momentum kick),
//     the leapfrog loop around it, the two effective-action evaluations and the Metropolis step of a trajectory are
//     device-side loops: fthmc_ft_trajectory / _ft_leapfrog / _ft_force / _ft_action / _flow_forward are one launch each;
//   * conv1, conv2 and conv2^T are the same implicit GEMMs on v_mfma_f64_16x16x4_f64 over the same weight blocks
//     (flow_common.h), conv3 / conv3^T / conv1^T / the tan-mixture transform the same VALU stages as in the tiled kernels;
//   * the activation stash keeps the layout of struct Stash (flow_mfma_common.h) in the caller's workspace: it is
//     written and read back by the same workgroup within one launch (L2-resident).
//
// Reference: GaugeEquivCouplingLayer.forward (fthmc/utils/layers.py:196-202, 348-371), ft_action / ft_force
// (fthmc/utils/qed_helpers.py:212-242), the leapfrog and accept step of ipynb/ft_hmc.py:394-435.
#include "flow_mfma_common.h"
#include <stdlib.h>

namespace {

using namespace fthmc;
using namespace fthmc_flow;

typedef double double2_t __attribute__((ext_vector_type(2)));

// timing-only builds (results wrong): bit 0 no stash stores, 1 no border duplicates, 2 identity activation, 3 no weight loads,
// 4 no stash loads in the backward.  tools/small_knobs.sh builds and times them; the product build has FT_KNOB = 0.
#ifndef FT_KNOB
#define FT_KNOB 0
#endif

template <int L> struct GS {
    static constexpr int N = L * L, NA = N / 4, NF = N / 2, NPAIR = N / 2;
    static constexpr int PL = L + 2, RS = PL;                    // periodic plane: one-site border of wrapped duplicates
    static constexpr int PSZ = ps_round16(PL * RS);              // plane stride (MFMA operand planes)
    static constexpr int NAS = 64;
    static_assert(L % 4 == 0 && L >= 8 && L <= 16, "whole-lattice kernel: L = 8, 12, 16");
    static_assert(NA <= 64 && N <= NT / 2 + NT / 2 && (NPAIR + 15) / 16 <= NW, "one MFMA tile per wave, one wave of active sites");
    // LDS plan (doubles)
    static constexpr int X = 0;                                  // [2][N] links being transformed (a sweep runs in place)
    static constexpr int XL = X + 2 * N;                         // [2][N] latent links of the trajectory
    static constexpr int GP = XL + 2 * N;                        // [N] plaquette gradient
    static constexpr int DIR = GP + N;                           // [N] a layer's contribution to it
    static constexpr int PA = DIR + N;                           // [4][NAS] per active site: P, cos P/2, sin P/2
    static constexpr int IN = PA + 4 * NAS;                      // fwd [2][PSZ] cos, sin   | bwd [3][PSZ] g_out
    static constexpr int A8 = IN + 3 * PSZ;                      // fwd [8][PSZ] h1         | bwd gz2, then conv1^T partials
    static constexpr int B8 = A8 + 8 * PSZ;                      // fwd [8][PSZ] h2         | bwd gz1
    static constexpr int ST = B8 + 8 * PSZ;                      // [8][3][NAS] conv3 partials
    static constexpr int T2 = ST + 8 * 3 * NAS;                  // [2 NMIX + 1][NAS] y_k, 1 / D_k per component, t
    static constexpr int SW = T2 + (2 * NMIX + 1) * NAS;         // [2][LF_LDS] weight blocks: the pass in flight, the next pass
    static constexpr int RED = SW + 2 * LF_LDS;
    static_assert(LF_LDS >= LB_SIZE, "one buffer size for forward and backward blocks");
    static constexpr int STT = RED + 16;                         // [8] (S_eff, plaq, Q) of x and of the proposal, K0, log det J
    static constexpr int PROF = STT + 8;                         // [32] cycle sums of a profiling run
    static constexpr int SIZE = PROF + 32;
    static_assert(8 * 2 * NF <= 8 * PSZ, "conv1^T partials fit over gz2");
    static_assert(SIZE * 8 <= 160 * 1024, "LDS of one CU");
};

// value v of site (r, c) into a periodic plane: the site itself and its wrapped duplicates on the border
template <int L, int RS>
__device__ __forceinline__ void put1(double* p, int r, int c, double v) {
    const int m = (r + 1) * RS + c + 1;
    const int rr = r == 0 ? (L + 1) * RS : (r == L - 1 ? 0 : -1);
    const int cc = c == 0 ? L + 1 : (c == L - 1 ? 0 : -1);
    p[m] = v;
    if (FT_KNOB & 2) return;
    if (rr >= 0) p[rr + c + 1] = v;
    if (cc >= 0) p[(r + 1) * RS + cc] = v;
    if (rr >= 0 && cc >= 0) p[rr + cc] = v;
}
// the same for two planes PS apart (the channel pair 2 g, 2 g + 1 of an MFMA lane)
template <int L, int RS, int PS>
__device__ __forceinline__ void put2(double* p, int r, int c, double va, double vb) {
    const int m = (r + 1) * RS + c + 1;
    const int rr = r == 0 ? (L + 1) * RS : (r == L - 1 ? 0 : -1);
    const int cc = c == 0 ? L + 1 : (c == L - 1 ? 0 : -1);
    p[m] = va; p[PS + m] = vb;
    if (FT_KNOB & 2) return;
    if (rr >= 0) { p[rr + c + 1] = va; p[PS + rr + c + 1] = vb; }
    if (cc >= 0) { p[(r + 1) * RS + cc] = va; p[PS + (r + 1) * RS + cc] = vb; }
    if (rr >= 0 && cc >= 0) { p[rr + cc] = va; p[PS + rr + cc] = vb; }
}

// interior only: planes whose readers are VALU stages (conv3, conv3^T, conv1^T) wrap their own indices, no border needed
template <int RS>
__device__ __forceinline__ void put1i(double* p, int r, int c, double v) { p[(r + 1) * RS + c + 1] = v; }
template <int RS, int PS>
__device__ __forceinline__ void put2i(double* p, int r, int c, double va, double vb) {
    const int m = (r + 1) * RS + c + 1;
    p[m] = va; p[PS + m] = vb;
}
// padded-plane row (or column) index of lattice line v - 1 + k, k = 0, 1, 2, wrapped
template <int L>
__device__ __forceinline__ void wrap3(int v, int (&o)[3]) {
    o[0] = (v == 0 ? L - 1 : v - 1) + 1; o[1] = v + 1; o[2] = (v == L - 1 ? 0 : v + 1) + 1;
}

// Global-memory accesses of this kernel go through pointers typed as such.  Its pointers pass through empty asm statements that
// pin them in SGPRs (stash(), gz(), block_of(), the kernarg-segment reads of cold()) and come out GENERIC: every stash / field
// access was a FLAT instruction with a 64-bit VGPR address, and FLAT instructions count on lgkmcnt as well as vmcnt -- every LDS
// wait of a pass then also waited for its stash stores to reach memory.
__device__ __forceinline__ double gld(const double* p) { return *(const FT_G double*)p; }
__device__ __forceinline__ void gst(double* p, double v) { *(FT_G double*)p = v; }
__device__ __forceinline__ void gst2(double* p, double2_t v) { *(FT_G double2_t*)p = v; }
__device__ __forceinline__ double2_t ldg2(const double* p) {
    if (FT_KNOB & 16) return double2_t{0.5, 0.25};
    return *(const FT_G double2_t*)p;
}

// What a pass over one layer needs from the kernel arguments, and nothing else: the full argument block stays in the
// kernarg segment and is read where it is used (cold()), so that a dozen output pointers do not sit in SGPRs (and spill)
// through the whole trajectory.
struct Hot {
    const double* wint;
    double* stash;
    long long* dbg;
    int B, nl, act;
    double* gz;              // training sweep: the pre-activation gradients of every layer (null otherwise)
};
typedef const __attribute__((address_space(4))) SmallArgs ColdArgs;        // the kernarg segment is constant memory: scalar loads
__device__ __forceinline__ ColdArgs& cold() {
    ColdArgs* p = (ColdArgs*)__builtin_amdgcn_kernarg_segment_ptr();                     // the one explicit kernel argument, offset 0
    asm volatile("" : "+s"(p));
    return *p;
}

// the stash values a backward pass multiplies by, loaded one pass ahead
struct BwdPre { double tcv[4 * NMIX], fcs, fsn, d2v[4], d1v[4]; };

// TRAIN: the instance that runs training sweeps (stashes h1, h2, writes the pre-activation gradients); a template parameter so
// that the trajectory / force instances carry none of it (the kernel is register- and SGPR-bound)
template <int L, bool TRAIN> struct Chain {
    using G = GS<L>;
    static constexpr int NWC = (LF_LDS + NT - 1) / NT;         // weight-block doubles per thread (LF_LDS >= LB_SIZE)
    double* sm;
    Hot A;
    int b, tid, lane, wave;
    int wcur;                                                    // which of the two LDS weight buffers the pass in flight reads
    double pfw[NWC];                                             // the next pass's weight block, on its way
    BwdPre pre;                                                  // this backward pass's stash values
    long long last;                                              // profiling runs (A.dbg): time of the previous stamp
    __device__ Chain(double* sm_, const Hot& A_, int b_) : sm(sm_), A(A_), b(b_) {
        tid = threadIdx.x; lane = tid & 63; wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        __builtin_assume(tid >= 0 && tid < NT && wave >= 0 && wave < NW);       // the tile maps fold their wave-uniform cases
        wcur = 0;
        last = A.dbg ? (long long)__builtin_readcyclecounter() : 0;
    }
    // profiling runs only: thread 0 adds the cycles since its previous stamp to slot k of this chain's record
    __device__ __forceinline__ void stamp(int k) {
        if (A.dbg && tid == 0) {
            const long long t = (long long)__builtin_readcyclecounter();
            reinterpret_cast<long long*>(sm + G::PROF)[k] += t - last;      // LDS: a global read-modify-write would be timed itself
            last = (long long)__builtin_readcyclecounter();
        }
    }
    // uniform values re-derived per use on the scalar unit: made opaque, or every per-layer pointer is hoisted out of the
    // layer loops, outlives its SGPRs and comes back through v_readlane (a VALU slot each)
    __device__ __forceinline__ int chain() const { int c = b; asm volatile("" : "+s"(c)); return c; }
    __device__ __forceinline__ double* stash(int l) const {
        double* p = A.stash; int B_ = A.B;
        asm volatile("" : "+s"(p), "+s"(B_));
        return (p) + (size_t)l * ((size_t)B_ * (TRAIN ? 35 : 19) * L * L);   // kernels.h flow_stash_doubles (training: + h1, h2)
    }
    // training sweep: this chain's slice of layer l's pre-activation gradients (kernels.h FlowLayerArgs::gz): gz2 [n][8],
    // gz1 [n][8] channel-minor, g_out [n/4][4]
    __device__ __forceinline__ double* gz(int l) const {
        double* p = A.gz; int B_ = A.B, c = b;
        asm volatile("" : "+s"(p), "+s"(B_), "+s"(c));
        return (p) + ((size_t)l * B_ + c) * (size_t)(17 * L * L);
    }
    __device__ __forceinline__ double* sW() const { return sm + G::SW + wcur * LF_LDS; }

    // ---- weight blocks: the pass in flight reads one LDS buffer while the next pass's block travels global -> registers
    //      (issued at the top of the pass) -> the other buffer (committed at its end, behind the pass's last LDS reads of it)
    static __device__ __forceinline__ const double* block_of(const double* wint, bool bwd, int l) {
        asm volatile("" : "+s"(wint));
        const int mu = l & 1;
        return (wint) + (size_t)l * FLOW_WINT + (bwd ? (mu == 0 ? WBWD1 : WBWD) : (mu == 0 ? WFWD0 : WFWD1));
    }
    __device__ __forceinline__ void weights_issue(bool bwd, int l) {
        int tid = this->tid;
        asm volatile("" : "+v"(tid));
        const double* wb = block_of(A.wint, bwd, l);
        const int n = bwd ? LB_SIZE : LF_LDS;
#pragma unroll
        for (int k = 0; k < NWC; ++k) pfw[k] = (FT_KNOB & 8) ? 0.01 : ldu(wb, (unsigned)min(tid + k * NT, n - 1));
    }
    __device__ __forceinline__ void weights_commit() {            // the size of the larger block: the tail of a smaller one is never read
        int tid = this->tid;
        asm volatile("" : "+v"(tid));
        double* dst = sm + G::SW + (wcur ^ 1) * LF_LDS;
#pragma unroll
        for (int k = 0; k < NWC; ++k) if (tid + k * NT < LF_LDS) dst[tid + k * NT] = pfw[k];
    }

    // active site a (compact index of struct Stash) -> lattice site
    static __device__ __forceinline__ void active_site(int a, int mu, int off, int& i, int& j) {
        if (mu == 0) { i = fdiv<L / 4>(a); j = off + 4 * (a - i * (L / 4)); }
        else { const int m = fdiv<L>(a); j = a - m * L; i = off + 4 * m; }
    }
    // frozen site f in [0, N / 2): line q along the stripes, h-th frozen line across them
    static __device__ __forceinline__ void frozen_site(int f, int mu, int off, int& r, int& c) {
        const int h = fdiv<L>(f), q = f - h * L, x = 4 * (h >> 1) + ((off + 1 + (h & 1)) & 3);
        if (mu == 0) { r = q; c = x; } else { c = q; r = x; }
    }
    // Pairs ACROSS the stripe lines (conv1 forward, conv2^T): tile `wave` = pair position u across the lines, lane i = position
    // v along them; site 0 = (v, 2 u) for mu = 0 (site 1 = next column), (2 u, v) for mu = 1 (site 1 = next row).  A tile
    // holds one u, so the stripe classes of its four-line input window are wave-uniform: conv1 runs on the two frozen lines
    // only, conv2^T skips the dead one (flow_fwd.hip, flow_bwd_gather.hip do the same on their windows).
    __device__ __forceinline__ bool pair_site(int lane, int mu, int& r, int& c) const {
        const int u = wave < L / 2 ? wave : L / 2 - 1, v_ = lane & 15, v = v_ < L ? v_ : L - 1;
        if (mu == 0) { r = v; c = 2 * u; } else { r = 2 * u; c = v; }
        return wave < L / 2 && v_ < L;
    }

    // ---- one coupling layer forward, in place on the links in LDS.  STASH: also write what the backward needs.
    //      log J of the layer is returned in thread 0 (want_logj), 0 elsewhere.  (nb, nl_): the pass that follows, or nl_ < 0.
    //      MU = the layer's stripe direction (l & 1) as a compile-time constant: the selects on it fold and each instance
    //      carries one of the two conv2 code paths (as in the tiled kernels).
    __device__ __forceinline__ double layer_fwd(int l, const bool STASH_, bool want_logj, bool nb, int nl_) {
        return (l & 1) ? layer_fwd_t<1>(l, STASH_, want_logj, nb, nl_) : layer_fwd_t<0>(l, STASH_, want_logj, nb, nl_);
    }
    template <int MU> __device__ double layer_fwd_t(int l, const bool STASH_, bool want_logj, bool nb, int nl_) {
        const bool STASH = (FT_KNOB & 1) ? false : STASH_;
        constexpr int N = G::N, NA = G::NA, NAS = G::NAS, RS = G::RS, PSZ = G::PSZ;
        // thread coordinates opaque per pass: otherwise every index expression of every stage is hoisted out of the layer
        // loops and lives in registers for the whole kernel (hundreds of them: they spilled)
        int tid = this->tid, lane = this->lane;
        asm volatile("" : "+v"(tid), "+v"(lane));
        constexpr int mu = MU;
        const int off = (l >> 1) & 3, act = A.act;
        double* sX = sm + G::X;  double* sIn = sm + G::IN;  double* sH1 = sm + G::A8;  double* sH2 = sm + G::B8;
        double* sPA = sm + G::PA;  double* sST = sm + G::ST;
        const double* sWc = sW();
        const Stash sv = STASH ? stash_view(stash(l), A.B, chain(), N) : Stash{};
        if (nl_ >= 0) weights_issue(nb, nl_);
        // ---- plaquettes, net input (cos P, sin P on the frozen lines, (1, 0) elsewhere), P / 2 at the active sites
        if (tid < N) {
            const int i = fdiv<L>(tid), j = tid - i * L;
            const int ip = i + 1 == L ? 0 : i + 1, jp = j + 1 == L ? 0 : j + 1;
            const double p = sX[tid] - sX[N + tid] - sX[i * L + jp] + sX[N + ip * L + j];
            const int sel = ((mu == 0 ? j : i) - off) & 3;
            const bool frozen = sel == 1 || sel == 2;
            double sn = 0.0, cs = 1.0;
            if (frozen || sel == 0) ft_sincos(frozen ? p : 0.5 * p, &sn, &cs);
            put1<L, RS>(sIn, i, j, frozen ? cs : 1.0);
            put1<L, RS>(sIn + PSZ, i, j, frozen ? sn : 0.0);
            if (sel == 0) {
                const int a = stash_active_idx(i, j, L, mu);
                sPA[a] = p; sPA[NAS + a] = cs; sPA[2 * NAS + a] = sn;
            }
            if (STASH && frozen) {
                double* cs_ = sv.cs + stash_frozen_idx(i, j, L, mu, off);
                gst(cs_, cs); gst(cs_ + (N >> 1), sn);
            }
        }
        lds_barrier();
        stamp(0);

        double* const st_d1 = STASH ? sv.d1 + 2 * (lane >> 4) : nullptr;
        double* const st_d2 = STASH ? sv.d2 + 2 * (lane >> 4) : nullptr;
        const bool HST = TRAIN && STASH;                                   // training: the weight gradients need h1, h2
        double* const st_h1 = HST ? sv.h1 + 2 * (lane >> 4) : nullptr;
        double* const st_h2 = HST ? sv.h2 + 2 * (lane >> 4) : nullptr;
        // ---- conv1 (2 -> 8) + act on the frozen taps: pairs across the stripe lines (columns for mu = 0, rows for mu = 1)
        {
            // K = 2 frozen lines x 3 taps along x 2 channels = 3 MFMA steps; the constant lines (cos, sin) = (1, 0) are in the
            // bias table BC (flow_common.h), indexed by the stripe class s4 of the window's first line
            const int g = lane >> 4, i = lane & 15, cN = i & 7, dd = i >> 3;
            int pr, pc;
            const bool ok = pair_site(lane, mu, pr, pc);
            const int u = mu == 0 ? pc >> 1 : pr >> 1, v = mu == 0 ? pr : pc;
            const int s4 = (2 * u - 1 - off) & 3;                        // wave-uniform
            const int fl = ((g >> 1) + 1 - s4) & 3;                      // this lane group's frozen line of the window
            const int lstep = mu == 0 ? 1 : RS, astep = mu == 0 ? RS : 1;
            const double* a0 = sIn + (g & 1) * PSZ + (2 * u + fl) * lstep + v * astep;
            const double* sP1 = sWc + LF_P1;
            const double* wp = sP1 + cN + (g & 1) * 48 + (fl + 1 - dd) * 8;
            const double* bc = sP1 + LF_BC + s4 * 16 + 2 * g;            // bias + constant lines: the accumulator's start value
            double4_t acc = {bc[0], bc[1], bc[8], bc[9]};
#pragma unroll
            for (int t = 0; t < 3; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wp[t * 96], a0[t * astep], acc, 0, 0, 0);
            double z[4] = {acc[0], acc[1], acc[2], acc[3]};
            double h[4], d[4];
            if (FT_KNOB & 4) { for (int q = 0; q < 4; ++q) { h[q] = z[q]; d[q] = 1.0; } } else
            act_eval4(z, act, h, d);
            if (ok) {
                const int dr = mu == 0 ? 0 : 1, dc = mu == 0 ? 1 : 0;
                put2<L, RS, PSZ>(sH1 + 2 * g * PSZ, pr, pc, h[0], h[1]);
                put2<L, RS, PSZ>(sH1 + 2 * g * PSZ, pr + dr, pc + dc, h[2], h[3]);
                if (STASH) {
                    const int at = pr * L + pc;
                    gst2(st_d1 + 8 * (size_t)at, double2_t{d[0], d[1]});
                    gst2(st_d1 + 8 * (size_t)(at + dr * L + dc), double2_t{d[2], d[3]});
                    if (HST) {
                        gst2(st_h1 + 8 * (size_t)at, double2_t{h[0], h[1]});
                        gst2(st_h1 + 8 * (size_t)(at + dr * L + dc), double2_t{h[2], h[3]});
                    }
                }
            }
        }
        lds_barrier();
        stamp(1);

        // ---- conv2 (8 -> 8) + act: pairs = rows for mu = 0, columns for mu = 1 (table P2)
        auto conv2_epi = [&](int g, bool ok, int r, int c, int dr, int dc, double (&z)[4]) {
            double h[4], d[4];                                           // the bias came in through the accumulator (bias2)
            if (FT_KNOB & 4) { for (int q = 0; q < 4; ++q) { h[q] = z[q]; d[q] = 1.0; } } else
            act_eval4(z, act, h, d);
            if (!ok) return;
            put2i<RS, PSZ>(sH2 + 2 * g * PSZ, r, c, h[0], h[1]);
            put2i<RS, PSZ>(sH2 + 2 * g * PSZ, r + dr, c + dc, h[2], h[3]);
            if (STASH) {
                const int at = r * L + c;
                gst2(st_d2 + 8 * (size_t)at, double2_t{d[0], d[1]});
                gst2(st_d2 + 8 * (size_t)(at + dr * L + dc), double2_t{d[2], d[3]});
                if (HST) {
                    gst2(st_h2 + 8 * (size_t)at, double2_t{h[0], h[1]});
                    gst2(st_h2 + 8 * (size_t)(at + dr * L + dc), double2_t{h[2], h[3]});
                }
            }
        };
        double4_t bias2;
        { const double b0 = sWc[LF_B1 + 2 * (lane >> 4)], b1 = sWc[LF_B1 + 2 * (lane >> 4) + 1]; bias2 = double4_t{b0, b1, b0, b1}; }
        if (mu == 0) {
            mfma_stage<KConv2Row, G::NPAIR, RS, PSZ, false, false, 1>(sH1, sWc + LF_P2, wave, lane,
                [&](int p) { const int pr = fdiv<L>(p); return 2 * pr * RS + (p - pr * L); },
                [&](int g, int p, bool ok, double (&z)[4], int) { const int pr = fdiv<L>(p); conv2_epi(g, ok, 2 * pr, p - pr * L, 1, 0, z); },
                nullptr, bias2);
        } else {
            mfma_stage<KConv2Col, G::NPAIR, RS, PSZ, false, false, 1>(sH1, sWc + LF_P2, wave, lane,
                [&](int p) { const int r = fdiv<L / 2>(p); return r * RS + 2 * (p - r * (L / 2)); },
                [&](int g, int p, bool ok, double (&z)[4], int) { const int r = fdiv<L / 2>(p); conv2_epi(g, ok, r, 2 * (p - r * (L / 2)), 0, 1, z); },
                nullptr, bias2);
        }
        // conv3's 27 weights of this wave's input channel: scalar loads straight from the weight block
        typedef const double __attribute__((address_space(4))) * cdptr;
        double w3[27];
        {
            cdptr w3p = (cdptr)(size_t)(block_of(A.wint, false, l) + LF_W2 + wave * 9);
            // wide scalar loads written out (s_load_dwordx16 + s_load_dwordx2 per output channel, as in flow_fwd.hip: the pass that
            // would merge 27 s_load_dwordx2 is off for this kernel)
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
        stamp(2);

        // ---- conv3 (8 -> 3) at the NA active sites; one input channel per wave
        const bool alane = lane < NA;
        int ai = 0, aj = 0;
        active_site(alane ? lane : 0, mu, off, ai, aj);
        if (alane) {
            double acc[3] = {0.0, 0.0, 0.0};
            int ro[3], co_[3];
            wrap3<L>(ai, ro); wrap3<L>(aj, co_);
            double hv9[9];                                               // the nine taps first, then the 27 FMAs (see the transform below)
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) hv9[tp] = sH2[wave * PSZ + ro[tp / 3] * RS + co_[tp % 3]];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[k] = fma(hv9[tp], w3[k * 9 + tp], acc[k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) sST[(wave * 3 + k) * NAS + lane] = acc[k];
        }
        lds_barrier();
        stamp(3);

        // ---- tan-mixture transform: wave k evaluates mixture component k, wave NMIX the shift t (flow_fwd.hip, same
        //      arithmetic); then ONE wave: the new plaquette, log J and the link update of the lane's own active link
        double* sT2 = sm + G::T2;
        if (wave < NMIX && alane) {
            // the eight partials first, then their sum in channel order: one chain's latency IS this kernel's time, and left to
            // itself the scheduler (short of registers at L = 16) waits for every read before it issues the next
            double sq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) sq[q] = sST[(q * 3 + wave) * NAS + lane];
            const double cs = sPA[NAS + lane], sn = sPA[2 * NAS + lane];
            double sk = sWc[LF_B2 + wave];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) sk += sq[q];
            const double es = ft_exp(sk), ems = ft_rcp(es);
            const double cs2 = cs * cs, sn2 = sn * sn, sincs = sn * cs;
            const double invD = ft_rcp(ems * cs2 + es * sn2);
            sT2[(wave * 2 + 1) * NAS + lane] = invD;
            sT2[(wave * 2 + 0) * NAS + lane] = ft_wrap_pm_pi(2 * ft_atan(es * (sn / cs)));
            if (STASH) {
                const double sinP = 2.0 * sincs, invD2 = invD * invD;
                double* tc = sv.tc + (size_t)wave * N + 4 * (size_t)lane;
                gst2(tc, double2_t{sinP * invD / NMIX, (ems * cs2 - es * sn2) * invD2});
                gst2(tc + 2, double2_t{invD / NMIX, sinP * 0.5 * (es - ems) * invD2});
            }
        }
        if (wave == NMIX && alane) {
            double tq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) tq[q] = sST[(q * 3 + NMIX) * NAS + lane];
            double tv = sWc[LF_B2 + NMIX];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) tv += tq[q];
            sT2[2 * NMIX * NAS + lane] = tv;
        }
        if (nl_ >= 0) weights_commit();
        lds_barrier();
        stamp(4);
        double logj = 0.0;
        if (wave == 0) {
            double lj = 0.0;
            if (alane) {
                double ysum = 0.0, si = 0.0;
#pragma unroll
                for (int k = 0; k < NMIX; ++k) { ysum += sT2[(k * 2) * NAS + lane]; si += sT2[(k * 2 + 1) * NAS + lane]; }
                const double d = ft_wrap(ysum / NMIX + sT2[2 * NMIX * NAS + lane]) - sPA[lane];
                const int at = ai * L + aj;
                if (mu == 0) sX[at] = ft_wrap(d + sX[at]); else sX[N + at] = ft_wrap(-d + sX[N + at]);
                if (want_logj) lj = log(si) - log((double)NMIX);
            }
            if (want_logj) {
                const double tot = ft_wave_sum(lj);
                if (lane == 0) logj = tot;
            }
        }
        lds_barrier();
        if (nl_ >= 0) wcur ^= 1;
        stamp(5);
        return logj;
    }

    // the stash values of backward pass l (struct Stash), each in the thread that multiplies by it; four groups, so that a
    // pass can refill each group for the pass behind it as soon as it has consumed it (no second copy in registers)
    __device__ __forceinline__ void issue_tc(int tid, int l, BwdPre& q) const {
        constexpr int N = G::N, NA = G::NA;
        const Stash sv = stash_view(stash(l), A.B, chain(), N);
        const int a = tid < NA ? tid : 0;                                  // transform adjoint at active site tid
#pragma unroll
        for (int k = 0; k < 4 * NMIX; k += 2) {
            const double2_t t2 = ldg2(sv.tc + (size_t)(k >> 2) * N + 4 * a + (k & 3));
            q.tcv[k] = t2.x; q.tcv[k + 1] = t2.y;
        }
    }
    __device__ __forceinline__ void issue_cs(int tid, int l, BwdPre& q) const {
        constexpr int N = G::N, NF = G::NF;
        const int mu = l & 1, off = (l >> 1) & 3;
        const Stash sv = stash_view(stash(l), A.B, chain(), N);
        int fr, fc;
        frozen_site(tid < NF ? tid : 0, mu, off, fr, fc);                  // cos / sin of frozen site tid
        const int ic = stash_frozen_idx(fr, fc, L, mu, off);
        if (FT_KNOB & 16) { q.fcs = 0.6; q.fsn = 0.8; } else { q.fcs = gld(sv.cs + ic); q.fsn = gld(sv.cs + (N >> 1) + ic); }
    }
    __device__ __forceinline__ void issue_d2(int tid, int l, BwdPre& q) const {
        constexpr int N = G::N;
        const Stash sv = stash_view(stash(l), A.B, chain(), N);
        const int c3half = tid >= N ? 1 : 0, c3s = tid - c3half * N;       // conv3^T task = (site, half of the channels)
        const double* pl = sv.d2 + 8 * (size_t)(tid < 2 * N ? c3s : 0) + 4 * c3half;
        const double2_t va = ldg2(pl), vb = ldg2(pl + 2);
        q.d2v[0] = va.x; q.d2v[1] = va.y; q.d2v[2] = vb.x; q.d2v[3] = vb.y;
    }
    __device__ __forceinline__ void issue_d1(int lane, int l, BwdPre& q) const {
        constexpr int N = G::N;
        const int mu = l & 1;
        const Stash sv = stash_view(stash(l), A.B, chain(), N);
        int r, c;
        pair_site(lane, mu, r, c);                                         // conv2^T epilogue: channels 2 g, 2 g + 1 at both sites
        const int s0 = r * L + c, s1 = s0 + (mu == 0 ? 1 : L);
        const double2_t va = ldg2(sv.d1 + 8 * (size_t)s0 + 2 * (lane >> 4)), vb = ldg2(sv.d1 + 8 * (size_t)s1 + 2 * (lane >> 4));
        q.d1v[0] = va.x; q.d1v[1] = va.y; q.d1v[2] = vb.x; q.d1v[3] = vb.y;
    }
    __device__ __forceinline__ void bwd_issue(int l, BwdPre& q) const {
        int t = tid, ln = lane;
        asm volatile("" : "+v"(t), "+v"(ln));
        issue_tc(t, l, q); issue_cs(t, l, q); issue_d2(t, l, q); issue_d1(ln, l, q);
    }

    // ---- one coupling layer backward from the stash (gather form, flow_bwd_gather.hip): gP += this layer's contribution.
    //      `pre` holds this pass's stash values (bwd_issue); (nb, nl_): the pass that follows, or nl_ < 0.
    __device__ __forceinline__ void layer_bwd(int l, double cb, bool nb, int nl_) {
        if (l & 1) layer_bwd_t<1>(l, cb, nb, nl_); else layer_bwd_t<0>(l, cb, nb, nl_);
    }
    template <int MU> __device__ void layer_bwd_t(int l, double cb, bool nb, int nl_) {
        constexpr int N = G::N, NA = G::NA, NF = G::NF, RS = G::RS, PSZ = G::PSZ;
        constexpr int mu = MU;
        const int off = (l >> 1) & 3;
        int tid = this->tid, lane = this->lane;                            // opaque per pass (see layer_fwd)
        asm volatile("" : "+v"(tid), "+v"(lane));
        double* sGP = sm + G::GP;  double* sGO = sm + G::IN;  double* sGZ2 = sm + G::A8;
        double* sD1 = sm + G::B8;  double* sPart = sm + G::A8;
        const double* sWc = sW();
        double* const gzo = TRAIN ? gz(l) : nullptr;                       // training sweep: gz2 [N][8] | gz1 [N][8] | g_out [N / 4][4]
        // the next pass's weight block is requested first; its stash values (when it is a backward pass) refill `pre`
        // group by group behind this pass's last use of each
        const bool refill = nl_ >= 0 && nb;
        if (nl_ >= 0) weights_issue(nb, nl_);
        const bool ttask = tid < NA, ftask = tid < NF, c3task = tid < 2 * N;
        const int c3half = tid >= N ? 1 : 0, c3s = tid - c3half * N;
        int fr = 0, fc = 0;
        frozen_site(ftask ? tid : 0, mu, off, fr, fc);
        int pr_ = 0, pc_ = 0;
        const bool pok = pair_site(lane, mu, pr_, pc_);
        __builtin_amdgcn_sched_barrier(0);

        // ---- adjoint of the tan-mixture transform at the active sites; their own gP is complete here: nobody else reads
        //      the gradient of an active plaquette (a task reads its own and its passive neighbour's)
        if (ttask) {
            int i, j;
            active_site(tid, mu, off, i, j);
            const double g0 = sGP[i * L + j];
            const double g1 = mu == 0 ? sGP[i * L + (j == 0 ? L - 1 : j - 1)] : sGP[(i == 0 ? L - 1 : i - 1) * L + j];
            const double gdelta = g0 - g1;
            double csum = 0.0, esum = 0.0;
#pragma unroll
            for (int k = 0; k < NMIX; ++k) { csum += pre.tcv[4 * k + 2]; esum += pre.tcv[4 * k + 3]; }
            const double tsum = NMIX * csum;
            double rs = __builtin_amdgcn_rcp(tsum);
            rs = fma(fma(-tsum, rs, 1.0), rs, rs);
            rs = fma(fma(-tsum, rs, 1.0), rs, rs);
            const double cbr = cb * rs;
#pragma unroll
            for (int k = 0; k < NMIX; ++k) put1i<RS>(sGO + k * PSZ, i, j, gdelta * pre.tcv[4 * k] + cbr * pre.tcv[4 * k + 1]);
            put1i<RS>(sGO + NMIX * PSZ, i, j, gdelta);
            sGP[i * L + j] = g0 + (gdelta * (csum - 1.0) - cbr * esum);
            if (gzo) {                                                     // g_out record of active site tid: dL/ds_0, dL/ds_1, dL/dt, 0
                static_assert(NMIX == 2, "g_out record");
                double* po = gzo + 16 * (size_t)N + 4 * (size_t)tid;
                gst2(po, double2_t{gdelta * pre.tcv[0] + cbr * pre.tcv[1], gdelta * pre.tcv[4] + cbr * pre.tcv[5]});
                gst2(po + 2, double2_t{gdelta, 0.0});
            }
        }
        if (refill) issue_tc(tid, nl_, pre);
        lds_barrier();
        stamp(8);

        // ---- conv3^T on the VALU (the one tap line that holds an active site), times act'(z2) -> gz2
        if (c3task) {
            const int r = fdiv<L>(c3s), c = c3s - r * L;
            const int ksel = ((mu == 0 ? c : r) + 1 - off) & 3;           // the one kx (mu = 0) / ky (mu = 1) whose source is active
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            if (ksel <= 2) {
                int ro[3], co_[3];                                       // padded index of line r - 1 + k: tap ky reads ro[2 - ky]
                wrap3<L>(r, ro); wrap3<L>(c, co_);
#pragma unroll
                for (int co = 0; co < 3; ++co) {
                    double wq[3][4], g0[3];
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) {
                        const int ky = mu == 0 ? kk : ksel, kx = mu == 0 ? ksel : kk;
                        g0[kk] = sGO[co * PSZ + ro[2 - ky] * RS + co_[2 - kx]];
#pragma unroll
                        for (int k = 0; k < 4; ++k) wq[kk][k] = sWc[LB_W2 + (co * 8 + c3half * 4 + k) * 9 + ky * 3 + kx];
                    }
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] = fma(g0[kk], wq[kk][k], acc[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = ksel <= 2 ? pre.d2v[k] * acc[k] : 0.0;
            put2<L, RS, PSZ>(sGZ2 + (c3half * 4) * PSZ, r, c, acc[0], acc[1]);
            put2<L, RS, PSZ>(sGZ2 + (c3half * 4 + 2) * PSZ, r, c, acc[2], acc[3]);
            if (gzo) {
                double* po = gzo + 8 * (size_t)c3s + 4 * c3half;
                gst2(po, double2_t{acc[0], acc[1]}); gst2(po + 2, double2_t{acc[2], acc[3]});
            }
        }
        if (refill) issue_d2(tid, nl_, pre);
        lds_barrier();
        stamp(9);

        // ---- conv2^T (MFMA over the flipped, transposed table T2), times act'(z1) -> gz1
        {
            auto epi = [&](int g, int, bool ok, double (&z)[4], int) {
                if (!ok) return;
                const double v0 = z[0] * pre.d1v[0], v1 = z[1] * pre.d1v[1], v2 = z[2] * pre.d1v[2], v3 = z[3] * pre.d1v[3];
                put2i<RS, PSZ>(sD1 + 2 * g * PSZ, pr_, pc_, v0, v1);
                put2i<RS, PSZ>(sD1 + 2 * g * PSZ, pr_ + (mu == 0 ? 0 : 1), pc_ + (mu == 0 ? 1 : 0), v2, v3);
                if (gzo) {
                    const int s0 = pr_ * L + pc_, s1 = s0 + (mu == 0 ? 1 : L);
                    gst2(gzo + 8 * (size_t)N + 8 * (size_t)s0 + 2 * g, double2_t{v0, v1});
                    gst2(gzo + 8 * (size_t)N + 8 * (size_t)s1 + 2 * g, double2_t{v2, v3});
                }
            };
            // the pair's four-line window holds exactly one line on which gz2 is zero (class 2: no active site within reach):
            // its six K steps are skipped; which line it is depends on u only (wave-uniform)
            const int g = lane >> 4, i = lane & 15;
            const int u = wave < L / 2 ? wave : L / 2 - 1;
            const int kd = (off + 3 - 2 * u) & 3;
            const double* wp = sWc + LB_T2 + KConv2Row::wlane(g, i & 7, i >> 3);
            const double* a0 = sGZ2 + g * PSZ + pr_ * RS + pc_;             // padded origin of the pair window = site 0 - (1, 1)
            double4_t acc;
            if (mu == 0) {
                switch (kd) {
                    case 0: acc = conv2t_tile<KConv2Col, 4, 0, RS, PSZ>(wp, a0); break;
                    case 1: acc = conv2t_tile<KConv2Col, 4, 1, RS, PSZ>(wp, a0); break;
                    case 2: acc = conv2t_tile<KConv2Col, 4, 2, RS, PSZ>(wp, a0); break;
                    default: acc = conv2t_tile<KConv2Col, 4, 3, RS, PSZ>(wp, a0); break;
                }
            } else {
                switch (kd) {
                    case 0: acc = conv2t_tile<KConv2Row, 3, 0, RS, PSZ>(wp, a0); break;
                    case 1: acc = conv2t_tile<KConv2Row, 3, 1, RS, PSZ>(wp, a0); break;
                    case 2: acc = conv2t_tile<KConv2Row, 3, 2, RS, PSZ>(wp, a0); break;
                    default: acc = conv2t_tile<KConv2Row, 3, 3, RS, PSZ>(wp, a0); break;
                }
            }
            double z4[4] = {acc[0], acc[1], acc[2], acc[3]};
            epi(g, 0, pok, z4, 0);
        }
        typedef const double __attribute__((address_space(4))) * cdptr;
        double w0s[18];
        {
            cdptr wq = (cdptr)(size_t)(block_of(A.wint, true, l) + LB_W0 + wave * 18);
            typedef double double8c_t __attribute__((ext_vector_type(8)));                   // wide scalar loads (8 + 8 + 2), as above
            typedef const double8c_t __attribute__((address_space(4))) * cd8ptr;
            const double8c_t va = *(cd8ptr)(wq), vb = *(cd8ptr)(wq + 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) { w0s[k] = va[k]; w0s[8 + k] = vb[k]; }
            w0s[16] = wq[16]; w0s[17] = wq[17];
        }
        if (refill) issue_d1(lane, nl_, pre);
        lds_barrier();
        stamp(10);

        // ---- conv1^T at the frozen sites: wave = hidden channel, the sum over the channels through LDS
        for (int f = lane; f < NF; f += 64) {
            int r, c;
            frozen_site(f, mu, off, r, c);
            const double* gz = sD1 + wave * PSZ;                           // source site (r + 1 - ky, c + 1 - kx), wrapped
            int ro[3], co_[3];
            wrap3<L>(r, ro); wrap3<L>(c, co_);
            double gv[9], gc = 0.0, gs = 0.0;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) gv[tp] = gz[ro[2 - tp / 3] * RS + co_[2 - tp % 3]];
            __builtin_amdgcn_sched_barrier(0);                             // all nine reads in flight before the first FMA waits
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) { gc = fma(gv[tp], w0s[tp], gc); gs = fma(gv[tp], w0s[9 + tp], gs); }
            sPart[(wave * 2 + 0) * NF + f] = gc;
            sPart[(wave * 2 + 1) * NF + f] = gs;
        }
        lds_barrier();
        stamp(11);
        // the (cos, sin) adjoint lands in the gradient of the frozen plaquette itself: nothing else touches it in this pass
        if (ftask) {
            double pc[8], ps[8];                                        // all sixteen partials first (see the transform above)
#pragma unroll
            for (int co = 0; co < 8; ++co) { pc[co] = sPart[(co * 2 + 0) * NF + tid]; ps[co] = sPart[(co * 2 + 1) * NF + tid]; }
            const double gp0 = sGP[fr * L + fc];
            __builtin_amdgcn_sched_barrier(0);
            double gct = 0.0, gst = 0.0;
#pragma unroll
            for (int co = 0; co < 8; ++co) { gct += pc[co]; gst += ps[co]; }
            sGP[fr * L + fc] = gp0 + (-pre.fsn * gct + pre.fcs * gst);
        }
        if (refill) issue_cs(tid, nl_, pre);
        if (nl_ >= 0) weights_commit();
        lds_barrier();
        if (nl_ >= 0) wcur ^= 1;
        stamp(12);
    }

    // ---- Wilson pieces on the links in sX ------------------------------------------------------------------
    __device__ __forceinline__ double plaq_at(const double* sX, int s) const {
        constexpr int N = G::N;
        const int i = fdiv<L>(s), j = s - i * L;
        const int ip = i + 1 == L ? 0 : i + 1, jp = j + 1 == L ? 0 : j + 1;
        return sX[s] - sX[N + s] - sX[i * L + jp] + sX[N + ip * L + j];
    }
    // S_W = -beta sum cos P, Q = sum wrap(P) / 2 pi of the field in sX; results valid in every thread
    __device__ void action_charge(double beta, double& S, double& Q) {
        constexpr int N = G::N;
        const double* sX = sm + G::X;
        double* red = sm + G::RED;
        double c = 0.0, q = 0.0;
        if (tid < N) {
            const int i = fdiv<L>(tid), j = tid - i * L;
            const int ip = i + 1 == L ? 0 : i + 1, jp = j + 1 == L ? 0 : j + 1;
            const double a = sX[tid], bb = sX[N + tid], cc = sX[i * L + jp], d = sX[N + ip * L + j];
            double sn_, cs_;
            ft_sincos(a + d - cc - bb, &sn_, &cs_);                        // summation order of BatchAction._u1_plaq
            c = cs_;
            q = ft_wrap(a - bb - cc + d);                                  // summation order of batch_plaqs
        }
        c = ft_block_sum(c, red);
        q = ft_block_sum(q, red);
        S = (-beta) * c;
        Q = q / FT_TWO_PI;
    }
    // gP = beta sin P of the flowed field: seeds the backward sweep
    __device__ void wilson_seed(double beta) {
        constexpr int N = G::N;
        if (tid < N) {
            double sn_, cs_;
            ft_sincos(plaq_at(sm + G::X, tid), &sn_, &cs_);
            sm[G::GP + tid] = beta * sn_;
        }
        lds_barrier();
    }
    // force of site tid from sGP (adjoint of the plaquette stencil)
    __device__ __forceinline__ void site_force(double& f0, double& f1) const {
        const double* g = sm + G::GP;
        const int i = fdiv<L>(tid), j = tid - i * L;
        const int im = i == 0 ? L - 1 : i - 1, jm = j == 0 ? L - 1 : j - 1;
        const double gc = g[tid];
        f0 = gc - g[i * L + jm];
        f1 = g[im * L + j] - gc;
    }
};

enum { SM_ACTION = 0, SM_FORCE = 1, SM_LEAPFROG = 2, SM_TRAJ = 3, SM_TRAIN = 4 };

// One launch = a sequence of SWEEPS over the layers of one chain.  A sweep copies the latent links, runs every layer forward
// in place, and then either evaluates the Wilson action and charge of the flowed field (EVAL sweep: S_eff = S_W - log det J)
// or seeds the plaquette gradient with beta sin P and runs every layer backward (FORCE sweep: gP of S_eff).
//   action:      [EVAL]
//   force:       [FORCE] -> F
//   leapfrog:    x += dt/2 v; nstep x ([FORCE]; v -= dt F; x += dt v (dt/2 after the last))
//   trajectory:  [EVAL] (unless state_in carries it), the leapfrog, regularize, [EVAL], Metropolis
//   training:    ONE sweep that is both (fthmc/train.py:191-210): every layer forward with the stash (+ h1, h2) AND log J,
//                the Wilson action of the flowed field -> x, logq, logp; the seed (beta / B) sin P, every layer backward with
//                dL/dlogJ = -1 / B, writing each layer's pre-activation gradients for k_flow_wgrad
// The sweep loop has ONE call site of the layer bodies (the kernel is register- and code-size-bound otherwise).
// TRAJ: the launch is a whole trajectory (fthmc_ft_trajectory: the hot entry point of the small lattices): the other entry
// points' branches leave that instance.  DBG: the stage stamps of tools/small_profile.py (A.dbg) exist in their own instance
// of the generic kernel only: compiled out, config 2 runs 4 % faster (0.497 -> 0.477 ms per trajectory).
template <int L, bool TRAIN, bool TRAJ, bool DBG>
__global__ FT_LDS_B64 __launch_bounds__(NT, 2) void k_ft_small(SmallArgs Aarg) {
    using G = GS<L>;
    constexpr int N = G::N;
    __shared__ __attribute__((aligned(16))) double sm[G::SIZE];
    const int b = blockIdx.x;
    const Hot hot{(Aarg.wint), (Aarg.stash), DBG ? (Aarg.dbg) : nullptr, Aarg.B, Aarg.nl, Aarg.act, TRAIN ? (Aarg.gz) : nullptr};
    Chain<L, TRAIN> C(sm, hot, b);
    const int tid = C.tid;
    double* red = sm + G::RED;
    const int mode = TRAIN ? (int)SM_TRAIN : (TRAJ ? (int)SM_TRAJ : Aarg.mode), nl = hot.nl;
    const double beta = Aarg.beta, dt = Aarg.dt;
    const bool have_state = Aarg.state_in != nullptr;
    {
        const double* xb = cold().x + (size_t)b * 2 * N;
        for (int s = tid; s < 2 * N; s += NT) sm[G::XL + s] = xb[s];
    }
    if (hot.dbg && tid < 32) reinterpret_cast<long long*>(sm + G::PROF)[tid] = 0;
    const bool moves = mode == SM_LEAPFROG || mode == SM_TRAJ;
    constexpr bool train = TRAIN;
    const int nforce = mode == SM_ACTION ? 0 : ((mode == SM_FORCE || train) ? 1 : Aarg.nstep);
    // sweeps it = first .. last: it < 0 and it == nforce are EVAL sweeps, 0 <= it < nforce FORCE sweeps
    const int first = (mode == SM_ACTION || (mode == SM_TRAJ && !have_state)) ? -1 : 0;
    const int last = mode == SM_TRAJ ? nforce : (mode == SM_ACTION ? -1 : nforce - 1);
    // momenta of site tid in registers
    double v0 = 0.0, v1 = 0.0;
    if (moves && tid < N) { const double* vb = cold().v + (size_t)b * 2 * N; v0 = vb[tid]; v1 = vb[N + tid]; }
    double* stt = sm + G::STT;                                             // thread 0's scalars live in LDS, not in registers of every lane
    if (mode == SM_TRAJ) {
        if (have_state && tid == 0) { const double* si = cold().state_in; stt[0] = si[b]; stt[1] = si[hot.B + b]; stt[2] = si[2 * hot.B + b]; }
        const double k0 = ft_block_sum(v0 * v0 + v1 * v1, red);
        if (tid == 0) stt[6] = k0;
    }
    // the first pass's weight block
    C.weights_issue(false, 0);
    C.weights_commit();
    C.wcur ^= 1;
    lds_barrier();

    for (int it = first; it <= last; ++it) {
        const bool force = it >= 0 && it < nforce;
        if (it == 0 && moves && tid < N) { sm[G::XL + tid] += 0.5 * dt * v0; sm[G::XL + N + tid] += 0.5 * dt * v1; }   // own site
        if (it == nforce && mode == SM_TRAJ && tid < N) {                  // end of the MD: regularize (ipynb/ft_hmc.py:426)
            sm[G::XL + tid] = ft_regularize(sm[G::XL + tid]); sm[G::XL + N + tid] = ft_regularize(sm[G::XL + N + tid]);
        }
        if (tid < N) { sm[G::X + tid] = sm[G::XL + tid]; sm[G::X + N + tid] = sm[G::XL + N + tid]; }
        lds_barrier();
        C.stamp(16);
        double ld = 0.0;
        for (int l = 0; l < nl; ++l) {
            // the pass behind this one: the next layer, the first backward pass, or the first layer of the next sweep
            const bool nb = l + 1 == nl && force;
            const int nl_ = l + 1 < nl ? l + 1 : (force ? nl - 1 : (it < last ? 0 : -1));
            ld += C.layer_fwd(l, force, !force || train, nb, nl_);
        }
        if (force) {
            double bscale = beta, cb = -1.0;
            if (train) {
                // the loss pieces of this chain (train.py:191-206): x = F(xi), logq = -2 L^2 log(2 pi) - log det J, logp = -S_W(x)
                double S, Q;
                C.action_charge(beta, S, Q);
                ColdArgs& At = cold();
                if (tid == 0) {
                    if (At.logq) At.logq[b] = -(double)(2 * N) * log(FT_TWO_PI) - ld;
                    if (At.logp) At.logp[b] = -S;
                }
                if (At.x_out && tid < N) { At.x_out[(size_t)b * 2 * N + tid] = sm[G::X + tid]; At.x_out[(size_t)b * 2 * N + N + tid] = sm[G::X + N + tid]; }
                bscale = beta / (double)hot.B; cb = -1.0 / (double)hot.B;      // d mean_b (S_W - log det J) / d .
            }
            // the stash of this sweep was written by other threads of this workgroup: complete and visible before it is read
            __syncthreads();
            C.bwd_issue(nl - 1, C.pre);
            C.wilson_seed(bscale);
            C.stamp(17);
            for (int l = nl - 1; l >= 0; --l) C.layer_bwd(l, cb, l > 0, l > 0 ? l - 1 : (it < last ? 0 : -1));
            if (train) continue;
            if (tid < N) {
                double f0, f1;
                C.site_force(f0, f1);
                if (mode == SM_FORCE) { double* F = cold().F + (size_t)b * 2 * N; F[tid] = f0; F[N + tid] = f1; }
                else {
                    v0 -= dt * f0; v1 -= dt * f1;
                    const double a = it == nforce - 1 ? 0.5 * dt : dt;
                    sm[G::XL + tid] += a * v0; sm[G::XL + N + tid] += a * v1;
                }
            }
            C.stamp(18);
        } else {
            double S, Q;
            C.action_charge(beta, S, Q);
            if (tid == 0) {                                                // ld lives in thread 0
                double* st = stt + (it < 0 ? 0 : 3);
                st[0] = S - ld; st[1] = (-S) / (beta * (double)N); st[2] = Q; stt[7] = ld;
            }
            C.stamp(19);
        }
    }

    ColdArgs& A = cold();                                           // outputs: read from the kernarg segment here
    if (mode == SM_ACTION) {
        if (tid == 0) {
            if (A.S_eff) A.S_eff[b] = stt[0];
            if (A.logdet) A.logdet[b] = stt[7];
            if (A.plaq) A.plaq[b] = stt[1];
            if (A.Q) A.Q[b] = stt[2];
        }
        if (A.x_out && tid < N) { A.x_out[(size_t)b * 2 * N + tid] = sm[G::X + tid]; A.x_out[(size_t)b * 2 * N + N + tid] = sm[G::X + N + tid]; }
        return;
    }
    if (mode == SM_LEAPFROG) {
        if (tid < N) {
            A.x_out[(size_t)b * 2 * N + tid] = sm[G::XL + tid]; A.x_out[(size_t)b * 2 * N + N + tid] = sm[G::XL + N + tid];
            A.v_out[(size_t)b * 2 * N + tid] = v0; A.v_out[(size_t)b * 2 * N + N + tid] = v1;
        }
        return;
    }
    if (mode != SM_TRAJ) return;                                            // (training: everything was written inside the sweep)
    if (A.dbg && tid == 0) for (int k = 0; k < 32; ++k) A.dbg[(size_t)b * 32 + k] = reinterpret_cast<long long*>(sm + G::PROF)[k];
    // ---- Metropolis (ipynb/ft_hmc.py:427-435)
    const double k1 = ft_block_sum(v0 * v0 + v1 * v1, red);
    if (tid == 0) {
        const double h0 = stt[0] + 0.5 * stt[6], h1 = stt[3] + 0.5 * k1;
        const double d = h1 - h0;
        const bool ok = A.u[b] < exp(-d);
        if (A.dH) A.dH[b] = d;
        if (A.acc) A.acc[b] = ok ? 1.0 : 0.0;
        if (A.H0) A.H0[b] = h0;
        if (A.H1) A.H1[b] = h1;
        const double* sel = stt + (ok ? 3 : 0);
        if (A.state_out) { A.state_out[b] = sel[0]; A.state_out[A.B + b] = sel[1]; A.state_out[2 * A.B + b] = sel[2]; }
        if (A.plaq) A.plaq[b] = sel[1];
        if (A.Q) A.Q[b] = sel[2];
        red[0] = ok ? 1.0 : 0.0;
    }
    __syncthreads();
    const bool ok = red[0] > 0.5;
    if (tid < N) {
        const double* xb = A.x + (size_t)b * 2 * N;
        A.x_out[(size_t)b * 2 * N + tid] = ok ? sm[G::XL + tid] : xb[tid];
        A.x_out[(size_t)b * 2 * N + N + tid] = ok ? sm[G::XL + N + tid] : xb[N + tid];
    }
}

int g_small = -1;        // -1: not decided yet (FTHMC_SMALL_PATH=0 in the environment switches the path off: A/B runs)

}  // namespace

namespace fthmc {

void set_small_path(int v) { g_small = v; }
int get_small_path() {
    if (g_small < 0) { const char* e = getenv("FTHMC_SMALL_PATH"); g_small = e ? (atoi(e) != 0) : 1; }
    return g_small;
}
bool ft_small_shape(int L, int nl) { return nl >= 1 && (L == 8 || L == 12 || L == 16); }

int launch_ft_small(const SmallArgs& a, int L, hipStream_t s) {
    const dim3 grid(a.B), block(NT);
    if (!flow_stash_fits32(a.B, L, true)) return FTHMC_ERR_UNSUPPORTED;                  // stash_view: 32-bit plane offsets
    if (a.mode == SM_TRAIN) {
        if (!a.gz) return FTHMC_ERR_ARG;
        switch (L) {
            case 8: hipLaunchKernelGGL((k_ft_small<8, true, false, false>), grid, block, 0, s, a); break;
            case 12: hipLaunchKernelGGL((k_ft_small<12, true, false, false>), grid, block, 0, s, a); break;
            case 16: hipLaunchKernelGGL((k_ft_small<16, true, false, false>), grid, block, 0, s, a); break;
            default: return FTHMC_ERR_UNSUPPORTED;
        }
        FT_LAUNCH_CHECK(); return FTHMC_OK;
    }
    if (a.dbg) {                                            // diagnostic launches: the generic kernel with its stage stamps
        switch (L) {
            case 8: hipLaunchKernelGGL((k_ft_small<8, false, false, true>), grid, block, 0, s, a); break;
            case 12: hipLaunchKernelGGL((k_ft_small<12, false, false, true>), grid, block, 0, s, a); break;
            case 16: hipLaunchKernelGGL((k_ft_small<16, false, false, true>), grid, block, 0, s, a); break;
            default: return FTHMC_ERR_UNSUPPORTED;
        }
        FT_LAUNCH_CHECK(); return FTHMC_OK;
    }
    if (a.mode == SM_TRAJ) {
        switch (L) {
            case 8: hipLaunchKernelGGL((k_ft_small<8, false, true, false>), grid, block, 0, s, a); break;
            case 12: hipLaunchKernelGGL((k_ft_small<12, false, true, false>), grid, block, 0, s, a); break;
            case 16: hipLaunchKernelGGL((k_ft_small<16, false, true, false>), grid, block, 0, s, a); break;
            default: return FTHMC_ERR_UNSUPPORTED;
        }
        FT_LAUNCH_CHECK(); return FTHMC_OK;
    }
    switch (L) {
        case 8: hipLaunchKernelGGL((k_ft_small<8, false, false, false>), grid, block, 0, s, a); break;
        case 12: hipLaunchKernelGGL((k_ft_small<12, false, false, false>), grid, block, 0, s, a); break;
        case 16: hipLaunchKernelGGL((k_ft_small<16, false, false, false>), grid, block, 0, s, a); break;
        default: return FTHMC_ERR_UNSUPPORTED;
    }
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}

}  // namespace fthmc
