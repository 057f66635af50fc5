// extern "C" entry points of libfthmc_hip.so (include/fthmc_hip.h) and the host-side
// kernel sequencing behind them.  Every function only enqueues work on the caller's
// stream; scratch comes from the caller's workspace, so the whole sequence of a
// trajectory can be captured into a hipGraph by the caller.
#include "common.h"
#include "kernels.h"
#include <math.h>

#include <stdio.h>
#include <stdlib.h>

using namespace fthmc;

namespace fthmc {
static thread_local char g_last_error[256] = "";
void note_hip_error(hipError_t e, const char* file, int line) {
    snprintf(g_last_error, sizeof(g_last_error), "%s (%s:%d)", hipGetErrorString(e), file, line);
}
}  // namespace fthmc

namespace {

constexpr size_t ALIGN = 32;   // doubles (256 B)
inline size_t up(size_t n) { return (n + ALIGN - 1) / ALIGN * ALIGN; }

struct WS {
    double *wint, *X, *gp, *gp2, *gp_part, *lj_part, *scal, *act_part, *xa, *va, *xb, *vb, *gw_part, *gw_tmp, *stash, *gz;
    double *hbuf, *gbuf;   // generic net shapes (flow_generic.hip): activation scratch, gradient ping-pong
    size_t n2;       // doubles per field batch: B * 2 * L * L
    size_t gw_rows, gw_tmp_rows;   // rows of FLOW_GW_STRIDE doubles in gw_part / gw_tmp
    size_t total;    // doubles
};

// scal slots, each B doubles
enum { SC_S = 0, SC_Q, SC_PLAQ, SC_LOGDET, SC_K, SC_H0, SC_H1, SC_OLD0, SC_OLD1, SC_OLD2, SC_NEW0, SC_NEW1, SC_NEW2, SC_SEFF, SC_N };

// What a call runs with: the net's shape (an ARGUMENT of the call: fthmc_arch_t), the two process-wide debug switches read
// ONCE per call, the caller's canonical weights and stream.  Nothing below reads a global.
struct Ctx {
    FlowArch A;
    bool mfma;               // fthmc_set_variant: MFMA kernels (default) / VALU kernels
    bool small_on;           // fthmc_set_small_path
    const double* wcan;      // the caller's canonical weights (generic net shapes read them as they are)
    hipStream_t s;
    bool gen() const { return !A.is_default(); }
    bool small(int L, int nl) const { return small_on && mfma && A.is_default() && ft_small_shape(L, nl); }
};
inline int make_ctx(const fthmc_arch_t* arch, void* stream, Ctx* c) {
    c->A = flow_arch_default();
    if (arch) { int rc = make_flow_arch(arch->n_hidden, arch->hidden, arch->kernel_size, arch->n_mix, arch->final_tanh, &c->A); if (rc != FTHMC_OK) return rc; }
    c->mfma = get_flow_variant() == 1;
    c->small_on = get_small_path() != 0;
    c->wcan = nullptr;
    c->s = ft_stream(stream);
    return FTHMC_OK;
}

WS ws_layout(const FlowArch& A, double* base, int B, int L, int nl, bool train = false) {
    WS w{};
    const size_t n1 = (size_t)B * L * L, n2 = 2 * n1;
    const size_t nt = flow_ntiles_max(L);
    size_t o = 0;
    auto take = [&](size_t n) { double* p = base ? base + o : nullptr; o += up(n); return p; };
    w.n2 = n2;
    // the weight expansions: a FIXED head of FLOW_WHEAD_LAYERS layer regions in every layout (kernels.h: no call of any shape puts
    // another region there, so the stamps of k_pack_weights vouch for what sits behind them), longer for deeper flows
    w.wint = take((size_t)(nl > FLOW_WHEAD_LAYERS ? nl : FLOW_WHEAD_LAYERS) * FLOW_WINT);
    w.X = take((size_t)nl * n2);
    w.gp = take(n1);
    w.gp2 = take(nl > 0 ? n1 : 0);                       // second plaquette-gradient field (gather-form backward)
    w.gp_part = take(nl > 0 ? (size_t)B * flow_gp_part_max(L) : 0);
    w.lj_part = take((size_t)(nl > 0 ? nl : 1) * B * nt);               // logJ partials [layer][chain][tile] of a sweep
    w.scal = take((size_t)SC_N * B);
    w.act_part = take((size_t)32 * B);                   // launch_action_charge's wave sums (few chains of a large lattice)
    w.xa = take(n2); w.va = take(n2); w.xb = take(n2); w.vb = take(n2);
    // small lattices in training: the weight-gradient partials (and reduction rows) of ALL layers at once
    const size_t nlw = train && A.is_default() && ft_small_shape(L, nl) ? (size_t)nl : 1;
    w.gw_rows = nl > 0 ? nlw * B * nt : 0;
    w.gw_tmp_rows = nl > 0 ? nlw * FLOW_REDUCE_GROUPS : 0;
    if (train && nl > 0 && A.is_default() && flow_bwd_train_shape(L) && flow_stash_fits32(B, L, true)) {   // the shapes force_gp takes the fused path on
        // the fused training backward (flow_bwd_train.hip) leaves every layer's partials side by side: ONE reduction behind the sweep
        const size_t np = (size_t)flow_bwd_train_nparts(B, L), ng = (size_t)flow_reduce_groups((int)np);
        if (w.gw_rows < (size_t)nl * np) w.gw_rows = (size_t)nl * np;
        if (w.gw_tmp_rows < (size_t)nl * ng) w.gw_tmp_rows = (size_t)nl * ng;
    }
    w.gw_part = take(w.gw_rows * FLOW_GW_STRIDE);
    w.gw_tmp = take(w.gw_tmp_rows * FLOW_GW_STRIDE);
    const bool gen = !A.is_default();
    // activation stash of a force evaluation (generic shapes: every layer's planes, at least one region as scratch)
    w.stash = take(gen ? (size_t)(nl > 0 ? nl : 1) * A.stash_doubles(B, L) : (size_t)nl * flow_stash_doubles(B, L, train));
    // training: pre-activation gradients of the layer in flight; small lattices (flow_small.hip: one launch runs all layers
    // forward and backward): of every layer
    w.gz = take(train && nl > 0 && !gen ? flow_gz_doubles(B, L) * (ft_small_shape(L, nl) ? (size_t)nl : 1) : 0);
    w.hbuf = take(gen ? (size_t)B * A.cmax() * L * L : 0);
    w.gbuf = take(gen ? (size_t)2 * B * A.cmax() * L * L : 0);
    w.total = o;
    return w;
}

// which kernel family serves a call, and with it the tile geometry of its partial buffers
inline int flow_rev(const Ctx& C, const FlowLayerArgs& a, hipStream_t s) {
    return C.mfma ? launch_flow_rev_mfma(a, s) : launch_flow_rev(a, s);
}
inline int flow_fwd(const Ctx& C, const FlowLayerArgs& a, hipStream_t s) {
    return C.mfma ? launch_flow_fwd_mfma(a, s) : launch_flow_fwd(a, s);
}
inline size_t ws_doubles(const FlowArch& A, int B, int L, int nl, bool train = false) { return ws_layout(A, nullptr, B, L, nl, train).total; }

inline bool bad_shape(int B, int L) { return B <= 0 || L < 4 || (L % 4) != 0; }

// The caller's canonical weights: the tuned kernels read their own expansion (k_pack_weights -> W.wint), the kernels for
// other net shapes (flow_generic.hip) read the canonical layout itself.
// wver: the caller's statement of the weights' content version (the `_v` entry points; 0 = none: expand).  The library keeps
// nothing between calls: the launch compares the token of (version, address of w) with the stamps the last expansion left in
// the workspace and expands the layers whose stamps differ -- a stale or wrong version costs an expansion, never a result.
inline unsigned long long weights_token(const double* w, uint64_t wver) {
    if (wver == 0) return 0ull;
    unsigned long long z = (unsigned long long)wver ^ ((unsigned long long)(uintptr_t)w * 0xD6E8FEB86659FD93ull);
    z ^= z >> 32; z *= 0xD6E8FEB86659FD93ull; z ^= z >> 32;            // a 64-bit mix: versions 1, 2, 3 ... do not look alike
    return z | 2ull;                                                     // never 0
}
inline int use_weights(Ctx& C, const double* w, int nl, const WS& W, hipStream_t s, uint64_t wver = 0) {
    C.wcan = w;
    return C.A.is_default() ? launch_pack_weights(w, nl, W.wint, s, weights_token(w, wver)) : FTHMC_OK;
}
inline GenLayerArgs gen_args(const Ctx& C, const WS& w, int l, int B, int L, int act, bool own_region) {
    GenLayerArgs g{};
    g.arch = C.A;
    g.w = C.wcan + (size_t)l * C.A.params();
    g.stash = w.stash + (own_region ? (size_t)l * C.A.stash_doubles(B, L) : 0);
    g.hbuf = w.hbuf; g.gbuf = w.gbuf;
    g.B = B; g.L = L; g.mu = l % 2; g.off = (l / 2) % 4; g.act = act;
    return g;
}

// small lattices: the fused single-launch path (flow_small.hip)
inline SmallArgs small_args(const double* x, const WS& w, int nl, int B, int act, double beta, int mode) {
    SmallArgs a{};
    a.x = x; a.wint = w.wint; a.stash = w.stash; a.beta = beta; a.mode = mode; a.B = B; a.nl = nl; a.act = act;
    return a;
}

#define FT_TRY(expr) do { int rc_ = (expr); if (rc_ != FTHMC_OK) return rc_; } while (0)

// 1 (default): the training backward of the tiled-exactly shapes computes its weight gradients itself (flow_bwd_train.hip);
// 0: k_flow_bwd_gather writes the pre-activation gradients and k_flow_wgrad reads them back (rounds 2-5; A/B builds: EXTRA=-DFT_FUSED_WGRAD=0)
#ifndef FT_FUSED_WGRAD
#define FT_FUSED_WGRAD 1
#endif

// Forward sweep x -> X[0..nl-1] (X[l] = output of layer l).  logdet (device [B]) optional.
// parts_only: leave the log J partials of every layer in w.lj_part ([layer][chain][tile]) and skip the summing launch (the
// caller folds them into its own reduction: launch_traj_energy); tuned kernels only.
int sweep_forward(const Ctx& C, const double* x, const WS& w, int nl, int B, int L, int act, double* logdet,
                  hipStream_t s, bool stash = false, bool train = false, bool parts_only = false) {
    if (C.gen()) {                                // any other net shape: plain kernels, one stash region per layer
        for (int l = 0; l < nl; ++l) {
            GenLayerArgs g = gen_args(C, w, l, B, L, act, stash);
            g.x = l == 0 ? x : w.X + (size_t)(l - 1) * w.n2;
            g.y = w.X + (size_t)l * w.n2;
            g.logj = logdet; g.logj_accumulate = l > 0;
            FT_TRY(launch_gen_fwd(g, false, s));
        }
        return FTHMC_OK;
    }
    for (int l = 0; l < nl; ++l) {
        FlowLayerArgs a{};
        a.stash = stash ? w.stash + (size_t)l * flow_stash_doubles(B, L, train) : nullptr;
        a.stash_h = train ? 1 : 0;
        // the layers behind this one write their stash before the backward reads this one's: beyond FT_STASH_FAR_BYTES of it the
        // 256 MB Infinity Cache will have let go of this layer's (the forward then stores it past the caches: flow_fwd.hip)
        a.stash_far = stash && (size_t)(nl - 1 - l) * flow_stash_doubles(B, L, train) * sizeof(double) >= FT_STASH_FAR_BYTES ? 1 : 0;
        a.x = l == 0 ? x : w.X + (size_t)(l - 1) * w.n2;
        a.wint = w.wint + (size_t)l * FLOW_WINT;
        a.y = w.X + (size_t)l * w.n2;
        // logJ partials of all layers side by side, summed by ONE launch behind the sweep (layer by layer, in order)
        a.logj_part = (logdet || parts_only) ? w.lj_part + (size_t)l * B * flow_fwd_geom(C.mfma).ntiles(L) : nullptr;
        a.B = B; a.L = L; a.mu = l % 2; a.off = (l / 2) % 4; a.act = act;
        FT_TRY(flow_fwd(C, a, s));
    }
    if (logdet && nl > 0) FT_TRY(launch_sum_parts(w.lj_part, B, flow_fwd_geom(C.mfma).ntiles(L), 1.0, 0, logdet, s, nl));
    return FTHMC_OK;
}

inline const double* phys_field(const double* x, const WS& w, int nl) {
    return nl == 0 ? x : w.X + (size_t)(nl - 1) * w.n2;
}

// S_eff (and friends) of x; leaves the checkpoints in w.X
int eval_action(const Ctx& C, const double* x, const WS& w, int nl, int B, int L, int act, double beta,
                double* S_eff, double* logdet, double* plaq, double* Q, hipStream_t s) {
    double* ld = logdet ? logdet : w.scal + (size_t)SC_LOGDET * B;
    if (nl > 0) FT_TRY(sweep_forward(C, x, w, nl, B, L, act, ld, s));
    double* S = w.scal + (size_t)SC_S * B;
    FT_TRY(launch_action_charge(phys_field(x, w, nl), B, L, beta, S, Q, plaq, s, w.act_part));
    if (S_eff) FT_TRY(launch_lincomb(S, 1.0, nl > 0 ? ld : nullptr, -1.0, 0.0, S_eff, B, s));
    return FTHMC_OK;
}

// Plaquette-gradient field of sum_b S_eff (scaled): gp = scale*beta*sin P(F(x)) + sum_l gP_l,
// with dL/dlogJ = glogj.  gw != null also accumulates weight gradients (training).
int force_gp(const Ctx& C, const double* x, const WS& w, int nl, int B, int L, int act, double beta_scaled,
             double glogj, double* gw, hipStream_t s, bool have_forward = false) {
    if (C.gen()) {                                // any other net shape (flow_generic.hip)
        if (nl > 0 && !have_forward) FT_TRY(sweep_forward(C, x, w, nl, B, L, act, nullptr, s, true));
        double* gcur = (nl & 1) ? w.gp2 : w.gp;
        double* galt = gcur == w.gp ? w.gp2 : w.gp;
        FT_TRY(launch_wilson_gp(phys_field(x, w, nl), B, L, beta_scaled, gcur, s));
        for (int l = nl - 1; l >= 0; --l) {
            GenLayerArgs g = gen_args(C, w, l, B, L, act, true);
            g.up_gp = gcur; g.glogj_const = glogj; g.gp_out = galt;
            g.gw = gw ? gw + (size_t)l * C.A.params() : nullptr;
            FT_TRY(launch_gen_bwd(g, s));
            double* t_ = gcur; gcur = galt; galt = t_;
        }
        return FTHMC_OK;
    }
    // MFMA path without weight gradients: the forward sweep stashes act'(z1), act'(z2), s per site
    // and the backward kernels read them back instead of recomputing the network
    // (training: the caller ran the forward with the h planes stashed too, have_forward = true)
    const bool mfma = C.mfma;
    const bool stash = mfma && (gw == nullptr ? !have_forward : have_forward);
    const bool train = gw != nullptr && stash;
    if (nl > 0 && !have_forward) FT_TRY(sweep_forward(C, x, w, nl, B, L, act, nullptr, s, stash));
    // stash path: gather-form backward, gP ping-pongs between two fields and ends in w.gp
    double* gcur = (stash && (nl & 1)) ? w.gp2 : w.gp;
    double* galt = gcur == w.gp ? w.gp2 : w.gp;
    FT_TRY(launch_wilson_gp(phys_field(x, w, nl), B, L, beta_scaled, gcur, s));
#if FT_FUSED_WGRAD
    const bool fused = gw && train && flow_bwd_train_built() && flow_bwd_train_shape(L) && flow_stash_fits32(B, L, true);
    const int npf = fused ? (int)flow_bwd_train_nparts(B, L) : 0;
    // the layers' partials side by side and ONE reduction behind the sweep (two launches instead of two per layer), where the
    // workspace has the rows (a training layout: ws_layout)
    const bool one_reduction = fused && (size_t)nl * npf <= w.gw_rows && (size_t)nl * flow_reduce_groups(npf) <= w.gw_tmp_rows;
#endif
    for (int l = nl - 1; l >= 0; --l) {
        FlowLayerArgs a{};
        a.x = l == 0 ? x : w.X + (size_t)(l - 1) * w.n2;
        a.wint = w.wint + (size_t)l * FLOW_WINT;
        a.up_gp = gcur;
        a.glogj_const = glogj;
        a.gp_part = w.gp_part;
        a.gw_part = w.gw_part;
        a.B = B; a.L = L; a.mu = l % 2; a.off = (l / 2) % 4; a.act = act;
        if (stash) {
            a.stash = w.stash + (size_t)l * flow_stash_doubles(B, L, train);
            a.gp_out = galt;
#if FT_FUSED_WGRAD
            if (fused) {
                // training: the layer's backward and its weight gradients in ONE kernel (flow_bwd_train.hip: the pre-activation
                // gradients never leave LDS), one partial per workgroup
                if (one_reduction) a.gw_part = w.gw_part + (size_t)l * npf * FLOW_GW_STRIDE;
                FT_TRY(launch_flow_bwd_train(a, s));
                if (!one_reduction) FT_TRY(launch_reduce_gw(w.gw_part, npf, 1.0, 0, gw + (size_t)l * FTHMC_W_PER_LAYER, w.gw_tmp, s));
                double* t_ = gcur; gcur = galt; galt = t_;
                continue;
            }
#endif
            a.gz = train ? w.gz : nullptr;
            FT_TRY(launch_flow_bwd_gather(a, s));
            if (gw) {                                                 // weight gradients from the pre-activation gradients
                a.tpw = flow_wgrad_tpw(B, L, 1);
                FT_TRY(launch_flow_wgrad(a, s));
                FT_TRY(launch_reduce_gw(w.gw_part, flow_wgrad_nparts(B, L, a.tpw), 1.0, 0,
                                        gw + (size_t)l * FTHMC_W_PER_LAYER, w.gw_tmp, s));
            }
            double* t_ = gcur; gcur = galt; galt = t_;
            continue;
        }
        FT_TRY(launch_flow_bwd(a, gw != nullptr, s));                // VALU variant: scatter form + gather
        if (gw) FT_TRY(launch_reduce_gw(w.gw_part, B * flow_geom(false).ntiles(L), 1.0, 0,
                                        gw + (size_t)l * FTHMC_W_PER_LAYER, w.gw_tmp, s));
        FT_TRY(launch_gather_gp(w.gp_part, B, L, flow_geom(false), 1, gcur, s));
    }
#if FT_FUSED_WGRAD
    if (one_reduction && nl > 0) FT_TRY(launch_reduce_gw(w.gw_part, npf, 1.0, 0, gw, w.gw_tmp, s, nl, (size_t)npf * FLOW_GW_STRIDE));
#endif
    return FTHMC_OK;
}

// leapfrog in the latent field; result in w.xa / w.va
// xreg (optional): regularize(result x), written by the last kick (the end point of a trajectory, ipynb/ft_hmc.py:426)
int ft_leapfrog_ws(const Ctx& C, const double* x, const double* v, const WS& w, int nl, int B, int L, int act,
                   double beta, double dt, int nstep, hipStream_t s, double* xreg = nullptr) {
    FT_TRY(launch_axpy_copy(x, v, 0.5 * dt, w.xa, w.va, w.n2, s));       // first half drift + the working copy of the momenta
    for (int k = 0; k < nstep; ++k) {
        FT_TRY(force_gp(C, w.xa, w, nl, B, L, act, beta, -1.0, nullptr, s));
        FT_TRY(launch_kick_from_gp(w.gp, w.va, w.xa, nullptr, B, L, dt, k == nstep - 1 ? 0.5 * dt : dt, s,
                                   k == nstep - 1 ? xreg : nullptr));
    }
    return FTHMC_OK;
}

// plain leapfrog; result pointers returned through xo/po (ping-pong inside the workspace)
int leapfrog_ws(const double* x, const double* p, const WS& w, int B, int L, double beta, double dt,
                int nstep, double** xo, double** po, hipStream_t s) {
    const double* xi = x; const double* pi = p;
    double* xs[2] = {w.xa, w.xb}; double* ps[2] = {w.va, w.vb};
    int cur = 0;
    for (int k = 0; k < nstep; ++k) {
        FT_TRY(launch_leap_step(xi, pi, xs[cur], ps[cur], B, L, beta, k == 0 ? 0.5 * dt : dt, dt, s));
        xi = xs[cur]; pi = ps[cur]; cur ^= 1;
    }
    // final half drift into the free x buffer
    FT_TRY(launch_axpy(xi, pi, 0.5 * dt, xs[cur], w.n2, s));
    *xo = xs[cur]; *po = const_cast<double*>(pi);
    return FTHMC_OK;
}

}  // namespace

extern "C" {

#ifndef FTHMC_SRC_SHA
#define FTHMC_SRC_SHA "unknown"
#endif
// "... src <fingerprint>": the kernel sources this library was built from (tools/csrc_sha.py, csrc/Makefile)
#ifdef FT_DRYRUN      // the sanitizer build (make san): launches are no-ops -- the name says so, fthmc_amd/_lib.py refuses to load it as the product
#define FTHMC_BUILD_KIND " DRYRUN (host-side sanitizer build: launches nothing)"
#else
#define FTHMC_BUILD_KIND ""
#endif
const char* fthmc_version(void) { return "fthmc_hip 0.5 (gfx950) src " FTHMC_SRC_SHA FTHMC_BUILD_KIND; }

int fthmc_set_variant(int v) {
    if (v != 0 && v != 1) return FTHMC_ERR_ARG;
    if (const char* e = getenv("FTHMC_LEAP_ROWS")) set_leap_rows(atoi(e) != 0);     // measurement switch, read with the variant
    set_flow_variant(v);
    return FTHMC_OK;
}
int fthmc_get_variant(void) { return get_flow_variant(); }

int fthmc_arch_params(const fthmc_arch_t* arch) {
    Ctx C;
    if (make_ctx(arch, nullptr, &C) != FTHMC_OK) return FTHMC_ERR_UNSUPPORTED;
    return C.A.params();
}

int fthmc_set_small_path(int on) {
    if (on != 0 && on != 1) return FTHMC_ERR_ARG;
    set_small_path(on);
    return FTHMC_OK;
}
int fthmc_get_small_path(void) { return get_small_path(); }

const char* fthmc_last_error(void) { return fthmc::g_last_error; }

const char* fthmc_strerror(int code) {
    switch (code) {
        case FTHMC_OK: return "ok";
        case FTHMC_ERR_ARG: return "bad argument (null pointer or shape; L must be a multiple of 4)";
        case FTHMC_ERR_UNSUPPORTED: return "unsupported configuration";
        case FTHMC_ERR_LAUNCH: return "HIP launch failed";
        case FTHMC_ERR_WS: return "workspace too small (see fthmc_ws_bytes)";
        default: return "unknown error";
    }
}

size_t fthmc_ws_bytes(const fthmc_arch_t* arch, int B, int L, int n_layers) {
    Ctx C;
    if (B <= 0 || L <= 0 || n_layers < 0 || make_ctx(arch, nullptr, &C) != FTHMC_OK) return 0;
    return ws_doubles(C.A, B, L, n_layers) * sizeof(double);
}

size_t fthmc_train_ws_bytes(const fthmc_arch_t* arch, int B, int L, int n_layers) {
    Ctx C;
    if (B <= 0 || L <= 0 || n_layers < 0 || make_ctx(arch, nullptr, &C) != FTHMC_OK) return 0;
    return ws_doubles(C.A, B, L, n_layers, true) * sizeof(double);
}

// the call's context (shape from `arch`, the debug switches, the stream), then its workspace view
#define FT_CTX(arch_)                                                               \
    (void)hipGetLastError();   /* drop stale (non-sticky) errors left by the host framework */ \
    Ctx C; { const int rc_ = make_ctx((arch_), stream, &C); if (rc_ != FTHMC_OK) return rc_; } \
    hipStream_t s = C.s; (void)s
#define FT_WS(nl)                                                                   \
    if (!ws || ws_bytes < ws_doubles(C.A, B, L, (nl)) * sizeof(double)) return FTHMC_ERR_WS; \
    const WS W = ws_layout(C.A, static_cast<double*>(ws), B, L, (nl))

int fthmc_wrap(const double* x, double* out, size_t n, void* stream) {
    if (!x || !out) return FTHMC_ERR_ARG;
    return launch_wrap(x, out, n, 0, ft_stream(stream));
}
int fthmc_regularize(const double* x, double* out, size_t n, void* stream) {
    if (!x || !out) return FTHMC_ERR_ARG;
    return launch_wrap(x, out, n, 1, ft_stream(stream));
}
int fthmc_plaquettes(const double* x, double* P, int B, int L, void* stream) {
    if (!x || !P || bad_shape(B, L)) return FTHMC_ERR_ARG;
    return launch_plaq(x, P, B, L, ft_stream(stream));
}
int fthmc_wilson_action_charge(const double* x, int B, int L, double beta, double* S, double* Q,
                               double* plaq, void* stream) {
    if (!x || bad_shape(B, L)) return FTHMC_ERR_ARG;
    return launch_action_charge(x, B, L, beta, S, Q, plaq, ft_stream(stream));
}
int fthmc_wilson_force(const double* x, int B, int L, double beta, double* F, void* stream) {
    if (!x || !F || bad_shape(B, L)) return FTHMC_ERR_ARG;
    return launch_wilson_force(x, B, L, beta, F, ft_stream(stream));
}
int fthmc_kinetic(const double* v, int B, int L, double* K, void* stream) {
    if (!v || !K || bad_shape(B, L)) return FTHMC_ERR_ARG;
    return launch_kinetic(v, B, L, K, ft_stream(stream));
}

int fthmc_stats_accumulate(const double* acc, const double* plaq, const double* Q, double* qold, const double* dH, int B,
                           double* vec8, void* stream) {
    if (!acc || !plaq || !Q || !qold || !dH || !vec8 || B <= 0) return FTHMC_ERR_ARG;
    return launch_stats_accumulate(acc, plaq, Q, qold, dH, B, vec8, ft_stream(stream));
}

int fthmc_random_momenta(const int64_t* seeds, int B, int n_per_chain, double* v, double* u, void* stream) {
    if (!seeds || !v || B <= 0 || n_per_chain <= 0) return FTHMC_ERR_ARG;
    return launch_random_momenta(seeds, B, n_per_chain, v, u, ft_stream(stream));
}

size_t fthmc_ws_head_bytes(void) { return up((size_t)FLOW_WHEAD_LAYERS * FLOW_WINT) * sizeof(double); }

int fthmc_pack_weights(const double* w, const fthmc_arch_t* arch, int n_layers, uint64_t weights_version, void* ws, size_t ws_bytes,
                       void* stream) {
    if (!w || n_layers <= 0) return FTHMC_ERR_ARG;
    FT_CTX(arch);
    if (!ws || ws_bytes < up((size_t)(n_layers > FLOW_WHEAD_LAYERS ? n_layers : FLOW_WHEAD_LAYERS) * FLOW_WINT) * sizeof(double)) return FTHMC_ERR_WS;
    const WS W = ws_layout(C.A, static_cast<double*>(ws), 1, 4, n_layers);      // the expansion is the workspace's first region whatever B, L
    return use_weights(C, w, n_layers, W, s, weights_version);
}

int fthmc_chain_seeds(int64_t seed, int64_t lo, int B, int64_t traj, int64_t* counter, int advance, int64_t* seeds, void* stream) {
    if (!seeds || B <= 0 || (advance && !counter)) return FTHMC_ERR_ARG;
    return launch_chain_seeds(seed, lo, B, traj, counter, advance, seeds, ft_stream(stream));
}

int fthmc_random_uniform(const int64_t* seeds, int B, int n_per_chain, double lo, double hi, double* out, void* stream) {
    if (!seeds || !out || B <= 0 || n_per_chain <= 0 || !(hi > lo)) return FTHMC_ERR_ARG;
    return launch_random_uniform(seeds, B, n_per_chain, lo, hi, out, ft_stream(stream));
}

int fthmc_train_metrics(const double* xi, const double* x, const double* logq, const double* logp, int B, int L,
                        double beta, double dkl_factor, double* row, void* ws, size_t ws_bytes, void* stream) {
    if (!xi || !x || !logq || !logp || !row || bad_shape(B, L) || !(beta != 0.0)) return FTHMC_ERR_ARG;
    FT_CTX(nullptr);
    FT_WS(0);
    double* q = W.scal + (size_t)SC_Q * B; double* qi = W.scal + (size_t)SC_OLD0 * B;
    FT_TRY(launch_action_charge(x, B, L, beta, nullptr, q, nullptr, s, W.act_part));
    FT_TRY(launch_action_charge(xi, B, L, beta, nullptr, qi, nullptr, s, W.act_part));
    return launch_train_metrics(logq, logp, q, qi, B, 1.0 / (beta * L * L), dkl_factor, row, s);
}

int fthmc_adam_step(double* w, const double* gw, double* exp_avg, double* exp_avg_sq, double* hyper, size_t n,
                    double beta1, double beta2, double eps, double weight_decay, int decoupled, void* stream) {
    if (!w || !gw || !exp_avg || !exp_avg_sq || !hyper || n == 0) return FTHMC_ERR_ARG;
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0)) return FTHMC_ERR_ARG;
    return launch_adam(w, gw, exp_avg, exp_avg_sq, hyper, n, beta1, beta2, eps, weight_decay, decoupled != 0, ft_stream(stream));
}

int fthmc_leapfrog(const double* x, const double* p, int B, int L, double beta, double dt, int nstep,
                   double* x_out, double* p_out, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !p || !x_out || !p_out || bad_shape(B, L) || nstep < 1) return FTHMC_ERR_ARG;
    FT_CTX(nullptr);
    FT_WS(0);
    double *xo, *po;
    FT_TRY(leapfrog_ws(x, p, W, B, L, beta, dt, nstep, &xo, &po, s));
    if (hipMemcpyAsync(x_out, xo, W.n2 * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(p_out, po, W.n2 * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return FTHMC_ERR_LAUNCH;
    return FTHMC_OK;
}

int fthmc_hmc_trajectory(const double* x, const double* v, const double* u, int B, int L, double beta,
                         double dt, int nstep, double* x_new, double* dH, double* acc, double* H0,
                         double* H1, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !v || !u || !x_new || bad_shape(B, L) || nstep < 1) return FTHMC_ERR_ARG;
    FT_CTX(nullptr);
    // L <= 64 (x_new must not alias x): one persistent launch per trajectory, state in LDS / registers
    if (L <= 64 && C.mfma && x_new != x)
        return launch_hmc_trajectory_fused(x, v, u, B, L, beta, dt, nstep, x_new, dH, acc, H0, H1, s);
    FT_WS(0);
    double* S = W.scal + (size_t)SC_S * B; double* K = W.scal + (size_t)SC_K * B;
    double* h0 = H0 ? H0 : W.scal + (size_t)SC_H0 * B;
    double* h1 = H1 ? H1 : W.scal + (size_t)SC_H1 * B;
    FT_TRY(launch_action_charge(x, B, L, beta, S, nullptr, nullptr, s));
    FT_TRY(launch_kinetic(v, B, L, K, s));
    FT_TRY(launch_lincomb(S, 1.0, K, 0.5, 0.0, h0, B, s));
    double *xo, *po;
    FT_TRY(leapfrog_ws(x, v, W, B, L, beta, dt, nstep, &xo, &po, s));
    FT_TRY(launch_wrap(xo, xo, W.n2, 1, s));                       // xr = regularize(x_)
    FT_TRY(launch_action_charge(xo, B, L, beta, S, nullptr, nullptr, s));
    FT_TRY(launch_kinetic(po, B, L, K, s));
    FT_TRY(launch_lincomb(S, 1.0, K, 0.5, 0.0, h1, B, s));
    return launch_metropolis(x, xo, u, h0, h1, B, L, 0, x_new, dH, acc, nullptr, nullptr, nullptr, 0, s);
}

int fthmc_flow_layer_fwd(const double* x, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off, int act,
                         double* y, double* logJ, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !w || !y || bad_shape(B, L) || mu < 0 || mu > 1 || off < 0 || off > 3) return FTHMC_ERR_ARG;
    if (act < 0 || act > 2) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    FT_WS(1);
    FT_TRY(use_weights(C, w, 1, W, s));
    if (C.gen()) {
        GenLayerArgs g = gen_args(C, W, 0, B, L, act, false);
        g.mu = mu; g.off = off; g.x = x; g.y = y; g.logj = logJ;
        return launch_gen_fwd(g, false, s);
    }
    FlowLayerArgs a{};
    a.x = x; a.wint = W.wint; a.y = y; a.logj_part = W.lj_part;
    a.B = B; a.L = L; a.mu = mu; a.off = off; a.act = act;
    FT_TRY(flow_fwd(C, a, s));
    if (logJ) FT_TRY(launch_sum_parts(W.lj_part, B, flow_fwd_geom(C.mfma).ntiles(L), 1.0, 0, logJ, s));
    return FTHMC_OK;
}

int fthmc_flow_layer_rev(const double* y, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off, int act,
                         double tol, double* x, double* logJ, void* ws, size_t ws_bytes, void* stream) {
    if (!y || !w || !x || bad_shape(B, L) || mu < 0 || mu > 1 || off < 0 || off > 3) return FTHMC_ERR_ARG;
    if (act < 0 || act > 2) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    FT_WS(1);
    FT_TRY(use_weights(C, w, 1, W, s));
    if (C.gen()) {
        GenLayerArgs g = gen_args(C, W, 0, B, L, act, false);
        g.mu = mu; g.off = off; g.x = y; g.y = x; g.logj = logJ; g.tol = tol;
        return launch_gen_fwd(g, true, s);
    }
    FlowLayerArgs a{};
    a.x = y; a.wint = W.wint; a.y = x; a.logj_part = W.lj_part; a.tol = tol;
    a.B = B; a.L = L; a.mu = mu; a.off = off; a.act = act;
    FT_TRY(flow_rev(C, a, s));
    if (logJ) FT_TRY(launch_sum_parts(W.lj_part, B, flow_geom(false).ntiles(L), 1.0, 0, logJ, s));
    return FTHMC_OK;
}

// plaquette-level map: the coupling kernel with the plaquette field as input and output (MFMA kernels only)
static int plaq_coupling(const double* P, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off, int act, double tol,
                         bool rev, double* out, double* logJ, void* ws, size_t ws_bytes, void* stream) {
    if (!P || !w || !out || bad_shape(B, L) || mu < 0 || mu > 1 || off < 0 || off > 3) return FTHMC_ERR_ARG;
    if (act < 0 || act > 2) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    if (!C.mfma && C.A.is_default()) return FTHMC_ERR_UNSUPPORTED;
    FT_WS(1);
    FT_TRY(use_weights(C, w, 1, W, s));
    if (C.gen()) {
        GenLayerArgs g = gen_args(C, W, 0, B, L, act, false);
        g.mu = mu; g.off = off; g.pin = P; g.pout = out; g.logj = logJ; g.tol = tol;
        return launch_gen_fwd(g, rev, s);
    }
    FlowLayerArgs a{};
    a.x = P; a.pin = P; a.pout = out; a.wint = W.wint; a.logj_part = W.lj_part; a.tol = tol;
    a.B = B; a.L = L; a.mu = mu; a.off = off; a.act = act;
    FT_TRY(rev ? launch_flow_rev_mfma(a, s) : launch_flow_fwd_mfma(a, s));
    if (logJ) FT_TRY(launch_sum_parts(W.lj_part, B, flow_fwd_geom(true).ntiles(L), 1.0, 0, logJ, s));
    return FTHMC_OK;
}

int fthmc_plaq_coupling_fwd(const double* P, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off, int act,
                            double* fP, double* logJ, void* ws, size_t ws_bytes, void* stream) {
    return plaq_coupling(P, w, arch, B, L, mu, off, act, 0.0, false, fP, logJ, ws, ws_bytes, stream);
}

int fthmc_plaq_coupling_rev(const double* fP, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off, int act,
                            double tol, double* P, double* logJ, void* ws, size_t ws_bytes, void* stream) {
    return plaq_coupling(fP, w, arch, B, L, mu, off, act, tol, true, P, logJ, ws, ws_bytes, stream);
}

// VJP of the plaquette-level map fP = NCPPlaqCouplingLayer.forward(P) (layers.py:348-371): gP = d/dP [sum gfP fP + sum_b glogJ logJ],
// gw (optional) the same wrt the weights.  fP = P + delta at the active sites and P elsewhere, so
//     gP = gfP + (the link-level layer's plaquette gradient for the upstream link gradient gy[mu] = +-gfP),
// i.e. the link-level backward kernels serve it unchanged: the upstream gradient is dressed as that link field
// (launch_plane_from), the forward runs on the plaquette field itself (FlowLayerArgs::pin), gP = W.gp + gfP.
int fthmc_plaq_coupling_bwd(const double* P, const double* w, const fthmc_arch_t* arch, const double* gfP, const double* glogJ,
                            int B, int L, int mu, int off, int act, double* gP, double* gw, void* ws, size_t ws_bytes, void* stream) {
    if (!P || !w || !gfP || !glogJ || !gP || bad_shape(B, L) || mu < 0 || mu > 1 || off < 0 || off > 3) return FTHMC_ERR_ARG;
    if (act < 0 || act > 2) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    if (!C.mfma && C.A.is_default()) return FTHMC_ERR_UNSUPPORTED;          // as fthmc_plaq_coupling_fwd: MFMA kernels (or the plain ones)
    if (!ws || ws_bytes < ws_doubles(C.A, B, L, 1, gw != nullptr) * sizeof(double)) return FTHMC_ERR_WS;
    const WS W = ws_layout(C.A, static_cast<double*>(ws), B, L, 1, gw != nullptr);
    FT_TRY(use_weights(C, w, 1, W, s));
    double* fake = W.xa;                                                     // [B][2][L][L]: only plane mu is read
    FT_TRY(launch_plane_from(gfP, B, L, mu, mu == 0 ? 1.0 : -1.0, fake, s));
    const size_t n1 = (size_t)B * L * L;
    if (C.gen()) {
        GenLayerArgs g = gen_args(C, W, 0, B, L, act, false);
        g.mu = mu; g.off = off; g.pin = P;
        FT_TRY(launch_gen_fwd(g, false, s));
        g.up_link = fake; g.glogj = glogJ; g.gp_out = W.gp; g.gw = gw;
        FT_TRY(launch_gen_bwd(g, s));
        return launch_axpy(W.gp, gfP, 1.0, gP, n1, s);
    }
    FlowLayerArgs a{};
    a.x = P; a.pin = P; a.wint = W.wint; a.up_link = fake; a.glogj = glogJ;
    a.gw_part = W.gw_part; a.stash = W.stash; a.stash_h = gw ? 1 : 0;
    a.B = B; a.L = L; a.mu = mu; a.off = off; a.act = act;
    FT_TRY(launch_flow_fwd_mfma(a, s));                                      // fills the stash (no output field, no log J)
    a.gp_out = W.gp;
    a.gz = gw ? W.gz : nullptr;
    FT_TRY(launch_flow_bwd_gather(a, s));
    if (gw) {
        a.tpw = flow_wgrad_tpw(B, L, 1);
        FT_TRY(launch_flow_wgrad(a, s));
        FT_TRY(launch_reduce_gw(W.gw_part, flow_wgrad_nparts(B, L, a.tpw), 1.0, 0, gw, W.gw_tmp, s));
    }
    return launch_axpy(W.gp, gfP, 1.0, gP, n1, s);
}

// VJP of one layer.  `stash` != null: the forward's activation stash (fthmc_flow_layer_fwd_stash) -- nothing is recomputed;
// else the layer is run forward first from `x`.
static int layer_bwd_impl(const double* x, const double* stash, const double* w, const fthmc_arch_t* arch, const double* gy, const double* glogJ,
                          int B, int L, int mu, int off, int act, double* gx, double* gw, void* ws, size_t ws_bytes, void* stream) {
    if ((!x && !stash) || !w || !gy || !glogJ || !gx || bad_shape(B, L) || mu < 0 || mu > 1 || off < 0 || off > 3)
        return FTHMC_ERR_ARG;
    if (act < 0 || act > 2) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    const bool mfma = C.mfma;
    if (stash && !mfma && C.A.is_default()) return FTHMC_ERR_UNSUPPORTED;      // the VALU variant has no stash
    if (!ws || ws_bytes < ws_doubles(C.A, B, L, 1, gw != nullptr && mfma) * sizeof(double)) return FTHMC_ERR_WS;
    const WS W = ws_layout(C.A, static_cast<double*>(ws), B, L, 1, gw != nullptr && mfma);
    FT_TRY(use_weights(C, w, 1, W, s));
    if (C.gen()) {
        // the adjoint seeded by the link gradient; W.gp holds the layer's plaquette gradient alone
        GenLayerArgs g = gen_args(C, W, 0, B, L, act, false);
        g.mu = mu; g.off = off; g.x = x;
        if (stash) g.stash = const_cast<double*>(stash);
        else FT_TRY(launch_gen_fwd(g, false, s));                 // forward once: fills the layer's planes
        g.up_link = gy; g.glogj = glogJ; g.gp_out = W.gp; g.gw = gw;
        FT_TRY(launch_gen_bwd(g, s));
        return launch_adj_add(W.gp, gy, B, L, gx, s);
    }
    FlowLayerArgs a{};
    a.x = x; a.wint = W.wint; a.up_link = gy; a.glogj = glogJ;
    a.gp_part = W.gp_part; a.gw_part = W.gw_part;
    a.B = B; a.L = L; a.mu = mu; a.off = off; a.act = act;
    if (mfma) {
        // the gather-form backward seeded by the link gradient; gP_out holds the layer's plaquette gradient alone (no
        // upstream gP field).  Without a caller's stash: forward once with the stash (no link update, no log J).
        if (stash) { a.stash = const_cast<double*>(stash); a.stash_h = 1; }
        else {
            a.stash = W.stash; a.stash_h = gw ? 1 : 0;
            FT_TRY(launch_flow_fwd_mfma(a, s));
        }
        a.gp_out = W.gp;
        a.gz = gw ? W.gz : nullptr;
        FT_TRY(launch_flow_bwd_gather(a, s));
        if (gw) {
            a.tpw = flow_wgrad_tpw(B, L, 1);
            FT_TRY(launch_flow_wgrad(a, s));
            FT_TRY(launch_reduce_gw(W.gw_part, flow_wgrad_nparts(B, L, a.tpw), 1.0, 0, gw, W.gw_tmp, s));
        }
        return launch_adj_add(W.gp, gy, B, L, gx, s);
    }
    FT_TRY(launch_flow_bwd(a, gw != nullptr, s));
    if (gw) FT_TRY(launch_reduce_gw(W.gw_part, B * flow_geom(false).ntiles(L), 1.0, 0, gw, W.gw_tmp, s));
    FT_TRY(launch_gather_gp(W.gp_part, B, L, flow_geom(false), 0, W.gp, s));
    return launch_adj_add(W.gp, gy, B, L, gx, s);
}

int fthmc_flow_layer_bwd(const double* x, const double* w, const fthmc_arch_t* arch, const double* gy, const double* glogJ,
                         int B, int L, int mu, int off, int act, double* gx, double* gw, void* ws,
                         size_t ws_bytes, void* stream) {
    if (!x) return FTHMC_ERR_ARG;
    return layer_bwd_impl(x, nullptr, w, arch, gy, glogJ, B, L, mu, off, act, gx, gw, ws, ws_bytes, stream);
}

size_t fthmc_layer_stash_bytes(const fthmc_arch_t* arch, int B, int L) {
    Ctx C;
    if (B <= 0 || L <= 0 || make_ctx(arch, nullptr, &C) != FTHMC_OK) return 0;
    if (C.gen()) return C.A.stash_doubles(B, L) * sizeof(double);
    return C.mfma ? flow_stash_doubles(B, L, true) * sizeof(double) : 0;
}

int fthmc_flow_layer_fwd_stash(const double* x, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off, int act, double* y,
                               double* logJ, double* stash, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !w || !y || !stash || bad_shape(B, L) || mu < 0 || mu > 1 || off < 0 || off > 3) return FTHMC_ERR_ARG;
    if (act < 0 || act > 2 || fthmc_layer_stash_bytes(arch, B, L) == 0) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    FT_WS(1);
    FT_TRY(use_weights(C, w, 1, W, s));
    if (C.gen()) {
        GenLayerArgs g = gen_args(C, W, 0, B, L, act, false);
        g.mu = mu; g.off = off; g.x = x; g.y = y; g.logj = logJ; g.stash = stash;
        return launch_gen_fwd(g, false, s);
    }
    FlowLayerArgs a{};
    a.x = x; a.wint = W.wint; a.y = y; a.logj_part = W.lj_part; a.stash = stash; a.stash_h = 1;
    a.B = B; a.L = L; a.mu = mu; a.off = off; a.act = act;
    FT_TRY(launch_flow_fwd_mfma(a, s));
    if (logJ) FT_TRY(launch_sum_parts(W.lj_part, B, flow_fwd_geom(true).ntiles(L), 1.0, 0, logJ, s));
    return FTHMC_OK;
}

int fthmc_flow_layer_bwd_stash(const double* stash, const double* w, const fthmc_arch_t* arch, const double* gy, const double* glogJ, int B, int L,
                               int mu, int off, int act, double* gx, double* gw, void* ws, size_t ws_bytes, void* stream) {
    if (!stash) return FTHMC_ERR_ARG;
    return layer_bwd_impl(nullptr, stash, w, arch, gy, glogJ, B, L, mu, off, act, gx, gw, ws, ws_bytes, stream);
}

int fthmc_flow_forward_v(const double* x, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                       double* y, double* logdet, void* ws, size_t ws_bytes, void* stream, uint64_t weights_version) {
    if (!x || (n_layers > 0 && !w) || bad_shape(B, L) || n_layers < 0) return FTHMC_ERR_ARG;
    if (act < 0 || act > 2) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    FT_WS(n_layers);
    FT_TRY(use_weights(C, w, n_layers, W, s, weights_version));
    double* ld = logdet ? logdet : W.scal + (size_t)SC_LOGDET * B;
    if (n_layers == 0 && hipMemsetAsync(ld, 0, (size_t)B * sizeof(double), s) != hipSuccess) return FTHMC_ERR_LAUNCH;
    if (C.small(L, n_layers)) {
        SmallArgs a = small_args(x, W, n_layers, B, act, 1.0, 0);
        a.x_out = y; a.logdet = ld;
        return launch_ft_small(a, L, s);
    }
    FT_TRY(sweep_forward(C, x, W, n_layers, B, L, act, ld, s));
    if (y && hipMemcpyAsync(y, phys_field(x, W, n_layers), W.n2 * sizeof(double),
                            hipMemcpyDeviceToDevice, s) != hipSuccess) return FTHMC_ERR_LAUNCH;
    return FTHMC_OK;
}

int fthmc_flow_forward(const double* x, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                       double* y, double* logdet, void* ws, size_t ws_bytes, void* stream) {
    return fthmc_flow_forward_v(x, w, arch, n_layers, B, L, act, y, logdet, ws, ws_bytes, stream, 0);
}

int fthmc_flow_reverse_v(const double* y, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act, double tol,
                       double* x, double* logdet, void* ws, size_t ws_bytes, void* stream, uint64_t weights_version) {
    if (!y || !x || (n_layers > 0 && !w) || bad_shape(B, L) || n_layers < 0) return FTHMC_ERR_ARG;
    if (act < 0 || act > 2) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    FT_WS(n_layers);
    FT_TRY(use_weights(C, w, n_layers, W, s, weights_version));
    double* ld = logdet ? logdet : W.scal + (size_t)SC_LOGDET * B;
    if (hipMemsetAsync(ld, 0, (size_t)B * sizeof(double), s) != hipSuccess) return FTHMC_ERR_LAUNCH;
    // The last layer maps y -> x, the others run in place on x.  In place is safe: a layer only rewrites its ACTIVE links,
    // every plaquette the kernel USES (the frozen ones of its window, the active ones of its own tile) is built from
    // links the layer never writes plus the workgroup's own active links, which it loads before it stores them; the
    // active / passive plaquettes of the halo, which a neighbour's update can tear, are never read.
    const double* src = y;
    if (C.gen()) {
        for (int l = n_layers - 1; l >= 0; --l) {
            GenLayerArgs g = gen_args(C, W, l, B, L, act, false);
            g.x = src; g.y = x; g.logj = ld; g.logj_accumulate = 1; g.tol = tol;
            FT_TRY(launch_gen_fwd(g, true, s));
            src = x;
        }
        if (n_layers == 0 && x != y && hipMemcpyAsync(x, y, W.n2 * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess)
            return FTHMC_ERR_LAUNCH;
        return FTHMC_OK;
    }
    for (int l = n_layers - 1; l >= 0; --l) {
        FlowLayerArgs a{};
        a.x = src; a.wint = W.wint + (size_t)l * FLOW_WINT; a.y = x; a.logj_part = W.lj_part; a.tol = tol;
        a.B = B; a.L = L; a.mu = l % 2; a.off = (l / 2) % 4; a.act = act;
        FT_TRY(flow_rev(C, a, s));
        FT_TRY(launch_sum_parts(W.lj_part, B, flow_geom(false).ntiles(L), 1.0, 1, ld, s));
        src = x;
    }
    if (n_layers == 0 && x != y && hipMemcpyAsync(x, y, W.n2 * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return FTHMC_ERR_LAUNCH;
    return FTHMC_OK;
}

int fthmc_flow_reverse(const double* y, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act, double tol,
                       double* x, double* logdet, void* ws, size_t ws_bytes, void* stream) {
    return fthmc_flow_reverse_v(y, w, arch, n_layers, B, L, act, tol, x, logdet, ws, ws_bytes, stream, 0);
}

int fthmc_ft_action_v(const double* x, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act, double beta,
                    double* S_eff, double* logdet, double* plaq, double* Q, void* ws, size_t ws_bytes,
                    void* stream, uint64_t weights_version) {
    if (!x || (n_layers > 0 && !w) || bad_shape(B, L) || n_layers < 0) return FTHMC_ERR_ARG;
    if (act < 0 || act > 2) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    FT_WS(n_layers);
    FT_TRY(use_weights(C, w, n_layers, W, s, weights_version));
    if (n_layers == 0 && logdet && hipMemsetAsync(logdet, 0, (size_t)B * sizeof(double), s) != hipSuccess) return FTHMC_ERR_LAUNCH;
    if (C.small(L, n_layers)) {
        SmallArgs a = small_args(x, W, n_layers, B, act, beta, 0);
        a.S_eff = S_eff; a.logdet = logdet; a.plaq = plaq; a.Q = Q;
        return launch_ft_small(a, L, s);
    }
    return eval_action(C, x, W, n_layers, B, L, act, beta, S_eff, logdet, plaq, Q, s);
}

int fthmc_ft_action(const double* x, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act, double beta,
                    double* S_eff, double* logdet, double* plaq, double* Q, void* ws, size_t ws_bytes,
                    void* stream) {
    return fthmc_ft_action_v(x, w, arch, n_layers, B, L, act, beta, S_eff, logdet, plaq, Q, ws, ws_bytes, stream, 0);
}

int fthmc_ft_force_v(const double* x, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act, double beta,
                   double* F, void* ws, size_t ws_bytes, void* stream, uint64_t weights_version) {
    if (!x || !F || (n_layers > 0 && !w) || bad_shape(B, L) || n_layers < 0) return FTHMC_ERR_ARG;
    if (act < 0 || act > 2) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    FT_WS(n_layers);
    FT_TRY(use_weights(C, w, n_layers, W, s, weights_version));
    if (C.small(L, n_layers)) {
        SmallArgs a = small_args(x, W, n_layers, B, act, beta, 1);
        a.F = F;
        return launch_ft_small(a, L, s);
    }
    FT_TRY(force_gp(C, x, W, n_layers, B, L, act, beta, -1.0, nullptr, s));
    return launch_kick_from_gp(W.gp, nullptr, nullptr, F, B, L, 0.0, 0.0, s);
}

int fthmc_ft_force(const double* x, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act, double beta,
                   double* F, void* ws, size_t ws_bytes, void* stream) {
    return fthmc_ft_force_v(x, w, arch, n_layers, B, L, act, beta, F, ws, ws_bytes, stream, 0);
}

int fthmc_ft_leapfrog_v(const double* x, const double* v, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L,
                      int act, double beta, double dt, int nstep, double* x_out, double* v_out,
                      void* ws, size_t ws_bytes, void* stream, uint64_t weights_version) {
    if (!x || !v || !x_out || !v_out || (n_layers > 0 && !w) || bad_shape(B, L) || n_layers < 0 || nstep < 1)
        return FTHMC_ERR_ARG;
    if (act < 0 || act > 2) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    FT_WS(n_layers);
    FT_TRY(use_weights(C, w, n_layers, W, s, weights_version));
    if (C.small(L, n_layers)) {
        SmallArgs a = small_args(x, W, n_layers, B, act, beta, 2);
        a.v = v; a.dt = dt; a.nstep = nstep; a.x_out = x_out; a.v_out = v_out;
        return launch_ft_small(a, L, s);
    }
    FT_TRY(ft_leapfrog_ws(C, x, v, W, n_layers, B, L, act, beta, dt, nstep, s));
    if (hipMemcpyAsync(x_out, W.xa, W.n2 * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(v_out, W.va, W.n2 * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return FTHMC_ERR_LAUNCH;
    return FTHMC_OK;
}

int fthmc_ft_leapfrog(const double* x, const double* v, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L,
                      int act, double beta, double dt, int nstep, double* x_out, double* v_out,
                      void* ws, size_t ws_bytes, void* stream) {
    return fthmc_ft_leapfrog_v(x, v, w, arch, n_layers, B, L, act, beta, dt, nstep, x_out, v_out, ws, ws_bytes, stream, 0);
}

int fthmc_ft_trajectory_v(const double* x, const double* v, const double* u, const double* w, const fthmc_arch_t* arch, int n_layers,
                        int B, int L, int act, double beta, double dt, int nstep, int mode, double* x_new,
                        double* dH, double* acc, double* H0, double* H1, double* plaq, double* Q,
                        const double* state_in, double* state_out,
                        void* ws, size_t ws_bytes, void* stream, uint64_t weights_version) {
    if (!x || !v || !u || !x_new || (n_layers > 0 && !w) || bad_shape(B, L) || n_layers < 0 || nstep < 1)
        return FTHMC_ERR_ARG;
    if (act < 0 || act > 2 || (mode != FTHMC_MODE_MD && mode != FTHMC_MODE_LITERAL)) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    FT_WS(n_layers);
    double* K = W.scal + (size_t)SC_K * B;
    double* h0 = H0 ? H0 : W.scal + (size_t)SC_H0 * B;
    double* h1 = H1 ? H1 : W.scal + (size_t)SC_H1 * B;
    // per-chain state triples [S_eff, plaq, Q]: of x (old), of the proposal (neu), of x_new (sel)
    double* old = W.scal + (size_t)SC_OLD0 * B;      // slots SC_OLD0.. : 3 consecutive
    double* neu = W.scal + (size_t)SC_NEW0 * B;
    double* sel = state_out ? state_out : W.scal + (size_t)SC_S * B;
    FT_TRY(use_weights(C, w, n_layers, W, s, weights_version));
    if (mode == FTHMC_MODE_MD && C.small(L, n_layers)) {          // the whole trajectory in one launch
        SmallArgs a = small_args(x, W, n_layers, B, act, beta, 3);
        a.v = v; a.u = u; a.dt = dt; a.nstep = nstep; a.x_out = x_new; a.state_in = state_in; a.state_out = state_out;
        a.dH = dH; a.acc = acc; a.H0 = H0; a.H1 = H1; a.plaq = plaq; a.Q = Q;
        return launch_ft_small(a, L, s);
    }
    if (mode == FTHMC_MODE_MD && C.A.is_default() && n_layers > 0) {
        // tuned kernels: the scalars of either end of the trajectory come from ONE launch each (launch_traj_energy: log det J
        // from the sweep's partials, S_W / Q / plaq of the flowed field, the kinetic term, H), the first drift and the copy of
        // the momenta are one pass, the last kick also writes the regularized end point, the Metropolis kernel hands plaq / Q
        // out itself: 5 small launches per trajectory where there were 15
        const int np = flow_fwd_geom(C.mfma).ntiles(L);
        if (!state_in) FT_TRY(sweep_forward(C, x, W, n_layers, B, L, act, nullptr, s, false, false, true));
        FT_TRY(launch_traj_energy(phys_field(x, W, n_layers), B, L, beta, W.lj_part, np, n_layers, state_in, v, old, h0, s));
        FT_TRY(ft_leapfrog_ws(C, x, v, W, n_layers, B, L, act, beta, dt, nstep, s, W.xb));
        FT_TRY(sweep_forward(C, W.xb, W, n_layers, B, L, act, nullptr, s, false, false, true));
        FT_TRY(launch_traj_energy(phys_field(W.xb, W, n_layers), B, L, beta, W.lj_part, np, n_layers, nullptr, W.va, neu, h1, s));
        return launch_metropolis(x, W.xb, u, h0, h1, B, L, 0, x_new, dH, acc, old, neu, sel, 3, s, plaq, Q);
    }
    if (state_in) {       // chained trajectories: S_eff and observables of x are the previous call's state_out
        if (hipMemcpyAsync(old, state_in, (size_t)3 * B * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess)
            return FTHMC_ERR_LAUNCH;
    } else {
        FT_TRY(eval_action(C, x, W, n_layers, B, L, act, beta, old, nullptr, old + B, old + 2 * B, s));
    }
    FT_TRY(launch_kinetic(v, B, L, K, s));
    FT_TRY(launch_lincomb(old, 1.0, K, 0.5, 0.0, h0, B, s));
    const double* vend;
    if (mode == FTHMC_MODE_MD) {
        FT_TRY(ft_leapfrog_ws(C, x, v, W, n_layers, B, L, act, beta, dt, nstep, s));
        FT_TRY(launch_wrap(W.xa, W.xb, W.n2, 1, s));              // regularize (ipynb/ft_hmc.py:426)
        vend = W.va;
    } else {
        FT_TRY(launch_axpy(x, v, 0.5 * dt, W.xa, W.n2, s));       // ft_hmc.py:187 (Q2)
        FT_TRY(launch_wrap(W.xa, W.xb, W.n2, 0, s));              // wrap (ft_hmc.py:208)
        vend = v;
    }
    FT_TRY(eval_action(C, W.xb, W, n_layers, B, L, act, beta, neu, nullptr, neu + B, neu + 2 * B, s));
    FT_TRY(launch_kinetic(vend, B, L, K, s));
    FT_TRY(launch_lincomb(neu, 1.0, K, 0.5, 0.0, h1, B, s));
    // state of x_new without another sweep: select per chain
    FT_TRY(launch_metropolis(x, W.xb, u, h0, h1, B, L, 0, x_new, dH, acc, old, neu, sel, 3, s));
    if (plaq && hipMemcpyAsync(plaq, sel + B, (size_t)B * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return FTHMC_ERR_LAUNCH;
    if (Q && hipMemcpyAsync(Q, sel + 2 * B, (size_t)B * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return FTHMC_ERR_LAUNCH;
    return FTHMC_OK;
}

int fthmc_ft_trajectory(const double* x, const double* v, const double* u, const double* w, const fthmc_arch_t* arch, int n_layers,
                        int B, int L, int act, double beta, double dt, int nstep, int mode, double* x_new,
                        double* dH, double* acc, double* H0, double* H1, double* plaq, double* Q,
                        const double* state_in, double* state_out,
                        void* ws, size_t ws_bytes, void* stream) {
    return fthmc_ft_trajectory_v(x, v, u, w, arch, n_layers, B, L, act, beta, dt, nstep, mode, x_new, dH, acc, H0, H1, plaq, Q, state_in, state_out, ws, ws_bytes, stream, 0);
}

int fthmc_train_grad(const double* xi, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act, double beta,
                     double* x, double* logq, double* logp, double* gw, void* ws, size_t ws_bytes,
                     void* stream) {
    if (!xi || !w || bad_shape(B, L) || n_layers < 1) return FTHMC_ERR_ARG;
    if (act < 0 || act > 2) return FTHMC_ERR_UNSUPPORTED;
    FT_CTX(arch);
    if (!ws || ws_bytes < ws_doubles(C.A, B, L, n_layers, true) * sizeof(double)) return FTHMC_ERR_WS;
    const WS W = ws_layout(C.A, static_cast<double*>(ws), B, L, n_layers, true);
    FT_TRY(use_weights(C, w, n_layers, W, s));
    if (gw && C.small(L, n_layers)) {
        // small lattices: ONE launch runs the forward sweep (stash + h1, h2 + log J), the loss pieces and the backward sweep of
        // every chain (flow_small.hip, training sweep); then the weight gradients layer by layer from the gz it left behind
        SmallArgs a = small_args(xi, W, n_layers, B, act, beta, 4);
        a.x_out = x; a.logq = logq; a.logp = logp; a.gz = W.gz;
        FT_TRY(launch_ft_small(a, L, s));
        FlowLayerArgs f{};                                            // every layer's weight gradient in ONE launch + one reduction
        f.stash = W.stash; f.gz = W.gz; f.gw_part = W.gw_part;
        f.B = B; f.L = L; f.act = act;
        f.nlb = n_layers;
        f.stash_lstride = flow_stash_doubles(B, L, true); f.gz_lstride = flow_gz_doubles(B, L);
        f.tpw = flow_wgrad_tpw(B, L, n_layers);
        f.gwp_lstride = (size_t)flow_wgrad_nparts(B, L, f.tpw) * FLOW_GW_STRIDE;
        FT_TRY(launch_flow_wgrad(f, s));
        return launch_reduce_gw(W.gw_part, flow_wgrad_nparts(B, L, f.tpw), 1.0, 0, gw, W.gw_tmp, s, n_layers, f.gwp_lstride);
    }
    double* ld = W.scal + (size_t)SC_LOGDET * B;
    double* S = W.scal + (size_t)SC_S * B;
    // one forward sweep serves both the outputs (x, logq, logp) and the backward pass; with the MFMA
    // kernels it stashes act', s and h of every layer for the weight-gradient backward
    const bool mfma = C.mfma || C.gen();       // paths whose backward reads the forward's stash
    FT_TRY(sweep_forward(C, xi, W, n_layers, B, L, act, ld, s, mfma && gw != nullptr, mfma && gw != nullptr));
    if (x || logq || logp) {
        FT_TRY(launch_action_charge(phys_field(xi, W, n_layers), B, L, beta, S, nullptr, nullptr, s, W.act_part));
        const double lp0 = -(double)(2 * L * L) * log(FT_TWO_PI);
        if (logq) FT_TRY(launch_lincomb(ld, -1.0, nullptr, 0.0, lp0, logq, B, s));
        if (logp) FT_TRY(launch_lincomb(S, -1.0, nullptr, 0.0, 0.0, logp, B, s));
        if (x && hipMemcpyAsync(x, phys_field(xi, W, n_layers), W.n2 * sizeof(double),
                                hipMemcpyDeviceToDevice, s) != hipSuccess) return FTHMC_ERR_LAUNCH;
    }
    if (gw) {
        // d mean_b(S_W - logdet) / dw : seed beta/B on the Wilson term, -1/B on every logJ
        FT_TRY(force_gp(C, xi, W, n_layers, B, L, act, beta / B, -1.0 / B, gw, s, /*have_forward=*/true));
    }
    return FTHMC_OK;
}

int fthmc_time_kernel(int kind, const double* x, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off, int act,
                      double beta, int reps, double* ms_avg_host, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !ms_avg_host || bad_shape(B, L) || reps < 1 || kind < 0 || kind > 3) return FTHMC_ERR_ARG;
    if (kind < 2 && !w) return FTHMC_ERR_ARG;
    FT_CTX(arch);
    if (C.gen() && kind < 2) return FTHMC_ERR_UNSUPPORTED;         // the tuned kernels serve the default net shape
    FT_WS(1);
    FlowLayerArgs a{};
    if (kind < 2) {
        FT_TRY(use_weights(C, w, 1, W, s));
        FT_TRY(launch_wilson_gp(x, B, L, beta, W.gp, s));
        a.x = x; a.wint = W.wint; a.y = W.X; a.logj_part = W.lj_part;
        a.up_gp = W.gp; a.glogj_const = -1.0; a.gp_part = W.gp_part; a.gp_out = W.gp2;
        a.B = B; a.L = L; a.mu = mu; a.off = off; a.act = act;
#ifdef FT_DIAG
        if (const char* e = getenv("FTHMC_DBG_STOP")) a.dbg_stop = atoi(e);
#endif
        if (C.mfma) {            // the hot path: forward stashes, backward reads the stash
            a.stash = W.stash;
            a.logj_part = nullptr;       // ... and asks for no log J: exactly what a layer of a force sweep launches (the SWEEP / FS instances)
            FT_TRY(launch_flow_fwd_mfma(a, s));
        }
    } else {
        int64_t* seeds = reinterpret_cast<int64_t*>(W.scal);             // B int64 seeds = 0 .. (any values do)
        if (hipMemsetAsync(seeds, 0, (size_t)B * sizeof(int64_t), s) != hipSuccess) return FTHMC_ERR_LAUNCH;
        FT_TRY(launch_random_momenta(seeds, B, 2 * L * L, W.va, nullptr, s));
    }
    // the events are created after the last early return and destroyed on every path below
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return FTHMC_ERR_LAUNCH;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return FTHMC_ERR_LAUNCH; }
    int rc = FTHMC_OK;
    for (int it = -2; it < reps && rc == FTHMC_OK; ++it) {          // two untimed warm-up launches
        if (it == 0) (void)hipEventRecord(e0, s);
        if (kind == 0) rc = flow_fwd(C, a, s);
        else if (kind == 1) rc = a.stash ? launch_flow_bwd_gather(a, s) : launch_flow_bwd(a, false, s);
        else if (kind == 2) rc = launch_leap_step(x, W.va, W.xa, W.vb, B, L, beta, 0.05, 0.1, s);
        else rc = launch_hmc_trajectory_fused(x, W.va, W.scal + B, B, L, beta, 0.1, 10, W.xa, nullptr, nullptr, nullptr, nullptr, s);
    }
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_avg_host = (double)ms / reps;
    return rc;
}

int fthmc_time_small(const double* x, const double* v, const double* u, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                     double beta, double dt, int nstep, int reps, double* ms_avg_host, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !v || !u || !w || !ms_avg_host || bad_shape(B, L) || n_layers < 1 || nstep < 1 || reps < 1) return FTHMC_ERR_ARG;
    FT_CTX(arch);
    if (!C.small(L, n_layers)) return FTHMC_ERR_UNSUPPORTED;
    FT_WS(n_layers);
    FT_TRY(use_weights(C, w, n_layers, W, s));
    SmallArgs a = small_args(x, W, n_layers, B, act, beta, 3);
    a.v = v; a.u = u; a.dt = dt; a.nstep = nstep; a.x_out = W.xb;
    // H0 is evaluated in the launch (no state_in): nstep force sweeps + 2 action sweeps
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return FTHMC_ERR_LAUNCH;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return FTHMC_ERR_LAUNCH; }
    int rc = FTHMC_OK;
    for (int it = -2; it < reps && rc == FTHMC_OK; ++it) {          // two untimed warm-up launches
        if (it == 0) (void)hipEventRecord(e0, s);
        rc = launch_ft_small(a, L, s);
    }
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_avg_host = (double)ms / reps;
    return rc;
}

int fthmc_small_profile(const double* x, const double* v, const double* u, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L,
                        int act, double beta, double dt, int nstep, double* cycles_host32, void* ws, size_t ws_bytes,
                        void* stream) {
    if (!x || !v || !u || !w || !cycles_host32 || bad_shape(B, L) || n_layers < 1 || nstep < 1) return FTHMC_ERR_ARG;
    FT_CTX(arch);
    if (!C.small(L, n_layers)) return FTHMC_ERR_UNSUPPORTED;
    FT_WS(n_layers);
    long long* dbg = reinterpret_cast<long long*>(W.gw_part);            // unused by the force path; B * 32 stamps fit
    if (hipMemsetAsync(dbg, 0, (size_t)B * 32 * sizeof(long long), s) != hipSuccess) return FTHMC_ERR_LAUNCH;
    FT_TRY(use_weights(C, w, n_layers, W, s));
    SmallArgs a = small_args(x, W, n_layers, B, act, beta, 3);
    a.v = v; a.u = u; a.dt = dt; a.nstep = nstep; a.x_out = W.xb; a.dbg = dbg;
    FT_TRY(launch_ft_small(a, L, s));
    long long* h = (long long*)malloc((size_t)B * 32 * sizeof(long long));
    if (!h) return FTHMC_ERR_ARG;
    if (hipMemcpyAsync(h, dbg, (size_t)B * 32 * sizeof(long long), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) { free(h); return FTHMC_ERR_LAUNCH; }
    for (int k = 0; k < 32; ++k) {
        cycles_host32[k] = 0.0;
        for (int b = 0; b < B; ++b) cycles_host32[k] += (double)h[(size_t)b * 32 + k] / B;
    }
    free(h);
    return FTHMC_OK;
}

int fthmc_profile_stages(int kind, const double* x, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off, int act,
                         double beta, double* cycles_host16, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !w || !cycles_host16 || bad_shape(B, L) || kind < 0 || kind > 3) return FTHMC_ERR_ARG;
    FT_CTX(arch);
    if (C.gen()) return FTHMC_ERR_UNSUPPORTED;                     // the tuned kernels serve the default net shape
    const bool train = kind >= 2;                                  // kind 2: the backward in training mode (also writes A.gz); 3: k_flow_wgrad behind it
    if (!ws || ws_bytes < ws_doubles(C.A, B, L, 1, train) * sizeof(double)) return FTHMC_ERR_WS;
    const WS W = ws_layout(C.A, static_cast<double*>(ws), B, L, 1, train);
    const size_t nrec = (size_t)B * (kind >= 1 ? flow_gather_geom() : flow_fwd_geom(true)).ntiles(L);
    // stamp buffer: a workspace region the profiled launch does not write (B * ntiles * 16 stamps fit in either)
    long long* dbg = reinterpret_cast<long long*>(train ? W.gp_part : W.gw_part);
    if (hipMemsetAsync(dbg, 0, nrec * 16 * sizeof(long long), s) != hipSuccess) return FTHMC_ERR_LAUNCH;
    FT_TRY(use_weights(C, w, 1, W, s));
    FT_TRY(launch_wilson_gp(x, B, L, beta, W.gp, s));
    FlowLayerArgs a{};
    a.x = x; a.wint = W.wint; a.y = W.X; a.logj_part = train ? W.lj_part : nullptr;      // kind 0, 1: the launch of a force sweep
    a.up_gp = W.gp; a.glogj_const = -1.0; a.gp_part = W.gp_part; a.gp_out = W.gp2; a.dbg = dbg;
    a.B = B; a.L = L; a.mu = mu; a.off = off; a.act = act;
    if (kind == 0) a.stash = W.stash;             // the forward as a force sweep launches it: with the activation stash, without log J
    if (kind >= 1) {                              // stash backward needs the forward's stash first
        a.stash = W.stash; a.stash_h = train ? 1 : 0; a.gw_part = W.gw_part; a.gz = train ? W.gz : nullptr; a.dbg = nullptr;
        FT_TRY(launch_flow_fwd_mfma(a, s));
        a.dbg = kind == 3 ? nullptr : dbg;
    }
    FT_TRY(kind == 0 ? launch_flow_fwd_mfma(a, s) : launch_flow_bwd_gather(a, s));
    size_t nused = nrec;                                           // records the profiled launch stamps
    if (kind == 3) { a.dbg = dbg; a.tpw = flow_wgrad_tpw(B, L, 1); nused = (size_t)flow_wgrad_nparts(B, L, a.tpw); FT_TRY(launch_flow_wgrad(a, s)); }
    long long* h = (long long*)malloc(nrec * 16 * sizeof(long long));
    if (!h) return FTHMC_ERR_ARG;
    if (hipMemcpyAsync(h, dbg, nrec * 16 * sizeof(long long), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) { free(h); return FTHMC_ERR_LAUNCH; }
    for (int k = 0; k < 16; ++k) cycles_host16[k] = 0.0;
    for (size_t r = 0; r < nrec; ++r)
        for (int k = 1; k < 16; ++k) {
            const int ref = k == 15 ? 14 : kind == 3 ? k - 1 : (kind >= 1 && k >= 6) ? 0 : (kind == 0 && k == 7) ? 1 : (kind == 0 && k == 11) ? 2 : k - 1;   // forward 7..12 (-DFT_DIAG builds): inside conv1 / conv2      // backward, slots 6..13: per-wave arrival at the first barrier
            if (h[r * 16 + k] && h[r * 16 + ref]) cycles_host16[k] += (double)(h[r * 16 + k] - h[r * 16 + ref]) / nused;
        }
    free(h);
    return FTHMC_OK;
}

}  // extern "C"
