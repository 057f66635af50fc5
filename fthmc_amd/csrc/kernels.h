// Internal launch interface between the translation units of libfthmc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// FlowLayerArgs::stash_far (api.hip sweep_forward): a layer's stash counts as gone from the caches when the layers behind it write this
// much before the backward returns to it -- the 256 MB Infinity Cache over two chain groups, each writing and reading back
#ifndef FT_STASH_FAR_BYTES
#define FT_STASH_FAR_BYTES ((size_t)64 << 20)
#endif

namespace fthmc {

// ---- wilson.hip
int launch_wrap(const double* x, double* o, size_t n, int reg, hipStream_t s);
int launch_axpy(const double* x, const double* p, double a, double* o, size_t n, hipStream_t s);
int launch_plaq(const double* x, double* P, int B, int L, hipStream_t s);
// wave_part (optional): 32 B doubles of scratch; with it a few chains of a large lattice (B < 128, L >= 128) run one wave per
// workgroup instead of one workgroup per chain -- the same sums in the same order, bit-identical
int launch_action_charge(const double* x, int B, int L, double beta, double* S, double* Q,
                         double* plaq, hipStream_t s, double* wave_part = nullptr);
int launch_kinetic(const double* v, int B, int L, double* K, hipStream_t s);
int launch_lincomb(const double* a, double ca, const double* b, double cb, double c0, double* out,
                   int B, hipStream_t s);
int launch_stats_accumulate(const double* acc, const double* plaq, const double* Q, double* qold, const double* dH, int B,
                            double* vec, hipStream_t s);
int launch_wilson_force(const double* x, int B, int L, double beta, double* F, hipStream_t s);
void set_leap_rows(int v);      // 1 (default): row-strip leapfrog kernel when L % 64 == 0; 0: 16 x 16 tiles always
int launch_leap_step(const double* x, const double* p, double* xo, double* po, int B, int L,
                     double beta, double a, double dt, hipStream_t s);
int launch_wilson_gp(const double* x, int B, int L, double beta, double* gp, hipStream_t s);
// whole plain-HMC trajectory in one launch (L <= 64: links in LDS, momenta in registers)
int launch_hmc_trajectory_fused(const double* x, const double* v, const double* u, int B, int L, double beta,
                                double dt, int nstep, double* x_new, double* dH, double* acc, double* H0,
                                double* H1, hipStream_t s);
int launch_kick_from_gp(const double* gp, double* v, double* xq, double* Fout, int B, int L,
                        double dt, double a, hipStream_t s, double* xreg = nullptr);   // xreg: also regularize(x') (end of the MD)
// the scalars of one end of a flowed trajectory in one launch: (S_eff, plaq, Q) of the flowed field (or carried over: state_in)
// and H = S_eff + sum v^2 / 2
int launch_traj_energy(const double* xphys, int B, int L, double beta, const double* lj_part, int np, int nsets,
                       const double* state_in, const double* v, double* trip, double* H, hipStream_t s);
int launch_axpy_copy(const double* x, const double* p, double a, double* xo, double* po, size_t n, hipStream_t s);
int launch_plane_from(const double* g, int B, int L, int mu, double sign, double* out, hipStream_t s);   // out[b][mu][:] = sign * g[b][:]
int launch_metropolis(const double* x_old, const double* x_prop, const double* u, const double* H0,
                      const double* H1, int B, int L, int xform, double* x_new, double* dH,
                      double* acc, const double* obs_old, const double* obs_new, double* obs_out,
                      int n_obs, hipStream_t s, double* o1 = nullptr, double* o2 = nullptr);   // o1, o2: selected observables 1, 2

int launch_train_metrics(const double* logq, const double* logp, const double* q, const double* qi, int B,
                         double inv_beta_vol, double dkl_factor, double* row, hipStream_t s);
int launch_adam(double* p, const double* g, double* m, double* v, double* hp, size_t n, double b1, double b2, double eps,
                double wd, int decoupled, hipStream_t s);       // hp: [t, lr, ticket] on the device

// ---- rng.hip
int launch_random_momenta(const int64_t* seeds, int B, int n, double* v, double* u, hipStream_t s);
int launch_random_uniform(const int64_t* seeds, int B, int n, double lo, double hi, double* out, hipStream_t s);
int launch_chain_seeds(int64_t seed, int64_t lo, int B, int64_t traj, int64_t* counter, int advance, int64_t* seeds, hipStream_t s);

// ---- flow.hip
constexpr int FLOW_TILE = 16;                 // VALU variant (flow.hip): 16 x 16 sites per tile
constexpr int FLOW_R0 = FLOW_TILE + 6;        // plaquette / net-input window edge
constexpr int FLOW_N0 = FLOW_R0 * FLOW_R0;    // window size of one gP partial
constexpr int FLOW_WINT = 9856;               // doubles per layer, kernel-side weight layout
// The expansion of layer l is guarded by STAMPS inside the layer's own region (its last FLOW_WSTAMPS doubles, one per
// workgroup of k_pack_weights): a workgroup that finds its stamp equal to the token of the call (caller's weight version,
// address of the canonical weights, layer) leaves at once, otherwise it expands its slice and writes the token.  The
// first FLOW_WHEAD_LAYERS layer regions are a FIXED head of every workspace layout (ws_layout): no other region of any call,
// whatever its shape, overlaps them, so a stamp vouches for the data it sits behind; layers beyond the head are expanded
// by every call.
constexpr int FLOW_WSTAMPS = 8, FLOW_WSTAMP0 = FLOW_WINT - FLOW_WSTAMPS;
constexpr int FLOW_WHEAD_LAYERS = 64;
constexpr int FLOW_GW_STRIDE = 960;           // doubles per (chain, tile) weight-gradient partial

// tile geometry of a variant: partial buffers are indexed [chain][tile][window]
struct FlowGeom {
    int tr, tc;
    int nti(int L) const { return (L + tr - 1) / tr; }
    int ntj(int L) const { return (L + tc - 1) / tc; }
    int ntiles(int L) const { return nti(L) * ntj(L); }
    int n0() const { return (tr + 6) * (tc + 6); }
};
// tile of the MFMA forward kernel: 16 x 16 measured faster than 8 x 16 at 3 workgroups per CU (79 vs 94 us)
constexpr int MF_FWD_TR = 16, MF_FWD_TC = 16;
inline FlowGeom flow_fwd_geom(bool mfma) { return mfma ? FlowGeom{MF_FWD_TR, MF_FWD_TC} : FlowGeom{FLOW_TILE, FLOW_TILE}; }
inline FlowGeom flow_geom(bool) { return FlowGeom{FLOW_TILE, FLOW_TILE}; }          // VALU variant: partial buffers [chain][tile][window]
// workspace sizing: the larger of the two variants
inline size_t flow_ntiles_max(int L) {
    size_t a = flow_geom(false).ntiles(L), b = 2 * FlowGeom{16, 16}.ntiles(L);   // VALU / forward tiles; weight-gradient partials: two per 16 x 16 tile
    return a > b ? a : b;
}
inline size_t flow_gp_part_max(int L) { return (size_t)flow_geom(false).ntiles(L) * flow_geom(false).n0(); }

// canonical (955/layer, PyTorch order) -> kernel layout (FLOW_WINT/layer)
// token = 0: expand unconditionally (and clear the stamps); else see FLOW_WSTAMPS above
int launch_pack_weights(const double* w, int n_layers, double* wint, hipStream_t s, unsigned long long token = 0ull);

struct FlowLayerArgs {
    const double* x;         // [B][2][L][L] layer input
    const double* wint;      // this layer's weights, kernel layout
    double* y;               // fwd / rev: output field (may alias x)
    const double* pin;       // fwd / rev, plaquette-level map: input plaquette field [B][L][L] (then x is unused)
    double* pout;            // fwd / rev, plaquette-level map: output plaquette field [B][L][L]
    double* logj_part;       // fwd / rev: [B][ntiles]
    const double* up_link;   // bwd: upstream link gradient [B][2][L][L] or null
    const double* up_gp;     // bwd: upstream plaquette-gradient field [B][L][L] or null
    const double* glogj;     // bwd: [B] or null (then glogj_const)
    double glogj_const;
    double* gp_part;         // bwd, scatter form: [B][ntiles][FLOW_N0]
    double* gp_out;          // bwd, gather form: plaquette-gradient field after this layer [B][L][L] (not up_gp)
    double* gw_part;         // bwd with wgrad: [B*ntiles][FLOW_GW_STRIDE]
    double tol;              // rev
    long long* dbg;          // optional: per-(chain,tile) stage time stamps [16] (diagnostic runs only)
    double* stash;           // optional: this layer's activation stash (MFMA forward writes, stash backward reads)
    int stash_h;             // forward: also stash h1, h2 (training: the weight gradients need them)
    int stash_far;           // forward: this layer's stash will have left the caches when the backward comes for it (sweep_forward)
    double* gz;              // training: gradients wrt the pre-activations of this layer, written by the backward kernel at
                             // every tile's own sites and read by k_flow_wgrad: per chain gz2 [n][8], gz1 [n][8]
                             // (channel-minor), g_out [n/4][4] (dL/ds_0, dL/ds_1, dL/dt, 0 at the active sites, compact)
    int dbg_stop;            // -DFT_DIAG builds: the forward kernel returns after this stage (instruction counts per stage); 0 = run all
    int B, L, mu, off, act;
    // k_flow_wgrad over SEVERAL layers in one launch (small lattices: every layer's gz and stash are there at once): layers
    // 0 .. nlb - 1 (mu, off from the layer index), their stash / gz / partial regions `*_lstride` doubles apart; nlb = 0: one layer
    int nlb;
    size_t stash_lstride, gz_lstride, gwp_lstride;
    int tpw;                 // k_flow_wgrad: (chain, tile) items a workgroup walks (flow_wgrad_tpw; 0 = 1)
    int wg_ns;               // k_flow_wgrad: stride of the walk = workgroups that stand on consecutive tiles (set by the launcher)
};
int launch_flow_fwd(const FlowLayerArgs& a, hipStream_t s);
int launch_flow_rev(const FlowLayerArgs& a, hipStream_t s);
int launch_flow_bwd(const FlowLayerArgs& a, bool wgrad, hipStream_t s);
// flow_fwd.hip: MFMA forward (same arguments and results as launch_flow_fwd; optionally writes the stash)
int launch_flow_fwd_mfma(const FlowLayerArgs& a, hipStream_t s);
// the inverse layer on the same kernel (conv net unchanged, scalar map inverted by safeguarded Newton)
int launch_flow_rev_mfma(const FlowLayerArgs& a, hipStream_t s);
// flow_bwd_gather.hip: backward from the stash in gather form: a tile produces the complete
// gP_out = up_gp + layer contribution of its own sites (a.gp_out, out of place), no partial windows;
// with a.gz (training) it also writes the pre-activation gradients for launch_flow_wgrad
constexpr int MG_TR = 16, MG_TC = 16;
inline FlowGeom flow_gather_geom() { return FlowGeom{MG_TR, MG_TC}; }
int launch_flow_bwd_gather(const FlowLayerArgs& a, hipStream_t s);
// flow_wgrad.hip: weight gradients of one layer from a.gz and the stashed h1, h2, cos / sin: a workgroup walks a.tpw
// (chain, 16 x 16 tile) items and writes TWO 955-entry partials (halves of the tiles' sites) to
// a.gw_part [flow_wgrad_nparts(B, L, tpw)][FLOW_GW_STRIDE]
int launch_flow_wgrad(const FlowLayerArgs& a, hipStream_t s);
inline size_t flow_gz_doubles(int B, int L) { return (size_t)B * 17 * L * L; }
inline int flow_wgrad_parts(int L) { return 2 * FlowGeom{MG_TR, MG_TC}.ntiles(L); }          // per chain at one item per workgroup: sizes the workspace
// items per workgroup: as many as leave one workgroup for each of the 512 slots of the chip (two per CU), at most 8
inline int flow_wgrad_tpw(int B, int L, int nlayers) {
    const long items = (long)B * FlowGeom{MG_TR, MG_TC}.ntiles(L) * (nlayers > 0 ? nlayers : 1);
    const long t = items / 512;
    return t < 1 ? 1 : t > 8 ? 8 : (int)t;
}
// workgroups of one XCD that walk side by side (one per slot: 32 CUs x 2; fewer when the launch is smaller than the chip)
inline int flow_wgrad_ns(int B, int L, int tpw) {
    const int items = B * FlowGeom{MG_TR, MG_TC}.ntiles(L), n = (items + 8 * tpw - 1) / (8 * tpw);
    return n < 1 ? 1 : n > 64 ? 64 : n;
}
// partials of a launch: one per workgroup that has an item (k_flow_wgrad numbers those 0, 1, 2, ...)
inline int flow_wgrad_nparts(int B, int L, int tpw) {
    const int items = B * FlowGeom{MG_TR, MG_TC}.ntiles(L), ns = flow_wgrad_ns(B, L, tpw);
    const int k0 = items / (tpw * ns), rem = items - k0 * tpw * ns;
    return k0 * ns + (rem < ns ? rem : ns);
}
// flow_bwd_train.hip: the training backward WITH the layer's weight gradients in one kernel (the pre-activation gradients never
// leave LDS): one workgroup per CU walks (chain, tile) items and writes ONE 955-entry partial to a.gw_part
// [flow_bwd_train_nparts(B, L)][FLOW_GW_STRIDE]; a.gp_out as launch_flow_bwd_gather.  Built for the shapes 16 x 16 tiles divide with
// L a power of two (flow_bwd_train_shape); FTHMC_ERR_UNSUPPORTED otherwise: the caller keeps the two-kernel form.
int launch_flow_bwd_train(const FlowLayerArgs& a, hipStream_t s);
bool flow_bwd_train_built();          // false in the act'(z1)-recompute build (FT_RECOMP_D1: the fused kernel reads act'(z1) from the stash)
inline bool flow_bwd_train_shape(int L) { return L >= 32 && (L & (L - 1)) == 0; }
// items per workgroup: as many as leave one workgroup per CU (256), at most 64
inline int flow_bwd_train_tpw(int B, int L) {
    const long items = (long)B * FlowGeom{MG_TR, MG_TC}.ntiles(L);
    const long t = (items + 255) / 256;
    return t < 1 ? 1 : t > 64 ? 64 : (int)t;
}
// workgroups of one XCD that walk side by side (one per CU: 32; fewer when the launch is smaller than the chip)
inline int flow_bwd_train_ns(int B, int L, int tpw) {
    const long items = (long)B * FlowGeom{MG_TR, MG_TC}.ntiles(L), n = (items + 8 * tpw - 1) / (8 * tpw);
    return n < 1 ? 1 : n > 32 ? 32 : (int)n;
}
// partials of one launch (= its workgroups that walk at least one item); the launcher serves flow_stash_fits32 shapes only, where
// this fits an int with room to spare -- the arithmetic is 64-bit for the callers that ask before they check (ws_layout)
inline long flow_bwd_train_nparts(int B, int L) {
    const int tpw = flow_bwd_train_tpw(B, L), ns = flow_bwd_train_ns(B, L, tpw);
    const long items = (long)B * FlowGeom{MG_TR, MG_TC}.ntiles(L);
    const long k0 = items / ((long)tpw * ns), rem = items - k0 * tpw * ns;
    return k0 * ns + (rem < ns ? rem : ns);
}
// doubles per layer of the stash (layout: flow_mfma_common.h struct Stash): 19 per site, 35 with h1, h2 (training)
inline size_t flow_stash_doubles(int B, int L, bool train = false) { return (size_t)B * (train ? 35 : 19) * L * L; }
// the tuned kernels form a chain's plane offsets inside one layer's stash in 32 bits (flow_mfma_common.h: uniform_at, stash_view)
// -- one layer's stash below 32 GiB; beyond that the launchers return FTHMC_ERR_UNSUPPORTED
inline bool flow_stash_fits32(int B, int L, bool train) { return flow_stash_doubles(B, L, train) < ((size_t)1 << 32); }
// ranges the tuned kernels state as assumptions (__builtin_assume in k_flow_fwd / k_flow_bwd_gather): checked by their launchers
inline bool flow_shape_ok(int B, int L, int off) { return B > 0 && B <= (1 << 20) && L >= 4 && L <= 8192 && (L & 3) == 0 && off >= 0 && off < 4; }
// ---- flow_generic.hip: any s/t net shape (hidden sizes, kernel size, mixture components); plain kernels, HBM-resident planes
constexpr int FLOW_ARCH_MAXH = 8;
// Shape of the s/t conv net: 2 -> hid[0] -> ... -> hid[nh - 1] -> nmix + 1 channels, k x k kernels.  A VALUE that travels with
// every call (fthmc_arch_t of the C ABI; NULL there = the default): nothing about the shape is process state.
struct FlowArch {
    int nh; int hid[FLOW_ARCH_MAXH]; int k; int nmix;
    int tanh_out;            // a tanh behind the last conv (make_conv_net(use_final_tanh=True))
    bool is_default() const { return nh == 2 && hid[0] == 8 && hid[1] == 8 && k == 3 && nmix == 2 && !tanh_out; }   // the tuned kernels serve it
    int chan(int i) const { return i == 0 ? 2 : (i <= nh ? hid[i - 1] : nmix + 1); }                   // channels in front of conv i
    int params() const {                                    // doubles per layer in the canonical (PyTorch-order) weight layout
        int p = 0;
        for (int i = 0; i <= nh; ++i) p += chan(i + 1) * chan(i) * k * k + chan(i + 1);
        return p;
    }
    int cmax() const { int m = 2; for (int i = 1; i <= nh + 1; ++i) m = chan(i) > m ? chan(i) : m; return m; }   // widest activation
    int csum() const { int c = 0; for (int i = 1; i <= nh + 1; ++i) c += chan(i); return c; }
    // per layer: P [B][n], IN [B][2][n], Z_1 .. Z_{nh+1} [B][c_i][n]
    size_t stash_doubles(int B, int L) const { return (size_t)B * L * L * (3 + csum()); }
};
inline FlowArch flow_arch_default() { return FlowArch{2, {8, 8, 0, 0, 0, 0, 0, 0}, 3, 2, 0}; }
// validated copy of a caller's shape (FTHMC_ERR_UNSUPPORTED beyond the limits of flow_generic.hip)
int make_flow_arch(int n_hidden, const int* hidden_sizes, int kernel_size, int n_mix, int final_tanh, FlowArch* out);
struct GenLayerArgs {
    FlowArch arch;           // the net's shape
    const double* x;         // [B][2][L][L] layer input (null with pin)
    const double* pin;       // plaquette-level map: input plaquette field [B][L][L]
    const double* w;         // this layer's weights, canonical layout
    double* y;               // forward / reverse: output links (may alias x), or null
    double* pout;            // plaquette-level map: output plaquette field
    double* logj;            // [B] or null
    int logj_accumulate;     // logj[b] += instead of =
    double tol;              // reverse
    double* stash;           // this layer's region (arch.stash_doubles): written by the forward, read by the backward
    double* hbuf;            // [B][cmax][n] scratch: activations of the conv input
    double* gbuf;            // [2][B][cmax][n] scratch: gradients, ping-pong
    const double* up_gp;     // backward: upstream plaquette gradient [B][L][L] or null
    const double* up_link;   // backward: upstream link gradient [B][2][L][L] or null
    const double* glogj;     // backward: [B] or null (then glogj_const)
    double glogj_const;
    double* gp_out;          // backward: plaquette gradient behind this layer (not up_gp)
    double* gw;              // backward: this layer's weight gradient (canonical layout) or null
    int B, L, mu, off, act;
};
int launch_gen_fwd(const GenLayerArgs& a, bool rev, hipStream_t s);
int launch_gen_bwd(const GenLayerArgs& a, hipStream_t s);

// ---- flow_small.hip: L <= 16, one workgroup per chain, whole sequences of the flowed path in one launch
struct SmallArgs {
    const double* x;         // [B][2][L][L] latent links
    const double* v;         // leapfrog / trajectory: momenta
    const double* u;         // trajectory: accept uniforms [B]
    const double* wint;      // n_layers * FLOW_WINT, kernel weight layout
    double* stash;           // n_layers * flow_stash_doubles(B, L): the activation stash of a force evaluation
    const double* state_in;  // trajectory: [3][B] (S_eff, plaq, Q) of x, or null
    double* state_out;       // trajectory: [3][B] of x_new, or null
    double* x_out;           // action: F(x) (or null); leapfrog: x'; trajectory: x_new
    double* v_out;           // leapfrog: v'
    double* F;               // force
    double *dH, *acc, *H0, *H1, *S_eff, *logdet, *plaq, *Q;   // per chain [B], each may be null
    double *logq, *logp;     // training: [B]
    double* gz;              // training: n_layers * flow_gz_doubles(B, L): every layer's pre-activation gradients (for k_flow_wgrad)
    double beta, dt;
    int nstep, mode, B, nl, act;   // mode: 0 action (forward sweep), 1 force, 2 leapfrog, 3 trajectory, 4 training sweep
    long long* dbg;          // profiling runs: [B][32] cycles per stage, accumulated by thread 0 of each chain (else null)
};
bool ft_small_shape(int L, int n_layers);       // the fused path is built for this lattice size (default net shape, MFMA kernels)
int launch_ft_small(const SmallArgs& a, int L, hipStream_t s);
void set_small_path(int v);
int get_small_path();
// 0: VALU kernels everywhere; 1 (default): MFMA kernels for forward and backward-wrt-x
void set_flow_variant(int v);
int get_flow_variant();
// out[b] (+)= sign * sum_t part[b][t]
int launch_sum_parts(const double* part, int B, int nparts, double sign, int accumulate,
                     double* out, hipStream_t s, int nsets = 1);   // nsets partial sets [set][B][nparts], summed in order
// gp[b][i][j] (+)= sum of every partial window position that maps to (i, j)
int launch_gather_gp(const double* gp_part, int B, int L, FlowGeom g, int accumulate, double* gp, hipStream_t s);
// gx = gy + adj(gp)
int launch_adj_add(const double* gp, const double* gy, int B, int L, double* gx, hipStream_t s);
// gw[idx] (+)= scale * sum_p gw_part[p][idx], idx < 955
// tmp: FLOW_REDUCE_GROUPS * FLOW_GW_STRIDE doubles of scratch (two-level reduction), or null
constexpr int FLOW_REDUCE_GROUPS = 128;
// rows of tmp the two-level reduction of `nparts` partials uses per layer (0: one level, no tmp)
inline int flow_reduce_groups(int nparts) {
    if (nparts <= 64) return 0;
    int groups = (nparts + 31) / 32; if (groups > FLOW_REDUCE_GROUPS) groups = FLOW_REDUCE_GROUPS;
    const int chunk = (nparts + groups - 1) / groups;
    return (nparts + chunk - 1) / chunk;
}
int launch_reduce_gw(const double* gw_part, int nparts, double scale, int accumulate, double* gw,
                     double* tmp, hipStream_t s, int nlayers = 1, size_t part_lstride = 0);   // nlayers > 1: layer l reads gw_part + l * part_lstride, writes gw + l * 955; tmp: nlayers * flow_reduce_groups(nparts) rows

}  // namespace fthmc
