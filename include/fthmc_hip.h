/* fthmc_hip.h -- C ABI of the MI355X (gfx950) ftHMC hot path.
 *
 * The reference (nftqcd/fthmc) is pure Python on PyTorch and has no FFI of its
 * own; each entry point below replaces the Python callable cited next to it
 * (paths relative to the reference root).  All pointers are DEVICE pointers to
 * contiguous fp64 data unless marked `host`.  Nothing here allocates or
 * synchronises: every call only enqueues kernels on `stream` (a hipStream_t
 * passed as void*; NULL = the null stream) and returns 0 or a negative
 * FTHMC_ERR_* code.  Scratch space is caller-owned (`ws`, sized by
 * fthmc_ws_bytes) so that calls can be captured into a hipGraph.  A workspace
 * belongs to one stream at a time: calls in flight on different streams (e.g.
 * two groups of independent chains) need a workspace each.
 *
 * Field layout: x[B][2][L][L], angle of the U(1) link in radians, mu-major
 * (the reference's [batch, Nd, Nt, Nx]).  L % 4 == 0.
 * Flow weights: n_layers * FTHMC_W_PER_LAYER doubles; per layer the six
 * nn.Conv2d tensors of `layers[i].plaq_coupling.net` in state_dict order and
 * PyTorch [Cout][Cin][kh][kw] layout:
 *     w0[8][2][3][3] b0[8] w1[8][8][3][3] b1[8] w2[3][8][3][3] b2[3]
 * (hidden_sizes=[8,8], kernel_size=3, n_mixture_comps=2: the reference default,
 * fthmc/config.py:283-303).  Layer i uses mu = i % 2, off = (i / 2) % 4
 * (fthmc/utils/layers.py:409-412).
 */
#ifndef FTHMC_HIP_H
#define FTHMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FTHMC_W_PER_LAYER 955

#define FTHMC_OK               0
#define FTHMC_ERR_ARG         -1   /* bad shape / null pointer              */
#define FTHMC_ERR_UNSUPPORTED -2   /* architecture outside the built kernels */
#define FTHMC_ERR_LAUNCH      -3   /* hipGetLastError() after a launch       */
#define FTHMC_ERR_WS          -4   /* workspace too small                    */

/* activation_fn of the s/t conv net (fthmc/utils/layers.py:117-135) */
#define FTHMC_ACT_SILU        0
#define FTHMC_ACT_RELU        1
#define FTHMC_ACT_LEAKY_RELU  2

/* trajectory modes */
#define FTHMC_MODE_MD         0   /* intended integrator, ipynb/ft_hmc.py:394-435 */
#define FTHMC_MODE_LITERAL    1   /* FieldTransformation.leapfrog as packaged, fthmc/ft_hmc.py:180-188 */

const char* fthmc_version(void);
const char* fthmc_strerror(int code);
/* text of the HIP error behind the last FTHMC_ERR_LAUNCH on this thread ("" if none) */
const char* fthmc_last_error(void);

/* Kernel variant of the coupling-layer forward / backward-wrt-x kernels: 1 (default) MFMA f64 16x16x4 implicit-GEMM
 * convolutions, 0 fp64-VALU convolutions.  Same results to rounding.  A process-wide DEBUG switch (A/B measurement and
 * cross-checks), like fthmc_set_small_path below: each entry point reads both once, when it is entered; they are the only
 * mutable state of the library and nothing a production caller touches. */
int fthmc_set_variant(int v);
int fthmc_get_variant(void);
/* Shape of the s/t conv net, an ARGUMENT of every entry point that runs the net (the reference passes it to the layer
 * constructors: make_conv_net fthmc/utils/layers.py:138-167, make_u1_equiv_layers :399-429; TrainConfig.hidden_sizes /
 * kernel_size / n_s_nets, fthmc/config.py:283-303): in_channels 2 -> hidden[0] -> ... -> hidden[n_hidden - 1] -> n_mix + 1,
 * square kernels of odd size, `n_mix` mixture components, `final_tanh` != 0: a tanh behind the last conv (make_conv_net's
 * use_final_tanh, layers.py:144,163-164; the reference passes False, :419).  NULL = the default (2, {8, 8}, 3, 2, 0) -- the reference default and
 * every BASELINE config -- which runs on the tuned kernels; any other shape runs on plain kernels (csrc/flow_generic.hip:
 * same results, one launch per operation, activations through HBM).  Weights: n_layers * fthmc_arch_params(arch) doubles,
 * per layer [w0 b0 w1 b1 ...] in PyTorch order; the workspace sizes follow the shape.  Limits: n_hidden <= 8, hidden sizes
 * <= 256, kernel_size <= 15 (and kernel_size / 2 <= L), n_mix <= 64; FTHMC_ERR_UNSUPPORTED otherwise.
 * The tuned kernels address one layer's activation stash with 32-bit element offsets: B * 19 * L * L (training: * 35)
 * must stay below 2^32 doubles (32 GiB per layer; FTHMC_ERR_UNSUPPORTED beyond -- shard the chains instead).
 * The shape is read during the call only: the library keeps no net shape between calls, two threads may run two different
 * flows on two streams at the same time. */
typedef struct fthmc_arch_t {
    int n_hidden;
    int hidden[8];
    int kernel_size;
    int n_mix;
    int final_tanh;
} fthmc_arch_t;
int fthmc_arch_params(const fthmc_arch_t* arch);   /* doubles per layer; 955 for the default */
/* Lattices of L = 8, 12, 16 take a fused path by default (csrc/flow_small.hip): one workgroup holds a whole chain in
 * LDS, and fthmc_ft_trajectory / _ft_leapfrog / _ft_force / _ft_action / _flow_forward are ONE launch each instead of
 * one launch per layer.  0 switches it off (the tiled kernels then serve every L): A/B runs and parity tests. */
int fthmc_set_small_path(int on);

/* Weight versions (the `_v` entry points below and fthmc_pack_weights).  The tuned kernels read a kernel-layout EXPANSION of
 * the canonical weights (77 KB per layer) that every entry point which runs the net writes into the head of its workspace
 * first (one launch, ~7 us; the reference packs nothing: its convs read nn.Conv2d parameters, fthmc/utils/layers.py:138-167).
 * A caller whose weights stay put between calls -- a sampler replaying a captured trajectory -- states their CONTENT VERSION
 * (any 64-bit number it changes whenever the weights' contents change; 0 = no statement): the launch then compares, on the
 * device, a token of (version, address of w, layer) with the stamp the last expansion left next to each layer's expansion
 * and expands only the layers whose stamp differs.  The library keeps no state between calls and trusts nothing: a wrong,
 * stale or reused version, another tensor, a call without a version in between (it clears the stamps it overwrites) all end
 * in an expansion, i.e. in the same numbers.  The one promise left with the caller is the meaning of the number: equal
 * version + equal address = equal contents.
 * The expansions of the first 64 layers are the HEAD of every workspace layout (fthmc_ws_head_bytes() bytes; no call of
 * any shape keeps anything else there); deeper layers are expanded by every call.  A workspace must be zero in its head
 * before its first use with a version (fresh memory may hold the stamps of an earlier life of the same address). */
size_t fthmc_ws_head_bytes(void);

/* Expand the canonical weights (n_layers x params, the layout of every `w` argument below) into the workspace under
 * `weights_version` (0: unconditionally) and do nothing else: what every entry point that runs the net does first. */
int fthmc_pack_weights(const double* w, const fthmc_arch_t* arch, int n_layers, uint64_t weights_version,
                       void* ws, size_t ws_bytes, void* stream);
int fthmc_get_small_path(void);

/* Bytes of scratch the flow / trajectory entry points need for (B, L, n_layers). */
size_t fthmc_ws_bytes(const fthmc_arch_t* arch, int B, int L, int n_layers);
/* Scratch for fthmc_train_grad (larger: the forward also stashes h1, h2 of every layer). */
size_t fthmc_train_ws_bytes(const fthmc_arch_t* arch, int B, int L, int n_layers);

/* ---- angle maps ------------------------------------------------------- */
/* out = remainder(x + pi, 2 pi) - pi.  fthmc/utils/layers.py:41-43 (torch_mod),
 * fthmc/utils/qed_helpers.py:49-50 (torch_wrap), fthmc/ft_hmc.py:173-175 (wrap). */
int fthmc_wrap(const double* x, double* out, size_t n, void* stream);
/* fthmc/utils/qed_helpers.py:40-42 (regularize) */
int fthmc_regularize(const double* x, double* out, size_t n, void* stream);

/* ---- Wilson action / plaquette / topological charge ------------------- */
/* P[b][i][j] = x0 - x1 - x0[i][j+1] + x1[i+1][j].
 * fthmc/utils/qed_helpers.py:94-105 (batch_plaqs), :80-90 (compute_u1_plaq). */
int fthmc_plaquettes(const double* x, double* P, int B, int L, void* stream);
/* S[b] = -beta sum cos P; Q[b] = sum wrap(P) / 2pi; plaq[b] = -S / (beta L^2) (formed that way on every path, tiled and
 * small-lattice alike, as the reference forms it, fthmc/hmc.py:125: at beta = 0 it is 0 / 0 = NaN there and here).
 * Any of S/Q/plaq may be NULL.  fthmc/utils/qed_helpers.py:177-186 (BatchAction),
 * :108-116 (batch_charges), fthmc/hmc.py:125 (plaq). */
int fthmc_wilson_action_charge(const double* x, int B, int L, double beta,
                               double* S, double* Q, double* plaq, void* stream);
/* F = dS/dx.  fthmc/utils/qed_helpers.py:265-272 (force, via autograd there). */
int fthmc_wilson_force(const double* x, int B, int L, double beta, double* F, void* stream);

/* ---- plain HMC --------------------------------------------------------- */
/* x_, p_ = leapfrog(x, p): fthmc/utils/qed_helpers.py:275-295.  x_out/p_out must
 * not alias x/p.  ws: fthmc_ws_bytes(NULL, B, L, 0). */
int fthmc_leapfrog(const double* x, const double* p, int B, int L, double beta,
                   double dt, int nstep, double* x_out, double* p_out,
                   void* ws, size_t ws_bytes, void* stream);
/* K[b] = sum_b v^2 (no 1/2).  Used for H = S + K/2 (qed_helpers.py:301) */
int fthmc_kinetic(const double* v, int B, int L, double* K, void* stream);
/* Run statistics of one trajectory (the per-trajectory metrics of fthmc/ft_hmc.py:266-270, 311-331 and hmc.py:118-149) folded
 * into running sums on the device: vec8[0..7] += sum over the B chains of (1, acc, plaq, Q, Q^2, |Q - qold|, dH, exp(-dH)),
 * then qold <- Q.  One launch, fixed summation order. */
int fthmc_stats_accumulate(const double* acc, const double* plaq, const double* Q, double* qold, const double* dH, int B,
                           double* vec8, void* stream);
/* One trajectory per chain with supplied momenta v[B][2][L][L] and uniforms u[B]:
 * H0 = S(x) + v^2/2; leapfrog; xr = regularize(x_); dH = H1 - H0;
 * acc = u < exp(-dH); x_new = acc ? xr : x.   fthmc/utils/qed_helpers.py:298-311
 * (there the whole tensor is one system; chains are independent here, identical
 * for B = 1).  Outputs dH[B], acc[B] (0.0 / 1.0); H0/H1 may be NULL. */
int fthmc_hmc_trajectory(const double* x, const double* v, const double* u,
                         int B, int L, double beta, double dt, int nstep,
                         double* x_new, double* dH, double* acc, double* H0, double* H1,
                         void* ws, size_t ws_bytes, void* stream);

/* Momentum refresh for the production path: v[b][0..n) ~ N(0,1), u[b] ~ U[0,1) from a
 * Philox4x32-10 stream keyed by seeds[b] (device int64[B]).  Replaces torch.randn_like /
 * torch.rand of fthmc/utils/qed_helpers.py:300,306 and fthmc/ft_hmc.py:204,212; a chain's
 * draws depend only on its seed, so sharding chains over GPUs does not change them.
 * u may be NULL. */
int fthmc_random_momenta(const int64_t* seeds, int B, int n_per_chain, double* v, double* u,
                         void* stream);

/* ---- coupling layers ---------------------------------------------------- */
/* (y, logJ[B]) = GaugeEquivCouplingLayer.forward(x): fthmc/utils/layers.py:196-202
 * with NCPPlaqCouplingLayer.forward :348-371.  w: one layer (955 doubles).
 * y may alias x. */
int fthmc_flow_layer_fwd(const double* x, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off,
                         int act, double* y, double* logJ,
                         void* ws, size_t ws_bytes, void* stream);
/* VJP of the layer wrt x: gx = d/dx [ sum(gy * y) + sum_b glogJ[b] logJ[b] ]
 * (what autograd does for fthmc/utils/qed_helpers.py:226-242 and train.py:210).
 * gw != NULL additionally returns the same VJP wrt the 955 weights; the workspace must then hold
 * fthmc_train_ws_bytes(arch, B, L, 1). */
int fthmc_flow_layer_bwd(const double* x, const double* w, const fthmc_arch_t* arch, const double* gy, const double* glogJ,
                         int B, int L, int mu, int off, int act,
                         double* gx, double* gw,
                         void* ws, size_t ws_bytes, void* stream);
/* The same pair for callers that keep a layer's activations between its forward and its backward (an autograd graph:
 * fthmc/utils/layers.py:196-202 under loss.backward(), train.py:210): the forward also fills `stash`
 * (fthmc_layer_stash_bytes(arch, B, L) bytes, caller-owned), the backward reads it and recomputes nothing.
 * fthmc_layer_stash_bytes is 0 where no stash exists (fthmc_set_variant(0)): use fthmc_flow_layer_bwd there. */
size_t fthmc_layer_stash_bytes(const fthmc_arch_t* arch, int B, int L);
int fthmc_flow_layer_fwd_stash(const double* x, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off, int act,
                               double* y, double* logJ, double* stash, void* ws, size_t ws_bytes, void* stream);
int fthmc_flow_layer_bwd_stash(const double* stash, const double* w, const fthmc_arch_t* arch, const double* gy, const double* glogJ,
                               int B, int L, int mu, int off, int act, double* gx, double* gw,
                               void* ws, size_t ws_bytes, void* stream);
/* (x, logJ[B]) = GaugeEquivCouplingLayer.reverse(y): fthmc/utils/layers.py:204-210,
 * :373-396.  The scalar inverse is solved per site to |f(x) - y| <= tol by
 * safeguarded Newton/bisection on [-pi, pi] (the reference bisects to a global
 * 1e-6, layers.py:294-320).  x may alias y (a layer rewrites only its active links, and every
 * plaquette a workgroup uses is built from links the layer leaves alone plus its own). */
int fthmc_flow_layer_rev(const double* y, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off,
                         int act, double tol, double* x, double* logJ,
                         void* ws, size_t ws_bytes, void* stream);

/* Plaquette-level coupling map on a plaquette field P[B][L][L] (no links involved):
 * (fP, logJ[B]) = NCPPlaqCouplingLayer.forward(P): fthmc/utils/layers.py:348-371 -- P' at the active
 * plaquettes, P at the frozen and passive ones; and its inverse NCPPlaqCouplingLayer.reverse(fP):
 * fthmc/utils/layers.py:373-396 (logJ of the inverse = -sum log dP'/dP; `tol` as fthmc_flow_layer_rev).
 * Served by the MFMA kernels only (FTHMC_ERR_UNSUPPORTED with fthmc_set_variant(0)).  logJ may be NULL. */
int fthmc_plaq_coupling_fwd(const double* P, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off, int act,
                            double* fP, double* logJ, void* ws, size_t ws_bytes, void* stream);
int fthmc_plaq_coupling_rev(const double* fP, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off, int act,
                            double tol, double* P, double* logJ, void* ws, size_t ws_bytes, void* stream);
/* VJP of fthmc_plaq_coupling_fwd (autograd through NCPPlaqCouplingLayer.forward, layers.py:348-371, on a plaquette field):
 * gP = d/dP [ sum(gfP * fP) + sum_b glogJ[b] logJ[b] ]; gw != NULL additionally the same wrt the layer's weights (the workspace
 * must then hold fthmc_train_ws_bytes(arch, B, L, 1)).  The layer is run forward once inside. */
int fthmc_plaq_coupling_bwd(const double* P, const double* w, const fthmc_arch_t* arch, const double* gfP, const double* glogJ,
                            int B, int L, int mu, int off, int act, double* gP, double* gw,
                            void* ws, size_t ws_bytes, void* stream);

/* ---- whole flow ---------------------------------------------------------- */
/* Each entry point of this section has a `_v` twin with a trailing `weights_version` (see "Weight versions" above);
 * the plain form is the twin with version 0. */
/* y = F(x), logdet[B] = sum_l logJ_l.  fthmc/ft_hmc.py:143-150 (flow_forward),
 * fthmc/utils/qed_helpers.py:191-198 (ft_flow).  y, logdet may be NULL. */
int fthmc_flow_forward(const double* x, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                       double* y, double* logdet, void* ws, size_t ws_bytes, void* stream);
int fthmc_flow_forward_v(const double* x, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                       double* y, double* logdet, void* ws, size_t ws_bytes, void* stream, uint64_t weights_version);
/* x = F^-1(y), logdet[B].  fthmc/ft_hmc.py:152-160, qed_helpers.py:201-209.  x may alias y: the last layer maps
 * y -> x and the remaining layers run in place on x (see fthmc_flow_layer_rev). */
int fthmc_flow_reverse(const double* y, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                       double tol, double* x, double* logdet,
                       void* ws, size_t ws_bytes, void* stream);
int fthmc_flow_reverse_v(const double* y, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                       double tol, double* x, double* logdet,
                       void* ws, size_t ws_bytes, void* stream, uint64_t weights_version);
/* S_eff[b] = S_W(F(x)) - logdet.  fthmc/utils/qed_helpers.py:212-223 (ft_action),
 * fthmc/ft_hmc.py:135-141.  Optional outputs (NULL to skip): logdet[B],
 * plaq[B], Q[B] of the physical field F(x). */
int fthmc_ft_action(const double* x, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                    double beta, double* S_eff, double* logdet, double* plaq, double* Q,
                    void* ws, size_t ws_bytes, void* stream);
int fthmc_ft_action_v(const double* x, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                    double beta, double* S_eff, double* logdet, double* plaq, double* Q,
                    void* ws, size_t ws_bytes, void* stream, uint64_t weights_version);
/* F = d(sum_b S_eff)/dx.  fthmc/utils/qed_helpers.py:226-242 (ft_force),
 * fthmc/ft_hmc.py:162-171. */
int fthmc_ft_force(const double* x, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                   double beta, double* F, void* ws, size_t ws_bytes, void* stream);
int fthmc_ft_force_v(const double* x, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                   double beta, double* F, void* ws, size_t ws_bytes, void* stream, uint64_t weights_version);
/* x_, v_ = leapfrog with ft_force.  ipynb/ft_hmc.py:394-418. */
int fthmc_ft_leapfrog(const double* x, const double* v, const double* w, const fthmc_arch_t* arch, int n_layers,
                      int B, int L, int act, double beta, double dt, int nstep,
                      double* x_out, double* v_out, void* ws, size_t ws_bytes, void* stream);
int fthmc_ft_leapfrog_v(const double* x, const double* v, const double* w, const fthmc_arch_t* arch, int n_layers,
                      int B, int L, int act, double beta, double dt, int nstep,
                      double* x_out, double* v_out, void* ws, size_t ws_bytes, void* stream, uint64_t weights_version);
/* One ftHMC trajectory per chain in the latent field x with supplied v, u[B].
 * mode FTHMC_MODE_MD: ipynb/ft_hmc.py:420-435 without the flow-inverse wrapper;
 * FTHMC_MODE_LITERAL: fthmc/ft_hmc.py:190-224 as packaged.
 * Outputs: x_new (latent), dH[B], acc[B] (0/1), and plaq[B], Q[B] of F(x_new)
 * (fthmc/ft_hmc.py:266-270, 311-313); H0/H1/plaq/Q may be NULL.
 * Chaining: state_out[3][B] (nullable) receives [S_eff, plaq, Q] of x_new; passing it back as
 * state_in of the next call (x = this x_new) skips the H0 flow sweep, which would recompute
 * exactly these numbers (the reference recomputes, ft_hmc.py:205).  state_in NULL = stateless. */
int fthmc_ft_trajectory(const double* x, const double* v, const double* u, const double* w, const fthmc_arch_t* arch,
                        int n_layers, int B, int L, int act, double beta, double dt, int nstep,
                        int mode, double* x_new, double* dH, double* acc,
                        double* H0, double* H1, double* plaq, double* Q,
                        const double* state_in, double* state_out,
                        void* ws, size_t ws_bytes, void* stream);
int fthmc_ft_trajectory_v(const double* x, const double* v, const double* u, const double* w, const fthmc_arch_t* arch,
                        int n_layers, int B, int L, int act, double beta, double dt, int nstep,
                        int mode, double* x_new, double* dH, double* acc,
                        double* H0, double* H1, double* plaq, double* Q,
                        const double* state_in, double* state_out,
                        void* ws, size_t ws_bytes, void* stream, uint64_t weights_version);

/* ---- training ------------------------------------------------------------ */
/* Reverse-KL loss pieces and weight gradients for a fixed prior draw xi
 * (fthmc/train.py:191-210, fthmc/utils/samplers.py:40-56):
 *   x = F(xi); logq = -2 L^2 log(2 pi) - logdet; logp = -S_W(x);
 *   loss = mean(logq - logp); gw = d loss / d w  (n_layers*955).
 * Outputs (any may be NULL): x[B][2][L][L], logq[B], logp[B], gw.
 * ws: fthmc_train_ws_bytes(arch, B, L, n_layers). */
int fthmc_train_grad(const double* xi, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                     double beta, double* x, double* logq, double* logp, double* gw,
                     void* ws, size_t ws_bytes, void* stream);

/* The prior draw of a training step on the device: out[b][0..n) ~ U[lo, hi) from Philox4x32-10 keyed by the per-chain
 * seed (MultivariateUniform.sample_n, fthmc/utils/distributions.py:65-76, called at fthmc/train.py:191 through
 * apply_flow_to_prior, fthmc/utils/samplers.py:40-56).  A chain's draw depends on its seed only, not on the sharding. */
int fthmc_random_uniform(const int64_t* seeds, int B, int n_per_chain, double lo, double hi, double* out, void* stream);

/* Per-chain seeds of one trajectory / training step, formed on the device: seeds[b] = the 63-bit SplitMix64 mix of
 * (seed, global chain id lo + b, t) -- the same numbers as the host helper of the multi-GPU layer (fthmc_amd/parallel.py
 * chain_seeds; SURVEY 7 "RNG": a chain's stream is keyed by its global id, never by the rank).  t = traj + *counter when
 * `counter` (a device int64) is given, else traj; with advance != 0 the kernel then adds 1 to *counter, so that a
 * captured launch draws fresh seeds at every replay without a host-to-device copy.  Replaces the host side of
 * torch.randn_like / prior.sample_n in the loops of fthmc/ft_hmc.py:204 and fthmc/train.py:191. */
int fthmc_chain_seeds(int64_t seed, int64_t lo, int B, int64_t traj, int64_t* counter, int advance, int64_t* seeds, void* stream);

/* Metrics of one training step from its pieces (train_step, fthmc/train.py:206-228; calc_dkl / calc_ess,
 * fthmc/utils/distributions.py:23-37; batch_charges, fthmc/utils/qed_helpers.py:108-116), ONE rank's batch:
 *   row[0] = loss_dkl = dkl_factor * mean(logq - logp)       row[1] = ess = exp(2 lse(logw) - lse(2 logw)) / B
 *   row[2 + k B + b], k = 0..4:  logp, logq, q = Q(x), dq = |Q(x) - Q(xi)|, plaq = logp / (beta L^2)   of chain b
 * (2 + 5 B doubles: what train_step returns, stacked, so that a training loop copies ONE buffer to the host, when it
 * wants to look).  ws: fthmc_ws_bytes(NULL, B, L, 0). */
int fthmc_train_metrics(const double* xi, const double* x, const double* logq, const double* logp, int B, int L,
                        double beta, double dkl_factor, double* row, void* ws, size_t ws_bytes, void* stream);

/* The optimizer step of the training loop on the flat parameter buffer: torch.optim.Adam (decoupled = 0; fthmc/train.py:297
 * optim.Adam(model['layers'].parameters(), lr=config.base_lr)) or AdamW (decoupled = 1; train.py:86), torch's update formulas
 * element by element, in ONE launch over all n = n_layers * params values:  w, exp_avg, exp_avg_sq updated in place from gw.
 * hyper: THREE doubles on the device -- [0] the number of steps taken so far (the launch moves it on), [1] the learning rate,
 * [2] scratch, zero-initialised -- so that a captured launch can be replayed step after step and a scheduler changes the
 * rate by writing hyper[1]. */
int fthmc_adam_step(double* w, const double* gw, double* exp_avg, double* exp_avg_sq, double* hyper, size_t n,
                    double beta1, double beta2, double eps, double weight_decay, int decoupled, void* stream);

/* ---- measurement hook (the only entry point that synchronises) ---------- */
/* Average duration in milliseconds (host double) of `reps` back-to-back launches of one
 * coupling-layer kernel on `stream`, bracketed by HIP events recorded on that stream.
 * kind 0: forward kernel; 1: backward-wrt-x kernel of the force path (reads the forward's
 * activation stash with the MFMA variant, recomputes the forward with the VALU variant);
 * 2: fused plain-HMC leapfrog step (Wilson force stencil); 3: whole plain-HMC trajectory of 10
 * steps in one launch (L <= 64). */
int fthmc_time_kernel(int kind, const double* x, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off,
                      int act, double beta, int reps, double* ms_avg_host,
                      void* ws, size_t ws_bytes, void* stream);

/* Measurement hook: average milliseconds per launch of the small-lattice fused trajectory kernel (two action sweeps,
 * nstep force sweeps, Metropolis), `reps` launches back to back between two HIP events on `stream`.  Synchronises. */
int fthmc_time_small(const double* x, const double* v, const double* u, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L, int act,
                     double beta, double dt, int nstep, int reps, double* ms_avg_host, void* ws, size_t ws_bytes, void* stream);
/* Measurement hook: one trajectory on the small-lattice fused path (fthmc_set_small_path) with cycle stamps of thread 0
 * of every chain; cycles_host32[k] = mean over chains of the cycles spent in stage k, summed over the trajectory
 * (0..6 forward layer stages, 8..13 backward layer stages, 16 copy, 17 Wilson seed, 18 kick, 19 action / charge).
 * Synchronises. */
int fthmc_small_profile(const double* x, const double* v, const double* u, const double* w, const fthmc_arch_t* arch, int n_layers, int B, int L,
                        int act, double beta, double dt, int nstep, double* cycles_host32, void* ws, size_t ws_bytes,
                        void* stream);
/* Diagnostic (synchronises, mallocs on the host): one launch of the MFMA forward (kind 0), stash backward
 * (kind 1), training backward (kind 2, ws: fthmc_train_ws_bytes) or weight-gradient (kind 3, same ws) kernel with per-workgroup
 * cycle stamps at every stage boundary;
 * cycles_host16[k] = mean cycles spent between stamp k-1 and stamp k. */
int fthmc_profile_stages(int kind, const double* x, const double* w, const fthmc_arch_t* arch, int B, int L, int mu, int off,
                         int act, double beta, double* cycles_host16,
                         void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FTHMC_HIP_H */
