// Coupling layers with ANY s/t conv net: hidden_sizes of any length and width, any odd kernel_size, any number of
// mixture components (make_conv_net fthmc/utils/layers.py:138-167, make_u1_equiv_layers :399-429 accept all of them).
//
// The tuned kernels (flow_fwd.hip, flow_bwd_gather.hip, flow_small.hip) are built for the reference default 2 -> 8 -> 8 -> 3,
// k = 3, two mixture components -- what every reference script and every BASELINE config uses.  Anything else takes this
// file: one plain kernel per operation, activations as [B][C][L][L] planes in the caller's workspace, one thread per output
// element, circular taps by index wrap, fixed-order reductions.  Same mathematics (GaugeEquivCouplingLayer.forward / .reverse
// layers.py:188-210, NCPPlaqCouplingLayer :348-396, the tan-mixture transform :58-90) and the same analytic adjoint as the
// tuned kernels (flow_fwd.hip stash coefficients, flow_bwd_gather.hip); correctness first -- a layer costs a dozen launches
// and moves its activations through HBM.  The shape travels with the call (GenLayerArgs::arch; api.hip routes on is_default()).
#include "common.h"
#include "kernels.h"
#include "flow_common.h"

#define FT_TRY_RC(expr) do { int rc_ = (expr); if (rc_ != FTHMC_OK) return rc_; } while (0)

namespace {

using namespace fthmc;
using namespace fthmc_flow;

__device__ __forceinline__ int wrapc(int v, int L) { return v < 0 ? v + L : (v >= L ? v - L : v); }

__device__ __forceinline__ void act1(double z, int act, double& h, double& d) { act_eval(z, act, h, d); }

// ---- plaquettes and the net input (cos P, sin P on the frozen stripes, (1, 0) elsewhere)
__global__ void k_gen_input(const double* __restrict__ x, const double* __restrict__ pin, double* __restrict__ P,
                            double* __restrict__ IN, int L, int mu, int off) {
    const int b = blockIdx.y, n = L * L;
    const double* x0 = x ? x + (size_t)b * 2 * n : nullptr;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const int i = s / L, j = s - i * L;
        double p;
        if (pin) p = pin[(size_t)b * n + s];
        else {
            const int ip = i + 1 == L ? 0 : i + 1, jp = j + 1 == L ? 0 : j + 1;
            p = x0[s] - x0[n + s] - x0[i * L + jp] + x0[n + ip * L + j];
        }
        const int cls = ft_stripe(i, j, mu, off);
        double sn = 0.0, cs = 1.0;
        if (cls == 1 || cls == 2) ft_sincos(p, &sn, &cs);
        P[(size_t)b * n + s] = p;
        IN[((size_t)b * 2 + 0) * n + s] = cs;
        IN[((size_t)b * 2 + 1) * n + s] = sn;
    }
}

// ---- H = act(Z), elementwise
__global__ void k_gen_act(const double* __restrict__ Z, double* __restrict__ H, size_t n, int act) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { double h, d; act1(Z[i], act, h, d); H[i] = h; }
}

// ---- the optional tanh behind the last conv (make_conv_net(use_final_tanh=True), layers.py:163-164): Z <- tanh(Z) in place
//      (the raw pre-activation is not needed again: tanh' = 1 - tanh^2), and its adjoint G <- G (1 - Z^2)
__global__ void k_gen_tanh(double* __restrict__ Z, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) Z[i] = tanh(Z[i]);
}
__global__ void k_gen_tanh_bwd(const double* __restrict__ Z, double* __restrict__ G, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) G[i] *= 1.0 - Z[i] * Z[i];
}

// ---- circular conv: Z[b][co][s] = bias[co] + sum_{ci, ky, kx} w[co][ci][ky][kx] A[b][ci][s + (ky - r, kx - r)]
__global__ void k_gen_conv(const double* __restrict__ A, int cin, int cout, int k, const double* __restrict__ w,
                           const double* __restrict__ bias, double* __restrict__ Z, int L) {
    const int b = blockIdx.z, co = blockIdx.y, n = L * L, r = k / 2;
    const double* wc = w + (size_t)co * cin * k * k;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const int i = s / L, j = s - i * L;
        double z = bias[co];
        for (int ci = 0; ci < cin; ++ci) {
            const double* a = A + ((size_t)b * cin + ci) * n;
            for (int ky = 0; ky < k; ++ky) {
                const int ii = wrapc(i + ky - r, L) * L;
                for (int kx = 0; kx < k; ++kx) z = fma(wc[(ci * k + ky) * k + kx], a[ii + wrapc(j + kx - r, L)], z);
            }
        }
        Z[((size_t)b * cout + co) * n + s] = z;
    }
}

// s_k, t of the net at site s -> the transform's pieces
struct MixOut { double newP, si; };
__device__ __forceinline__ MixOut mix_forward(const double* __restrict__ Zb, int n, int s, int K, double Pa) {
    double sn, cs;
    ft_sincos(0.5 * Pa, &sn, &cs);
    const double tn = sn / cs, cs2 = cs * cs, sn2 = sn * sn;
    double ysum = 0.0, si = 0.0;
    for (int k = 0; k < K; ++k) {
        const double es = ft_exp(Zb[(size_t)k * n + s]), ems = ft_rcp(es);
        ysum += ft_wrap_pm_pi(2 * ft_atan(es * tn));
        si += ft_rcp(ems * cs2 + es * sn2);
    }
    return MixOut{ft_wrap(ysum / K + Zb[(size_t)K * n + s]), si};
}

// ---- tan-mixture transform at the active sites + link update (or plaquette-level output); one workgroup per chain
template <bool REV>
__global__ __launch_bounds__(256) void k_gen_transform(const double* __restrict__ x, const double* __restrict__ P,
                                                       const double* __restrict__ Z, double* __restrict__ y,
                                                       double* __restrict__ pout, double* __restrict__ logJ, int accumulate,
                                                       int L, int mu, int off, int K, double tol) {
    __shared__ double red[16];
    const int b = blockIdx.x, n = L * L, na = n / 4;
    const double* Pb = P + (size_t)b * n;
    const double* Zb = Z + (size_t)b * (K + 1) * n;
    if (y && x && y != x) {
        const double* xb = x + (size_t)b * 2 * n;
        for (int s = threadIdx.x; s < 2 * n; s += blockDim.x) y[(size_t)b * 2 * n + s] = xb[s];
        __syncthreads();
    }
    if (pout) { for (int s = threadIdx.x; s < n; s += blockDim.x) pout[(size_t)b * n + s] = Pb[s]; __syncthreads(); }
    double lj = 0.0;
    for (int a = threadIdx.x; a < na; a += blockDim.x) {
        int i, j;
        if (mu == 0) { i = a / (L / 4); j = off + 4 * (a - i * (L / 4)); } else { const int m = a / L; j = a - m * L; i = off + 4 * m; }
        const int s = i * L + j;
        const double Pa = Pb[s];
        double newP;
        if (!REV) {
            const MixOut m = mix_forward(Zb, n, s, K, Pa);
            newP = m.newP;
            lj += log(m.si) - log((double)K);
        } else {
            // inverse: solve mean_k y_k(xs) = wrap(P' - t) by safeguarded Newton (monotone map, derivative mean_k 1 / D_k;
            // the reference bisects to a global 1e-6, layers.py:294-320)
            const double target = ft_wrap(Pa - Zb[(size_t)K * n + s]);
            double lo = -FT_PI, hi = FT_PI, xs = target, fp = 1.0;
            for (int it = 0; it < 200; ++it) {
                double sn, cs;
                ft_sincos(0.5 * xs, &sn, &cs);
                const double tn = sn / cs;
                double f = 0.0;
                fp = 0.0;
                for (int k = 0; k < K; ++k) {
                    const double es = ft_exp(Zb[(size_t)k * n + s]), ems = ft_rcp(es);
                    f += ft_wrap_pm_pi(2 * ft_atan(es * tn));
                    fp += 1.0 / (ems * cs * cs + es * sn * sn);
                }
                f /= K; fp /= K;
                const double err = target - f;
                if (fabs(err) <= tol) break;
                if (err > 0) lo = xs; else hi = xs;
                double xn = xs + err / fp;
                if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
                if (xn == xs) break;
                xs = xn;
            }
            newP = xs;
            lj += -log(fp);
        }
        if (pout) pout[(size_t)b * n + s] = newP;
        if (y) {
            const double d = newP - Pa;
            double* yb = y + (size_t)b * 2 * n;
            if (mu == 0) yb[s] = ft_wrap(d + yb[s]); else yb[n + s] = ft_wrap(-d + yb[n + s]);
        }
    }
    if (logJ) {
        const double tot = ft_block_sum(lj, red);
        if (threadIdx.x == 0) logJ[b] = (accumulate ? logJ[b] : 0.0) + tot;
    }
}

// ---- adjoint of the transform: G = dL/d(net output) (zero off the active sites), gp_out = upstream + the active sites' part
__global__ void k_gen_transform_bwd(const double* __restrict__ P, const double* __restrict__ Z,
                                    const double* __restrict__ up_gp, const double* __restrict__ up_link,
                                    const double* __restrict__ glogj, double glogj_const, double* __restrict__ G,
                                    double* __restrict__ gp_out, int L, int mu, int off, int K) {
    const int b = blockIdx.y, n = L * L;
    const double* Zb = Z + (size_t)b * (K + 1) * n;
    double* Gb = G + (size_t)b * (K + 1) * n;
    const double cb = glogj ? glogj[b] : glogj_const;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const int i = s / L, j = s - i * L;
        const double g_up = up_gp ? up_gp[(size_t)b * n + s] : 0.0;
        if (ft_stripe(i, j, mu, off) != 0) {
            for (int k = 0; k <= K; ++k) Gb[(size_t)k * n + s] = 0.0;
            gp_out[(size_t)b * n + s] = g_up;
            continue;
        }
        double gdelta;
        if (up_link) gdelta = mu == 0 ? up_link[((size_t)b * 2 + 0) * n + s] : -up_link[((size_t)b * 2 + 1) * n + s];
        else {
            const int sp = mu == 0 ? i * L + (j == 0 ? L - 1 : j - 1) : (i == 0 ? L - 1 : i - 1) * L + j;   // the passive neighbour
            gdelta = g_up - up_gp[(size_t)b * n + sp];
        }
        const double Pa = P[(size_t)b * n + s];
        double sn, cs;
        ft_sincos(0.5 * Pa, &sn, &cs);
        const double cs2 = cs * cs, sn2 = sn * sn, sinP = 2.0 * sn * cs;
        double csum = 0.0, esum = 0.0;
        for (int k = 0; k < K; ++k) {                                  // C_k = 1 / (K D_k), E_k (struct Stash, flow_mfma_common.h)
            const double es = ft_exp(Zb[(size_t)k * n + s]), ems = ft_rcp(es);
            const double invD = ft_rcp(ems * cs2 + es * sn2);
            csum += invD / K;
            esum += sinP * 0.5 * (es - ems) * invD * invD;
        }
        const double cbr = cb / (K * csum);
        for (int k = 0; k < K; ++k) {
            const double es = ft_exp(Zb[(size_t)k * n + s]), ems = ft_rcp(es);
            const double invD = ft_rcp(ems * cs2 + es * sn2);
            Gb[(size_t)k * n + s] = gdelta * (sinP * invD / K) + cbr * ((ems * cs2 - es * sn2) * invD * invD);   // dL/ds_k
        }
        Gb[(size_t)K * n + s] = gdelta;                                                                      // dL/dt
        gp_out[(size_t)b * n + s] = g_up + gdelta * (csum - 1.0) - cbr * esum;
    }
}

// ---- conv^T: Gin[b][ci][s] = sum_{co, ky, kx} w[co][ci][ky][kx] Gz[b][co][s - (ky - r, kx - r)], times act'(Zprev)
__global__ void k_gen_conv_bwd_data(const double* __restrict__ Gz, int cout, int cin, int k, const double* __restrict__ w,
                                    const double* __restrict__ Zprev, int act, double* __restrict__ Gin, int L) {
    const int b = blockIdx.z, ci = blockIdx.y, n = L * L, r = k / 2;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const int i = s / L, j = s - i * L;
        double g = 0.0;
        for (int co = 0; co < cout; ++co) {
            const double* gz = Gz + ((size_t)b * cout + co) * n;
            const double* wc = w + ((size_t)co * cin + ci) * k * k;
            for (int ky = 0; ky < k; ++ky) {
                const int ii = wrapc(i - (ky - r), L) * L;
                for (int kx = 0; kx < k; ++kx) g = fma(wc[ky * k + kx], gz[ii + wrapc(j - (kx - r), L)], g);
            }
        }
        if (Zprev) { double h, d; act1(Zprev[((size_t)b * cin + ci) * n + s], act, h, d); g *= d; }
        Gin[((size_t)b * cin + ci) * n + s] = g;
    }
}

// ---- (cos, sin) adjoint at the frozen plaquettes: gp += -sin P g_cos + cos P g_sin
__global__ void k_gen_input_bwd(const double* __restrict__ IN, const double* __restrict__ Gin, double* __restrict__ gp,
                                int L, int mu, int off) {
    const int b = blockIdx.y, n = L * L;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const int cls = ft_stripe(s / L, s % L, mu, off);
        if (cls == 1 || cls == 2)
            gp[(size_t)b * n + s] += -IN[((size_t)b * 2 + 1) * n + s] * Gin[((size_t)b * 2 + 0) * n + s] +
                                     IN[((size_t)b * 2 + 0) * n + s] * Gin[((size_t)b * 2 + 1) * n + s];
    }
}

// ---- weight gradients of one conv: one workgroup per weight (and per bias), fixed-order sum over chains and sites
__global__ __launch_bounds__(256) void k_gen_conv_bwd_w(const double* __restrict__ Gz, const double* __restrict__ A, int cin,
                                                        int cout, int k, int B, int L, double* __restrict__ gw,
                                                        double* __restrict__ gb) {
    __shared__ double red[16];
    const int n = L * L, r = k / 2, nw = cout * cin * k * k;
    const int idx = blockIdx.x;
    double acc = 0.0;
    if (idx < nw) {
        const int kx = idx % k, ky = (idx / k) % k, ci = (idx / (k * k)) % cin, co = idx / (k * k * cin);
        for (int b = 0; b < B; ++b) {
            const double* gz = Gz + ((size_t)b * cout + co) * n;
            const double* a = A + ((size_t)b * cin + ci) * n;
            for (int s = threadIdx.x; s < n; s += blockDim.x) {
                const int i = s / L, j = s - i * L;
                acc = fma(gz[s], a[wrapc(i + ky - r, L) * L + wrapc(j + kx - r, L)], acc);
            }
        }
    } else {
        const int co = idx - nw;
        for (int b = 0; b < B; ++b) {
            const double* gz = Gz + ((size_t)b * cout + co) * n;
            for (int s = threadIdx.x; s < n; s += blockDim.x) acc += gz[s];
        }
    }
    acc = ft_block_sum(acc, red);
    if (threadIdx.x == 0) { if (idx < nw) gw[idx] = acc; else gb[idx - nw] = acc; }
}

inline int sgrid(int n) { int g = (n + 255) / 256; return g > 64 ? 64 : (g < 1 ? 1 : g); }
inline int egrid(size_t n) { size_t g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g ? g : 1)); }

}  // namespace

namespace fthmc {

int make_flow_arch(int nh, const int* hid, int k, int nmix, int final_tanh, FlowArch* out) {
    if (nh < 0 || nh > FLOW_ARCH_MAXH || k < 1 || (k & 1) == 0 || k > 15 || nmix < 1 || nmix > 64) return FTHMC_ERR_UNSUPPORTED;
    for (int i = 0; i < nh; ++i) if (!hid || hid[i] < 1 || hid[i] > 256) return FTHMC_ERR_UNSUPPORTED;
    FlowArch a{};
    a.nh = nh; a.k = k; a.nmix = nmix; a.tanh_out = final_tanh != 0;
    for (int i = 0; i < nh; ++i) a.hid[i] = hid[i];
    *out = a;
    return FTHMC_OK;
}

namespace {
struct GenStash { double *P, *IN, *Z[FLOW_ARCH_MAXH + 1]; };
GenStash gen_view(const FlowArch& A, double* base, int B, int L) {
    GenStash v{};
    const size_t n = (size_t)L * L;
    v.P = base; v.IN = base + (size_t)B * n;
    double* p = base + (size_t)B * 3 * n;
    for (int i = 0; i <= A.nh; ++i) { v.Z[i] = p; p += (size_t)B * A.chan(i + 1) * n; }
    return v;
}
struct GenW { const double *w, *b; };
GenW gen_w(const FlowArch& A, const double* wl, int conv) {          // canonical layout [w0 b0 w1 b1 ...], PyTorch [Cout][Cin][k][k]
    const double* p = wl;
    for (int i = 0; i < conv; ++i) p += A.chan(i + 1) * A.chan(i) * A.k * A.k + A.chan(i + 1);
    return GenW{p, p + A.chan(conv + 1) * A.chan(conv) * A.k * A.k};
}
}  // namespace

// net of one layer on the plaquettes of a.x (or a.pin): fills the layer's stash region (P, IN, every pre-activation)
static int gen_net(const GenLayerArgs& a, const GenStash& st, hipStream_t s) {
    const FlowArch& A_ = a.arch;
    const int n = a.L * a.L, nh = A_.nh, k = A_.k;
    hipLaunchKernelGGL(k_gen_input, dim3(sgrid(n), a.B), dim3(256), 0, s, a.x, a.pin, st.P, st.IN, a.L, a.mu, a.off);
    FT_LAUNCH_CHECK();
    const double* A = st.IN;
    for (int i = 0; i <= nh; ++i) {
        const int cin = A_.chan(i), cout = A_.chan(i + 1);
        const GenW W = gen_w(A_, a.w, i);
        hipLaunchKernelGGL(k_gen_conv, dim3(sgrid(n), cout, a.B), dim3(256), 0, s, A, cin, cout, k, W.w, W.b, st.Z[i], a.L);
        FT_LAUNCH_CHECK();
        if (i < nh) {
            hipLaunchKernelGGL(k_gen_act, dim3(egrid((size_t)a.B * cout * n)), dim3(256), 0, s, st.Z[i], a.hbuf, (size_t)a.B * cout * n, a.act);
            FT_LAUNCH_CHECK();
            A = a.hbuf;
        }
    }
    if (A_.tanh_out) {
        const size_t nz = (size_t)a.B * (A_.nmix + 1) * n;
        hipLaunchKernelGGL(k_gen_tanh, dim3(egrid(nz)), dim3(256), 0, s, st.Z[nh], nz);
        FT_LAUNCH_CHECK();
    }
    return FTHMC_OK;
}

int launch_gen_fwd(const GenLayerArgs& a, bool rev, hipStream_t s) {
    if (a.arch.k / 2 > a.L) return FTHMC_ERR_UNSUPPORTED;    // a circular pad wider than the lattice (torch's Conv2d refuses it too)
    const GenStash st = gen_view(a.arch, a.stash, a.B, a.L);
    FT_TRY_RC(gen_net(a, st, s));
    if (rev) hipLaunchKernelGGL(k_gen_transform<true>, dim3(a.B), dim3(256), 0, s, a.x, st.P, st.Z[a.arch.nh], a.y, a.pout, a.logj,
                                a.logj_accumulate, a.L, a.mu, a.off, a.arch.nmix, a.tol);
    else hipLaunchKernelGGL(k_gen_transform<false>, dim3(a.B), dim3(256), 0, s, a.x, st.P, st.Z[a.arch.nh], a.y, a.pout, a.logj,
                            a.logj_accumulate, a.L, a.mu, a.off, a.arch.nmix, 0.0);
    FT_LAUNCH_CHECK();
    return FTHMC_OK;
}

// backward of one layer from its stash region: gp_out = upstream + layer contribution; gw (optional): this layer's weights
int launch_gen_bwd(const GenLayerArgs& a, hipStream_t s) {
    if (a.arch.k / 2 > a.L) return FTHMC_ERR_UNSUPPORTED;
    const FlowArch& A_ = a.arch;
    const int n = a.L * a.L, nh = A_.nh, k = A_.k, K = A_.nmix;
    const GenStash st = gen_view(A_, a.stash, a.B, a.L);
    double* G = a.gbuf;
    double* G2 = a.gbuf + (size_t)a.B * A_.cmax() * n;
    hipLaunchKernelGGL(k_gen_transform_bwd, dim3(sgrid(n), a.B), dim3(256), 0, s, st.P, st.Z[nh], a.up_gp, a.up_link, a.glogj,
                       a.glogj_const, G, a.gp_out, a.L, a.mu, a.off, K);
    FT_LAUNCH_CHECK();
    if (A_.tanh_out) {
        const size_t nz = (size_t)a.B * (K + 1) * n;
        hipLaunchKernelGGL(k_gen_tanh_bwd, dim3(egrid(nz)), dim3(256), 0, s, st.Z[nh], G, nz);
        FT_LAUNCH_CHECK();
    }
    for (int i = nh; i >= 0; --i) {
        const int cin = A_.chan(i), cout = A_.chan(i + 1);
        const GenW W = gen_w(A_, a.w, i);
        if (a.gw) {
            const double* A = st.IN;
            if (i > 0) {
                hipLaunchKernelGGL(k_gen_act, dim3(egrid((size_t)a.B * cin * n)), dim3(256), 0, s, st.Z[i - 1], a.hbuf, (size_t)a.B * cin * n, a.act);
                FT_LAUNCH_CHECK();
                A = a.hbuf;
            }
            double* gwl = a.gw + (W.w - a.w);
            hipLaunchKernelGGL(k_gen_conv_bwd_w, dim3(cout * cin * k * k + cout), dim3(256), 0, s, G, A, cin, cout, k, a.B, a.L,
                               gwl, gwl + (size_t)cout * cin * k * k);
            FT_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(k_gen_conv_bwd_data, dim3(sgrid(n), cin, a.B), dim3(256), 0, s, G, cout, cin, k, W.w,
                           i > 0 ? st.Z[i - 1] : nullptr, a.act, G2, a.L);
        FT_LAUNCH_CHECK();
        double* t = G; G = G2; G2 = t;
    }
    hipLaunchKernelGGL(k_gen_input_bwd, dim3(sgrid(n), a.B), dim3(256), 0, s, st.IN, G, a.gp_out, a.L, a.mu, a.off);
    FT_LAUNCH_CHECK();
    return FTHMC_OK;
}

}  // namespace fthmc
