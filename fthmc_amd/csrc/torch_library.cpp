// TORCH_LIBRARY(fthmc_hip): the torch operator boundary of SURVEY.md 8(b), level 2, as a COMPILED dispatcher library.
//
// Every operator is a thin entry over the C ABI of libfthmc_hip.so (include/fthmc_hip.h): inputs are contiguous fp64 tensors
// on the HIP device, outputs and the per-call workspace are allocated by torch (caching allocator), the launch goes to
// at::hip::getCurrentHIPStream() and nothing synchronises; errors become c10::Error (RuntimeError).  Only the device dispatch
// key is registered: a CPU tensor fails in the dispatcher (there is no CPU fallback).  The s/t net's shape travels in the
// schema as plain integers (n_mix, hidden, kernel_size) and becomes the fthmc_arch_t of the call.
// Shape functions for tracing and the autograd formulas (which call the backward operators below) are attached from Python
// (fthmc_amd/torch_ops.py: torch.library.register_fake / register_autograd on these definitions).
//
// Built by csrc/Makefile with g++ (host code only; links libfthmc_hip.so next to it and PyTorch's libraries).
// Reference call sites each operator replaces: fthmc_amd/torch_ops.py docstring.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>

#include "../../include/fthmc_hip.h"

namespace {

using at::Tensor;
using IntList = c10::optional<at::IntArrayRef>;

// every operator runs with its first tensor's device current: the workspace and the outputs are allocated there and the launch
// goes to THAT device's current stream, whatever device the calling thread had selected (several GPUs in one process)
#define FT_DEVICE_GUARD(t) const c10::hip::OptionalHIPGuardMasqueradingAsCUDA device_guard_((t).device())

void* cur_stream(const Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

const double* cp(const Tensor& t) { return t.const_data_ptr<double>(); }
double* mp(Tensor& t) { return t.mutable_data_ptr<double>(); }

Tensor dev64(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), name, ": tensor lives on ", t.device(), "; fthmc_hip runs on the MI355X only (no CPU fallback)");
    TORCH_CHECK(t.scalar_type() == at::kDouble, name, ": dtype ", t.scalar_type(), "; the HIP path computes in float64");
    return t.contiguous();
}
Tensor field(const Tensor& t, const char* name) {
    Tensor f = dev64(t, name);
    TORCH_CHECK(f.dim() == 4 && f.size(1) == 2 && f.size(2) == f.size(3), name, ": expected [B, 2, L, L], got ", f.sizes());
    TORCH_CHECK(f.size(2) % 4 == 0, name, ": L = ", f.size(2), " must be a multiple of 4 (stripe masks have period 4)");
    return f;
}
void ok(int rc, const char* what) {
    TORCH_CHECK(rc == FTHMC_OK, what, " failed: ", fthmc_strerror(rc), " (code ", rc, ") ", rc == FTHMC_ERR_LAUNCH ? fthmc_last_error() : "");
}

// the fthmc_arch_t of a call from the schema's integers (hidden = None: the reference default [8, 8])
struct Arch {
    fthmc_arch_t a;
    bool is_default;
    Arch(int64_t n_mix, IntList hidden, int64_t kernel_size) {
        a = fthmc_arch_t{};
        a.n_mix = (int)n_mix; a.kernel_size = (int)kernel_size;
        if (hidden.has_value()) {
            TORCH_CHECK(hidden->size() <= 8, "hidden: at most 8 hidden layers");
            a.n_hidden = (int)hidden->size();
            for (size_t i = 0; i < hidden->size(); ++i) a.hidden[i] = (int)(*hidden)[i];
        } else { a.n_hidden = 2; a.hidden[0] = 8; a.hidden[1] = 8; }
        is_default = a.n_hidden == 2 && a.hidden[0] == 8 && a.hidden[1] == 8 && a.kernel_size == 3 && a.n_mix == 2;
    }
    const fthmc_arch_t* ptr() const { return is_default ? nullptr : &a; }
    int64_t params() const { const int p = fthmc_arch_params(ptr()); TORCH_CHECK(p > 0, "net shape beyond the limits of the HIP kernels"); return p; }
};

struct Workspace {
    Tensor buf; size_t bytes;
    Workspace(const Tensor& like, const fthmc_arch_t* arch, int B, int L, int nl, bool train = false) {
        bytes = train ? fthmc_train_ws_bytes(arch, B, L, nl) : fthmc_ws_bytes(arch, B, L, nl);
        TORCH_CHECK(bytes > 0, "unsupported shape (B = ", B, ", L = ", L, ", n_layers = ", nl, ")");
        buf = at::empty({(int64_t)((bytes + 7) / 8)}, like.options());
    }
    void* ptr() { return buf.mutable_data_ptr<double>(); }
};

Tensor weights(const Tensor& w, int64_t expect, const char* name) {
    Tensor f = dev64(w, name).reshape({-1});
    TORCH_CHECK(f.numel() == expect, name, ": expected ", expect, " doubles for this net shape, got ", f.numel());
    return f;
}
Tensor perchain(const Tensor& t, int64_t B, const char* name) {
    Tensor f = dev64(t, name).reshape({-1});
    TORCH_CHECK(f.numel() == B, name, ": expected ", B, " entries, got ", f.numel());
    return f;
}

// ---------------------------------------------------------------- Wilson
std::tuple<Tensor, Tensor, Tensor> wilson_action_charge(const Tensor& x_, double beta) {
    FT_DEVICE_GUARD(x_);
    Tensor x = field(x_, "x");
    const int B = (int)x.size(0), L = (int)x.size(2);
    Tensor S = at::empty({B}, x.options()), Q = at::empty({B}, x.options()), plaq = at::empty({B}, x.options());
    ok(fthmc_wilson_action_charge(cp(x), B, L, beta, mp(S), mp(Q), mp(plaq), cur_stream(x)), "fthmc_wilson_action_charge");
    return {S, Q, plaq};
}
Tensor wilson_force(const Tensor& x_, double beta) {
    FT_DEVICE_GUARD(x_);
    Tensor x = field(x_, "x");
    Tensor F = at::empty_like(x);
    ok(fthmc_wilson_force(cp(x), (int)x.size(0), (int)x.size(2), beta, mp(F), cur_stream(x)), "fthmc_wilson_force");
    return F;
}
std::tuple<Tensor, Tensor, Tensor> hmc_trajectory(const Tensor& x_, const Tensor& v_, const Tensor& u_, double beta, double dt, int64_t nstep) {
    FT_DEVICE_GUARD(x_);
    Tensor x = field(x_, "x"), v = field(v_, "v");
    const int B = (int)x.size(0), L = (int)x.size(2);
    Tensor u = perchain(u_, B, "u");
    TORCH_CHECK(v.sizes() == x.sizes(), "v: shaped like x expected");
    Tensor xn = at::empty_like(x), dH = at::empty({B}, x.options()), acc = at::empty({B}, x.options());
    Workspace ws(x, nullptr, B, L, 0);
    ok(fthmc_hmc_trajectory(cp(x), cp(v), cp(u), B, L, beta, dt, (int)nstep, mp(xn), mp(dH), mp(acc), nullptr, nullptr, ws.ptr(), ws.bytes,
                            cur_stream(x)), "fthmc_hmc_trajectory");
    return {xn, dH, acc};
}

// ---------------------------------------------------------------- one coupling layer
std::tuple<Tensor, Tensor> flow_layer_fwd(const Tensor& x_, const Tensor& w_, int64_t mu, int64_t off, int64_t n_mix, int64_t act,
                                          IntList hidden, int64_t kernel_size) {
    FT_DEVICE_GUARD(x_);
    Tensor x = field(x_, "x");
    Arch A(n_mix, hidden, kernel_size);
    Tensor w = weights(w_, A.params(), "w");
    const int B = (int)x.size(0), L = (int)x.size(2);
    Tensor y = at::empty_like(x), logJ = at::empty({B}, x.options());
    Workspace ws(x, A.ptr(), B, L, 1);
    ok(fthmc_flow_layer_fwd(cp(x), cp(w), A.ptr(), B, L, (int)mu, (int)off, (int)act, mp(y), mp(logJ), ws.ptr(), ws.bytes, cur_stream(x)),
       "fthmc_flow_layer_fwd");
    return {y, logJ};
}
std::tuple<Tensor, Tensor> layer_bwd(const Tensor& x_, const Tensor& gy_, const Tensor& glogJ_, const Tensor& w_, int64_t mu, int64_t off,
                                     int64_t n_mix, int64_t act, IntList hidden, int64_t kernel_size, bool need_gw) {
    FT_DEVICE_GUARD(x_);
    Tensor x = field(x_, "x"), gy = field(gy_, "gy");
    Arch A(n_mix, hidden, kernel_size);
    Tensor w = weights(w_, A.params(), "w");
    const int B = (int)x.size(0), L = (int)x.size(2);
    Tensor glogJ = perchain(glogJ_, B, "glogJ");
    TORCH_CHECK(gy.sizes() == x.sizes(), "gy: shaped like x expected");
    Tensor gx = at::empty_like(x), gw = need_gw ? at::empty({w.numel()}, x.options()) : Tensor();
    Workspace ws(x, A.ptr(), B, L, 1, need_gw);
    ok(fthmc_flow_layer_bwd(cp(x), cp(w), A.ptr(), cp(gy), cp(glogJ), B, L, (int)mu, (int)off, (int)act, mp(gx), need_gw ? mp(gw) : nullptr,
                            ws.ptr(), ws.bytes, cur_stream(x)), "fthmc_flow_layer_bwd");
    return {gx, gw};
}
Tensor flow_layer_bwd_x(const Tensor& x, const Tensor& gy, const Tensor& glogJ, const Tensor& w, int64_t mu, int64_t off, int64_t n_mix,
                        int64_t act, IntList hidden, int64_t kernel_size) {
    return std::get<0>(layer_bwd(x, gy, glogJ, w, mu, off, n_mix, act, hidden, kernel_size, false));
}
Tensor flow_layer_bwd_w(const Tensor& x, const Tensor& gy, const Tensor& glogJ, const Tensor& w, int64_t mu, int64_t off, int64_t n_mix,
                        int64_t act, IntList hidden, int64_t kernel_size) {
    return std::get<1>(layer_bwd(x, gy, glogJ, w, mu, off, n_mix, act, hidden, kernel_size, true));
}
std::tuple<Tensor, Tensor> flow_layer_bwd(const Tensor& x, const Tensor& gy, const Tensor& glogJ, const Tensor& w, int64_t mu, int64_t off,
                                          int64_t n_mix, int64_t act, IntList hidden, int64_t kernel_size) {
    return layer_bwd(x, gy, glogJ, w, mu, off, n_mix, act, hidden, kernel_size, true);
}
std::tuple<Tensor, Tensor> flow_layer_rev(const Tensor& y_, const Tensor& w_, int64_t mu, int64_t off, int64_t n_mix, int64_t act, double tol,
                                          IntList hidden, int64_t kernel_size) {
    FT_DEVICE_GUARD(y_);
    Tensor y = field(y_, "y");
    Arch A(n_mix, hidden, kernel_size);
    Tensor w = weights(w_, A.params(), "w");
    const int B = (int)y.size(0), L = (int)y.size(2);
    Tensor x = at::empty_like(y), logJ = at::empty({B}, y.options());
    Workspace ws(y, A.ptr(), B, L, 1);
    ok(fthmc_flow_layer_rev(cp(y), cp(w), A.ptr(), B, L, (int)mu, (int)off, (int)act, tol, mp(x), mp(logJ), ws.ptr(), ws.bytes, cur_stream(y)),
       "fthmc_flow_layer_rev");
    return {x, logJ};
}

// ---------------------------------------------------------------- whole flow
std::tuple<Tensor, Tensor, Tensor> ft_action_force(const Tensor& x_, const Tensor& w_all, int64_t n_layers, double beta, int64_t act,
                                                   int64_t n_mix, IntList hidden, int64_t kernel_size) {
    FT_DEVICE_GUARD(x_);
    Tensor x = field(x_, "x");
    Arch A(n_mix, hidden, kernel_size);
    Tensor w = weights(w_all, n_layers * A.params(), "w_all");
    const int B = (int)x.size(0), L = (int)x.size(2), nl = (int)n_layers;
    Tensor S = at::empty({B}, x.options()), logdet = at::empty({B}, x.options()), F = at::empty_like(x);
    Workspace ws(x, A.ptr(), B, L, nl);
    ok(fthmc_ft_action(cp(x), cp(w), A.ptr(), nl, B, L, (int)act, beta, mp(S), mp(logdet), nullptr, nullptr, ws.ptr(), ws.bytes, cur_stream(x)),
       "fthmc_ft_action");
    ok(fthmc_ft_force(cp(x), cp(w), A.ptr(), nl, B, L, (int)act, beta, mp(F), ws.ptr(), ws.bytes, cur_stream(x)), "fthmc_ft_force");
    return {S, logdet, F};
}
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> fthmc_trajectory(const Tensor& x_, const Tensor& v_, const Tensor& u_, const Tensor& w_all,
                                                                    int64_t n_layers, double beta, double dt, int64_t nstep, int64_t mode,
                                                                    int64_t act, int64_t n_mix, IntList hidden, int64_t kernel_size) {
    FT_DEVICE_GUARD(x_);
    TORCH_CHECK(mode == FTHMC_MODE_MD || mode == FTHMC_MODE_LITERAL, "mode: expected 0 (md) or 1 (literal), got ", mode);
    Tensor x = field(x_, "x"), v = field(v_, "v");
    Arch A(n_mix, hidden, kernel_size);
    Tensor w = weights(w_all, n_layers * A.params(), "w_all");
    const int B = (int)x.size(0), L = (int)x.size(2), nl = (int)n_layers;
    Tensor u = perchain(u_, B, "u");
    TORCH_CHECK(v.sizes() == x.sizes(), "v: shaped like x expected");
    auto o = x.options();
    Tensor xn = at::empty_like(x), dH = at::empty({B}, o), acc = at::empty({B}, o), plaq = at::empty({B}, o), Q = at::empty({B}, o);
    Workspace ws(x, A.ptr(), B, L, nl);
    ok(fthmc_ft_trajectory(cp(x), cp(v), cp(u), cp(w), A.ptr(), nl, B, L, (int)act, beta, dt, (int)nstep, (int)mode, mp(xn), mp(dH), mp(acc),
                           nullptr, nullptr, mp(plaq), mp(Q), nullptr, nullptr, ws.ptr(), ws.bytes, cur_stream(x)), "fthmc_ft_trajectory");
    return {xn, dH, acc, plaq, Q};
}
std::tuple<Tensor, Tensor, Tensor, Tensor> train_grad(const Tensor& xi_, const Tensor& w_all, int64_t n_layers, double beta, int64_t act,
                                                      int64_t n_mix, IntList hidden, int64_t kernel_size) {
    FT_DEVICE_GUARD(xi_);
    Tensor xi = field(xi_, "xi");
    Arch A(n_mix, hidden, kernel_size);
    Tensor w = weights(w_all, n_layers * A.params(), "w_all");
    const int B = (int)xi.size(0), L = (int)xi.size(2), nl = (int)n_layers;
    auto o = xi.options();
    Tensor x = at::empty_like(xi), logq = at::empty({B}, o), logp = at::empty({B}, o), gw = at::empty({w.numel()}, o);
    Workspace ws(xi, A.ptr(), B, L, nl, true);
    ok(fthmc_train_grad(cp(xi), cp(w), A.ptr(), nl, B, L, (int)act, beta, mp(x), mp(logq), mp(logp), mp(gw), ws.ptr(), ws.bytes, cur_stream(xi)),
       "fthmc_train_grad");
    return {x, logq, logp, gw};
}

}  // namespace

TORCH_LIBRARY(fthmc_hip, m) {
    m.def("wilson_action_charge(Tensor x, float beta) -> (Tensor, Tensor, Tensor)");
    m.def("wilson_force(Tensor x, float beta) -> Tensor");
    m.def("hmc_trajectory(Tensor x, Tensor v, Tensor u, float beta, float dt, int nstep) -> (Tensor, Tensor, Tensor)");
    m.def("flow_layer_fwd(Tensor x, Tensor w, int mu, int off, int n_mix, int act, int[]? hidden=None, int kernel_size=3) -> (Tensor, Tensor)");
    m.def("flow_layer_bwd_x(Tensor x, Tensor gy, Tensor glogJ, Tensor w, int mu, int off, int n_mix, int act, int[]? hidden=None, int kernel_size=3) -> Tensor");
    m.def("flow_layer_bwd_w(Tensor x, Tensor gy, Tensor glogJ, Tensor w, int mu, int off, int n_mix, int act, int[]? hidden=None, int kernel_size=3) -> Tensor");
    m.def("flow_layer_bwd(Tensor x, Tensor gy, Tensor glogJ, Tensor w, int mu, int off, int n_mix, int act, int[]? hidden=None, int kernel_size=3) -> (Tensor, Tensor)");
    m.def("flow_layer_rev(Tensor y, Tensor w, int mu, int off, int n_mix, int act, float tol, int[]? hidden=None, int kernel_size=3) -> (Tensor, Tensor)");
    m.def("ft_action_force(Tensor x, Tensor w_all, int n_layers, float beta, int act, int n_mix=2, int[]? hidden=None, int kernel_size=3) -> (Tensor, Tensor, Tensor)");
    m.def("fthmc_trajectory(Tensor x, Tensor v, Tensor u, Tensor w_all, int n_layers, float beta, float dt, int nstep, int mode, int act, int n_mix=2, int[]? hidden=None, int kernel_size=3) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("train_grad(Tensor xi, Tensor w_all, int n_layers, float beta, int act, int n_mix=2, int[]? hidden=None, int kernel_size=3) -> (Tensor, Tensor, Tensor, Tensor)");
}

// "CUDA" is the dispatch key of HIP devices in PyTorch-ROCm
TORCH_LIBRARY_IMPL(fthmc_hip, CUDA, m) {
    m.impl("wilson_action_charge", &wilson_action_charge);
    m.impl("wilson_force", &wilson_force);
    m.impl("hmc_trajectory", &hmc_trajectory);
    m.impl("flow_layer_fwd", &flow_layer_fwd);
    m.impl("flow_layer_bwd_x", &flow_layer_bwd_x);
    m.impl("flow_layer_bwd_w", &flow_layer_bwd_w);
    m.impl("flow_layer_bwd", &flow_layer_bwd);
    m.impl("flow_layer_rev", &flow_layer_rev);
    m.impl("ft_action_force", &ft_action_force);
    m.impl("fthmc_trajectory", &fthmc_trajectory);
    m.impl("train_grad", &train_grad);
}
