// Building blocks shared by the MFMA coupling-layer kernels (flow_fwd.hip, flow_bwd_gather.hip):
// tile geometry, the implicit-GEMM conv stage, the XCD-aware block map.
#pragma once
#include "flow_common.h"

// These kernels are written for ONE chip: 160 KB of LDS per CU (their static LDS exceeds the 64 KB of other gfx9
// parts), v_mfma_f64_16x16x4_f64, wave64.  Anything else would fail late, at launch, with an opaque error.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the fthmc MFMA kernels target gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

// The MFMA kernels feed every matrix instruction from LDS with two 8-byte reads per lane.  hipcc would pair such reads into
// ds_read2_b64, which the LDS serves at 128 B/clk instead of the 256 B/clk of ds_read_b64 (MI355X_MICROARCH.md, LDS table):
// the merging pass is off for these kernels (a per-function target feature, device pass only: as a command-line feature it
// made the host pass print "not a recognized feature" once per file); the IR-level vectorizer is off through the Makefile
// (-mllvm -amdgpu-load-store-vectorizer=0).  Explicit 16-byte accesses in the source are unaffected.
#if defined(__HIP_DEVICE_COMPILE__)
#define FT_LDS_B64 __attribute__((target("no-load-store-opt")))
#else
#define FT_LDS_B64
#endif

// accumulator chains of the MFMA stages (0 = default rule); a build-time knob for A/B runs
#ifndef FT_NCH
#define FT_NCH 0
#endif

// 1: the backward kernel recomputes act'(z1) (conv1 on the matrix cores over the stashed cos / sin of the frozen
// plaquettes, on the pair map of conv2^T so that the values land in the registers that multiply by them, + one sigmoid
// per value) instead of reading it back: the forward kernel then writes 9 (training: 25) instead of 17 (33) doubles per
// site and layer.  0 (default): act'(z1) travels through the stash.  Measured in round 3 (tools/abn.sh, config 3, one
// device, alternating): forward 26.2 -> 24.9 us per 64-chain launch, backward 19.8 -> 22.6 us (full batch 35.1 -> 41.8),
// trajectory 6.375 -> 6.578 ms: the backward is not as idle as its load phase suggests, the 33 MFMAs + 11 sigmoid
// epilogues per workgroup cost more than the forward saves.  Kept as a build switch (make EXTRA=-DFT_RECOMP_D1=1).
#ifndef FT_RECOMP_D1
#define FT_RECOMP_D1 0
#endif

// 1 (default): the act'(z1) plane of a mu = 0 layer is stored TRANSPOSED ([j][i][8]: site index j L + i).  Both kernels walk that
// plane down the stripe lines -- conv1 and conv2^T pair their output sites ACROSS the lines, so the 16 lanes of a tile are 16
// positions ALONG a line: 16 consecutive rows for mu = 0 --, and row-major storage made every one of those 16-byte accesses its own
// 64-byte piece of a line 4 KB from the next (16 half lines per wave instruction where mu = 1 touches 8 whole ones): the mu = 0
// instances were 4 % (forward) and 10 % (backward) slower than the mu = 1 ones, all of it in load / store issue.
#ifndef FT_D1_T
#define FT_D1_T 1
#endif
// The forward's stash stores carry the non-temporal hint (flow_fwd.hip launch_fwd: the SWEEP = 5, 6 instances) where the backward
// will not find the stash in a cache again.  2 (default): one layer's stash is FT_NT_MIN_BYTES or more, OR the layers behind this
// one write FT_STASH_FAR_BYTES (kernels.h) or more before the backward comes back to it (FlowLayerArgs::stash_far: at the headline
// shape every layer but the last two); 1: the first rule only; 3: the second only; 0: never (A/B knob)
#ifndef FT_NT_STASH
#define FT_NT_STASH 2
#endif
#ifndef FT_NT_MIN_BYTES
#define FT_NT_MIN_BYTES ((size_t)128 << 20)
#endif
// 1: the training backward's stash LOADS carry the hint too; 2: also k_flow_bwd_gather's (A/B knob)
#ifndef FT_NT_LOAD
#define FT_NT_LOAD 0
#endif
// 1 (default): the act'(z2) plane holds the live stripe lines only, in order (stash_live_idx).  conv3 reads h2 within one site of
// an active line, so every fourth line (x = off + 2 mod 4) is dead: never written, never used.  Row-major storage left the dead
// COLUMNS of a mu = 0 layer inside the cache lines the backward fetches (record = 64 B, line = 128 B: a window row cost 10 lines
// for 15 live records; per-kernel counters: 32 % more L2 read requests and 11 % more HBM reads than the mu = 1 instance, whose
// dead ROWS are skipped whole).  Compact, both instances read the same from HBM; the backward's mu = 0 / mu = 1 gap went from 8.5 % to 5.6 %.
#ifndef FT_D2_C
#define FT_D2_C 1
#endif

namespace fthmc_flow {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int NT = 512;                 // threads per workgroup (8 waves)
constexpr int NW = NT / 64;

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the wave's global loads
// and stores (s_waitcnt vmcnt(0)), which would stall every stage boundary on HBM round trips of the
// stash stores / prefetched windows.  No thread of these kernels reads global data another thread wrote.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Integer index math of these kernels is written for the full-rate VALU ops: v_mul_hi_u32 / v_mul_lo_u32 /
// v_mad_u64_u32 run at quarter rate, and the per-thread index set-up ahead of the first loads used to cost
// more cycles than the HBM round trip it precedes.
//   mul24: operands < 2^23 (lattice extents, window indices)
//   fdiv<D>(u) = u / D for 0 <= u < 32768, 2 <= D <= 32, via one 24-bit multiply (exact: u (M D - 2^20) < 2^20)
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
template <int D> __device__ __forceinline__ int fdiv(int u) {
    static_assert(D >= 1 && D <= 32, "small divisors");
    if (D == 1) return u;
    if ((D & (D - 1)) == 0) return (int)((unsigned)u >> __builtin_ctz(D));
    constexpr unsigned M = ((1u << 20) + D - 1) / D;
    return (int)(__umul24((unsigned)u, M) >> 20);
}
// base + (a launch- or wave-uniform element offset forced into SGPRs): loads through the result take the
// scalar-base form (global_load v, v_offset32, s[base]), their per-lane address is one 32-bit offset and
// costs no 64-bit VALU add.  The kernel-argument base keeps the pointer's global address space.
__device__ __forceinline__ size_t uniform_u64(size_t v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((size_t)hi << 32) | lo;
}
template <class T> __device__ __forceinline__ T* uniform_ptr(T* base, size_t off) { return base + uniform_u64(off); }
// the same with a 32-bit element offset (the launchers refuse shapes whose stash exceeds 2^32 doubles): offsets of a chain's
// planes are then 32-bit scalar products, where size_t arithmetic costs a 64-bit multiply (s_mul_i32 + s_mul_hi + adds) per base
template <class T> __device__ __forceinline__ T* uniform_at(T* base, unsigned off) {
    return base + (size_t)(unsigned)__builtin_amdgcn_readfirstlane(off);
}
// element `idx` (32-bit, idx * 8 < 2^32) of a uniform base: the byte offset is formed in 32 bits so that the
// load can be base-in-SGPRs + one VGPR offset
#define FT_G __attribute__((address_space(1)))                    // global memory, said in the pointer type (see flow_small.hip: gld / gst)
__device__ __forceinline__ double ldu(const double* base, unsigned idx) {
    return *(const FT_G double*)((const FT_G char*)base + idx * 8u);
}
// the same for an index that comes out of a branch (idle lanes read element 0): hipcc otherwise carries the ZERO-EXTENDED offset
// through the join and forms a 64-bit VGPR address per load (v_mov 0 + v_lshl_add_u64); the empty asm pins the 32-bit byte offset
// behind the join, the load keeps its base-in-SGPRs form
__device__ __forceinline__ unsigned ft_off32(unsigned idx) {
    unsigned o = idx * 8u;
    asm("" : "+v"(o));
    return o;
}
__device__ __forceinline__ double ldu_j(const double* base, unsigned idx) {
    return *(const FT_G double*)((const FT_G char*)base + ft_off32(idx));
}
// stores through a uniform base + a 32-bit element index: the address is base-in-SGPRs + one VGPR offset (one shift per
// store); through a per-lane 64-bit pointer every stash store cost a sign extension, a 64-bit shift and a 64-bit add
typedef double double2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stu(double* base, unsigned idx, double v) {
    *(FT_G double*)((FT_G char*)base + idx * 8u) = v;
}
__device__ __forceinline__ void stu2(double* base, unsigned idx, double2u_t v) {
    *(FT_G double2u_t*)((FT_G char*)base + idx * 8u) = v;
}
// ... and the same with the non-temporal hint (NTS, a template flag of the caller's instance): the stash of a TRAINING sweep is
// 280 bytes per site and layer that nobody reads before the whole sweep has gone by (FT_NT_STASH, flow_fwd.hip)
template <bool NTS> __device__ __forceinline__ void sts(double* base, unsigned idx, double v) {
    FT_G double* p = (FT_G double*)((FT_G char*)base + idx * 8u);
    if (NTS) __builtin_nontemporal_store(v, p); else *p = v;
}
template <bool NTS> __device__ __forceinline__ void sts2(double* base, unsigned idx, double2u_t v) {
    FT_G double2u_t* p = (FT_G double2u_t*)((FT_G char*)base + idx * 8u);
    if (NTS) __builtin_nontemporal_store(v, p); else *p = v;
}
// Wrapped lattice coordinate (v mod L) of window line v, -L <= v.  FAST (L exceeds the window by a
// margin, chosen at launch): the line wraps at most once, two selects.  Otherwise (small test lattices)
// division by the launch-uniform L through its reciprocal (wrap_magic): n M >> 32 = floor(n / L) for
// n, L < 2^16 with M = floor(2^32 / L) + 1.
__device__ __forceinline__ unsigned wrap_magic(int L) { return 0xFFFFFFFFu / (unsigned)L + 1u; }
// POW2 (L a power of two: 64, 128, 256 -- the EXACT instances of the coupling kernels; `magic` then carries L - 1): one v_and.
template <bool FAST, bool POW2 = false>
__device__ __forceinline__ int wrap_line(int v, int L, unsigned magic) {
    if (POW2) return (int)((unsigned)v & magic);
    if (FAST) return (int)min(min((unsigned)v, (unsigned)(v - L)), (unsigned)(v + L));   // -L <= v < 2 L: one v_min3_u32
    const unsigned nn = (unsigned)(v + L);
    return (int)(nn - (unsigned)L * __umulhi(nn, magic));
}
// a lattice at least this much larger than the tile takes the FAST wrap (window lines reach 4 sites out)
inline bool wrap_fast_ok(int L, int tr, int tc) { return L >= tr + 8 && L >= tc + 8; }

// Plane stride (doubles): smallest value >= n that is = 18 (mod 32).  The four k-lanes of an
// A read (ds_read_b64, 64 banks) then overlap in only 2 of 32 doubles, and the eight channel
// lanes of an epilogue ds_write_b64 (32 banks = 16 doubles) land on eight different bank pairs
// (with = 16 (mod 32) they would all hit the same pair: 8-way conflict on every store).
constexpr int ps_round(int n) { return ((n - 18 + 31) / 32) * 32 + 18; }
// Plane stride of the planes an MFMA stage reads its activation operand from: = 16 (mod 32).  A ds_read_b64 is banked
// over 32 lanes = 16 pairs x two K lane groups g, g + 1 (one plane apart): consecutive pair sites + 16 fill the 32
// double-wide banks exactly (with 18 two lanes collide and the read takes a third LDS cycle).
#ifndef FT_PSMOD
#define FT_PSMOD 16
#endif
constexpr int ps_round16(int n) { return ((n - FT_PSMOD + 31) / 32) * 32 + FT_PSMOD; }

template <int TR, int TC> struct Geom {
    static constexpr int R0R = TR + 6, R0C = TC + 6, N0 = R0R * R0C;   // plaquette / input window
    static constexpr int R1R = TR + 4, R1C = TC + 4, N1 = R1R * R1C;   // h1 window
    // LDS row stride of the h1 planes: odd, so that a column of sites (the conv1 epilogue's stores for mu = 0 run down one:
    // 16 consecutive rows per MFMA tile) spreads over 16 bank pairs; with the window width (20) they collapsed onto 8
    static constexpr int RS1 = R1C + 1;
    static constexpr int R2R = TR + 2, R2C = TC + 2, N2 = R2R * R2C;   // h2 window
    static constexpr int N3 = TR * TC, NA = N3 / 4;                    // tile, active sites
    static constexpr int NAS = NA <= 32 ? 32 : 64;                     // lane stride of per-active-site scratch
    static constexpr int PS0 = ps_round16(N0), PS1 = ps_round16(R1R * RS1), PS2 = ps_round(N2);   // net input, h1: MFMA operands
    static_assert(PS0 % 32 == FT_PSMOD && PS1 % 32 == FT_PSMOD && PS2 % 32 == 18, "bank layout");
    static_assert(PS0 >= N0 && PS1 >= R1R * RS1 && PS2 >= N2, "plane size");
    static_assert(TR % 4 == 0 && TC % 4 == 0 && NA <= 64, "tile shape");
};


// Per-layer activation stash written by the forward kernel and read back by the gather-form backward
// (n = L * L; per chain b):
//   d1  [n][8]     act'(z1), channel-minor (mu = 0 layers: site index TRANSPOSED, j L + i: FT_D1_T)
//   d2  [3n/4][8]  act'(z2), channel-minor, the LIVE stripe lines only, compact (FT_D2_C, stash_live_idx; the tiled kernels --
//                  k_ft_small keeps [n][8] with its dead lines unwritten; the plane is n records either way): a window row of a
//                  mu = 0 layer is 15 consecutive records, not 20 of which every fourth is fetched with its cache line and dropped
//                  (64 B per site: a window row of 20 sites is 10 cache lines for all channels, not 8 x 2..3,
//                   and the channel pair (2 g, 2 g + 1) of a lane is one 16-byte access)
//   tc  [K][n/4][4] adjoint coefficients of the tan-mixture transform at the ACTIVE sites, compact, component-major:
//                  per mixture component k the four values A_k, B_k, C_k, E_k.  With g = upstream dL/d delta,
//                  cb = dL/dlogJ and the softmax normaliser rs = 1 / (K sum_k C_k)  (C_k = 1 / (K D_k)):
//                  dL/ds_k = g A_k + cb rs B_k,   dL/dP = -g + sum_k (g C_k - cb rs E_k)
//   cs  [2][n/2]   cos P, sin P at the FROZEN sites (the net input), compact
//   h1, h2 [n][8]  hidden activations, channel-minor (training only)
// = 16 + 2 + 1 = 19 doubles per site and layer (35 with h1, h2): kernels.h flow_stash_doubles().
struct Stash { double *d1, *d2, *tc, *cs, *h1, *h2; };
__device__ __forceinline__ Stash stash_view(double* base, int B, int b, int n) {
    const unsigned bn = (unsigned)b * (unsigned)n, Bn = (unsigned)B * (unsigned)n;     // 35 B n < 2^32: checked by the launchers
    Stash v;
    v.d1 = base + (size_t)(8u * bn);
    v.d2 = base + (size_t)(8u * (Bn + bn));
    v.tc = base + (size_t)(16u * Bn + 2u * bn);
    v.cs = base + (size_t)(18u * Bn + bn);
    v.h1 = base + (size_t)(19u * Bn + 8u * bn);
    v.h2 = base + (size_t)(27u * Bn + 8u * bn);
    return v;
}
// compact index of an active site (i, j): every 4th column (mu = 0) or row (mu = 1)
__device__ __forceinline__ int stash_active_idx(int i, int j, int L, int mu) {
    // unsigned 24-bit multiply-adds (coordinates are lattice sites): the signed form came out as v_bfe_i32 + the quarter-rate v_mad_u64_u32
    return (int)(mu == 0 ? __umul24((unsigned)i, (unsigned)(L >> 2)) + (unsigned)(j >> 2) : __umul24((unsigned)(i >> 2), (unsigned)L) + (unsigned)j);
}
// act'(z2) plane: index of site (i, j) when only the live stripe lines are stored (FT_D2_C): conv3 reads h2 within one site of an
// active line, so the lines x = off + 2 (mod 4) are never written nor used; live lines in order, 3 of every 4
template <bool POW2> __device__ __forceinline__ int stash_live_line(int x, int L, int off) {     // compact index of stripe line x
    int u = x - off - 3;                                  // (x - off - 3) mod L: class 0, 1, 2 live (off - 1, off, off + 1), 3 dead
    if (POW2) u &= L - 1;
    else { u += u < 0 ? L : 0; u += u < 0 ? L : 0; }      // twice: x - off - 3 >= -6 and L may be 4
    return mul24(u >> 2, 3) + (u & 3);                    // 24-bit multiply-add (a plain one becomes the quarter-rate v_mad_u64_u32)
}
template <bool POW2> __device__ __forceinline__ int stash_live_idx(int i, int j, int L, int mu, int off) {
    const int cx = stash_live_line<POW2>(mu == 0 ? j : i, L, off);
    return mu == 0 ? mul24(i, 3 * (L >> 2)) + cx : mul24(cx, L) + j;
}
// compact index of a frozen site (stripe classes 1, 2 of its line): two of every 4 columns / rows
__device__ __forceinline__ int stash_frozen_idx(int i, int j, int L, int mu, int off) {
    int u = (mu == 0 ? j : i) - off - 1;
    if (u < 0) u += L;
    const int f = 2 * (u >> 2) + (u & 1);
    return mu == 0 ? mul24(i, L >> 1) + f : mul24(f, L) + j;
}

// LDS copy of the layer's weight block (flow_common.h: LF_* forward, LB_* backward)
// One implicit-GEMM stage on v_mfma_f64_16x16x4_f64 over NPAIR "pair sites": a pair is two adjacent
// output sites (rows r, r + 1 of one column, or columns c, c + 1 of one row) that share a 4 x 3 (3 x 4)
// input window, so N = 16 = 8 output channels x the 2 sites of the pair and K = 12 window taps x KC input
// channels (9 of the 12 taps are non-zero for each site: 75 % useful MACs instead of the 50 % of a half-empty N).
// The weights are the MFMA's A operand and the activations its B operand, so D comes out transposed:
// D[row = (channel, site of pair)][col = pair].  A lane (g = lane >> 4, i = lane & 15) then holds one pair
// (col = i) and the rows g + 4 q: all site arithmetic of the epilogue (offsets, bounds, stash address) happens
// once per lane and tile instead of once per value.
// Operand addressing: K step t of lane group g reads
//     activation  A[amap(pair) + KO::alane(g) + KO::aimm(t)]      weight  W[wl + KO::bimm(t)]
// with aimm / bimm compile-time constants (they become the offset field of the ds_read) and wl, the lane part of
// the weight index, computed once per stage by the caller from (g, cN = i & 7, dd = i >> 3) against a zero-padded
// weight table (flow_common.h), so the K loop holds no VALU instruction at all.
//   epi(g, p, ok, z, it): z[q] = output channel 2 g + (q & 1) (= ft_chan of MFMA row g + 4 (q & 1), flow_common.h)
//               at site (q >> 1) of pair p; all four values
//               of a lane at once so that their chains interleave; it = which of the wave's tiles (a
//               compile-time constant with UNROLL, so the caller can keep per-tile operands in registers)
// K orders (RSA / PSA: row / plane stride of the activation planes):
struct KConv2Row {      // 8 -> 8 channels, pairs = rows (r, r + 1): k = (tap = t >> 1 of the 4 x 3 window, ci = 4 (t & 1) + g)
    static constexpr int NSTEP = 24;
    template <int RSA, int PSA> static __device__ __forceinline__ int alane(int g) { return g * PSA; }
    template <int RSA, int PSA> static constexpr int aimm(int t) { return (t & 1) * 4 * PSA + ((t >> 1) / 3) * RSA + (t >> 1) % 3; }
    // table [kx][g][h][r5][cN]: W[cN][4 h + g][ky4 - dd][kx] sits at r5 = ky4 - dd + 1
    static constexpr int bimm(int t) { return ((t >> 1) % 3) * 320 + (t & 1) * 40 + ((t >> 1) / 3) * 8; }
    static __device__ __forceinline__ int wlane(int g, int cN, int dd) { return cN + g * 80 + (1 - dd) * 8; }
};
struct KConv2Col {      // pairs = columns (c, c + 1): 3 x 4 window, table [cN][ci][ky][c5]
    static constexpr int NSTEP = 24;
    template <int RSA, int PSA> static __device__ __forceinline__ int alane(int g) { return g * PSA; }
    template <int RSA, int PSA> static constexpr int aimm(int t) { return (t & 1) * 4 * PSA + ((t >> 1) / 4) * RSA + (t >> 1) % 4; }
    static constexpr int bimm(int t) { return ((t >> 1) / 4) * 320 + (t & 1) * 40 + ((t >> 1) % 4) * 8; }
    static __device__ __forceinline__ int wlane(int g, int cN, int dd) { return cN + g * 80 + (1 - dd) * 8; }
};
struct KConv1 {         // 2 -> 8 channels, pairs = rows: k = (ci = g & 1, tap = t + 6 (g >> 1)): taps 6.. are the window's rows 2, 3
    static constexpr int NSTEP = 6;
    template <int RSA, int PSA> static __device__ __forceinline__ int alane(int g) { return (g & 1) * PSA + (g >> 1) * 2 * RSA; }
    template <int RSA, int PSA> static constexpr int aimm(int t) { return (t / 3) * RSA + t % 3; }
    static constexpr int bimm(int t) { return (t % 3) * 96 + (t / 3) * 8; }       // table [kx][ci: 48][r5: 8][cN]
    static __device__ __forceinline__ int wlane(int g, int cN, int dd) { return cN + (g & 1) * 48 + (2 * (g >> 1) + 1 - dd) * 8; }
};
// Diagnostic builds (-DFT_DIAG) only: cycle stamp of wave 0 inside a stage, after everything the wave has issued is back
__device__ __forceinline__ void stampx(long long* slot) {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t_;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if ((threadIdx.x & 63) == 0) *slot = (long long)t_;          // callers pick the wave
}
// DEFER: the epilogues run after ALL of the workgroup's MFMA loops (results parked in registers, one workgroup
//        barrier in between), so that an epilogue may overwrite the planes the MFMAs read (needs UNROLL)
// NCHAIN: independent accumulator chains per tile (0: 4, or 3 for short K)
template <class KO, int NPAIR, int RSA, int PSA, bool UNROLL, bool DEFER, int NCHAIN, class AMap, class Epi>
// acc0: the accumulator's start value (the bias of this lane's rows: the epilogue then adds nothing)
__device__ __forceinline__ void mfma_stage(const double* __restrict__ A, const double* __restrict__ Wt,
                                           int wave, int lane, AMap amap, Epi epi, long long* dbg = nullptr,
                                           double4_t acc0 = double4_t{0.0, 0.0, 0.0, 0.0}) {
    constexpr int NSTEP = KO::NSTEP, NTILE = (NPAIR + 15) / 16;
    const int g = lane >> 4, i = lane & 15;
    const double* wp = Wt + KO::wlane(g, i & 7, i >> 3);
    const int al = KO::template alane<RSA, PSA>(g);
    constexpr int NIT = (NTILE + NW - 1) / NW, NUNR = UNROLL ? NIT : 1;
    static_assert(!DEFER || UNROLL, "deferred epilogues keep their tiles in registers");
    double zs[DEFER ? NIT : 1][4];
#pragma unroll NUNR
    for (int it = 0; it < NIT; ++it) {
        const int tile = wave + it * NW;
        if (tile >= NTILE) break;
        const int p_ = tile * 16 + i;
        const bool ok = p_ < NPAIR;
        const int p = ok ? p_ : NPAIR - 1;                   // padding lanes: any valid address
        const double* a0 = A + amap(p) + al;
        // independent accumulator chains keep the matrix pipe busy when a wave is alone on it
        constexpr int NCH = NCHAIN > 0 ? NCHAIN : (NSTEP >= 8 ? 4 : 3);
        double4_t accs[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) accs[ch] = ch == 0 ? acc0 : double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) {
            accs[t % NCH] = __builtin_amdgcn_mfma_f64_16x16x4f64(wp[KO::bimm(t)], a0[KO::template aimm<RSA, PSA>(t)], accs[t % NCH], 0, 0, 0);
#ifdef FT_MFMA_FENCE      // a scheduling fence every FT_MFMA_FENCE K steps: bounds how many operand reads are hoisted (register budget)
            if (t % FT_MFMA_FENCE == FT_MFMA_FENCE - 1) __builtin_amdgcn_sched_barrier(0);
#endif
        }
        double4_t acc = accs[0];
#pragma unroll
        for (int ch = 1; ch < NCH; ++ch) acc += accs[ch];
#ifdef FT_DIAG
        if (dbg && wave == 0) { asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3])); stampx(dbg + 2 * it); }
#endif
        if (DEFER) { zs[it][0] = acc[0]; zs[it][1] = acc[1]; zs[it][2] = acc[2]; zs[it][3] = acc[3]; }
        else { double z4[4] = {acc[0], acc[1], acc[2], acc[3]}; epi(g, p, ok, z4, it); }
#ifdef FT_DIAG
        if (dbg && wave == 0) stampx(dbg + 2 * it + 1);
#endif
    }
    if (DEFER) {
        lds_barrier();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int tile = wave + it * NW;
            if (tile < NTILE) {
                const int p_ = tile * 16 + i;
                const bool ok = p_ < NPAIR;
                epi(g, ok ? p_ : NPAIR - 1, ok, zs[it], it);
            }
        }
    }
}

// One conv2^T tile: the 18 live K steps of a pair window whose line KD (of its four lines across the pairing direction,
// NL = lines per tap row of KO's window) is dead.  Straight-line code per (K order, KD): the operand reads pipeline freely.
template <class KO, int NL, int KD, int RSA, int PSA>
__device__ __forceinline__ double4_t conv2t_tile(const double* __restrict__ wp, const double* __restrict__ a0) {
    double4_t accs[2] = {double4_t{0.0, 0.0, 0.0, 0.0}, double4_t{0.0, 0.0, 0.0, 0.0}};
    int s = 0;
#pragma unroll
    for (int t = 0; t < KO::NSTEP; ++t) {
        // tap t >> 1 of the window; the line across the pairing direction: its column (4-wide window) / its row (4-high window)
        const int line = NL == 4 ? (t >> 1) % 4 : (t >> 1) / 3;
        if (line == KD) continue;
        accs[s & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(wp[KO::bimm(t)], a0[KO::template aimm<RSA, PSA>(t)], accs[s & 1], 0, 0, 0);
        ++s;
    }
    return accs[0] + accs[1];
}

// conv3^T of the tiled backward kernels on the matrix cores (flow_bwd_gather.hip, flow_bwd_train.hip; weight table T3 and the
// idea: flow_common.h LT3_*).  Output = gz2 on the tile+2 window (W2 x W2 sites, LDS planes [8][PS2], row stride RS2) = act'(z2)
// times the sum over (co, taps) of g_out on the tile+3 window (planes [3][N3W], row stride W3C; written on the active lines only).
// Pairs across the stripe lines: position pu across (sites 2 pu, 2 pu + 1), v along.  A pair's four-line window starts at line
// 2 pu - 1; its active line is window line u = (a1 - 2 pu) & 3 (a1 = first active line of the tile+3 window: c0 / r0 of the
// kernels), i.e. tile+3 line 2 pu + u, and member dd sees it through the tap across ka = dd + 2 - u -- outside 0..2 the member
// sits on a DEAD line (no active site within reach): its weights are the table's zero lines, its act'(z2) was never stashed, the
// result is an exact 0.  The weights are the MFMA's A operand, shared by the 16 pairs of a tile: u must be the tile's, and it
// depends on the parity of pu only -- tiles are class-pure: class cls = pu & 1, NTC tiles per class, pair index inside the
// class with the coordinate that runs along a lattice ROW fastest (the act'(z2) records of a tile's lanes are then close in
// the stash).
template <int MU, int W2> struct Conv3T {
    static constexpr int NU = W2 / 2, NCL = NU / 2, NPC = NCL * W2, NTC = (NPC + 15) / 16, NTILE = 2 * NTC, NIT = (NTILE + NW - 1) / NW;
    static_assert(NU % 2 == 0, "as many even as odd pair positions");
    int pu, v, u;            // this lane's pair of the tile; u: window line of the active line (tile-uniform)
    bool ok;                 // a pair of the window (the last tile of a class is not full)
    __device__ __forceinline__ Conv3T(int tile, int lane, int a1) {
        const int cls = tile >= NTC ? 1 : 0, p_ = (tile - cls * NTC) * 16 + (lane & 15);
        ok = tile < NTILE && p_ < NPC;
        const int p = ok ? p_ : 0;
        int k;
        if (MU == 0) { v = fdiv<NCL>(p); k = p - v * NCL; } else { k = fdiv<W2>(p); v = p - k * W2; }
        pu = 2 * k + cls;
        u = (a1 - 2 * cls) & 3;
    }
    __device__ __forceinline__ bool dead(int dd) const { return dd ? u == 0 : u == 3; }
    // tile+2 coordinates of member dd
    __device__ __forceinline__ int row(int dd) const { return MU == 0 ? v : 2 * pu + dd; }
    __device__ __forceinline__ int col(int dd) const { return MU == 0 ? 2 * pu + dd : v; }
};
// one tile: 3 MFMAs; returns z[q] = (channel 2 g + (q & 1), member q >> 1) of the lane's pair, BEFORE the act'(z2) factor
template <int MU, int W2, int W3C, int N3W>
__device__ __forceinline__ double4_t conv3t_tile(const double* __restrict__ sGO, const double* __restrict__ sT3, const Conv3T<MU, W2>& P, int lane) {
    const int g = lane >> 4, i = lane & 15, cN = i & 7, dd = i >> 3;
    const double* wp = sT3 + g * LT3_CO + (dd + 3 - P.u) * 8 + cN;
    const int la3 = 2 * P.pu + P.u;                                        // the active line in tile+3 coordinates
    const double* a0 = sGO + (g < 3 ? g : 0) * N3W + (MU == 0 ? (P.v + 2) * W3C + la3 : la3 * W3C + P.v + 2);
    constexpr int astep = MU == 0 ? W3C : 1;                               // LDS step along the lines: tap t reads position v + 1 - t
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wp[t * LT3_T], a0[-t * astep], acc, 0, 0, 0);
    return acc;
}

// Live-line map of a window whose every 4th line (first one d0) is dead: index of the l-th live line.
// (the quotient by 3 through fdiv: a plain `/ 3` compiles to the quarter-rate v_mul_hi_u32)
__device__ __forceinline__ int live_line(int l, int d0) {
    const int e = l - d0, q = 1 + fdiv<3>(max(e, 0));                    // branch-free: one select
    return l + (e >= 0 ? q : 0);
}


// XCD-aware block -> (chain, tile) map.  Blocks are dealt round-robin over the 8 XCDs (chains b and b + 8
// share one), and each XCD has its own L2: all tiles of a chain go to the same XCD, consecutively,
// so the halo re-reads of neighbouring tiles (links, stashed activations) hit that XCD's L2
// instead of going out to the fabric once per XCD.  Speed only: any placement is correct.
struct BlockTile { int b, tile, ti, tj; };
// grid = (8 * ntj, nti, ceil(B / 8)): the linear block id the dispatcher deals round-robin over the 8 XCDs is
// x + gridDim.x * (y + gridDim.y * z) and gridDim.x is a multiple of 8, so XCD = blockIdx.x & 7; the rest
// of the coordinates come out of the block index without a division.
__device__ __forceinline__ bool block_tile(int B, int nti, int ntj, BlockTile& t, int zb = -1) {   // zb: the chain group when blockIdx.z carries more (layers)
    const int xcd = blockIdx.x & 7;
    t.tj = blockIdx.x >> 3;
    t.ti = blockIdx.y;
    t.tile = t.ti * ntj + t.tj;
    t.b = (zb < 0 ? (int)blockIdx.z : zb) * 8 + xcd;
    return t.b < B;
}
inline dim3 xcd_grid(int B, int nti, int ntj) { return dim3(8 * ntj, nti, (B + 7) / 8); }

}  // namespace fthmc_flow
