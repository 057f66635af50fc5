// Counter-based momentum refresh: Philox4x32-10 keyed by a per-chain 64-bit seed,
// counter = element index.  A chain's draws depend only on its seed, never on
// how the batch is sharded over GPUs (SURVEY 7 "RNG").  Box-Muller in fp64.
#include "common.h"
#include "kernels.h"

namespace {

struct u4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u4 philox4x32_10(u4 c, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = u4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += W0; k1 += W1;
    }
    return c;
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {     // (0, 1]
    const uint64_t m = (((uint64_t)hi << 32) | lo) >> 11;
    return ((double)m + 1.0) * (1.0 / 9007199254740992.0);
}

// v[b][0..n) ~ N(0,1) (pairs from one Philox block), u[b] ~ U[0,1)
__global__ void k_random_momenta(const int64_t* __restrict__ seeds, int n, double* __restrict__ v,
                                 double* __restrict__ u) {
    const int b = blockIdx.y;
    const uint64_t sd = (uint64_t)seeds[b];
    const uint32_t k0 = (uint32_t)sd, k1 = (uint32_t)(sd >> 32);
    const int npair = (n + 1) / 2;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npair; p += gridDim.x * blockDim.x) {
        const u4 r = philox4x32_10(u4{(uint32_t)p, 0u, 0u, 0u}, k0, k1);
        const double u1 = u53(r.x, r.y), u2 = u53(r.z, r.w);
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs; sincos(FT_TWO_PI * u2, &sn, &cs);
        v[(size_t)b * n + 2 * p] = rad * cs;
        if (2 * p + 1 < n) v[(size_t)b * n + 2 * p + 1] = rad * sn;
    }
    if (u && blockIdx.x == 0 && threadIdx.x == 0) {
        const u4 r = philox4x32_10(u4{0u, 0u, 1u, 0u}, k0, k1);      // separate counter plane
        u[b] = 1.0 - u53(r.x, r.y);                                   // [0, 1)
    }
}

// out[b][0..n) ~ U[lo, hi): the prior draw of the reverse-KL training step (MultivariateUniform.sample_n,
// fthmc/utils/distributions.py:65-76), two values per Philox block, counter plane 2 (disjoint from the momenta's)
__global__ void k_random_uniform(const int64_t* __restrict__ seeds, int n, double lo, double hi, double* __restrict__ out) {
    const int b = blockIdx.y;
    const uint64_t sd = (uint64_t)seeds[b];
    const uint32_t k0 = (uint32_t)sd, k1 = (uint32_t)(sd >> 32);
    const int npair = (n + 1) / 2;
    const double w = hi - lo;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npair; p += gridDim.x * blockDim.x) {
        const u4 r = philox4x32_10(u4{(uint32_t)p, 0u, 2u, 0u}, k0, k1);
        out[(size_t)b * n + 2 * p] = fma(1.0 - u53(r.x, r.y), w, lo);             // 1 - (0, 1] = [0, 1)
        if (2 * p + 1 < n) out[(size_t)b * n + 2 * p + 1] = fma(1.0 - u53(r.z, r.w), w, lo);
    }
}

// seeds[b] = SplitMix64 mix of (seed, global chain id, trajectory) & (2^63 - 1): the arithmetic of parallel.chain_seeds.
// One workgroup: every thread reads the counter before thread 0 moves it on.
__global__ void k_chain_seeds(uint64_t seed, uint64_t lo, int B, uint64_t traj, int64_t* __restrict__ counter, int advance,
                              int64_t* __restrict__ seeds) {
    const uint64_t t = traj + (counter ? (uint64_t)*counter : 0ull);
    __syncthreads();
    const uint64_t base = (seed * 2ull + 1ull) * 0x2545F4914F6CDD1Dull + t * 0xBF58476D1CE4E5B9ull;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        uint64_t z = (lo + (uint64_t)b) * 0x9E3779B97F4A7C15ull + base;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        seeds[b] = (int64_t)(z & 0x7FFFFFFFFFFFFFFFull);
    }
    if (advance && threadIdx.x == 0) *counter = (int64_t)(t - traj + 1ull);
}

}  // namespace

namespace fthmc {
int launch_chain_seeds(int64_t seed, int64_t lo, int B, int64_t traj, int64_t* counter, int advance, int64_t* seeds, hipStream_t s) {
    hipLaunchKernelGGL(k_chain_seeds, dim3(1), dim3(256), 0, s, (uint64_t)seed, (uint64_t)lo, B, (uint64_t)traj, counter, advance, seeds);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_random_uniform(const int64_t* seeds, int B, int n, double lo, double hi, double* out, hipStream_t s) {
    int gx = ((n + 1) / 2 + 255) / 256; if (gx > 32) gx = 32; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_random_uniform, dim3(gx, B), dim3(256), 0, s, seeds, n, lo, hi, out);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
int launch_random_momenta(const int64_t* seeds, int B, int n, double* v, double* u, hipStream_t s) {
    int gx = ((n + 1) / 2 + 255) / 256; if (gx > 32) gx = 32; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_random_momenta, dim3(gx, B), dim3(256), 0, s, seeds, n, v, u);
    FT_LAUNCH_CHECK(); return FTHMC_OK;
}
}  // namespace fthmc
