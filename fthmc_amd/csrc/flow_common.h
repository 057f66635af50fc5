// Shared constants / helpers of the coupling-layer kernels (flow.hip: VALU variant,
// flow_mfma.hip: MFMA variant).
#pragma once
#include "common.h"
#include "kernels.h"

namespace fthmc_flow {

using namespace fthmc;

constexpr int FT = FLOW_TILE;
constexpr int R0 = FT + 6, R1 = FT + 4, R2 = FT + 2;
constexpr int N0 = R0 * R0, N1 = R1 * R1, N2 = R2 * R2, N3 = FT * FT;
constexpr int NACT = N3 / 4;                  // active sites per tile (64 = one wave)
constexpr int NMIX = 2;

// canonical per-layer offsets (PyTorch [Cout][Cin][3][3])
constexpr int CW0 = 0, CB0 = 144, CW1 = 152, CB1 = 728, CW2 = 736, CB2 = 952;
// kernel layout offsets
constexpr int W1F = 0;      // [ci 2][tap 9][co 8]
constexpr int B1 = 144;     // [8]
constexpr int W2F = 152;    // [ci 8][tap 9][co 8]
constexpr int B2 = 728;     // [8]
constexpr int W3F = 736;    // [ci 8][tap 9][co 4] (co 3 = 0)
constexpr int B3 = 1024;    // [4]
constexpr int W3B = 1028;   // [co 3][tap 9][ci 8]
constexpr int W2B = 1244;   // [co 8][tap 9][ci 8]
constexpr int W1B = 1820;   // [co 8][tap 9][ci 2]
static_assert(W1B + 144 <= FLOW_WINT, "weight layout");

// MFMA B-operand packs (one double per lane per k-step; see flow_mfma.hip)
constexpr int WB1 = 1968;            // conv1   (K = 2 ci x 12 taps = 24)  ->  6 steps x 64 lanes
constexpr int WB2 = WB1 + 6 * 64;    // conv2   (K = 8 ci x 12 taps = 96)  -> 24 steps
constexpr int WB3T = WB2 + 24 * 64;  // conv3^T (K = 3 co x 12 taps = 36)  ->  9 steps
constexpr int WB2T = WB3T + 9 * 64;  // conv2^T (K = 8 co x 12 taps = 96)  -> 24 steps
static_assert(WB2T + 24 * 64 <= FLOW_WINT, "weight layout");

__device__ __forceinline__ void act_eval(double z, int act, double& h, double& d) {
    if (act == FTHMC_ACT_SILU) {
        const double sg = 1.0 / (1.0 + exp(-z));
        h = z * sg;
        d = sg * (1.0 + z * (1.0 - sg));
    } else if (act == FTHMC_ACT_RELU) {
        h = z > 0.0 ? z : 0.0;  d = z > 0.0 ? 1.0 : 0.0;
    } else {
        h = z > 0.0 ? z : 0.01 * z;  d = z > 0.0 ? 1.0 : 0.01;
    }
}


}  // namespace fthmc_flow
