// Weight-blob layout of one SCPA block of PAN (reference architectures/PAN_arch.py:58-105) as the MFMA A fragments the fused block kernels read from LDS:
// shared by csrc/pan_scpa.hip (fp16 arithmetic) and csrc/pan_scpa_split.hip (the fp32-accurate form on (hi, lo) operand pairs: two blobs of this layout).
#pragma once
namespace innfer {
namespace scpa {

constexpr int TW = 32, HC = TW + 4;
constexpr int r2(int t) { return t ? 8 : 12; }          // real rows of 16-row tile t of a 32-row (20-channel) panel: row rho <-> channel 8 (rho >> 2) + 4 t + (rho & 3)
constexpr int r4(int t) { return t < 2 ? 12 : 8; }      // ... of a 64-row (40-channel) panel: row rho <-> channel 16 (rho >> 2) + 4 t + (rho & 3)
// compact weight blob (bytes): only real rows, only k-octets that can be non-zero
constexpr int C1_ROW = 5 * 16, K_ROW = 3 * 16, C3_ROW = 6 * 16;
constexpr int OFF_C1A = 0, C1_T1 = 12 * C1_ROW, C1_SIZE = 20 * C1_ROW;            // conv1_a: [t][row][oct 0..3 = k-step 0, oct 4 = k 32..39]
constexpr int OFF_C1B = OFF_C1A + C1_SIZE;
constexpr int K_T1 = 12 * K_ROW, K_TAP = 20 * K_ROW;                              // a 20 -> 20 conv: [tap][t][row][oct 0..2]
constexpr int OFF_K1 = OFF_C1B + C1_SIZE, OFF_K3 = OFF_K1 + 9 * K_TAP, OFF_K4 = OFF_K3 + 9 * K_TAP, OFF_K2 = OFF_K4 + 9 * K_TAP;
constexpr int OFF_C3 = OFF_K2 + K_TAP;                                             // conv3: [t 0..3][row][oct 0..2 = a', 3..5 = b']
constexpr int c3_t(int t) { return (t < 2 ? t * 12 : 24 + (t - 2) * 8) * C3_ROW; }
constexpr int OFF_B2 = OFF_C3 + 40 * C3_ROW;                                       // k2's bias: 32 floats by channel (20 real)
constexpr int OFF_ZERO = OFF_B2 + 128;                                            // 16 zero bytes: what a lane reads for a structurally-zero fragment
constexpr int W_BYTES = OFF_ZERO + 16;
static_assert(W_BYTES == 34064, "blob layout");

}  // namespace scpa
}  // namespace innfer
