// Weight-blob layout of one SCPA block of PAN (reference architectures/PAN_arch.py:58-105) as the MFMA A fragments the fused block kernels read from LDS, and the
// lane constants that go with it: shared by csrc/pan_scpa.hip (fp16 arithmetic) and csrc/pan_scpa_split.hip (the fp32-accurate form on (hi, lo) operand pairs: two
// blobs of this layout).
//
// Round 6: the 20 -> 20 convs' panels and the LDS images of A / B / Y are laid out for the bank rules of ds_read_b128 -- four groups of 16 lanes, {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63}, each needing 16 distinct 16-byte slots mod 256 B.  Row-major panels ((3 row + octet) * 16) and
// pixel-major 48-byte pixels put two lanes of a group on one slot: 0.41 - 0.43 of both kernels' LDS cycles were conflict cycles (profiles/r6/pmc_pan.txt, pan_fp32/).
//   * a tap block of a 20 -> 20 conv: [12 rows of tile 0 | 8 rows of tile 1 | zero | zero] (62 slots) with each tile's rows in a PERMUTED order sigma: the slots
//     3 sigma(row) + octet of a group's real lanes are distinct mod 16 (sigma from the segment structure of 3 * {0..11} / 3 * {0..7} mod 16: rows whose slots are
//     consecutive mod 16 must sit on the same side of the {0-3} | {4-11} split of a group).  Lanes of rows beyond the real ones read the fragment of a real row of
//     their OWN group (identical addresses broadcast; their output rows are channels that only meet zero weights), lanes of k-octet 3 -- structural padding -- the
//     block's zero slot, whose residue mod 16 no real fragment of their group has (slot 60 for tile 0's reads, 61 for tile 1's).
//     => fragment address = lane constant (kfrag_t0 / kfrag_t1) + immediate: no select, no address arithmetic per read;
//   * pixel images OCTET-major: octet o of pixel P at o * plane + 16 P with plane = 16 * pixels a multiple of 256 B: a group's lanes read 16 consecutive pixels of one
//     or two octet planes; lanes of k-octet 3 read octet 2 of their own pixel (finite data in a slot of their own, against the zero slot of the weights).
#pragma once
namespace innfer {
namespace scpa {

constexpr int TW = 32, HC = TW + 4;
constexpr int r2(int t) { return t ? 8 : 12; }          // real rows of 16-row tile t of a 32-row (20-channel) panel: row rho <-> channel 8 (rho >> 2) + 4 t + (rho & 3)
constexpr int r4(int t) { return t < 2 ? 12 : 8; }      // ... of a 64-row (40-channel) panel: row rho <-> channel 16 (rho >> 2) + 4 t + (rho & 3)
constexpr int C1_ROW = 5 * 16, C3_ROW = 6 * 16;
constexpr int C1_T1 = 12 * C1_ROW, C1_SIZE = 20 * C1_ROW;                         // conv1_a / conv1_b, row-major: [t][row][oct 0..3 = k-step 0, oct 4 = k 32..39]
constexpr int K_T1 = 36 * 16, K_Z0 = 60 * 16, K_Z1 = 61 * 16, K_TAP = 62 * 16;    // a tap block of a 20 -> 20 conv (see above)
constexpr int SIG0[12] = {7, 2, 8, 3, 9, 4, 10, 5, 0, 11, 6, 1};                  // position of row rho of tile 0 / tile 1 inside its part of a tap block
constexpr int SIG1[8] = {6, 1, 7, 2, 5, 0, 3, 4};
constexpr unsigned long long sig_nib(const int* t, int n) { unsigned long long v = 0; for (int i = n - 1; i >= 0; --i) v = (v << 4) | (unsigned)t[i]; return v; }
// (the 20 -> 20 convs first: their lo twins -- a second blob behind the first -- stay within a 16-bit offset of the lane constants)
constexpr int OFF_K1 = 0, OFF_K3 = OFF_K1 + 9 * K_TAP, OFF_K4 = OFF_K3 + 9 * K_TAP, OFF_K2 = OFF_K4 + 9 * K_TAP;
constexpr int OFF_C1A = OFF_K2 + K_TAP, OFF_C1B = OFF_C1A + C1_SIZE;
constexpr int OFF_C3 = OFF_C1B + C1_SIZE;                                          // conv3, row-major: [t 0..3][row][oct 0..2 = a', 3..5 = b']
constexpr int c3_t(int t) { return (t < 2 ? t * 12 : 24 + (t - 2) * 8) * C3_ROW; }
constexpr int OFF_B2 = OFF_C3 + 40 * C3_ROW;                                       // k2's bias: 32 floats by channel (20 real)
constexpr int OFF_ZERO = OFF_B2 + 128;                                            // 16 zero bytes: what a lane reads for a structurally-zero fragment of conv1 / conv3
constexpr int W_BYTES = OFF_ZERO + 16;
static_assert(W_BYTES == 34960, "blob layout");

// the lane's fragment offset inside a tap block, tile 0 / tile 1 (li = lane & 15: the panel row, lg = lane >> 4: the k-octet)
__device__ __forceinline__ int kfrag_t0(int li, int lg) {
    const int r = li < 12 ? li : li - 12;
    return lg < 3 ? (3 * (int)((sig_nib(SIG0, 12) >> (4 * r)) & 15) + lg) * 16 : K_Z0;
}
__device__ __forceinline__ int kfrag_t1(int li, int lg) {
    const int r = li < 8 ? li : (li < 12 ? li - 4 : li - 12);
    return lg < 3 ? K_T1 + (3 * (int)((sig_nib(SIG1, 8) >> (4 * r)) & 15) + lg) * 16 : K_Z1;
}

}  // namespace scpa
}  // namespace innfer
