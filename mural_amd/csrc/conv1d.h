// Generic fp32 Conv1d building block (direct convolution on the vector ALU) used by the INDEL U-Net and by the
// per-layer (training-mode) SNV path.  Tensors are [B][C][L] fp32 contiguous, like the reference's NCL layout.
#pragma once
#include <vector>

#include "common.h"

namespace mural {

enum ConvAct { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_SOFTPLUS = 3 };

struct Conv1dArgs {
  const float* in;       // [B][Cin][Lin]
  const float* wt;       // weights re-laid-out as [Cin][K][Cout] (output channel contiguous)
  const float* bias;     // [Cout] or nullptr
  float* out;            // [B][Cout][Lout]
  int B, Cin, Lin, Cout, Lout;
  int K, stride, pad;
  int up;                // nearest-neighbour upsampling of the input by `up` (virtual input length Lin * up)
  // optional per-input-channel affine (+ReLU) applied to in-range input values BEFORE the zero padding:
  //   x' = pre_s[ci] * (pre_relu ? max(x, 0) : x) + pre_t[ci]          (BatchNorm in front of a conv)
  const float* pre_s;
  const float* pre_t;
  int pre_relu;
  int act;               // ConvAct applied to (acc + bias)
  const float* res1;     // optional residuals added AFTER the activation, [B][Cout][Lout]
  const float* res2;
  int rt;                // set by launch_conv1d: batch rows packed into one 64-lane tile (short rows), 1 otherwise
  // phases > 1 (MFMA kernel only): the polyphase form of conv(upsample_phases(x)).  Output column phases * i + p only sees the source
  // columns i + d for a few offsets d, so the K taps collapse to KJ = K summed taps per phase: wt is [Cin][KJ][Cout * phases] (row
  // index co * phases + p), K = KJ, pad = -d_min, stride = up = 1, Lout = Lin * phases.  7 taps -> 3 at phases 4 / 5.
  int phases;
};

int launch_conv1d(const Conv1dArgs& a, hipStream_t stream);        // routes between the two kernels below
int launch_conv1d_valu(const Conv1dArgs& a, hipStream_t stream);   // direct form on the vector ALU (conv1d.hip)
// implicit-GEMM version on fp32 MFMA for layers with >= 16 output channels (conv1d_mfma.hip); launch_conv1d routes to it
bool conv1d_mfma_supported(const Conv1dArgs& a);
int launch_conv1d_mfma(const Conv1dArgs& a, hipStream_t stream);
// stride-1 convs with 3 / 5 / 7 taps, <= 32 channels on long rows, operands straight from global memory (conv1d_direct.hip)
bool conv1d_direct_supported(const Conv1dArgs& a);
int launch_conv1d_direct(const Conv1dArgs& a, hipStream_t stream);
// the polyphase form of an upsampling conv (Conv1dArgs::phases, three source columns per phase) on the same barrier-free scheme
bool conv1d_direct_poly_supported(const Conv1dArgs& a);
int launch_conv1d_direct_poly(const Conv1dArgs& a, hipStream_t stream);
// host: weights [Cin][K][Cout] of conv_K(upsample_up(x)) -> polyphase weights [Cin][KJ][Cout * up] (Conv1dArgs::phases)
void conv1d_phase_weights(const float* w, int Cin, int K, int Cout, int up, std::vector<float>* out, int* KJ, int* pad_out);

// Fused ConvBlock of the INDEL U-Net (reference MuRaL/model/model_indel.py:6-19, eval mode, BatchNorms folded):
//   out = x + W1 . SiLU(W5 * x + b5) + b1 [+ res2],  W5: k=5 conv C -> 2C, W1: 1x1 conv 2C -> C.
// The 2C-channel intermediate lives in registers only.
struct ConvBlockArgs {
  const float* x;        // [B][C][L] (unused with a front)
  const float* w5;       // [C][5][2C]
  const float* b5;       // [2C]
  const float* w1;       // [2C][C]
  const float* b1;       // [C]
  const float* res2;     // optional [B][C][L] (decoder: encoder skip)
  float* out;            // [B][C][L] (unused with a tail)
  int B, C, L;
  // optional tail (model_indel.py:172-175): max over positions of Softplus(W_b . ReLU(W_a . out + b_a) + b_b), both 1x1 convs
  // C -> C with weights [Cin][Cout]; tail_max: [B][convblock_tiles()][C] per-workgroup maxima, reduced by the consumer
  // optional front: x itself is not read but produced in LDS as conv_k7(upsample_u(f_in)) + f_b, a stride-1 k=7 conv
  // Cf -> C with weights [Cf][7][C] (the decoder's Upsample + Conv1d + BN, model_indel.py:117-123, or the first encoder conv)
  const float* f_in;     // [B][Cf][Lf], Lf * f_up == L
  const float* f_w;
  const float* f_b;
  int Cf, Lf, f_up;
  int f_stride;          // convblock_deep.hip only (0 / 1 elsewhere): the front is a STRIDED k=7 conv, L == (Lf - 1) / f_stride + 1, f_up == 1
  const float* f_pw;     // optional (f_up == 4): polyphase front weights [4 phases][Cf][3][C] (taps summed per source column)
  // optional source of the front input (first encoder level, Cf == 4, f_up == 1): instead of reading f_in, the workgroup decodes its
  // span of the window from the packed genome (site g_pos[b], strand g_strand[b], window origin g_off) and evaluates the layer in
  // front of the U-Net -- the strand-symmetrising Conv1d(4, 4, k) + BN of model_indel.py:29-32 / :154-155, folded into ONE conv
  // on the host, or the identity when the model has none -- per symbol from a table: symtab[15 symbols][sym_taps][4] (+ sym_bias[4]
  // on columns inside the window).  Neither the one-hot window nor the layer's output ever exist in HBM.
  MuralGenome genome;
  const int64_t* g_pos;
  const uint8_t* g_strand;
  int g_off;
  const float* symtab;
  const float* sym_bias;
  int sym_taps;
  // optional, instead of the genome (persistent first-level kernel only): one symbol byte per column, [B][Lf] (0 .. 14 the MuRaL symbols,
  // 16: no symbol -- the column's floats are read from f_in [B][4][Lf], which must then be set)
  const uint8_t* sym_in;
  // optional (genome source): the per-symbol layer and the k=7 front composed into ONE conv of 6 + sym_taps taps, summed per group of
  // three taps over A C G T (indel_enc0_compose, indel_level0.hip): the persistent first-level kernel takes the launch then
  const float* e0_t3;
  const float* e0_t1;
  const float* e0_bias;
  // optional, with the composed front: the strided conv of the NEXT level (8 -> 16 channels, k = 7, stride 4, pad 3; weights
  // [8][7][16] + bias) is emitted by the same launch from the block's outputs while they are in LDS -- d_out [B][16][d_L],
  // d_L = (L - 1) / 4 + 1; `out` is still written (the decoder's skip)
  const float* d_w;
  const float* d_b;
  float* d_out;
  int d_L;
  const float* ta_w;
  const float* ta_b;
  const float* tb_w;
  const float* tb_b;
  float* tail_max;
};
bool convblock_supported(int C);
int convblock_tiles(int L, bool front);   // upper bound of the workgroups per row (sizing of tail_max)
int convblock_tiles_of(const ConvBlockArgs& a);   // workgroups per row of this launch = entries per row of its tail_max
int launch_convblock(const ConvBlockArgs& a, hipStream_t stream);
// fp32-MFMA version of the plain block (no front, no tail) for C = 16 / 24 (convblock_mfma.hip); launch_convblock routes to it
bool convblock_mfma_supported(const ConvBlockArgs& a);
int launch_convblock_mfma(const ConvBlockArgs& a, hipStream_t stream);


// the first encoder level from the packed genome as one persistent launch with a composed, table-driven front (indel_level0.hip)
void indel_enc0_compose(const float* fw, const float* fb, const float* symtab, const float* sym_bias, int st, std::vector<float>* t3,
                        std::vector<float>* t1, std::vector<float>* bias);
bool indel_enc0_supported(const ConvBlockArgs& a);
int launch_indel_enc0(const ConvBlockArgs& a, hipStream_t stream);
// the last decoder level (polyphase front from 16 channels, + skip, with or without the out_conv tail), persistent as well
bool indel_dec0_supported(const ConvBlockArgs& a);
int launch_indel_dec0(const ConvBlockArgs& a, hipStream_t stream);
// the 32-channel block on rows of up to 80 columns (the U-Net's fourth level): one row per workgroup pass (convblock_deep.hip)
bool convblock_deep_supported(const ConvBlockArgs& a);
int launch_convblock_deep(const ConvBlockArgs& a, hipStream_t stream);

// y[b][c] = max_l x[b][c][l]
int launch_rowmax(const float* x, int64_t rows, int L, float* y, hipStream_t stream);

}  // namespace mural
