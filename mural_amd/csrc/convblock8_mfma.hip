// The level-0 ConvBlocks of the INDEL U-Net (8 channels, rows of the full window length) on fp32 MFMA.
//
// Reference: MuRaL/model/model_indel.py:6-19 (ConvBlock), :35-42 / :117-134 (the convs in front of the first encoder and the last
// decoder block), :136-149 / :172-175 (out_conv + max over positions); eval mode, BatchNorms folded on the host.  Same contract and
// the same ConvBlockArgs as convblock_kernel<8, TAIL, true> (conv1d.hip), which stays as the VALU fallback; one workgroup = 252
// consecutive output positions of one batch row, persistent over (row, tile).
//
//   block    GEMM 1  D1[16 hidden][16 pos] += W5[hidden][(tap, ci)] x[ci][pos + tap - 2]: K = 5 x 8 = 10 k-steps; the hidden width of
//            the 8-channel block is exactly one 16-row MFMA tile.  SiLU on the accumulators, which then ARE the B operand of
//            GEMM 2 (1x1 conv 16 -> 8: 4 k-steps, 8 of the 16 output rows are padding).
//   encoder  front (k=7 conv 4 -> 8, model_indel.py:35-38): 224 multiply-adds per position on the vector ALU, lane = position, as
//            in the VALU kernel -- its input tile either staged from the strand-symmetrised tensor or decoded from the packed genome
//            (ConvBlockArgs::symtab).
//   decoder  front (Upsample(4) + k=7 conv 16 -> 8 as a polyphase 3-tap GEMM on the SOURCE columns, 32 rows = (channel, phase)):
//            2 M-blocks x 12 k-steps per 16 source columns; lane (source column, kk) ends up with the 4 phases = 4 consecutive
//            output positions of channel 4 mb + kk: one 16-byte store into the block-input tile.  Then + encoder skip, out_conv
//            (two 1x1 convs, ReLU / Softplus) and the maximum over the tile's positions (lane = position, vector ALU: 128 multiply-adds).
// The packed-FMA kernel spends one v_pk_fma_f32 per 4 FLOP and waits on scalar weight loads; here the block's 1536 FLOP per
// position are 14 MFMAs per 16 positions with both weight matrices resident in 14 VGPRs.
#include <cstdlib>

#include "conv1d.h"
#include "mfma_tile.h"

namespace mural {
unsigned long long* g_cb8_stamps = nullptr;      // diagnostic (tools/phase_stamps_cb8.py): per-workgroup phase sums, or nullptr
namespace {

constexpr int C8 = 8;
constexpr int C8_OUT = 252;        // output positions per tile (a multiple of 4: polyphase origin; = the VALU front kernel's tile)
constexpr int C8_PITCH = 272;      // = 16 (mod 32) floats
constexpr int C8_SPITCH = 80;      // decoder source tile pitch (67 columns used), = 16 (mod 32)
using f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ float silu8(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float softplus8(float v) { return v > 20.f ? v : log1pf(__expf(v)); }

// (the weight tables of the vector-ALU stages are separate __restrict__ parameters: only then are their wave-uniform reads scalar loads;
// read through the by-value struct they become per-lane loads that the compiler keeps in ~130 VGPRs for the whole persistent loop)
template <bool DEC>
__global__ __launch_bounds__(256) void convblock8_mfma_kernel(const ConvBlockArgs a, const float* __restrict__ f_w,
                                                              const float* __restrict__ f_b, const float* __restrict__ ta_w,
                                                              const float* __restrict__ ta_b, const float* __restrict__ tb_w,
                                                              const float* __restrict__ tb_b, unsigned long long* stamps) {
  unsigned long long t_prev = stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define CB8_STAMP(id)                                                          \
  if (stamps && threadIdx.x == 0) {                                            \
    const unsigned long long t_now = __builtin_amdgcn_s_memrealtime();         \
    stamps[8 * blockIdx.x + (id)] += t_now - t_prev;                           \
    t_prev = t_now;                                                            \
  }
  // xt: block input x, tile origin = position l0 - 4; ot: block output (block + residual), origin l0; fin: front input
  __shared__ __attribute__((aligned(16))) float xt[C8 * C8_PITCH];
  __shared__ __attribute__((aligned(16))) float ot[C8 * C8_PITCH];
  __shared__ __attribute__((aligned(16))) float fin[DEC ? 16 * C8_SPITCH + 32 : 4 * C8_PITCH];   // (+ the overhang of the fifth block's reads)
  __shared__ float stab[15 * 15 * 4 + 4];            // ConvBlockArgs::symtab | sym_bias (encoder, packed-genome source)
  __shared__ float wmax[4 * C8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4;

  // A fragments, lane (m = n16, kk)
  float a1[10], a2[4];
#pragma unroll
  for (int s = 0; s < 10; ++s) a1[s] = a.w5[((4 * (s & 1) + kk) * 5 + (s >> 1)) * 16 + n16];
#pragma unroll
  for (int s = 0; s < 4; ++s) a2[s] = n16 < C8 ? a.w1[(4 * kk + s) * C8 + n16] : 0.f;
  float af[2][12];
  if (DEC) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int s = 0; s < 12; ++s) {
        const int d = s >> 2, ci = 4 * (s & 3) + kk, row = 16 * mb + n16, co = row >> 2, p = row & 3;
        af[mb][s] = a.f_pw[(((size_t)p * 16 + ci) * 3 + d) * C8 + co];
      }
  }
  const f32x4 bias1 = ld4(a.b5 + 4 * kk);
  const f32x4 bias2 = kk < 2 ? ld4(a.b1 + 4 * kk) : splat(0.f);

  if (!DEC && a.symtab != nullptr) {
    for (int i = tid; i < 15 * a.sym_taps * 4; i += 256) stab[i] = a.symtab[i];
    if (tid < 4) stab[15 * 15 * 4 + tid] = a.sym_bias[tid];
  }
  const int tiles_per_row = (a.L + C8_OUT - 1) / C8_OUT;
  const int64_t total_tiles = (int64_t)a.B * tiles_per_row;
#pragma unroll 1
  for (int64_t tix = blockIdx.x; tix < total_tiles; tix += gridDim.x) {
    const int b = (int)(tix / tiles_per_row);
    const int tile_no = (int)(tix - (int64_t)b * tiles_per_row);
    const int l0 = tile_no * C8_OUT;
    __syncthreads();                                    // the previous tile is consumed
    CB8_STAMP(0);
    if (!DEC) {
      // ------------------------------------------------------------ encoder front: x[l0 - 2 + tid], tid = 0 .. 255
      // front input columns l0 - 5 .. l0 + 256 (262 of them) at fin[ci][0 .. 261]
      const int r0 = l0 - 5;
      if (a.symtab != nullptr) {
        uint8_t* symb = reinterpret_cast<uint8_t*>(ot);  // symbols of columns r0 - h .. (the output tile is written later)
        const int h = a.sym_taps >> 1;
        const int nsym = 262 + 2 * h;
        const int64_t ws = a.g_pos[b] + a.g_off;
        const bool neg = a.g_strand[b] != 0;
        for (int i = tid; i < nsym; i += 256) {
          const int j = r0 - h + i;
          uint32_t sy = SYM_PAD;
          if (j >= 0 && j < a.Lf) {
            sy = genome_sym_iupac(a.genome, neg ? ws + (a.Lf - 1 - j) : ws + j);
            if (neg) sy = sym_complement(sy);
          }
          symb[i] = (uint8_t)sy;
        }
        __syncthreads();
        for (int i = tid; i < 4 * 262; i += 256) {
          const int ci = i / 262, rr = i - ci * 262;
          const int r = r0 + rr;
          float v = 0.f;
          if (r >= 0 && r < a.Lf) {
            v = stab[15 * 15 * 4 + ci];
            for (int k = 0; k < a.sym_taps; ++k) {
              const uint32_t sy = symb[rr + k];
              if (sy != SYM_PAD) v += stab[(sy * a.sym_taps + k) * 4 + ci];
            }
          }
          fin[ci * C8_PITCH + rr] = v;
        }
      } else {
        const float* fsrc = a.f_in + (size_t)b * 4 * a.Lf;
        float v[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {                   // 4 x 262 floats = 5 per thread, all in flight
          const int i = tid + 256 * u;
          const int ci = i / 262, rr = i - ci * 262;
          const int r = r0 + rr;
          const bool ok = i < 4 * 262 && r >= 0 && r < a.Lf;
          v[u] = fsrc[ok ? (size_t)ci * a.Lf + r : 0];
          if (!ok) v[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int i = tid + 256 * u;
          if (i < 4 * 262) fin[(i / 262) * C8_PITCH + (i % 262)] = v[u];
        }
      }
      __syncthreads();
      {
        const int l = l0 - 2 + tid;
        f32x2 t[C8 / 2];
#pragma unroll
        for (int c = 0; c < C8 / 2; ++c) t[c] = f32x2{f_b[2 * c], f_b[2 * c + 1]};
#pragma unroll 1
        for (int ci = 0; ci < 4; ++ci) {
          const float* frow = fin + ci * C8_PITCH + tid;   // column (l - 3 + k) - r0 = tid + k
#pragma unroll
          for (int k = 0; k < 7; ++k) {
            const float* __restrict__ wk = f_w + (size_t)(ci * 7 + k) * C8;   // wave-uniform: scalar loads
            const float xv = frow[k];
            const f32x2 x2 = {xv, xv};
#pragma unroll
            for (int c = 0; c < C8 / 2; ++c) t[c] = __builtin_elementwise_fma(x2, f32x2{wk[2 * c], wk[2 * c + 1]}, t[c]);
          }
        }
        const bool in = l >= 0 && l < a.L;               // the k=5 conv zero-pads ITS input
#pragma unroll
        for (int c = 0; c < C8; ++c) xt[c * C8_PITCH + tid + 2] = in ? ((c & 1) ? t[c >> 1].y : t[c >> 1].x) : 0.f;
      }
    } else {
      // ------------------------------------------------------------ decoder front: polyphase GEMM on the source columns
      // source columns i0 - 1 .. i0 + 65 (i0 = l0 / 4 - 1) at fin[ci][0 .. 66]; MFMA blocks cover sources i0 .. i0 + 63, i.e. the
      // tile entries 0 .. 255 (positions l0 - 4 .. l0 + 251); positions l0 + 252, l0 + 253 (source i0 + 64) come from the vector ALU
      const int i0 = l0 / 4 - 1;
      const float* fsrc = a.f_in + (size_t)b * 16 * a.Lf;
      float v[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {                     // 16 x 67 floats = 5 per thread (the last round is partial)
        const int i = tid + 256 * u;
        const int ci = i / 67, rr = i - ci * 67;
        const int r = i0 - 1 + rr;
        const bool ok = i < 16 * 67 && r >= 0 && r < a.Lf;
        v[u] = fsrc[ok ? (size_t)ci * a.Lf + r : 0];
        if (!ok) v[u] = 0.f;                             // zero padding of the upsampled tensor
      }
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int i = tid + 256 * u;
        if (i < 16 * 67) fin[(i / 67) * C8_SPITCH + (i % 67)] = v[u];
      }
      __syncthreads();
      CB8_STAMP(1);
      // five 16-source blocks: one per wave, and wave 0 also takes the block behind them, of which only its first column (source
      // i0 + 64 = positions l0 + 252 .. l0 + 255) is inside the tile
#pragma unroll 1
      for (int nb = wave; nb < 5; nb += 4) {
        const float* sp = fin + kk * C8_SPITCH + 16 * nb + n16;   // source column (i0 + 16 nb + n16) + d - 1 sits at sp[d]
        f32x4 acc[2] = {splat(0.f), splat(0.f)};
#pragma unroll
        for (int s = 0; s < 12; ++s) {
          const float bv = sp[4 * (s & 3) * C8_SPITCH + (s >> 2)];
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][s], bv, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][s], bv, acc[1], 0, 0, 0);
        }
        const int j = 4 * (16 * nb + n16);               // tile entry of phase 0 (position l0 - 4 + j)
        if (j < 260) {
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const int co = 4 * mb + kk;
            const float fb = f_b[co];
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int l = l0 - 4 + j + r;
              o[r] = (l >= 0 && l < a.L) ? acc[mb][r] + fb : 0.f;
            }
            st4(xt + co * C8_PITCH + j, o);
          }
        }
      }
    }
    __syncthreads();
    CB8_STAMP(2);

    // -------------------------------------------------------------- the block: 16 position blocks, 4 per wave
#pragma unroll 1
    for (int pb = 0; pb < 4; ++pb) {
      const int p = 64 * wave + 16 * pb + n16;          // tile-relative output position (l0 + p)
      if (l0 + 64 * wave + 16 * pb >= a.L) break;       // wave-uniform: nothing of this block is inside the row
      const float* xp = xt + kk * C8_PITCH + p + 2;     // x[ci = 4 cq + kk][l0 + p + t - 2] sits at xp[4 cq * pitch + t]
      f32x4 acc1 = bias1;
#pragma unroll
      for (int s = 0; s < 10; ++s) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], xp[4 * (s & 1) * C8_PITCH + (s >> 1)], acc1, 0, 0, 0);
      f32x4 acc2 = bias2;
#pragma unroll
      for (int s = 0; s < 4; ++s) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[s], silu8(acc1[s]), acc2, 0, 0, 0);
      if (kk < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * kk + r;
          ot[c * C8_PITCH + p] = xt[c * C8_PITCH + p + 4] + acc2[r];
        }
      }
    }
    __syncthreads();
    CB8_STAMP(3);

    if (!DEC) {
      // ------------------------------------------------------------ stream the tile out as 16-byte pieces (L % 4 == 0) or scalars
      if ((a.L & 3) == 0) {
        for (int i = tid; i < C8 * (C8_OUT / 4); i += 256) {
          const int c = i / (C8_OUT / 4), q = i - c * (C8_OUT / 4);
          const int l = l0 + 4 * q;
          if (l < a.L) st4(a.out + ((size_t)b * C8 + c) * a.L + l, ld4(ot + c * C8_PITCH + 4 * q));
        }
      } else {
        for (int i = tid; i < C8 * C8_OUT; i += 256) {
          const int c = i / C8_OUT, j = i - c * C8_OUT;
          if (l0 + j < a.L) a.out[((size_t)b * C8 + c) * a.L + l0 + j] = ot[c * C8_PITCH + j];
        }
      }
    } else {
      // ------------------------------------------------------------ + encoder skip, out_conv, maximum over the tile (lane = position)
      const int l = l0 + tid;
      const bool live = tid < C8_OUT && l < a.L;
      float v[C8];
      {
        // the skip tensor's 8 values together (one branch, not one -- with a full wait behind its load -- per channel)
        float sk[C8];
#pragma unroll
        for (int c = 0; c < C8; ++c) sk[c] = 0.f;
        if (a.res2 && live) {
#pragma unroll
          for (int c = 0; c < C8; ++c) sk[c] = a.res2[((size_t)b * C8 + c) * a.L + l];
        }
#pragma unroll
        for (int c = 0; c < C8; ++c) v[c] = ot[c * C8_PITCH + (tid < C8_OUT ? tid : 0)] + sk[c];
      }
      if (a.tail_max == nullptr) {
        if (live) {
#pragma unroll
          for (int c = 0; c < C8; ++c) a.out[((size_t)b * C8 + c) * a.L + l] = v[c];
        }
      } else {
        f32x2 t[C8 / 2];
#pragma unroll
        for (int c = 0; c < C8 / 2; ++c) t[c] = f32x2{ta_b[2 * c], ta_b[2 * c + 1]};
#pragma unroll
        for (int j = 0; j < C8; ++j) {
          const f32x2 v2 = {v[j], v[j]};
#pragma unroll
          for (int c = 0; c < C8 / 2; ++c) t[c] = __builtin_elementwise_fma(v2, f32x2{ta_w[j * C8 + 2 * c], ta_w[j * C8 + 2 * c + 1]}, t[c]);
        }
        f32x2 u[C8 / 2];
#pragma unroll
        for (int c = 0; c < C8 / 2; ++c) u[c] = f32x2{tb_b[2 * c], tb_b[2 * c + 1]};
#pragma unroll
        for (int j = 0; j < C8; ++j) {
          const float r = fmaxf((j & 1) ? t[j >> 1].y : t[j >> 1].x, 0.f);
          const f32x2 r2 = {r, r};
#pragma unroll
          for (int c = 0; c < C8 / 2; ++c) u[c] = __builtin_elementwise_fma(r2, f32x2{tb_w[j * C8 + 2 * c], tb_w[j * C8 + 2 * c + 1]}, u[c]);
        }
#pragma unroll
        for (int c = 0; c < C8; ++c) {
          float sp = softplus8((c & 1) ? u[c >> 1].y : u[c >> 1].x);
          if (!live) sp = 0.f;                           // Softplus > 0: 0 is the identity of the maximum
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) sp = fmaxf(sp, __shfl_xor(sp, off));
          if (lane == 0) wmax[wave * C8 + c] = sp;
        }
        __syncthreads();
        if (tid < C8)
          a.tail_max[((size_t)b * tiles_per_row + tile_no) * C8 + tid] =
              fmaxf(fmaxf(wmax[tid], wmax[C8 + tid]), fmaxf(wmax[2 * C8 + tid], wmax[3 * C8 + tid]));
      }
    }
    CB8_STAMP(4);
    if (stamps && threadIdx.x == 0) stamps[8 * blockIdx.x + 7] += 1;
  }
#undef CB8_STAMP
}

}  // namespace

// the 8-channel block with a front: the first encoder level (k=7 front from a 4-channel tensor or from the packed genome) and the
// last decoder level (polyphase front from 16 channels, upsampling 4, with or without the out_conv tail)
// Measured (MI355X, 20480 positions at L = 8000): 510-520 k positions/s with these kernels against 570 k with the packed-FMA kernels of
// conv1d.hip -- on this part v_pk_fma_f32 and v_mfma_f32_16x16x4_f32 have the SAME peak rate (64 FLOP per cycle and SIMD), the 1x1
// conv pads its 8 output rows to 16, and a block's two dependent MFMA chains (10 + 4 deep, SiLU in between) leave the pipe idle at
// the occupancy its LDS tile allows.  Parity-green, kept for reference, OFF unless MURAL_CONVBLOCK8_MFMA=1.
bool convblock8_mfma_supported(const ConvBlockArgs& a) {
  if (a.C != C8 || getenv("MURAL_DEBUG_CONVBLOCK_VALU") || !getenv("MURAL_CONVBLOCK8_MFMA")) return false;
  const bool enc = (a.f_in != nullptr || a.symtab != nullptr) && a.Cf == 4 && a.f_up == 1 && a.tail_max == nullptr && a.res2 == nullptr;
  const bool dec = a.f_in != nullptr && a.symtab == nullptr && a.Cf == 16 && a.f_up == 4 && a.f_pw != nullptr && (a.L & 3) == 0;
  return enc || dec;
}

int launch_convblock8_mfma(const ConvBlockArgs& a, hipStream_t stream) {
  const int64_t tiles = (int64_t)a.B * ((a.L + C8_OUT - 1) / C8_OUT);
  const dim3 grid((unsigned)(tiles < 2048 ? tiles : 2048));
  if (a.Cf == 16) hipLaunchKernelGGL(convblock8_mfma_kernel<true>, grid, dim3(256), 0, stream, a, a.f_w, a.f_b, a.ta_w, a.ta_b, a.tb_w, a.tb_b, g_cb8_stamps);
  else hipLaunchKernelGGL(convblock8_mfma_kernel<false>, grid, dim3(256), 0, stream, a, a.f_w, a.f_b, a.ta_w, a.ta_b, a.tb_w, a.tb_b, g_cb8_stamps);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural
