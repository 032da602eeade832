// The 32-channel ConvBlock of the INDEL U-Net's fourth level (rows of 80 columns and fewer) as one launch.
//
// Reference: MuRaL/model/model_indel.py:6-19 (ConvBlock: x + BN(1x1(SiLU(BN(k5(x)))))), eval mode, BatchNorms folded on the host.  Same
// contract as the other ConvBlock launches (ConvBlockArgs: x, w5 [32][5][64], b5, w1 [64][32], b1, optional skip res2, out).
//
// On rows this short the fused blocks of the long levels do not apply (a lane / a wave owns positions of ONE row: 80 columns are five
// 16-column blocks), so the block ran as two launches of the tiled MFMA conv -- 53 + 38 us per 2048 rows for 26 us of MFMA time (4.0 GFLOP):
// their staging is per-element vector work (25 - 39 vector instructions per MFMA), and every launch of the forward sits on its SIMDs'
// issue time.  Here a workgroup takes one row at a time:
//   k = 5 conv  D[64 hidden][80] = W5[hidden][(tap, ci)] x[ci][col + tap - 2]: wave w owns hidden rows 16 w .. 16 w + 15 for all five
//               column blocks (five independent accumulators), its 40 A fragments stay in registers for the whole launch, the B operand
//               is one ds_read_b32 of the row image per MFMA (shared by nobody: each wave needs every column of every channel);
//   SiLU        on the accumulators, parked in LDS as the 1x1 conv's B operand;
//   1x1 conv    D[32][80]: wave w owns output rows 16 (w % 2) .. + 15 of the column blocks w / 2, w / 2 + 2, w / 2 + 4 (16 fragments);
//   epilogue    + block input (from the row image) + skip, 64 contiguous bytes per channel and block.
// The next row's ten dwords per thread are requested right after the barrier that publishes this row's image and land during the two
// matrix phases; two workgroups per CU (fragments + prefetch + five accumulators: no spill at 256 registers, 43 KB of LDS each).
#include <cstdlib>

#include "conv1d.h"
#include "mfma_tile.h"

namespace mural {
namespace {

constexpr int DB_C = 32, DB_H = 64;
constexpr int DB_NB = 5;            // 16-column blocks of a row (L <= 80)
constexpr int DB_PX = 112;          // row-image pitch, = 16 (mod 32) floats: the four channel rows of a B-operand read sit 16 banks apart
constexpr int DB_PH = 112;

__device__ __forceinline__ float silu_db(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

// FRONT: the block input is not read but produced here as the level's strided k = 7 conv (Cf <= 24 channels -> 32, stride f_stride, pad 3,
// BatchNorm folded) of the row above -- the encoder's fourth level, 24 x 400 -> 32 x 80 at stride 5: the source row (with three zero
// columns on either side) takes the place of the hidden tile in LDS (dead before the k = 5 phase writes it), the conv is 42 k-steps
// of the same MFMA with the operand read at a lane stride of `f_stride` floats (5: conflict-free), wave w: output rows 16 (w % 2) ..,
// column blocks w / 2, + 2, + 4.  The next source row travels in registers (ten 16-byte pieces per thread) under the matrix phases.
// Before: a launch of the tiled conv, 63 us per 2048 rows at 26 vector instructions per MFMA.
constexpr int DB_FC = 24, DB_PF = 432;      // front channels (max), source-row pitch: = 16 (mod 32), >= Lf + 6
constexpr int DB_FQ = 10;                   // 16-byte pieces of a source row per thread

template <int NBR, bool FRONT>      // 16-column blocks of a row
__global__ __launch_bounds__(256, 2) void convblock_deep32_kernel(const ConvBlockArgs a, const float* __restrict__ w5, const float* __restrict__ b5,
                                                                  const float* __restrict__ w1, const float* __restrict__ b1,
                                                                  const float* __restrict__ fw, const float* __restrict__ fb) {
  __shared__ __attribute__((aligned(16))) float xS[DB_C * DB_PX + 4];  // entry j of a channel row = column j - 2 (zeros outside the row); + a dump slot
  constexpr int HF = FRONT && DB_FC * DB_PF + 4 > DB_H * DB_PH ? DB_FC * DB_PF + 4 : DB_H * DB_PH;
  __shared__ __attribute__((aligned(16))) float hS[HF];      // the hidden tile; FRONT: first the source row, entry j of a channel = column j - 3
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4;
  const int L = a.L;
  constexpr int nbr = NBR;
  // A fragments, lane (m = n16, kk): k = 5 conv, k-step s = (tap s / 8, channel quad s % 8): ci = 4 (s % 8) + kk
  float a5[40], a1[16];
#pragma unroll
  for (int s = 0; s < 40; ++s) a5[s] = w5[((4 * (s & 7) + kk) * 5 + (s >> 3)) * DB_H + 16 * wave + n16];
#pragma unroll
  for (int s = 0; s < 16; ++s) a1[s] = w1[(4 * s + kk) * DB_C + 16 * (wave & 1) + n16];
  const f32x4 bias5 = ld4(b5 + 16 * wave + 4 * kk), bias1 = ld4(b1 + 16 * (wave & 1) + 4 * kk);
  // the image entries no row load writes: columns -2, -1 and L .. 16 nbr + 1 of every channel
  for (int i = tid; i < DB_C * DB_PX; i += 256) {
    const int j = i % DB_PX;
    if (j < 2 || j >= L + 2) xS[i] = 0.f;
  }
  const uint32_t row_bytes = (uint32_t)DB_C * (uint32_t)L * 4u;
  // front: 7 (Cf / 4) fragments, k-step s = (tap s / (Cf / 4), channel quad s % (Cf / 4)); channels past Cf: zero weights
  const int Cf = FRONT ? a.Cf : 0, Lf = FRONT ? a.Lf : 0, fst = FRONT ? a.f_stride : 1;
  // (in LDS, fragment-major [k-step][row block][lane]: 42 more registers beside 56 fragments, the travelling source row and five
  // accumulators spilled 41)
  constexpr int FKS = 7 * (DB_FC / 4);
  __shared__ __attribute__((aligned(16))) float afS[FRONT ? FKS * 2 * 64 : 4];
  f32x4 biasf = {0.f, 0.f, 0.f, 0.f};
  if constexpr (FRONT) {
    for (int i = tid; i < FKS * 2 * 64; i += 256) {
      const int fs = i >> 7, fmb = (i >> 6) & 1, fl = i & 63, ci = 4 * (fs % (DB_FC / 4)) + (fl >> 4);
      afS[i] = ci < Cf ? fw[(ci * 7 + fs / (DB_FC / 4)) * DB_C + 16 * fmb + (fl & 15)] : 0.f;
    }
    biasf = ld4(fb + 16 * (wave & 1) + 4 * kk);
    for (int i = tid; i < DB_FC * DB_PF + 4; i += 256) hS[i] = 0.f;      // (zero columns around the source row, zero rows past Cf)
  }
  const uint32_t frow_bytes = (uint32_t)Cf * (uint32_t)Lf * 4u;

  // LDS slot of each of this thread's ten dwords of a row (rows are [32][L] floats, dword i = tid + 256 u); past the row: the dump slot
  int slot[10];
#pragma unroll
  for (int u = 0; u < 10; ++u) {
    const int i = tid + 256 * u, ci = i / L;
    slot[u] = i < DB_C * L ? ci * DB_PX + (i - ci * L) + 2 : DB_C * DB_PX;
  }
  float v[FRONT ? 1 : 10];
  f32x4 vq[FRONT ? DB_FQ : 1];
  auto request = [&](int64_t row) {      // (a row past the launch's last: a descriptor of no bytes, the loads return 0 and are never stored)
    const bool in = row < a.B;
    if constexpr (FRONT) {
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.f_in) + (in ? (size_t)row * Cf * Lf : 0), 0,
                                                                          in ? (int)frow_bytes : 0, 0x00020000);
#pragma unroll
      for (int u = 0; u < DB_FQ; ++u) vq[u] = buf_ld4(rx, 16u * (uint32_t)(tid + 256 * u));
    } else {
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (in ? (size_t)row * DB_C * L : 0), 0,
                                                                          in ? (int)row_bytes : 0, 0x00020000);
#pragma unroll
      for (int u = 0; u < 10; ++u) v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, 4u * (uint32_t)(tid + 256 * u), 0, 0));
    }
  };
  if constexpr (FRONT) __syncthreads();      // (the zero fill of the source-row image is complete before the first row lands in it)
  request(blockIdx.x);

#pragma unroll 1
  for (int64_t row = blockIdx.x; row < a.B; row += gridDim.x) {
    // ------------------------------------------------------------------ the row -> LDS; the next row's ten dwords requested behind it
    if constexpr (FRONT) {
      int t = tid;      // (per row from an opaque index: ten hoisted offsets are ten registers)
      asm volatile("" : "+v"(t));
#pragma unroll
      for (int u = 0; u < DB_FQ; ++u) {
        const int e = 4 * (t + 256 * u);      // (Lf % 4 == 0: a piece stays inside its channel row)
        if (e < Cf * Lf) {
          const int ci = e / Lf, col = e - ci * Lf;
          float* d = hS + ci * DB_PF + col + 3;
          d[0] = vq[u][0]; d[1] = vq[u][1]; d[2] = vq[u][2]; d[3] = vq[u][3];
        }
      }
      // the zero columns around the row: the previous row's hidden tile has been here (channel rows past Cf keep whatever finite
      // values it left: their weights are zero)
      const int gap = DB_PF - Lf;      // entries Lf + 3 .. PF - 1 and 0 .. 2 of every channel row
      for (int i = t; i < Cf * gap; i += 256) {
        const int ci = i / gap, j = i - ci * gap;
        hS[ci * DB_PF + (j < 3 ? j : Lf + j)] = 0.f;
      }
    } else {
#pragma unroll
      for (int u = 0; u < 10; ++u) xS[slot[u]] = v[u];
    }
    __syncthreads();
    request(row + gridDim.x);
    // the skip operand of this wave's output blocks (up to twelve dwords per lane), requested before the matrix phases as well
    const int mb = wave & 1;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res2 ? a.res2 + (size_t)row * DB_C * L : w5), 0,
                                                                        a.res2 ? (int)row_bytes : 0, 0x00020000);
    float sk[3][4];
    uint32_t off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int nb = (wave >> 1) + 2 * i, col = 16 * nb + n16;
      off[i] = col < L ? ((uint32_t)(16 * mb + 4 * kk) * (uint32_t)L + (uint32_t)col) * 4u : 0x80000000u;      // (past the row: no access)
#pragma unroll
      for (int r = 0; r < 4; ++r) sk[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rk, off[i], (uint32_t)r * (uint32_t)L * 4u, 0));
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FRONT) {
      // ---------------------------------------------------------------- the strided k = 7 conv of the source row -> the block's input image
      const float* fbase = hS + kk * DB_PF + fst * n16;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int nb = (wave >> 1) + 2 * i;
        if (nb < nbr) {      // (wave-uniform)
          f32x4 o = biasf;
          const float* fp = fbase + fst * 16 * nb;
#pragma unroll
          for (int s = 0; s < FKS; ++s)
          {
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(afS[(2 * s + mb) * 64 + lane], fp[4 * (s % (DB_FC / 4)) * DB_PF + s / (DB_FC / 4)], o, 0, 0, 0);
            if (s % 6 == 5) __builtin_amdgcn_sched_barrier(0);
          }
          const int col = 16 * nb + n16;
          if (col < L) {
#pragma unroll
            for (int r = 0; r < 4; ++r) xS[(16 * mb + 4 * kk + r) * DB_PX + col + 2] = o[r];
          }
        }
      }
      __syncthreads();
    }
    // ------------------------------------------------------------------ k = 5 conv + SiLU: this wave's 16 hidden rows, every column block
    {
      const float* xb = xS + kk * DB_PX + n16;
      f32x4 acc[NBR];
#pragma unroll
      for (int nb = 0; nb < NBR; ++nb) acc[nb] = bias5;
#pragma unroll
      for (int s = 0; s < 40; ++s) {
#pragma unroll
        for (int nb = 0; nb < NBR; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a5[s], xb[4 * (s & 7) * DB_PX + 16 * nb + (s >> 3)], acc[nb], 0, 0, 0);
        if (FRONT && (s & 7) == 7) __builtin_amdgcn_sched_barrier(0);      // (keeps the operand reads from running dozens of registers ahead)
      }
#pragma unroll
      for (int nb = 0; nb < NBR; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) hS[(16 * wave + 4 * kk + r) * DB_PH + 16 * nb + n16] = silu_db(acc[nb][r]);
      }
    }
    __syncthreads();
    // ------------------------------------------------------------------ 1x1 conv, + block input, + skip, out
    {
      const float* hb = hS + kk * DB_PH + n16;
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)row * DB_C * L, 0, (int)row_bytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int nb = (wave >> 1) + 2 * i;
        if (nb < nbr) {      // (wave-uniform)
          const int col = 16 * nb + n16;
          f32x4 o = bias1;
#pragma unroll
          for (int s = 0; s < 16; ++s) o = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], hb[4 * s * DB_PH + 16 * nb], o, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = (o[r] + xS[(16 * mb + 4 * kk + r) * DB_PX + col + 2]) + sk[i][r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), ro, off[i], (uint32_t)r * (uint32_t)L * 4u, 0);
          }
        }
      }
    }
    __syncthreads();      // the next row's load overwrites the image
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The two deepest levels (40 channels x 16 columns, 48 channels x 8 columns): a row is one 16-column block or half of one, so the
// column blocks of a workgroup pass are ROWS -- NW = 2 C / 16 waves (5 / 6), wave w owns hidden rows 16 w .. 16 w + 15 of the k = 5
// conv for the pass's NW column blocks (5 rows of 16 columns / 12 rows of 8), then the whole 1x1 conv (three 16-row blocks, the last
// padded with zero weights) of column block w.  A channel row of the image holds the pass's rows side by side, each with its own two
// zero columns on either side, so the taps of one row never read its neighbour.  Before: two launches of the tiled conv per block,
// 22 + 14 us for 1 - 2 us of matrix work each.
template <int C, int L>
struct TinyGeo {
  static constexpr int H = 2 * C, NW = H / 16, MB2 = (C + 15) / 16, KS5 = C * 5 / 4, KS1 = H / 4;
  static constexpr int RPB = 16 / L, RP = NW * RPB, W = L + 4;      // rows per column block / per pass, width of a row image
  static constexpr int PX = ((RP * W + 31) / 32) * 32 + 16;         // pitches = 16 (mod 32)
  static constexpr int PH = ((16 * NW + 31) / 32) * 32 + 16;
  static constexpr int PER = (RP * C * L + 64 * NW - 1) / (64 * NW); // dwords of a pass per thread
  static_assert(C % 4 == 0 && 16 % L == 0 && H % 16 == 0, "geometry");
};

template <int C, int L, int NT = 64 * (2 * C / 16)>
__global__ __launch_bounds__(NT, 1) void convblock_tiny_kernel(const ConvBlockArgs a, const float* __restrict__ w5,
                                                                                  const float* __restrict__ b5, const float* __restrict__ w1,
                                                                                  const float* __restrict__ b1) {
  using G = TinyGeo<C, L>;
  static_assert(NT == 64 * G::NW, "one wave per 16 hidden rows");
  __shared__ __attribute__((aligned(16))) float xS[C * G::PX + 4];
  __shared__ __attribute__((aligned(16))) float hS[G::H * G::PH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4;
  constexpr uint32_t ROWB = (uint32_t)C * L * 4u;
  float v[G::PER];
  auto request = [&](int64_t row0) {      // rows past the batch: outside the descriptor, the loads return 0
    const int64_t left = a.B - row0;
    const int rows = left <= 0 ? 0 : (left < G::RP ? (int)left : G::RP);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (rows ? (size_t)row0 * C * L : 0), 0,
                                                                        (int)(rows * ROWB), 0x00020000);
#pragma unroll
    for (int u = 0; u < G::PER; ++u) v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, 4u * (uint32_t)(tid + NT * u), 0, 0));
  };
  request((int64_t)blockIdx.x * G::RP);
  // k = 5 fragments in registers; the 1x1 conv's (72 per lane at 48 channels: five / six waves share four SIMDs, 256 registers each) in
  // LDS, padded to whole 16-row blocks with zeros
  __shared__ __attribute__((aligned(16))) float w1S[G::H * 16 * G::MB2];
  float a5[G::KS5];
#pragma unroll
  for (int s = 0; s < G::KS5; ++s) a5[s] = w5[((4 * (s % (C / 4)) + kk) * 5 + s / (C / 4)) * G::H + 16 * wave + n16];
  for (int i = tid; i < G::H * 16 * G::MB2; i += NT) {
    const int k = i / (16 * G::MB2), c = i % (16 * G::MB2);
    w1S[i] = c < C ? w1[k * C + c] : 0.f;
  }
  const f32x4 bias5 = ld4(b5 + 16 * wave + 4 * kk);
  __shared__ __attribute__((aligned(16))) float b1S[16 * G::MB2];
  if (tid < 16 * G::MB2) b1S[tid] = tid < C ? b1[tid] : 0.f;
  for (int i = tid; i < C * G::PX + 4; i += NT) xS[i] = 0.f;      // (the zero columns stay; the row loads overwrite the rest each pass)
  __syncthreads();
  const int lane_x = (n16 / L) * G::W + n16 % L;      // this lane's column inside a column block's row images

#pragma unroll 1
  for (int64_t row0 = (int64_t)blockIdx.x * G::RP; row0 < a.B; row0 += (int64_t)gridDim.x * G::RP) {
    {
      // LDS slot of each of this thread's dwords of a pass (RP rows of [C][L] floats, contiguous in memory); derived per pass from an
      // opaque thread index: twelve hoisted slots are twelve registers this kernel does not have
      int t = tid;
      asm volatile("" : "+v"(t));
#pragma unroll
      for (int u = 0; u < G::PER; ++u) {
        const int i = t + NT * u, rr = i / (C * L), ci = (i / L) % C, col = i % L;
        xS[i < G::RP * C * L ? ci * G::PX + rr * G::W + col + 2 : C * G::PX] = v[u];
      }
    }
    __syncthreads();
    request(row0 + (int64_t)gridDim.x * G::RP);
    // skip operand of this wave's column block (rows row0 + RPB wave ..), every output channel block
    const int64_t left = a.B - row0;
    const int rows = left < G::RP ? (int)left : G::RP;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res2 ? a.res2 + (size_t)row0 * C * L : w5), 0,
                                                                        a.res2 ? (int)(rows * ROWB) : 0, 0x00020000);
    const int rr = wave * G::RPB + n16 / L;            // this lane's row of the pass in the 1x1 phase
    float sk[G::MB2][4];
    uint32_t off[G::MB2];
#pragma unroll
    for (int mb = 0; mb < G::MB2; ++mb) {
      const int c0 = 16 * mb + 4 * kk;
      off[mb] = c0 < C ? (uint32_t)((rr * C + c0) * L + n16 % L) * 4u : 0x80000000u;      // (C % 4 == 0: a lane's four channels are in or out together)
#pragma unroll
      for (int r = 0; r < 4; ++r) sk[mb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rk, off[mb], (uint32_t)r * L * 4u, 0));
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const float* xb = xS + kk * G::PX + lane_x;
      f32x4 acc[G::NW];
#pragma unroll
      for (int nb = 0; nb < G::NW; ++nb) acc[nb] = bias5;
#pragma unroll
      for (int s = 0; s < G::KS5; ++s) {
#pragma unroll
        for (int nb = 0; nb < G::NW; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a5[s], xb[4 * (s % (C / 4)) * G::PX + nb * G::RPB * G::W + s / (C / 4)], acc[nb], 0, 0, 0);
        if (s % 6 == 5) __builtin_amdgcn_sched_barrier(0);      // (keeps the operand reads from running a hundred registers ahead)
      }
#pragma unroll
      for (int nb = 0; nb < G::NW; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) hS[(16 * wave + 4 * kk + r) * G::PH + 16 * nb + n16] = silu_db(acc[nb][r]);
      }
    }
    __syncthreads();
    {
      const float* hb = hS + kk * G::PH + 16 * wave + n16;
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)row0 * C * L, 0, (int)(rows * ROWB), 0x00020000);
      f32x4 o[G::MB2];
#pragma unroll
      for (int mb = 0; mb < G::MB2; ++mb) o[mb] = *reinterpret_cast<const f32x4*>(b1S + 16 * mb + 4 * kk);
#pragma unroll
      for (int s = 0; s < G::KS1; ++s) {
        const float h = hb[4 * s * G::PH];
#pragma unroll
        for (int mb = 0; mb < G::MB2; ++mb)
          o[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1S[(4 * s + kk) * 16 * G::MB2 + 16 * mb + n16], h, o[mb], 0, 0, 0);
      }
#pragma unroll
      for (int mb = 0; mb < G::MB2; ++mb) {
        const int c0 = 16 * mb + 4 * kk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float xin = xS[(c0 + r < C ? c0 + r : 0) * G::PX + rr * G::W + n16 % L + 2];
          const float val = (o[mb][r] + xin) + sk[mb][r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, val), ro, off[mb], (uint32_t)r * L * 4u, 0);
        }
      }
    }
    __syncthreads();
  }
}

template <int C, int L>
int launch_tiny(const ConvBlockArgs& a, hipStream_t stream) {
  using G = TinyGeo<C, L>;
  const int64_t passes = (a.B + G::RP - 1) / G::RP;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    hipDeviceProp_t prop;
    MURAL_HIP_CHECK(hipGetDevice(&dev));
    MURAL_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    (void)n;      // (two workgroups per CU where they fit: measured slower, 36.6 vs 32.8 us at 40 channels -- the weights' traffic doubles)
    cus = prop.multiProcessorCount;
  }
  const dim3 grid((unsigned)(passes < cus ? passes : cus));
  hipLaunchKernelGGL((convblock_tiny_kernel<C, L>), grid, dim3(64 * G::NW), 0, stream, a, a.w5, a.b5, a.w1, a.b1);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

bool tiny_geometry(int C, int L) { return (C == 40 && L == 16) || (C == 48 && L == 8); }

}  // namespace

bool convblock_deep_supported(const ConvBlockArgs& a) {
  const bool off = dev_env("MURAL_INDEL_DEEP") && atoi(dev_env("MURAL_INDEL_DEEP")) == 0;
  const bool geo = (a.C == DB_C && a.L >= 1 && a.L <= 16 * DB_NB) || tiny_geometry(a.C, a.L);
  bool input = a.x != nullptr && a.f_in == nullptr;
  if (a.f_in != nullptr && a.C == DB_C) {      // the strided k = 7 front (convblock_deep32_kernel<., true>)
    const bool foff = dev_env("MURAL_INDEL_DEEP_FRONT") && atoi(dev_env("MURAL_INDEL_DEEP_FRONT")) == 0;
    input = !foff && a.f_w && a.f_b && a.f_up == 1 && a.f_pw == nullptr && a.f_stride >= 1 && a.f_stride <= 8 && a.Cf >= 4 && a.Cf <= DB_FC &&
            (a.Cf & 3) == 0 && (a.Lf & 3) == 0 && a.Lf + 6 <= DB_PF && a.Cf * a.Lf <= 4 * 256 * DB_FQ && (a.Lf - 1) / a.f_stride + 1 == a.L &&
            a.f_stride * (16 * ((a.L + 15) / 16) - 1) + 6 < DB_PF;
  }
  return !off && geo && input && a.out != nullptr && a.symtab == nullptr &&
         a.tail_max == nullptr && (uint64_t)a.C * a.L * 4 < (1ull << 31);
}

int launch_convblock_deep(const ConvBlockArgs& a, hipStream_t stream) {
  if (a.B == 0) return MURAL_OK;
  if (a.C == 40) return launch_tiny<40, 16>(a, stream);
  if (a.C == 48) return launch_tiny<48, 8>(a, stream);
  static int cap = 0;
  if (cap == 0) {
    int dev = 0, n = 0;
    hipDeviceProp_t prop;
    MURAL_HIP_CHECK(hipGetDevice(&dev));
    MURAL_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    MURAL_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, convblock_deep32_kernel<DB_NB, true>, 256, 0));
    cap = prop.multiProcessorCount * (n > 0 ? n : 1);
  }
  const dim3 grid((unsigned)(a.B < cap ? a.B : cap));
#define MURAL_DEEP_CASE(N)                                                                                                            \
  case N:                                                                                                                             \
    if (a.f_in) hipLaunchKernelGGL((convblock_deep32_kernel<N, true>), grid, dim3(256), 0, stream, a, a.w5, a.b5, a.w1, a.b1, a.f_w, a.f_b);    \
    else hipLaunchKernelGGL((convblock_deep32_kernel<N, false>), grid, dim3(256), 0, stream, a, a.w5, a.b5, a.w1, a.b1, a.f_w, a.f_b);          \
    break;
  switch ((a.L + 15) >> 4) {
    MURAL_DEEP_CASE(1) MURAL_DEEP_CASE(2) MURAL_DEEP_CASE(3) MURAL_DEEP_CASE(4) MURAL_DEEP_CASE(5)
    default: MURAL_REQUIRE(false, "internal: deep ConvBlock launched on rows of more than 80 columns");
  }
#undef MURAL_DEEP_CASE
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural
