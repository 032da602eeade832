// The first encoder level of the INDEL U-Net fed from the packed genome, as ONE persistent launch.
//
// Reference: MuRaL/model/model_indel.py:29-32 / :154-155 (strand-symmetrising Conv1d(4, 4, k) + BN in front of the U-Net), :35-38
// (first encoder conv 4 -> 8, k = 7, + BN), :6-19 (ConvBlock), MuRaL/data/preprocessing.py:756-816 (the one-hot window); eval mode,
// BatchNorms folded on the host.  Same contract and the same ConvBlockArgs as convblock_kernel<8, false, true, true> with a
// genome source (conv1d.hip), which stays as the fallback (MURAL_INDEL_ENC0=0) and as the parity partner of the tests.
//
// What the phase stamps of that kernel said (tools/phase_stamps_indel_l0.py, 2048 positions of L = 8000): of a workgroup's 12.2 us,
// 7.7 us pass before its front input is staged -- kernel arguments, the site's position, the genome words, a barrier, 7 dependent
// table reads per column -- and 2.3 us in the k = 7 conv (112 packed FMAs per position); the block itself is 2.3 us.  Here
//   * a workgroup walks a contiguous range of (row, tile) pairs and requests the genome words of tile t + 1 before it computes tile t:
//     no tile waits for its input, and the fragments / tables are fetched once per workgroup;
//   * the two linear layers in front of the block (per-symbol table of the strand-symmetrising conv, then the k = 7 conv) are ONE conv
//     of TT = 13 taps (7 without the symmetrising layer) from symbols to 8 channels.  The host composes it and sums it per group of
//     three taps over A C G T: 4 (2) reads of a [64 codes][8 channels] table + one single-tap read per position instead of 7 table
//     reads + 112 packed FMAs.  A lane's 3-mer codes come from three bit planes of the tile's symbols (wave ballots: low bit, high
//     bit, not-A-C-G-T), one funnel shift each;
//   * a position whose TT-column neighbourhood holds anything but A C G T -- N, an ambiguity code, the zero padding outside the window,
//     which is also every position within 3 of the window's ends, where the k = 7 conv pads ITS input -- takes the exact two-layer
//     form on the lane (49 table reads); a wave without such a lane never enters it.
// The block behind the front is the split form's: k = 5 conv 8 -> 16 and 1x1 conv 16 -> 8 on v_mfma_f32_16x16x4_f32, SiLU on the vector ALU.
#include <cstdlib>

#include "conv1d.h"
#include "mfma_tile.h"

namespace mural {
extern unsigned long long* g_cb8_stamps;      // diagnostic (conv1d.hip): per-workgroup phase sums, or nullptr
namespace {

constexpr int E0_OUT = 252;         // output positions per tile (= the split form's tile, so tail / tile bookkeeping is shared)
constexpr int E0_OUT_DOWN = 248;    // ... of the form that also emits the next level's stride-4 conv (62 columns of it per tile)
constexpr int E0_PITCH = 272;       // = 16 (mod 32) floats
constexpr int E0_C = 8;

__device__ __forceinline__ float silu0(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

struct E0Words { uint32_t w[2], m[2]; };   // genome words of a thread's one or two columns of a tile (2-bit bases, not-ACGT mask)

// BYTES: the window's symbols come from one byte per column (ConvBlockArgs::sym_in [B][Lf], the dense entry's one-hot windows
// classified by dense_to_symbols_kernel) instead of the packed genome; a column that is no MuRaL symbol (E0_SYM_DENSE) sends its
// neighbourhood through the exact form, which reads that column's four floats from the dense window itself (f_in [B][4][Lf]).
constexpr uint32_t E0_SYM_DENSE = 16;
template <int TT, bool STAMPS, bool DOWN, bool BYTES = false>
__global__ __launch_bounds__(256, 4) void indel_enc0_kernel(const ConvBlockArgs a, const float* __restrict__ w5, const float* __restrict__ b5,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            int tiles_per_row, long long total_tiles, unsigned long long* stamps) {
  constexpr int HW = (TT - 1) / 2;           // half width of the composed conv
  constexpr int ST = TT - 6;                 // taps of the per-symbol layer in front (7 or 1)
  constexpr int NG = (TT - 1) / 3;           // groups of three taps; the last tap goes alone
  constexpr int NSYM = 256 + TT - 1;         // columns a tile decodes: positions l0 + XO - HW .. l0 + XO + 255 + HW
  // DOWN: the tile also emits the next level's strided conv (8 -> 16, k = 7, stride 4, model_indel.py:39-42) of its outputs: 62 columns
  // of it per tile need the block's outputs from three positions in front of the tile's own 248, so the tile's origin moves by four
  constexpr int OUTW = DOWN ? E0_OUT_DOWN : E0_OUT;      // outputs stored per tile
  constexpr int XO = DOWN ? -6 : -2;                     // tile entry p <-> position l0 + XO + p; block output o <-> position l0 + XO + 2 + o
  constexpr int OLO = DOWN ? 4 : 0;                      // block outputs OLO .. OLO + OUTW - 1 are the tile's own
  static_assert(TT == 13 || TT == 7, "composed taps");
  unsigned long long t_prev = STAMPS ? __builtin_amdgcn_s_memrealtime() : 0ull;
  if (STAMPS && threadIdx.x == 0) stamps[8 * blockIdx.x + 6] = t_prev;      // (absolute: when the workgroup started)
#define E0_STAMP(id)                                                           \
  if (STAMPS && threadIdx.x == 0) {                                            \
    const unsigned long long t_now = __builtin_amdgcn_s_memrealtime();         \
    stamps[8 * blockIdx.x + (id)] += t_now - t_prev;                           \
    t_prev = t_now;                                                            \
  }
  __shared__ __attribute__((aligned(16))) float tile[E0_C * E0_PITCH];       // block input x, entry p = position l0 - 2 + p
  // rows of the 3-mer table at a pitch of 12 floats: a lane reads 32 bytes of a random row, and at a pitch of 8 the rows start on
  // 8 distinct bank positions (conflict rate 0.43 of the launch's LDS cycles); 12 gives 16
  constexpr int T3P = 12;
  __shared__ __attribute__((aligned(16))) float t3s[NG * 64 * T3P];
  __shared__ __attribute__((aligned(16))) float t1s[4 * E0_C + E0_C];        // single-tap table | bias of the composed conv
  __shared__ __attribute__((aligned(16))) float stabS[15 * ST * 4 + 4];      // exact form: per-symbol layer | its bias
  __shared__ __attribute__((aligned(16))) float fwS[4 * 7 * E0_C + E0_C];    // exact form: k = 7 conv [ci][k][co] | its bias
  __shared__ __attribute__((aligned(16))) float biasS[16 + E0_C + 16];       // b5 | b1 | bias of the strided conv (read per tile: fewer registers across the loop)
  __shared__ uint32_t planes[3 * 12];                                        // low bit | high bit | not-ACGT, 320 columns each (+ pad)
  __shared__ uint8_t symb[320];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = tid & 15, kk = (tid >> 4) & 3;

  for (int i = tid; i < NG * 64 * E0_C; i += 256) t3s[(i >> 3) * T3P + (i & 7)] = a.e0_t3[i];
  if (tid < 5 * E0_C) t1s[tid] = tid < 4 * E0_C ? a.e0_t1[tid] : a.e0_bias[tid - 4 * E0_C];
  for (int i = tid; i < 15 * ST * 4 + 4; i += 256) stabS[i] = i < 15 * ST * 4 ? a.symtab[i] : a.sym_bias[i - 15 * ST * 4];
  if (tid < 4 * 7 * E0_C + E0_C) fwS[tid] = tid < 4 * 7 * E0_C ? a.f_w[tid] : a.f_b[tid - 4 * 7 * E0_C];
  // the dead lanes behind the last output of a tile (o = 252 .. 255) read entries 256 .. 259: they meet live values in the paired 1x1
  // conv (multiplied by the zero half of A), so they must be finite
  if (tid < 4 * E0_C) tile[(tid >> 2) * E0_PITCH + 256 + (tid & 3)] = 0.f;
  // A fragments, lane (m = n16, kk): k = 5 conv, k-step s = (tap s / 2, ci 4 (s % 2) + kk); 1x1 conv on block PAIRS (b, b + 2)
  // (A = [W1 0] against block b, [0 W1] against block b + 2: one register per k-step, the zero half made by a lane mask at the use)
  float a5[10], a1w[4];
#pragma unroll
  for (int s = 0; s < 10; ++s) a5[s] = w5[((4 * (s & 1) + kk) * 5 + (s >> 1)) * 16 + n16];
#pragma unroll
  for (int q = 0; q < 4; ++q) a1w[q] = w1[(4 * kk + q) * E0_C + (n16 & 7)];
  if (tid < 16 + E0_C) biasS[tid] = tid < 16 ? b5[tid] : b1[tid - 16];
  // DOWN: A fragments of the strided conv, k-step s = (tap s / 2, ci 4 (s % 2) + kk), weights [8][7][16]
  float ad[DOWN ? 14 : 1];
  if constexpr (DOWN) {
#pragma unroll
    for (int s = 0; s < 14; ++s) ad[s] = a.d_w[((4 * (s & 1) + kk) * 7 + (s >> 1)) * 16 + n16];
    if (tid < 16) biasS[16 + E0_C + tid] = a.d_b[tid];
  }

  const long long first = total_tiles * (long long)blockIdx.x / (long long)gridDim.x;
  const long long last = total_tiles * ((long long)blockIdx.x + 1) / (long long)gridDim.x;
  if (first >= last) return;
  int b = (int)(first / tiles_per_row);
  int tile_no = (int)(first - (long long)b * tiles_per_row);
  const int Lf = a.Lf;
  const long long glen = BYTES ? (long long)a.B * a.Lf : a.genome.length;      // BYTES: "genome" = the symbol rows end to end, a row's origin = b Lf

  // a column's place in the genome: window column j of row (ws, neg)
  auto column = [&](int i, int l0, long long ws, bool neg, bool& inwin, bool& ing, long long& g) {
    const int j = l0 + XO - HW + i;
    inwin = (unsigned)j < (unsigned)Lf;
    g = neg ? ws + (long long)(Lf - 1 - j) : ws + (long long)j;
    ing = inwin && g >= 0 && g < glen;
  };
  auto request = [&](int tid, int l0, long long ws, bool neg, E0Words& p) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (r == 1 && wave != 0) break;                       // columns 256 .. : the first wave's second round
      bool inwin, ing;
      long long g;
      column(tid + 256 * r, l0, ws, neg, inwin, ing, g);
      const long long gi = ing ? g : 0;
      if constexpr (BYTES) {
        p.w[r] = a.sym_in[gi];
        p.m[r] = 0u;
      } else {
        p.w[r] = a.genome.packed2[gi >> 4];
        p.m[r] = a.genome.nmask[gi >> 5];
      }
    }
  };

  // (a row's origin and strand are wave-uniform: kept in scalar registers)
  auto uniform64 = [](long long v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  long long ws = BYTES ? (long long)b * Lf : uniform64(a.g_pos[b] + a.g_off);
  bool neg = BYTES ? false : __builtin_amdgcn_readfirstlane((int)a.g_strand[b]) != 0;
  E0Words cur;
  cur.w[1] = cur.m[1] = 0u;
  request(threadIdx.x, tile_no * OUTW, ws, neg, cur);
  E0_STAMP(0);

#pragma unroll 1
  for (long long tix = first; tix < last; ++tix) {
    const int l0 = tile_no * OUTW;
    // (the lane's address pieces are loop-invariant and the compiler keeps them in ~20 registers: 91 in all, five waves per SIMD --
    // re-deriving them per tile from an opaque thread index fits seven waves per SIMD at 59 registers and is 3 % slower: the
    // launch is bound by instruction issue, not by latency, so registers are cheaper than instructions)
    const int tid = threadIdx.x;
    const int lane = tid & 63, n16 = tid & 15, kk = (tid >> 4) & 3;
    // ---------------------------------------------------------------- symbols of the tile's columns -> bit planes (+ bytes for the exact form)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (r == 1 && wave != 0) break;
      const int i = tid + 256 * r;
      bool inwin, ing;
      long long g;
      column(i, l0, ws, neg, inwin, ing, g);
      uint32_t sy;
      if constexpr (BYTES) {
        sy = cur.w[r];      // (every column of the window has its byte: ing == inwin)
      } else {
        const bool masked = ((cur.m[r] >> (uint32_t)(g & 31)) & 1u) != 0u;
        sy = (cur.w[r] >> (2u * (uint32_t)(g & 15))) & 3u;
        sy = (ing && !masked) ? sy : (uint32_t)SYM_N;
        if (ing && masked && a.genome.n_amb > 0) sy = genome_sym_iupac(a.genome, g);      // ambiguity codes: the sparse side table
        if (neg) sy = sym_complement(sy);
      }
      sy = (inwin && i < NSYM) ? sy : (uint32_t)SYM_PAD;
      const unsigned long long blo = __ballot((sy & 1u) != 0u), bhi = __ballot((sy & 2u) != 0u), bna = __ballot(sy >= 4u);
      symb[i] = (uint8_t)sy;
      if (lane == 0) {
        const int wq = 2 * (wave + 4 * r);
        planes[wq] = (uint32_t)blo;
        planes[wq + 1] = (uint32_t)(blo >> 32);
        planes[12 + wq] = (uint32_t)bhi;
        planes[12 + wq + 1] = (uint32_t)(bhi >> 32);
        planes[24 + wq] = (uint32_t)bna;
        planes[24 + wq + 1] = (uint32_t)(bna >> 32);
      }
    }
    __syncthreads();                       // (also: the previous tile's block has read the tile)
    E0_STAMP(1);
    // ---------------------------------------------------------------- the next tile's genome words, in flight under this tile's work
    int nb = b, ntile = tile_no + 1;
    if (ntile == tiles_per_row) {
      ntile = 0;
      ++nb;
    }
    long long nws = ws;
    bool nneg = neg;
    E0Words nxt;
    nxt.w[0] = nxt.m[0] = nxt.w[1] = nxt.m[1] = 0u;
    if (tix + 1 < last) {
      if (nb != b) {
        nws = BYTES ? (long long)nb * Lf : uniform64(a.g_pos[nb] + a.g_off);
        nneg = BYTES ? false : __builtin_amdgcn_readfirstlane((int)a.g_strand[nb]) != 0;
      }
      request(tid, ntile * OUTW, nws, nneg, nxt);
    }
    // ---------------------------------------------------------------- front: x[.][l0 - 2 + tid]
    {
      const int l = l0 + XO + tid;
      const int wd = tid >> 5;
      const uint32_t sh = (uint32_t)(tid & 31);
      const uint32_t lo = __builtin_amdgcn_alignbit(planes[wd + 1], planes[wd], sh);
      const uint32_t hi = __builtin_amdgcn_alignbit(planes[12 + wd + 1], planes[12 + wd], sh);
      const uint32_t na = __builtin_amdgcn_alignbit(planes[24 + wd + 1], planes[24 + wd], sh);
      f32x4 x0 = ld4(t1s + 4 * E0_C), x1 = ld4(t1s + 4 * E0_C + 4);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const uint32_t code = ((lo >> (3 * g)) & 7u) | (((hi >> (3 * g)) & 7u) << 3);
        const float* r = t3s + (g * 64 + (int)code) * T3P;
        x0 += ld4(r);
        x1 += ld4(r + 4);
      }
      {
        const uint32_t s = ((lo >> (TT - 1)) & 1u) | (((hi >> (TT - 1)) & 1u) << 1);
        const float* r = t1s + (int)s * E0_C;
        x0 += ld4(r);
        x1 += ld4(r + 4);
      }
      if ((na & ((1u << TT) - 1u)) != 0u) {
        // exact form: S = per-symbol layer on the columns inside the window, x = k = 7 conv of S with S zero outside the window
        f32x4 e0 = ld4(fwS + 4 * 7 * E0_C), e1 = ld4(fwS + 4 * 7 * E0_C + 4);
#pragma unroll 1
        for (int k2 = 0; k2 < 7; ++k2) {
          const int j = l + k2 - 3;
          if ((unsigned)j >= (unsigned)Lf) continue;
          f32x4 S = ld4(stabS + 15 * ST * 4);
          if constexpr (BYTES) {
#pragma unroll 1
            for (int k = 0; k < ST; ++k) {      // (rolled: this instance's slow path also reads the dense window; unrolled it spilled 49 registers)
              const uint32_t sy = symb[tid + k2 + k];
              if (sy == E0_SYM_DENSE) {         // no symbol: the layer is linear in the column's four floats
                const int jc = l0 + XO - HW + tid + k2 + k;
                const float* dv = a.f_in + (size_t)b * 4 * Lf + jc;
#pragma unroll 1
                for (int ci = 0; ci < 4; ++ci) S += splat(dv[(size_t)ci * Lf]) * ld4(stabS + (ci * ST + k) * 4);
              } else if (sy != SYM_PAD) S += ld4(stabS + ((int)sy * ST + k) * 4);
            }
          } else {
#pragma unroll
          for (int k = 0; k < ST; ++k) {
            const uint32_t sy = symb[tid + k2 + k];
            if (sy != SYM_PAD) S += ld4(stabS + ((int)sy * ST + k) * 4);
          }
          }
#pragma unroll
          for (int ci = 0; ci < 4; ++ci) {
            const float* wr = fwS + (ci * 7 + k2) * E0_C;
            e0 += splat(S[ci]) * ld4(wr);
            e1 += splat(S[ci]) * ld4(wr + 4);
          }
        }
        x0 = e0;
        x1 = e1;
      }
      const bool in = l >= 0 && l < a.L;                 // the k = 5 conv zero-pads ITS input
      if (!in) x0 = x1 = splat(0.f);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        tile[c * E0_PITCH + tid] = x0[c];
        tile[(c + 4) * E0_PITCH + tid] = x1[c];
      }
    }
    __syncthreads();
    E0_STAMP(2);
    // ---------------------------------------------------------------- the block: output o = 64 wave + 16 bk + n16 <-> position l0 + o,
    // its five taps at tile entries o .. o + 4
    {
      const int o0 = 64 * wave + n16;
      const float* xb = tile + kk * E0_PITCH + o0;
      int bo = 4 * kk;
      asm volatile("" : "+v"(bo));                     // (the bias reads stay inside the loop)
      const f32x4 bias5 = ld4(biasS + bo);
      f32x4 acc[4] = {bias5, bias5, bias5, bias5};
#pragma unroll
      for (int s = 0; s < 10; ++s)
#pragma unroll
        for (int bk = 0; bk < 4; ++bk)
          acc[bk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a5[s], xb[4 * (s & 1) * E0_PITCH + 16 * bk + (s >> 1)], acc[bk], 0, 0, 0);
      float h[4][4];
#pragma unroll
      for (int bk = 0; bk < 4; ++bk)
#pragma unroll
        for (int q = 0; q < 4; ++q) h[bk][q] = silu0(acc[bk][q]);
      E0_STAMP(3);
      const f32x4 bias1 = ld4(biasS + 16 + (bo & 4));
      f32x4 o[2] = {bias1, bias1};
      uint32_t lowm = n16 < 8 ? 0xffffffffu : 0u;
      asm volatile("" : "+v"(lowm));                   // (not hoisted into eight more live registers)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float wa = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a1w[q]) & lowm);
#pragma unroll
        for (int p = 0; p < 2; ++p) o[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, h[p][q], o[p], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float wb = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a1w[q]) & ~lowm);
#pragma unroll
        for (int p = 0; p < 2; ++p) o[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb, h[p + 2][q], o[p], 0, 0, 0);
      }
      // lane (n16, kk) holds channels 4 (kk % 2) + q of blocks p + 2 (kk / 2), p = 0, 1: + block input, out
      const int cb = 4 * (kk & 1);
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * E0_C * a.L, 0, (int)((uint32_t)E0_C * (uint32_t)a.L * 4u), 0x00020000);
      float v[2][4];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int ob = o0 + 16 * (p + 2 * (kk >> 1));
        const int l = l0 + XO + 2 + ob;
        const bool live = ob >= OLO && ob < OLO + OUTW && l < a.L;
        uint32_t off = ((uint32_t)cb * (uint32_t)a.L + (uint32_t)l) * 4u;
        off = live ? off : 0x80000000u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[p][q] = o[p][q] + tile[(cb + q) * E0_PITCH + ob + 2];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v[p][q]), ro, off, (uint32_t)q * (uint32_t)a.L * 4u, 0);
        }
      }
      if constexpr (DOWN) {
        // ------------------------------------------------------------ the next level's strided conv of this tile's outputs: the block
        // outputs go back into the tile (entry = block output index; zero outside the row: the conv pads ITS input), then column
        // c = 16 wave + n16 of the tile's 62 (row column l0 / 4 + c) is D[16 channels] = W[16][(tap, ci)] out[ci][4 c + 1 + tap]
        __syncthreads();                   // every wave has read its block input
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int ob = o0 + 16 * (p + 2 * (kk >> 1));
          const int l = l0 + XO + 2 + ob;
          const bool in = l >= 0 && l < a.L;
#pragma unroll
          for (int q = 0; q < 4; ++q) tile[(cb + q) * E0_PITCH + ob] = in ? v[p][q] : 0.f;
        }
        __syncthreads();
        const int c = 16 * wave + n16;
        const float* yb = tile + kk * E0_PITCH + 4 * c + 1;
        f32x4 dacc = ld4(biasS + 16 + E0_C + bo);
#pragma unroll
        for (int s = 0; s < 14; ++s) dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ad[s], yb[4 * (s & 1) * E0_PITCH + (s >> 1)], dacc, 0, 0, 0);
        const int col = l0 / 4 + c;
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(a.d_out + (size_t)b * 16 * a.d_L, 0, (int)(16u * (uint32_t)a.d_L * 4u), 0x00020000);
        uint32_t doff = ((uint32_t)(4 * kk) * (uint32_t)a.d_L + (uint32_t)col) * 4u;
        doff = (c < OUTW / 4 && col < a.d_L) ? doff : 0x80000000u;
        // (the four rows as named scalars pinned behind the MFMA chain: without the pin the four stores below came out as four
        // stores of row 0's register, the other three overwritten by the address arithmetic -- caught by the packed-entry tests)
        float d0 = dacc.x, d1 = dacc.y, d2 = dacc.z, d3 = dacc.w;
        asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        const uint32_t rowb = (uint32_t)a.d_L * 4u;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, d0), rd, doff, 0u, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, d1), rd, doff, rowb, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, d2), rd, doff, 2u * rowb, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, d3), rd, doff, 3u * rowb, 0);
      }
    }
    E0_STAMP(4);
    if (STAMPS && threadIdx.x == 0) stamps[8 * blockIdx.x + 7] += 1;
    cur = nxt;
    ws = nws;
    neg = nneg;
    b = nb;
    tile_no = ntile;
  }
#undef E0_STAMP
}


// ------------------------------------------------------------------------------------------------ last decoder level
// Upsample(4) + Conv1d(16 -> 8, k = 7) + BN as a polyphase GEMM on the source columns, the ConvBlock, + encoder skip, out_conv
// (1x1, BN, ReLU, 1x1, Softplus) and the maximum over positions (model_indel.py:117-134, :136-149, :172-175), persistent like the
// encoder kernel above: the 16 x 66 source columns of tile t + 1 and the skip values of tile t are requested before tile t's matrix
// work; every fragment stays in registers (four waves per SIMD: the launch is bound by instruction issue, not latency; LDS only
// stages them once per workgroup); the maximum over a row's positions
// is carried in registers from tile to tile and leaves the workgroup once per row segment (the other tiles' slots get 0, the
// identity of a maximum of Softplus values).  Tile geometry = convblock_kernel<8, true, true, true>'s polyphase form: 248 outputs.
constexpr int D0_OUT = 248;         // = CB_FRONT_OUT_POLY (conv1d.hip): tail_max has one slot per such tile
constexpr int D0_SPITCH = 68;       // source tile pitch (66 columns used)

template <bool STAMPS>
__global__ __launch_bounds__(256, 4) void indel_dec0_kernel(const ConvBlockArgs a, const float* __restrict__ w5, const float* __restrict__ b5,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            int tiles_per_row, long long total_tiles, unsigned long long* stamps) {
  unsigned long long t_prev = STAMPS ? __builtin_amdgcn_s_memrealtime() : 0ull;
  if (STAMPS && threadIdx.x == 0) stamps[8 * blockIdx.x + 6] = t_prev;
#define D0_STAMP(id)                                                           \
  if (STAMPS && threadIdx.x == 0) {                                            \
    const unsigned long long t_now = __builtin_amdgcn_s_memrealtime();         \
    stamps[8 * blockIdx.x + (id)] += t_now - t_prev;                           \
    t_prev = t_now;                                                            \
  }
  __shared__ __attribute__((aligned(16))) float tile[E0_C * E0_PITCH];       // block input x, entry p = position l0 - 4 + p
  __shared__ __attribute__((aligned(16))) float fin[16 * D0_SPITCH];         // source columns l0 / 4 - 2 .. l0 / 4 + 63
  __shared__ float afS[24 * 64];                                             // polyphase A fragments [mb][s][lane of (n16, kk)]
  __shared__ float tS[8 * 64];                                               // tail A fragments: diag(Wa, Wa) | diag(Wb, Wb)
  __shared__ float a5S[10 * 64];                                             // k = 5 conv A fragments (in registers they cost ten spills here)
  __shared__ __attribute__((aligned(16))) float biasS[16 + E0_C + E0_C + E0_C + E0_C];   // b5 | b1 | front bias | tail a | tail b
  __shared__ float red[4 * E0_C];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = tid & 15, kk = (tid >> 4) & 3;

  if (tid < 64) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int s = 0; s < 12; ++s) {
        const int d = s >> 2, ci = 4 * (s & 3) + kk, row = 16 * mb + n16, co = row >> 2, ph = row & 3;
        afS[(mb * 12 + s) * 64 + tid] = a.f_pw[(((size_t)ph * 16 + ci) * 3 + d) * E0_C + co];
      }
#pragma unroll
    for (int s = 0; s < 10; ++s) a5S[s * 64 + tid] = w5[((4 * (s & 1) + kk) * 5 + (s >> 1)) * 16 + n16];
    const bool own = (n16 >> 3) == (kk >> 1);                                // diag(W, W) on the paired layout
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      tS[q * 64 + tid] = (own && a.tail_max) ? a.ta_w[(4 * (kk & 1) + q) * E0_C + (n16 & 7)] : 0.f;
      tS[(4 + q) * 64 + tid] = (own && a.tail_max) ? a.tb_w[(4 * (kk & 1) + q) * E0_C + (n16 & 7)] : 0.f;
    }
  }
  if (tid < 16) biasS[tid] = b5[tid];
  else if (tid < 24) biasS[tid] = b1[tid - 16];
  else if (tid < 32) biasS[tid] = a.f_b[tid - 24];
  else if (tid < 40) biasS[tid] = a.tail_max ? a.ta_b[tid - 32] : 0.f;
  else if (tid < 48) biasS[tid] = a.tail_max ? a.tb_b[tid - 40] : 0.f;
  if (tid < 4 * E0_C) tile[(tid >> 2) * E0_PITCH + 256 + (tid & 3)] = 0.f;     // entries 256 .. 259: read by dead lanes only, must be finite
  float a1w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a1w[q] = w1[(4 * kk + q) * E0_C + (n16 & 7)];
  __syncthreads();
  // (four waves per SIMD: this launch is bound by instruction issue, not by latency -- the fragments of the polyphase front, the k = 5
  // conv and the tail stay in registers, 42 LDS reads per wave and tile less than at six waves per SIMD)
  float afr[24], a5r[10], tfr[8];
#pragma unroll
  for (int s2 = 0; s2 < 24; ++s2) afr[s2] = afS[s2 * 64 + lane];
#pragma unroll
  for (int s2 = 0; s2 < 10; ++s2) a5r[s2] = a5S[s2 * 64 + lane];
#pragma unroll
  for (int s2 = 0; s2 < 8; ++s2) tfr[s2] = tS[s2 * 64 + lane];

  const long long first = total_tiles * (long long)blockIdx.x / (long long)gridDim.x;
  const long long last = total_tiles * ((long long)blockIdx.x + 1) / (long long)gridDim.x;
  if (first >= last) return;
  int b = (int)(first / tiles_per_row);
  int tile_no = (int)(first - (long long)b * tiles_per_row);
  const int Lf = a.Lf, L = a.L;
  const uint32_t src_row_bytes = 16u * (uint32_t)Lf * 4u, out_row_bytes = (uint32_t)E0_C * (uint32_t)L * 4u;

  // a thread's five source values of a tile: channel tid / 16, columns i0 - 1 + tid % 16 + 16 u (u = 4: two columns)
  auto request = [&](int tid, int bb, int l0, float (&v)[5]) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.f_in) + (size_t)bb * 16 * Lf, 0, (int)src_row_bytes, 0x00020000);
    const int ci = tid >> 4, c0 = l0 / 4 - 2 + (tid & 15);
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int c = c0 + 16 * u;
      const bool ok = (unsigned)c < (unsigned)Lf && (u < 4 || (tid & 15) < 2);
      const uint32_t off = ok ? (uint32_t)(ci * Lf + c) * 4u : 0x80000000u;      // outside the row: the zero padding of the upsampled tensor
      v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
    }
  };
  float cur[5];
  request(threadIdx.x, b, tile_no * D0_OUT, cur);
  float rmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};     // running maximum of the raw tail values, channels 4 (kk % 2) + q
  D0_STAMP(0);

#pragma unroll 1
  for (long long tix = first; tix < last; ++tix) {
    const int l0 = tile_no * D0_OUT;
    // the lane's indices are re-derived from an opaque copy of the thread index in every iteration: hoisted out of the loop, the address
    // pieces made from them (and the 42 fragment reads from LDS) do not fit the register budget of six waves per SIMD and come back
    // as scratch reloads behind a full wait -- which is the wait for the next tile's prefetch
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, n16 = tid & 15, kk = (tid >> 4) & 3;
    {
      const int ci = tid >> 4, rr0 = tid & 15;
#pragma unroll
      for (int u = 0; u < 5; ++u)
        if (u < 4 || rr0 < 2) fin[ci * D0_SPITCH + rr0 + 16 * u] = cur[u];
    }
    __syncthreads();                       // (also: the previous tile's block has read the tile)
    D0_STAMP(1);
    int nb = b, ntile = tile_no + 1;
    if (ntile == tiles_per_row) {
      ntile = 0;
      ++nb;
    }
    float nxt[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (tix + 1 < last) request(tid, nb, ntile * D0_OUT, nxt);
    // the skip values of this tile's outputs: lane (n16, kk) owns channels 4 (kk % 2) + q of blocks p + 2 (kk / 2), output t = position l0 - 2 + t
    const int cb = 4 * (kk & 1);
    const int t0 = 64 * wave + n16;
    float sk[2][4];
    bool live[2];
    {
      const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res2 ? a.res2 + (size_t)b * E0_C * L : w5), 0,
                                                                          a.res2 ? (int)out_row_bytes : 0, 0x00020000);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int t = t0 + 16 * (p + 2 * (kk >> 1));
        const int l = l0 - 2 + t;
        live[p] = (t >= 2) & (t < 2 + D0_OUT) & (l < L);
        uint32_t off = ((uint32_t)cb * (uint32_t)L + (uint32_t)l) * 4u;
        off = live[p] ? off : 0x80000000u;
#pragma unroll
        for (int q = 0; q < 4; ++q) sk[p][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rk, off, (uint32_t)q * (uint32_t)L * 4u, 0));
      }
    }
    // ---------------------------------------------------------------- polyphase front: this wave's 16 source columns -> 64 tile entries
    {
      const float* sp = fin + kk * D0_SPITCH + 16 * wave + n16;      // source column (i0 + 16 wave + n16) + d - 1 at sp[d]
      f32x4 accf[2] = {splat(0.f), splat(0.f)};
#pragma unroll
      for (int s = 0; s < 12; ++s) {
        const float bv = sp[4 * (s & 3) * D0_SPITCH + (s >> 2)];
        accf[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[s], bv, accf[0], 0, 0, 0);
        accf[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[12 + s], bv, accf[1], 0, 0, 0);
      }
      const int j = 64 * wave + 4 * n16;               // tile entry of phase 0 (position l0 - 4 + j)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int co = 4 * mb + kk;
        const float fb = biasS[24 + co];
        f32x4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int l = l0 - 4 + j + r;
          o4[r] = (l >= 0 && l < L) ? accf[mb][r] + fb : 0.f;      // the k = 5 conv zero-pads ITS input
        }
        st4(tile + co * E0_PITCH + j, o4);
      }
    }
    __syncthreads();
    D0_STAMP(2);
    // ---------------------------------------------------------------- the block, + skip, tail
    {
      const float* xb = tile + kk * E0_PITCH + t0;       // output t's five taps at entries t .. t + 4
      int bo = 4 * kk;
      asm volatile("" : "+v"(bo));
      const f32x4 bias5 = ld4(biasS + bo);
      f32x4 acc[4] = {bias5, bias5, bias5, bias5};
#pragma unroll
      for (int s = 0; s < 10; ++s) {
        const float a5s = a5r[s];
#pragma unroll
        for (int bk = 0; bk < 4; ++bk)
          acc[bk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a5s, xb[4 * (s & 1) * E0_PITCH + 16 * bk + (s >> 1)], acc[bk], 0, 0, 0);
      }
      float h[4][4];
#pragma unroll
      for (int bk = 0; bk < 4; ++bk)
#pragma unroll
        for (int q = 0; q < 4; ++q) h[bk][q] = silu0(acc[bk][q]);
      D0_STAMP(3);
      const f32x4 bias1 = ld4(biasS + 16 + (bo & 4));
      f32x4 o[2] = {bias1, bias1};
      uint32_t lowm = n16 < 8 ? 0xffffffffu : 0u;
      asm volatile("" : "+v"(lowm));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float wa = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a1w[q]) & lowm);
#pragma unroll
        for (int p = 0; p < 2; ++p) o[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, h[p][q], o[p], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float wb = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a1w[q]) & ~lowm);
#pragma unroll
        for (int p = 0; p < 2; ++p) o[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb, h[p + 2][q], o[p], 0, 0, 0);
      }
      float v[2][4];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int t = t0 + 16 * (p + 2 * (kk >> 1));
#pragma unroll
        for (int q = 0; q < 4; ++q) v[p][q] = (o[p][q] + tile[(cb + q) * E0_PITCH + t + 2]) + sk[p][q];
      }
      D0_STAMP(4);
      if (a.tail_max == nullptr) {
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * E0_C * L, 0, (int)out_row_bytes, 0x00020000);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int l = l0 - 2 + t0 + 16 * (p + 2 * (kk >> 1));
          uint32_t off = ((uint32_t)cb * (uint32_t)L + (uint32_t)l) * 4u;
          off = live[p] ? off : 0x80000000u;
#pragma unroll
          for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v[p][q]), ro, off, (uint32_t)q * (uint32_t)L * 4u, 0);
        }
      } else {
        const f32x4 biasA = ld4(biasS + 32 + (bo & 4)), biasB = ld4(biasS + 40 + (bo & 4));
        f32x4 ta[2] = {biasA, biasA};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int p = 0; p < 2; ++p) ta[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(tfr[q], v[p][q], ta[p], 0, 0, 0);
        f32x4 u[2] = {biasB, biasB};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int p = 0; p < 2; ++p) u[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(tfr[4 + q], fmaxf(ta[p][q], 0.f), u[p], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) rmax[q] = fmaxf(rmax[q], fmaxf(live[0] ? u[0][q] : -INFINITY, live[1] ? u[1][q] : -INFINITY));
        // Softplus is non-decreasing: the raw values are reduced, the activation applied once per row segment
        const bool seg_end = nb != b || tix + 1 == last;
        float* slot = a.tail_max + ((size_t)b * tiles_per_row + tile_no) * E0_C;
        if (!seg_end) {
          if (tid < E0_C) slot[tid] = 0.f;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float m = rmax[q];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
            m = fmaxf(m, __shfl_xor(m, 32));
            if (n16 == 0 && kk < 2) red[wave * E0_C + cb + q] = m;
            rmax[q] = -INFINITY;
          }
          __syncthreads();
          if (tid < E0_C) {
            const float m = fmaxf(fmaxf(red[tid], red[E0_C + tid]), fmaxf(red[2 * E0_C + tid], red[3 * E0_C + tid]));
            const float e = __expf(m);
            slot[tid] = m > 20.f ? m : (m < -15.f ? e : __logf(1.f + e));      // torch.nn.Softplus(beta = 1, threshold = 20), as apply_act (conv1d.hip)
          }
        }
      }
    }
    D0_STAMP(5);
    if (STAMPS && threadIdx.x == 0) stamps[8 * blockIdx.x + 7] += 1;
#pragma unroll
    for (int u = 0; u < 5; ++u) cur[u] = nxt[u];
    b = nb;
    tile_no = ntile;
  }
#undef D0_STAMP
}

}  // namespace

// host: [Cin = 4][7][8] conv weights + bias behind a per-symbol layer symtab[15][st][4] + sym_bias[4] -> the composed conv per group of
// three taps over A C G T (t3: [ng][64][8], code = low bits of the three symbols | high bits << 3), its last tap alone (t1: [4][8]) and
// its bias (on positions whose 7 columns of the layer in front all lie inside the window)
void indel_enc0_compose(const float* fw, const float* fb, const float* symtab, const float* sym_bias, int st, std::vector<float>* t3,
                        std::vector<float>* t1, std::vector<float>* bias) {
  const int TT = 7 + st - 1, NG = (TT - 1) / 3;
  std::vector<double> T((size_t)4 * TT * 8, 0.0);       // [symbol A C G T][tap][co]
  for (int sy = 0; sy < 4; ++sy)
    for (int k2 = 0; k2 < 7; ++k2)
      for (int k = 0; k < st; ++k)
        for (int ci = 0; ci < 4; ++ci)
          for (int co = 0; co < 8; ++co)
            T[((size_t)sy * TT + k2 + k) * 8 + co] += (double)fw[(ci * 7 + k2) * 8 + co] * (double)symtab[((size_t)sy * st + k) * 4 + ci];
  t3->assign((size_t)NG * 64 * 8, 0.f);
  for (int g = 0; g < NG; ++g)
    for (int code = 0; code < 64; ++code)
      for (int co = 0; co < 8; ++co) {
        double acc = 0.0;
        for (int u = 0; u < 3; ++u) {
          const int sy = ((code >> u) & 1) | (((code >> (3 + u)) & 1) << 1);
          acc += T[((size_t)sy * TT + 3 * g + u) * 8 + co];
        }
        (*t3)[((size_t)g * 64 + code) * 8 + co] = (float)acc;
      }
  t1->assign(4 * 8, 0.f);
  for (int sy = 0; sy < 4; ++sy)
    for (int co = 0; co < 8; ++co) (*t1)[sy * 8 + co] = (float)T[((size_t)sy * TT + TT - 1) * 8 + co];
  bias->assign(8, 0.f);
  for (int co = 0; co < 8; ++co) {
    double acc = fb[co];
    for (int k2 = 0; k2 < 7; ++k2)
      for (int ci = 0; ci < 4; ++ci) acc += (double)fw[(ci * 7 + k2) * 8 + co] * (double)sym_bias[ci];
    (*bias)[co] = (float)acc;
  }
}

// diagnostic (tools/phase_stamps_indel_l0.py): MURAL_DEBUG_CB_STAMP_ONLY = enc | dec stamps only that launch
static unsigned long long* l0_stamps(bool enc) {
  if (g_cb8_stamps)
    if (const char* only = dev_env("MURAL_DEBUG_CB_STAMP_ONLY"))
      if ((only[0] == 'e') != enc) return nullptr;
  return g_cb8_stamps;
}

bool indel_enc0_supported(const ConvBlockArgs& a) {
  const bool off = dev_env("MURAL_INDEL_ENC0") && atoi(dev_env("MURAL_INDEL_ENC0")) == 0;      // (read per launch: the tests switch it)
  if (a.sym_in && !(a.f_in && (int64_t)a.B * a.Lf < (int64_t(1) << 40))) return false;      // (byte source: the dense window backs it)
  return !off && a.C == 8 && a.symtab && a.e0_t3 && a.e0_t1 && a.e0_bias && a.Cf == 4 && a.f_up == 1 && (a.sym_taps == 7 || a.sym_taps == 1) &&
         a.tail_max == nullptr && a.res2 == nullptr && a.Lf == a.L;
}

int launch_indel_enc0(const ConvBlockArgs& a, hipStream_t stream) {
  const bool down = a.d_out != nullptr;
  if (down) MURAL_REQUIRE(a.d_w && a.d_b && (a.L & 3) == 0 && a.d_L == (a.L - 1) / 4 + 1, "level-0 launch: bad geometry of the strided conv behind it");
  const int outw = down ? E0_OUT_DOWN : E0_OUT;
  const int tiles_per_row = (a.L + outw - 1) / outw;
  const long long total = (long long)a.B * tiles_per_row;
  if (total == 0) return MURAL_OK;
  unsigned long long* stamps = l0_stamps(true);
  const bool bytes = a.sym_in != nullptr;
  if (bytes) stamps = nullptr;      // (the byte-source instances carry no phase stamps)
  const int v = (a.sym_taps == 7 ? 0 : 1) + (stamps ? 2 : 0) + (down ? 4 : 0) + (bytes ? 8 : 0);      // the instance that is launched
  static int wg_per_cu[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    MURAL_HIP_CHECK(hipGetDevice(&dev));
    MURAL_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount;
  }
#define MURAL_E0_CASES(X) \
  switch (v) {            \
    case 0: X(13, false, false); break; \
    case 1: X(7, false, false); break;  \
    case 2: X(13, true, false); break;  \
    case 3: X(7, true, false); break;   \
    case 4: X(13, false, true); break;  \
    case 5: X(7, false, true); break;   \
    case 6: X(13, true, true); break;   \
    case 7: X(7, true, true); break;    \
    case 8: X(13, false, false, true); break; \
    case 9: X(7, false, false, true); break;  \
    case 12: X(13, false, true, true); break; \
    default: X(7, false, true, true); break;  \
  }
  if (wg_per_cu[v] == 0) {
    int n = 0;
#define MURAL_E0_OCC(...) MURAL_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, indel_enc0_kernel<__VA_ARGS__>, 256, 0))
    MURAL_E0_CASES(MURAL_E0_OCC)
#undef MURAL_E0_OCC
    wg_per_cu[v] = n > 7 ? 7 : (n > 0 ? n : 1);
  }
  static const int cap = dev_env("MURAL_INDEL_ENC0_WGS") ? atoi(dev_env("MURAL_INDEL_ENC0_WGS")) : 0;      // experiment: workgroups per CU
  const long long want = (long long)cus * (cap > 0 ? cap : wg_per_cu[v]);
  const dim3 grid((unsigned)(total < want ? total : want));
#define MURAL_E0(...) hipLaunchKernelGGL((indel_enc0_kernel<__VA_ARGS__>), grid, dim3(256), 0, stream, a, a.w5, a.b5, a.w1, a.b1, tiles_per_row, total, stamps)
  MURAL_E0_CASES(MURAL_E0)
#undef MURAL_E0
#undef MURAL_E0_CASES
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

bool indel_dec0_supported(const ConvBlockArgs& a) {
  const bool off = dev_env("MURAL_INDEL_DEC0") && atoi(dev_env("MURAL_INDEL_DEC0")) == 0;
  return !off && a.C == 8 && a.f_in != nullptr && a.symtab == nullptr && a.f_pw != nullptr && a.Cf == 16 && a.f_up == 4 && (a.L & 3) == 0 && a.Lf * 4 == a.L &&
         a.x == nullptr && (a.tail_max != nullptr || a.out != nullptr) && (a.tail_max == nullptr || (a.ta_w && a.ta_b && a.tb_w && a.tb_b));
}

int launch_indel_dec0(const ConvBlockArgs& a, hipStream_t stream) {
  const int tiles_per_row = (a.L + D0_OUT - 1) / D0_OUT;
  const long long total = (long long)a.B * tiles_per_row;
  if (total == 0) return MURAL_OK;
  unsigned long long* const stamps = l0_stamps(false);
  const int v = stamps ? 1 : 0;
  static int wg_per_cu[2] = {0, 0}, cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    MURAL_HIP_CHECK(hipGetDevice(&dev));
    MURAL_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount;
  }
  if (wg_per_cu[v] == 0) {
    int n = 0;
    if (v == 0) MURAL_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, indel_dec0_kernel<false>, 256, 0));
    else MURAL_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, indel_dec0_kernel<true>, 256, 0));
    wg_per_cu[v] = n > 0 ? n : 1;
  }
  static const int cap = dev_env("MURAL_INDEL_DEC0_WGS") ? atoi(dev_env("MURAL_INDEL_DEC0_WGS")) : 0;      // experiment: workgroups per CU
  const long long want = (long long)cus * (cap > 0 ? cap : wg_per_cu[v]);
  const dim3 grid((unsigned)(total < want ? total : want));
  if (stamps) hipLaunchKernelGGL(indel_dec0_kernel<true>, grid, dim3(256), 0, stream, a, a.w5, a.b5, a.w1, a.b1, tiles_per_row, total, stamps);
  else hipLaunchKernelGGL(indel_dec0_kernel<false>, grid, dim3(256), 0, stream, a, a.w5, a.b5, a.w1, a.b1, tiles_per_row, total, stamps);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural
