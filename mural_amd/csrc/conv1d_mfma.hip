// Generic fp32 Conv1d as an implicit GEMM on v_mfma_f32_16x16x4_f32 (gfx950): the same contract as conv1d.hip's vector-ALU
// kernel (Conv1dArgs: stride, nearest-neighbour upsampling, per-channel pre-op, bias, activation, two residuals; [B][C][L]
// tensors, weights [Cin][K][Cout]) for layers with at least 16 output channels.
//
// Why: the deep U-Net levels of the INDEL model (model_indel.py:21-176: 32..96 channels on rows of 80 / 16 / 8 columns) are
// GEMMs with M = Cout, N = batch x columns, K = Cin x taps whose WEIGHTS outweigh a row's activations.  The direct kernel
// gives a workgroup 64 output columns and streams every weight through the scalar cache once per workgroup -- 100-150 us per
// launch for a few MFLOP per position.  Here the N axis is the flattened (row, column) index of TR whole rows (short rows) or a
// 64-column segment of one row (long rows).  The four waves tile the (output-channel blocks) x (16-column blocks) grid 2 x 2
// (4 x 1 over the columns when there is one channel block): per (4 input channels, tap) step a wave loads its <= 3 weight
// fragments as coalesced 64-byte global loads (shared by the other waves through L1 / L2, one step ahead of their use) and its
// <= 2 input fragments from LDS, and issues their outer product of MFMAs.
//
//   A (16 x 4)  = W[co = 16 mb + (lane & 15)][ci0 + (lane >> 4)][tap]      global, [Cin][K][Cout] layout: 16 lanes = 64 contiguous bytes
//   B (4 x 16)  = X[ci0 + (lane >> 4)][column (lane & 15) of the block, tap] LDS tile [row][ci][span], pre-op / upsampling / zero
//                                                                            padding applied while staging
//   D (16 x 16) : lane holds rows 4 (lane >> 4) .. + 3 of column (lane & 15)
#include <vector>

#include "conv1d.h"

namespace mural {
namespace {

typedef float g4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float cg_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_SILU: return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
    case ACT_SOFTPLUS: {
      const float e = __expf(v);
      return v > 20.f ? v : (v < -15.f ? e : __logf(1.f + e));
    }
    default: return v;
  }
}

struct ConvGArgs {
  Conv1dArgs a;
  int TR;        // batch rows per tile (row regime), 1 in the segment regime
  int seg;       // 1: a tile is a segc-column segment of one row (grid = segments x rows)
  int segc;      // columns of a segment-regime tile (seg_cols)
  int span;      // staged input columns per (row, channel)
  int span_p;    // LDS row stride (== 16 mod 32: the four channel rows of a fragment read land on disjoint bank halves)
  FastDiv d_span, d_cin, d_up, d_lout;   // the staging loop's index arithmetic costs more than its loads with hardware division
  int Mrows;     // GEMM rows: Cout, or Cout * phases in the polyphase form
  int Lcols;     // GEMM columns per batch row: Lout, or the source length Lin in the polyphase form
  int ph;        // phases (1: plain)
  FastDiv d_ph;
  int mslice;    // 16-row blocks per workgroup
  unsigned long long* stamps;      // diagnostic (tools/phase_stamps_conv1d.py): 5 s_memrealtime values per workgroup, or nullptr
};

constexpr int CG_MSLICE = 6;    // at most this many 16-row blocks per workgroup; more rows go to further workgroups along grid.z

constexpr int CG_NBW_SEG = 3;   // segment regime: at most this many 16-column blocks per wave (2, 3 and 4 measure the same over the
                                // U-Net forward, within the +-2 % run-to-run spread; 3 keeps the accumulators at 36 registers)

// 16-column blocks per wave of a segment-regime tile: CG_NBW_SEG for long rows, fewer when a whole row of 65..128 columns fits
// (80-column rows on a 128 / 256-column tile would spend 40..70 % of their MFMAs on padding)
inline int seg_nbw(int Mrows, int Lcols) {
  const int wn = Mrows > 16 ? 2 : 4;
  int nbw = (Lcols + 16 * wn - 1) / (16 * wn);
  if (nbw < 2) nbw = 2;
  return nbw < CG_NBW_SEG ? nbw : CG_NBW_SEG;
}
// columns of a segment-regime tile: the column-block groups (4 with one channel block, else 2) x blocks per wave x 16
inline int seg_cols(int Mrows, int Lcols) { return (Mrows > 16 ? 2 : 4) * seg_nbw(Mrows, Lcols) * 16; }

template <int MW, int KT, int NBW>     // MW: output-channel blocks per wave (1..3); NBW: 16-column blocks per wave
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(const ConvGArgs g, const float* __restrict__ wt, const float* __restrict__ bias) {
  extern __shared__ float tile[];   // [TR][Cin][span_p]
  const Conv1dArgs& a = g.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4;
  const int b0 = g.seg ? blockIdx.y : blockIdx.x * g.TR;
  const int SEG = g.segc;
  const int l0 = g.seg ? blockIdx.x * SEG : 0;
  const int in0 = l0 * a.stride - a.pad;      // virtual (upsampled) input index of the first staged column
  const int Lv = a.Lin * a.up;
  if (g.stamps && tid == 0) g.stamps[5 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + 0] = __builtin_amdgcn_s_memrealtime();
  // ---- stage: all 256 threads over the flattened (row, channel, column) index, UN loads of a thread REALLY in flight: every load
  //      goes through a range-checked descriptor with its offset chosen by a select (a refused element -- zero padding, a row past
  //      the batch -- aims past the descriptor and reads 0), the pre-op runs on the whole round afterwards.  (The loads used to sit
  //      behind `if (in range)` with the pre-op inside the branch: the compiler then waits for each load inside its own branch, a
  //      global round trip per element instead of one per round -- 2-7 us of a 10-18 us launch, tools/phase_stamps_conv1d.py.)
  {
    const int total = g.TR * a.Cin * g.span;
    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((size_t)a.B * a.Cin * a.Lin * 4), 0x00020000);
    const bool has_pre = a.pre_s != nullptr || a.pre_t != nullptr || a.pre_relu != 0;      // wave-uniform
    const __amdgpu_buffer_rsrc_t rps =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pre_s ? a.pre_s : a.in), 0, a.pre_s ? a.Cin * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rpt =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pre_t ? a.pre_t : a.in), 0, a.pre_t ? a.Cin * 4 : 0, 0x00020000);
    constexpr int UN = 8;      // (16 cost 160 registers -- two workgroups per CU -- for the index arithmetic in flight with the loads)
    for (int i0 = tid; i0 < total; i0 += 256 * UN) {
      float x[UN];
      int dsto[UN];
      uint32_t okmask = 0;
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + 256 * u;
        const int rc = (int)g.d_span.divnb((uint32_t)i), j = i - rc * g.span;
        const int r = (int)g.d_cin.divnb((uint32_t)rc), ci = rc - r * a.Cin;
        const int v = in0 + j;
        const bool in_tile = i < total;
        const bool ok = in_tile & (b0 + r < a.B) & (v >= 0) & (v < Lv);
        uint32_t off = (uint32_t)(((b0 + r) * a.Cin + ci) * a.Lin + (int)g.d_up.divnb((uint32_t)(ok ? v : 0))) * 4u;
        asm volatile("" : "+v"(off));      // (a select below, not a branch)
        x[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, ok ? off : 0x80000000u, 0, 0));
        dsto[u] = in_tile ? rc * g.span_p + j : -1;
        okmask |= ok ? (1u << u) : 0u;
      }
      if (has_pre) {      // x' = pre_s * (relu?)(x) + pre_t on in-range values, zero padding stays zero
        float ps[UN], pt[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int i = i0 + 256 * u;
          const int rc = (int)g.d_span.divnb((uint32_t)i);
          const int ci = rc - (int)g.d_cin.divnb((uint32_t)rc) * a.Cin;
          const uint32_t co = (okmask >> u) & 1u ? (uint32_t)ci * 4u : 0x80000000u;
          ps[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rps, co, 0, 0));
          pt[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rpt, co, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const float xv = a.pre_relu ? fmaxf(x[u], 0.f) : x[u];
          const float sc = a.pre_s ? ps[u] : 1.f;
          x[u] = (okmask >> u) & 1u ? fmaf(sc, xv, pt[u]) : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (dsto[u] >= 0) tile[dsto[u]] = x[u];
    }
  }
  __syncthreads();
  if (g.stamps && tid == 0) g.stamps[5 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + 1] = __builtin_amdgcn_s_memrealtime();
  const int ncols = g.seg ? (g.Lcols - l0 < SEG ? g.Lcols - l0 : SEG) : g.TR * g.Lcols;
  constexpr int K = KT;
  // wave -> (channel-block group wm, column-block group wn)
  const int mb_all = (g.Mrows + 15) >> 4;
  const int m_base = blockIdx.z * g.mslice;                 // this workgroup's slice of the channel blocks
  const int mb_total = mb_all - m_base < g.mslice ? mb_all - m_base : g.mslice;
  const int WM = mb_all >= 2 ? 2 : 1, WN = 4 / WM;
  const int wm = wave % WM, wn = wave / WM;
  const int m_first = wm * MW;                              // this wave's channel blocks: m_first .. m_first + MW - 1 (< mb_total)
  const int nbw = g.seg ? NBW : 4 / WN;                     // row regime: 64 columns = 4 blocks over the WN groups
  // per owned column block: LDS offset of the lane's column, validity
  int boff[NBW], brow[NBW], bcol[NBW];
  bool bval[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int blk = wn * nbw + j;
    const int col = 16 * blk + n16;
    bval[j] = j < nbw && col < ncols;
    int r = 0, l = bval[j] ? col : 0;
    if (!g.seg) {
      r = (int)g.d_lout.div((uint32_t)l);
      l -= r * g.Lcols;
    }
    brow[j] = r;
    bcol[j] = l;
    boff[j] = (r * a.Cin + kk) * g.span_p + l * a.stride;
  }
  // weight fragment addresses (bytes into wt, lane part): channel blocks past Cout aim past the descriptor and read zeros
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, (int)((size_t)a.Cin * KT * g.Mrows * 4), 0x00020000);
  uint32_t woff[MW];
#pragma unroll
  for (int m = 0; m < MW; ++m) {
    const int co = 16 * (m_base + m_first + m) + n16;
    const bool wok = m_first + m < mb_total && co < g.Mrows;
    woff[m] = wok ? (uint32_t)(kk * KT * g.Mrows + co) * 4u : 0x80000000u;
  }
  g4 acc[MW][NBW];
#pragma unroll
  for (int m = 0; m < MW; ++m)
#pragma unroll
    for (int j = 0; j < NBW; ++j) acc[m][j] = g4{0.f, 0.f, 0.f, 0.f};
  // weight fragments travel CG_PF steps (of 4 input channels) ahead of their MFMAs: a step's MFMAs are shorter than a global round
  // trip whenever a wave owns few blocks (the deep levels: a dozen steps of <= 30 MFMAs), and one step of lead left every step
  // waiting out most of that round trip -- 15-25 us per launch for a microsecond of arithmetic.  The wave-uniform part of the
  // address rides in the scalar offset: one address register per channel block.
  constexpr int CG_PF = 48 / (K * MW) >= 16 ? 16 : (48 / (K * MW) >= 2 ? 48 / (K * MW) : 2);      // about 48 registers of lead
  static_assert((CG_PF - 1) * K * MW <= 63, "newer loads in flight at a use must fit the vmcnt range");
  float wq[CG_PF][K][MW];
  auto fetch = [&](float (&dst)[K][MW], int ci) {
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
      for (int m = 0; m < MW; ++m)
        dst[t][m] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, woff[m], (uint32_t)((ci * K + t) * g.Mrows) * 4u, 0));
  };
  // blocks of this wave that hold any column of the tile (a short problem's tile is 16 or 32 columns wide: three of the four column
  // blocks of the 64-column layout, and the waves that own nothing else, would spend their MFMAs on padding -- 4 of the 6 us of the
  // MFMA phase at 16 columns)
  const int nv = __builtin_amdgcn_readfirstlane(min(nbw, max(0, (ncols - 16 * wn * nbw + 15) >> 4)));
  if (nv > 0 && m_first < mb_total) {
#pragma unroll
    for (int p = 0; p < CG_PF; ++p) fetch(wq[p], 4 * p < a.Cin ? 4 * p : a.Cin - 4);
    for (int cb = 0; cb < a.Cin; cb += 4 * CG_PF) {
#pragma unroll
      for (int p = 0; p < CG_PF; ++p) {
        const int ci0 = cb + 4 * p;
        if (ci0 >= a.Cin) break;
        float ac[K][MW];
#pragma unroll
        for (int t = 0; t < K; ++t)
#pragma unroll
          for (int m = 0; m < MW; ++m) ac[t][m] = wq[p][t][m];
        // (always fetched, the last steps re-fetch the final one: a load under a branch makes the compiler's wait-count bookkeeping
        // merge the two paths pessimistically -- it then waits for the whole queue at the end of every round of CG_PF steps)
        fetch(wq[p], ci0 + 4 * CG_PF < a.Cin ? ci0 + 4 * CG_PF : a.Cin - 4);
        const float* trow = tile + (size_t)ci0 * g.span_p;
#pragma unroll
        for (int t = 0; t < K; ++t) {
          float bv[NBW];
#pragma unroll
          for (int j = 0; j < NBW; ++j) bv[j] = trow[boff[j] + t];
#pragma unroll
          for (int j = 0; j < NBW; ++j)
            if (j < nv) {
#pragma unroll
              for (int m = 0; m < MW; ++m) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[t][m], bv[j], acc[m][j], 0, 0, 0);
            }
        }
      }
    }
  }
  if (g.stamps && tid == 0) {
    asm volatile("s_nop 0" ::: "memory");
    g.stamps[5 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + 2] = __builtin_amdgcn_s_memrealtime();
  }
  // ---- epilogue: bias, activation, residuals, straight from the accumulators.  Plain: 16 lanes write 16 consecutive columns of
  //      one output channel.  Polyphase: row co * ph + p of source column i is output column ph * i + p of channel co, so a lane's
  //      four rows are (mostly) four consecutive output columns.  (Measured: routing the tile through LDS to store whole 16-byte
  //      runs is no faster -- the kernel is bound by its serial stage / MFMA / store phases at 3 workgroups per CU, not by the
  //      width of the stores.)
  //      Every load of the epilogue -- the bias of a lane's rows, both residuals of all its elements -- is issued up front through
  //      range-checked descriptors (an absent tensor has a descriptor of 0 bytes and reads 0; a refused element aims past it): with
  //      `if (res1) v += res1[o]` per element the compiler put a full wait behind each of the up to 108 loads of a lane, 15-20 us
  //      of serial round trips per workgroup on launches whose arithmetic takes one.
  {
    const uint32_t obytes = (uint32_t)((size_t)a.B * a.Cout * a.Lout * 4);      // (launch_conv1d_mfma refuses tensors of 2 GB and more)
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias ? bias : wt), 0, bias ? a.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res1 ? a.res1 : wt), 0, a.res1 ? (int)obytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res2 ? a.res2 : wt), 0, a.res2 ? (int)obytes : 0, 0x00020000);
    const bool any_res = a.res1 != nullptr || a.res2 != nullptr;      // wave-uniform
    // bias of all the lane's rows first (one round trip), then one channel block at a time: its offsets, its residual loads (all in
    // flight together), bias / activation, its stores -- a block's 12 x 3 temporaries instead of the tile's 36 x 3 keep three
    // workgroups per CU resident on the long rows
    float bz[MW][4];
    int cos[MW][4], subs[MW][4];
    bool roks[MW][4];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * (m_base + m_first + m) + 4 * kk + q;
        const bool rok = (m_first + m < mb_total) & (row < g.Mrows);
        int co = row, sub = 0;
        if (g.ph > 1) {
          co = (int)g.d_ph.div((uint32_t)row);
          sub = row - co * g.ph;
        }
        cos[m][q] = co; subs[m][q] = sub; roks[m][q] = rok;
        bz[m][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, rok ? (uint32_t)co * 4u : 0x80000000u, 0, 0));
      }
#pragma unroll
    for (int m = 0; m < MW; ++m) {
      uint32_t off[NBW][4];
      float e1[NBW][4], e2[NBW][4], v[NBW][4];
#pragma unroll
      for (int j = 0; j < NBW; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = roks[m][q] & bval[j] & (b0 + brow[j] < a.B);
          const int lo = (l0 + bcol[j]) * g.ph + subs[m][q];
          uint32_t o = (uint32_t)(((b0 + brow[j]) * a.Cout + cos[m][q]) * a.Lout + lo) * 4u;
          asm volatile("" : "+v"(o));      // (a select below, not a branch)
          off[j][q] = ok ? o : 0x80000000u;
          e1[j][q] = 0.f;
          e2[j][q] = 0.f;
        }
      if (any_res) {
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            e1[j][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr1, off[j][q], 0, 0));
            e2[j][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr2, off[j][q], 0, 0));
          }
      }
      // bias and activation, the activation chosen once per block (a switch per element is four branches and the code of every case
      // in the instruction stream 36 times: these launches are short enough for instruction fetch to show)
#pragma unroll
      for (int j = 0; j < NBW; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = acc[m][j][q] + bz[m][q];
      if (a.act == ACT_RELU) {
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) v[j][q] = fmaxf(v[j][q], 0.f);
      } else if (a.act == ACT_SILU) {
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) v[j][q] = cg_act(v[j][q], ACT_SILU);
      } else if (a.act == ACT_SOFTPLUS) {
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) v[j][q] = cg_act(v[j][q], ACT_SOFTPLUS);
      }
#pragma unroll
      for (int j = 0; j < NBW; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v[j][q] + e1[j][q] + e2[j][q]), ro, off[j][q], 0, 0);
    }
  }
  if (g.stamps && tid == 0) {
    g.stamps[5 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + 3] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g.stamps[5 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + 4] = __builtin_amdgcn_s_memrealtime();
  }
}

using ConvGFn = void (*)(const ConvGArgs, const float*, const float*);

template <int MW, int NBW>
ConvGFn pick_taps(int K) {
  switch (K) {
    case 1: return conv1d_mfma_kernel<MW, 1, NBW>;
    case 3: return conv1d_mfma_kernel<MW, 3, NBW>;
    case 5: return conv1d_mfma_kernel<MW, 5, NBW>;
    case 7: return conv1d_mfma_kernel<MW, 7, NBW>;
    default: return nullptr;
  }
}

template <int NBW>
ConvGFn pick_mw(int mb_total, int K, int mslice) {
  const int slice = mb_total < mslice ? mb_total : mslice;
  const int mw = mb_total >= 2 ? (slice + 1) / 2 : 1;
  switch (mw) {
    case 1: return pick_taps<1, NBW>(K);
    case 2: return pick_taps<2, NBW>(K);
    default: return pick_taps<3, NBW>(K);
  }
}

ConvGFn pick(int mb_total, int K, int seg, int mslice) {
  if (!seg || seg == 2) return pick_mw<2>(mb_total, K, mslice);      // seg: 0 = row regime, else blocks per wave of the segment tile
  return pick_mw<CG_NBW_SEG>(mb_total, K, mslice);
}

struct ConvGArgs;
size_t lds_bytes(const ConvGArgs& g);

int pad_span(int span) {      // smallest stride >= span that is 16 mod 32
  const int rem = span & 31;
  return rem <= 16 ? span - rem + 16 : span - rem + 48;
}

bool plan(const Conv1dArgs& a, ConvGArgs* g) {
  const int ph = a.phases > 1 ? a.phases : 1;
  if (a.Cin % 4 != 0 || a.Cin < 4 || (a.K != 1 && a.K != 3 && a.K != 5 && a.K != 7)) return false;
  if (ph > 1 && (a.stride != 1 || a.up != 1 || a.Lout != a.Lin * ph || a.pre_s || a.pre_t || a.pre_relu)) return false;
  g->a = a;
  g->ph = ph;
  g->Mrows = a.Cout * ph;
  g->Lcols = ph > 1 ? a.Lin : a.Lout;
  if (g->Mrows < 16 || (ph == 1 && g->Mrows > 16 * CG_MSLICE)) return false;
  if ((uint64_t)a.B * a.Cout * a.Lout * 4 >= (1ull << 31) || (uint64_t)a.B * a.Cin * a.Lin * 4 >= (1ull << 31)) return false;      // buffer descriptors address 2 GB
  if (g->Lcols > 64) {          // longer than the 64-column tile of the row regime
    g->seg = 1;
    g->TR = 1;
    g->segc = seg_cols(g->Mrows, g->Lcols);
    g->span = (g->segc - 1) * a.stride + a.K;
  } else {
    g->seg = 0;
    g->span = (g->Lcols - 1) * a.stride + a.K;
    int tr = 64 / g->Lcols;
    if (tr < 1) tr = 1;
    while (tr > 1 && (size_t)tr * a.Cin * pad_span(g->span) * 4 > 64 * 1024) --tr;
    // few rows: smaller tiles, more workgroups -- a launch of a few dozen workgroups is a chain of staging round trips, not arithmetic
    while (tr > 1 && (a.B + tr - 1) / tr < 128 && (tr / 2) * g->Lcols >= 16) tr /= 2;
    g->TR = tr;
  }
  g->span_p = pad_span(g->span);
  g->d_span = FastDiv::make((uint32_t)g->span);
  g->d_cin = FastDiv::make((uint32_t)a.Cin);
  g->d_up = FastDiv::make((uint32_t)a.up);
  g->d_lout = FastDiv::make((uint32_t)g->Lcols);
  g->d_ph = FastDiv::make((uint32_t)ph);
  g->mslice = CG_MSLICE;
  {
    // a short problem (the deep levels: a few dozen tiles) spreads its channel blocks over more workgroups: a workgroup's MFMA chain
    // and the launch's tail shrink with the slice, the re-staged input tile is a few KB
    const int mb_all = (g->Mrows + 15) / 16;
    const int64_t tiles = g->seg ? (int64_t)((g->Lcols + g->segc - 1) / g->segc) * a.B : (a.B + g->TR - 1) / g->TR;
    while (g->mslice > 2 && tiles * ((mb_all + g->mslice - 1) / g->mslice) < 128) g->mslice = g->mslice > 3 ? 3 : 2;
    // ... and down to one block per wave while the launch stays within three workgroups per CU: a wave with one block fetches its
    // weights a dozen steps ahead (48 registers of lead / K) and its MFMA phase stops being a chain of weight round trips
    if (tiles * ((mb_all + 1) / 2) <= 768) g->mslice = 2;
  }
  return lds_bytes(*g) <= 96 * 1024;
}

size_t lds_bytes(const ConvGArgs& g) { return (size_t)g.TR * g.a.Cin * g.span_p * 4; }

}  // namespace

// Host: polyphase weights of conv_K(upsample_up(x)) with zero padding (K - 1) / 2 on the upsampled signal.  Output column up * i + p
// reads the upsampled columns up * i + p + t - pad, i.e. the source columns i + floor((p + t - pad) / up): taps with the same
// offset d are summed.  in: [Cin][K][Cout]; out: [Cin][KJ][Cout * up] with row co * up + p; *pad_out = -d_min.
void conv1d_phase_weights(const float* w, int Cin, int K, int Cout, int up, std::vector<float>* out, int* KJ, int* pad_out) {
  const int pad = (K - 1) / 2;
  auto fl = [&](int v) { return v >= 0 ? v / up : -((-v + up - 1) / up); };
  const int dmin = fl(-pad), dmax = fl(up - 1 + K - 1 - pad);
  const int kj = dmax - dmin + 1;
  out->assign((size_t)Cin * kj * Cout * up, 0.f);
  for (int ci = 0; ci < Cin; ++ci)
    for (int t = 0; t < K; ++t)
      for (int p = 0; p < up; ++p) {
        const int d = fl(p + t - pad) - dmin;
        for (int co = 0; co < Cout; ++co)
          (*out)[((size_t)ci * kj + d) * Cout * up + (size_t)co * up + p] += w[((size_t)ci * K + t) * Cout + co];
      }
  *KJ = kj;
  *pad_out = -dmin;
}

bool conv1d_mfma_supported(const Conv1dArgs& a) {
  ConvGArgs g;
  return plan(a, &g);
}

static unsigned long long* g_conv1d_stamps = nullptr;
void conv1d_mfma_set_stamps(unsigned long long* p) { g_conv1d_stamps = p; }

int launch_conv1d_mfma(const Conv1dArgs& a, hipStream_t stream) {
  if (a.B == 0 || a.Lout == 0) return MURAL_OK;
  ConvGArgs g;
  MURAL_REQUIRE(plan(a, &g), "conv1d (MFMA): unsupported geometry");
  g.stamps = g_conv1d_stamps;
  const size_t lds = lds_bytes(g);
  const int mb_all = (g.Mrows + 15) / 16;
  ConvGFn fn = pick(mb_all, a.K, g.seg ? seg_nbw(g.Mrows, g.Lcols) : 0, g.mslice);
  MURAL_REQUIRE(fn, "conv1d (MFMA): no kernel for %d taps", a.K);
  if (lds > 64 * 1024)
    MURAL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int segc = g.segc;
  const unsigned mz = (unsigned)((mb_all + g.mslice - 1) / g.mslice);
  const dim3 grid = g.seg ? dim3((g.Lcols + segc - 1) / segc, a.B, mz) : dim3((a.B + g.TR - 1) / g.TR, 1, mz);
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, stream, g, a.wt, a.bias);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural
