// Wave-private form of the SNV tower kernel for the throughput launches (gfx950 / CDNA4).
//
// Reference semantics: MuRaL/model/model_snv.py:473-513 (tower part of Network2.forward), :794-812 (ResBlock), :515-523 (head) --
// the same layers, layer roles, fragments, k order and epilogue arithmetic as snv_towers_fused (snv_tower.hip), which stays the
// kernel of the latency-shaped small calls and of the debug dumps.  What changes is who owns what:
//
//   * a WAVE owns whole sites: all 32 output channels (both 16-row M-blocks) of every column of its own Pw sites.  A layer's
//     output feeds only the same wave's next layer, so there is NO workgroup barrier between layers (snv_towers_fused pays
//     one per layer: ~1.5 k cycles of barrier + pipeline fill against 6.5 k cycles of MFMAs); the waves of a CU drift apart
//     and one wave's entry / pooling / head phases run under the other wave's MFMAs on the same SIMD.
//   * both M-blocks share every B operand: one ds_read_b128 feeds 8 MFMAs instead of 4, and the two M-blocks ARE the two
//     independent accumulator chains the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 asks for.
//   * the layer runs IN PLACE on one LDS image per wave: a wave's LDS operations execute in order, block b's output columns
//     16b+1 .. 16b+16 are written after the tap-0 operands of block b+1 (the only later read that touches them) were read.
//     One image of <= 146 columns x 128 B = 18.7 KB per wave: four waves per workgroup, two workgroups per CU.
//   * weight fragments: 48 VGPRs per layer (both M-blocks), single-buffered -- the next layer's fragments are requested into
//     the registers of each tap group right after the LAST block's MFMAs of that group were issued (>= 500 cycles before their
//     first use, against ~200 cycles of L2-hit latency).
//
// Launches (per chunk of <= 131072 sites, like the workgroup-tile form): (large | mid) x (first conv stage | short stages + fc
// (+ head)), with Pw = 1 | 2 sites per wave in the first conv stage (136 / 137 columns = 9 blocks) and 6 | 5 in the short stages.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "snv_tower_conv.h"

namespace mural {

template <int M>
using ModeTag = std::integral_constant<int, M>;
// geometry of one tower for a window of Lwin columns with Pw sites per wave (the arithmetic of plan_wave_geometry); L[i] < 1: no fit
__host__ __device__ constexpr TowerGeom wave_tower_geom(int tw, int Lwin, int Pw) {
  constexpr int pools[2][3][3] = {{{15, 15, 7}, {7, 7, 3}, {3, 3, 1}}, {{3, 3, 1}, {3, 3, 1}, {3, 3, 1}}};
  TowerGeom g{};
  g.L1 = tw == 0 ? Lwin : 2 * SNV_MID_HALF + 1;
  g.col0 = tw == 0 ? 0 : Lwin / 2 - SNV_MID_HALF;
  int L = g.L1;
  for (int i = 0; i < 3; ++i) {
    g.pk[i] = pools[tw][i][0];
    g.ps[i] = pools[tw][i][1];
    g.pp[i] = pools[tw][i][2];
    L = (L + 2 * g.pp[i] - g.pk[i]) / g.ps[i] + 1;
    g.L[i] = L;
    if (L < 1) return g;
    g.Sc[i] = L + 1;
    g.NC[i] = 1 + Pw * g.Sc[i];
    g.nb[i] = (g.NC[i] + 15) / 16;
    g.dL[i] = FastDiv::make((uint32_t)L);
    g.dSc[i] = FastDiv::make((uint32_t)g.Sc[i]);
  }
  return g;
}
constexpr int SHIP_LWIN = 2001;      // distal_radius 1000: the window the specialised instantiations are compiled for

constexpr int MODE_GENERIC = -1;      // epilogue role from the run-time LayerK (layers that share one code body)
constexpr int TW_NBW = 9;            // 16-column blocks a wave owns at most
constexpr int TW_DUMP = 128;          // floats per wave behind its LDS regions: 32 16-byte dump slots (lane & 31)
constexpr uint32_t TW_BLK = 2048u;   // bytes between consecutive blocks of the image (16 columns x 128 B; the swizzle key has period 16)

struct WaveAddr {
  uint32_t rd[6];    // byte offset of B-operand chunk (tap t, half h) for block 0: rd[2t+h]
  uint32_t wr[2];    // byte offset of this lane's output chunk of M-block 0 / 1 for block 0
  uint32_t vmask;    // bit b: this lane's column of block b carries data (not separator / padding)
  uint32_t dump;     // byte offset (from the image) of this lane's dump slot: where the lanes of gap columns store (TW_DUMP)
};

__device__ __forceinline__ WaveAddr wave_setup(const TowerGeom& g, int st, int Pw, int n16, int kk, uint32_t dump0) {
  WaveAddr a;
  a.dump = dump0 + 16u * (uint32_t)((16 * kk + n16) & 31);
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) a.rd[2 * t + h] = 4u * (uint32_t)lds_off(n16 + t, 4 * h + kk);
  a.wr[0] = 4u * (uint32_t)lds_off(n16 + 1, kk);
  a.wr[1] = 4u * (uint32_t)lds_off(n16 + 1, 4 + kk);
  a.vmask = 0;
#pragma unroll
  for (int b = 0; b < TW_NBW; ++b) a.vmask |= col_is_data(n16 + 16 * b, g.dSc[st], g.Sc[st], g.L[st], Pw) ? (1u << b) : 0u;
  return a;
}

// four k-steps (4 g .. 4 g + 3) of both M-blocks' fragments: two 16-byte loads per lane; w4 = layer base of wfrag4 + 4 * lane
// (wf = descriptor of the tower's wfrag4 table, layer_bytes = byte offset of the layer in it: scalar; lane16 = 16 * lane)
struct FragSrc {
  __amdgpu_buffer_rsrc_t wf;
  uint32_t layer_bytes;      // wave-uniform; ~0u: no request
  uint32_t lane16;
};
__device__ __forceinline__ void load_frag4(float (&a0)[SNV_KSTEPS], float (&a1)[SNV_KSTEPS], const FragSrc& w4, int g) {
  const f32x4 u = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w4.wf, w4.lane16, w4.layer_bytes + 1024u * g, 0));
  const f32x4 v = __builtin_bit_cast(
      f32x4, __builtin_amdgcn_raw_buffer_load_b128(w4.wf, w4.lane16, w4.layer_bytes + 1024u * (SNV_KSTEPS / 4 + g), 0));
  a0[4 * g] = u.x;
  a0[4 * g + 1] = u.y;
  a0[4 * g + 2] = u.z;
  a0[4 * g + 3] = u.w;
  a1[4 * g] = v.x;
  a1[4 * g + 1] = v.y;
  a1[4 * g + 2] = v.z;
  a1[4 * g + 3] = v.w;
}

// One conv tap (8 k-steps) of a block for both M-blocks -- the two accumulator chains alternate, pair by pair -- with the PREVIOUS
// block's epilogue of one M-block (epi_burst) behind the first pair.  A scheduling barrier closes every pair: the
// backend's own grouping serialises each chain (8 dependent MFMAs in a row: 40 instead of 32 cycles each).

// The whole epilogue of one M-block of a block as ONE burst of vector instructions (measured, tools/microbench/mfma_valu.hip: vector
// ALU work does not run beside this wave's or its SIMD partner's MFMAs -- every v_* instruction costs 2.5 - 5 cycles of MFMA issue and
// every switch MFMA -> VALU -> MFMA a further 8 - 16 -- so the epilogue is as few instructions in as few groups as it can be):
// packed fp32 forms for the BatchNorm map and the residual update, separator / padding lanes masked at the store.  Same roundings as epilogue() in snv_tower_conv.h.
// MODE (layer_mode, compile time): the role of the layer decides which pieces exist at all -- a raw layer (MODE_RES_LAST) stores its
// accumulators as they are (ps = 1, pt = 0, lo = -inf), only MODE_RES_FIRST / MODE_ENTRY touch the residual stream.
template <bool FINAL, int MODE>
__device__ __forceinline__ void epi_burst(const LayerK& k, const f32x4& acc, f32x4& xr, const f32x4& ps, const f32x4& pt, char* img,
                                          uint32_t off, bool valid, uint32_t dump) {
  f32x2 o01 = {acc.x, acc.y}, o23 = {acc.z, acc.w};
  if constexpr (MODE == MODE_GENERIC) {      // role from the run-time constants (kx == 1: every layer but the entry conv)
    o01 = __builtin_elementwise_fma(f32x2{ps.x, ps.y}, f32x2{fmaxf(acc.x, k.lo), fmaxf(acc.y, k.lo)}, f32x2{pt.x, pt.y});
    o23 = __builtin_elementwise_fma(f32x2{ps.z, ps.w}, f32x2{fmaxf(acc.z, k.lo), fmaxf(acc.w, k.lo)}, f32x2{pt.z, pt.w});
  } else if constexpr (MODE != MODE_RES_LAST) {
    o01 = f32x2{fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f)};
    o23 = f32x2{fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f)};
    if constexpr (MODE != MODE_FINAL) {      // conv3 is raw behind its ReLU
      o01 = __builtin_elementwise_fma(f32x2{ps.x, ps.y}, o01, f32x2{pt.x, pt.y});
      o23 = __builtin_elementwise_fma(f32x2{ps.z, ps.w}, o23, f32x2{pt.z, pt.w});
    }
  }
  if constexpr (!FINAL) {      // the last layer of a launch: the residual registers are dead (and may already hold prefetched data)
    if constexpr (MODE == MODE_GENERIC) {
      const f32x2 ku = {k.ku, k.ku};
      const f32x2 x01 = __builtin_elementwise_fma(f32x2{acc.x, acc.y}, ku, f32x2{xr.x, xr.y});
      const f32x2 x23 = __builtin_elementwise_fma(f32x2{acc.z, acc.w}, ku, f32x2{xr.z, xr.w});
      xr = f32x4{x01.x, x01.y, x23.x, x23.y};
    }
    if constexpr (MODE == MODE_RES_FIRST) xr += acc;      // z = x1 + x0 keeps the outer skip (model_snv.py:477-479)
    if constexpr (MODE == MODE_ENTRY) xr = acc;
  }
  // separator / padding columns are zero when a stage starts and stay zero: their lanes store into the wave's dump slots instead
  // (one select on the address instead of four on the values; no branch)
  lds_st4(img, valid ? off : dump, f32x4{o01.x, o01.y, o23.x, o23.y});
}

__device__ __forceinline__ f32x4 relu_bn_pk(const f32x4& v, const f32x4& s, const f32x4& t) {
  const f32x2 o01 = __builtin_elementwise_fma(f32x2{s.x, s.y}, f32x2{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)}, f32x2{t.x, t.y});
  const f32x2 o23 = __builtin_elementwise_fma(f32x2{s.z, s.w}, f32x2{fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}, f32x2{t.z, t.w});
  return f32x4{o01.x, o01.y, o23.x, o23.y};
}

template <int MODE>
__device__ __forceinline__ f32x4 acc_init_pk(const LayerK& k, const f32x4& pb, const f32x4& xr) {
  if constexpr (MODE == MODE_GENERIC) {
    const f32x2 kr = {k.kr, k.kr};
    const f32x2 a01 = __builtin_elementwise_fma(f32x2{xr.x, xr.y}, kr, f32x2{pb.x, pb.y});
    const f32x2 a23 = __builtin_elementwise_fma(f32x2{xr.z, xr.w}, kr, f32x2{pb.z, pb.w});
    return f32x4{a01.x, a01.y, a23.x, a23.y};
  } else if constexpr (MODE == MODE_RES_FIRST || MODE == MODE_RES_LAST) {
    return pb + xr;      // accumulator starts from bias + residual
  } else {
    return pb;
  }
}

template <int T, bool EPI, bool FINAL, int MODE>
__device__ __forceinline__ void mfma_tap_epi(const float (&a0)[SNV_KSTEPS], const float (&a1)[SNV_KSTEPS], const f32x4 (&bv)[2], f32x4& acc0,
                                             f32x4& acc1, const LayerK& k, const f32x4& pa, f32x4& xr, const f32x4& ps,
                                             const f32x4& pt, char* img, uint32_t off, bool valid, uint32_t dump) {
#define MURAL_TAP_PAIR(I)                                                                                       \
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[8 * T + (I)], bv[(I) >> 2][(I) & 3], acc0, 0, 0, 0);          \
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[8 * T + (I)], bv[(I) >> 2][(I) & 3], acc1, 0, 0, 0);          \
  if constexpr (EPI && (I) == 0) epi_burst<FINAL, MODE>(k, pa, xr, ps, pt, img, off, valid, dump);              \
  __builtin_amdgcn_sched_barrier(0);
  MURAL_TAP_PAIR(0)
  MURAL_TAP_PAIR(1)
  MURAL_TAP_PAIR(2)
  MURAL_TAP_PAIR(3)
  MURAL_TAP_PAIR(4)
  MURAL_TAP_PAIR(5)
  MURAL_TAP_PAIR(6)
  MURAL_TAP_PAIR(7)
#undef MURAL_TAP_PAIR
}

// Request of the NEXT unit's stage-1 activations (first-stage launches): wave-uniform descriptor of that unit's first row and the
// scalars of the lane-offset formula.  on == false: nothing is requested inside the layer.
struct XReq {
  __amdgpu_buffer_rsrc_t base;
  FastDiv dSc;
  uint32_t Sc, L, rows, x0_cols, lane_col, kk16;
  bool on;
};
// byte offset of this lane's 16 bytes of block b (M-block 0; M-block 1 sits 64 bytes further); lanes whose column holds no
// data read the unit's first column (their values never reach a data column)
__device__ __forceinline__ uint32_t xreq_offset(const XReq& x, int b) {
  const uint32_t c = 16u * b + x.lane_col;
  const uint32_t u = c - 1u;
  const uint32_t p = x.dSc.div(u);
  const uint32_t j2 = u - p * x.Sc;
  const bool ok = c >= 1u && p < x.rows && j2 < x.L;
  return (ok ? ((p * x.x0_cols + j2) << 7) : 0u) + x.kk16;
}

// One 32->32 k=3 conv layer of this wave, in place on `img`, NB blocks wide (compile time: straight-line code, no copies at block
// boundaries).  wn: fragments of the layer that follows (wfrag4 layout + 4 * lane), requested into a0 / a1 tap group by tap group
// behind the last block's MFMAs of that group.  FINAL: last layer of a launch -- the residual registers die with each block's
// accumulator start and are not updated; with xq.on block b's registers at once receive the next unit's stage-1 activations,
// which then travel under the rest of the layer and the pooling instead of stalling the next entry.
template <int NB, bool FINAL, bool USEX, int MODE>
__device__ __forceinline__ void conv_layer_wave(char* img, const WaveAddr& sa, const LayerK& k, float (&a0)[SNV_KSTEPS],
                                                float (&a1)[SNV_KSTEPS], const FragSrc& wn, const f32x4 (&pb)[2], const f32x4 (&ps)[2],
                                                const f32x4 (&pt)[2], f32x4 (&xr0)[TW_NBW], f32x4 (&xr1)[TW_NBW], const XReq& xq) {
  static_assert(NB >= 1 && NB <= TW_NBW, "blocks per wave");
  f32x4 X[2], Y[2], Z[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    X[h] = lds_ld4(img, sa.rd[h]);
    Y[h] = lds_ld4(img, sa.rd[2 + h]);
    Z[h] = lds_ld4(img, sa.rd[4 + h]);
  }
  f32x4 pa0 = splat(0.f), pa1 = splat(0.f);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const bool last = b == NB - 1;
    const int bp = b > 0 ? b - 1 : 0;
    const bool vprev = (sa.vmask >> bp) & 1u;
    // USEX == false: a layer whose accumulators start from the bias alone (kr == 0) must not touch the residual registers --
    // in the short-stage launches they already hold the next unit's input tile
    f32x4 acc0 = USEX ? acc_init_pk<MODE>(k, pb[0], xr0[b]) : pb[0], acc1 = USEX ? acc_init_pk<MODE>(k, pb[1], xr1[b]) : pb[1];
    __builtin_amdgcn_sched_barrier(0);
    if (FINAL && xq.on) {
      const uint32_t vo = xreq_offset(xq, b);
      xr0[b] = buf_ld4(xq.base, vo);
      xr1[b] = buf_ld4(xq.base, vo + 64u);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_tap_epi<0, false, FINAL, MODE>(a0, a1, X, acc0, acc1, k, pa0, xr0[bp], ps[0], pt[0], img, 0u, true, 0u);
    // tap-0 operands of the next block: read BEFORE this block's epilogue overwrites image column 16(b+1) (in-place rule)
    if (!last) {
#pragma unroll
      for (int h = 0; h < 2; ++h) X[h] = lds_ld4(img, sa.rd[h] + TW_BLK * (b + 1));
    } else if (wn.layer_bytes != ~0u) {
      load_frag4(a0, a1, wn, 0);
      load_frag4(a0, a1, wn, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the previous block's epilogue rides behind this block's tap-1 (M-block 0) and tap-2 (M-block 1) MFMA pairs
    if (b > 0) mfma_tap_epi<1, true, FINAL, MODE>(a0, a1, Y, acc0, acc1, k, pa0, xr0[bp], ps[0], pt[0], img, sa.wr[0] + TW_BLK * bp, vprev, sa.dump);
    else mfma_tap_epi<1, false, FINAL, MODE>(a0, a1, Y, acc0, acc1, k, pa0, xr0[bp], ps[0], pt[0], img, 0u, true, 0u);
    if (!last) {
#pragma unroll
      for (int h = 0; h < 2; ++h) Y[h] = lds_ld4(img, sa.rd[2 + h] + TW_BLK * (b + 1));
    } else if (wn.layer_bytes != ~0u) {
      load_frag4(a0, a1, wn, 2);
      load_frag4(a0, a1, wn, 3);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (b > 0) mfma_tap_epi<2, true, FINAL, MODE>(a0, a1, Z, acc0, acc1, k, pa1, xr1[bp], ps[1], pt[1], img, sa.wr[1] + TW_BLK * bp, vprev, sa.dump);
    else mfma_tap_epi<2, false, FINAL, MODE>(a0, a1, Z, acc0, acc1, k, pa1, xr1[bp], ps[1], pt[1], img, 0u, true, 0u);
    if (!last) {
#pragma unroll
      for (int h = 0; h < 2; ++h) Z[h] = lds_ld4(img, sa.rd[4 + h] + TW_BLK * (b + 1));
    } else if (wn.layer_bytes != ~0u) {
      load_frag4(a0, a1, wn, 4);
      load_frag4(a0, a1, wn, 5);
    }
    pa0 = acc0;
    pa1 = acc1;
    __builtin_amdgcn_sched_barrier(0);
  }
  // the last block's epilogue has no MFMAs to hide behind
  {
    f32x4 dx0 = xr0[NB - 1], dx1 = xr1[NB - 1];      // FINAL: the registers may hold prefetched data, the update goes to a dead copy
    const bool v = (sa.vmask >> (NB - 1)) & 1u;
    epi_burst<FINAL, MODE>(k, pa0, FINAL ? dx0 : xr0[NB - 1], ps[0], pt[0], img, sa.wr[0] + TW_BLK * (NB - 1), v, sa.dump);
    epi_burst<FINAL, MODE>(k, pa1, FINAL ? dx1 : xr1[NB - 1], ps[1], pt[1], img, sa.wr[1] + TW_BLK * (NB - 1), v, sa.dump);
  }
}

template <int NB, bool FINAL, bool USEX, int MODE>
__device__ __forceinline__ void conv_layer_any(char* img, const WaveAddr& sa, int nb, const LayerK& k, float (&a0)[SNV_KSTEPS],
                                               float (&a1)[SNV_KSTEPS], const FragSrc& wn, const f32x4 (&pb)[2], const f32x4 (&ps)[2],
                                               const f32x4 (&pt)[2], f32x4 (&xr0)[TW_NBW], f32x4 (&xr1)[TW_NBW], const XReq& xq) {
  // block counts are compile-time constants (plan_wave_geometry only admits geometries with an instance); NB == 0 marks the
  // stage a launch phase does not have (its call sits in a branch that phase never takes)
  (void)nb;
  if constexpr (NB > 0) conv_layer_wave<NB, FINAL, USEX, MODE>(img, sa, k, a0, a1, wn, pb, ps, pt, xr0, xr1, xq);
}

// first x0 column (counted over the whole buffer) of launch row `row`: a site's row at the tower's column offset -- or, for the
// segment launches of long windows (SnvFwdArgs::seg_n, one row per wave), segment row % seg_n of site row / seg_n in place
__device__ __forceinline__ size_t wave_x0_column(const SnvFwdArgs& args, int64_t row, int x0_cols, int x0c) {
  if (args.seg_n > 0) {
    const int64_t site = row / args.seg_n;
    const int k = (int)(row - site * args.seg_n);
    return (size_t)site * (size_t)x0_cols + (size_t)(args.seg_col0 + k * args.seg_step);
  }
  return (size_t)row * (size_t)x0_cols + (size_t)x0c;
}

// Stage-1 activations of unit `unit` straight into the residual registers, in MFMA accumulator layout (lane = column n16 of each
// block, channels 4 kk .. + 3 of both M-blocks): one wave-uniform buffer descriptor per unit + a 32-bit lane offset worked out on
// the spot (a division by multiply-high per block) -- nothing per-lane survives between units, so nothing is spilled.  Columns
// without data (separators, padding) and rows behind the last site read as zero.
__device__ __forceinline__ void wave_request_x0(const SnvFwdArgs& args, const TowerGeom& g, int Pw, int x0_cols, int x0c,
                                                f32x4 (&xr0)[TW_NBW], f32x4 (&xr1)[TW_NBW], int64_t unit, int64_t n_units, int n16, int kk) {
  const int64_t row0 = unit * Pw;
  const bool any = unit < n_units;
  const __amdgpu_buffer_rsrc_t base = uniform_rsrc(args.x0 + wave_x0_column(args, any ? row0 : 0, x0_cols, x0c) * 32);
  const int rows = any ? (int)(args.n - row0 < Pw ? args.n - row0 : Pw) : 0;      // sites of this unit that exist
  uint32_t lane_col = (uint32_t)n16;
  asm volatile("" : "+v"(lane_col));      // opaque: keeps the offsets below from being precomputed for the whole launch
  if (rows == Pw) {
    // a whole unit (all but the last): every lane loads -- lanes whose column holds no data (separators, padding) read the unit's
    // first column instead, their values never reach a data column -- so there is no divergent control flow around the loads
#pragma unroll
    for (int b = 0; b < TW_NBW; ++b) {
      if (b < g.nb[0]) {
        const uint32_t c = 16u * b + lane_col;
        const uint32_t u = c - 1u;
        const uint32_t p = g.dSc[0].div(u);
        const uint32_t j2 = u - p * (uint32_t)g.Sc[0];
        const bool ok = c >= 1u && p < (uint32_t)rows && j2 < (uint32_t)g.L[0];
        const uint32_t vo = (ok ? ((p * (uint32_t)x0_cols + j2) << 7) : 0u) + 16u * (uint32_t)kk;
        xr0[b] = buf_ld4(base, vo);
        xr1[b] = buf_ld4(base, vo + 64u);
      } else {
        xr0[b] = splat(0.f);
        xr1[b] = splat(0.f);
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < TW_NBW; ++b) {
    xr0[b] = splat(0.f);
    xr1[b] = splat(0.f);
    if (b < g.nb[0]) {
      const uint32_t c = 16u * b + lane_col;
      const uint32_t u = c - 1u;                        // c == 0 wraps: p becomes huge and the lane reads nothing
      const uint32_t p = g.dSc[0].div(u);
      const uint32_t j2 = u - p * (uint32_t)g.Sc[0];
      if (c >= 1u && p < (uint32_t)rows && j2 < (uint32_t)g.L[0]) {
        const uint32_t vo = ((p * (uint32_t)x0_cols + j2) << 7) + 16u * (uint32_t)kk;
        xr0[b] = buf_ld4(base, vo);
        xr1[b] = buf_ld4(base, vo + 64u);
      }
    }
  }
}

// zero the image columns of stage `st` that hold no data: the separators, the padding behind the last site and the column behind
// the last block (lanes cooperate: 8 chunks per column)
__device__ __forceinline__ void wave_zero_gaps(float* img, const TowerGeom& g, int st, int Pw, int lane) {
  const int Sc = g.Sc[st];
  const int nz = 1 + Pw + (16 * g.nb[st] - g.NC[st]) + 1;
  for (int task = lane; task < nz * 8; task += 64) {
    const int k = task >> 3;
    const int c = (k <= Pw) ? k * Sc : g.NC[st] + (k - Pw - 1);
    st4(img + lds_off(c + 1, task & 7), splat(0.f));
  }
}

// PHASE 1: first conv stage of ONE tower (entry BN/ReLU image, four ResBlock convs, max-pool 2 + BN -> s3).
// PHASE 2: the two short stages of ONE tower (conv2, four ResBlock convs, max-pool 3 + BN, conv3), global max, fc; the mid
//          tower's launch also runs the head.
// NBA / NBB: 16-column blocks per wave of the launch's main stage / of the last stage (PHASE 2), fixed at compile time for the
// instances plan_wave_geometry admits (0: the launch phase has no such stage).
// Arrival counters per CU (never reset: the workgroups resident on a CU at any time hold consecutive counts).  The two
// workgroups of a CU are symmetric and start together, so left alone their waves run in lockstep: both in their conv layers
// (sharing the MFMA pipe), then both in their entry / pooling phases (pipe idle).  Every second arrival therefore starts
// `stagger` x 8 k cycles late; from then on one wave's boundary phases fall under its SIMD partner's MFMAs.
__device__ int g_cu_arrivals[1024];

__device__ __forceinline__ uint32_t cu_key() {
  // HW_REG_HW_ID (id 4): cu_id [11:8], sh_id [12], se_id [15:13]; HW_REG_XCC_ID (id 20): xcc_id [3:0]
  const uint32_t hw = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4);      // 8 bits from bit 8
  const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // 4 bits from bit 0
  return ((xcc & 7u) << 7) | (hw & 127u);
}

// diagnostic: accumulate one wave's cycles per phase into args.stamps[block][id] (only when stamps != nullptr; tools/phase_stamps.py)
#define SNVW_STAMP(id)                                                               \
  do {                                                                               \
    if (args.stamps != nullptr && lane == 0 && wave == 0) {                          \
      const unsigned long long _t = __builtin_amdgcn_s_memtime();                    \
      args.stamps[(size_t)SNVW_BX * 32 + (PHASE == 1 ? 0 : 8) + 16 * tw_i + (id)] += _t - t_prev; \
      t_prev = _t;                                                                   \
    }                                                                                \
  } while (0)

// TWC / PWC: tower and sites per wave of an instantiation compiled for the shipped window (SHIP_LWIN): the whole geometry is a
// compile-time constant there -- no divisions by run-time constants, no geometry words fetched from the kernel arguments inside the
// unit loop (with ~100 SGPRs live the backend re-reads them with s_load + wait at every use); TWC < 0: geometry from the arguments.
template <int PHASE, int NBA, int NBB, int TWC = -1, int PWC = 0>
__global__ __launch_bounds__(SNV_THREADS, 2) void snv_tower_wave(const SnvFwdArgs args) {
#define SNVW_BX blockIdx.x
#define SNVW_GX gridDim.x
#include "snv_tower_wave_body.inc"
}
#undef SNVW_BX
#undef SNVW_GX

// Up to three first-stage jobs of ONE instance in one launch (blockIdx.y = job; a workgroup beyond its job's grid leaves at once): the
// long-window path's three launches of this stage -- the large tower's equal segments, its last segment, the mid tower -- each paid
// ~14 us of start-up for 22 us of work per unit at a few hundred windows per call.  Same body, the job's own workgroup count as the stride.
struct SnvFwdJobs {
  SnvFwdArgs j[3];
  int grid[3];
};
static_assert(sizeof(SnvFwdJobs) <= 4096, "kernel arguments: 4 KB");
#define SNVW_BX bx_job
#define SNVW_GX gx_job
template <int PHASE, int NBA, int NBB, int TWC = -1, int PWC = 0>
__global__ __launch_bounds__(SNV_THREADS, 2) void snv_tower_wave_jobs(const SnvFwdJobs jobs) {
  const SnvFwdArgs& args = jobs.j[blockIdx.y];
  const unsigned gx_job = (unsigned)jobs.grid[blockIdx.y];
  const unsigned bx_job = blockIdx.x;
  if (bx_job >= gx_job) return;
#include "snv_tower_wave_body.inc"
}
#undef SNVW_BX
#undef SNVW_GX

// ====================================================================================================================
// Edge tiles of the cross-position reuse path (snv_reuse.hip) in the wave-private form: a wave owns
// EW_P = 7 sites x (9 + 9 edge columns + separator) = 134 columns = the nine blocks of a first-stage launch, runs the four
// ResBlock convs on them with the layer code above (same fragments, k order, epilogue forms -- the per-window kernel's own
// operation sequence), and pools: interior windows are gathered from the shared rows S, the windows that touch the edge
// pyramids read the wave's image (+ R rows when mixed).  Replaces the workgroup-tile snv_edge_kernel (two LDS buffers, a
// workgroup barrier per layer) for the shipped layer shape.
// ====================================================================================================================
constexpr int EW_NB = 9;
constexpr int EW_WST = 32;      // floats per wave behind the image: window starts of this unit and of the next one (2 x 8 int64)

__device__ __forceinline__ int64_t edge_wave_wstart(const EdgeArgs& args, int64_t row) {
  if (row >= args.n) return -1;
  const int64_t gp = args.pos[row];
  const int neg = args.strand[row] != 0;
  const int64_t t = neg ? args.glen - 1 - gp : gp;
  const int64_t w = t + args.woff - args.t0[neg];
  if (w < 0 || w + args.L1 > args.nb || args.F[neg] == nullptr) return -1;   // the caller's bounds / strand mask were wrong
  return w | ((int64_t)neg << 62);
}

// raw x0 of the lane's edge columns of one unit, in accumulator layout (both M-blocks): gathers from the shared rows.  The (site,
// column) of a lane's column in block b is launch-invariant; it is worked out here from an opaque lane index so that nothing per
// block is kept (and spilled) across the unit loop.
__device__ __forceinline__ void edge_wave_request(const EdgeArgs& args, const int64_t* wst, f32x4 (&xr0)[TW_NBW], f32x4 (&xr1)[TW_NBW],
                                                  int n16_in, int kk) {
  int n16 = n16_in;
  asm volatile("" : "+v"(n16));
  const float* safe = args.F[0] != nullptr ? args.F[0] : args.F[1];
#pragma unroll
  for (int b = 0; b < EW_NB; ++b) {
    const uint32_t c = 16u * b + (uint32_t)n16;
    const uint32_t u = c - 1u;
    const uint32_t p = u / (uint32_t)RU_SC;      // c == 0 wraps: p is huge and the lane reads nothing
    const uint32_t j = u - p * (uint32_t)RU_SC;
    const bool col = c >= 1u && p < (uint32_t)EW_P && j < (uint32_t)RU_L;
    const int q = (int)j < RU_EC ? (int)j : args.L2 - RU_L + (int)j;
    const int64_t ws = wst[col ? p : 0];
    const bool ok = col && ws >= 0;
    const int set = (int)((ws >> 62) & 1);
    const int64_t w = ws & ((1ll << 62) - 1);
    const float* src = args.F[set] + (size_t)(w + (int64_t)args.D * q) * 32;
    if (q == 0) src = args.El[set] + (size_t)w * 32;
    else if (q == args.L2 - 1 && args.right_pad) src = args.Er[set] + (size_t)(w + args.L1 - 1) * 32;
    if (!ok) src = safe;
    const f32x4 v0 = ld4(src + 4 * kk), v1 = ld4(src + 16 + 4 * kk);
    xr0[b] = ok ? v0 : splat(0.f);
    xr1[b] = ok ? v1 : splat(0.f);
  }
}

__global__ __launch_bounds__(SNV_THREADS, 2) void snv_edge_wave(const EdgeArgs args, int* unit_counter) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4;
  const TowerDev& tw = args.tw;
  constexpr int Pw = EW_P;
  // the edge tile's geometry: Pw sites x (RU_L data columns + separator) behind the guard column
  TowerGeom g{};
  g.L[0] = RU_L;
  g.Sc[0] = RU_SC;
  g.NC[0] = 1 + Pw * RU_SC;
  g.nb[0] = EW_NB;
  g.dL[0] = FastDiv::make(RU_L);
  g.dSc[0] = FastDiv::make(RU_SC);
  // LDS: par = ex_s[2][32] (entry, pool) | ex_t[2][32] | lpar[4 layers][3][32]; per wave: image | window starts | dump slots
  constexpr int nbuf = (16 * EW_NB + 2) * SNV_C;
  constexpr int lpar0 = 4 * SNV_C;
  constexpr int par_floats = lpar0 + 4 * 3 * SNV_C;
  constexpr int wave_floats = nbuf + EW_WST + TW_DUMP;
  float* par = smem;
  float* img = smem + par_floats + wave * wave_floats;
  int64_t* wst_buf = reinterpret_cast<int64_t*>(img + nbuf);      // [2][8]
  const uint32_t dump0 = 4u * (uint32_t)(nbuf + EW_WST);
  if (tid < SNV_C) {
    par[tid] = tw.ex_s[EX_RB1_ENTRY * 32 + tid];
    par[SNV_C + tid] = tw.ex_s[EX_BN_MID * 32 + tid];
    par[2 * SNV_C + tid] = tw.ex_t[EX_RB1_ENTRY * 32 + tid];
    par[3 * SNV_C + tid] = tw.ex_t[EX_BN_MID * 32 + tid];
  }
  for (int i = tid; i < 4 * SNV_C; i += SNV_THREADS) {
    const int l = i >> 5, c = i & 31;
    par[lpar0 + (l * 3 + 0) * SNV_C + c] = tw.bias[l * 32 + c];
    par[lpar0 + (l * 3 + 1) * SNV_C + c] = tw.post_s[l * 32 + c];
    par[lpar0 + (l * 3 + 2) * SNV_C + c] = tw.post_t[l * 32 + c];
  }
  if (lane < 8) st4(img + lds_off(0, lane), splat(0.f));
  else if (lane < 16) st4(img + lds_off(16 * EW_NB + 1, lane - 8), splat(0.f));
  wave_zero_gaps(img, g, 0, Pw, lane);      // once: no layer stores into a gap column
  __syncthreads();

  const int64_t n_units = (args.n + Pw - 1) / Pw;
  const bool dyn = unit_counter != nullptr;
  const int64_t unit_step = (int64_t)gridDim.x * SNV_WAVES;
  int ticket = 0;
  int64_t unit0 = (int64_t)blockIdx.x * SNV_WAVES + wave;
  if (dyn) {
    if (lane == 0) ticket = atomicAdd(unit_counter, 1);
    unit0 = __builtin_amdgcn_readfirstlane(ticket);
    if (lane == 0) ticket = atomicAdd(unit_counter, 1);
  }
  char* imgb = reinterpret_cast<char*>(img);
  const int chv0 = 4 * kk, chv1 = 16 + 4 * kk;
  f32x4 xr0[TW_NBW], xr1[TW_NBW];
  float a0[SNV_KSTEPS], a1[SNV_KSTEPS];
  FragSrc fsrc;
  fsrc.wf = uniform_rsrc(tw.wfrag4);
  fsrc.lane16 = 16u * (uint32_t)lane;
  fsrc.layer_bytes = 0u;
#pragma unroll
  for (int g4 = 0; g4 < SNV_KSTEPS / 4; ++g4) load_frag4(a0, a1, fsrc, g4);
  const WaveAddr sa = wave_setup(g, 0, Pw, n16, kk, dump0);
  XReq xoff{};
  xoff.on = false;
  int cur = 0;
  if (lane < 8) wst_buf[lane] = lane < Pw ? edge_wave_wstart(args, unit0 * Pw + lane) : -1;
  edge_wave_request(args, wst_buf, xr0, xr1, n16, kk);
  const int cg = lane & 7;

  for (int64_t unit = unit0, next_unit = 0; unit < n_units; unit = next_unit) {
    const int64_t row0 = unit * Pw;
    if (dyn) {
      next_unit = __builtin_amdgcn_readfirstlane(ticket);
      if (lane == 0) ticket = atomicAdd(unit_counter, 1);
    } else {
      next_unit = unit + unit_step;
    }
    const int64_t* wst = wst_buf + 8 * cur;
    // window starts of the next unit: the position loads hide under the convs
    if (lane < 8) wst_buf[8 * (cur ^ 1) + lane] = (lane < Pw && next_unit < n_units) ? edge_wave_wstart(args, next_unit * Pw + lane) : -1;
    // ---- entry: BN(ReLU(x0)) image of the edge columns; the raw values stay in the residual registers
    {
      const f32x4 es0 = ld4(par + chv0), et0 = ld4(par + 2 * SNV_C + chv0);
      const f32x4 es1 = ld4(par + chv1), et1 = ld4(par + 2 * SNV_C + chv1);
#pragma unroll
      for (int b = 0; b < EW_NB; ++b) {
        const bool v = (sa.vmask >> b) & 1u;
        lds_st4(imgb, v ? sa.wr[0] + TW_BLK * b : sa.dump, relu_bn_pk(xr0[b], es0, et0));
        lds_st4(imgb, v ? sa.wr[1] + TW_BLK * b : sa.dump, relu_bn_pk(xr1[b], es1, et1));
      }
    }
    // ---- the four ResBlock convs on the edge tile
    auto run_layer = [&](int layer, auto mode_tag, auto final_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
      constexpr bool FINAL = decltype(final_tag)::value;
      const LayerK lk = layer_consts(layer_mode(layer));
      int lofs = lpar0 + layer * 3 * SNV_C;
      asm volatile("" : "+s"(lofs));
      const float* lp = par + lofs;
      const f32x4 pb[2] = {ld4(lp + chv0), ld4(lp + chv1)};
      const f32x4 ps[2] = {ld4(lp + SNV_C + chv0), ld4(lp + SNV_C + chv1)};
      const f32x4 pt[2] = {ld4(lp + 2 * SNV_C + chv0), ld4(lp + 2 * SNV_C + chv1)};
      FragSrc wn = fsrc;
      wn.layer_bytes = (uint32_t)(layer < 3 ? layer + 1 : 0) * SNV_WFRAG * 4u;
      conv_layer_any<EW_NB, FINAL, true, MODE>(imgb, sa, EW_NB, lk, a0, a1, wn, pb, ps, pt, xr0, xr1, xoff);
    };
    for (int layer = 0; layer < 3; ++layer) run_layer(layer, ModeTag<MODE_GENERIC>{}, std::false_type{});
    run_layer(3, ModeTag<MODE_RES_LAST>{}, std::true_type{});
    // the residual registers are dead: the next unit's gathers fly under the pooling
    if (next_unit < n_units) edge_wave_request(args, wst_buf + 8 * (cur ^ 1), xr0, xr1, n16, kk);
    // ---- maxpool2 + BN -> s3.  Interior windows (columns u_lo .. u_hi): gathers from S, four rounds in flight at a time; the few
    //      windows that touch the edge pyramids read the image (+ R rows when mixed) and get the BN here.
    {
      const f32x4 pool_s = ld4(par + SNV_C + 4 * cg), pool_t = ld4(par + 3 * SNV_C + 4 * cg);
      const int n_int = args.u_hi - args.u_lo + 1;
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      // interior windows are a copy S -> s3: site by site (window start and destination row wave-uniform: scalar address math, one
      // buffer descriptor each), a round = 8 pooled columns x 8 channel groups; the lane offsets of the rounds are worked out once
      // per unit, a site costs three loads and three stores and no vector arithmetic
      const int n_rounds = (n_int + 7) >> 3;
      const int u_l = lane_o >> 3;
      const uint32_t sstep = (uint32_t)(args.D * args.ps2) * 128u;      // bytes between consecutive pooled columns on the base axis
      uint32_t so[3], dof[3];
      bool on[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int u = args.u_lo + 8 * r + u_l;
        on[r] = r < n_rounds && u <= args.u_hi;
        so[r] = (uint32_t)u * sstep + 16u * (uint32_t)cg;
        dof[r] = (uint32_t)u * 128u + 16u * (uint32_t)cg;
      }
      const float* s_any = args.S[0] != nullptr ? args.S[0] : args.S[1];
      for (int p = 0; p < Pw; ++p) {
        if (row0 + p >= args.n) break;
        const int64_t wsv = wst[p];
        const int64_t ws = ((int64_t)__builtin_amdgcn_readfirstlane((int)(wsv >> 32)) << 32) |
                           (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(wsv & 0xffffffffll));
        const bool ok = ws >= 0;
        const int64_t w = ws & ((1ll << 62) - 1);
        const __amdgpu_buffer_rsrc_t srs = uniform_rsrc(ok ? args.S[(int)((ws >> 62) & 1)] + (size_t)w * 32 : s_any);
        const __amdgpu_buffer_rsrc_t drs = uniform_rsrc(args.s3 + (size_t)(row0 + p) * args.L3 * 32);
        f32x4 sv[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          sv[r] = splat(__uint_as_float(0x7FC00000u));
          if (on[r] && ok) sv[r] = buf_ld4(srs, so[r]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
          if (on[r]) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sv[r]), drs, dof[r], 0, 0);
        for (int r = 3; r < n_rounds; ++r) {      // more than 24 interior columns (not in the shipped geometries)
          const int u = args.u_lo + 8 * r + u_l;
          if (u <= args.u_hi) {
            const f32x4 v = ok ? buf_ld4(srs, (uint32_t)u * sstep + 16u * (uint32_t)cg) : splat(__uint_as_float(0x7FC00000u));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), drs, (uint32_t)u * 128u + 16u * (uint32_t)cg, 0, 0);
          }
        }
      }
      const int n_edge = args.L3 - n_int;
#pragma unroll 1
      for (int task = lane_o; task < Pw * n_edge * 8; task += 64) {
        const int pj = task >> 3;
        const int p = pj / n_edge, e = pj - p * n_edge;
        const int u = e < args.u_lo ? e : args.u_hi + 1 + (e - args.u_lo);
        if (row0 + p >= args.n) continue;
        const int64_t ws = wst[p];
        f32x4 m = splat(__uint_as_float(0x7FC00000u));
        if (ws >= 0) {
          const int set = (int)(ws >> 62);
          const int64_t w = ws & ((1ll << 62) - 1);
          const int jlo = u * args.ps2 - args.pp2;
          const int lo = jlo < 0 ? 0 : jlo;
          const int hi = (jlo + args.pk2 - 1) < (args.L2 - 1) ? (jlo + args.pk2 - 1) : (args.L2 - 1);
          f32x4 rv[7];
#pragma unroll
          for (int d = 0; d < 7; ++d) {           // the model's second pools are 7 / 3 wide: every row read in flight together
            rv[d] = splat(-INFINITY);
            const int q = lo + d;
            if (q <= hi) {
              if (q < RU_EV) rv[d] = ld4(img + lds_off(1 + p * RU_SC + q + 1, cg));
              else if (q > args.L2 - 1 - RU_EV) rv[d] = ld4(img + lds_off(1 + p * RU_SC + (q - (args.L2 - RU_L)) + 1, cg));
              else rv[d] = ld4(args.R[set] + (size_t)(w + (int64_t)args.D * q) * 32 + 4 * cg);
            }
          }
          m = max4(max4(max4(rv[0], rv[1]), max4(rv[2], rv[3])), max4(max4(rv[4], rv[5]), rv[6]));
          for (int q = lo + 7; q <= hi; ++q) {     // wider pools (not in the model): plain loop
            f32x4 v;
            if (q < RU_EV) v = ld4(img + lds_off(1 + p * RU_SC + q + 1, cg));
            else if (q > args.L2 - 1 - RU_EV) v = ld4(img + lds_off(1 + p * RU_SC + (q - (args.L2 - RU_L)) + 1, cg));
            else v = ld4(args.R[set] + (size_t)(w + (int64_t)args.D * q) * 32 + 4 * cg);
            m = max4(m, v);
          }
          m = f32x4{fmaf(pool_s.x, m.x, pool_t.x), fmaf(pool_s.y, m.y, pool_t.y), fmaf(pool_s.z, m.z, pool_t.z),
                    fmaf(pool_s.w, m.w, pool_t.w)};
        }
        st4(args.s3 + ((size_t)(row0 + p) * args.L3 + u) * 32 + 4 * cg, m);
      }
    }
    cur ^= 1;
  }
}

static bool tower_dynamic_units() {
  const char* e = dev_env("MURAL_TOWER_DYNAMIC_UNITS");
  return e && atoi(e) != 0 && !dev_env("MURAL_DEBUG_TOWER_STATIC_UNITS");
}

size_t edge_wave_lds_bytes() {
  return (size_t)(4 * SNV_C + 4 * 3 * SNV_C + SNV_WAVES * ((16 * EW_NB + 2) * SNV_C + EW_WST + TW_DUMP)) * 4;
}

int launch_snv_edge_wave(const EdgeArgs& e, int* unit_counter, hipStream_t stream) {
  const int64_t n_units = (e.n + EW_P - 1) / EW_P;
  const int64_t n_wg = (n_units + SNV_WAVES - 1) / SNV_WAVES;
  if (n_wg == 0) return MURAL_OK;
  const int grid = (int)(n_wg < 512 ? n_wg : 512);
  if (n_units < 4 * (int64_t)grid * SNV_WAVES || !tower_dynamic_units()) unit_counter = nullptr;
  static DynLdsOnce lds;
  if (int rc = lds.ensure(&snv_edge_wave)) return rc;
  hipLaunchKernelGGL(snv_edge_wave, dim3(grid), dim3(SNV_THREADS), edge_wave_lds_bytes(), stream, e, unit_counter);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

// per-wave geometry: Pw sites per wave; returns the LDS bytes of a four-wave workgroup (0: does not fit a wave)
size_t plan_wave_geometry(SnvFwdArgs& a, int Lwin, int Pw, int n_class, int tower, int phase) {
  int maxcols = 0;
  for (int tw = 0; tw < 2; ++tw) {
    a.geom[tw] = wave_tower_geom(tw, Lwin, Pw);
    const TowerGeom& g = a.geom[tw];
    for (int i = 0; i < 3; ++i) {
      if (g.L[i] < 1) return 0;
      if (tw != tower) continue;
      if ((phase == 1 && i != 0) || (phase == 2 && i == 0)) continue;
      if (g.nb[i] > TW_NBW) return 0;
      if (phase == 1 && g.nb[i] != TW_NBW) return 0;      // the first-stage kernel is compiled for nine blocks per wave only
      maxcols = std::max(maxcols, 16 * g.nb[i] + 2);
    }
  }
  if (Pw > 31) return 0;      // the x0 plan keeps the site index in five bits
  // the short-stage kernel exists only with the shipped window's geometry at compile time (a run-time-geometry instance -- a switch
  // over nine block counts x three layer roles -- compiled with ~750 spilled VGPRs and its results once depended on unrelated
  // code motion): other windows keep the workgroup-tile kernel for their short stages
  if (phase == 2 && !(Lwin == SHIP_LWIN && Pw == (tower == 0 ? 6 : 5))) return 0;
  a.P = Pw;
  a.Lwin = Lwin;
  a.tw_first = tower;
  a.tw_last = tower;
  a.phase = phase;
  a.wave = 1;
  a.stagger = 0;      // measured: no effect (the waves of a CU do not run in lockstep); kept as a diagnostic
  if (const char* e = dev_env("MURAL_DEBUG_TOWER_STAGGER")) a.stagger = atoi(e);
  a.x0_cols = a.geom[0].L[0] + a.geom[1].L[0];
  a.nbuf = maxcols * SNV_C;
  const size_t par = (size_t)(2 * EX_COUNT * SNV_C + n_class * SNV_C + SNV_MAXCLASS + 4 + 6 * 3 * SNV_C);
  const size_t per_wave = (size_t)a.nbuf + (phase == 2 ? (size_t)Pw * SNV_C + 3 * (size_t)Pw * SNV_MAXCLASS : 0) + TW_DUMP;
  return (par + SNV_WAVES * per_wave) * 4;
}

// n <= 3 first-stage launches of the run-time-geometry instance as ONE launch (snv_tower_wave_jobs); a job without units is dropped
int launch_snv_tower_wave_jobs(const SnvFwdArgs* jobs_in, const size_t* lds_bytes, int n, hipStream_t stream) {
  MURAL_REQUIRE(n >= 1 && n <= 3, "wave-private tower launch: %d jobs", n);
  SnvFwdJobs jobs;
  std::memset(&jobs, 0, sizeof(jobs));
  int gx = 0, ny = 0;
  size_t lds = 0;
  int64_t wg_of[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const SnvFwdArgs& a = jobs_in[i];
    MURAL_REQUIRE(a.phase == 1 && a.Lwin != SHIP_LWIN, "wave-private tower jobs: first-stage launches of the run-time-geometry instance only");
    const int64_t n_units = (a.n + a.P - 1) / a.P;
    const int64_t n_wg = (n_units + SNV_WAVES - 1) / SNV_WAVES;
    if (n_wg == 0) continue;
    jobs.j[ny] = a;
    jobs.j[ny].unit_counter = nullptr;      // units at a fixed stride
    wg_of[ny] = n_wg;
    lds = lds_bytes[i] > lds ? lds_bytes[i] : lds;
    ++ny;
  }
  if (ny == 0) return MURAL_OK;
  // every job gets what a launch of its own would: up to 512 workgroups (two per CU are resident; the hardware hands the later jobs'
  // workgroups the slots the earlier ones free).  Sharing 512 workgroups between the jobs in proportion to their units -- every wave
  // the same number of units, nothing waiting for a slot -- measured SLOWER (0.56 -> 0.48 at 2048 windows): the jobs' units do not cost
  // the same, and a fixed split cannot even that out.
  for (int i = 0; i < ny; ++i) {
    jobs.grid[i] = (int)(wg_of[i] < 512 ? wg_of[i] : 512);
    gx = jobs.grid[i] > gx ? jobs.grid[i] : gx;
  }
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&snv_tower_wave_jobs<1, 9, 0>)) return rc;
  hipLaunchKernelGGL((snv_tower_wave_jobs<1, 9, 0>), dim3(gx, ny), dim3(SNV_THREADS), lds, stream, jobs);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

int launch_snv_tower_wave(const SnvFwdArgs& a_in, size_t lds_bytes, hipStream_t stream) {
  SnvFwdArgs a = a_in;
  const int64_t n_units = (a.n + a.P - 1) / a.P;
  const int64_t n_wg = (n_units + SNV_WAVES - 1) / SNV_WAVES;
  if (n_wg == 0) return MURAL_OK;
  // two workgroups per CU are resident: with more work than that every wave walks its units in a grid-stride loop, so the
  // prologue (parameter staging, first fragments, first activations with their full latency) is paid once per wave
  int resident = 512;
  if (const char* e = dev_env("MURAL_DEBUG_TOWER_GRID")) resident = std::max(1, atoi(e));
  const int grid = (int)(n_wg < resident ? n_wg : resident);
  // Units at a fixed stride (wave w: units w, w + waves, ...).  The ticket counter (MURAL_TOWER_DYNAMIC_UNITS=1: a returning atomic per
  // unit, requested a unit ahead) was the default until the end of round 5 and is 2 - 5 % SLOWER on every batch size measured
  // (500 k sites: 16.52 vs 16.85 M bases/s, 0.753 vs 0.770 of the roof; 77 777 sites: 15.77 vs 16.61 M): the units cost the same, so
  // the fixed stride has no tail to repair, and the counter's waves end a unit apart (64 +- 1 units each).  With the counter, units
  // go through it only when every wave has several to take.
  if (n_units < 4 * (int64_t)grid * SNV_WAVES || !tower_dynamic_units()) a.unit_counter = nullptr;
  if (const char* e = dev_env("MURAL_DEBUG_TOWER_LDS")) {      // diagnostic: inflate the LDS request (one workgroup per CU: occupancy study)
    const size_t v = (size_t)atol(e);
    if (v > lds_bytes && v <= 160 * 1024) lds_bytes = v;
  }
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&snv_tower_wave<1, 9, 0, 0, 1>, &snv_tower_wave<1, 9, 0, 1, 2>, &snv_tower_wave<2, 8, 4, 0, 6>,
                              &snv_tower_wave<2, 8, 3, 1, 5>, &snv_tower_wave<1, 9, 0>)) return rc;
  // the instantiations compiled for the shipped window: the launch's geometry must be exactly theirs
  const int t = a.tw_first;
  const int ship_pw = a.phase == 1 ? (t == 0 ? 1 : 2) : (t == 0 ? 6 : 5);
  // MURAL_DEBUG_TOWER_RUNTIME_GEOM: A/B switch of the tests -- the first-stage launches through the instance that reads its geometry
  // from the arguments (bitwise the same results)
  bool ship = a.Lwin == SHIP_LWIN && a.P == ship_pw && !(a.phase == 1 && dev_env("MURAL_DEBUG_TOWER_RUNTIME_GEOM"));
  if (ship) {
    const TowerGeom want = wave_tower_geom(t, SHIP_LWIN, ship_pw), other = wave_tower_geom(0, SHIP_LWIN, 1);
    const TowerGeom& have = a.geom[t];
    for (int i = 0; i < 3; ++i)
      ship = ship && have.L[i] == want.L[i] && have.Sc[i] == want.Sc[i] && have.NC[i] == want.NC[i] && have.nb[i] == want.nb[i] &&
             have.pk[i] == want.pk[i] && have.ps[i] == want.ps[i] && have.pp[i] == want.pp[i];
    ship = ship && a.geom[0].L[0] == other.L[0] && a.x0_cols == other.L[0] + wave_tower_geom(1, SHIP_LWIN, 1).L[0];
  }
  const dim3 gr(grid), bl(SNV_THREADS);
  if (a.phase == 1) {
    if (ship && t == 0) hipLaunchKernelGGL((snv_tower_wave<1, 9, 0, 0, 1>), gr, bl, lds_bytes, stream, a);
    else if (ship) hipLaunchKernelGGL((snv_tower_wave<1, 9, 0, 1, 2>), gr, bl, lds_bytes, stream, a);
    else hipLaunchKernelGGL((snv_tower_wave<1, 9, 0>), gr, bl, lds_bytes, stream, a);      // plan_wave_geometry: nine blocks
  } else {
    if (ship && t == 0) hipLaunchKernelGGL((snv_tower_wave<2, 8, 4, 0, 6>), gr, bl, lds_bytes, stream, a);
    else if (ship) hipLaunchKernelGGL((snv_tower_wave<2, 8, 3, 1, 5>), gr, bl, lds_bytes, stream, a);
    else {
      set_error("internal: short-stage wave launch without the shipped geometry");
      return MURAL_E_INVALID;
    }
  }
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural
