// Generic fp32 Conv1d (direct form, vector ALU): one workgroup = 64 output positions x all output channels of one
// batch row.  The input span of the tile (all input channels) is staged once in LDS with the optional BN(+ReLU) pre-op
// and nearest-neighbour upsampling applied; each wave then walks output-channel groups of COG channels, the weights of a
// group arriving through wave-uniform (scalar) loads, so the inner loop is 1 LDS read + COG FMAs per (ci, tap).
// Used by the INDEL U-Net (reference MuRaL/model/model_indel.py:6-176: k=7/5/1 convs, strides 1/4/5/2, Upsample,
// SiLU / ReLU / Softplus, residual adds).
#include <cstdlib>

#include "conv1d.h"

namespace mural {

// Activations on the hardware transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each): the accurate libm forms
// cost as many instructions per output as the 8-channel convolutions themselves.  Relative error ~1e-6, far inside the
// 1e-4 / 1e-5 parity budget of the INDEL scores (tests/test_gpu_indel.py).
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_SILU: return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
    case ACT_SOFTPLUS: {                                        // torch.nn.Softplus(beta=1, threshold=20)
      const float e = __expf(v);
      return v > 20.f ? v : (v < -15.f ? e : __logf(1.f + e));  // log(1 + e) = e to fp32 precision below -15
    }
    default: return v;
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int COG, int KT, bool WIDE>   // KT: compile-time tap count (0 = runtime a.K)
// WIDE = false: workgroup = 64 output positions, the 4 waves share the output-channel groups;
// WIDE = true : workgroup = 256 output positions, each wave takes 64 of them and walks every channel group (used when
//               there are fewer than 4 groups, so that no wave idles on the 8/16-channel layers)
__global__ __launch_bounds__(256) void conv1d_kernel(const Conv1dArgs a, const float* __restrict__ wt,
                                                      const float* __restrict__ bias) {
  extern __shared__ float tile[];   // [Cin][TWp]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Short rows (Lout <= 32, the deep U-Net levels): RT = a.rt batch rows share one 64-lane tile, lane = (row, position);
  // otherwise (RT = 1) a workgroup covers 64 / 256 positions of one row.
  const int RT = WIDE ? 1 : a.rt;
  const int b0 = blockIdx.y * RT;
  constexpr int TLB = WIDE ? 256 : 64;
  const int l0 = blockIdx.x * TLB;
  const int TW = ((RT > 1 ? a.Lout : TLB) - 1) * a.stride + a.K;   // input span of one row of the tile
  // Row pitch.  Stride 4 with seven taps (the 8 -> 16 encoder conv of the U-Net): a lane's taps are the 7 floats from 4 * position
  // on -- two aligned 16-byte reads on a pitch of a multiple of 4 (lanes 16 bytes apart: conflict-free), where seven 4-byte reads at a
  // lane stride of 4 floats hit 8 of the 32 banks (r04 PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.64 for this launch).
  // Otherwise an odd pitch: channel rows start on different banks.
  const bool vec4 = KT == 7 && a.stride == 4 && RT == 1;
  const int TWp = vec4 ? ((TW + 4) & ~3) : (TW | 1);
  const int in0 = l0 * a.stride - a.pad;       // first (virtual, upsampled) input index of the tile
  const int Lv = a.Lin * a.up;
  const size_t in_bytes = (size_t)a.B * a.Cin * a.Lin * sizeof(float);
  if (a.up == 1 && in_bytes < (size_t(1) << 31)) {
    // 16-byte pieces of the input rows (4-byte aligned: a strided tile starts anywhere), all of a wave's pieces of a channel row in
    // flight together: the 4-byte form below walks a 1027-column row of the stride-4 encoder conv in 4 rounds of 4 loads -- eight
    // serial round trips per workgroup for its two channels, most of that launch's 318 us (tools/phase notes in DESIGN.md 3.3).
    // Pieces that cross a row end are refused here (the descriptor's offset is replaced by one past its range: zeros) and patched
    // element by element afterwards.
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)in_bytes, 0x00020000);
    const int nq = (TW + 3) >> 2;
    constexpr int UNQ = 5;
    for (int rc = wave; rc < RT * a.Cin; rc += 4) {
      const int r = RT > 1 ? rc / a.Cin : 0, ci = rc - r * a.Cin;
      const bool row_ok = b0 + r < a.B;
      const uint32_t soff = (uint32_t)(((row_ok ? b0 + r : 0) * a.Cin + ci) * a.Lin) * 4u;
      const float ps = a.pre_s ? a.pre_s[ci] : 1.f, pt = a.pre_t ? a.pre_t[ci] : 0.f;
      const bool has_pre = a.pre_s != nullptr || a.pre_t != nullptr || a.pre_relu != 0;
      float* trow = tile + rc * TWp;
      for (int q0 = lane; q0 < nq; q0 += 64 * UNQ) {
        f32x4 v[UNQ];
#pragma unroll
        for (int u = 0; u < UNQ; ++u) {
          const int q = q0 + 64 * u, pos = in0 + 4 * q;
          const bool full = row_ok & (q < nq) & (pos >= 0) & (pos + 3 < Lv);
          uint32_t off = (uint32_t)pos * 4u;
          asm volatile("" : "+v"(off));
          v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, full ? off : 0x80000000u, soff, 0));
        }
#pragma unroll
        for (int u = 0; u < UNQ; ++u) {
          const int q = q0 + 64 * u, pos = in0 + 4 * q;
          if (q >= nq) continue;
          const bool full = row_ok & (pos >= 0) & (pos + 3 < Lv);
          if (!full && row_ok) {                          // a piece across a row end (a lane or two per row): element by element
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int pe = pos + e;
              v[u][e] = (pe >= 0 && pe < Lv) ? a.in[(size_t)soff / 4 + pe] : 0.f;
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = 4 * q + e, pe = pos + e;
            float xv = v[u][e];
            if (has_pre) {                               // zero padding applies after the pre-op, like nn.Conv1d after a BatchNorm
              const bool in = row_ok & (pe >= 0) & (pe < Lv);
              if (a.pre_relu) xv = fmaxf(xv, 0.f);
              xv = in ? fmaf(ps, xv, pt) : 0.f;
            }
            if (j < TW) trow[j] = xv;
          }
        }
      }
    }
  } else
  for (int rc = wave; rc < RT * a.Cin; rc += 4) {
    const int r = RT > 1 ? rc / a.Cin : 0, ci = rc - r * a.Cin;
    const bool row_ok = b0 + r < a.B;
    const float* src = a.in + ((size_t)(row_ok ? b0 + r : 0) * a.Cin + ci) * a.Lin;
    const float ps = a.pre_s ? a.pre_s[ci] : 1.f, pt = a.pre_t ? a.pre_t[ci] : 0.f;
    // UN loads of a lane in flight before the first is used: a round that waits for its own load costs a global round trip, and
    // a strided 256-column tile has 17 rounds per (row, channel)
    constexpr int UN = 4;
    for (int j0 = lane; j0 < TW; j0 += 64 * UN) {
      float xv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int v = in0 + j0 + 64 * u;
        const bool ok = row_ok && j0 + 64 * u < TW && v >= 0 && v < Lv;
        xv[u] = src[ok ? (a.up == 1 ? v : v / a.up) : 0];
        if (!ok) xv[u] = 0.f;                  // zero padding (applied after the pre-op, like nn.Conv1d after a BN)
        else {
          if (a.pre_relu) xv[u] = fmaxf(xv[u], 0.f);
          xv[u] = fmaf(ps, xv[u], pt);
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (j0 + 64 * u < TW) tile[rc * TWp + j0 + 64 * u] = xv[u];
    }
  }
  __syncthreads();
  int lloc = WIDE ? 64 * wave + lane : lane;   // output position inside the tile
  int b = b0, trow0 = 0;
  bool live = true;
  if (RT > 1) {
    const int r = lane / a.Lout;
    lloc = lane - r * a.Lout;
    b = b0 + r;
    live = r < RT && b < a.B;
    trow0 = (live ? r : 0) * a.Cin * TWp;
  }
  const int l = l0 + lloc;
  const int ngroups = a.Cout / COG;
  for (int cg = WIDE ? 0 : wave; cg < ngroups; cg += WIDE ? 1 : 4) {
    // two output channels per v_pk_fma_f32: the input value is broadcast to both halves, the weight pair is a scalar pair
    f32x2 acc[COG / 2];
#pragma unroll
    for (int c = 0; c < COG / 2; ++c)
      acc[c] = bias ? f32x2{bias[cg * COG + 2 * c], bias[cg * COG + 2 * c + 1]} : f32x2{0.f, 0.f};
    const float* __restrict__ w = wt + cg * COG;
    const int K = KT ? KT : a.K;
#pragma unroll 2
    for (int ci = 0; ci < a.Cin; ++ci) {   // unrolled: several input channels' scalar weight loads in flight together
      const float* trow = tile + trow0 + ci * TWp + lloc * a.stride;
      if (KT) {
        float x[KT ? KT : 1];
        bool done = false;
        if constexpr (KT == 7) {
          if (vec4) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(trow), v = *reinterpret_cast<const f32x4*>(trow + 4);
            x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w;
            x[4] = v.x; x[5] = v.y; x[6] = v.z;
            done = true;
          }
        }
        if (!done) {
#pragma unroll
          for (int k = 0; k < KT; ++k) x[k] = trow[k];      // all taps' LDS reads and scalar weight loads in flight together
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const float* __restrict__ wk = w + (size_t)(ci * KT + k) * a.Cout;   // wave-uniform address: scalar loads
          const f32x2 x2 = {x[k], x[k]};
#pragma unroll
          for (int c = 0; c < COG / 2; ++c) acc[c] = __builtin_elementwise_fma(x2, f32x2{wk[2 * c], wk[2 * c + 1]}, acc[c]);
        }
      } else {
        for (int k = 0; k < K; ++k) {
          const float xv = trow[k];
          const f32x2 x2 = {xv, xv};
          const float* __restrict__ wk = w + (size_t)(ci * K + k) * a.Cout;
#pragma unroll
          for (int c = 0; c < COG / 2; ++c) acc[c] = __builtin_elementwise_fma(x2, f32x2{wk[2 * c], wk[2 * c + 1]}, acc[c]);
        }
      }
    }
    if (live && l < a.Lout) {
      // the residuals of all COG channels are requested together (one branch per tensor, not one -- with a full wait behind its
      // load -- per element)
      const size_t o0 = ((size_t)b * a.Cout + cg * COG) * a.Lout + l;
      float e[COG];
#pragma unroll
      for (int c = 0; c < COG; ++c) e[c] = 0.f;
      if (a.res1) {
#pragma unroll
        for (int c = 0; c < COG; ++c) e[c] = a.res1[o0 + (size_t)c * a.Lout];
      }
      if (a.res2) {
        float e2[COG];
#pragma unroll
        for (int c = 0; c < COG; ++c) e2[c] = a.res2[o0 + (size_t)c * a.Lout];
#pragma unroll
        for (int c = 0; c < COG; ++c) e[c] += e2[c];
      }
#pragma unroll
      for (int c = 0; c < COG; ++c) a.out[o0 + (size_t)c * a.Lout] = apply_act((c & 1) ? acc[c >> 1].y : acc[c >> 1].x, a.act) + e[c];
    }
  }
}

using Conv1dFn = void (*)(const Conv1dArgs, const float*, const float*);

template <int COG, bool WIDE>
static Conv1dFn pick_taps(int K) {
  switch (K) {
    case 1: return conv1d_kernel<COG, 1, WIDE>;
    case 3: return conv1d_kernel<COG, 3, WIDE>;
    case 5: return conv1d_kernel<COG, 5, WIDE>;
    case 7: return conv1d_kernel<COG, 7, WIDE>;
    default: return conv1d_kernel<COG, 0, WIDE>;
  }
}

template <bool WIDE>
static Conv1dFn pick_kernel(int K, int cog) {
  return cog == 16 ? pick_taps<16, WIDE>(K) : (cog == 8 ? pick_taps<8, WIDE>(K) : pick_taps<4, WIDE>(K));
}

int launch_conv1d(const Conv1dArgs& a, hipStream_t stream) {
  if (a.B == 0 || a.Lout == 0) return MURAL_OK;
  MURAL_REQUIRE(a.up >= 1 && a.stride >= 1 && a.K >= 1, "conv1d: bad geometry");
  static const bool use_mfma = !(dev_env("MURAL_CONV1D_MFMA") && atoi(dev_env("MURAL_CONV1D_MFMA")) == 0);   // A/B switch of the tools
  // measured at 2048 rows (tools/gpu_debug_conv1d.py): the implicit GEMM wins on short rows (<= 128 columns: 2-4 x) and on long
  // rows with a deep reduction (Cin * K >= 224 then; >= 160 since the MFMA kernel stopped waiting per element: 24 -> 32 k7 at 400 columns
  // 144 -> 66 us); long rows with few input channels are bound by the output stream and the direct
  // kernel's 64-column tiles keep more of them in flight.  A small launch (a training batch of 128 rows at the U-Net's 400-column
  // level: the direct kernel's 256-column tiles make 256 workgroups of a 70 us chain each) goes to the implicit GEMM too, and so do
  // the upsampling convs with a medium reduction (measured on the training step, tools/r4_step_classes.py: 74 -> 22, 30 -> 17, 55 -> 42 us).
  // long rows, few channels, several taps at stride 1: the barrier-free kernel (neither tiled kernel covers its stage / compute /
  // store phases on these; measured on the training step: 35-53 us -> see DESIGN.md)
  static const bool use_direct = !(dev_env("MURAL_CONV1D_DIRECT") && atoi(dev_env("MURAL_CONV1D_DIRECT")) == 0);
  // (measured at 2048 rows too, tools/gpu_debug_conv1d.py: 16 -> 20 k7 at 517 columns 258 / 188 / 159 us valu / mfma / direct, 4 -> 32 k7
  // at 600 columns 83 / 114 / 50; three taps fill 3 of its 8 tap slots and lose to the vector ALU)
  if (use_direct && use_mfma && a.K >= 5 && (int64_t)a.B * a.Lout >= 32768 && a.Lout >= 64 && conv1d_direct_supported(a))
    return launch_conv1d_direct(a, stream);
  static const bool all_mfma = dev_env("MURAL_CONV1D_MFMA") && atoi(dev_env("MURAL_CONV1D_MFMA")) == 2;
  const bool small_launch = (int64_t)a.B * a.Lout <= 128 * 512 && a.Lout <= 512;
  if (use_mfma && conv1d_mfma_supported(a) &&
      (all_mfma || a.Lout <= 128 || a.Cin * a.K >= 160 || small_launch || (a.up > 1 && a.Cin * a.K >= 128 && (int64_t)a.B * a.Lout <= 128 * 2048) ||
       (a.Cin * a.K >= 96 && a.Cout >= 32 && a.Lout <= 1024)))
    return launch_conv1d_mfma(a, stream);
  return launch_conv1d_valu(a, stream);
}

int launch_conv1d_valu(const Conv1dArgs& a_in, hipStream_t stream) {
  Conv1dArgs a = a_in;
  if (a.B == 0 || a.Lout == 0) return MURAL_OK;
  MURAL_REQUIRE(a.Cout % 4 == 0, "conv1d: Cout must be a multiple of 4 (got %d)", a.Cout);
  MURAL_REQUIRE(a.up >= 1 && a.stride >= 1 && a.K >= 1, "conv1d: bad geometry");
  // output channels per accumulator group: 16 halves the LDS reads per FMA (used when a wave still gets a group)
  const int cog = (a.Cout % 16 == 0) ? 16 : (a.Cout % 8 == 0 ? 8 : 4);
  const int ngroups = a.Cout / cog;
  auto pitch = [&](size_t span, bool one_row) -> size_t {      // the kernel's row pitch (see there)
    return (a.K == 7 && a.stride == 4 && one_row) ? ((span + 4) & ~(size_t)3) : (span | 1);
  };
  const size_t lds_wide = (size_t)a.Cin * pitch((size_t)255 * a.stride + a.K, true) * sizeof(float);
  const bool wide = ngroups < 4 && lds_wide <= 64 * 1024 && a.Lout > 64;
  a.rt = 1;
  if (!wide && a.Lout <= 32) {                       // short rows: pack batch rows into the 64 lanes of a tile
    a.rt = 64 / a.Lout;
    while (a.rt > 1 && (size_t)a.rt * a.Cin * ((((size_t)a.Lout - 1) * a.stride + a.K) | 1) * sizeof(float) > 60 * 1024) --a.rt;
  }
  const int span = (a.rt > 1 ? a.Lout - 1 : (wide ? 255 : 63)) * a.stride + a.K;
  const size_t lds = (size_t)a.rt * a.Cin * pitch((size_t)span, a.rt == 1) * sizeof(float);
  MURAL_REQUIRE(lds <= 160 * 1024, "conv1d: input tile of %zu bytes exceeds LDS", lds);
  const int tlb = wide ? 256 : 64;
  const dim3 grid(a.rt > 1 ? 1 : (a.Lout + tlb - 1) / tlb, (a.B + a.rt - 1) / a.rt);
  Conv1dFn fn = wide ? pick_kernel<true>(a.K, cog) : pick_kernel<false>(a.K, cog);
  if (lds > 64 * 1024)
    MURAL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        160 * 1024));
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, stream, a, a.wt, a.bias);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

// ------------------------------------------------------------------------------------------------ fused ConvBlock
// One workgroup = 256 consecutive positions of one batch row; a lane owns one position and ALL channels of it: the k=5
// conv fills 2C accumulators from the LDS tile (weights as scalar pairs, v_pk_fma_f32), SiLU runs on them in place, the
// 1x1 conv contracts them to C outputs, the block input is added back from the tile.  HBM traffic: read x, write out
// (+ read res2) instead of also writing and re-reading the 2C-channel intermediate.
int g_convblock8_form = -1;      // validation hook (mural_debug_convblock): 0 vector-ALU form, 1 split form, -1 the environment's choice
constexpr int CB_FRONT_FLOATS = 2048;          // front input tile: Cf x (262 / up + 3) floats
constexpr int CB_FRONT_OUT = 252;              // output positions per workgroup of the front variant
constexpr int CB_FRONT_OUT_POLY = 248;         // ... of the split form with the polyphase front on the matrix cores

// floats of the split form's front-input region (also holds the tail's 32 partial maxima)
__host__ __device__ inline int convblock_front_floats(int Cf, int f_up, bool front) {
  return front ? ((Cf * (262 / f_up + 3) + 3) & ~3) : 32;
}

// MF (8 channels): everything behind the front -- the k=5 conv 8 -> 16, the 1x1 conv 16 -> 8 and the tail's two 1x1 convs -- runs on
// v_mfma_f32_16x16x4_f32 while front, SiLU and the adds stay on the vector ALU.  fp32 MFMA and v_pk_fma_f32 have the SAME peak rate,
// so moving the whole block to the matrix pipe gains nothing (tried in round 3: slower, removed in round 6); splitting it does: a SIMD's eight waves
// are in different phases, the two pipes work side by side, and each carries about half of what the vector ALU alone carried.
//   k=5 conv  D[16 hidden][16 positions] += W5[hidden][(tap, ci)] x[ci][pos + tap - 2], 10 k-steps, four position blocks per wave
//             (four independent accumulators); the B operand is one ds_read_b32 per MFMA at an immediate offset.
//   1x1 conv  blocks taken in PAIRS (b, b + 2): rows 0-7 of the output tile are the 8 channels of block b, rows 8-15 those of block
//             b + 2 (A = [W1 0] against block b's SiLU values, then [0 W1] against block b + 2's) -- no padding rows, and the
//             accumulator layout (lane (n, kk): rows 4 kk + q) of the conv in front IS the B layout with the k index permuted.
//   tail      the paired layout is closed under a 1x1 conv 8 -> 8 with A = diag(W, W): 4 MFMAs per pair and conv.
template <int C, bool TAIL, bool FRONT, bool MF>
__global__ __launch_bounds__(256, MF ? (FRONT ? 6 : (TAIL ? 7 : 8)) : 1) void convblock_kernel(const ConvBlockArgs a, const float* __restrict__ w5,
                                                        const float* __restrict__ b5, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const float* __restrict__ ta_w,
                                                        const float* __restrict__ ta_b, const float* __restrict__ tb_w,
                                                        const float* __restrict__ tb_b, const float* __restrict__ f_w,
                                                        const float* __restrict__ f_b, unsigned long long* stamps) {
  // diagnostic (tools/phase_stamps_cb8.py, split form only): wall-clock ticks (100 MHz) of the workgroup's first thread at entry, front
  // input staged, block input ready, SiLU done, block output ready, exit -- 8 words per workgroup
#define CB_STAMP(id)                                                                                              \
  if (MF && stamps && threadIdx.x == 0) {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    const size_t wg_ = (size_t)blockIdx.y * gridDim.x + blockIdx.x;                                               \
    if (wg_ < 65536) stamps[8 * wg_ + (id)] = __builtin_amdgcn_s_memrealtime();                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
  }
  CB_STAMP(0);
  if (MF && stamps && threadIdx.x == 0) {      // word 7: where the workgroup runs (HW_ID: CU / SE, XCC_ID)
    const size_t wg_ = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (wg_ < 65536)
      stamps[8 * wg_ + 7] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |          // HW_REG_HW_ID, 32 bits
                            ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32);  // HW_REG_XCC_ID
  }
  static_assert(!MF || C == 8, "the matrix-core form serves the 8-channel block: its hidden width is one 16-row MFMA tile");
  // pitch: odd for the vector-ALU form (lane = position); = 16 (mod 32) for the matrix form, whose operand reads have the lane
  // groups kk = 0, 1 of a half wave 1 x pitch apart
  constexpr int C2 = 2 * C, TW = 256 + 4, TWp = MF ? 272 : (TW | 1);
  // the vector form keeps its three arrays static; the split form carves tile | front input | polyphase weights out of ONE dynamic
  // allocation sized by the host for what the launch uses (encoder 13 KB, decoder 19 KB instead of 22.5 KB: the LDS granule made that
  // six workgroups per CU)
  extern __shared__ __attribute__((aligned(16))) float cb_dyn[];
  __shared__ float tile_s[MF ? 4 : C * TWp];
  float* const tile = MF ? cb_dyn : tile_s;
  // SiLU outputs parked for the rolled 1x1 loop, HP of the 2C rows at a time (a lane only ever touches its own column, so the
  // passes need no barrier): at 8 channels two passes of 8 rows keep the buffer at 8 KB and the workgroup at 17 KB of LDS = 8
  // workgroups per CU, which is what hides the scalar weight loads of the inner loops
  constexpr int HP = C == 8 ? 8 : C2;
  __shared__ float hs_s[MF ? 4 : HP * 256];
  float* const hs = MF ? cb_dyn + C * TWp : hs_s;
  static_assert(!FRONT || HP * 256 >= CB_FRONT_FLOATS, "the front input tile borrows the SiLU buffer");
  constexpr int PW_FLOATS = (FRONT && C == 8) ? 4 * 16 * 3 * 8 : 4;      // polyphase front weights of the four phases (Cf <= 16)
  __shared__ __attribute__((aligned(16))) float pw_s[MF ? 4 : PW_FLOATS];
  float* const pwS = MF ? hs + convblock_front_floats(a.Cf, a.f_up, FRONT) : pw_s;
  // split form, decoder: the polyphase front itself runs on the matrix cores (POLY, below) -- its weights are A fragments in registers
  const bool POLY = MF && FRONT && a.f_pw != nullptr && a.Cf == 16 && a.f_up == 4;
  if (FRONT && C == 8 && !POLY && a.f_pw != nullptr && a.Cf * 3 * C * 4 <= PW_FLOATS)
    for (int i = threadIdx.x; i < a.Cf * 3 * C * 4; i += 256) pwS[i] = a.f_pw[i];      // visible behind the front's first barrier
  float* fin = hs;                             // front input tile: dead before the first SiLU output is parked (a barrier in between),
                                               // and 8 KB less LDS is two more workgroups per CU to hide the scalar weight loads
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  // without a front a workgroup covers 256 positions (tile = positions l0-2 .. l0+257); with one it covers CB_FRONT_OUT =
  // 252: lane tid computes the block INPUT at position l0-2+tid (256 of them, no second pass for the halo) and, for
  // 2 <= tid < 254, the block OUTPUT at that same position
  // (POLY: 248 outputs = 62 source columns + one of halo each side = the 64 source columns of four 16-column MFMA blocks, one per wave;
  // the tile then starts at position l0 - 4, not l0 - 2)
  const int OUTW = FRONT ? (POLY ? CB_FRONT_OUT_POLY : CB_FRONT_OUT) : 256;
  const int l0 = blockIdx.x * OUTW;
  float af[MF && FRONT ? 2 : 1][MF && FRONT ? 12 : 1];      // POLY: A fragments of the front, lane (row 16 mb + n16 = 4 co + phase, kk)
  if (!FRONT) {
    const float* src = a.x + (size_t)b * C * a.L;
    constexpr int UN = 4;                        // loads of a thread in flight (a round per load = a global round trip per round)
    for (int i0 = tid; i0 < C * TW; i0 += 256 * UN) {
      float v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + 256 * u;
        const int ci = i / TW, j = i - ci * TW;
        const int l = l0 - 2 + j;
        const bool ok = i < C * TW && l >= 0 && l < a.L;
        v[u] = src[ok ? (size_t)ci * a.L + l : 0];
        if (!ok) v[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + 256 * u;
        if (i < C * TW) {
          const int ci = i / TW, j = i - ci * TW;
          tile[ci * TWp + j] = v[u];
        }
      }
    }
  } else {
    // the block input of positions l0-2 .. l0+253 needs the (virtual, upsampled) front input at l0-5 .. l0+256
    const int up = a.f_up;
    const int v0 = l0 - 5;                                   // first virtual index (may be negative)
    const int r0 = v0 >= 0 ? v0 / up : -((-v0 + up - 1) / up);   // floor(v0 / up)
    const int span = (l0 + 256) / up - r0 + 1;               // real columns staged per channel
    const float* fsrc = a.f_in + (size_t)b * a.Cf * a.Lf;
    constexpr int UN = 5;                                    // 16 x 67 (decoder) and 4 x 262 (encoder) floats = 5 per thread
    if (a.symtab != nullptr) {
      // front input from the packed genome (Cf == 4, up == 1): symbols of columns r0 - h .. r0 + span - 1 + h (h = sym_taps / 2)
      // into LDS bytes (strand-oriented: '-' rows read the window backwards and complemented), then per staged column the
      // sym_taps table rows of its neighbourhood; columns outside [0, Lf) hold the zero padding of the front conv, symbols
      // outside it the zero padding of the table layer
      uint8_t* symb = reinterpret_cast<uint8_t*>(tile);      // the block-input tile is written only after the front
      float* stab = tile + 128;                              // [15][sym_taps][4] table + bias[4], behind the symbol bytes
      for (int i = tid; i < 15 * a.sym_taps * 4; i += 256) stab[i] = a.symtab[i];
      if (tid < 4) stab[15 * 15 * 4 + tid] = a.sym_bias[tid];
      const int h = a.sym_taps >> 1;
      const int nsym = span + 2 * h;
      const int64_t ws = a.g_pos[b] + a.g_off;
      const bool neg = a.g_strand[b] != 0;
      for (int i = tid; i < nsym; i += 256) {
        const int j = r0 - h + i;                            // window column
        uint32_t sy = SYM_PAD;
        if (j >= 0 && j < a.Lf) {
          sy = genome_sym_iupac(a.genome, neg ? ws + (a.Lf - 1 - j) : ws + j);
          if (neg) sy = sym_complement(sy);
        }
        symb[i] = (uint8_t)sy;
      }
      __syncthreads();
      // a thread owns a staged column and all four channels of it: one symbol read and one 16-byte table read per tap (the same
      // sums in the same order as one thread per (channel, column), a quarter of the instructions -- this kernel is bound by
      // vector-instruction issue, PMC: SQ_ACTIVE_INST_VALU 0.97)
      for (int rr = tid; rr < span; rr += 256) {
        const int r = r0 + rr;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (r >= 0 && r < a.Lf) {
          v = *reinterpret_cast<const f32x4*>(stab + 15 * 15 * 4);
          for (int k = 0; k < a.sym_taps; ++k) {
            const uint32_t sy = symb[rr + k];
            if (sy != SYM_PAD) v += *reinterpret_cast<const f32x4*>(stab + (sy * a.sym_taps + k) * 4);
          }
        }
        fin[rr] = v.x;
        fin[span + rr] = v.y;
        fin[2 * span + rr] = v.z;
        fin[3 * span + rr] = v.w;
      }
    } else if ((a.Cf & (a.Cf - 1)) == 0 && a.Cf <= 64 && span <= UN * (256 / a.Cf)) {
      // a power-of-two channel count: 256 / Cf threads per channel walk its span (no division per element -- a run-time one costs
      // ~20 vector instructions, five of them per thread were a tenth of this kernel's instruction count)
      const int tpc = 256 / a.Cf, sh = 31 - __builtin_clz(tpc);
      const int ci = tid >> sh, rr0 = tid & (tpc - 1);
      float v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int rr = rr0 + tpc * u;
        const int r = r0 + rr;
        const bool ok = rr < span && r >= 0 && r < a.Lf;
        v[u] = fsrc[ok ? (size_t)ci * a.Lf + r : 0];
        if (!ok) v[u] = 0.f;                                 // zero padding of the upsampled tensor
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (rr0 + tpc * u < span) fin[ci * span + rr0 + tpc * u] = v[u];
    } else
    for (int i0 = tid; i0 < a.Cf * span; i0 += 256 * UN) {
      float v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + 256 * u;
        const int ci = i / span, rr = i - ci * span;
        const int r = r0 + rr;
        const bool ok = i < a.Cf * span && r >= 0 && r < a.Lf;
        v[u] = fsrc[ok ? (size_t)ci * a.Lf + r : 0];
        if (!ok) v[u] = 0.f;                                 // zero padding of the upsampled tensor
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (i0 + 256 * u < a.Cf * span) fin[i0 + 256 * u] = v[u];
    }
    // (requested behind the staging loads, in front of their barrier: not live across the staging)
    if constexpr (MF && FRONT) {
      if (POLY) {
        const int n16f = threadIdx.x & 15, kkf = (threadIdx.x >> 4) & 3;
  #pragma unroll
        for (int mb = 0; mb < 2; ++mb)
  #pragma unroll
          for (int s = 0; s < 12; ++s) {
            const int d = s >> 2, ci = 4 * (s & 3) + kkf, row = 16 * mb + n16f, co = row >> 2, ph = row & 3;
            af[mb][s] = a.f_pw[(((size_t)ph * 16 + ci) * 3 + d) * C + co];
          }
      }
    }
    __syncthreads();
    CB_STAMP(1);
    if (MF && POLY) {
      if constexpr (MF) {
        // polyphase front as a GEMM on the SOURCE columns: rows = (channel, phase) -- 32 = two M-blocks --, k = (tap d of 3, 16 input
        // channels) = 12 k-steps, columns = this wave's 16 source columns l0 / 4 - 1 + 16 wave + n.  Lane (n, kk) ends up with the four
        // phases = four consecutive positions of channel 4 mb + kk: one 16-byte store into the tile (origin l0 - 4).  Against the
        // vector form: 24 MFMAs for 192 packed FMAs (the same pipe time), but 12 LDS reads per wave instead of 144 -- the 96 broadcast
        // 16-byte weight reads alone kept the CU's one LDS pipe as busy as its SIMDs.
        const int wvf = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int n16f = tid & 15, kkf = (tid >> 4) & 3;
        const float* sp = fin + kkf * span + 16 * wvf + n16f;      // x[ci = 4 cq + kk][source + d - 1] = sp[4 cq span + d]
        f32x4 accf[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 12; ++s) {
          const float bv = sp[4 * (s & 3) * span + (s >> 2)];
          accf[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][s], bv, accf[0], 0, 0, 0);
          accf[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][s], bv, accf[1], 0, 0, 0);
        }
        const int j = 64 * wvf + 4 * n16f;                         // tile entry of phase 0 (position l0 - 4 + j)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const int co = 4 * mb + kkf;
          const float fb = f_b[co];
          float o4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int l = l0 - 4 + j + r;
            o4[r] = (l >= 0 && l < a.L) ? accf[mb][r] + fb : 0.f;      // the k=5 conv zero-pads ITS input
          }
          *reinterpret_cast<f32x4*>(tile + co * TWp + j) = f32x4{o4[0], o4[1], o4[2], o4[3]};
        }
      }
    } else if (a.f_pw) {
      // polyphase front (f_up == 4, workgroup origin l0 a multiple of 4): position l only sees the source columns l / 4 + d,
      // d in {-1, 0, 1}, with the taps that share a column summed on the host per phase l % 4.  Wave w takes the tile entries
      // j = 4 lane + w -- one phase per wave, so the phase's weights stay wave-uniform (scalar loads) -- and does 3 / 7 of the
      // multiply-adds of the direct form below.
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler too: scalar weight loads
      const int j = 4 * (tid & 63) + wv;
      const int l = l0 - 2 + j;
      const int ph = (wv + 2) & 3;                           // l % 4 for every l >= 0 of the tile (l0 % 4 == 0)
      const int i0 = (l >= 0 ? (l >> 2) : 0) - 1 - r0;       // fin column of source column l / 4 - 1
      f32x2 t[C / 2];
#pragma unroll
      for (int c = 0; c < C / 2; ++c) t[c] = f32x2{f_b[2 * c], f_b[2 * c + 1]};
      if (C == 8 && a.Cf * 3 * C * 4 <= PW_FLOATS) {
        // the phase's weights from LDS (staged once per workgroup, broadcast 16-byte reads): as scalar loads they are three
        // waited-for round trips per 12 packed FMAs (the scalar registers hold two channels' worth at most), from LDS a dozen
        // reads are in flight
        const float* pwl = pwS + ph * a.Cf * 3 * C;
#pragma unroll 4
        for (int ci = 0; ci < a.Cf; ++ci) {
          const float* frow = fin + ci * span + i0;
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(pwl + (ci * 3 + d) * C), w1 = *reinterpret_cast<const f32x4*>(pwl + (ci * 3 + d) * C + 4);
            const float xv = frow[d];
            const f32x2 x2 = {xv, xv};
            t[0] = __builtin_elementwise_fma(x2, f32x2{w0.x, w0.y}, t[0]);
            t[1] = __builtin_elementwise_fma(x2, f32x2{w0.z, w0.w}, t[1]);
            t[2] = __builtin_elementwise_fma(x2, f32x2{w1.x, w1.y}, t[2]);
            t[3] = __builtin_elementwise_fma(x2, f32x2{w1.z, w1.w}, t[3]);
          }
        }
      } else {
      const float* __restrict__ pw = a.f_pw + (size_t)ph * a.Cf * 3 * C;
#pragma unroll 2
      for (int ci = 0; ci < a.Cf; ++ci) {
        const float* frow = fin + ci * span + i0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const float* __restrict__ wk = pw + (size_t)(ci * 3 + d) * C;    // wave-uniform: scalar loads
          const float xv = frow[d];
          const f32x2 x2 = {xv, xv};
#pragma unroll
          for (int c = 0; c < C / 2; ++c) t[c] = __builtin_elementwise_fma(x2, f32x2{wk[2 * c], wk[2 * c + 1]}, t[c]);
        }
      }
      }
      const bool in = l >= 0 && l < a.L;
#pragma unroll
      for (int c = 0; c < C; ++c) tile[c * TWp + j] = in ? ((c & 1) ? t[c >> 1].y : t[c >> 1].x) : 0.f;
    } else {
      const int j = tid;
      const int l = l0 - 2 + j;
      f32x2 t[C / 2];
#pragma unroll
      for (int c = 0; c < C / 2; ++c) t[c] = f32x2{f_b[2 * c], f_b[2 * c + 1]};
      int ridx[7];
      if (up == 1) {                                         // first encoder level: no upsampling, no division
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          const int v = l - 3 + k;
          ridx[k] = (v >= 0 ? v : r0) - r0;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          const int v = l - 3 + k;                           // virtual input index of tap k, >= v0
          ridx[k] = (v >= 0 ? v / up : r0) - r0;             // v < 0 only if r0 < 0: column 0 then holds a zero
        }
      }
#pragma unroll 1
      for (int ci = 0; ci < a.Cf; ++ci) {
        const float* frow = fin + ci * span;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          const float* __restrict__ wk = f_w + (size_t)(ci * 7 + k) * C;   // wave-uniform: scalar loads
          const float xv = frow[ridx[k]];
          const f32x2 x2 = {xv, xv};
#pragma unroll
          for (int c = 0; c < C / 2; ++c) t[c] = __builtin_elementwise_fma(x2, f32x2{wk[2 * c], wk[2 * c + 1]}, t[c]);
        }
      }
      const bool in = l >= 0 && l < a.L;                     // the k=5 conv zero-pads ITS input
#pragma unroll
      for (int c = 0; c < C; ++c) tile[c * TWp + j] = in ? ((c & 1) ? t[c >> 1].y : t[c >> 1].x) : 0.f;
    }
  }
  // matrix form with a front: the front fills tile entries 0 .. 255; the taps of the two dead lanes behind the last output reach entries
  // 256 .. 257, and a dead lane's values DO meet live ones here (the paired 1x1 multiplies them by the zero half of A: 0 x NaN) -- so
  // those entries must not be whatever the previous workgroup left in LDS
  if (MF && FRONT && tid < 4 * C) tile[(tid >> 2) * TWp + 256 + (tid & 3)] = 0.f;
  // matrix form: the A fragments (lane (m = lane % 16, kk = lane / 16) holds A[m][kk] of a k-step), requested behind the front (its
  // registers are free again) and in front of its barrier
  const int n16 = tid & 15, kk = (tid >> 4) & 3;
  float a5[MF ? 10 : 1], a1a[MF ? 4 : 1], a1b[MF ? 4 : 1];
  f32x4 bias5 = {0.f, 0.f, 0.f, 0.f}, bias1 = bias5;
  if constexpr (MF) {
#pragma unroll
    for (int s = 0; s < 10; ++s) a5[s] = w5[((4 * (s & 1) + kk) * 5 + (s >> 1)) * C2 + n16];      // k = (tap s / 2, ci 4 (s % 2) + kk)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float w = w1[(4 * kk + q) * C + (n16 & 7)];                                            // k = (q, kk): hidden channel 4 kk + q
      a1a[q] = n16 < 8 ? w : 0.f;
      a1b[q] = n16 < 8 ? 0.f : w;
    }
    bias5 = *reinterpret_cast<const f32x4*>(b5 + 4 * kk);
    bias1 = *reinterpret_cast<const f32x4*>(b1 + 4 * (kk & 1));
  }
  __syncthreads();
  CB_STAMP(2);
  if constexpr (MF) {
    // virtual lane t = 64 wave + 16 block + n16 plays the role the lane plays in the vector form (tile index toff(t), position l(t))
    const int t0 = (tid & ~63) + n16;
    auto toff_of = [&](int t) { return (FRONT && !POLY) ? (t >= 2 ? t - 2 : 0) : t; };      // (POLY: the tile starts two positions earlier)
    f32x4 acc[4];
    const float* xb[4];
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) {
      xb[bk] = tile + kk * TWp + toff_of(t0 + 16 * bk);
      acc[bk] = bias5;
    }
#pragma unroll
    for (int s = 0; s < 10; ++s)
#pragma unroll
      for (int bk = 0; bk < 4; ++bk)
        acc[bk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a5[s], xb[bk][4 * (s & 1) * TWp + (s >> 1)], acc[bk], 0, 0, 0);
    float h[4][4];                                 // SiLU values (separate scalars: element-wise updates of the accumulator vectors in place
                                                   // have been miscompiled before)
    // the tail's fragments are requested here: in flight under the MFMAs, not live above them
    float tA[TAIL ? 4 : 1], tB[TAIL ? 4 : 1];
    f32x4 biasA = {0.f, 0.f, 0.f, 0.f}, biasB = biasA;
    if constexpr (TAIL) {
      const bool own = (n16 >> 3) == (kk >> 1);                                                     // diag(W, W)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float wa = ta_w[(4 * (kk & 1) + q) * C + (n16 & 7)], wb = tb_w[(4 * (kk & 1) + q) * C + (n16 & 7)];
        tA[q] = own ? wa : 0.f;
        tB[q] = own ? wb : 0.f;
      }
      biasA = *reinterpret_cast<const f32x4*>(ta_b + 4 * (kk & 1));
      biasB = *reinterpret_cast<const f32x4*>(tb_b + 4 * (kk & 1));
    }
#pragma unroll
    for (int bk = 0; bk < 4; ++bk)
#pragma unroll
      for (int q = 0; q < 4; ++q) h[bk][q] = apply_act(acc[bk][q], ACT_SILU);
    CB_STAMP(3);
    f32x4 o[2] = {bias1, bias1};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int p = 0; p < 2; ++p) o[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1a[q], h[p][q], o[p], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int p = 0; p < 2; ++p) o[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1b[q], h[p + 2][q], o[p], 0, 0, 0);
    // lane (n16, kk) now holds channels 4 (kk % 2) + q of blocks p + 2 (kk / 2), p = 0, 1: + block input + skip
    const int cb = 4 * (kk & 1);
    const size_t rowbase = (size_t)b * C * a.L;
    const uint32_t row_bytes = (uint32_t)C * (uint32_t)a.L * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res2 ? a.res2 + rowbase : w5), 0,
                                                                        a.res2 ? (int)row_bytes : 0, 0x00020000);
    bool live[2];
    uint32_t goff[2];
    float v[2][4];
    {
      float sk[2][4], xin[2][4];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int t = t0 + 16 * (p + 2 * (kk >> 1));
        const int l = FRONT ? l0 - 2 + t : l0 + t;
        live[p] = FRONT ? ((t >= 2) & (t < 2 + OUTW) & (l < a.L)) : (l < a.L);
        uint32_t off = ((uint32_t)cb * (uint32_t)a.L + (uint32_t)l) * 4u;
        off = live[p] ? off : 0x80000000u;
        asm volatile("" : "+v"(off));
        goff[p] = off;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sk[p][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, (uint32_t)q * (uint32_t)a.L * 4u, 0));
          xin[p][q] = tile[(cb + q) * TWp + toff_of(t) + 2];
        }
      }
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[p][q] = (o[p][q] + xin[p][q]) + sk[p][q];
    }
    CB_STAMP(4);
    if constexpr (!TAIL) {
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + rowbase, 0, (int)row_bytes, 0x00020000);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v[p][q]), ro, goff[p], (uint32_t)q * (uint32_t)a.L * 4u, 0);
      CB_STAMP(5);
      if (stamps) {      // (diagnostic only) how long the stores take to drain
        __builtin_amdgcn_s_waitcnt(0);
        CB_STAMP(6);
      }
      return;
    } else {
      f32x4 ta[2] = {biasA, biasA};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < 2; ++p) ta[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(tA[q], v[p][q], ta[p], 0, 0, 0);
      f32x4 u[2] = {biasB, biasB};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < 2; ++p) u[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(tB[q], fmaxf(ta[p][q], 0.f), u[p], 0, 0, 0);
      // Softplus is non-decreasing: reduce the raw values (lanes of one channel: the 16 columns and the two block halves), apply once
      float* red = hs;                             // the front input tile is dead (barrier above)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float m = fmaxf(live[0] ? u[0][q] : -INFINITY, live[1] ? u[1][q] : -INFINITY);
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (n16 == 0 && kk < 2) red[(tid >> 6) * C + cb + q] = m;
      }
      __syncthreads();
      if (tid < C)
        a.tail_max[((size_t)b * gridDim.x + blockIdx.x) * C + tid] =
            apply_act(fmaxf(fmaxf(red[tid], red[C + tid]), fmaxf(red[2 * C + tid], red[3 * C + tid])), ACT_SOFTPLUS);
      CB_STAMP(5);
      return;
    }
  }
  const int toff = FRONT ? (tid >= 2 ? tid - 2 : 0) : tid;    // tile index of this lane's first k=5 tap
  const int l = FRONT ? l0 - 2 + tid : l0 + tid;
  const bool live = FRONT ? (tid >= 2 && tid < 2 + CB_FRONT_OUT && l < a.L) : (l < a.L);   // lane owns a real output
  f32x2 h[C];                                  // 2C accumulators as pairs
#pragma unroll
  for (int j = 0; j < C; ++j) h[j] = f32x2{b5[2 * j], b5[2 * j + 1]};
#pragma unroll 1
  for (int ci = 0; ci < C; ++ci) {
    const float* trow = tile + ci * TWp + toff;
    float x[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) x[k] = trow[k];
    // at most ~64 weights in scalar registers at a time: more spills SGPRs into VGPR lanes (v_readlane per use)
    constexpr int TG = (C2 <= 16) ? 4 : (C2 <= 32 ? 2 : 1);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (k % TG == 0) __builtin_amdgcn_sched_barrier(0);
      const float* __restrict__ wk = w5 + (size_t)(ci * 5 + k) * C2;      // wave-uniform: scalar loads
      const f32x2 x2 = {x[k], x[k]};
#pragma unroll
      for (int j = 0; j < C; ++j) h[j] = __builtin_elementwise_fma(x2, f32x2{wk[2 * j], wk[2 * j + 1]}, h[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int j = 0; j < C; ++j) h[j] = f32x2{apply_act(h[j].x, ACT_SILU), apply_act(h[j].y, ACT_SILU)};
  // 1x1 conv 2C -> C.  Its 2C x C weights do not fit the scalar register file if the loop over the 2C inputs is unrolled
  // (and it must be, to index registers): the lane parks its 2C values in LDS (column tid: conflict-free) and walks them
  // in a rolled loop, one scalar weight row in flight at a time.
  f32x2 o[C / 2];
#pragma unroll
  for (int c = 0; c < C / 2; ++c) o[c] = f32x2{b1[2 * c], b1[2 * c + 1]};
#pragma unroll
  for (int j0 = 0; j0 < C2; j0 += HP) {
#pragma unroll
    for (int j = 0; j < HP / 2; ++j) {
      hs[(2 * j) * 256 + tid] = h[j0 / 2 + j].x;
      hs[(2 * j + 1) * 256 + tid] = h[j0 / 2 + j].y;
    }
#pragma unroll 2
    for (int j = 0; j < HP; ++j) {
      const float* __restrict__ wj = w1 + (size_t)(j0 + j) * C;
      const float hv = hs[j * 256 + tid];
      const f32x2 h2 = {hv, hv};
#pragma unroll
      for (int c = 0; c < C / 2; ++c) o[c] = __builtin_elementwise_fma(h2, f32x2{wj[2 * c], wj[2 * c + 1]}, o[c]);
    }
  }
  float v[C];
  {
    // the skip tensor's C values are requested together (one branch, not one -- with a full wait behind its load -- per channel)
    float sk[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sk[c] = 0.f;
    if (a.res2 && live) {
#pragma unroll
      for (int c = 0; c < C; ++c) sk[c] = a.res2[((size_t)b * C + c) * a.L + l];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = ((c & 1) ? o[c >> 1].y : o[c >> 1].x) + tile[c * TWp + toff + 2] + sk[c];
  }
  if (!TAIL) {
    if (live) {
#pragma unroll
      for (int c = 0; c < C; ++c) a.out[((size_t)b * C + c) * a.L + l] = v[c];
    }
    return;
  }
  // tail: two 1x1 convs on the lane's own position, then the maximum over the row's positions
  f32x2 t[C / 2];
#pragma unroll
  for (int c = 0; c < C / 2; ++c) t[c] = f32x2{ta_b[2 * c], ta_b[2 * c + 1]};
#pragma unroll
  for (int j = 0; j < C; ++j) {
    const f32x2 v2 = {v[j], v[j]};
#pragma unroll
    for (int c = 0; c < C / 2; ++c) t[c] = __builtin_elementwise_fma(v2, f32x2{ta_w[j * C + 2 * c], ta_w[j * C + 2 * c + 1]}, t[c]);
  }
  f32x2 u[C / 2];
#pragma unroll
  for (int c = 0; c < C / 2; ++c) u[c] = f32x2{tb_b[2 * c], tb_b[2 * c + 1]};
#pragma unroll
  for (int j = 0; j < C; ++j) {
    const float r = fmaxf((j & 1) ? t[j >> 1].y : t[j >> 1].x, 0.f);
    const f32x2 r2 = {r, r};
#pragma unroll
    for (int c = 0; c < C / 2; ++c) u[c] = __builtin_elementwise_fma(r2, f32x2{tb_w[j * C + 2 * c], tb_w[j * C + 2 * c + 1]}, u[c]);
  }
  __syncthreads();                               // hs is free again
  // Softplus is non-decreasing, so the maximum over positions of Softplus(u) is Softplus(maximum of u): the workgroup reduces
  // the raw values and C threads apply the activation once.  The reduction goes through LDS (a lane parks its C values, 64
  // threads take 32 positions of one channel each, C threads finish): a butterfly of wave shuffles is 6 x C permutes with their
  // address arithmetic per lane, and this kernel is bound by vector-instruction issue.
  static_assert(HP * 256 >= C * 256, "the parked values fit the SiLU buffer");
#pragma unroll
  for (int c = 0; c < C; ++c) hs[c * 256 + tid] = live ? ((c & 1) ? u[c >> 1].y : u[c >> 1].x) : -INFINITY;
  __syncthreads();
  float part = -INFINITY;
  if (tid < 8 * C) {                             // (channel, 32-position slice)
    const int c = tid >> 3, sl = tid & 7;
    const float* row = hs + c * 256 + 32 * sl;
#pragma unroll
    for (int i = 0; i < 32; ++i) part = fmaxf(part, row[(i + 4 * sl + (c >> 1)) & 31]);      // rotated starts: the 64 threads spread over all banks
  }
  __syncthreads();
  if (tid < 8 * C) hs[tid] = part;
  __syncthreads();
  if (tid < C) {
    float m = hs[8 * tid];
#pragma unroll
    for (int i = 1; i < 8; ++i) m = fmaxf(m, hs[8 * tid + i]);
    a.tail_max[((size_t)b * gridDim.x + blockIdx.x) * C + tid] = apply_act(m, ACT_SOFTPLUS);
  }
}

bool convblock_supported(int C) { return C == 8 || C == 16 || C == 24; }   // LDS: tile + 2C x 256 floats

// upper bound of the workgroups per row (sizing of tail_max); the exact count of a launch: convblock_tiles_of
int convblock_tiles(int L, bool front) { return front ? (L + CB_FRONT_OUT_POLY - 1) / CB_FRONT_OUT_POLY : (L + 255) / 256; }

static bool convblock8_valu_form() {
  static const bool valu8_env = dev_env("MURAL_CONVBLOCK8_VALU") != nullptr && dev_env("MURAL_CONVBLOCK8_VALU")[0] == '1';
  return g_convblock8_form >= 0 ? g_convblock8_form == 0 : valu8_env;
}
static bool convblock_poly_mfma(const ConvBlockArgs& a) {
  return a.C == 8 && !convblock8_valu_form() && a.f_in != nullptr && a.symtab == nullptr && a.f_pw != nullptr && a.Cf == 16 && a.f_up == 4;
}
// workgroups per row of THIS launch = entries per row of its tail_max
int convblock_tiles_of(const ConvBlockArgs& a) {
  const bool front = a.f_in != nullptr || a.symtab != nullptr;
  if (!front) return (a.L + 255) / 256;
  if (!convblock_mfma_supported(a) && convblock_poly_mfma(a)) return (a.L + CB_FRONT_OUT_POLY - 1) / CB_FRONT_OUT_POLY;
  return (a.L + CB_FRONT_OUT - 1) / CB_FRONT_OUT;
}

unsigned long long* g_cb8_stamps = nullptr;      // diagnostic (debug flavour: mural_debug_cb8_set_stamps): per-workgroup phase sums of the level-0 blocks      // validation hook (mural_debug_convblock): 0 vector-ALU form, 1 split form, -1 the environment's choice

template <bool TAIL, bool FRONT>
static void launch_convblock_t(const ConvBlockArgs& a, hipStream_t stream) {
  const dim3 grid(convblock_tiles_of(a), a.B);
  // diagnostic (tools/phase_stamps_indel_l0.py): MURAL_DEBUG_CB_STAMP_ONLY = enc | dec stamps only the genome-fed / the tail launch
  unsigned long long* stamps = g_cb8_stamps;
  if (stamps)
    if (const char* only = dev_env("MURAL_DEBUG_CB_STAMP_ONLY"))
      if ((only[0] == 'e') != (a.symtab != nullptr) || (only[0] == 'd') != (a.tail_max != nullptr)) stamps = nullptr;
#define MURAL_CB(CN)                                                                                                        \
  hipLaunchKernelGGL((convblock_kernel<CN, TAIL, FRONT, MFV>), grid, dim3(256), MFV ? mf_lds : 0, stream, a, a.w5, a.b5, a.w1, a.b1, a.ta_w, a.ta_b, \
                     a.tb_w, a.tb_b, a.f_w, a.f_b, stamps)
  // MURAL_CONVBLOCK8_VALU=1: the 8-channel block entirely on the vector ALU (A/B switch for the split form)
  const size_t mf_lds = (size_t)(8 * 272 + convblock_front_floats(a.Cf, a.f_up, FRONT) + ((a.f_pw && !convblock_poly_mfma(a)) ? 4 * 16 * 3 * 8 : 0)) * sizeof(float);
  const bool valu8 = convblock8_valu_form();
  switch (a.C) {
    case 8: {
      if (valu8) {
        constexpr bool MFV = false;
        MURAL_CB(8);
      } else {
        constexpr bool MFV = true;
        MURAL_CB(8);
      }
      break;
    }
    case 16: { constexpr bool MFV = false; MURAL_CB(16); break; }
    default: { constexpr bool MFV = false; MURAL_CB(24); break; }
  }
#undef MURAL_CB
}

int launch_convblock(const ConvBlockArgs& a, hipStream_t stream) {
  if (a.B == 0 || a.L == 0) return MURAL_OK;
  if (convblock_deep_supported(a)) return launch_convblock_deep(a, stream);      // 32 channels, rows of up to 80 columns
  MURAL_REQUIRE(convblock_supported(a.C), "convblock: %d channels not instantiated", a.C);
  if (a.symtab) MURAL_REQUIRE(a.Cf == 4 && a.f_up == 1 && a.C == 8 && a.sym_taps >= 1 && (a.sym_taps & 1) && a.sym_taps <= 15 && a.sym_bias &&
                              ((a.g_pos && a.g_strand) || a.sym_in), "convblock: the genome-fed front serves the 4-channel input of the first level");
  if (a.sym_in) MURAL_REQUIRE(a.symtab && a.f_in && indel_enc0_supported(a), "convblock: a symbol-byte source is served by the persistent first-level kernel only");
  if (a.f_in || a.symtab) {
    MURAL_REQUIRE(a.f_w && a.f_b && a.f_up >= 1 && a.Lf * a.f_up == a.L, "convblock: bad front geometry");
    MURAL_REQUIRE(a.Cf * (262 / a.f_up + 3) <= CB_FRONT_FLOATS, "convblock: front input tile does not fit");
    MURAL_REQUIRE(!a.f_pw || a.f_up == 4, "convblock: the polyphase front serves an upsampling factor of 4");
  }
  if (a.tail_max) MURAL_REQUIRE(a.ta_w && a.ta_b && a.tb_w && a.tb_b, "convblock: tail weights missing");
  if (convblock_mfma_supported(a)) return launch_convblock_mfma(a, stream);
  if (indel_enc0_supported(a)) return launch_indel_enc0(a, stream);
  if (convblock_poly_mfma(a) && indel_dec0_supported(a)) return launch_indel_dec0(a, stream);
  const bool front = a.f_in != nullptr || a.symtab != nullptr;
  if (a.tail_max && front) launch_convblock_t<true, true>(a, stream);
  else if (a.tail_max) launch_convblock_t<true, false>(a, stream);
  else if (front) launch_convblock_t<false, true>(a, stream);
  else launch_convblock_t<false, false>(a, stream);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

// one wave per row: max over the row
__global__ __launch_bounds__(256) void rowmax_kernel(const float* __restrict__ x, int64_t rows, int L, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = x + row * L;
  float m = -INFINITY;
  for (int i = lane; i < L; i += 64) m = fmaxf(m, p[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (lane == 0) y[row] = m;
}

int launch_rowmax(const float* x, int64_t rows, int L, float* y, hipStream_t stream) {
  if (rows == 0) return MURAL_OK;
  hipLaunchKernelGGL(rowmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, rows, L, y);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural
