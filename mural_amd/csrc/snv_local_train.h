// The local branch of Network0 / Network2 inside the composed training step, three launches per direction (+ one for the three weight
// gradients) instead of 9 + 16 (included by train_ops.hip behind its helpers; gfx950 / CDNA4).
//
// Reference: MuRaL/model/model_snv.py:322-339, 451-468 under model.train() (training.py:424-427) -- the shared Embedding(65, 5) of every
// k-mer column -> Dropout -> [Linear -> ReLU -> BatchNorm1d -> Dropout] x 2 -> Linear.  The batch-statistics BatchNorm is the only
// operation that needs the whole batch, so the forward cuts there:
//     F1: gather + dropout + Linear1, batch sums of relu(y)          F2: BatchNorm1 (finalised by every workgroup from the sums)
//     + dropout + Linear2, batch sums                                 F3: BatchNorm2 + dropout + Linear3
// and the backward, whose BatchNorm needs the batch sums of (d, d * xhat), at the mirrored places:
//     B3: d logits x W3 -> dropout mask, sums of BatchNorm2          B2: BatchNorm2-backward apply + ReLU mask while the operand is
//     loaded, x W2 -> dropout mask, sums of BatchNorm1                 B1: the same for layer 1, x W1 -> dropout mask -> embedding scatter
//     W: the three weight / bias gradients in one launch.
// What was there before -- one launch per op, the Linear on 64 workgroups that each copied the whole weight matrix into LDS -- took
// 19 % of the step's kernel time for 0.6 GFLOP (profiles/r04_train_step_rocprof_summary.txt: linear_mfma_kernel at mfma_busy 0.01).
//
// Kernel shape: a workgroup owns 16 batch rows (one MFMA M-tile), its four waves share the 16-column output blocks.  Both MFMA operands
// come straight from memory in a PERMUTED reduction order: lane (m, q) of v_mfma_f32_16x16x4_f32 takes k = q * KQ + s in step s, so its
// KQ values are one contiguous run of a row of x (A) or of W (B) -- no LDS image, no barrier in the main loop; the transform in front
// of the Linear (gather / ReLU + BatchNorm affine + dropout, or the BatchNorm-backward apply) runs on the lane's run while it sits in
// registers.  The weights (57 KB at most) stay in L2.
#pragma once

namespace mural {
namespace ltrain {

constexpr uint64_t LT_GOLD = 0x9E3779B97F4A7C15ull;
constexpr uint32_t LT_OOB = 0x80000000u;      // offset no descriptor covers
typedef float lt_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lt_keep(uint64_t seed, uint64_t index, float p, float keep_scale, float v) {      // dropout_kernel's draw
  const uint64_t r = mix64(seed + LT_GOLD * (index + 1));
  const float u = (float)(r >> 40) * (1.f / 16777216.f);
  return u >= p ? v * keep_scale : 0.f;
}

struct LtFwd {
  const int64_t* cat;          // layer 1: [B][cols] k-mer ids; x[b][k] = E[cat[b][k / 5]][k % 5]
  const float* E;
  int cols, emb_rows;
  const float* x;              // layers 2, 3: [B][K] raw output of the Linear below
  const double* acc;           // batch sums of relu(x), relu(x)^2 ([MURAL_BN_SLOTS][2][K]) or nullptr: no BatchNorm in front (layer 1)
  const float* gamma;
  const float* beta;
  float eps, momentum;
  float* running_mean;
  float* running_var;
  float* state;                // [4][K] scale | shift | mean | invstd, written by workgroup 0
  float p;                     // dropout behind the BatchNorm (behind the gather for layer 1)
  uint64_t seed;
  const uint64_t* seed_dev;
  float* xt;                   // [B][K] the Linear's input as it was used (saved for the weight gradient)
  const float* W;              // [N][K]
  const float* bias;
  float* y;                    // [B][N]
  double* acc_out;             // batch sums of relu(y), relu(y)^2 for the next BatchNorm, or nullptr
  int64_t B;
  int K, N;
};

// a lane's run of KQ consecutive floats base[o .. o + KQ) of a tensor of `tot` floats (4-byte aligned): 16-byte loads, then dwords; a
// run that would cross the tensor's end (the batch's last row, the weight's last row) is read element by element, zeros behind the end
typedef f32x4_t lt_f32x4_u __attribute__((aligned(4)));
template <int KQ>
__device__ __forceinline__ void lt_run(const float* __restrict__ base, int64_t o, int64_t tot, float (&v)[KQ]) {
  if (o + KQ <= tot) {
    const float* p = base + o;
#pragma unroll
    for (int s = 0; s + 4 <= KQ; s += 4) {
      const f32x4_t t = *reinterpret_cast<const lt_f32x4_u*>(p + s);
      v[s] = t[0]; v[s + 1] = t[1]; v[s + 2] = t[2]; v[s + 3] = t[3];
    }
#pragma unroll
    for (int s = KQ & ~3; s < KQ; ++s) v[s] = p[s];
  } else {
    // (branch-free: `o + s < tot ? base[o + s] : 0` is a load behind a branch with a wait of its own per element -- KQ serial round trips
    // in every wave that holds a lane of the tensor's last row, which set the duration of the whole launch: 16 -> 6 us)
    // one descriptor over the whole tensor (wave-uniform; the launchers keep tensors under 2 GB), the run in the lane offset
    const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(tot * 4), 0x00020000);
#pragma unroll
    for (int s = 0; s < KQ; ++s) v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d, 4u * (uint32_t)(o + s), 0, 0));
  }
}

template <int KQ, bool EMB>
__global__ __launch_bounds__(256) void lt_fwd_kernel(const LtFwd a) {
  extern __shared__ __attribute__((aligned(16))) float cst[];      // [K][2] scale, shift
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, q = lane >> 4;
  const int K = a.K, N = a.N;
  const bool bn = !EMB && a.acc != nullptr;
  if (bn) {
    const double n = (double)a.B;
    for (int c = tid; c < K; c += 256) {
      double v1[MURAL_BN_SLOTS], v2[MURAL_BN_SLOTS];
#pragma unroll
      for (int k = 0; k < MURAL_BN_SLOTS; ++k) {
        v1[k] = a.acc[((size_t)k * 2 + 0) * K + c];
        v2[k] = a.acc[((size_t)k * 2 + 1) * K + c];
      }
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int k = 0; k < MURAL_BN_SLOTS; ++k) {
        s1 += v1[k];
        s2 += v2[k];
      }
      const double mean = s1 / n;
      double var = s2 / n - mean * mean;
      if (var < 0.0) var = 0.0;
      const double invstd = 1.0 / sqrt(var + (double)a.eps);
      const float sc = (float)(a.gamma[c] * invstd), sh = (float)(a.beta[c] - mean * a.gamma[c] * invstd);
      cst[2 * c] = sc;
      cst[2 * c + 1] = sh;
      if (blockIdx.x == 0) {
        a.state[c] = sc;
        a.state[K + c] = sh;
        a.state[2 * K + c] = (float)mean;
        a.state[3 * K + c] = (float)invstd;
        if (a.running_mean) {
          const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
          a.running_mean[c] = (float)((1.0 - a.momentum) * a.running_mean[c] + a.momentum * mean);
          a.running_var[c] = (float)((1.0 - a.momentum) * a.running_var[c] + a.momentum * unbiased);
        }
      }
    }
    __syncthreads();
  }
  uint64_t seed = a.seed;
  if (a.seed_dev) seed += *a.seed_dev;
  const float keep_scale = 1.f / (1.f - a.p);
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  // ---- A: this lane's run of row (row0 + m), transformed
  float af[KQ];
  {
    const int64_t row = row0 + m;
    const bool rv = row < a.B;
    const int64_t rr = rv ? row : a.B - 1;
    const int k0 = q * KQ;
    if (EMB) {
#pragma unroll
      for (int s = 0; s < KQ; ++s) {
        const int k = k0 + s;
        const int kc = k < K ? k : K - 1;
        const int col = kc / 5, d = kc - 5 * col;
        int64_t id = a.cat[rr * a.cols + col];
        id = id < 0 ? 0 : (id >= a.emb_rows ? a.emb_rows - 1 : id);
        af[s] = a.E[id * 5 + d];
      }
    } else {
      // (the run of the last quarter may reach into the next row: masked below)
      lt_run<KQ>(a.x, rr * K + k0, a.B * K, af);
      if (bn) {
#pragma unroll
        for (int s = 0; s < KQ; ++s) {
          const int k = k0 + s < K ? k0 + s : K - 1;
          const lt_f32x2 c2 = *reinterpret_cast<const lt_f32x2*>(cst + 2 * k);
          af[s] = fmaf(c2.x, fmaxf(af[s], 0.f), c2.y);
        }
      }
    }
    // (wave-uniform conditions once around the loops, stores through a range-checked descriptor with the offset chosen by a select:
    // a branch per element -- `if (p > 0)`, `if (wave == 0 && valid)` -- keeps the scheduler from overlapping the elements' latencies)
    if (a.p > 0.f) {
#pragma unroll
      for (int s = 0; s < KQ; ++s) af[s] = lt_keep(seed, (uint64_t)(rr * K + k0 + s), a.p, keep_scale, af[s]);
    }
#pragma unroll
    for (int s = 0; s < KQ; ++s) af[s] = (rv && k0 + s < K) ? af[s] : 0.f;
    if (wave == 0) {
      const __amdgpu_buffer_rsrc_t xo = __builtin_amdgcn_make_buffer_rsrc(a.xt, 0, (int)(a.B * K * 4), 0x00020000);
#pragma unroll
      for (int s = 0; s < KQ; ++s)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, af[s]), xo, (rv && k0 + s < K) ? 4u * (uint32_t)(row * K + k0 + s) : LT_OOB, 0, 0);
    }
  }
  // ---- output blocks of this wave
  const int nblk = (N + 15) >> 4;
  for (int nb = wave; nb < nblk; nb += 4) {
    const int n = 16 * nb + m;
    const bool nv = n < N;
    const int nn = nv ? n : N - 1;
    float bf[KQ];
    // (no masks on B: the A operand is zero for k >= K, and an output column n >= N is never stored)
    lt_run<KQ>(a.W, (int64_t)nn * K + q * KQ, (int64_t)N * K, bf);
    const float bv0 = (a.bias ? a.bias : a.W)[nn];      // (unconditional load, select behind it)
    const float bv = (nv && a.bias) ? bv0 : 0.f;
    f32x4_t acc = {bv, bv, bv, bv};
#pragma unroll
    for (int s = 0; s < KQ; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc, 0, 0, 0);
    // D[row 4 q + r][column m]
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + 4 * q + r;
      if (row < a.B && nv) {
        a.y[row * N + n] = acc[r];
        const float w = fmaxf(acc[r], 0.f);
        s1 += w;
        s2 += w * w;
      }
    }
    if (a.acc_out) {
      s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
      if (q == 0 && nv) {
        double* slot = a.acc_out + (size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 * N;
        atomicAdd(&slot[n], (double)s1);
        atomicAdd(&slot[N + n], (double)s2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------- backward
struct LtBwd {
  // gradient arriving at this layer's Linear output: `du` as it is (top layer: d logits), or -- BatchNorm-backward apply + ReLU mask of the
  // BatchNorm BEHIND this Linear, made while the operand is loaded -- from du = the dropout-masked gradient of the BatchNorm output
  const float* du;             // [B][O]
  const float* lin;            // [B][O] this Linear's raw output (ReLU mask, xhat), or nullptr: du is used as it is
  const float* state;          // [4][O] of that BatchNorm
  const float* gamma;
  const double* acc;           // its completed sums of (du, du * xhat)
  float* dgamma;               // written by workgroup 0
  float* dbeta;
  float* g;                    // [B][O] the gradient of the Linear's output as used (saved for the weight gradient; nullptr with lin == nullptr)
  const float* W;              // [O][I]
  // what happens to d x = g W ([B][I]): dropout mask of the layer below, then
  float p;
  uint64_t seed;
  const uint64_t* seed_dev;
  float* dd;                   // [B][I] stored, and the sums of (dd, dd * xhat) of the BatchNorm below taken -- or
  const float* lin_below;      // [B][I]
  const float* state_below;    // [4][I]
  double* acc_below;
  const int64_t* cat;          // bottom layer: scattered into the embedding gradient dE[emb_rows][5] (zeroed by the caller)
  float* dE;
  int cols, emb_rows;
  int64_t B;
  int I, O;
  unsigned long long* stamps;  // diagnostic (mural_debug_lt_set_stamps): [launch slot][workgroup][8] wall-clock ticks of wave 0
};
#define LT_STAMP(K) do { if (a.stamps && tid == 0) a.stamps[8 * blockIdx.x + (K)] = __builtin_amdgcn_s_memrealtime(); } while (0)

template <int KQ, bool BOTTOM, bool APPLY>
__global__ __launch_bounds__(256) void lt_bwd_kernel(const LtBwd a) {
  extern __shared__ __attribute__((aligned(16))) float cst[];      // [O][8]: gamma * invstd, mean(du), mean(du * xhat), mean, invstd, - - - ; bottom: | dE image [emb_rows][5]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, q = lane >> 4;
  const int I = a.I, O = a.O;
  constexpr bool apply = APPLY;
  float* se = cst + (apply ? 8 * O : 0);
  LT_STAMP(0);
  if (apply) {
    const double n = (double)a.B;
    for (int c = tid; c < O; c += 256) {
      double v1[MURAL_BN_SLOTS], v2[MURAL_BN_SLOTS];
#pragma unroll
      for (int k = 0; k < MURAL_BN_SLOTS; ++k) {
        v1[k] = a.acc[((size_t)k * 2 + 0) * O + c];
        v2[k] = a.acc[((size_t)k * 2 + 1) * O + c];
      }
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int k = 0; k < MURAL_BN_SLOTS; ++k) {
        s1 += v1[k];
        s2 += v2[k];
      }
      cst[8 * c] = a.gamma[c] * a.state[3 * O + c];
      cst[8 * c + 1] = (float)(s1 / n);
      cst[8 * c + 2] = (float)(s2 / n);
      cst[8 * c + 3] = a.state[2 * O + c];
      cst[8 * c + 4] = a.state[3 * O + c];
      if (blockIdx.x == 0) {
        a.dgamma[c] = (float)s2;
        a.dbeta[c] = (float)s1;
      }
    }
  }
  if (BOTTOM)
    for (int i = tid; i < a.emb_rows * 5; i += 256) se[i] = 0.f;
  if (apply || BOTTOM) __syncthreads();
  LT_STAMP(1);
  uint64_t seed = a.seed;
  if (a.seed_dev) seed += *a.seed_dev;
  const float keep_scale = 1.f / (1.f - a.p);
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  // ---- A: the lane's run of the gradient row (row0 + m) over the reduction index o
  float af[KQ];
  {
    const int64_t row = row0 + m;
    const bool rv = row < a.B;
    const int64_t rr = rv ? row : a.B - 1;
    const int k0 = q * KQ;
    lt_run<KQ>(a.du, rr * O + k0, a.B * O, af);
    if constexpr (apply) {      // bn_bwd_apply_kernel's arithmetic: g = relu'(lin) * gamma * invstd * (du - mean(du) - xhat * mean(du * xhat))
      float lraw[KQ];
      lt_run<KQ>(a.lin, rr * O + k0, a.B * O, lraw);
#pragma unroll
      for (int s = 0; s < KQ; ++s) {
        const int kc = k0 + s < O ? k0 + s : O - 1;
        const f32x4_t c4 = *reinterpret_cast<const f32x4_t*>(cst + 8 * kc);
        const float is = cst[8 * kc + 4];
        const float xh = (fmaxf(lraw[s], 0.f) - c4[3]) * is;
        const float gq = c4[0] * (af[s] - c4[1] - xh * c4[2]);
        af[s] = lraw[s] > 0.f ? gq : 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < KQ; ++s) af[s] = (rv && k0 + s < O) ? af[s] : 0.f;
    if (apply && wave == 0) {      // (stores through a range-checked descriptor, offsets by select: no branch per element)
      const __amdgpu_buffer_rsrc_t go = __builtin_amdgcn_make_buffer_rsrc(a.g, 0, (int)(a.B * O * 4), 0x00020000);
#pragma unroll
      for (int s = 0; s < KQ; ++s)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, af[s]), go, (rv && k0 + s < O) ? 4u * (uint32_t)(row * O + k0 + s) : LT_OOB, 0, 0);
    }
  }
  LT_STAMP(2);
  // ---- blocks of 16 input features; B operand: W[o = q KQ + s][i = 16 nb + m]
  const int64_t rows_here = a.B - row0 < 16 ? a.B - row0 : 16;
  const __amdgpu_buffer_rsrc_t wd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W), 0, O * I * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ld = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BOTTOM ? a.W : a.lin_below + row0 * I), 0,
                                                                       BOTTOM ? 0 : (int)(rows_here * I * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t sd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BOTTOM ? a.W : a.state_below), 0, BOTTOM ? 0 : 4 * I * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t cd = __builtin_amdgcn_make_buffer_rsrc(const_cast<int64_t*>(BOTTOM ? a.cat + row0 * a.cols : nullptr), 0,
                                                                       BOTTOM ? (int)(rows_here * a.cols * 8) : 0, 0x00020000);
  const int nblk = (I + 15) >> 4;
  for (int nb = wave; nb < nblk; nb += 4) {
    const int i = 16 * nb + m;
    const bool iv = i < I;
    // Every load of a block goes through a range-checked descriptor with its offset chosen by a select (a refused element aims past
    // the descriptor and reads 0): `valid ? W[..] : 0` is compiled to a load behind a branch with a full wait of its own -- 38 serial
    // round trips per block ("a global load behind a branch costs a full wait").
    float bf[KQ];
#pragma unroll
    for (int s = 0; s < KQ; ++s) {
      const int o = q * KQ + s;
      // (o >= O lies behind the descriptor by itself and reads 0; a column i >= I reads a neighbour's weight into an output column
      // that is never stored -- no per-step masks: 38 of them, block-invariant, were hoisted into 76 scalar registers and spilled)
      bf[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wd, 4u * (uint32_t)(o * I + i), 0, 0));
    }
    // what the epilogue needs from memory, requested in front of the MFMAs: the BatchNorm below's input / the k-mer ids of the rows
    float lb[4];
    uint32_t idv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + 4 * q + r;
      const bool ok = row < a.B && iv;
      if (BOTTOM) idv[r] = __builtin_amdgcn_raw_buffer_load_b32(cd, ok ? 8u * (uint32_t)((row - row0) * a.cols + i / 5) : LT_OOB, 0, 0);
      else lb[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ld, ok ? 4u * (uint32_t)((row - row0) * I + i) : LT_OOB, 0, 0));
    }
    const uint32_t so = (!BOTTOM && iv) ? 4u * (uint32_t)(2 * I + i) : LT_OOB;
    const float mu = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sd, so, 0, 0));
    const float is = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sd, so == LT_OOB ? LT_OOB : so + 4u * (uint32_t)I, 0, 0));
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KQ; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc, 0, 0, 0);
    if (nb == wave) LT_STAMP(3);
    // D[row 4 q + r][feature m]: dropout mask of the layer below, then sums + store, or the embedding scatter
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + 4 * q + r;
      if (row < a.B && iv) {
        float v = acc[r];
        if (a.p > 0.f) v = lt_keep(seed, (uint64_t)(row * I + i), a.p, keep_scale, v);
        if (BOTTOM) {
          const int d = i - 5 * (i / 5);
          int id = (int)idv[r];                  // low word of the int64 id (ids are < 2^31; negative ids read as large: clamped to the last row below like the
          id = id < 0 ? 0 : (id >= a.emb_rows ? a.emb_rows - 1 : id);      // forward's gather does after its own 64-bit clamp)
          atomicAdd(&se[id * 5 + d], v);
        } else {
          a.dd[row * I + i] = v;
          const float xh = (fmaxf(lb[r], 0.f) - mu) * is;
          s1 += v;
          s2 += v * xh;
        }
      }
    }
    if (!BOTTOM) {
      s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
      if (q == 0 && iv) {
        double* slot = a.acc_below + (size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 * I;
        atomicAdd(&slot[i], (double)s1);
        atomicAdd(&slot[I + i], (double)s2);
      }
    }
  }
  LT_STAMP(4);
  if (BOTTOM) {
    __syncthreads();
    LT_STAMP(5);
    for (int i = tid; i < a.emb_rows * 5; i += 256)
      if (se[i] != 0.f) atomicAdd(&a.dE[i], se[i]);
  }
  LT_STAMP(6);
}
#undef LT_STAMP

// ---- the three weight / bias gradients in one launch: linear_wgrad_mfma_kernel's 16 x 16 blocks of dW, job by block range
struct LtWgradJobs {
  const float* dy[3];
  const float* x[3];
  float* dW[3];
  float* db[3];
  int I[3], O[3], first[4];
  int64_t B;
};
__global__ __launch_bounds__(1024) void lt_wgrad_kernel(const LtWgradJobs jobs) {
  __shared__ float red[16][16 * 16 + 16];
  const int j = (int)blockIdx.x >= jobs.first[2] ? 2 : ((int)blockIdx.x >= jobs.first[1] ? 1 : 0);
  const float* __restrict__ dy = jobs.dy[j];
  const float* __restrict__ x = jobs.x[j];
  const int I = jobs.I[j], O = jobs.O[j];
  const int64_t B = jobs.B;
  const int blk = blockIdx.x - jobs.first[j];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4;
  const int nbi = (I + 15) / 16;
  const int ob = blk / nbi, ib = blk - ob * nbi;
  const int o = 16 * ob + n16, i = 16 * ib + n16;
  const bool ov = o < O, iv = i < I;
  const int64_t per = ((B + 15) / 16 + 3) & ~(int64_t)3;            // rows per wave, a multiple of 4
  const int64_t r0 = wave * per, r1 = (r0 + per < B) ? r0 + per : B;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int64_t r = r0; r < r1; r += 32) {
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {                                      // 16 loads in flight per lane
      const int64_t row = r + 4 * u + kk;
      const bool rv = row < r1;
      av[u] = (rv && ov) ? dy[row * O + o] : 0.f;
      bv[u] = (rv && iv) ? x[row * I + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
      bsum += av[u];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(4 * kk + r) * 16 + n16] = acc[r];
  bsum += __shfl_xor(bsum, 16);
  bsum += __shfl_xor(bsum, 32);
  if (lane < 16) red[wave][256 + lane] = bsum;
  __syncthreads();
  if (tid < 256 + 16) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][tid];
    if (tid < 256) {
      const int oo = 16 * ob + (tid >> 4), ii = 16 * ib + (tid & 15);
      if (oo < O && ii < I) jobs.dW[j][(size_t)oo * I + ii] = t;
    } else if (ib == 0 && jobs.db[j]) {
      const int oo = 16 * ob + (tid - 256);
      if (oo < O) jobs.db[j][oo] = t;
    }
  }
}

template <bool EMB>
int lt_launch_fwd(const LtFwd& a, hipStream_t stream) {
  const int kq = (a.K + 3) / 4;
  const unsigned grid = (unsigned)((a.B + 15) / 16);
  const size_t lds = (size_t)2 * a.K * 4;
#define LT_F(Q) if (kq <= Q) { hipLaunchKernelGGL((lt_fwd_kernel<Q, EMB>), dim3(grid), dim3(256), lds, stream, a); MURAL_HIP_CHECK(hipGetLastError()); return MURAL_OK; }
  LT_F(19) LT_F(24) LT_F(38) LT_F(64)
#undef LT_F
  set_error("local branch (fused): a Linear input of %d features is beyond the kernels built (256)", a.K);
  return MURAL_E_INVALID;
}

unsigned long long* g_lt_stamps = nullptr;      // diagnostic (mural_debug_lt_set_stamps): three launches x 256 workgroups x 8
template <bool BOTTOM>
int lt_launch_bwd(LtBwd a, hipStream_t stream) {
  a.stamps = g_lt_stamps ? g_lt_stamps + (size_t)(BOTTOM ? 2 : (a.lin ? 1 : 0)) * 256 * 8 : nullptr;
  const int kq = (a.O + 3) / 4;
  const unsigned grid = (unsigned)((a.B + 15) / 16);
  const size_t lds = (size_t)(a.lin ? 8 * a.O : 0) * 4 + (BOTTOM ? (size_t)a.emb_rows * 5 * 4 : 0);
#define LT_B(Q)                                                                                                                \
  if (kq <= Q) {                                                                                                               \
    if (a.lin) hipLaunchKernelGGL((lt_bwd_kernel<Q, BOTTOM, true>), dim3(grid), dim3(256), lds, stream, a);                    \
    else hipLaunchKernelGGL((lt_bwd_kernel<Q, BOTTOM, false>), dim3(grid), dim3(256), lds, stream, a);                         \
    MURAL_HIP_CHECK(hipGetLastError());                                                                                        \
    return MURAL_OK;                                                                                                           \
  }
  LT_B(2) LT_B(19) LT_B(38) LT_B(64)
#undef LT_B
  set_error("local branch (fused): a Linear output of %d features is beyond the kernels built (256)", a.O);
  return MURAL_E_INVALID;
}

}  // namespace ltrain

// does the fused form serve this shape?  (every Linear dimension within the kernels' reduction runs; MURAL_TRAIN_LOCAL_OPS=1 keeps the
// per-op launches for A/B runs and parity tests of both)
// The bottom backward launch keeps the embedding-gradient table (emb_rows x 5 floats) and eight rows of the Linear's output in LDS:
// tables beyond the 64 KB a launch gets without opting in (local_order >= 6: 4097 rows = 82 KB) and batches beyond the launchers'
// 32-bit offsets take the per-op route, which has neither limit.
bool local_train_fused_ok(int in1, int h1, int h2, int nc, int emb_rows, int64_t B) {
  const char* e = dev_env("MURAL_TRAIN_LOCAL_OPS");
  if (e && atoi(e) != 0) return false;
  const size_t bottom_lds = ((size_t)8 * (size_t)h1 + (size_t)emb_rows * 5) * 4;
  if (bottom_lds > (size_t)64 * 1024 || B > (1 << 20)) return false;
  return in1 <= 256 && h1 <= 256 && h2 <= 256 && nc <= 256 && in1 >= 1 && h1 >= 1 && h2 >= 1 && nc >= 1;
}

int local_train_fwd(const int64_t* cat, const float* E, int cols, int emb_rows, int64_t B, const int* dims /* in1, h1, h2, nc */,
                    const float* const* W, const float* const* bias, const float* const* gamma, const float* const* beta,
                    float* const* running_mean, float* const* running_var, float* const* state, double* const* acc_f, const float* drop,
                    const uint64_t* seeds, const uint64_t* seed_dev, float eps, float momentum, float* const* xt, float* const* lin,
                    float* logits, hipStream_t stream) {
  using namespace ltrain;
  if (B == 0) return MURAL_OK;
  MURAL_REQUIRE(B <= (1 << 20), "local branch (fused): batches of up to 2^20 rows (32-bit offsets into its tensors)");
  LtFwd a{};
  a.B = B; a.eps = eps; a.momentum = momentum; a.seed_dev = seed_dev;
  // F1: gather + dropout + Linear1
  a.cat = cat; a.E = E; a.cols = cols; a.emb_rows = emb_rows; a.p = drop[0]; a.seed = seeds[0]; a.xt = xt[0];
  a.W = W[0]; a.bias = bias[0]; a.y = lin[0]; a.acc_out = acc_f[0]; a.K = dims[0]; a.N = dims[1];
  if (int rc = lt_launch_fwd<true>(a, stream)) return rc;
  // F2 / F3: BatchNorm (batch sums of the launch before) + dropout + Linear
  for (int l = 1; l < 3; ++l) {
    a.cat = nullptr; a.E = nullptr; a.x = lin[l - 1]; a.acc = acc_f[l - 1]; a.gamma = gamma[l - 1]; a.beta = beta[l - 1];
    a.running_mean = running_mean[l - 1]; a.running_var = running_var[l - 1]; a.state = state[l - 1];
    a.p = drop[l]; a.seed = seeds[l]; a.xt = xt[l];
    a.W = W[l]; a.bias = bias[l]; a.y = l == 2 ? logits : lin[l]; a.acc_out = l == 2 ? nullptr : acc_f[l]; a.K = dims[l]; a.N = dims[l + 1];
    if (int rc = lt_launch_fwd<false>(a, stream)) return rc;
  }
  return MURAL_OK;
}

int local_train_bwd(const int64_t* cat, int cols, int emb_rows, int64_t B, const int* dims, const float* dlogits, const float* const* W,
                    const float* const* gamma, const float* const* state, double* const* acc_b, const float* drop, const uint64_t* seeds,
                    const uint64_t* seed_dev, const float* const* xt, const float* const* lin, float* const* dd /* [2] */, float* const* g /* [2] */,
                    float* const* dW, float* const* db, float* const* dgamma, float* const* dbeta, float* dE, hipStream_t stream) {
  using namespace ltrain;
  if (B == 0) return MURAL_OK;
  MURAL_HIP_CHECK(hipMemsetAsync(dE, 0, (size_t)emb_rows * 5 * 4, stream));
  LtBwd a{};
  a.B = B; a.seed_dev = seed_dev;
  // B3: d logits x W3 -> dropout mask of layer 2's output, sums of BatchNorm2
  a.du = dlogits; a.lin = nullptr; a.W = W[2]; a.I = dims[2]; a.O = dims[3];
  a.p = drop[2]; a.seed = seeds[2]; a.dd = dd[1]; a.lin_below = lin[1]; a.state_below = state[1]; a.acc_below = acc_b[1];
  if (int rc = lt_launch_bwd<false>(a, stream)) return rc;
  // B2: BatchNorm2-backward apply on the way in, x W2 -> dropout mask, sums of BatchNorm1
  a.du = dd[1]; a.lin = lin[1]; a.state = state[1]; a.gamma = gamma[1]; a.acc = acc_b[1]; a.dgamma = dgamma[1]; a.dbeta = dbeta[1]; a.g = g[1];
  a.W = W[1]; a.I = dims[1]; a.O = dims[2];
  a.p = drop[1]; a.seed = seeds[1]; a.dd = dd[0]; a.lin_below = lin[0]; a.state_below = state[0]; a.acc_below = acc_b[0];
  if (int rc = lt_launch_bwd<false>(a, stream)) return rc;
  // B1: BatchNorm1-backward apply, x W1 -> dropout mask of the embedding -> scatter
  a.du = dd[0]; a.lin = lin[0]; a.state = state[0]; a.gamma = gamma[0]; a.acc = acc_b[0]; a.dgamma = dgamma[0]; a.dbeta = dbeta[0]; a.g = g[0];
  a.W = W[0]; a.I = dims[0]; a.O = dims[1];
  a.p = drop[0]; a.seed = seeds[0]; a.dd = nullptr; a.lin_below = nullptr; a.state_below = nullptr; a.acc_below = nullptr;
  a.cat = cat; a.dE = dE; a.cols = cols; a.emb_rows = emb_rows;
  if (int rc = lt_launch_bwd<true>(a, stream)) return rc;
  // W: dW_l = g_l^T xt_l, db_l = sum g_l
  LtWgradJobs jobs{};
  jobs.B = B;
  const float* gl[3] = {g[0], g[1], dlogits};
  int first = 0;
  for (int l = 0; l < 3; ++l) {
    jobs.dy[l] = gl[l]; jobs.x[l] = xt[l]; jobs.dW[l] = dW[l]; jobs.db[l] = db[l]; jobs.I[l] = dims[l]; jobs.O[l] = dims[l + 1];
    jobs.first[l] = first;
    first += ((dims[l + 1] + 15) / 16) * ((dims[l] + 15) / 16);
  }
  jobs.first[3] = first;
  hipLaunchKernelGGL(lt_wgrad_kernel, dim3((unsigned)first), dim3(1024), 0, stream, jobs);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural
