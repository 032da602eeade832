// Channel-last kernels of the composed SNV training step (snv_train.hip): every activation of a tower lives as [B][L][32] fp32,
// so a tile of R batch rows is ONE contiguous block of memory whose 16-byte pieces are the 16-byte chunks of the LDS image
// [column][32 channels] that the prediction kernel uses (mfma_tile.h: XOR-swizzled chunks, ds_read_b128 operands) -- staging,
// MFMA operand reads and the stream-out move 4 channels per instruction, and per-channel sums ride in registers because a
// thread always handles the same 4 channels.  (The NCL kernels of conv32_mfma.hip transpose through LDS with 4-byte accesses:
// measured there, scatter + stream-out + column sums cost as much as the MFMA phases.)
//
//   conv32cl_fwd_kernel : y = conv32(BN(act(x))) [+ bias] [+ res1 + res2] [relu], BatchNorm finalised from the batch sums in the
//                         prologue (workgroup 0 writes the state and the running statistics), batch sums of act(y) for the next
//                         BatchNorm taken while the tile streams out       (reference: nn.Conv1d / BatchNorm1d of model_snv.py:350-430
//                         under model.train(), training.py:424)
//   conv32cl_bwd_kernel : weight / bias gradient partial rows, dz = input gradient of the conv, and the BatchNorm-backward sums
//                         (sum dz, sum dz * xhat) from one staging of dy and x          (loss.backward(), training.py:427)
//   streaming kernels   : BatchNorm statistics / backward apply, max-pools with arg-max, global max -- same math as train_ops.hip
#include <cstdlib>
#include <cstring>

#include "conv32_cl.h"
#include "conv32_jobs.h"

namespace mural {
namespace {

constexpr int CL_DEPTH = 4;        // 16-byte global loads a thread keeps in flight while staging / streaming
constexpr int CL_CUS = 256;        // compute units of an MI355X: the grid is one resident set of workgroups walking the tiles

struct ClTile {                    // tile of R rows on a flattened column axis with zero separators (stage 0 of a TowerGeom)
  TowerGeom g;
  int R, L, B;
  int nbuf;                        // floats per LDS image
};

bool cl_tile(int B, int L, ClTile* t) {
  std::memset(t, 0, sizeof(*t));
  const int Sc = L + 1;
  int r = (16 * 2 * SNV_NB2MAX - 1) / Sc;      // the tallest tile wins (measured: 9 / 12-block tiles are 7 % slower over the step)
  if (r < 1) return false;
  if (r > B) r = B;
  const int balanced = (B + 1023) / 1024;       // large batches of short rows: >= 1024 tiles rather than the tallest tile
  if (B >= 1024 && r > balanced) r = balanced;
  t->R = r;
  t->L = L;
  t->B = B;
  TowerGeom& g = t->g;
  g.L[0] = L;
  g.Sc[0] = Sc;
  g.NC[0] = 1 + r * Sc;
  g.nb[0] = (g.NC[0] + 15) / 16;
  g.dL[0] = FastDiv::make((uint32_t)L);
  g.dSc[0] = FastDiv::make((uint32_t)Sc);
  t->nbuf = (16 * g.nb[0] + 2) * CL_C;
  return g.nb[0] <= 2 * SNV_NB2MAX;
}

// zero the image columns that hold no data: guard + leading separator, the separator behind every row, everything behind the
// last row present (rows < R on the last tile) up to the end of the image
__device__ __forceinline__ void cl_zero_gaps(float* img, const ClTile& t, int rows, int tid) {
  const int Sc = t.g.Sc[0];
  const int ncols = 16 * t.g.nb[0] + 2;
  const int first_free = 2 + rows * Sc;                       // image column behind the separator of the last present row
  const int n_tail = ncols - first_free;
  const int n = 2 + rows + (n_tail > 0 ? n_tail : 0);
  for (int task = tid; task < n * 8; task += SNV_THREADS) {
    const int k = task >> 3;
    int pc;
    if (k < 2) pc = k;                                        // guard, leading separator
    else if (k < 2 + rows) pc = 1 + (k - 1) * Sc;             // separator behind row k - 2
    else pc = first_free + (k - 2 - rows);
    st4(img + lds_off(pc, task & 7), splat(0.f));
  }
}

// stage `rows` batch rows (contiguous [rows * L][32] floats at src) into the image: v' = t + s * (act(v) - m) per channel (aff != 0)
template <bool AFF, bool SUM>
__device__ __forceinline__ void cl_stage(const float* __restrict__ src, int rows, const ClTile& t, f32x4 s4, f32x4 t4, f32x4 m4, int relu,
                                         float* img, int tid, f32x4* colsum) {
  const int total = rows * t.L * 8;
  const int Sc = t.g.Sc[0];
  const int chunk = tid & 7;
  for (int base = tid; base < total; base += SNV_THREADS * CL_DEPTH) {
    f32x4 v[CL_DEPTH];
#pragma unroll
    for (int q = 0; q < CL_DEPTH; ++q) {
      const int task = base + q * SNV_THREADS;
      v[q] = task < total ? ld4(src + (size_t)task * 4) : splat(0.f);
    }
#pragma unroll
    for (int q = 0; q < CL_DEPTH; ++q) {
      const int task = base + q * SNV_THREADS;
      if (task >= total) break;
      const uint32_t col = (uint32_t)task >> 3;
      const uint32_t r = t.g.dL[0].div(col);
      const int l = (int)(col - r * (uint32_t)t.L);
      f32x4 x = v[q];
      if (SUM) *colsum += x;
      if (relu) x = max4(x, splat(0.f));
      if (AFF) x = f32x4{fmaf(s4.x, x.x - m4.x, t4.x), fmaf(s4.y, x.y - m4.y, t4.y), fmaf(s4.z, x.z - m4.z, t4.z), fmaf(s4.w, x.w - m4.w, t4.w)};
      st4(img + lds_off(2 + (int)r * Sc + l, chunk), x);
    }
  }
}

// two per-channel sums of a 256-thread workgroup into its accumulator slot: lanes with the same chunk (tid & 7) hold the same four
// channels; they meet through shuffles inside a wave and through `red` (64 floats per wave, dead LDS) across the four waves, so a
// workgroup issues 64 double atomics instead of 256 -- at 1024 workgroups the atomics of one launch otherwise queue for ~10 us on
// the few L2 channels the 16 KB accumulator block maps to
__device__ __forceinline__ void cl_slot_add(f32x4 a1, f32x4 a2, double* slot, float* red, int tid) {
  const int lane = tid & 63, wave = tid >> 6, chunk = tid & 7;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v1 = a1[q], v2 = a2[q];
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
      v1 += __shfl_xor(v1, off);
      v2 += __shfl_xor(v2, off);
    }
    if (lane < 8) {
      red[wave * 64 + 4 * chunk + q] = v1;
      red[wave * 64 + CL_C + 4 * chunk + q] = v2;
    }
  }
  __syncthreads();
  if (tid < 64) atomicAdd(&slot[tid], (double)((red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid])));
}

// ------------------------------------------------------------------------------------------------------------ forward
struct ClFwdArgs {
  ClTile t;
  const float* x;
  float* y;
  const float* W;
  const float* bias;
  const float* res1;
  const float* res2;
  ClFin fin;
  int pre_relu, post_relu;
  double* stat_out;       // nullptr: no sums
  int stat_relu;
  int dbg;                // timing experiments (MURAL_DEBUG_CL): 1 no staging, 2 no MFMA phase, 4 no stream-out, 8 no residual loads
};

__global__ __launch_bounds__(SNV_THREADS, 2) void conv32cl_fwd_kernel(const ClFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const ClTile& t = a.t;
  float* bufA = smem;
  float* bufB = smem + t.nbuf;
  float* aux = smem + 2 * t.nbuf;                               // scale | beta | mean
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave & 1, cgp = wave >> 1;
  const int n16 = lane & 15, kk = lane >> 4;
  const int chv = 16 * mb + 4 * kk;
  const int chunk = tid & 7;
  float af[SNV_KSTEPS];
  cl_frags(a.W, 0, mb, n16, kk, af);
  const f32x4 pb = a.bias ? ld4(a.bias + chv) : splat(0.f);
  if (!(a.dbg & 32)) cl_finalize(a.fin, aux, reinterpret_cast<double*>(bufA), tid);
  const f32x4 s4 = ld4(aux + 4 * chunk), t4 = ld4(aux + CL_C + 4 * chunk), m4 = ld4(aux + 2 * CL_C + 4 * chunk);
  const TowerGeom& g = t.g;
  const StageAddr sa = stage_setup(g, 0, t.R, n16, kk, mb, cgp);
  const int nbw = g.nb[0] > cgp ? (g.nb[0] - cgp + 1) / 2 : 0;
  // residual plan: (row r, column l) of the lane's column in each owned block as an offset inside the tile, ~0u = no data
  uint32_t plan[SNV_NB2MAX];
#pragma unroll
  for (int i = 0; i < SNV_NB2MAX; ++i) {
    plan[i] = ~0u;
    const int c = 16 * (cgp + 2 * i) + n16;
    if (i < nbw && c >= 1) {
      const uint32_t u = (uint32_t)(c - 1);
      const uint32_t r = g.dSc[0].div(u);
      const uint32_t l = u - r * (uint32_t)g.Sc[0];
      if (r < (uint32_t)t.R && l < (uint32_t)t.L) plan[i] = (r << 16) | l;
    }
  }
  LayerK lk;
  lk.lo = a.post_relu ? 0.f : -INFINITY;
  lk.ku = 0.f;
  lk.kx = 1.f;
  lk.kr = 1.f;
  f32x4 sum1 = splat(0.f), sum2 = splat(0.f);
  const int64_t ntiles = ((int64_t)t.B + t.R - 1) / t.R;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * t.R;
    const int rows = (int)((t.B - b0) < t.R ? (t.B - b0) : t.R);
    const size_t base = (size_t)b0 * t.L * CL_C;
    if (!(a.dbg & 1)) cl_stage<true, false>(a.x + base, rows, t, s4, t4, m4, a.pre_relu, bufA, tid, nullptr);
    cl_zero_gaps(bufA, t, rows, tid);
    f32x4 xres[SNV_NB2MAX];
    // residual operands of the lane's columns: every load is issued (out-of-plan lanes re-read the tile's first element and drop
    // it), so the nine round trips overlap instead of queueing behind one branch each
#pragma unroll
    for (int i = 0; i < SNV_NB2MAX; ++i) xres[i] = splat(0.f);
    if (a.res1 && !(a.dbg & 8)) {
      const float* r2p = a.res2 ? a.res2 : a.res1;
      f32x4 u[SNV_NB2MAX], w[SNV_NB2MAX];
      bool ok[SNV_NB2MAX];
#pragma unroll
      for (int i = 0; i < SNV_NB2MAX; ++i) {
        ok[i] = i < nbw && plan[i] != ~0u && (int)(plan[i] >> 16) < rows;
        const size_t o = base + (ok[i] ? ((size_t)(plan[i] >> 16) * t.L + (plan[i] & 0xFFFFu)) * CL_C + chv : 0);
        u[i] = ld4(a.res1 + o);
        w[i] = ld4(r2p + o);
      }
#pragma unroll
      for (int i = 0; i < SNV_NB2MAX; ++i)
        if (ok[i]) xres[i] = a.res2 ? u[i] + w[i] : u[i];
    }
    __syncthreads();
    if (!(a.dbg & 2))
      conv_layer(reinterpret_cast<const char*>(bufA), reinterpret_cast<char*>(bufB), sa, nbw, lk, af, pb, splat(1.f), splat(0.f), xres);
    __syncthreads();
    // the tile leaves as one contiguous stream of 16-byte pieces; the sums of act(y) ride along in registers
    const int total = (a.dbg & 4) ? 0 : rows * t.L * 8;
    for (int task = tid; task < total; task += SNV_THREADS) {
      const uint32_t col = (uint32_t)task >> 3;
      const uint32_t r = g.dL[0].div(col);
      const int l = (int)(col - r * (uint32_t)t.L);
      const f32x4 v = ld4(bufB + lds_off(2 + (int)r * g.Sc[0] + l, chunk));
      st4(a.y + base + (size_t)task * 4, v);
      if (a.stat_out) {
        const f32x4 w = a.stat_relu ? max4(v, splat(0.f)) : v;
        sum1 += w;
        sum2 += f32x4{w.x * w.x, w.y * w.y, w.z * w.z, w.w * w.w};
      }
    }
    // (no barrier here: the next tile stages into bufA, and the barrier behind that staging separates this stream-out from the
    // next conv's writes to bufB)
  }
  if (a.stat_out && !(a.dbg & 16))     // bufA is dead: every wave is past the barrier in front of the last stream-out
    cl_slot_add(sum1, sum2, a.stat_out + (size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 * CL_C, bufA, tid);
}

// ------------------------------------------------------------------------------------------------------------ backward
struct ClBwdArgs {
  ClTile t;
  const float* dy;
  const float* x;
  const float* W;
  const float* state;     // scale | beta | mean | invstd of the BatchNorm in front of the conv
  int pre_relu;
  float* part;            // [grid][32*32*3 + 32]
  float* dz;
  double* stat_out;       // sum(dz), sum(dz * xhat)
  int dbg;                // timing experiments (MURAL_DEBUG_CL): 64 no staging, 128 no weight gradient, 256 no input gradient, 512 no stream-out
};

__global__ __launch_bounds__(SNV_THREADS, 2) void conv32cl_bwd_kernel(const ClBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const ClTile& t = a.t;
  float* bufG = smem;                                            // dy image
  float* bufA = smem + t.nbuf;                                   // BN(act(x)) image, later dz
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave & 1, cgp = wave >> 1;
  const int n16 = lane & 15, kk = lane >> 4;
  const int chunk = tid & 7;
  float af[SNV_KSTEPS];
  cl_frags(a.W, 1, mb, n16, kk, af);
  const f32x4 s4 = ld4(a.state + 4 * chunk), t4 = ld4(a.state + CL_C + 4 * chunk);
  const f32x4 mean4 = ld4(a.state + 2 * CL_C + 4 * chunk), inv4 = ld4(a.state + 3 * CL_C + 4 * chunk);
  const TowerGeom& g = t.g;
  const StageAddr sa = stage_setup(g, 0, t.R, n16, kk, mb, cgp);
  const int nbw = g.nb[0] > cgp ? (g.nb[0] - cgp + 1) / 2 : 0;
  f32x4 wacc[2][3][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int h = 0; h < 2; ++h) wacc[m][tp][h] = splat(0.f);
  f32x4 bsum = splat(0.f), sdz = splat(0.f), sdzx = splat(0.f);
  const int nk = 4 * g.nb[0];
  const int k_lo = wave * nk / 4, k_hi = (wave + 1) * nk / 4;
  LayerK lk;
  lk.lo = -INFINITY;
  lk.ku = 0.f;
  lk.kx = 1.f;
  lk.kr = 0.f;
  const int64_t ntiles = ((int64_t)t.B + t.R - 1) / t.R;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * t.R;
    const int rows = (int)((t.B - b0) < t.R ? (t.B - b0) : t.R);
    const size_t base = (size_t)b0 * t.L * CL_C;
    if (!(a.dbg & 64)) {
      cl_stage<false, true>(a.dy + base, rows, t, splat(1.f), splat(0.f), splat(0.f), 0, bufG, tid, &bsum);
      cl_stage<true, false>(a.x + base, rows, t, s4, t4, mean4, a.pre_relu, bufA, tid, nullptr);
    }
    cl_zero_gaps(bufG, t, rows, tid);
    cl_zero_gaps(bufA, t, rows, tid);
    __syncthreads();
    // ---- weight gradient: dW[co][ci][tap] += dy[col][co] * act[col + tap - 1][ci]; M = co, N = ci, K = 4 columns per step, this
    //      wave's quarter of the columns.  Logical column c sits at image column c + 1, its tap-t neighbour at c + t.
    for (int s = k_lo; s < ((a.dbg & 128) ? k_lo : k_hi); ++s) {
      const int pc = 4 * s + kk;
      float gv[2], bv[3][2];
#pragma unroll
      for (int m = 0; m < 2; ++m) gv[m] = bufG[lds_off(pc + 1, (16 * m + n16) >> 2) + (n16 & 3)];
#pragma unroll
      for (int tp = 0; tp < 3; ++tp)
#pragma unroll
        for (int h = 0; h < 2; ++h) bv[tp][h] = bufA[lds_off(pc + tp, (16 * h + n16) >> 2) + (n16 & 3)];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int tp = 0; tp < 3; ++tp)
#pragma unroll
          for (int h = 0; h < 2; ++h) wacc[m][tp][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[m], bv[tp][h], wacc[m][tp][h], 0, 0, 0);
    }
    __syncthreads();                                             // the BN(act(x)) image is dead: the input gradient goes over it
    {
      f32x4 xres[SNV_NB2MAX];
#pragma unroll
      for (int i = 0; i < SNV_NB2MAX; ++i) xres[i] = splat(0.f);
      if (!(a.dbg & 256))
        conv_layer(reinterpret_cast<const char*>(bufG), reinterpret_cast<char*>(bufA), sa, nbw, lk, af, splat(0.f), splat(1.f), splat(0.f), xres);
    }
    __syncthreads();
    // ---- stream dz out; sum dz and sum dz * xhat ride along (x is read again: an L2 hit a few microseconds after its staging)
    const int total = (a.dbg & 512) ? 0 : rows * t.L * 8;
    for (int bt = tid; bt < total; bt += SNV_THREADS * CL_DEPTH) {
      f32x4 xv[CL_DEPTH];
#pragma unroll
      for (int q = 0; q < CL_DEPTH; ++q) {
        const int task = bt + q * SNV_THREADS;
        xv[q] = task < total ? ld4(a.x + base + (size_t)task * 4) : splat(0.f);
      }
#pragma unroll
      for (int q = 0; q < CL_DEPTH; ++q) {
        const int task = bt + q * SNV_THREADS;
        if (task >= total) break;
        const uint32_t col = (uint32_t)task >> 3;
        const uint32_t r = g.dL[0].div(col);
        const int l = (int)(col - r * (uint32_t)t.L);
        const f32x4 v = ld4(bufA + lds_off(2 + (int)r * g.Sc[0] + l, chunk));
        st4(a.dz + base + (size_t)task * 4, v);
        f32x4 xr = xv[q];
        if (a.pre_relu) xr = max4(xr, splat(0.f));
        sdz += v;
        sdzx += f32x4{v.x * ((xr.x - mean4.x) * inv4.x), v.y * ((xr.y - mean4.y) * inv4.y), v.z * ((xr.z - mean4.z) * inv4.z),
                      v.w * ((xr.w - mean4.w) * inv4.w)};
      }
    }
    __syncthreads();
  }
  // BatchNorm-backward sums and the bias gradient: lanes with the same chunk hold the same 4 channels
  double* slot = a.stat_out + (size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 * CL_C;
  constexpr int NW = CL_C * CL_C * 3;
  __syncthreads();
  float* wred = smem;                                            // [4 waves][NW + 32] floats: the images are dead
  cl_slot_add(sdz, sdzx, slot, wred, tid);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v3 = bsum[q];
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) v3 += __shfl_xor(v3, off);
    if (lane < 8) wred[(size_t)wave * (NW + CL_C) + NW + 4 * chunk + q] = v3;    // every wave covers all 32 channels with its own columns
  }
  // D[row = co 4 kk + r][col = ci n16] of tile (m, tap, h) -> dW[16 m + 4 kk + r][16 h + n16][tap]
  float* mine = wred + (size_t)wave * (NW + CL_C);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[((16 * m + 4 * kk + r) * CL_C + 16 * h + n16) * 3 + tp] = wacc[m][tp][h][r];
  __syncthreads();
  float* dst = a.part + (size_t)blockIdx.x * (NW + CL_C);
  for (int i = tid; i < NW + CL_C; i += SNV_THREADS)
    dst[i] = (wred[i] + wred[(NW + CL_C) + i]) + (wred[2 * (NW + CL_C) + i] + wred[3 * (NW + CL_C) + i]);
}

// ------------------------------------------------------------------------------------------------------------ streaming kernels
// batch sums of act(x) and act(x)^2 per channel of a [rows][32] tensor (rows = B * L)
__global__ __launch_bounds__(256) void bn_stats_cl_kernel(const float* __restrict__ x, int64_t rows, int relu, double* __restrict__ acc) {
  f32x4 s1 = splat(0.f), s2 = splat(0.f);
  const int64_t total = rows * 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 v = ld4(x + i * 4);
    if (relu) v = max4(v, splat(0.f));
    s1 += v;
    s2 += f32x4{v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w};
  }
  __shared__ float red[256];
  cl_slot_add(s1, s2, acc + (size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 * CL_C, red, threadIdx.x);
}

// dx = a'(x) * gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)) [+ add1 + add2]; workgroup 0 writes dgamma / dbeta.
// blockIdx.y = job (conv32_jobs.h: the two towers in one launch)
struct BnApplyArgs2 { BnApplyJob j[TOWER_JOBS]; };
__global__ __launch_bounds__(256) void bn_bwd_apply_cl_kernel(const BnApplyArgs2 aa) {
  const BnApplyJob& a = aa.j[blockIdx.y];
  const float* __restrict__ dz = a.dz;
  const float* __restrict__ x = a.x;
  const float* __restrict__ add1 = a.add1;
  const float* __restrict__ add2 = a.add2;
  float* __restrict__ dx = a.dx;
  const int relu = a.relu;
  const double n = (double)a.rows;
  __shared__ float cst[4][CL_C];      // gamma * invstd, mean(dz), mean(dz * xhat), mean
  __shared__ float inv[CL_C];
  if (threadIdx.x < CL_C) {
    const int c = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < MURAL_BN_SLOTS; ++k) {
      s1 += a.acc[((size_t)k * 2 + 0) * CL_C + c];
      s2 += a.acc[((size_t)k * 2 + 1) * CL_C + c];
    }
    cst[0][c] = a.gamma[c] * a.state[3 * CL_C + c];
    cst[1][c] = (float)(s1 / n);
    cst[2][c] = (float)(s2 / n);
    cst[3][c] = a.state[2 * CL_C + c];
    inv[c] = a.state[3 * CL_C + c];
    if (blockIdx.x == 0) {
      a.dgamma[c] = (float)s2;
      a.dbeta[c] = (float)s1;
    }
  }
  __syncthreads();
  const int chunk = threadIdx.x & 7;
  const f32x4 k0 = ld4(&cst[0][4 * chunk]), m1 = ld4(&cst[1][4 * chunk]), m2 = ld4(&cst[2][4 * chunk]), mu = ld4(&cst[3][4 * chunk]),
              is = ld4(&inv[4 * chunk]);
  const int64_t total = a.rows * 8;
  // Every load of an iteration is issued before the first use, the optional residual gradients through range-checked descriptors (an
  // absent tensor has one of 0 bytes and reads 0): `add1 ? ld4(add1 + i * 4) : 0` put each of them behind a branch with a full wait.
  // Two 16-byte pieces per lane and iteration in flight.  (Tensors of 2 GB and more keep the plain loop.)
  const uint64_t bytes = (uint64_t)total * 16;
  if (bytes < (1ull << 31)) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, (int)bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(add1 ? add1 : x), 0, add1 ? (int)bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(add2 ? add2 : x), 0, add2 ? (int)bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(dx, 0, (int)bytes, 0x00020000);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i0 < total; i0 += 2 * stride) {
      f32x4 raw[2], d[2], r1[2], r2[2];
      uint32_t off[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t i = i0 + u * stride;
        off[u] = i < total ? (uint32_t)(i * 16) : 0x80000000u;
        raw[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off[u], 0, 0));
        d[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, off[u], 0, 0));
        r1[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1d, off[u], 0, 0));
        r2[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2d, off[u], 0, 0));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = relu ? fmaxf(raw[u][q], 0.f) : raw[u][q];
          const float xh = (v - mu[q]) * is[q];
          float gq = k0[q] * (d[u][q] - m1[q] - xh * m2[q]);
          if (relu && raw[u][q] <= 0.f) gq = 0.f;
          o[q] = (gq + r1[u][q]) + r2[u][q];
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{o[0], o[1], o[2], o[3]}), ro, off[u], 0, 0);
      }
    }
    return;
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 raw = ld4(x + i * 4), d = ld4(dz + i * 4);
    const f32x4 r1 = add1 ? ld4(add1 + i * 4) : splat(0.f), r2 = add2 ? ld4(add2 + i * 4) : splat(0.f);
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v = relu ? fmaxf(raw[q], 0.f) : raw[q];
      const float xh = (v - mu[q]) * is[q];
      float gq = k0[q] * (d[q] - m1[q] - xh * m2[q]);
      if (relu && raw[q] <= 0.f) gq = 0.f;
      o[q] = (gq + r1[q]) + r2[q];
    }
    st4(dx + i * 4, o);
  }
}

// MaxPool1d(k, s, p) on [B][L][32] -> [B][Lout][32] with the arg-max column (first maximum wins); thread = (row, pooled column,
// 4 channels); blockIdx.y = job
struct PoolFwdArgs2 { PoolFwdJob j[TOWER_JOBS]; };
__global__ __launch_bounds__(256) void maxpool_cl_fwd_kernel(const PoolFwdArgs2 aa) {
  const PoolFwdJob& a = aa.j[blockIdx.y];
  const float* __restrict__ x = a.x;
  float* __restrict__ y = a.y;
  int32_t* __restrict__ arg = a.arg;
  const int L = a.L, k = a.k, s = a.s, p = a.p;
  const int Lout = (L + 2 * p - k) / s + 1;
  __shared__ float red[256];
  f32x4 s1 = splat(0.f), s2 = splat(0.f);     // batch sums of the pooled values for the BatchNorm behind the pool (acc != nullptr)
  const int64_t total = a.B * Lout * 8;
  const uint64_t xbytes = (uint64_t)a.B * L * CL_C * 4;
  const __amdgpu_buffer_rsrc_t rxd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, xbytes < (1ull << 31) ? (int)xbytes : 0, 0x00020000);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int chunk = (int)(i & 7);
    const int64_t bc = i >> 3;
    const int64_t b = bc / Lout;
    const int jo = (int)(bc - b * Lout);
    const int jlo = jo * s - p;
    f32x4 m = splat(-INFINITY);
    int am[4] = {0, 0, 0, 0};
    if (k <= 8 && xbytes < (1ull << 31)) {
      // the window's columns requested together (a load per tap under `if (in range)` waited for each one in turn: seven round trips
      // per output at k = 7); a column outside the row aims past the descriptor and is masked to -inf, same comparison order
      f32x4 v[8];
      bool ok[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const int j = jlo + w;
        ok[w] = (w < k) & (j >= 0) & (j < L);
        uint32_t off = (uint32_t)(((b * L + j) * CL_C + 4 * chunk) * 4);
        asm volatile("" : "+v"(off));
        v[w] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxd, ok[w] ? off : 0x80000000u, 0, 0));
      }
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const int j = jlo + w;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool gt = ok[w] & (v[w][q] > m[q]);
          m[q] = gt ? v[w][q] : m[q];
          am[q] = gt ? j : am[q];
        }
      }
    } else {
      for (int w = 0; w < k; ++w) {
        const int j = jlo + w;
        if (j < 0 || j >= L) continue;
        const f32x4 v = ld4(x + ((size_t)(b * L + j) * CL_C) + 4 * chunk);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (v[q] > m[q]) {
            m[q] = v[q];
            am[q] = j;
          }
      }
    }
    st4(y + (size_t)bc * CL_C + 4 * chunk, m);
    int32_t* ap = arg + (size_t)bc * CL_C + 4 * chunk;
    ap[0] = am[0]; ap[1] = am[1]; ap[2] = am[2]; ap[3] = am[3];
    s1 += m;
    s2 += f32x4{m.x * m.x, m.y * m.y, m.z * m.z, m.w * m.w};
  }
  if (a.acc) cl_slot_add(s1, s2, a.acc + (size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 * CL_C, red, threadIdx.x);
}

// gather backward for disjoint windows (stride >= kernel: every pool of the model): dx[b][l][c] = arg[b][lo][c] == l ? dy[b][lo][c] : 0
struct PoolBwdArgs2 { PoolBwdJob j[TOWER_JOBS]; };
__global__ __launch_bounds__(256) void maxpool_cl_bwd_kernel(const PoolBwdArgs2 aa) {
  const PoolBwdJob& a = aa.j[blockIdx.y];
  const float* __restrict__ dy = a.dy;
  const int32_t* __restrict__ arg = a.arg;
  float* __restrict__ dx = a.dx;
  const int L = a.L, Lout = a.Lout, s = a.s, p = a.p;
  const int64_t total = a.B * L * 8;
  const bool fold = a.fdz != nullptr;
  __shared__ float cst[5][CL_C];      // fold: gamma * invstd | mean(dz) | mean(dz * xhat) | mean | invstd of the BatchNorm behind the pool
  if (fold) {
    if (threadIdx.x < CL_C) {
      const int c = threadIdx.x;
      const double n = (double)a.B * Lout;
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < MURAL_BN_SLOTS; ++k) {
        s1 += a.facc[((size_t)k * 2 + 0) * CL_C + c];
        s2 += a.facc[((size_t)k * 2 + 1) * CL_C + c];
      }
      cst[0][c] = a.fgamma[c] * a.fstate[3 * CL_C + c];
      cst[1][c] = (float)(s1 / n);
      cst[2][c] = (float)(s2 / n);
      cst[3][c] = a.fstate[2 * CL_C + c];
      cst[4][c] = a.fstate[3 * CL_C + c];
      if (blockIdx.x == 0) {
        a.fdgamma[c] = (float)s2;
        a.fdbeta[c] = (float)s1;
      }
    }
    __syncthreads();
  }
  const int chunk0 = threadIdx.x & 7;
  const f32x4 k0 = fold ? ld4(&cst[0][4 * chunk0]) : splat(0.f), m1 = fold ? ld4(&cst[1][4 * chunk0]) : splat(0.f),
              m2 = fold ? ld4(&cst[2][4 * chunk0]) : splat(0.f), mu = fold ? ld4(&cst[3][4 * chunk0]) : splat(0.f),
              is = fold ? ld4(&cst[4][4 * chunk0]) : splat(0.f);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int chunk = (int)(i & 7);      // (== chunk0: the grid stride is a multiple of 8)
    const int64_t bl = i >> 3;
    const int64_t b = bl / L;
    const int l = (int)(bl - b * L);
    const int jo = (l + p) / s;
    f32x4 o = splat(0.f);
    if (jo < Lout) {
      const size_t src = ((size_t)(b * Lout + jo)) * CL_C + 4 * chunk;
      f32x4 g;
      if (fold) {      // bn_bwd_apply_cl_kernel's arithmetic on the pooled element
        const f32x4 d = ld4(a.fdz + src), raw = ld4(a.fx + src);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = a.frelu ? fmaxf(raw[q], 0.f) : raw[q];
          const float xh = (v - mu[q]) * is[q];
          float gq = k0[q] * (d[q] - m1[q] - xh * m2[q]);
          if (a.frelu && raw[q] <= 0.f) gq = 0.f;
          g[q] = gq;
        }
      } else {
        g = ld4(dy + src);
      }
      const int32_t* ap = arg + src;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = ap[q] == l ? g[q] : 0.f;
    }
    st4(dx + (size_t)bl * CL_C + 4 * chunk, o);
  }
}

// global max over the columns of [B][L][32] -> feat [B][32] + arg-max column; relu: feat = max(0, .) -- the conv in front wrote its
// raw output and the ReLU of conv3 (model_snv.py:386-387) is taken here (max_l relu(v) = relu(max_l v); the backward masks on v > 0)
struct GmaxFwdArgs2 { GmaxFwdJob j[TOWER_JOBS]; };
__global__ __launch_bounds__(256) void gmax_cl_kernel(const GmaxFwdArgs2 aa) {
  const GmaxFwdJob& a = aa.j[blockIdx.y];
  const float* __restrict__ x = a.x;
  const int L = a.L;
  const int64_t total = a.B * 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int chunk = (int)(i & 7);
    const int64_t b = i >> 3;
    f32x4 m = splat(-INFINITY);
    int am[4] = {0, 0, 0, 0};
    for (int l = 0; l < L; ++l) {
      const f32x4 v = ld4(x + ((size_t)(b * L + l)) * CL_C + 4 * chunk);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (v[q] > m[q]) {
          m[q] = v[q];
          am[q] = l;
        }
    }
    if (a.relu) m = max4(m, splat(0.f));
    st4(a.feat + (size_t)b * CL_C + 4 * chunk, m);
    int32_t* ap = a.arg + (size_t)b * CL_C + 4 * chunk;
    ap[0] = am[0]; ap[1] = am[1]; ap[2] = am[2]; ap[3] = am[3];
  }
}

// backward of the global max and of the ReLU in front of it: dc3[b][l][c] = (l == arg[b][c] && c3[b][l][c] > 0) ? dfeat[b][c] : 0
// (c3: the conv output, raw or behind its ReLU -- the mask is the same)
struct GmaxBwdArgs2 { GmaxBwdJob j[TOWER_JOBS]; };
__global__ __launch_bounds__(256) void gmax_relu_bwd_cl_kernel(const GmaxBwdArgs2 aa) {
  const GmaxBwdJob& a = aa.j[blockIdx.y];
  const int L = a.L;
  const int64_t total = a.B * L * 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int chunk = (int)(i & 7);
    const int64_t bl = i >> 3;
    const int64_t b = bl / L;
    const int l = (int)(bl - b * L);
    const f32x4 g = ld4(a.dfeat + (size_t)b * CL_C + 4 * chunk), v = ld4(a.c3 + (size_t)bl * CL_C + 4 * chunk);
    const int32_t* ap = a.arg + (size_t)b * CL_C + 4 * chunk;
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = (ap[q] == l && v[q] > 0.f) ? g[q] : 0.f;
    st4(a.dx + (size_t)bl * CL_C + 4 * chunk, o);
  }
}

// phase-ablation switches of the timing tools (tools/time_conv32_cl.py), read once
int debug_phases() {
  static const int v = dev_env("MURAL_DEBUG_CL") ? atoi(dev_env("MURAL_DEBUG_CL")) : 0;
  return v;
}

int cl_grid(int64_t total, int cap = 8192) {
  const int64_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// ---- host entry points (snv_train.hip) --------------------------------------------------------------------------------------
int cl_conv32_supported(int L) {
  ClTile t;
  return cl_tile(2, L, &t) ? 1 : 0;
}

int cl_conv32_fwd(const float* x, int64_t B, int L, int pre_relu, const double* acc, const float* gamma, const float* beta, float eps,
                  float momentum, float* running_mean, float* running_var, float* state, const float* W, const float* bias, int post_relu,
                  const float* res1, const float* res2, double* acc_out, int out_relu, float* y, hipStream_t stream) {
  if (B == 0 || L == 0) return MURAL_OK;
  ClFwdArgs a;
  std::memset(&a, 0, sizeof(a));
  MURAL_REQUIRE(cl_tile((int)B, L, &a.t), "conv32 (channel-last): L = %d does not fit the LDS tile", L);
  if (!res1) { res1 = res2; res2 = nullptr; }
  a.x = x; a.y = y; a.W = W; a.bias = bias; a.res1 = res1; a.res2 = res2; a.pre_relu = pre_relu; a.post_relu = post_relu;
  a.stat_out = acc_out; a.stat_relu = out_relu;
  a.fin = ClFin{acc, (double)B * L, gamma, beta, eps, momentum, running_mean, running_var, state};
  a.dbg = debug_phases();
  const size_t lds = (size_t)(2 * a.t.nbuf + 3 * CL_C) * 4;
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&conv32cl_fwd_kernel)) return rc;
  const int64_t ntiles = (B + a.t.R - 1) / a.t.R;
  int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
  per_cu = per_cu > 4 ? 4 : (per_cu < 2 ? 2 : per_cu);
  const int64_t cap = (int64_t)CL_CUS * per_cu;
  hipLaunchKernelGGL(conv32cl_fwd_kernel, dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(SNV_THREADS), lds, stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

size_t cl_conv32_part_floats() { return (size_t)1024 * (CL_C * CL_C * 3 + CL_C); }

int cl_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int L, const float* state, int pre_relu, float* dz,
                  double* stat_out, float* part, int* nrow, hipStream_t stream) {
  ClBwdArgs a;
  std::memset(&a, 0, sizeof(a));
  MURAL_REQUIRE(cl_tile((int)B, L, &a.t), "conv32_bwd (channel-last): L = %d does not fit the LDS tile", L);
  a.dy = dy; a.x = x; a.W = W; a.state = state; a.pre_relu = pre_relu; a.part = part; a.dz = dz; a.stat_out = stat_out;
  a.dbg = debug_phases();
  const int64_t ntiles = (B + a.t.R - 1) / a.t.R;
  size_t lds = (size_t)2 * a.t.nbuf * 4;
  const size_t lds_red = (size_t)4 * (CL_C * CL_C * 3 + CL_C) * 4;
  lds = lds > lds_red ? lds : lds_red;
  int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
  per_cu = per_cu > 4 ? 4 : (per_cu < 2 ? 2 : per_cu);
  const int64_t cap = (int64_t)CL_CUS * per_cu;
  const int grid = (int)(ntiles < cap ? ntiles : cap);
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&conv32cl_bwd_kernel)) return rc;
  hipLaunchKernelGGL(conv32cl_bwd_kernel, dim3(grid), dim3(SNV_THREADS), lds, stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  *nrow = grid;
  return MURAL_OK;
}

int cl_bn_stats(const float* x, int64_t rows, int relu, double* acc, hipStream_t stream) {
  if (rows == 0) return MURAL_OK;
  hipLaunchKernelGGL(bn_stats_cl_kernel, dim3(cl_grid(rows * 8, 2048)), dim3(256), 0, stream, x, rows, relu, acc);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

int cl_bn_bwd_apply_jobs(const BnApplyJob* jobs, int n, hipStream_t stream) {
  BnApplyArgs2 aa;
  std::memset(&aa, 0, sizeof(aa));
  int gx = 0, ny = 0;
  for (int i = 0; i < n; ++i) {
    if (jobs[i].rows == 0) continue;
    aa.j[ny++] = jobs[i];
    const int g = cl_grid(jobs[i].rows * 8, 2048);
    gx = g > gx ? g : gx;
  }
  if (ny == 0) return MURAL_OK;
  hipLaunchKernelGGL(bn_bwd_apply_cl_kernel, dim3(gx, ny), dim3(256), 0, stream, aa);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

int cl_bn_bwd_apply(const float* dz, const float* x, int64_t rows, int relu, const float* state, const float* gamma, const double* acc,
                    const float* add1, const float* add2, float* dx, float* dgamma, float* dbeta, hipStream_t stream) {
  const BnApplyJob j{dz, x, rows, relu, state, gamma, acc, add1, add2, dx, dgamma, dbeta};
  return cl_bn_bwd_apply_jobs(&j, 1, stream);
}

int cl_maxpool_fwd_jobs(const PoolFwdJob* jobs, int n, hipStream_t stream) {
  PoolFwdArgs2 aa;
  std::memset(&aa, 0, sizeof(aa));
  int gx = 0, ny = 0;
  for (int i = 0; i < n; ++i) {
    const PoolFwdJob& j = jobs[i];
    const int Lout = (j.L + 2 * j.p - j.k) / j.s + 1;
    if (j.B * Lout == 0) continue;
    aa.j[ny++] = j;
    const int g = cl_grid(j.B * Lout * 8, j.acc ? 1024 : 8192);
    gx = g > gx ? g : gx;
  }
  if (ny == 0) return MURAL_OK;
  hipLaunchKernelGGL(maxpool_cl_fwd_kernel, dim3(gx, ny), dim3(256), 0, stream, aa);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

int cl_maxpool_fwd(const float* x, int64_t B, int L, int k, int s, int p, float* y, int32_t* arg, double* acc, hipStream_t stream) {
  const PoolFwdJob j{x, B, L, k, s, p, y, arg, acc};
  return cl_maxpool_fwd_jobs(&j, 1, stream);
}

int cl_maxpool_bwd_jobs(const PoolBwdJob* jobs, int n, hipStream_t stream) {
  PoolBwdArgs2 aa;
  std::memset(&aa, 0, sizeof(aa));
  int gx = 0, ny = 0;
  for (int i = 0; i < n; ++i) {
    const PoolBwdJob& j = jobs[i];
    MURAL_REQUIRE(j.s >= j.k, "channel-last max-pool backward serves disjoint windows (stride >= kernel)");
    if (j.B * j.L == 0) continue;
    aa.j[ny++] = j;
    const int g = cl_grid(j.B * j.L * 8, j.fdz ? 2048 : 8192);      // (a fold: every workgroup sums the BatchNorm's 32 slots first)
    gx = g > gx ? g : gx;
  }
  if (ny == 0) return MURAL_OK;
  hipLaunchKernelGGL(maxpool_cl_bwd_kernel, dim3(gx, ny), dim3(256), 0, stream, aa);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

int cl_maxpool_bwd(const float* dy, const int32_t* arg, int64_t B, int L, int Lout, int k, int s, int p, float* dx, hipStream_t stream) {
  PoolBwdJob j{};
  j.dy = dy; j.arg = arg; j.B = B; j.L = L; j.Lout = Lout; j.k = k; j.s = s; j.p = p; j.dx = dx;
  return cl_maxpool_bwd_jobs(&j, 1, stream);
}

// the pool's backward with the BatchNorm-backward apply of the conv behind the pool folded in (PoolBwdJob: fold)
int cl_maxpool_bwd_fold(const BnApplyJob& f, const int32_t* arg, int64_t B, int L, int Lout, int k, int s, int p, float* dx, hipStream_t stream) {
  MURAL_REQUIRE(f.dz && f.x && f.state && f.gamma && f.acc && f.dgamma && f.dbeta && !f.add1 && !f.add2 && f.rows == B * Lout,
                "max-pool backward: bad BatchNorm fold");
  PoolBwdJob j{};
  j.arg = arg; j.B = B; j.L = L; j.Lout = Lout; j.k = k; j.s = s; j.p = p; j.dx = dx;
  j.fdz = f.dz; j.fx = f.x; j.fstate = f.state; j.fgamma = f.gamma; j.facc = f.acc; j.frelu = f.relu; j.fdgamma = f.dgamma; j.fdbeta = f.dbeta;
  return cl_maxpool_bwd_jobs(&j, 1, stream);
}

int cl_gmax_fwd_jobs(const GmaxFwdJob* jobs, int n, hipStream_t stream) {
  GmaxFwdArgs2 aa;
  std::memset(&aa, 0, sizeof(aa));
  int gx = 0, ny = 0;
  for (int i = 0; i < n; ++i) {
    if (jobs[i].B == 0) continue;
    aa.j[ny++] = jobs[i];
    const int g = cl_grid(jobs[i].B * 8);
    gx = g > gx ? g : gx;
  }
  if (ny == 0) return MURAL_OK;
  hipLaunchKernelGGL(gmax_cl_kernel, dim3(gx, ny), dim3(256), 0, stream, aa);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

int cl_gmax_fwd(const float* x, int64_t B, int L, float* feat, int32_t* arg, hipStream_t stream) {
  const GmaxFwdJob j{x, B, L, 0, feat, arg};
  return cl_gmax_fwd_jobs(&j, 1, stream);
}

int cl_gmax_relu_bwd_jobs(const GmaxBwdJob* jobs, int n, hipStream_t stream) {
  GmaxBwdArgs2 aa;
  std::memset(&aa, 0, sizeof(aa));
  int gx = 0, ny = 0;
  for (int i = 0; i < n; ++i) {
    if (jobs[i].B * jobs[i].L == 0) continue;
    aa.j[ny++] = jobs[i];
    const int g = cl_grid(jobs[i].B * jobs[i].L * 8);
    gx = g > gx ? g : gx;
  }
  if (ny == 0) return MURAL_OK;
  hipLaunchKernelGGL(gmax_relu_bwd_cl_kernel, dim3(gx, ny), dim3(256), 0, stream, aa);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

int cl_gmax_relu_bwd(const float* dfeat, const int32_t* arg, const float* c3, int64_t B, int L, float* dx, hipStream_t stream) {
  const GmaxBwdJob j{dfeat, arg, c3, B, L, dx};
  return cl_gmax_relu_bwd_jobs(&j, 1, stream);
}

}  // namespace mural
