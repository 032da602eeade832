// fp32-MFMA kernels for the 32->32, k=3 convolutions of the SNV towers on HBM-resident [B][32][L] tensors: forward /
// input gradient (same kernel, different weight fragments) and weight + bias gradient.  Used by the training step
// (MuRaL/training.py:424-427 autograd over nn.Conv1d, here one explicit kernel per direction).
//
// As in snv_towers_fused, R batch rows share one flattened column axis with zero separator columns (= the conv's
// zero padding); the tile is staged in LDS as [column][32 ch] with the conflict-free chunk permutation of mfma_tile.h,
// with the optional BatchNorm(+ReLU) affine applied while staging.
//   forward : D[16 cout][16 col] += W[cout][k] act[k][col], K = 3 taps x 32 ch; wave = (M-block, column parity)
//   wgrad   : dW[cout][(tap, cin)] += dy[cout][col] * act[cin][col + tap - 1]: M = cout, N = (tap, cin) = 6 blocks of 16,
//             K = columns in steps of 4; each wave reduces a quarter of the tile's columns into 12 accumulator tiles
#include <cstring>

#include "mfma_tile.h"

namespace mural {

constexpr int C32 = 32;
constexpr int C32_KSTEPS = 24;
constexpr int C32_NB2MAX = 9;          // <= 18 blocks of 16 columns per tile
constexpr int C32_MAXCOLS = 16 * 2 * C32_NB2MAX;

struct Conv32Args {
  const float* x;        // [B][32][L]
  float* y;              // [B][32][L]
  const float* W;        // PyTorch [32][32][3]; dgrad: use the transposed, tap-flipped filter
  int dgrad;
  // per-channel sums taken in the epilogue (the producer of a tensor knows its values: no extra pass over HBM)
  int stat_mode;         // 0 none | 1: sum, sum of squares of act(y) (batch statistics of the NEXT BatchNorm)
                         //        | 2: sum(y), sum(y * xhat), xhat = (act(stat_x) - mean) * invstd (BatchNorm backward of y = dz)
  int stat_relu;         // act = relu
  const float* stat_x;   // mode 2: [B][32][L]
  const float* stat_mean;
  const float* stat_invstd;
  double* stat_out;      // accumulator block [MURAL_BN_SLOTS][2][32] (zeroed by the caller), copy picked by workgroup index
  const float* bias;     // [32] or nullptr
  const float* pre_s;    // [32] or nullptr: x' = pre_s * act(x) + pre_t, act = relu if pre_relu
  const float* pre_t;
  const float* res1;     // optional residuals [B][32][L]
  const float* res2;
  int pre_relu, post_relu;
  int B, L, R, Sc, NC, nb;
  FastDiv dSc, dL;
};

// stage R rows of x (contiguous in memory) into the LDS image, applying the BN(+ReLU) affine; separators / tail = 0.
// `aff` is an LDS copy of pre_s | pre_t (64 floats) or nullptr.
__device__ __forceinline__ void stage_rows(const float* __restrict__ x, int64_t b0, int B, int L, int R, int Sc, int NC, int nb,
                                           const FastDiv& dL, const float* aff, int pre_relu, float* img, int tid) {
  const int rows = (int)((B - b0) < R ? (B - b0) : R);
  const int total = rows * C32 * L;                      // a multiple of 4 (32 channels)
  const float* src = x + (size_t)b0 * C32 * L;           // 128-byte aligned: float4 loads
  for (int i0 = tid * 4; i0 < total; i0 += 256 * 4) {
    const f32x4 v4 = ld4(src + i0);
    uint32_t rc = dL.div((uint32_t)i0);
    int l = i0 - (int)rc * L;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (l >= L) { l -= L; ++rc; }                      // next (row, channel)
      const int r = (int)rc >> 5, ci = (int)rc & 31;
      float v = v4[e];
      if (pre_relu) v = fmaxf(v, 0.f);
      if (aff) v = fmaf(aff[ci], v, aff[C32 + ci]);
      img[lds_off(1 + r * Sc + l + 1, ci >> 2) + (ci & 3)] = v;
      ++l;
    }
  }
  // zero: both guard columns (logical -1 and 16 nb), the separator columns, rows missing from a partial last tile, tail
  const int nsep = R + 1, first_tail = 1 + rows * Sc;    // columns >= first_tail hold no data
  const int nz = 2 + nsep + (16 * nb - first_tail > 0 ? 16 * nb - first_tail : 0);
  for (int i = tid; i < nz * 8; i += 256) {
    const int k = i >> 3, cg = i & 7;
    int c;
    if (k == 0) c = -1;
    else if (k == 1) c = 16 * nb;
    else if (k < 2 + nsep) c = (k - 2) * Sc;
    else c = first_tail + (k - 2 - nsep);
    if (c <= 16 * nb) st4(img + lds_off(c + 1, cg), splat(0.f));
  }
}

template <int STAT>   // = Conv32Args::stat_mode, compile-time so that the plain forward does not carry the epilogue sums
__global__ __launch_bounds__(256) void conv32_mfma_kernel(const Conv32Args a) {
  extern __shared__ __attribute__((aligned(16))) float img[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave & 1, cgp = wave >> 1;
  const int n16 = lane & 15, kk = lane >> 4;
  float af[C32_KSTEPS];   // A fragments: k-step s = 8 tap + 4 half + q holds W[cout = 16 mb + n16][cin = 16 half + 4 kk + q][tap]
#pragma unroll
  for (int s = 0; s < C32_KSTEPS; ++s) {
    const int t = s / 8, h = (s % 8) / 4, q = s % 4;
    const int cin = 16 * h + 4 * kk + q;               // channel of the tensor being convolved
    const int cout = 16 * mb + n16;                    // channel of the tensor being produced
    af[s] = a.W[a.dgrad ? (cin * C32 + cout) * 3 + (2 - t) : (cout * C32 + cin) * 3 + t];
  }
  const int chv = 16 * mb + 4 * kk;
  float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};

  const f32x4 bias = a.bias ? ld4(a.bias + chv) : splat(0.f);
  // per-lane LDS byte offsets of block 0 of this wave (block i: + 4096 i)
  uint32_t rd[6];
  const int c0 = 16 * cgp + n16;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) rd[2 * t + h] = 4u * (uint32_t)lds_off(c0 + t, 4 * h + kk);
  const int nbw = a.nb > cgp ? (a.nb - cgp + 1) / 2 : 0;
  const char* in = reinterpret_cast<const char*>(img);
  float* aff = img + (16 * a.nb + 2) * C32;              // pre_s | pre_t
  if (a.pre_s && tid < 2 * C32) aff[tid] = tid < C32 ? a.pre_s[tid] : a.pre_t[tid - C32];
  const float* sms = aff + 2 * C32;                      // stat_mean | stat_invstd (mode 2)
  if (STAT == 2 && tid < 2 * C32) aff[2 * C32 + tid] = tid < C32 ? a.stat_mean[tid] : a.stat_invstd[tid - C32];
  const int64_t ntiles = ((int64_t)a.B + a.R - 1) / a.R;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * a.R;
    __syncthreads();
    stage_rows(a.x, b0, a.B, a.L, a.R, a.Sc, a.NC, a.nb, a.dL, a.pre_s ? aff : nullptr, a.pre_relu, img, tid);
    __syncthreads();
#pragma unroll
    for (int ip = 0; ip < (C32_NB2MAX + 1) / 2; ++ip) {
      const int i0 = 2 * ip, i1 = 2 * ip + 1;
      if (i0 < nbw) {
        const bool dual = i1 < nbw;
        f32x4 X0[2], X1[2], Y0[2], Y1[2], Z0[2], Z1[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          X0[h] = lds_ld4(in, rd[0 + h] + 4096u * i0); X1[h] = lds_ld4(in, rd[0 + h] + 4096u * i1);
          Y0[h] = lds_ld4(in, rd[2 + h] + 4096u * i0); Y1[h] = lds_ld4(in, rd[2 + h] + 4096u * i1);
          Z0[h] = lds_ld4(in, rd[4 + h] + 4096u * i0); Z1[h] = lds_ld4(in, rd[4 + h] + 4096u * i1);
        }
        f32x4 acc0 = bias, acc1 = bias;
        __builtin_amdgcn_sched_barrier(0);
        if (dual) {
          mfma_tap<true, C32_KSTEPS>(af, 0, X0, X1, acc0, acc1);
          mfma_tap<true, C32_KSTEPS>(af, 1, Y0, Y1, acc0, acc1);
          mfma_tap<true, C32_KSTEPS>(af, 2, Z0, Z1, acc0, acc1);
        } else {
          mfma_tap<false, C32_KSTEPS>(af, 0, X0, X1, acc0, acc1);
          mfma_tap<false, C32_KSTEPS>(af, 1, Y0, Y1, acc0, acc1);
          mfma_tap<false, C32_KSTEPS>(af, 2, Z0, Z1, acc0, acc1);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (e == 1 && !dual) break;
          const int c = 16 * (cgp + 2 * (e ? i1 : i0)) + n16;
          if (c < 1 || c >= a.NC) continue;
          const uint32_t u = (uint32_t)(c - 1);
          const uint32_t r = a.dSc.div(u);
          const int l = (int)(u - r * (uint32_t)a.Sc);
          if (l >= a.L || b0 + r >= a.B) continue;
          const f32x4 acc = e ? acc1 : acc0;
          const size_t o = ((size_t)(b0 + r) * C32 + chv) * a.L + l;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v = acc[q];
            if (a.post_relu) v = fmaxf(v, 0.f);
            const size_t oq = o + (size_t)q * a.L;
            if (a.res1) v += a.res1[oq];
            if (a.res2) v += a.res2[oq];
            a.y[oq] = v;
            if (STAT == 1) {
              const float t = a.stat_relu ? fmaxf(v, 0.f) : v;
              st1[q] += t;
              st2[q] += t * t;
            } else if (STAT == 2) {
              float r = a.stat_x[oq];
              if (a.stat_relu) r = fmaxf(r, 0.f);
              st1[q] += v;
              st2[q] += v * ((r - sms[chv + q]) * sms[C32 + chv + q]);
            }
          }
        }
      }
    }
  }
  if (STAT) {
    // lanes of one kk group hold the same 4 channels for 16 different columns; the two waves of an M-block meet in LDS
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        st1[q] += __shfl_xor(st1[q], off);
        st2[q] += __shfl_xor(st2[q], off);
      }
    }
    __syncthreads();                                   // the image is dead
    float* red = img;                                  // [4 waves][2][16]
    if (n16 == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        red[wave * 32 + 4 * kk + q] = st1[q];
        red[wave * 32 + 16 + 4 * kk + q] = st2[q];
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int which = tid >> 5, ch = tid & 31, m = ch >> 4, j = ch & 15;
      const float t = red[m * 32 + which * 16 + j] + red[(m + 2) * 32 + which * 16 + j];
      atomicAdd(&a.stat_out[((size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 + which) * C32 + ch], (double)t);
    }
  }
}

// ------------------------------------------------------------------------------------------------- weight gradient
struct Wgrad32Args {
  const float* dy;       // [B][32][L]
  const float* x;        // [B][32][L] pre-activation saved by the forward
  const float* pre_s;    // input of the conv was pre_s * act(x) + pre_t
  const float* pre_t;
  int pre_relu;
  int B, L, R, Sc, NC, nb;
  FastDiv dL;
  float* part;           // [grid][32*32*3 + 32] per-workgroup partial sums
  int img_floats;        // floats of one LDS image
};

__global__ __launch_bounds__(256) void wgrad32_mfma_kernel(const Wgrad32Args a) {
  extern __shared__ __attribute__((aligned(16))) float wimg[];
  float* gimg = wimg;                 // dy tile
  float* aimg = wimg + a.img_floats;  // BN(act(x)) tile
  float* aff = aimg + a.img_floats;   // pre_s | pre_t
  if (a.pre_s && threadIdx.x < 2 * C32) aff[threadIdx.x] = threadIdx.x < C32 ? a.pre_s[threadIdx.x] : a.pre_t[threadIdx.x - C32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4;
  f32x4 acc[2][3][2];                 // [M-block][tap][cin half]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[m][t][h] = splat(0.f);
  float bacc = 0.f;
  const int nk = 4 * a.nb;            // k-steps of 4 columns
  const int k_lo = wave * nk / 4, k_hi = (wave + 1) * nk / 4;
  const int64_t ntiles = ((int64_t)a.B + a.R - 1) / a.R;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * a.R;
    __syncthreads();
    stage_rows(a.dy, b0, a.B, a.L, a.R, a.Sc, a.NC, a.nb, a.dL, nullptr, 0, gimg, tid);
    stage_rows(a.x, b0, a.B, a.L, a.R, a.Sc, a.NC, a.nb, a.dL, a.pre_s ? aff : nullptr, a.pre_relu, aimg, tid);
    __syncthreads();
    for (int s = k_lo; s < k_hi; ++s) {
      const int c = 4 * s + kk;       // logical column of this lane's k element
      float g[2], bv[3][2];
#pragma unroll
      for (int m = 0; m < 2; ++m) g[m] = gimg[lds_off(c + 1, (16 * m + n16) >> 2) + (n16 & 3)];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) bv[t][h] = aimg[lds_off(c + t, (16 * h + n16) >> 2) + (n16 & 3)];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[m][t][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[m], bv[t][h], acc[m][t][h], 0, 0, 0);
    }
    {   // bias gradient: thread = (cout tid/8, column residue tid%8)
      const int co = tid >> 3, p8 = tid & 7;
      float sum = 0.f;
      for (int c = p8; c < 16 * a.nb; c += 8) sum += gimg[lds_off(c + 1, co >> 2) + (co & 3)];
      bacc += sum;
    }
  }
  bacc += __shfl_xor(bacc, 1);
  bacc += __shfl_xor(bacc, 2);
  bacc += __shfl_xor(bacc, 4);
  // D[row = cout 4kk+r][col = cin n16] of tile (m, t, h) -> dW[16m + 4kk + r][16h + n16][t].  The four waves (column
  // quarters) meet in LDS in a fixed order; one partial row per workgroup goes to HBM.
  constexpr int NW = C32 * C32 * 3;
  __syncthreads();                    // the images are dead: reuse them as [4][NW + 32]
  float* mine = wimg + (size_t)wave * (NW + C32);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[((16 * m + 4 * kk + r) * C32 + 16 * h + n16) * 3 + t] = acc[m][t][h][r];
  // each wave holds the bias sums of its own 8 output channels (in lanes 0, 8, ..); the other 24 entries of its row are 0
  const float sv = __shfl(bacc, (lane & 7) * 8);
  if (lane < C32) mine[NW + lane] = ((lane >> 3) == wave) ? sv : 0.f;
  __syncthreads();
  float* dst = a.part + (size_t)blockIdx.x * (NW + C32);
  for (int i = tid; i < NW + C32; i += 256)
    dst[i] = (wimg[i] + wimg[(NW + C32) + i]) + (wimg[2 * (NW + C32) + i] + wimg[3 * (NW + C32) + i]);
}

// 64 outputs x 16 slices of the partial rows per workgroup; fixed summation order -> reproducible gradients
__global__ __launch_bounds__(1024) void part_reduce_kernel(const float* __restrict__ part, int nrow, int nW, int nB,
                                                           float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float sh[16][64];
  const int o = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s = 0.f;
  if (i < nW + nB)
    for (int b = slice; b < nrow; b += 16) s += part[(size_t)b * (nW + nB) + i];
  sh[slice][o] = s;
  __syncthreads();
  if (slice == 0 && i < nW + nB) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][o];
    if (i < nW) dW[i] = t;
    else if (db) db[i - nW] = t;
  }
}

static bool tile_geometry(int B, int L, int* R, int* Sc, int* NC, int* nb) {
  *Sc = L + 1;
  int r = (C32_MAXCOLS - 1) / *Sc;
  if (r < 1) return false;
  if (r > B) r = B;
  *R = r;
  *NC = 1 + r * *Sc;
  *nb = (*NC + 15) / 16;
  return *nb <= 2 * C32_NB2MAX;
}

}  // namespace mural

using namespace mural;

extern "C" int mural_op_conv32_supported(int32_t L) { return (L + 2) <= C32_MAXCOLS ? 1 : 0; }

// y = conv32(pre(x)) [+bias] [relu] [+res1 +res2]; W is the PyTorch [32][32][3] weight (dgrad != 0: input-gradient
// filter, i.e. y = dL/dx for x = dL/dy).  stat_mode / stat_*: per-channel sums of the output taken in the epilogue
// (see Conv32Args); stat_out: accumulator block double[MURAL_BN_SLOTS][2][32] zeroed by the caller.
extern "C" int mural_op_conv32(const float* x, const float* W, const float* bias, float* y, int64_t B, int32_t L, int32_t dgrad,
                               const float* pre_s, const float* pre_t, int32_t pre_relu, int32_t post_relu, const float* res1,
                               const float* res2, int32_t stat_mode, int32_t stat_relu, const float* stat_x,
                               const float* stat_mean, const float* stat_invstd, double* stat_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0 || L == 0) return MURAL_OK;
  Conv32Args a;
  std::memset(&a, 0, sizeof(a));
  MURAL_REQUIRE(tile_geometry((int)B, L, &a.R, &a.Sc, &a.NC, &a.nb), "conv32: L = %d does not fit the LDS tile", L);
  MURAL_REQUIRE(stat_mode == 0 || stat_out, "conv32: stat_mode %d needs stat_out", stat_mode);
  MURAL_REQUIRE(stat_mode != 2 || (stat_x && stat_mean && stat_invstd), "conv32: stat_mode 2 needs stat_x / mean / invstd");
  a.x = x; a.y = y; a.W = W; a.dgrad = dgrad; a.bias = bias; a.pre_s = pre_s; a.pre_t = pre_t; a.res1 = res1; a.res2 = res2;
  a.pre_relu = pre_relu; a.post_relu = post_relu; a.B = (int)B; a.L = L;
  a.stat_mode = stat_mode; a.stat_relu = stat_relu; a.stat_x = stat_x; a.stat_mean = stat_mean; a.stat_invstd = stat_invstd;
  a.stat_out = stat_out;
  a.dSc = FastDiv::make((uint32_t)a.Sc);
  a.dL = FastDiv::make((uint32_t)L);
  const size_t lds = (size_t)(16 * a.nb + 2) * C32 * 4 + 4 * C32 * 4;
  const int64_t ntiles = (B + a.R - 1) / a.R;
  const int grid = (int)(ntiles < 1024 ? ntiles : 1024);
  if (stat_mode == 0) hipLaunchKernelGGL(conv32_mfma_kernel<0>, dim3(grid), dim3(256), lds, stream, a);
  else if (stat_mode == 1) hipLaunchKernelGGL(conv32_mfma_kernel<1>, dim3(grid), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(conv32_mfma_kernel<2>, dim3(grid), dim3(256), lds, stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

// dW [32][32][3], db [32] of y = conv32(pre(x)); part: float scratch of at least mural_op_conv32_wgrad_scratch() floats
extern "C" size_t mural_op_conv32_wgrad_scratch(void) { return (size_t)512 * (C32 * C32 * 3 + C32); }

extern "C" int mural_op_conv32_wgrad(const float* dy, const float* x, int64_t B, int32_t L, const float* pre_s, const float* pre_t,
                                     int32_t pre_relu, float* dW, float* db, float* part, size_t part_floats, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0 || L == 0) return MURAL_OK;
  Wgrad32Args a;
  std::memset(&a, 0, sizeof(a));
  MURAL_REQUIRE(tile_geometry((int)B, L, &a.R, &a.Sc, &a.NC, &a.nb), "conv32_wgrad: L = %d does not fit the LDS tile", L);
  a.dy = dy; a.x = x; a.pre_s = pre_s; a.pre_t = pre_t; a.pre_relu = pre_relu; a.B = (int)B; a.L = L;
  a.dL = FastDiv::make((uint32_t)L);
  a.part = part;
  a.img_floats = (16 * a.nb + 2) * C32;
  const int64_t ntiles = (B + a.R - 1) / a.R;
  const int grid = (int)(ntiles < 512 ? ntiles : 512);
  MURAL_REQUIRE(part && part_floats >= (size_t)grid * (C32 * C32 * 3 + C32), "conv32_wgrad: partial-sum scratch too small");
  size_t lds = (size_t)2 * a.img_floats * 4 + 2 * C32 * 4;
  const size_t lds_red = (size_t)4 * (C32 * C32 * 3 + C32) * 4;     // the in-workgroup reduction reuses the image space
  lds = lds > lds_red ? lds : lds_red;
  static bool attr_set = false;
  if (!attr_set) {
    MURAL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad32_mfma_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad32_mfma_kernel, dim3(grid), dim3(256), lds, stream, a);
  hipLaunchKernelGGL(part_reduce_kernel, dim3((C32 * C32 * 3 + C32 + 63) / 64), dim3(1024), 0, stream, part, grid,
                     C32 * C32 * 3, C32, dW, db);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}
