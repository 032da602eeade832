// Microbenchmark (GPU box): what does a wave of vector-ALU / LDS work cost the MFMA wave it shares a SIMD with?
// One 8-wave workgroup per CU (LDS-limited); waves 0-3 run an MFMA loop (two accumulator chains, v_mfma_f32_16x16x4_f32),
// waves 4-7 a partner loop chosen by MODE:  0 nothing, 1 independent v_fma chains, 2 one dependent v_fma chain, 3 LDS read/write
// (ds_read_b128 x3 + max + ds_write_b128, like the pooling phase), 4 MFMA as well, 5 integer address arithmetic (v_mad_u32, shifts).
// Prints the MFMA waves' rate (MFMA per 32 cycles of wall clock would be 1.0 at the nominal 2.4 GHz) and the partner's rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NOP>
__device__ __forceinline__ void pace(float& y0, float& y1, float& y2, float& y3, float& y4, float& y5, float& y6, float& y7) {
  __builtin_amdgcn_sched_barrier(0);
  if (NOP >= 11) {
    const float m = 1.0001f, c = 0.5f;
    if (NOP - 10 >= 1) y0 = __builtin_fmaf(y0, m, c);
    if (NOP - 10 >= 2) y1 = __builtin_fmaf(y1, m, c);
    if (NOP - 10 >= 3) y2 = __builtin_fmaf(y2, m, c);
    if (NOP - 10 >= 4) y3 = __builtin_fmaf(y3, m, c);
    if (NOP - 10 >= 5) y4 = __builtin_fmaf(y4, m, c);
    if (NOP - 10 >= 6) y5 = __builtin_fmaf(y5, m, c);
    if (NOP - 10 >= 7) y6 = __builtin_fmaf(y6, m, c);
    if (NOP - 10 >= 8) y7 = __builtin_fmaf(y7, m, c);
  }
  if (NOP == 1) asm volatile("s_nop 4");
  if (NOP == 2) asm volatile("s_nop 5");
  if (NOP == 3) asm volatile("s_nop 6");
  if (NOP == 4) asm volatile("s_nop 7");
  if (NOP == 5) asm volatile("s_nop 3");
  __builtin_amdgcn_sched_barrier(0);
}

template <int MODE, int NOP, int SWAP, int PRIO>
__global__ __launch_bounds__(512) void kco(float* out, unsigned long long* times, int iters, int piters, float seed) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave_hw = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wave = SWAP ? (wave_hw ^ 4) : wave_hw;      // role index: 0-3 MFMA, 4-7 partner
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = seed + i * 1e-6f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float res = 0.f;
  if (wave >= 4 && PRIO) __builtin_amdgcn_s_setprio(PRIO);
  if (wave < 4) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
    const float a = seed + lane, b = seed - lane;
    float y0 = seed, y1 = seed + 1, y2 = seed + 2, y3 = seed + 3, y4 = seed + 4, y5 = seed + 5, y6 = seed + 6, y7 = seed + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
        pace<NOP>(y0, y1, y2, y3, y4, y5, y6, y7);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
        pace<NOP>(y0, y1, y2, y3, y4, y5, y6, y7);
      }
    }
    res = acc0.x + acc1.y + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
  } else if (MODE == 1) {
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
    const float m = 1.0001f, c = 0.5f;
    for (int i = 0; i < piters; ++i) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        x0 = __builtin_fmaf(x0, m, c); x1 = __builtin_fmaf(x1, m, c); x2 = __builtin_fmaf(x2, m, c); x3 = __builtin_fmaf(x3, m, c);
        x4 = __builtin_fmaf(x4, m, c); x5 = __builtin_fmaf(x5, m, c); x6 = __builtin_fmaf(x6, m, c); x7 = __builtin_fmaf(x7, m, c);
      }
    }
    res = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  } else if (MODE == 2) {
    float x0 = seed;
    const float m = 1.0001f, c = 0.5f;
    for (int i = 0; i < piters; ++i) {
#pragma unroll
      for (int t = 0; t < 32; ++t) x0 = __builtin_fmaf(x0, m, c);
    }
    res = x0;
  } else if (MODE == 3) {
    float* base = lds + (wave - 4) * 2048 + lane * 4;
    f32x4 m = {0, 0, 0, 0};
    for (int i = 0; i < piters; ++i) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const f32x4 u = *(const f32x4*)(base + ((t * 256) & 1023)), v = *(const f32x4*)(base + ((t * 256 + 256) & 1023)),
                    w = *(const f32x4*)(base + ((t * 256 + 512) & 1023));
        m = __builtin_elementwise_max(__builtin_elementwise_max(u, v), __builtin_elementwise_max(w, m));
        *(f32x4*)(base + 1024 + ((t * 256) & 1023)) = m;
      }
    }
    res = m.x + m.y + m.z + m.w;
  } else if (MODE == 4) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
    const float a = seed + lane, b = seed - lane;
    for (int i = 0; i < piters; ++i) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
      }
    }
    res = acc0.x + acc1.y;
  } else if (MODE == 6) {
    unsigned s0 = __builtin_amdgcn_readfirstlane(lane), s1 = 3;
    for (int i = 0; i < piters; ++i) {
#pragma unroll
      for (int t = 0; t < 32; ++t) { s0 = s0 * 3u + s1; s1 = s1 ^ (s0 >> 3); }
    }
    res = (float)(s0 + s1);
  } else if (MODE == 5) {
    unsigned x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
    for (int i = 0; i < piters; ++i) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        x0 = x0 * 3u + (x1 >> 3); x1 = (x1 ^ x2) + 7u; x2 = x2 * 5u + (x3 & 255u); x3 = (x3 << 1) ^ x0;
      }
    }
    res = (float)(x0 + x1 + x2 + x3);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    times[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
    times[(blockIdx.x * 8 + wave) * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int MODE, int NOP = 0, int SWAP = 0, int PRIO = 0>
void run(const char* name, int iters, int piters, double partner_ops_per_iter) {
  float* out;
  unsigned long long* times;
  const int grid = 256;
  hipMalloc(&out, grid * 512 * 4);
  hipMalloc(&times, grid * 16 * 8);
  hipFuncSetAttribute((const void*)(kco<MODE, NOP, SWAP, PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kco<MODE, NOP, SWAP, PRIO>), dim3(grid), dim3(512), 100 * 1024, 0, out, times, iters, piters, 1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 16);
  hipMemcpy(h.data(), times, grid * 16 * 8, hipMemcpyDeviceToHost);
  double tm = 0, tp = 0;
  int same = 0;
  for (int b = 0; b < grid; ++b) {
    for (int w = 0; w < 4; ++w) tm += (double)h[(b * 8 + w) * 2];
    for (int w = 4; w < 8; ++w) tp += (double)h[(b * 8 + w) * 2];
    for (int w = 0; w < 4; ++w)
      if (((h[(b * 8 + w) * 2 + 1] >> 4) & 3) == ((h[(b * 8 + w + 4) * 2 + 1] >> 4) & 3)) ++same;
  }
  tm /= grid * 4; tp /= grid * 4;      // 100 MHz ticks
  const double mf = iters * 16.0;
  printf("%-28s mfma wave: %8.1f us, %.3f MFMA per 32 cycles @2.4GHz | partner: %8.1f us, %.2f ops per 4 cycles @2.4GHz | simd shared %d/1024\n",
         name, tm / 100, mf * 32 / (tm / 100 * 2400), tp / 100, partner_ops_per_iter * piters * 4 / (tp / 100 * 2400), same);
  hipFree(out); hipFree(times);
}

int main() {
  const int it = 40000;
  run<0>("partner idle", it, 0, 0);
  run<0, 12>("idle, mfma + 2 own fma", it, 0, 0);
  run<0, 14>("idle, mfma + 4 own fma", it, 0, 0);
  run<0, 16>("idle, mfma + 6 own fma", it, 0, 0);
  run<0, 17>("idle, mfma + 7 own fma", it, 0, 0);
  run<0, 18>("idle, mfma + 8 own fma", it, 0, 0);
  run<1, 14>("fma partner, mfma + 4 own fma", it, 40000, 32);
  run<6>("salu partner", it, 40000, 64);
  run<6>("salu (partner only)", 0, 40000, 64);
  return 0;
}
