// Microbenchmark (GPU box): cost of vector-ALU work between MFMAs.  Each wave repeats [G x v_mfma_f32_16x16x4_f32 on two alternating
// accumulator chains][K x independent v_fma_f32]; one workgroup per CU (LDS-limited) of 4 waves (one per SIMD) or 8 (two per SIMD).
// Prints cycles per iteration at the nominal 2.4 GHz next to the 32 G cycles the MFMAs alone need.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int G, int K, int KIND>
__global__ __launch_bounds__(512) void kmv(float* out, unsigned long long* times, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = seed + i * 1e-6f;
  __syncthreads();
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
  const float a = seed + lane, b = seed - lane;
  float y[8];
  for (int j = 0; j < 8; ++j) y[j] = seed + j;
  const float m = 1.0001f, c = 0.5f;
  float* base = lds + (threadIdx.x >> 6) * 1024 + lane * 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float vm = m + lane * 1e-9f, vc = c + lane * 1e-9f;
  asm volatile("" : "+v"(vm), "+v"(vc));
  for (int i = 0; i < iters; i += 4) {
#pragma unroll
   for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (KIND == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) y[k & 7] = __builtin_fmaf(y[k & 7], m, c);
    } else if (KIND == 1) {      // K LDS 16-byte stores
#pragma unroll
      for (int k = 0; k < K; ++k) *(f32x4*)(base + 256 * (k & 3)) = f32x4{y[0], y[1], y[2], y[3]};
    } else if (KIND == 2) {      // K LDS 16-byte loads (consumed one iteration later)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const f32x4 v = *(const f32x4*)(base + 256 * (k & 3));
        y[k & 7] += v.x;
      }
    } else if (KIND == 3) {      // K s_nop 0 (4 cycles each)
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("s_nop 0");
    } else if (KIND == 4) {      // v_fma with vector operands only
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[k & 7]) : "v"(vm), "v"(vc));
    } else if (KIND == 5) {      // v_max
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(y[k & 7]) : "v"(vc));
    } else if (KIND == 6) {      // v_mov
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("v_mov_b32 %0, %1" : "=v"(y[k & 7]) : "v"(vc));
    } else if (KIND == 7) {      // s_mov (scalar ALU)
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("s_mov_b32 s20, s21" ::: "s20");
    }
    __builtin_amdgcn_sched_barrier(0);
   }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) times[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  float s = acc0.x + acc1.y;
  for (int j = 0; j < 8; ++j) s += y[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int G, int K, int KIND = 0>
void run(int waves) {
  float* out;
  unsigned long long* times;
  const int grid = 256, iters = 200000 / G;
  hipMalloc(&out, grid * 512 * 4);
  hipMalloc(&times, grid * 8 * 8);
  hipFuncSetAttribute((const void*)(kmv<G, K, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kmv<G, K, KIND>), dim3(grid), dim3(64 * waves), 100 * 1024, 0, out, times, iters, 1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 8);
  hipMemcpy(h.data(), times, grid * 8 * 8, hipMemcpyDeviceToHost);
  double t = 0;
  for (int b = 0; b < grid; ++b) {
    double mx = 0;
    for (int w = 0; w < waves; ++w) mx = mx > (double)h[b * 8 + w] ? mx : (double)h[b * 8 + w];
    t += mx;
  }
  t /= grid;      // 100 MHz ticks: the last wave of a workgroup
  const double cyc = t / 100 * 2400 / iters;      // cycles per iteration of one wave
  const double per_simd = cyc / (waves / 4);      // the SIMD retires (waves / 4) iterations in that time
  static const char* kinds[] = {"v_fma", "ds_write_b128", "ds_read_b128", "s_nop 0", "v_fma vgpr", "v_max", "v_mov", "s_mov"};
  printf("G=%2d K=%2d %-14s waves/SIMD=%d: %7.1f cycles per iteration per SIMD (MFMA alone %4d) -> extra %6.1f = %5.2f per op; MFMA pipe %.3f\n", G, K,
         kinds[KIND], waves / 4, per_simd, 32 * G, per_simd - 32 * G, K ? (per_simd - 32 * G) / K : 0.0, 32 * G / per_simd);
  hipFree(out); hipFree(times);
}

int main() {
  for (int waves = 4; waves <= 8; waves += 4) {
    run<2, 0>(waves); run<8, 0>(waves);
    run<2, 1>(waves); run<2, 2>(waves); run<2, 4>(waves);
    run<2, 1, 4>(waves); run<2, 2, 4>(waves); run<2, 4, 4>(waves);
    run<2, 1, 5>(waves); run<2, 2, 5>(waves);
    run<2, 1, 6>(waves); run<2, 2, 6>(waves);
    run<2, 1, 7>(waves); run<2, 4, 7>(waves);
    run<2, 1, 3>(waves); run<2, 2, 3>(waves); run<2, 4, 3>(waves); run<2, 6, 3>(waves);
    run<8, 8, 4>(waves); run<8, 16, 4>(waves); run<16, 32, 4>(waves); run<48, 64, 4>(waves);
    run<2, 1, 1>(waves); run<8, 4, 1>(waves); run<2, 1, 2>(waves); run<8, 4, 2>(waves);
  }
  return 0;
}
