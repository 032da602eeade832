// Microbenchmark (GPU box): sustained rate of v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32 as a function of the number of
// independent accumulator chains per wave and of the waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int CH>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-7f, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CH; ++c) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
  float a = a0 + threadIdx.x * 1e-7f, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CH; ++c)
    for (int q = 0; q < 16; ++q) s += acc[c][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int chains, double flop_per_mfma, int wgs_per_cu, float* out) {
  const int iters = 4000, cus = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = cus * wgs_per_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)grid * 4 /*waves*/ * iters * 8 * chains;
  printf("%-10s chains %d  waves/SIMD %d : %7.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, chains, wgs_per_cu,
         mfmas * flop_per_mfma / (ms * 1e-3) / 1e12, (ms * 1e-3 * 2.4e9) / (mfmas / (cus * 4.0)));
}

int main() {
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
  for (int w : {1, 2, 4}) {
    run("16x16x4", k16<1>, 1, 2048, w, out);
    run("16x16x4", k16<2>, 2, 2048, w, out);
    run("16x16x4", k16<4>, 4, 2048, w, out);
    run("16x16x4", k16<8>, 8, 2048, w, out);
    run("32x32x2", k32<1>, 1, 4096, w, out);
    run("32x32x2", k32<2>, 2, 4096, w, out);
    run("32x32x2", k32<4>, 4, 4096, w, out);
  }
  return 0;
}
