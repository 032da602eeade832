// Microbenchmark (gfx950): do v_mfma_f32_16x16x4_f32 and v_pk_fma_f32 / v_fma_f32 / v_exp_f32 from DIFFERENT waves of a SIMD overlap?
// Every workgroup has 8 waves (2 per SIMD); mode 0: all waves run the MFMA loop, 1: all run the VALU loop, 2: even waves MFMA, odd
// waves VALU (same per-wave work as in 0 / 1 -> time = max if the pipes are independent, ~sum/2.. if they share hardware).
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VKIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || (mode == 2 && (wave & 1) == 0);
  float r = 0.f;
  if (do_mfma) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
      }
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    // 16 MFMAs of 32 cycles = 512 cycles per iteration on the matrix side; the VALU side: 128 instructions per iteration
    if (VKIND == 0) {
      f32x2 c[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = f32x2{(float)j, 1.f};
      const f32x2 x = {1.0001f, 0.9999f}, y = {1e-3f, 2e-3f};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) c[j] = __builtin_elementwise_fma(c[j], x, y);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) r += c[j].x + c[j].y;
    } else if (VKIND == 1) {
      float c[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = (float)j;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) c[j] = __builtin_fmaf(c[j], 1.0001f, 1e-3f);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) r += c[j];
    } else {
      float c[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = (float)j * 0.1f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_exp2f(c[j]) * 0.25f;      // v_exp_f32 + v_mul
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) r += c[j];
    }
  }
  if (r == 123.456f) out[threadIdx.x] = r;
}

template <int VKIND>
static float run(int mode, int iters, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<VKIND>, dim3(256 * 4), dim3(512), 0, 0, d, iters, mode);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<VKIND>, dim3(256 * 4), dim3(512), 0, 0, d, iters, mode);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  const int iters = 2000;
  const char* names[3] = {"v_pk_fma_f32", "v_fma_f32", "v_exp_f32+v_mul_f32"};
  // 1024 workgroups of 8 waves on 256 CUs: 4 workgroups per CU in sequence or together (register-light: all resident)
  for (int vk = 0; vk < 3; ++vk) {
    float t[3];
    for (int mode = 0; mode < 3; ++mode) t[mode] = vk == 0 ? run<0>(mode, iters, d) : (vk == 1 ? run<1>(mode, iters, d) : run<2>(mode, iters, d));
    printf("%-22s all-MFMA %8.1f us   all-VALU %8.1f us   half/half %8.1f us   (independent pipes: %.1f, shared: %.1f)\n", names[vk], t[0], t[1],
           t[2], (t[0] > t[1] ? t[0] : t[1]) / 2, (t[0] + t[1]) / 2);
  }
  // per-instruction cost: waves per SIMD = 1024 * 8 / 1024 = 8; MFMAs per wave = 16 * iters
  return 0;
}
