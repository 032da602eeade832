// Microbenchmark (GPU box): an MFMA loop shaped like the inner loops of the product kernels -- 16 MFMAs (4 accumulator chains)
// per iteration whose A/B operands are (0) loop-invariant registers, (1) registers refreshed by v_mov each iteration,
// (2) refreshed from LDS by four ds_read_b128 issued one iteration ahead.  One or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE>
__global__ __launch_bounds__(256) void kmix(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 4 + 4096];
  for (int i = threadIdx.x; i < 256 * 16 + 4096; i += 256) lds[i] = seed + i * 1e-6f;
  __syncthreads();
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  const float* base = lds + threadIdx.x * 4;
  f32x4 a = *(const f32x4*)(base), a2 = *(const f32x4*)(base + 1024), p0 = *(const f32x4*)(base + 2048), p1 = *(const f32x4*)(base + 3072);
  for (int i = 0; i < iters; ++i) {
    f32x4 an = a, a2n = a2, p0n = p0, p1n = p1;
    if (MODE == 2) {
      const int o = (i & 3) * 16;
      an = *(const f32x4*)(base + o);
      a2n = *(const f32x4*)(base + 1024 + o);
      p0n = *(const f32x4*)(base + 2048 + o);
      p1n = *(const f32x4*)(base + 3072 + o);
    } else if (MODE == 1) {
      asm volatile("v_mov_b32 %0, %1" : "=v"(an.x) : "v"(a.y));
      asm volatile("v_mov_b32 %0, %1" : "=v"(a2n.x) : "v"(a2.y));
      asm volatile("v_mov_b32 %0, %1" : "=v"(p0n.x) : "v"(p0.y));
      asm volatile("v_mov_b32 %0, %1" : "=v"(p1n.x) : "v"(p1.y));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], p0[t], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], p1[t], acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[t], p0[t], acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[t], p1[t], acc3, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    a = an; a2 = a2n; p0 = p0n; p1 = p1n;
  }
  f32x4 s = acc0 + acc1 + acc2 + acc3;
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

// The same FLOPs per iteration (32768) as 8 x v_mfma_f32_32x32x2_f32 on two 32-column accumulator tiles: a wave that owns all 32
// output channels reads each B value once instead of once per 16-channel M-block.  MODE 3: A fragments in registers, B from LDS
// (two ds_read_b128 per 8 MFMAs); MODE 4: A from LDS as well (four reads).
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int MODE>
__global__ __launch_bounds__(256) void kbig(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 4 + 4096];
  for (int i = threadIdx.x; i < 256 * 16 + 4096; i += 256) lds[i] = seed + i * 1e-6f;
  __syncthreads();
  f32x16 acc0 = {0}, acc1 = {0};
  const float* base = lds + threadIdx.x * 4;
  f32x4 a = *(const f32x4*)(base), a2 = *(const f32x4*)(base + 1024), p0 = *(const f32x4*)(base + 2048), p1 = *(const f32x4*)(base + 3072);
  for (int i = 0; i < iters; ++i) {
    f32x4 an = a, a2n = a2, p0n, p1n;
    const int o = (i & 3) * 16;
    p0n = *(const f32x4*)(base + 2048 + o);
    p1n = *(const f32x4*)(base + 3072 + o);
    if (MODE == 4) {
      an = *(const f32x4*)(base + o);
      a2n = *(const f32x4*)(base + 1024 + o);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(t & 1 ? a2[t] : a[t], p0[t], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(t & 1 ? a2[t] : a[t], p1[t], acc1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    a = an; a2 = a2n; p0 = p0n; p1 = p1n;
  }
  float s = 0.f;
  for (int q = 0; q < 16; ++q) s += acc0[q] + acc1[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int wgs_per_cu, float* out) {
  const int iters = 2000, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)grid * 4 * iters * 16;     // in 2048-FLOP units: 8 of the 32x32x2 instructions count as 16
  printf("%-34s waves/SIMD %d : %6.1f TFLOP/s, %5.1f cycles per MFMA per SIMD\n", name, wgs_per_cu, mfmas * 2048 / (ms * 1e-3) / 1e12,
         (ms * 1e-3 * 2.4e9) / (mfmas / 1024.0));
}

int main() {
  float* out;
  (void)hipMalloc(&out, sizeof(float) * 256 * 256 * 4);
  for (int w : {1, 2}) {
    run("invariant operands", kmix<0>, w, out);
    run("operands refreshed by v_mov", kmix<1>, w, out);
    run("operands from LDS (b128, 1 ahead)", kmix<2>, w, out);
    run("32x32x2: B from LDS, A in registers", kbig<3>, w, out);
    run("32x32x2: A and B from LDS", kbig<4>, w, out);
  }
  return 0;
}
