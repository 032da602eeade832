// Microbenchmark (gfx950): workgroup dispatch rate.  Near-empty kernels of 256-thread workgroups, 65536 of them, with and without
// static LDS and with a small / large register footprint: the launch time / 65536 is the cost of getting a workgroup on and off a CU.
// build: hipcc -O3 --offload-arch=gfx950 -o dispatch_rate dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_FLOATS, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int n) {
  __shared__ float s[LDS_FLOATS > 0 ? LDS_FLOATS : 1];
  if (LDS_FLOATS > 0) s[threadIdx.x] = (float)threadIdx.x;
  if (n == 12345) out[threadIdx.x] = s[(threadIdx.x * 7) % (LDS_FLOATS > 0 ? LDS_FLOATS : 1)];
}
template <int LDS_FLOATS, int THREADS>
static void run(const char* name, int wgs, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<LDS_FLOATS, THREADS>), dim3(wgs), dim3(THREADS), 0, 0, d, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<LDS_FLOATS, THREADS>), dim3(wgs), dim3(THREADS), 0, 0, d, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %6d workgroups: %8.1f us per launch = %6.2f ns per workgroup (%.0f per us)\n", name, wgs, ms * 1e3f / 5, ms * 1e6f / 5 / wgs,
         wgs / (ms * 1e3f / 5));
}
int main() {
  float* d;
  hipMalloc(&d, 1 << 20);
  run<0, 256>("256 threads, no LDS", 65536, d);
  run<4096, 256>("256 threads, 16 KB LDS", 65536, d);
  run<5760, 256>("256 threads, 22.5 KB LDS", 65536, d);
  run<0, 64>("64 threads, no LDS", 65536, d);
  run<0, 64>("64 threads, no LDS", 262144, d);
  run<0, 512>("512 threads, no LDS", 32768, d);
  run<0, 1024>("1024 threads, no LDS", 16384, d);
  run<4096, 256>("256 threads, 16 KB LDS", 16384, d);
  return 0;
}
