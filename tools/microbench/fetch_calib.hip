// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the read patterns of the product kernels (MI355X_MICROARCH.md, HBM: the counter
// reports half the bytes of a wide coalesced 16-byte-per-lane streaming read; "other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern").  Every kernel below reads a 1 GiB buffer (4 x the Infinity Cache, written by a fill
// kernel right before) exactly once and is named after its pattern; run as
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- tools/microbench/fetch_calib
// and divide the buffer size by each kernel's FETCH_SIZE x 1024 (tools/r5_fetch_calib.sh does both).
//   wide16     lane l reads the aligned 16-byte piece l of its wave's 1 KB (global_load_dwordx4): the guide's pattern
//   dword      lane l reads dword l of its wave's 256 bytes (the row loads of convblock_deep.hip, the BatchNorm maps' tails)
//   bufwide16  wide16 through a raw buffer descriptor (indel_level0.hip's requests)
//   quads4     the B operand of the barrier-free convs (conv1d_direct.hip, convblock_mfma.hip): lane (n, kk) reads the UNALIGNED 16-byte
//              quad x[row 4 j + kk][p + n .. p + n + 3] -- four rows per wave instruction, 76 contiguous bytes of each, the quads of
//              neighbouring lanes overlapping by 12 bytes; a wave walks its rows 16 positions at a time
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void fill(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)(i & 1023) * 1e-3f;
}

__global__ __launch_bounds__(256) void wide16(const float* __restrict__ p, size_t n4, float* out) {
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) s += reinterpret_cast<const f32x4*>(p)[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void dword(const float* __restrict__ p, size_t n, float* out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += p[i];
  if (s == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void bufwide16(const float* __restrict__ p, size_t n4, float* out) {
  // (descriptors address 4 GB: one per 1 GiB buffer is enough)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)0x7fffffff, 0x00020000);
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    s += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(i * 16), 0, 0));
  }
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}

// rows of L floats; a wave owns four rows and walks them 16 positions at a time (the last quads of a row run 3 floats into the next row)
__global__ __launch_bounds__(256) void quads4(const float* __restrict__ p, size_t rows, int L, float* out) {
  const int lane = threadIdx.x & 63, n = lane & 15, kk = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
  f32x4 s = {0, 0, 0, 0};
  for (size_t g = wave; g < rows / 4; g += nwaves) {
    const float* row = p + (4 * g + kk) * (size_t)L;
    for (int c = 0; c + 16 <= L; c += 16) {
      f32x4 v;
      __builtin_memcpy(&v, row + c + n, 16);      // unaligned 16-byte load
      s += v;
    }
  }
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}

int main() {
  const size_t bytes = (size_t)1 << 30, n = bytes / 4;
  float *p, *out;
  CK(hipMalloc(&p, bytes + 4096));
  CK(hipMalloc(&out, 4096));
  const int L = 2048;      // (row length of the quads4 walk: 512 K rows)
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, p, n + 1024);
    hipLaunchKernelGGL(wide16, dim3(2048), dim3(256), 0, 0, p, n / 4, out);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, p, n + 1024);
    hipLaunchKernelGGL(dword, dim3(2048), dim3(256), 0, 0, p, n, out);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, p, n + 1024);
    hipLaunchKernelGGL(bufwide16, dim3(2048), dim3(256), 0, 0, p, n / 4, out);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, p, n + 1024);
    hipLaunchKernelGGL(quads4, dim3(2048), dim3(256), 0, 0, p, n / L, L, out);
  }
  CK(hipDeviceSynchronize());
  printf("buffer bytes %zu\n", bytes);
  return 0;
}
