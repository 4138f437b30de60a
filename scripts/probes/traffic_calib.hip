// Calibration for the gfx950 FETCH_SIZE / WRITE_SIZE counters: kernels that read (or write) a KNOWN
// byte count (1 GiB, larger than the 256 MiB Infinity Cache) with the access widths our kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename T>
__global__ void read_k(const T* __restrict__ p, size_t n, unsigned* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < n; i += stride) { T v = p[i]; const unsigned* w = (const unsigned*)&v; for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k]; }
  if (acc == 0x12345678u) out[0] = acc;
}
template <typename T>
__global__ void write_k(T* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  T v; unsigned* w = (unsigned*)&v; for (unsigned k = 0; k < sizeof(T) / 4; ++k) w[k] = (unsigned)i + k;
  for (; i < n; i += stride) p[i] = v;
}
// the fwd_xy access shape: each lane reads 18 consecutive dwords, lanes 48 bytes apart
__global__ void read_seg48(const unsigned* __restrict__ p, size_t nseg, unsigned* out) {
  size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  if (s < nseg) { const unsigned* q = p + s * 12; for (int k = 0; k < 18; ++k) acc ^= q[k]; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void write_short(short* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = (short)i;
}
int main() {
  const size_t bytes = 1ull << 30;
  void* buf; unsigned* out; hipMalloc(&buf, bytes + 4096); hipMalloc(&out, 64);
  hipMemset(buf, 1, bytes);
  hipLaunchKernelGGL(read_k<unsigned>, dim3(8192), dim3(256), 0, 0, (const unsigned*)buf, bytes / 4, out);
  hipLaunchKernelGGL(read_k<uint2>, dim3(8192), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, out);
  hipLaunchKernelGGL(read_k<uint4>, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
  hipLaunchKernelGGL(read_seg48, dim3((unsigned)((bytes / 48 - 2 + 255) / 256)), dim3(256), 0, 0, (const unsigned*)buf, bytes / 48 - 2, out);
  hipLaunchKernelGGL(write_k<unsigned>, dim3(8192), dim3(256), 0, 0, (unsigned*)buf, bytes / 4);
  hipLaunchKernelGGL(write_k<uint4>, dim3(8192), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
  hipLaunchKernelGGL(write_short, dim3(8192), dim3(256), 0, 0, (short*)buf, bytes / 2);
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
