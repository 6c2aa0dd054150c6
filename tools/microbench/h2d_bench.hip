// micro-benchmark: pinned host -> device copies of a chunk's size, back to back on one stream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void pull_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) dst[i] = src[i];
}
int main(int argc, char **argv) {
  const size_t maxb = 8u << 20;
  char *h; char *d;
  CK(hipHostMalloc((void **)&h, maxb, hipHostMallocDefault));
  CK(hipMalloc((void **)&d, maxb));
  memset(h, 1, maxb);
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t sizes[] = {64u << 10, 144u << 10, 288u << 10, 575u << 10, 1150u << 10, 2300u << 10, 4600u << 10};
  for (size_t sz : sizes) {
    for (int w = 0; w < 20; ++w) CK(hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    const int reps = 200;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("memcpyAsync H2D %8zu B: %7.2f us  %6.2f GB/s\n", sz, us, sz / us / 1e3);
    for (int wg : {32, 128, 512, 2048}) {
      for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(pull_kernel, dim3(wg), dim3(256), 0, st, (const uint4 *)h, (uint4 *)d, sz / 16);
      CK(hipStreamSynchronize(st));
      t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(pull_kernel, dim3(wg), dim3(256), 0, st, (const uint4 *)h, (uint4 *)d, sz / 16);
      CK(hipStreamSynchronize(st));
      us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
      printf("  pull kernel %4d wg      : %7.2f us  %6.2f GB/s\n", wg, us, sz / us / 1e3);
    }
    // D2H too
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("  memcpyAsync D2H         : %7.2f us  %6.2f GB/s\n", us, sz / us / 1e3);
  }
  return 0;
}
