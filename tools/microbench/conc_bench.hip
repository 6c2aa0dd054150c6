// do kernels on two streams run side by side on this box?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void spin_kernel(unsigned long long ticks, unsigned *out) {
  const unsigned long long t0 = wall_clock64();
  unsigned k = 0;
  while (wall_clock64() - t0 < ticks) ++k;
  if (threadIdx.x == 0 && out) out[blockIdx.x] = k;
}
__global__ __launch_bounds__(256) void export_kernel(u32x4 *__restrict__ dst, const u32x4 *__restrict__ src, size_t n16) {
  const size_t step = (size_t)gridDim.x * 256u;
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += step) dst[i] = src[i];
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = 1300u << 10, n16 = bytes / 16;
  char *h, *d; unsigned *o;
  CK(hipHostMalloc((void **)&h, bytes, hipHostMallocDefault));
  CK(hipMalloc((void **)&d, bytes));
  CK(hipMalloc((void **)&o, 4096 * 4));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const unsigned long long ticks = 4000;  // wall_clock64 runs at 100 MHz: 40 us
  for (int w = 0; w < 200; ++w) { hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(256), 0, sa, ticks, o); hipLaunchKernelGGL(export_kernel, dim3(32), dim3(256), 0, sb, (u32x4 *)h, (const u32x4 *)d, n16); }
  CK(hipDeviceSynchronize());
  auto run = [&](const char *name, int a, int b, int wgs_b) -> int {
    // a: 0 none, 1 spin(32 wg) on A, 2 export on A ; b: 0 none, 1 spin(wgs_b) on B x3 kernels
    double sum = 0; const int reps = 100;
    for (int r = 0; r < reps; ++r) {
      CK(hipDeviceSynchronize());
      const double t0 = now_us();
      if (a == 1) hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(256), 0, sa, ticks, o);
      if (a == 2) hipLaunchKernelGGL(export_kernel, dim3(32), dim3(256), 0, sa, (u32x4 *)h, (const u32x4 *)d, n16);
      if (b) for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(spin_kernel, dim3(wgs_b), dim3(256), 0, sb, ticks / 4, o + 2048);
      CK(hipDeviceSynchronize());
      sum += now_us() - t0;
    }
    printf("%-60s %7.1f us\n", name, sum / reps);
    return 0;
  };
  if (run("A: spin 40 us (32 wg)", 1, 0, 0)) return 1;
  if (run("A: export 1.3 MB (32 wg)", 2, 0, 0)) return 1;
  if (run("B: 3 x spin 10 us (32 wg)", 0, 1, 32)) return 1;
  if (run("B: 3 x spin 10 us (1250 wg)", 0, 1, 1250)) return 1;
  if (run("A spin 40 | B 3 x spin 10 (32 wg)", 1, 1, 32)) return 1;
  if (run("A spin 40 | B 3 x spin 10 (1250 wg)", 1, 1, 1250)) return 1;
  if (run("A export | B 3 x spin 10 (32 wg)", 2, 1, 32)) return 1;
  if (run("A export | B 3 x spin 10 (1250 wg)", 2, 1, 1250)) return 1;
  return 0;
}
