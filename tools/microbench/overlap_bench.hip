// does a kernel that writes pinned host memory stretch the kernels running beside it on another stream?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void export_kernel(u32x4 *__restrict__ dst, const u32x4 *__restrict__ src, size_t n16, int nt) {
  const size_t step = (size_t)gridDim.x * 256u;
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += step) {
    if (nt) __builtin_nontemporal_store(src[i], dst + i);
    else dst[i] = src[i];
  }
}
// ~10 us of work on 1250 workgroups: a few dependent passes over a private 64 KB stretch of HBM
__global__ __launch_bounds__(256) void work_kernel(unsigned *__restrict__ buf, int rounds) {
  unsigned *p = buf + (size_t)blockIdx.x * 16384;
  unsigned acc = threadIdx.x;
  for (int r = 0; r < rounds; ++r)
    for (int k = threadIdx.x; k < 16384; k += 256) { acc = acc * 1664525u + p[k]; p[k] = acc; }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = 1300u << 10, n16 = bytes / 16;
  char *h_coh, *h_nc, *d_src; unsigned *d_work;
  CK(hipHostMalloc((void **)&h_coh, bytes, hipHostMallocDefault));
  CK(hipHostMalloc((void **)&h_nc, bytes, hipHostMallocNonCoherent));
  CK(hipMalloc((void **)&d_src, bytes));
  CK(hipMalloc((void **)&d_work, 1250u * 16384 * 4));
  CK(hipMemset(d_src, 1, bytes));
  CK(hipMemset(d_work, 1, 1250u * 16384 * 4));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  int rounds = 3;
  auto run = [&](const char *name, int mode, int nb) -> int {
    // mode 0: work kernels alone; 1: export to coherent; 2: export to non-coherent; 3: nt stores coherent; 4: hipMemcpyAsync D2H; 5: export alone
    double best = 1e30, sum = 0;
    const int reps = 60;
    for (int r = 0; r < reps + 10; ++r) {
      CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
      const double t0 = now_us();
      if (mode == 1 || mode == 5) hipLaunchKernelGGL(export_kernel, dim3(32), dim3(256), 0, sa, (u32x4 *)h_coh, (const u32x4 *)d_src, n16, 0);
      if (mode == 2) hipLaunchKernelGGL(export_kernel, dim3(32), dim3(256), 0, sa, (u32x4 *)h_nc, (const u32x4 *)d_src, n16, 0);
      if (mode == 3) hipLaunchKernelGGL(export_kernel, dim3(32), dim3(256), 0, sa, (u32x4 *)h_coh, (const u32x4 *)d_src, n16, 1);
      if (mode == 4) CK(hipMemcpyAsync(h_coh, d_src, bytes, hipMemcpyDeviceToHost, sa));
      if (mode != 5) for (int k = 0; k < nb; ++k) hipLaunchKernelGGL(work_kernel, dim3(1250), dim3(256), 0, sb, d_work, rounds);
      CK(hipStreamSynchronize(sb));
      const double t1 = now_us();
      CK(hipStreamSynchronize(sa));
      if (r >= 10) { best = std::min(best, t1 - t0); sum += t1 - t0; }
    }
    printf("%-44s x%d work kernels: stream B done after %7.1f us avg, %7.1f best\n", name, nb, sum / reps, best);
    return 0;
  };
  for (int nb : {1, 4, 8}) {
    if (run("work alone", 0, nb)) return 1;
    if (run("beside export kernel -> coherent pinned", 1, nb)) return 1;
    if (run("beside export kernel -> non-coherent pinned", 2, nb)) return 1;
    if (run("beside export kernel, nontemporal stores", 3, nb)) return 1;
    if (run("beside hipMemcpyAsync D2H", 4, nb)) return 1;
  }
  if (run("export kernel alone (B idle)", 5, 0)) return 1;
  return 0;
}
