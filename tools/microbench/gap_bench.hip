// what do the per-chunk events cost the launch stream?  four ~10 us kernels per "chunk", an H2D per chunk on another stream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void spin_kernel(unsigned long long ticks, unsigned *out) {
  const unsigned long long t0 = wall_clock64();
  unsigned k = 0;
  while (wall_clock64() - t0 < ticks) ++k;
  if (threadIdx.x == 0 && out) out[blockIdx.x] = k;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t up = 1150u << 10;
  char *h_up, *d_up; unsigned *o;
  CK(hipHostMalloc((void **)&h_up, up, hipHostMallocDefault));
  CK(hipMalloc((void **)&d_up, up));
  CK(hipMalloc((void **)&o, 4096 * 4));
  memset(h_up, 1, up);
  hipStream_t sl, su;
  CK(hipStreamCreate(&sl));
  CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
  const int NE = 64;
  hipEvent_t evU[NE], evD[NE], evDdev[NE];
  for (int i = 0; i < NE; ++i) {
    CK(hipEventCreateWithFlags(&evU[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&evD[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&evDdev[i], hipEventDisableTiming | hipEventReleaseToDevice));
  }
  const unsigned long long t10 = 1000;  // 10 us at 100 MHz
  for (int w = 0; w < 500; ++w) { hipLaunchKernelGGL(spin_kernel, dim3(1250), dim3(256), 0, sl, t10, o); CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, su)); }
  CK(hipDeviceSynchronize());
  const int chunks = 400;
  for (int outer = 0; outer < 2; ++outer)
  for (int mode = 0; mode < 7; ++mode) {
    // 0: kernels only; 1: + record after; 2: + wait on upload event before; 3: both (the product); 4: both, record with
    // ReleaseToDevice; 5: both, one 40 us kernel instead of four of 10; 6: record only every 4th chunk
    CK(hipDeviceSynchronize());
    const double t0 = now_us();
    for (int k = 0; k < chunks; ++k) {
      const int e = k % NE;
      const bool upl = mode == 2 || mode >= 3;
      if (upl) { CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, su)); CK(hipEventRecord(evU[e], su)); CK(hipStreamWaitEvent(sl, evU[e], 0)); }
      if (mode == 5) hipLaunchKernelGGL(spin_kernel, dim3(1250), dim3(256), 0, sl, 4 * t10, o);
      else for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(spin_kernel, dim3(1250), dim3(256), 0, sl, t10, o);
      if (mode == 1 || mode == 3 || mode == 5) CK(hipEventRecord(evD[e], sl));
      if (mode == 4) CK(hipEventRecord(evDdev[e], sl));
      if (mode == 6 && (k & 3) == 3) CK(hipEventRecord(evD[e], sl));
      if ((k & 15) == 15) CK(hipEventSynchronize(mode == 4 ? evDdev[(k - 8) % NE] : (mode == 1 || mode == 3 || mode == 5) ? evD[(k - 8) % NE] : evD[0]));  // (keeps the queue from running far ahead)
    }
    CK(hipDeviceSynchronize());
    const double us = (now_us() - t0) / chunks;
    static const char *nm[] = {"4 kernels", "4 kernels + record", "wait(upload) + 4 kernels", "wait + 4 kernels + record", "wait + 4 kernels + record(ReleaseToDevice)",
                               "wait + 1 kernel of 40 us + record", "wait + 4 kernels, record every 4th"};
    if (outer) printf("%-48s %6.1f us per chunk (kernels alone: 40)\n", nm[mode], us);
  }
  return 0;
}
