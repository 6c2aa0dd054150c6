// H2D and D2H copies of a chunk's size on two streams at once: do the DMA engines run them side by side?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t up = 1150u << 10, down = 1300u << 10;
  char *h_up, *h_down, *d_up, *d_down;
  CK(hipHostMalloc((void **)&h_up, up, hipHostMallocDefault));
  CK(hipHostMalloc((void **)&h_down, down, hipHostMallocDefault));
  CK(hipMalloc((void **)&d_up, up));
  CK(hipMalloc((void **)&d_down, down));
  memset(h_up, 1, up);
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev[64];
  for (int i = 0; i < 64; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  const int reps = 1000;
  for (int w = 0; w < 3000; ++w) { CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, s2)); }
  CK(hipDeviceSynchronize());
  for (int outer = 0; outer < 2; ++outer)
  for (int mode = 0; mode < 5; ++mode) {
    // 0: H2D only, 1: D2H only, 2: both, 3: both with an event recorded behind every copy, 4: both, D2H in four pieces
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipDeviceSynchronize());
      const double t0 = now_us();
      for (int r = 0; r < reps; ++r) {
        if (mode != 1) CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, s1));
        if (mode == 3) CK(hipEventRecord(ev[r & 31], s1));
        if (mode == 4) { for (int q = 0; q < 4; ++q) CK(hipMemcpyAsync(h_down + q * (down / 4), d_down + q * (down / 4), down / 4, hipMemcpyDeviceToHost, s2)); }
        else if (mode != 0) CK(hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, s2));
        if (mode == 3) CK(hipEventRecord(ev[32 + (r & 31)], s2));
      }
      CK(hipDeviceSynchronize());
      const double us = (now_us() - t0) / reps;
      if (pass) printf("mode %d: %.1f us per round (1.15 MB up%s)\n", mode, us, mode == 0 ? "" : mode == 1 ? " -- none; 1.3 MB down only" : " + 1.3 MB down");
    }
  }
  return 0;
}
