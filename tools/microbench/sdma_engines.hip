// how fast is a 1.45 MB device -> host copy on each SDMA engine right now?  (csrc/grim_sdma.cpp names the engine)
//   hipcc --offload-arch=gfx950 -O2 -I ../../py-graph-imputation_amd/csrc sdma_engines.hip ../../py-graph-imputation_amd/csrc/grim_sdma.cpp -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "grim_sdma.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const size_t bytes = 1455136;
  char *h, *d, *hu, *du;
  CK(hipHostMalloc((void **)&h, bytes, hipHostMallocDefault));
  CK(hipMalloc((void **)&d, bytes));
  CK(hipHostMalloc((void **)&hu, bytes, hipHostMallocDefault));
  CK(hipMalloc((void **)&du, bytes));
  CK(hipMemset(d, 1, bytes));
  CK(hipDeviceSynchronize());
  hipStream_t su;
  CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
  char bdf[64];
  CK(hipDeviceGetPCIBusId(bdf, 64, 0));
  for (int pass = 0; pass < 2; ++pass)
    for (unsigned e = 1; e <= 0x80; e <<= 1) {
      char env[32];
      snprintf(env, sizeof env, "0x%x", e);
      setenv("GRIM_SDMA_ENGINE", env, 1);
      const char *why = nullptr;
      GrimSdma *s = grim_sdma_open(bdf, 0, &why);
      if (!s) { printf("engine 0x%x: not opened (%s)\n", e, why ? why : "?"); continue; }
      uint64_t job = 0;
      if (grim_sdma_job_create(s, &job) != 0) { printf("no signal\n"); return 1; }
      double alone = 0, beside = 0;
      int ok = 1;
      for (int mode = 0; mode < 2 && ok; ++mode) {
        for (int w = 0; w < 5 && ok; ++w) ok = grim_sdma_d2h_issue(s, job, h, d, bytes) == 0 && grim_sdma_wait(s, job) == 0;
        const int reps = 40;
        const double t0 = now_us();
        for (int r = 0; r < reps && ok; ++r) {
          if (mode) CK(hipMemcpyAsync(du, hu, bytes, hipMemcpyHostToDevice, su));  // an upload beside it, as in the product
          ok = grim_sdma_d2h_issue(s, job, h, d, bytes) == 0 && grim_sdma_wait(s, job) == 0;
        }
        CK(hipStreamSynchronize(su));
        (mode ? beside : alone) = (now_us() - t0) / reps;
      }
      if (ok) printf("pass %d engine 0x%02x: %6.1f us per 1.45 MB down alone (%5.1f GB/s), %6.1f us with an upload beside it\n", pass, e, alone, bytes / alone / 1e3, beside);
      else printf("pass %d engine 0x%02x: copy refused\n", pass, e);
      grim_sdma_job_destroy(s, job);
      grim_sdma_close(s);
    }
  return 0;
}
