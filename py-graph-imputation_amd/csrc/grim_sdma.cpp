// grim_sdma.cpp -- see grim_sdma.h
#include "grim_sdma.h"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

struct GrimSdma {
  hsa_agent_t gpu, cpu;
  hsa_amd_sdma_engine_id_t engine;
  uint64_t spin_ticks;  // ~100 us of the signal clock
  uint32_t rec, avail;  // ROCr's recommended / available engines for device -> host at open time
  bool pinned_by_env;
};

namespace {
struct Agents {
  std::vector<hsa_agent_t> gpus, cpus;
};
hsa_status_t collect(hsa_agent_t a, void *data) {
  Agents *A = (Agents *)data;
  hsa_device_type_t t;
  if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  if (t == HSA_DEVICE_TYPE_GPU) A->gpus.push_back(a);
  else if (t == HSA_DEVICE_TYPE_CPU) A->cpus.push_back(a);
  return HSA_STATUS_SUCCESS;
}
}  // namespace

GrimSdma *grim_sdma_open(const char *hip_pci_bus_id, int hip_device_ordinal, const char **why) {
  static const char *dummy;
  if (!why) why = &dummy;
  *why = nullptr;
  if (hsa_init() != HSA_STATUS_SUCCESS) {  // (reference counted: the HIP runtime holds ROCr open already)
    *why = "hsa_init failed";
    return nullptr;
  }
  Agents A;
  if (hsa_iterate_agents(collect, &A) != HSA_STATUS_SUCCESS || A.gpus.empty() || A.cpus.empty()) {
    *why = "no GPU or CPU agent";
    hsa_shut_down();
    return nullptr;
  }
  // the HIP device among ROCr's agents: by PCI address ("dddd:bb:dd.f"); by position only when there is a single GPU
  int found = -1;
  unsigned dom = 0, bus = 0, dev = 0, fn = 0;
  if (hip_pci_bus_id && sscanf(hip_pci_bus_id, "%x:%x:%x.%x", &dom, &bus, &dev, &fn) == 4) {
    for (size_t i = 0; i < A.gpus.size(); ++i) {
      uint32_t bdf = 0, d = 0;
      if (hsa_agent_get_info(A.gpus[i], (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) continue;
      (void)hsa_agent_get_info(A.gpus[i], (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &d);
      if (((bdf >> 8) & 0xFF) == bus && ((bdf >> 3) & 0x1F) == dev && (bdf & 7) == fn && d == dom) {
        found = (int)i;
        break;
      }
    }
  }
  if (found < 0 && A.gpus.size() == 1 && hip_device_ordinal == 0) found = 0;
  if (found < 0) {
    *why = "the HIP device was not found among ROCr's agents";
    hsa_shut_down();
    return nullptr;
  }
  GrimSdma *s = new GrimSdma();
  s->gpu = A.gpus[(size_t)found];
  s->cpu = A.cpus[0];
  // the engine: ROCr's recommendation for device -> host, never engine 0 (where the HIP runtime puts the uploads)
  uint32_t rec = 0, avail = 0;
  (void)hsa_amd_memory_get_preferred_copy_engine(s->cpu, s->gpu, &rec);
  if (hsa_amd_memory_copy_engine_status(s->cpu, s->gpu, &avail) != HSA_STATUS_SUCCESS) avail = 0;
  uint32_t pick = rec & ~1u;
  if (!pick) pick = avail & ~1u;
  s->rec = rec;
  s->avail = avail;
  s->pinned_by_env = false;
  if (const char *e = getenv("GRIM_SDMA_ENGINE")) {
    pick = (uint32_t)strtoul(e, nullptr, 0);
    s->pinned_by_env = true;
  }
  if (!pick) {
    *why = "no SDMA engine besides engine 0";
    delete s;
    hsa_shut_down();
    return nullptr;
  }
  s->engine = (hsa_amd_sdma_engine_id_t)(pick & (0u - pick));
  uint64_t hz = 0;
  if (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &hz) != HSA_STATUS_SUCCESS || !hz) hz = 100000000ull;
  s->spin_ticks = hz / 10000;
  return s;
}

void grim_sdma_close(GrimSdma *s) {
  if (!s) return;
  delete s;
  hsa_shut_down();
}

uint32_t grim_sdma_engine(const GrimSdma *s) { return s ? (uint32_t)s->engine : 0; }

uint32_t grim_sdma_pick(GrimSdma *s, void *dst_host, const void *src_dev, size_t bytes, char *report, size_t report_len) {
  if (!s) return 0;
  if (report && report_len) report[0] = 0;
  if (s->pinned_by_env) return (uint32_t)s->engine;
  uint64_t job = 0;
  if (grim_sdma_job_create(s, &job) != 0) return (uint32_t)s->engine;
  uint64_t hz = 0;
  if (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &hz) != HSA_STATUS_SUCCESS || !hz) hz = 100000000ull;
  auto now = [] {
    uint64_t t = 0;
    hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP, &t);
    return t;
  };
  const hsa_amd_sdma_engine_id_t before = s->engine;
  uint32_t cand = (s->avail ? s->avail : s->rec) & 0xFEu;  // engines 1-7 (the ones beyond are no faster over PCIe: 160-220 us)
  if (!cand) cand = (uint32_t)before;
  double best = 0, t_of[16];
  uint32_t best_e = 0;
  size_t used = 0;
  for (int k = 0; k < 16; ++k) {
    t_of[k] = 0;
    const uint32_t e = 1u << k;
    if (!(cand & e)) continue;
    s->engine = (hsa_amd_sdma_engine_id_t)e;
    bool ok = grim_sdma_d2h_issue(s, job, dst_host, src_dev, bytes) == 0 && grim_sdma_wait(s, job) == 0;  // (the first copy on an engine sets its queue up)
    double fastest = 0;
    for (int r = 0; r < 3 && ok; ++r) {
      const uint64_t t0 = now();
      ok = grim_sdma_d2h_issue(s, job, dst_host, src_dev, bytes) == 0 && grim_sdma_wait(s, job) == 0;
      const double us = (double)(now() - t0) * 1e6 / (double)hz;
      if (ok && (fastest == 0 || us < fastest)) fastest = us;
    }
    if (!ok || fastest == 0) continue;
    t_of[k] = fastest;
    if (report && used + 24 < report_len) used += (size_t)snprintf(report + used, report_len - used, "%s0x%x %.0f us", used ? ", " : "", e, fastest);
    if (best == 0 || fastest < best) {
      best = fastest;
      best_e = e;
    }
  }
  // ROCr's recommended engines first among those within a quarter of the fastest
  uint32_t pick = 0;
  for (int pass = 0; pass < 2 && !pick; ++pass)
    for (int k = 0; k < 16 && !pick; ++k)
      if (t_of[k] > 0 && t_of[k] <= 1.25 * best && (pass == 1 || (s->rec & (1u << k)))) pick = 1u << k;
  if (!pick) pick = best_e;
  grim_sdma_job_destroy(s, job);
  s->engine = pick ? (hsa_amd_sdma_engine_id_t)pick : before;
  return pick;
}

int grim_sdma_job_create(GrimSdma *s, uint64_t *job) {
  if (!s || !job) return -1;
  hsa_signal_t sig;
  if (hsa_signal_create(0, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) return -1;
  *job = sig.handle;
  return 0;
}

void grim_sdma_job_destroy(GrimSdma *s, uint64_t job) {
  if (!s || !job) return;
  hsa_signal_t sig;
  sig.handle = job;
  hsa_signal_destroy(sig);
}

int grim_sdma_d2h_issue(GrimSdma *s, uint64_t job, void *dst_host, const void *src_dev, size_t bytes) {
  if (!s || !job) return -1;
  hsa_signal_t sig;
  sig.handle = job;
  hsa_signal_store_relaxed(sig, 1);
  const hsa_status_t st = hsa_amd_memory_async_copy_on_engine(dst_host, s->cpu, src_dev, s->gpu, bytes, 0, nullptr, sig, s->engine, true);
  if (st != HSA_STATUS_SUCCESS) {
    hsa_signal_store_relaxed(sig, 0);
    return -1;
  }
  return 0;
}

int grim_sdma_wait(GrimSdma *s, uint64_t job) {
  if (!s || !job) return -1;
  hsa_signal_t sig;
  sig.handle = job;
  // a short spin (the copy takes ~35 us; 100 us at most), then blocked waits
  hsa_signal_value_t v = hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, s->spin_ticks, HSA_WAIT_STATE_ACTIVE);
  while (v >= 1) v = hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
  return v < 0 ? -1 : 0;  // a failed copy leaves a negative value
}
