// grim_sdma.h -- results D2H on an SDMA engine of our own choosing (ROCr, not hipMemcpyAsync).
//
// Why: a stream's chunk goes up (1.1 MB of text) and its results come down (1.3-1.5 MB) for every 10 000 lines.  The HIP
// runtime gives each HIP stream "the first free engine" when the stream copies for the first time and keeps it: both the
// upload stream and the copy stream of a context get SDMA engine 0 (AMD_LOG_LEVEL=4: copy_engine=0x1 for every copy of a
// run), so the two directions of a full-duplex link take turns -- 63 us per 1.3 MB download in the pipeline against 33 us
// alone (profiles/r4_notes.md).  The way round that in round 3 was a copy KERNEL (grim_export_kernel), but a kernel that
// writes host memory does not run beside other kernels on this chip: the launch stream stands still for the copy's 30 us
// (build_tmp-style micro-benchmark in profiles/r4_notes.md: 44 us of kernels + a 36 us export kernel on another stream =
// 80 us; + an SDMA copy = 45 us).  hsa_amd_memory_async_copy_on_engine names the engine: downloads go to the engine ROCr
// recommends for device->host (mask 0x6 on MI355X: engine 1), uploads stay where the HIP runtime puts them (engine 0).
//
// Ordering: the caller has waited (on the host) for the kernels that wrote the source, so the copy needs no GPU-side
// dependency; its completion is an HSA signal the fetch thread waits for.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct GrimSdma;

// nullptr when ROCr cannot be reached, the device is not found among its agents, or no engine other than the uploads' is
// on offer: the caller keeps its other way (`why`, when given, says which).
GrimSdma *grim_sdma_open(const char *hip_pci_bus_id, int hip_device_ordinal, const char **why);
void grim_sdma_close(GrimSdma *s);
uint32_t grim_sdma_engine(const GrimSdma *s);  // the engine's bit (hsa_amd_sdma_engine_id_t)
// Chooses the engine by MEASUREMENT: a few copies of `bytes` from `src_dev` to `dst_host` on every engine on offer except
// engine 0, the fastest wins (ROCr's recommended engines first among equals).  Why: an engine can be slow for reasons nobody
// reports -- on MI355X engines 4-7 move 11 GB/s over PCIe where engines 0-3 move 40, and engine 1, the recommended one, drops
// to 9 GB/s for every later process once some process has allocated and freed tens of GB (profiles/r4_notes.md,
// tools/microbench/sdma_engines.hip).  Returns the engine's bit, 0 when no copy worked (the caller keeps another way).
// `report` (optional) gets one line of the timings.  GRIM_SDMA_ENGINE pins the choice and skips the measurement.
uint32_t grim_sdma_pick(GrimSdma *s, void *dst_host, const void *src_dev, size_t bytes, char *report, size_t report_len);

// one job = one completion signal, reused
int grim_sdma_job_create(GrimSdma *s, uint64_t *job);
void grim_sdma_job_destroy(GrimSdma *s, uint64_t job);
// device memory (hipMalloc) -> pinned host memory (hipHostMalloc); 0 on success
int grim_sdma_d2h_issue(GrimSdma *s, uint64_t job, void *dst_host, const void *src_dev, size_t bytes);
int grim_sdma_wait(GrimSdma *s, uint64_t job);
