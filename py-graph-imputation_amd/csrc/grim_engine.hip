// grim_engine.hip -- libgrim_hip.so: kernels + the C-ABI of include/grim_hip.h (gfx950 only).
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared grim_engine.hip -o libgrim_hip.so
// (-ffp-contract=off is REQUIRED: an fma in P1*P2*prior would change the last bit and with it
//  rankings that the reference decides on exact ties).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "grim_plan_a.h"
#include "grim_plan_b.h"
#include "grim_small.h"
#include "grim_medium.h"
#include "grim_mid.h"
#include "grim_tables.h"
#include "grim_tokdev.h"
#include "grim_engine_internal.h"
#include "grim_host_internal.h"
#include "grim_sdma.h"

// =================================================================================================
// Plan-A kernel: one workgroup per subject, pulled from a work counter.  Waves build the phase
// sides (one wave per side, wave64 ballots/prefix scans, LDS-staged running top-K); the whole
// workgroup then scores haplotype pairs, dedups, sums and ranks.
// =================================================================================================
__global__ __launch_bounds__(GRIM_WG, GRIM_WG_PER_CU) void grim_plan_a_kernel(DevArgs A) {
  __shared__ WgShared sh;
  __shared__ WaveTop wt[GRIM_NWAVE];
  const int tid = threadIdx.x;
  const int P = A.g.P;
  Slot S = make_slot(A, blockIdx.x);
  // hand-overs of the one-wave kernel (earlier in the stream): the heavier ones, at the back of its list, are
  // taken first, then this kernel's own subjects (heaviest first), then the light hand-overs
  const uint32_t n_bail = A.bail_list ? A.queue[5] : 0, n_bail_heavy = A.bail_list ? A.queue[7] : 0;
  // behind the mid-size kernel (grim_mid.h) this kernel only takes what that one handed over
  const uint32_t n_mid = A.mid_list ? A.queue[17] : 0;
  if (tid < GRIM_NWAVE * 4) ((unsigned long long *)sh.wctr)[tid] = 0;
  wg_arena(sh, wt);
  __syncthreads();
  for (;;) {
    if (tid == 0) sh.bc[3] = atomicAdd(A.queue, 1u);
    __syncthreads();
    const uint32_t w = sh.bc[3];
    if (A.mid_list ? w >= n_mid : w >= A.n_work + n_bail + n_bail_heavy) break;
    uint32_t si;
    if (A.mid_list)
      si = A.mid_list[w];
    else if (w < n_bail_heavy)
      si = A.bail_list[A.n_medium - 1u - w];
    else if (w - n_bail_heavy < A.n_work)
      si = A.order[w - n_bail_heavy];
    else
      si = A.bail_list[w - n_bail_heavy - A.n_work];
    if (tid < 16) ((uint32_t *)&sh.subj)[tid] = ((const uint32_t *)&A.subj[si])[tid];
    if (tid < GRIM_SIDES) {
      sh.Tn[tid] = 0;
      sh.cand_any[tid] = 0;
    }
    if (tid < (int)(sizeof(grim_subject_result) / 4)) ((uint32_t *)&sh.out)[tid] = 0;
    __syncthreads();
    STAMP_BEGIN();
    const unsigned long long t_subject = STAMP_NOW();
    (void)t_subject;
    enumerate_phases(sh);
    const double *prior = A.priors + (uint64_t)sh.subj.prior_idx * P * P;
    const int nph = sh.nph;
    const bool fits = prepare_lists(A, sh, S);
    bool kept = false;
    // open_phases; when no phase has candidates on both sides the reference rewrites the '/'-lists
    // (known alleles only, then the 10 most frequent) and opens again (impute.py:1619-1627)
    for (int stage = 0; fits; ++stage) {
      // one scan of the label for all sides when that is cheaper than opening them one by one (high ambiguity on a big graph)
      bool done = false;
      if (stage == 0 && shared_scan_wanted(A, sh)) done = build_sides_shared_scan(A, sh, S, prior, wt);
      if (stage == 0) HIST(0, done ? 1 : 0, 1);  // (diagnostic build) subjects by opening: bucket 0 side by side, bucket 1 shared scan
      if (!done)
        for (int s = wave_id(); s < 2 * nph; s += GRIM_NWAVE) build_side_plan_a(A, sh, S, prior, wt[wave_id()], s >> 1, s & 1, s);
      __syncthreads();
      kept = false;
      for (int i = 0; i < nph; ++i) kept |= (sh.cand_any[2 * i] && sh.cand_any[2 * i + 1]);
      if (kept || stage == 2) break;
      if (stage == 0) reduce_lists(A, sh, S, prior);
      apply_stage(A, sh, stage + 1);
    }
    STAMP(8);
    const unsigned long long t_sides = STAMP_NOW();
    HIST(1, sh.ntok, (t_sides - t_subject) / 100);  // us in the sides by number of alleles in the GL string
    HIST(2, sh.ntok, 1);
    uint8_t status = GRIM_ST_MISS, reason = 0, plan = 'a';
    if (!fits) {
      status = GRIM_ST_UNSUPPORTED;  // more than GRIM_RTOK_CAP/3 alleles in one GL string
      reason = 5;
    } else if (!kept) {
      // no phase at all: the reference returns its placeholder result and trips over it while writing
      // the phased file (impute.py:1607-1609, 2090-2097)
      status = GRIM_ST_NOPHASE;
    } else {
      const uint32_t np = pair_offsets(sh);
      int e = A.prm.n_ladder;
      if (np > 0) e = ladder_first(A, sh, S, prior, np);
      STAMP(9);
      uint32_t nU = 0;
      double mx = 0.0;
      if (e < A.prm.n_ladder) {
        double eps = A.prm.ladder[e];
        bool first = true;
        if (eps > 0.0) {
          pair_pass(A, sh, S, prior, np, eps, false, &mx);
          STAMP(10);
          eps = mx / 100000.0;  // impute.py:1685
          first = false;
        }
        nU = pair_pass(A, sh, S, prior, np, eps, true, &mx, first);
        STAMP(11);
      }
      const unsigned long long t_pairs = STAMP_NOW();
      (void)t_pairs;
      HIST(3, nU, 1);
      HIST(4, np, 1);
      HIST(5, sh.poff[0] + [&]() { uint32_t e = 0; for (int s = 0; s < 2 * sh.nph; ++s) e += sh.Tn[s]; return e; }(), 1);
      HIST(6, nU, nU);
      HIST(7, nU, (t_pairs - t_subject) / 100);
      if (nU > 0) {
        emit_tables(A, sh, S, nU, sh.out, si);
        STAMP(12);
        status = GRIM_ST_OK;
        if (tid == 0) sh.out.max_prob = mx;
      } else if (A.prm.planb) {
        status = GRIM_ST_UNSUPPORTED;  // replaced by the plan-B kernel's verdict when it runs
        reason = 2;
        if (tid == 0 && A.next_list) push_next(A, si, sh.subj.n_loci <= GRIM_HEAVY_LOCI);
      }
    }
    __syncthreads();
    if (tid == 0) {
      sh.out.status = status;
      sh.out.reason = reason;
      sh.out.plan = plan;
      A.res[si] = sh.out;
    }
    __syncthreads();
  }
  if (tid == 0) {
    unsigned long long c0 = 0, c1 = 0, c2 = 0;
    for (int wv = 0; wv < GRIM_NWAVE; ++wv) {
      c0 += sh.wctr[wv][0];
      c1 += sh.wctr[wv][1];
      c2 += sh.wctr[wv][2];
    }
    atomicAdd(&A.counters[0], c0);
    atomicAdd(&A.counters[1], c1);
    atomicAdd(&A.counters[2], c2);
  }
}


// The half-wave kernel writes a subject's rows at a fixed stride (no atomics on its hot path) into a staging region at
// the TOP of the row pool; most of that region stays empty (1 + ~1.3 of 11 rows per subject).  This kernel moves the
// rows that exist into the bump-allocated part of the pool -- one allocation per wave of 64 subjects -- and re-bases the
// subjects' row offsets, so that the batch's D2H copy carries ~75 instead of 352 bytes of rows per subject.
__global__ __launch_bounds__(64) void grim_small_compact_kernel(DevArgs A, const uint32_t *order_s, const SmallRec *recs, uint32_t n_small,
                                                                 uint32_t stage_base, uint32_t stride) {
  const uint32_t w = blockIdx.x * 64 + threadIdx.x;
  const int lane = lane_id();
  // a subject's staged rows are one run: [the row that is its .umug, .umug.pops and .pmug.pops row, its .pmug rows] from stage_base + w * stride
  uint32_t si = 0, cnt = 0;
  uint4 ro = make_uint4(0, 0, 0, 0), nr = make_uint4(0, 0, 0, 0);
  // record w's subject: from the class list (host-tokenised subjects), or from the record itself (device-tokenised
  // lines: GRIM_NONE where the line is not a device subject)
  if (w < n_small) si = order_s ? order_s[w] : recs[w].si;
  if (w < n_small && si != GRIM_NONE) {
    const uint32_t *r = (const uint32_t *)(A.res + si);  // dwords 3..6 row_off, 7..10 n_rows
    ro = make_uint4(r[3], r[4], r[5], r[6]);
    nr = make_uint4(r[7], r[8], r[9], r[10]);
    if ((nr.x | nr.y | nr.z | nr.w) != 0 && ro.z >= stage_base) cnt = GRIM_SMALL_ROWS_FIXED + nr.z;  // .z: GRIM_T_PMUG
  }
  uint32_t incl = cnt;  // inclusive prefix over the wave
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  const uint32_t total = __shfl(incl, 63);
  if (total == 0) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(A.row_head, total);
  base = __shfl(base, 0);
  if (base + total > stage_base) {  // the pool is full: reported like every other row overflow (the caller splits the batch)
    if (lane == 0) atomicExch(&A.counters[4], 1ull);
    return;
  }
  if (cnt == 0) return;
  const uint32_t src = stage_base + w * stride, dst = base + incl - cnt;
  const uint4 *sp = (const uint4 *)(A.rows + src);
  uint4 *dp = (uint4 *)(A.rows + dst);
  uint32_t k = 0;
  for (; k + 2 <= cnt; k += 2) {  // two rows (four 16-byte loads) in flight
    const uint4 a0 = sp[2 * k], a1 = sp[2 * k + 1], a2 = sp[2 * k + 2], a3 = sp[2 * k + 3];
    dp[2 * k] = a0;
    dp[2 * k + 1] = a1;
    dp[2 * k + 2] = a2;
    dp[2 * k + 3] = a3;
  }
  if (k < cnt) {
    const uint4 a0 = sp[2 * k], a1 = sp[2 * k + 1];
    dp[2 * k] = a0;
    dp[2 * k + 1] = a1;
  }
  const uint32_t delta = dst - src;  // modulo 2^32: offsets move down
  uint32_t *r = (uint32_t *)(A.res + si);
  r[3] = ro.x + delta;
  r[4] = ro.y + delta;
  r[5] = ro.z + delta;
  r[6] = ro.w + delta;
}

// A finished batch's results, device arena -> pinned host memory, by a few workgroups instead of the DMA engine: the
// engine moves a 1.3 MB block (10 000 config-2 subjects) at ~20 GB/s, which made the D2H copy the slowest stage of the
// stream (62 us per chunk against 33 us of kernels); sixteen-byte stores from 32 workgroups fill the PCIe link without
// taking the chip from the kernels of the next batches (the runtime's own blit kernel, which rocprofv3 forces, does take
// it: the half-wave kernel beside it stretches from 11 to 49 us).  GRIM_EXPORT_KERNEL=0: hipMemcpyAsync as before.
__global__ __launch_bounds__(256) void grim_export_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, uint64_t n16) {
  const uint64_t step = (uint64_t)gridDim.x * 256u;
  uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  for (; i + 3 * step < n16; i += 4 * step) {  // four loads in flight per lane
    const uint4 a = src[i], b = src[i + step], c = src[i + 2 * step], d = src[i + 3 * step];
    dst[i] = a;
    dst[i + step] = b;
    dst[i + 2 * step] = c;
    dst[i + 3 * step] = d;
  }
  for (; i < n16; i += step) dst[i] = src[i];
}

// zero the counters and work heads of a batch (one launch instead of several memsets)
__global__ void grim_reset_kernel(unsigned long long *counters, uint32_t *queue, uint32_t row_head0) {
  for (int i = threadIdx.x; i < GRIM_NCTR; i += blockDim.x) counters[i] = 0;
  if (threadIdx.x < GRIM_NQ) queue[threadIdx.x] = threadIdx.x == 1 ? row_head0 : 0u;
}

// End of a stage: the state block (counters, then the eight queue words) goes straight to the batch's pinned
// host copy, and -- unless Plan B still has work queued, which needs the heads as they are -- the device copy is
// zeroed for the next run.  One small kernel instead of a copy engine job now and a reset kernel next time.
__global__ void grim_finish_kernel(unsigned long long *state, unsigned long long *host_state, uint32_t row_head0, int after_plan_b) {
  __shared__ uint32_t pending;
  const uint32_t *queue = (const uint32_t *)(state + GRIM_NCTR);
  // subjects waiting for Plan B/C, or accepted pairs waiting for the table kernels: a second stage follows
  if (threadIdx.x == 0) pending = after_plan_b ? 0u : queue[2] + queue[6] + queue[9] + queue[10];
  for (int i = threadIdx.x; i < GRIM_NCTR + GRIM_NQ / 2; i += blockDim.x) host_state[i] = state[i];
  __syncthreads();
  if (pending) return;  // the second stage works on this state
  for (int i = threadIdx.x; i < GRIM_NCTR; i += blockDim.x) state[i] = 0;
  if (threadIdx.x < GRIM_NQ) ((uint32_t *)(state + GRIM_NCTR))[threadIdx.x] = threadIdx.x == 1 ? row_head0 : 0u;
}

// =================================================================================================
// host side
// =================================================================================================
struct grim_ctx {
  int device;
  hipStream_t stream;
  hipStream_t copy_stream;  // D2H of a finished batch while the next batch's kernels run (engine_batch_fetch_async)
  hipStream_t up_stream;    // H2D of the next batch's input while this batch's kernels run (engine_batch_load)
  GrimSdma *sdma;           // results D2H on an SDMA engine of its own (grim_sdma.h); nullptr: export kernel / hipMemcpyAsync
  int export_mode;          // 2 = grim_sdma, 1 = grim_export_kernel, 0 = hipMemcpyAsync on the copy stream
  // Two threads of a stream use one context at the same time (device thread: loads and launches; copy thread: waits,
  // second stages, D2H), so the error text is written and read under its own lock; grim_last_error hands out a
  // per-thread copy.  Everything else a context owns is either immutable after grim_create or guarded by run_mu.
  std::mutex err_mu;
  std::string err;
  std::mutex run_mu;        // binding the scratch block / launching a run's kernels: one thread at a time
  int n_cu;
  // per-workgroup scratch slots, kept across batches (13 GB at the default sizes: allocating them
  // per batch cost more than the kernels)
  void *scratch;
  uint64_t scratch_bytes;
  uint32_t *wctr;           // the table kernels' sliced work counters (grim_dev.h: slice_next), zeroed in front of every use
  // capacity batches a finished stream handed back: the next stream on this context takes them instead of
  // allocating (and pinning) a few hundred MB per slot again
  std::vector<grim_batch *> spare;
};

struct grim_graph {
  grim_ctx *ctx;
  DevGraph d;
  std::vector<void *> bufs;
  uint64_t bytes;
  uint32_t max_label;
};

struct grim_batch {
  grim_ctx *ctx;
  const grim_graph *g;
  DevArgs a;
  // Three arenas, laid out per load (engine_batch_plan): what the next batch holds decides the offsets, so ONE copy
  // moves a batch's whole input to the device and ONE copy brings its results back.
  //   in   (device + pinned mirror): run state | subjects | half-wave records | three class lists | tokens
  //   work (device only)           : hand-over lists, per-wave counters
  //   out  (device + pinned mirror): result headers | row pool
  uint8_t *d_in, *h_in, *d_work, *d_out, *h_out;
  uint64_t in_cap, work_cap, out_cap, h_out_cap;
  double *d_priors, *h_priors;  // prior matrices: their own small buffers, uploaded only when the set changes
  uint8_t *d_pool;              // the table kernels' arena (device only; grows at load time): pair records, per-item state,
  uint64_t pool_cap, pool_bytes; // bucket starts, cells, work units, bucket order, groups, probabilities in cell order
  uint64_t pool_want;            // records the last run asked for when it ran out (the next load sizes the pool for it)
  uint64_t pool_asked;           // pair records the last run asked for
  EngineLoad last_load;          // what the last engine_batch_load was given (priors not kept): a run that outgrew the pair pool loads again
  bool in_retry;
  uint32_t priors_cap, priors_up;
  uint64_t row_limit;   // rows a run may use
  EnginePlan plan;      // what the arenas are laid out for
  uint64_t off_tok, off_rows;  // byte offsets inside in / out
  EngineHost h;         // pointers into the pinned arenas (valid until the next plan)
  uint32_t n_subj, n_slots;
  uint32_t n_small_waves;
  bool small_ctr_pending;  // the half-wave kernel's per-wave counts have not been added to `counters` yet
  unsigned long long *hstate;  // pinned: counters + work/row heads of the last run
  SmallRec *d_small;
  uint32_t *d_os, *d_om;
  // device tokenizer (plan.text_cap != 0)
  const grim_devdict *dict;
  LineRec *d_lines;
  uint8_t *d_text;
  SmallRec *d_dsmall;   // its half-wave records, one slot per line (work arena)
  uint64_t off_subj;    // the input arena up to here is all a run without host-tokenised subjects needs
  uint32_t dev_lo, dev_hi, n_dev_lines, n_irregular;
  uint32_t n_host_small_waves;
  float ms_k;           // timing mode: the tokenizer kernel
  uint32_t n_medium;
  uint32_t n_small, n_general, small_stride;
  uint64_t scratch_need;  // bytes of per-workgroup scratch this batch's runs need (bound at run time)
  hipEvent_t ev_copy;  // behind the D2H copy of engine_batch_fetch_issue
  int ct_slot = -1;    // GRIM_DEBUG_COPYTIME
  uint64_t sdma_job = 0;  // completion signal of grim_sdma copies (made on first use)
  bool fetch_hsa = false; // the copy in flight was issued through grim_sdma
  bool up_pending = false; // an input copy is on the upload stream and the launch stream has not been ordered behind it yet
  hipEvent_t ev_up;    // behind the H2D copy of engine_batch_load: the batch's kernels wait for it
  hipEvent_t ev_done;  // recorded behind the last kernel of a stage: what engine_batch_wait waits for (not the whole stream --
                       // the device thread may have queued the next chunk's kernels behind it already)
  bool enqueued;       // stage 1 is in flight (engine_batch_enqueue without its engine_batch_wait)
  hipEvent_t ev[18];  // timing mode ([16]/[17] the mid-size kernel), kernel start/stop ([14]/[15] device tokenizer): [3]/[5] half-wave, [0]/[1] one-wave, [6]/[7] general, [4]/[2] Plan B,
                      // [8]/[9] table kernels of stage 1, [10]/[11] table kernels after Plan B
  bool timing;       // GRIM_TIMING=1 or grim_batch_set_timing: direct launches with per-kernel events instead of the graph replay
  hipGraphExec_t gexec;
  int graph_state;  // 0 not tried, 1 captured, -1 direct launches
  float ms_a, ms_b, ms_s, ms_g, ms_m, ms_t, ms_c;  // ms_c: the half-wave kernel's row compaction
  float ms_d;         // the mid-size kernel (grim_mid.h)
  double acc_ms[10];  // sums over the timed runs since timing was switched on (index = `which`)
  uint32_t n_timed;
  uint32_t rows_used;
  unsigned long long counters[8];
};

static thread_local std::string g_err;

static void set_err(grim_ctx *c, const std::string &m) {
  g_err = m;
  if (c) {
    std::lock_guard<std::mutex> lk(c->err_mu);
    c->err = m;
  }
}

#define HIPCHK(call, ctxp, ret)                                                        \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      set_err((ctxp), std::string(#call) + ": " + hipGetErrorString(e_));             \
      return ret;                                                                      \
    }                                                                                  \
  } while (0)

extern "C" int grim_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" grim_ctx *grim_create(int device_id) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    g_err = "grim_create: no HIP device visible";
    return nullptr;
  }
  if (device_id < 0 || device_id >= n) {
    g_err = "grim_create: device id out of range";
    return nullptr;
  }
  grim_ctx *c = new grim_ctx();
  c->device = device_id;
  c->copy_stream = nullptr;
  c->up_stream = nullptr;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess ||
      hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking) != hipSuccess) {
    g_err = "grim_create: cannot initialise device";
    delete c;
    return nullptr;
  }
  hipDeviceProp_t prop;
  c->n_cu = 256;
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  c->wctr = nullptr;
  if (hipMalloc((void **)&c->wctr, 4 * GRIM_WCTR_WORDS) != hipSuccess) {
    (void)hipGetLastError();
    g_err = "grim_create: cannot allocate the work counters";
    hipStreamDestroy(c->stream);
    hipStreamDestroy(c->copy_stream);
    hipStreamDestroy(c->up_stream);
    delete c;
    return nullptr;
  }
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->n_cu = prop.multiProcessorCount;
  // how a stream's results come down: GRIM_EXPORT=sdma|kernel|memcpy (GRIM_EXPORT_KERNEL=0/1 of round 3 still means
  // memcpy / kernel); default sdma when ROCr offers an engine besides the uploads', else the export kernel
  c->sdma = nullptr;
  c->export_mode = 1;
  {
    const char *e = getenv("GRIM_EXPORT");
    int want = 2;
    if (e && !strcmp(e, "kernel")) want = 1;
    else if (e && !strcmp(e, "memcpy")) want = 0;
    else if (!e && getenv("GRIM_EXPORT_KERNEL")) want = atoi(getenv("GRIM_EXPORT_KERNEL")) ? 1 : 0;
    if (want == 2) {
      char bdf[64] = {0};
      const char *why = nullptr;
      if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device_id) != hipSuccess) bdf[0] = 0;
      c->sdma = grim_sdma_open(bdf[0] ? bdf : nullptr, device_id, &why);
      if (c->sdma) {
        // the engine by measurement (grim_sdma_pick: a chunk's worth of results down on every engine on offer, ~1 ms)
        const size_t probe = 1500000;
        void *hp = nullptr, *dp = nullptr;
        char report[256];
        uint32_t got = 0;
        if (hipHostMalloc(&hp, probe, hipHostMallocDefault) == hipSuccess && hipMalloc(&dp, probe) == hipSuccess &&
            hipMemset(dp, 0, probe) == hipSuccess && hipDeviceSynchronize() == hipSuccess)
          got = grim_sdma_pick(c->sdma, hp, dp, probe, report, sizeof(report));
        (void)hipGetLastError();
        if (dp) (void)hipFree(dp);
        if (hp) (void)hipHostFree(hp);
        if (getenv("GRIM_DEBUG_STREAM")) fprintf(stderr, "grim: SDMA engines for the downloads: %s -> 0x%x\n", got ? report : "none worked", got);
        if (!got) {
          why = "no SDMA engine completed a test copy";
          grim_sdma_close(c->sdma);
          c->sdma = nullptr;
        }
      }
      if (!c->sdma && e) fprintf(stderr, "grim: GRIM_EXPORT=sdma is not available (%s); using the export kernel\n", why ? why : "?");
      want = c->sdma ? 2 : 1;
    }
    c->export_mode = want;
  }
  return c;
}

extern "C" int grim_export_engine(grim_ctx *c) {
  if (!c) return -1;
  return c->export_mode == 2 ? (int)grim_sdma_engine(c->sdma) : c->export_mode == 1 ? 0 : -1;
}

static void batch_destroy(grim_batch *b);

extern "C" void grim_destroy(grim_ctx *c) {
  if (!c) return;
  hipSetDevice(c->device);
  for (grim_batch *b : c->spare) batch_destroy(b);
  c->spare.clear();
  if (c->scratch) hipFree(c->scratch);
  if (c->wctr) hipFree(c->wctr);
  if (c->sdma) grim_sdma_close(c->sdma);
  c->sdma = nullptr;
  hipStreamDestroy(c->stream);
  if (c->copy_stream) hipStreamDestroy(c->copy_stream);
  if (c->up_stream) hipStreamDestroy(c->up_stream);
  delete c;
}

extern "C" const char *grim_last_error(grim_ctx *c) {
  if (c) {  // a copy private to the calling thread: the context's text may be rewritten by another thread meanwhile
    std::lock_guard<std::mutex> lk(c->err_mu);
    g_err = c->err;
  }
  return g_err.c_str();
}

template <typename T>
static T *upload(grim_ctx *c, std::vector<void *> &bufs, const T *src, size_t n, uint64_t *bytes) {
  void *p = nullptr;
  size_t sz = (n ? n : 1) * sizeof(T);
  if (hipMalloc(&p, sz) != hipSuccess) return nullptr;
  bufs.push_back(p);
  if (n && src && hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  if (bytes) *bytes += sz;
  return (T *)p;
}

static uint64_t host_mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

extern "C" grim_graph *grim_graph_upload(grim_ctx *c, const grim_graph_desc *d) {
  if (!c || !d) return nullptr;
  hipSetDevice(c->device);
  if (d->n_pops == 0 || d->n_pops > GRIM_MAXPOP) { set_err(c, "grim_graph_upload: population count out of range"); return nullptr; }
  if (d->n_loci == 0 || d->n_loci > GRIM_MAXL) { set_err(c, "grim_graph_upload: locus count out of range"); return nullptr; }
  if (d->n_nodes >= (1u << 23)) { set_err(c, "grim_graph_upload: more than 2^23 nodes"); return nullptr; }
  grim_graph *g = new grim_graph();
  g->ctx = c;
  g->bytes = 0;
  // exact-name index: open addressing over the 64-bit node keys
  // load <= 0.5; a graph whose index is far beyond the L2 anyway (every probe is a memory round trip, and the
  // rounds of linear probing are what a 64-lane wave waits for) gets load <= 0.25
  uint32_t cap = 64;
  static const int ht_factor = getenv("GRIM_HT_FACTOR") ? atoi(getenv("GRIM_HT_FACTOR")) : 0;
  while (cap < (uint64_t)(ht_factor > 1 ? ht_factor : (d->n_nodes > (1u << 18) ? 4 : 2)) * (uint64_t)d->n_nodes) cap <<= 1;
  std::vector<uint64_t> hk(cap, 0);
  std::vector<uint32_t> hv(cap, 0);
  bool unique_names = true;
  for (uint32_t i = 0; i < d->n_nodes; ++i) {
    uint64_t k = d->node_key[i];
    if (k == 0) { set_err(c, "grim_graph_upload: node with empty key"); delete g; return nullptr; }
    uint32_t h = (uint32_t)host_mix64(k) & (cap - 1);
    while (hk[h] != 0 && hk[h] != k) h = (h + 1) & (cap - 1);
    if (hk[h] == k) unique_names = false;
    hk[h] = k;  // a repeated name keeps the later row, like dict assignment (networkx_graph.py:53)
    hv[h] = i;
  }
  uint32_t max_deg = 0;
  for (uint32_t i = 0; i < d->n_nodes; ++i)
    if (d->a_start[i + 1] > d->a_start[i]) max_deg = std::max(max_deg, d->a_start[i + 1] - d->a_start[i]);
  uint32_t maxlab = 0;
  for (uint32_t m = 0; m < (1u << GRIM_MAXL); ++m) {
    uint32_t n = d->lab_start[m + 1] - d->lab_start[m];
    if (n > maxlab) maxlab = n;
  }
  g->max_label = maxlab;
  DevGraph &D = g->d;
  D.n_nodes = d->n_nodes;
  D.P = d->n_pops;
  D.n_loci = d->n_loci;
  D.full_mask = d->full_mask;
  D.ht_mask = cap - 1;
  D.n_conn = d->n_conn;
  D.node_key = upload(c, g->bufs, d->node_key, d->n_nodes, &g->bytes);
  D.node_mask = upload(c, g->bufs, d->node_mask, d->n_nodes, &g->bytes);
  D.freq = upload(c, g->bufs, d->freq, (size_t)d->n_nodes * d->n_pops, &g->bytes);
  D.a_start = upload(c, g->bufs, d->a_start, (size_t)d->n_nodes + 1, &g->bytes);
  D.a_nbr = upload(c, g->bufs, d->a_nbr, d->n_a_nbr, &g->bytes);
  D.b_conn = upload(c, g->bufs, d->b_conn, (size_t)d->n_nodes * GRIM_MAXL, &g->bytes);
  D.b_start = upload(c, g->bufs, d->b_start, (size_t)d->n_conn + 1, &g->bytes);
  D.b_nbr = upload(c, g->bufs, d->b_nbr, d->n_b_nbr, &g->bytes);
  D.lab_start = upload(c, g->bufs, d->lab_start, (1u << GRIM_MAXL) + 1, &g->bytes);
  D.lab_nodes = upload(c, g->bufs, d->lab_nodes, d->n_nodes, &g->bytes);
  {
    std::vector<uint64_t> lk(d->n_nodes);
    for (uint32_t i = 0; i < d->n_nodes; ++i) lk[i] = d->node_key[d->lab_nodes[i]];
    D.lab_key = upload(c, g->bufs, lk.data(), d->n_nodes, &g->bytes);
  }
  D.scan_ok = (unique_names && max_deg < (1u << 22) && d->n_pops <= 64) ? 1u : 0u;
  D.order_bad = d->label_order_bad;
  {
    std::vector<HtEnt> ht(cap);
    for (uint32_t i = 0; i < cap; ++i) {
      ht[i].key = hk[i];
      ht[i].val = hv[i];
      ht[i].pad = 0;
    }
    D.ht = upload(c, g->bufs, ht.data(), cap, &g->bytes);
  }
  {  // full-haplotype table
    uint32_t nfull = d->lab_start[d->full_mask + 1] - d->lab_start[d->full_mask];
    uint32_t fcap = 64;
    // load <= 1/8: every lane of a wave probes at once and the wave waits for its unluckiest lane, so the
    // number of linear-probing rounds matters more than the table's footprint (1 MB for the CAU graph);
    // measured on the bench kernel: 10.8 / 8.8 / 8.2 / 8.5 us at load 1/2, 1/4, 1/8, 1/16
    static const int fht_factor = getenv("GRIM_FHT_FACTOR") ? atoi(getenv("GRIM_FHT_FACTOR")) : 8;
    while (fcap < (uint64_t)(fht_factor > 1 ? fht_factor : 2) * (uint64_t)nfull) fcap <<= 1;
    std::vector<FullEnt> ft(fcap);
    memset(ft.data(), 0, sizeof(FullEnt) * fcap);
    for (uint32_t i = 0; i < d->n_nodes; ++i) {
      if (d->node_mask[i] != d->full_mask) continue;
      uint64_t k = d->node_key[i];
      uint32_t h = fht_hash(k) & (fcap - 1);
      while (ft[h].key != 0 && ft[h].key != k) h = (h + 1) & (fcap - 1);
      ft[h].key = k;
      ft[h].f0 = d->freq[(size_t)i * d->n_pops];
    }
    D.fht = upload(c, g->bufs, ft.data(), fcap, &g->bytes);
    D.fht_mask = fcap - 1;
  }
  if (!D.fht || !D.node_key || !D.node_mask || !D.freq || !D.a_start || !D.a_nbr || !D.b_conn || !D.b_start || !D.b_nbr ||
      !D.lab_start || !D.lab_nodes || !D.lab_key || !D.ht) {
    set_err(c, "grim_graph_upload: device allocation or copy failed");
    for (void *p : g->bufs) hipFree(p);
    delete g;
    return nullptr;
  }
  return g;
}

extern "C" void grim_graph_free(grim_graph *g) {
  if (!g) return;
  hipSetDevice(g->ctx->device);
  for (void *p : g->bufs) hipFree(p);
  delete g;
}

extern "C" uint64_t grim_graph_device_bytes(const grim_graph *g) { return g ? g->bytes : 0; }


static uint64_t align256(uint64_t x) { return (x + 255) & ~255ull; }

static thread_local int tl_device = -1;
static inline void use_device(int dev) {
  if (tl_device != dev) {
    hipSetDevice(dev);
    tl_device = dev;
  }
}

static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v ? atoi(v) : dflt;
}

uint32_t engine_small_stride(const grim_params *p) { return GRIM_SMALL_ROWS_FIXED + (p->n_results < 16 ? p->n_results : 16); }

uint64_t engine_rows_per_subject(const grim_params *p, uint32_t P) {
  return 2ull * p->n_results + 2ull * (p->n_pop_results < (uint64_t)P * P ? p->n_pop_results : (uint64_t)P * P);
}

static bool dev_realloc(uint8_t *&ptr, uint64_t bytes) {
  if (ptr) hipFree(ptr);
  ptr = nullptr;
  return hipMalloc((void **)&ptr, bytes ? bytes : 256) == hipSuccess;
}
static bool pin_realloc(uint8_t *&ptr, uint64_t bytes) {
  if (ptr) hipHostFree(ptr);
  ptr = nullptr;
  return hipHostMalloc((void **)&ptr, bytes ? bytes : 256, hipHostMallocDefault) == hipSuccess;
}

static std::atomic<uint64_t> g_moved[2];

// lay the arenas out for a batch of at most pl.n_subj subjects and pl.tok_cap tokens; grows them when needed (what
// they held is lost then) and points the device arguments and the host-side pointers at the new places
static bool batch_plan(grim_batch *b, const EnginePlan &pl) {
  const uint64_t n = pl.n_subj ? pl.n_subj : 1;
  uint64_t o = 0;
  auto take = [&](uint64_t bytes) { uint64_t r = o; o = align256(o + bytes); return r; };
  const uint64_t o_state = take(8ull * (GRIM_NCTR + GRIM_NQ / 2));
  // device tokenizer: line records and the chunk's text come first -- a run whose lines are all tokenised on the device
  // copies the arena only up to here
  const bool devtok = pl.text_cap != 0;
  const uint64_t o_lines = take(devtok ? sizeof(LineRec) * n : 0), o_text = take(devtok ? pl.text_cap + 64 : 0);
  const uint64_t o_subj = take(sizeof(grim_subject) * n);
  const uint64_t o_small = take(sizeof(SmallRec) * n);
  const uint64_t o_os = take(4 * n), o_om = take(4 * n), o_og = take(4 * n);
  const uint64_t o_tok = take(2 * ((pl.tok_cap ? pl.tok_cap : 1) + (devtok ? (uint64_t)TOK_LANES * n : 0)));  // device-written tokens behind the host's
  const uint64_t in_bytes = o;
  o = 0;
  const uint64_t small_waves = ((n + GRIM_WG / 32 - 1) / (GRIM_WG / 32)) * (GRIM_WG / 64);
  const uint64_t w_bail = take(4 * n), w_next = take(4 * n), w_ctr = take(4 * (4 * small_waves + 4));  // host- and device-tokenised launches
  const uint64_t w_mid = take(4 * n);
  const uint64_t w_dsmall = take(devtok ? sizeof(SmallRec) * n : 0);
  const uint64_t w_t1 = take(sizeof(TabWork) * 2 * n), w_t2 = take(sizeof(TabWork) * 2 * n);  // a subject queues at most two items
  const uint64_t work_bytes = o;
  o = 0;
  const uint64_t o_res = take(sizeof(grim_subject_result) * n);
  const uint64_t o_rows = take(sizeof(grim_row) * b->row_limit);
  const uint64_t out_bytes = o;
  bool ok = true;
  if (in_bytes > b->in_cap || !b->d_in) {
    const uint64_t want = in_bytes > b->in_cap ? in_bytes : b->in_cap;
    ok = ok && dev_realloc(b->d_in, want) && pin_realloc(b->h_in, want);
    b->in_cap = ok ? want : 0;
  }
  if (work_bytes > b->work_cap || !b->d_work) {
    const uint64_t want = work_bytes > b->work_cap ? work_bytes : b->work_cap;
    ok = ok && dev_realloc(b->d_work, want);
    b->work_cap = ok ? want : 0;
  }
  if (out_bytes > b->out_cap || !b->d_out) {
    const uint64_t want = out_bytes > b->out_cap ? out_bytes : b->out_cap;
    ok = ok && dev_realloc(b->d_out, want);
    b->out_cap = ok ? want : 0;
  }
  if (!ok) return false;
  b->plan = pl;
  b->off_tok = o_tok;
  b->off_rows = o_rows;
  DevArgs &A = b->a;
  A.counters = (unsigned long long *)(b->d_in + o_state);
  A.queue = (uint32_t *)(A.counters + GRIM_NCTR);
  A.row_head = A.queue + 1;
  A.next_count = A.queue + 2;
  A.subj = (const grim_subject *)(b->d_in + o_subj);
  b->d_small = (SmallRec *)(b->d_in + o_small);
  b->d_lines = devtok ? (LineRec *)(b->d_in + o_lines) : nullptr;
  b->d_text = devtok ? b->d_in + o_text : nullptr;
  b->d_dsmall = devtok ? (SmallRec *)(b->d_work + w_dsmall) : nullptr;
  b->off_subj = o_subj;
  b->h.lines = devtok ? (LineRec *)(b->h_in + o_lines) : nullptr;
  b->h.text = devtok ? b->h_in + o_text : nullptr;
  b->d_os = (uint32_t *)(b->d_in + o_os);
  b->d_om = (uint32_t *)(b->d_in + o_om);
  A.order = (const uint32_t *)(b->d_in + o_og);
  A.tok = (const uint16_t *)(b->d_in + o_tok);
  A.bail_list = (uint32_t *)(b->d_work + w_bail);
  A.next_list = (uint32_t *)(b->d_work + w_next);
  {
    static const int no_mid = env_int("GRIM_NO_MID", 0);  // test switch: everything the mid-size kernel takes goes to the general kernel
    A.mid_list = no_mid ? nullptr : (uint32_t *)(b->d_work + w_mid);
  }
  A.small_ctr = (uint32_t *)(b->d_work + w_ctr);
  A.t1_list = (TabWork *)(b->d_work + w_t1);
  A.t2_list = (TabWork *)(b->d_work + w_t2);
  A.res = (grim_subject_result *)(b->d_out + o_res);
  A.rows = (grim_row *)(b->d_out + o_rows);
  A.row_cap = (uint32_t)b->row_limit;
  A.priors = b->d_priors;
  A.next_cap = (uint32_t)n;
  b->h.subj = (grim_subject *)(b->h_in + o_subj);
  b->h.small = (SmallRec *)(b->h_in + o_small);
  b->h.order_s = (uint32_t *)(b->h_in + o_os);
  b->h.order_m = (uint32_t *)(b->h_in + o_om);
  b->h.order_g = (uint32_t *)(b->h_in + o_og);
  b->h.tok = (uint16_t *)(b->h_in + o_tok);
  b->h.res = (grim_subject_result *)b->h_out;            // set for real by the first fetch (the landing area grows on demand)
  b->h.rows = (grim_row *)(b->h_out ? b->h_out + o_rows : nullptr);
  return true;
}

// parameters, graph and the per-workgroup scratch layout of a batch (everything that is not a buffer)
static void batch_init_params(grim_batch *b, const grim_graph *g, const grim_params *p) {
  b->g = g;
  DevArgs &A = b->a;
  A.g = g->d;
  A.prm = *p;
  const uint32_t P = g->d.P;
  b->small_stride = engine_small_stride(p);
  // per-workgroup scratch layout (bytes per slot); the block itself belongs to the context and is bound at run time
  A.pair_cap = GRIM_MAXPH * p->top_n * p->top_n;
  uint32_t tab = 64;
  while (tab < 2 * A.pair_cap) tab <<= 1;
  A.tab_cap = tab;
  A.bset_cap = g->max_label;
  SlotLayout &L = A.lay;
  uint64_t o = 0;
  auto take = [&](uint64_t n) { uint64_t r = o; o = align256(o + n); return r; };
  L.Tp = take(8ull * GRIM_SIDES * GRIM_TOPCAP);
  L.Tm = take(8ull * GRIM_SIDES * GRIM_TOPCAP);
  L.Te = take(4ull * GRIM_SIDES * GRIM_TOPCAP);
  L.k0 = take(8ull * tab);
  L.k1 = take(8ull * tab);
  L.tmin = take(4ull * tab);
  L.tgid = take(4ull * tab);
  L.Useq = take(4ull * A.pair_cap);
  L.Uprob = take(8ull * A.pair_cap);
  L.Uslot = take(4ull * A.pair_cap);
  L.ska = take(8ull * A.pair_cap);
  L.skb = take(8ull * A.pair_cap);
  L.sva = take(4ull * A.pair_cap);
  L.svb = take(4ull * A.pair_cap);
  L.gsum = take(8ull * A.pair_cap);
  L.ghead = take(4ull * A.pair_cap);
  L.gstart = take(4ull * (A.pair_cap + 1));
  L.gcnt = take(4ull * (A.pair_cap > P * P ? A.pair_cap : P * P));
  L.qsum = take(8ull * P * P);
  L.qfirst = take(4ull * P * P);
  L.bset = take(4ull * GRIM_NWAVE * GRIM_MAXL * (uint64_t)A.bset_cap);
  L.comp = take(8ull * GRIM_COMP_CAP);
  {
    uint32_t pc = 64;
    while (pc < 2 * A.bset_cap) pc <<= 1;
    A.proj_cap = pc;
  }
  L.proj_k = take(8ull * GRIM_NWAVE * A.proj_cap);
  L.proj_p = take(4ull * GRIM_NWAVE * A.proj_cap);
  L.rtok = take(2ull * GRIM_RTOK_CAP);
  L.save = take(p->save_mode ? 8ull * GRIM_NWAVE * 2 * GRIM_SAVE_CAP * (P + 1) : 0);
  L.stride = align256(o);
  b->timing = env_int("GRIM_TIMING", 0) != 0;
  A.flags = (env_int("GRIM_TABLES_HBM", 0) ? GRIM_F_TABLES_HBM : 0u) | (env_int("GRIM_NO_NODUP", 0) ? GRIM_F_NO_NODUP : 0u) |
            (env_int("GRIM_NO_SIDEMASK", 0) ? GRIM_F_NO_SIDEMASK : 0u);
  memset(b->acc_ms, 0, sizeof(b->acc_ms));
  b->n_timed = 0;
  b->n_subj = b->n_small = b->n_medium = b->n_general = 0;
  b->rows_used = 0;
}

grim_batch *engine_batch_create(grim_ctx *c, const grim_graph *g, const grim_params *p, uint64_t row_limit, const EnginePlan *plan) {
  if (!c || !g || !p || !plan) return nullptr;
  use_device(c->device);
  if (p->top_n == 0 || p->top_n > GRIM_TOPCAP) { set_err(c, "grim_batch_upload: max_haplotypes_number_in_phase must be 1..128"); return nullptr; }
  if (p->n_ladder < 0 || p->n_ladder > GRIM_MAXLADDER) { set_err(c, "grim_batch_upload: epsilon ladder too long"); return nullptr; }
  if (row_limit > 0x7FFFFFF0ull) row_limit = 0x7FFFFFF0ull;
  if (row_limit < 64) row_limit = 64;
  grim_batch *b = nullptr;
  if (!c->spare.empty()) {  // a batch some earlier stream handed back: its arenas and events are reused
    b = c->spare.back();
    c->spare.pop_back();
    if (b->d_priors) hipFree(b->d_priors);  // sized in P x P units of the graph it served
    if (b->h_priors) hipHostFree(b->h_priors);
    b->d_priors = b->h_priors = nullptr;
    b->priors_cap = b->priors_up = 0;
  } else {
    b = new grim_batch();
    memset((void *)b, 0, sizeof(*b));
    b->ctx = c;
    bool ok = true;
    if (hipHostMalloc((void **)&b->hstate, 8 * (GRIM_NCTR + GRIM_NQ / 2)) != hipSuccess) {
      b->hstate = nullptr;
      ok = false;
    }
    for (int i = 0; i < 18 && ok; ++i) ok = hipEventCreate(&b->ev[i]) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&b->ev_done, hipEventDisableTiming | hipEventReleaseToSystem) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&b->ev_copy, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      set_err(c, "grim_batch: device or pinned-host allocation failed");
      batch_destroy(b);
      return nullptr;
    }
  }
  b->row_limit = row_limit;
  batch_init_params(b, g, p);
  if (!batch_plan(b, *plan)) {
    set_err(c, "grim_batch: device or pinned-host allocation failed");
    batch_destroy(b);
    return nullptr;
  }
  return b;
}

// hand a capacity batch back to its context for the next stream (at most 8 are kept)
void engine_batch_recycle(grim_batch *b) {
  if (!b) return;
  grim_ctx *c = b->ctx;
  use_device(c->device);
  hipStreamSynchronize(c->stream);
  if (b->gexec) {
    hipGraphExecDestroy(b->gexec);
    b->gexec = nullptr;
  }
  b->graph_state = 0;
  if (c->spare.size() >= 8) {
    batch_destroy(b);
    return;
  }
  c->spare.push_back(b);
}

int engine_batch_plan(grim_batch *b, const EnginePlan *plan) {
  if (!b || !plan) return -1;
  use_device(b->ctx->device);
  if (!batch_plan(b, *plan)) {
    set_err(b->ctx, "grim_batch: device or pinned-host allocation failed while growing a batch");
    return -1;
  }
  return 0;
}

void engine_set_error(grim_ctx *c, const char *msg) { set_err(c, msg); }
const EngineHost *engine_batch_host(grim_batch *b) { return b ? &b->h : nullptr; }
uint64_t engine_bytes_moved(const grim_batch *, int dir) { return g_moved[dir ? 1 : 0]; }

struct grim_devdict {
  grim_ctx *ctx;
  DevDict d;
  std::vector<void *> bufs;
};

static void pack_name(const char *p, size_t n, uint32_t (&w)[6]) {
  for (int i = 0; i < 6; ++i) w[i] = 0;
  for (size_t k = 0; k < n && k < GRIM_TOKNAME; ++k) w[k >> 2] |= (uint32_t)(uint8_t)p[k] << (24 - 8 * (k & 3));
}

grim_devdict *engine_devdict_create(grim_ctx *c, const DictSnap *snap) {
  if (!c || !snap) return nullptr;
  use_device(c->device);
  grim_devdict *dd = new grim_devdict();
  dd->ctx = c;
  memset(&dd->d, 0, sizeof(dd->d));
  dd->d.n_loci = snap->n_loci;
  for (uint32_t s = 0; s < GRIM_MAXL; ++s) dd->d.locus_len[s] = 0xFFFFFFFFu;  // matches no name
  for (const DictSnap::Locus &L : snap->loci) {
    if (L.slot >= GRIM_MAXL || L.name.empty() || L.name.size() > 8) continue;  // (longer locus names: host tokenizer)
    uint64_t v = 0;
    for (size_t k = 0; k < L.name.size(); ++k) v |= (uint64_t)(uint8_t)L.name[k] << (56 - 8 * k);
    dd->d.locus[L.slot] = v;
    dd->d.locus_len[L.slot] = (uint32_t)L.name.size();
  }
  bool ok = true;
  for (uint32_t s = 0; s < snap->n_loci && s < GRIM_MAXL && ok; ++s) {
    const uint32_t cnt = snap->base[s];
    uint32_t cap = 16;
    while (cap < 4ull * cnt) cap <<= 1;  // load <= 1/4: a wave waits for its unluckiest lane
    std::vector<DictEnt> tab(cap);
    memset(tab.data(), 0, sizeof(DictEnt) * cap);
    for (uint32_t id = 0; id < cnt; ++id) {
      const sv nm = snap->name(s, id);
      if (nm.empty() || nm.size() > GRIM_TOKNAME) continue;  // never found on the device: such a line goes to the host tokenizer
      DictEnt e;
      pack_name(nm.data(), nm.size(), e.w);
      e.hash = tokname_hash(e.w, (uint32_t)nm.size());
      e.meta = 0x80000000u | ((uint32_t)nm.size() << 16) | (id & 0xFFFFu);
      uint32_t k = e.hash & (cap - 1);
      while (tab[k].meta >> 31) k = (k + 1) & (cap - 1);
      tab[k] = e;
    }
    DictEnt *dt = upload(c, dd->bufs, tab.data(), cap, nullptr);
    ok = dt != nullptr;
    dd->d.tab[s] = dt;
    dd->d.mask[s] = cap - 1;
  }
  if (!ok) {
    set_err(c, "engine_devdict_create: device allocation or copy failed");
    engine_devdict_free(dd);
    return nullptr;
  }
  return dd;
}

void engine_devdict_free(grim_devdict *d) {
  if (!d) return;
  use_device(d->ctx->device);
  for (void *p : d->bufs) hipFree(p);
  delete d;
}

void engine_batch_set_dict(grim_batch *b, const grim_devdict *d) {
  if (b) b->dict = d;
}

uint32_t engine_batch_irregular(const grim_batch *b) { return b ? b->n_irregular : 0; }
uint32_t engine_graph_order_bad(const grim_graph *g) { return g ? g->d.order_bad : 0; }

// After grim_batch_run returned -2: when the PAIR POOL was what ran out and the demand fits `max_records`, the next
// engine_batch_load sizes the pool for it (returns 1: load and run the same subjects again); 0: something else
// overflowed, or the demand is beyond the limit -- split the batch.
int engine_batch_grow_pool(grim_batch *b, uint64_t max_records) {
  if (!b || b->pool_asked <= b->a.ppool_cap) return 0;
  uint64_t want = b->pool_asked + b->pool_asked / 8 + 4096;
  if (want > max_records || want > 0x7FFFFFF0ull) return 0;
  if (want <= b->pool_want) return 0;  // already tried with that much
  b->pool_want = want;
  return 1;
}

// The ROW pool of a batch whose run ran out of rows (not of pairs, buckets or work units): twice as many, up to max_rows.
// 1: grown -- the caller runs the same subjects again (their inputs are still in the pinned arena); 0: not the row pool, or
// already at the bound (the caller halves the range).  engine_batch_hint_rows: the same lesson for a sibling batch of the
// stream before its next load.  Only ever called between runs of the batch.
static int batch_set_rows(grim_batch *b, uint64_t rows) {
  const uint64_t old = b->row_limit;
  b->row_limit = rows;
  const EnginePlan pl = b->plan;
  if (!batch_plan(b, pl)) {
    (void)hipGetLastError();
    b->row_limit = old;
    if (!batch_plan(b, pl)) set_err(b->ctx, "grim_batch: device allocation failed while growing the row pool");
    return 0;
  }
  return 1;
}
int engine_batch_grow_rows(grim_batch *b, uint64_t max_rows) {
  if (!b || b->rows_used < b->a.row_cap || b->row_limit >= max_rows) return 0;
  use_device(b->ctx->device);
  uint64_t want = b->row_limit * 2;
  if (want > max_rows) want = max_rows;
  if (want > 0x7FFFFFF0ull) want = 0x7FFFFFF0ull;
  if (want <= b->row_limit) return 0;
  return batch_set_rows(b, want);
}
uint64_t engine_batch_row_limit(const grim_batch *b) { return b ? b->row_limit : 0; }
void engine_batch_hint_rows(grim_batch *b, uint64_t rows) {
  if (!b || rows <= b->row_limit) return;
  use_device(b->ctx->device);
  (void)batch_set_rows(b, rows);
}

// what a batch learnt about its pair pool (0: the default sizing was enough) / the same lesson for a sibling batch of the
// same stream, which will see the same kind of chunks: its next load sizes the pool accordingly instead of finding out
uint64_t engine_batch_pool_want(const grim_batch *b) { return b ? b->pool_want : 0; }
void engine_batch_hint_pool(grim_batch *b, uint64_t records) {
  if (b && records > b->pool_want) b->pool_want = records;
}

int engine_batch_load(grim_batch *b, const EngineLoad *ld) {
  if (!b || !ld) return -1;
  grim_ctx *c = b->ctx;
  use_device(c->device);
  DevArgs &A = b->a;
  if (ld->n_subj > b->plan.n_subj || ld->tok_used > b->plan.tok_cap) {
    set_err(c, "engine_batch_load: beyond what the batch was planned for");
    return -1;
  }
  hipStream_t st = c->stream;
  const uint32_t P = b->g->d.P;
  const uint64_t PP = (uint64_t)P * P;
  // prior matrices + one all-ones matrix for Plan B's second level (impute.py:1696-1700): only when the set changed
  if (ld->priors && (ld->n_priors != b->priors_up || !b->d_priors)) {
    if (ld->n_priors + 1 > b->priors_cap || !b->d_priors) {
      const uint32_t want = ld->n_priors + 1 > 2 * b->priors_cap ? ld->n_priors + 1 : 2 * b->priors_cap;
      if (b->d_priors) hipFree(b->d_priors);
      if (b->h_priors) hipHostFree(b->h_priors);
      b->d_priors = b->h_priors = nullptr;
      if (hipMalloc((void **)&b->d_priors, 8 * (uint64_t)want * PP) != hipSuccess ||
          hipHostMalloc((void **)&b->h_priors, 8 * (uint64_t)want * PP, hipHostMallocDefault) != hipSuccess) {
        set_err(c, "engine_batch_load: allocation of the prior matrices failed");
        return -1;
      }
      b->priors_cap = want;
    }
    if (ld->n_priors) memcpy(b->h_priors, ld->priors, 8 * (uint64_t)ld->n_priors * PP);
    for (uint64_t k = 0; k < PP; ++k) b->h_priors[(uint64_t)ld->n_priors * PP + k] = 1.0;
    HIPCHK(hipMemcpyAsync(b->d_priors, b->h_priors, 8 * ((uint64_t)ld->n_priors + 1) * PP, hipMemcpyHostToDevice, st), c, -1);
    g_moved[0] += 8 * ((uint64_t)ld->n_priors + 1) * PP;
    b->priors_up = ld->n_priors;
  }
  A.priors = b->d_priors;
  A.ones_prior = b->priors_up;
  if (ld->dev_hi > ld->dev_lo && (!b->dict || !b->d_lines || ld->text_bytes > b->plan.text_cap || ld->dev_hi > ld->n_subj)) {
    set_err(c, "engine_batch_load: the batch was not planned for the device tokenizer");
    return -1;
  }
  b->dev_lo = ld->dev_lo;
  b->dev_hi = ld->dev_hi;
  b->n_dev_lines = ld->dev_hi > ld->dev_lo ? ld->dev_hi - ld->dev_lo : 0;  // record slots of the device tokenizer: one per line of the range
  b->n_irregular = 0;
  b->n_subj = ld->n_subj;
  b->n_small = ld->n_small;
  b->n_medium = ld->n_medium;
  b->n_general = ld->n_general;
  b->last_load = *ld;
  b->last_load.priors = nullptr;
  A.n_medium = ld->n_medium;
  A.n_work = ld->n_general;
  // the run state travels with the input: clean counters and work heads, rows of the half-wave kernel's fixed region taken
  unsigned long long *hs = (unsigned long long *)b->h_in;
  memset(hs, 0, 8 * (GRIM_NCTR + GRIM_NQ / 2));
  // (rows are bump-allocated from 0; the half-wave kernel's staging region is the top of the pool)
  {
    const uint64_t staged = ((uint64_t)b->n_small + b->n_dev_lines) * b->small_stride;
    A.row_cap = (uint32_t)(b->row_limit - (staged <= b->row_limit ? staged : 0));
  }
  // everything up to the last token in use; a load without host-tokenised subjects ends with the chunk's text
  const bool host_subjects = ld->n_small + ld->n_medium + ld->n_general > 0;
  const uint64_t bytes = (!host_subjects && b->n_dev_lines) ? b->off_subj : b->off_tok + 2 * ld->tok_used;
  // the input goes up on the upload stream -- the copy of THIS batch overlaps the kernels of the batch before it -- and the
  // launch stream waits for it (the batch's buffers are its own: nothing else orders the two streams)
  HIPCHK(hipMemcpyAsync(b->d_in, b->h_in, bytes, hipMemcpyHostToDevice, c->up_stream), c, -1);
  HIPCHK(hipEventRecord(b->ev_up, c->up_stream), c, -1);
  b->up_pending = true;  // engine_batch_enqueue orders the launch stream behind it (or finds it done)
  g_moved[0] += bytes;
  {
    // arena of the table kernels: a pair pool of 512 records per subject that can reach them directly, a few for the
    // half-wave kernel's subjects (they get there through Plan B only), and the arrays of the three-kernel path sized by
    // it; running out of any of them is reported like a row-pool overflow
    uint64_t R = (1ull << 20) + 512ull * ((uint64_t)ld->n_medium + ld->n_general) + 16ull * ld->n_small;
    // subjects the host classified as heavy (high ambiguity) can accept tens of thousands of pairs each: half a subject's
    // worst case for the first 256 of them, so that a small batch of heavy subjects does not need a second run
    R += (uint64_t)(ld->n_general < 256u ? ld->n_general : 256u) * (A.pair_cap / 2);
    const char *env_pool = getenv("GRIM_PAIR_POOL");  // tests: start small so that the grow-and-run-again path is taken
    if (env_pool && atol(env_pool) > 0) R = (uint64_t)atol(env_pool);
    if (b->pool_want > R) R = b->pool_want;  // an earlier run of this batch ran out: it said how much it needed
    if (R > 0x7FFFFFF0ull) R = 0x7FFFFFF0ull;
    const uint64_t items = 2ull * ld->n_subj < R / 256 + 1 ? 2ull * ld->n_subj : R / 256 + 1;  // bigger work items at most
    // buckets: a table starts with fewer than n / 64 and the split kernels may re-deal it into twice as many up to three
    // times (n / 8 at most), two tables
    uint64_t cap_b = R / 4 + (items + 1) * ((uint64_t)P * P + 4), cap_u = R / 4 + items * 64 + 1024;
    if (cap_b > R + 4096) cap_b = R + 4096;
    uint64_t o = 0;
    auto take = [&](uint64_t bytes) { uint64_t r = o; o = align256(o + bytes); return r; };
    const uint64_t o_pool = take(sizeof(PairRec) * R), o_aux = take(sizeof(TabAux) * 2 * (uint64_t)(ld->n_subj ? ld->n_subj : 1)),
                   o_boff = take(4 * cap_b), o_cell = take(sizeof(CellRec) * cap_b), o_units = take(sizeof(TabUnit) * cap_u),
                   o_sort = take(4 * 3 * R), o_grp = take(sizeof(GrpRec) * 2 * R), o_prob = take(8 * R);
    if (o > b->pool_bytes || !b->d_pool) {
      HIPCHK(hipStreamSynchronize(st), c, -1);
      if (b->d_pool) hipFree(b->d_pool);
      b->d_pool = nullptr;
      b->pool_bytes = 0;
      if (hipMalloc((void **)&b->d_pool, o) != hipSuccess) {
        (void)hipGetLastError();
        set_err(c, "engine_batch_load: cannot allocate " + std::to_string((unsigned long long)(o >> 20)) + " MiB for the table kernels");
        return -1;
      }
      b->pool_bytes = o;
    }
    b->pool_cap = R;
    A.ppool = (PairRec *)(b->d_pool + o_pool);
    A.ppool_cap = (uint32_t)R;
    A.taux = (TabAux *)(b->d_pool + o_aux);
    A.tboff = (uint32_t *)(b->d_pool + o_boff);
    A.tcell = (CellRec *)(b->d_pool + o_cell);
    A.tboff_cap = (uint32_t)cap_b;
    A.tunits = (TabUnit *)(b->d_pool + o_units);
    A.tunits_cap = (uint32_t)cap_u;
    A.tstride = (uint32_t)R;
    A.psort = (uint32_t *)(b->d_pool + o_sort);
    A.pgrp = (GrpRec *)(b->d_pool + o_grp);
    A.pprob = (double *)(b->d_pool + o_prob);
  }
  const uint32_t per_block = GRIM_WG / 32;
  b->n_host_small_waves = ((b->n_small + per_block - 1) / per_block) * (GRIM_WG / 64);
  b->n_small_waves = b->n_host_small_waves + ((b->n_dev_lines + per_block - 1) / per_block) * (GRIM_WG / 64);
  uint32_t slots = (uint32_t)c->n_cu * GRIM_WG_PER_CU;
  static const int env_slots = env_int("GRIM_SLOTS", 0);
  if (env_slots > 0) slots = (uint32_t)env_slots;
  if (slots > ld->n_subj) slots = ld->n_subj;
  if (slots == 0) slots = 1;
  // (a device-tokenised subject can only ever reach the Plan-B kernel: it is the half-wave kernel's)
  b->n_slots = slots;
  b->scratch_need = (uint64_t)A.lay.stride * slots;
  if (b->gexec) {
    hipGraphExecDestroy(b->gexec);
    b->gexec = nullptr;
  }
  b->graph_state = 0;
  b->rows_used = 0;
  b->small_ctr_pending = false;
  return 0;
}

extern "C" grim_batch *grim_batch_upload(grim_ctx *c, const grim_graph *g, const grim_params *p, const grim_batch_desc *d) {
  if (!c || !g || !p || !d) return nullptr;
  use_device(c->device);
  const uint32_t P = g->d.P;
  // subject classes
  ClassRule rule;
  // (a loci_map that is not alphabetical -- DevGraph::order_bad -- goes through the general kernel only: its look-ups are
  // the ones that reproduce the reference's misses)
  rule.small_ok = (P == 1) && p->opt_threshold > 1 && !getenv("GRIM_NO_SMALL") && g->d.order_bad == 0;
  rule.medium_ok = !getenv("GRIM_NO_MEDIUM") && g->d.order_bad == 0;
  rule.medium_max_cost = getenv("GRIM_MEDIUM_MAXCOST") ? atof(getenv("GRIM_MEDIUM_MAXCOST")) : 0.0;
  rule.graph_loci = g->d.n_loci;
  rule.opt_threshold = p->opt_threshold;
  std::vector<uint32_t> os, om, og;
  for (uint32_t i = 0; i < d->n_subjects; ++i) {
    const int cls = grim_classify(rule, d->subjects[i]);
    (cls == GRIM_CLS_SMALL ? os : cls == GRIM_CLS_MEDIUM ? om : og).push_back(i);
  }
  {  // longest-processing-time-first; stable so equal-cost subjects stay in input order
    std::vector<double> cs(d->n_subjects);
    for (uint32_t i : og) cs[i] = grim_cost(d->subjects[i]);
    std::stable_sort(og.begin(), og.end(), [&](uint32_t a, uint32_t b2) { return cs[a] > cs[b2]; });
  }
  // rows: enough for every subject to fill all four tables (+ the gaps of the one-wave kernel's private row blocks:
  // at most as much again as its subjects use, plus one unfinished GRIM_ROW_GRAB block per resident wave)
  const uint64_t per = engine_rows_per_subject(p, P);
  uint64_t want = per * d->n_subjects + per * om.size() + 2ull * engine_small_stride(p) * os.size() + 1024 +
                  (uint64_t)GRIM_ROW_GRAB * c->n_cu * 32;
  const char *env_rows = getenv("GRIM_ROW_CAP");
  if (env_rows) want = strtoull(env_rows, nullptr, 10);
  if (want > 0x7FFFFFF0ull) want = 0x7FFFFFF0ull;
  EnginePlan plan{d->n_subjects ? d->n_subjects : 1u, d->n_tokens ? d->n_tokens : 1ull};
  grim_batch *b = engine_batch_create(c, g, p, want, &plan);
  if (!b) return nullptr;
  if (d->n_subjects) memcpy(b->h.subj, d->subjects, sizeof(grim_subject) * (size_t)d->n_subjects);
  if (d->n_tokens) memcpy(b->h.tok, d->tokens, 2 * (size_t)d->n_tokens);
  for (size_t k = 0; k < os.size(); ++k) {
    const grim_subject &sj = d->subjects[os[k]];
    grim_small_rec(sj, d->tokens + sj.tok_off, os[k], b->h.small[k]);
    b->h.order_s[k] = os[k];
  }
  if (!om.empty()) memcpy(b->h.order_m, om.data(), 4 * om.size());
  if (!og.empty()) memcpy(b->h.order_g, og.data(), 4 * og.size());
  std::vector<double> one_prior;
  const double *pri = d->priors;
  if (!d->n_priors) {  // no matrix given: subjects index matrix 0
    one_prior.assign((size_t)P * P, 1.0);
    pri = one_prior.data();
  }
  EngineLoad ld{d->n_subjects, (uint32_t)os.size(), (uint32_t)om.size(), (uint32_t)og.size(), d->n_tokens,
                d->n_priors ? d->n_priors : 1u, pri};
  const int lrc = engine_batch_load(b, &ld);  // sets the error text itself
  if (lrc != 0 || hipStreamSynchronize(c->stream) != hipSuccess) {
    if (lrc == 0) set_err(c, "grim_batch_upload: copy failed");
    grim_batch_free(b);
    return nullptr;
  }
  return b;
}

// the context's scratch block: bound when a run's kernels are launched (c->run_mu held).  Kernels use it only while they
// run -- nothing in it outlives a kernel -- and a context's kernels share one stream, so batches in flight at the same
// time may share the block; it is only ever replaced by a bigger one, after the stream has drained.
static int bind_scratch(grim_batch *b) {
  grim_ctx *c = b->ctx;
  if (b->scratch_need > c->scratch_bytes) {
    if (c->scratch) {
      hipStreamSynchronize(c->stream);
      hipFree(c->scratch);
    }
    c->scratch = nullptr;
    c->scratch_bytes = 0;
    if (hipMalloc(&c->scratch, b->scratch_need) != hipSuccess) {
      (void)hipGetLastError();
      set_err(c, "grim_batch_run: cannot allocate " + std::to_string((unsigned long long)(b->scratch_need >> 20)) + " MiB of scratch");
      return -1;
    }
    c->scratch_bytes = b->scratch_need;
  }
  b->a.scratch = (uint8_t *)c->scratch;
  b->a.wctr = c->wctr;
  return 0;
}

// The table kernels: the one-wave kernel for work items of up to GRIM_TAB_T1_MAX pairs; split, bucket and merge kernel
// for the rest.  All read their list lengths on the device (the kernels before them in the stream wrote them) and come
// back at once when there is nothing to do.  start/stop: timing mode, one interval around the four.
static void enqueue_tables(grim_batch *b, hipEvent_t start, hipEvent_t stop) {
  grim_ctx *c = b->ctx;
  DevArgs &A = b->a;
  const uint32_t cand = b->n_subj;
  uint32_t g1 = (uint32_t)c->n_cu * 14u, g2 = (uint32_t)c->n_cu * GRIM_TAB_WG_PER_CU;
  if (g1 > cand) g1 = cand;
  if (g2 > b->n_slots) g2 = b->n_slots;  // one scratch slot per workgroup
  if (g1 == 0) g1 = 1;
  if (g2 == 0) g2 = 1;
  // resident waves of the bucket kernel (registers or LDS, whichever binds): its units are dealt round robin over the grid, a
  // block that had to wait for a free slot would run its share after everybody else's
  static int bucket_per_cu = 0;
  if (bucket_per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, grim_tables_bucket_kernel, 64, 0) != hipSuccess || nb <= 0) {
      (void)hipGetLastError();
      nb = 8;
    }
    bucket_per_cu = nb > 32 ? 32 : nb;
  }
  const uint32_t g3 = (uint32_t)c->n_cu * (uint32_t)bucket_per_cu;
  // items of up to TW_MAXN pairs: a wave each, twelve split waves (LDS) / sixteen merge waves (registers) per CU
  uint32_t g4 = (uint32_t)c->n_cu * (160u * 1024u / (uint32_t)sizeof(SplitWave)), g5 = (uint32_t)c->n_cu * 16u;
  if (g4 > cand) g4 = cand;
  if (g5 > cand) g5 = cand;
  if (g4 == 0) g4 = 1;
  if (g5 == 0) g5 = 1;
  (void)hipMemsetAsync(c->wctr, 0, 4 * GRIM_WCTR_WORDS, c->stream);
  if (start && stop) {
    hipExtLaunchKernelGGL(grim_tables_wave_kernel, dim3(g1), dim3(64), 0, c->stream, start, nullptr, 0, A);
  } else {
    hipLaunchKernelGGL(grim_tables_wave_kernel, dim3(g1), dim3(64), 0, c->stream, A);
  }
  hipLaunchKernelGGL(grim_tables_split_wave_kernel, dim3(g4), dim3(64), 0, c->stream, A);
  hipLaunchKernelGGL(grim_tables_split_kernel, dim3(g2), dim3(GRIM_WG), 0, c->stream, A);
  hipLaunchKernelGGL(grim_tables_bucket_kernel, dim3(g3), dim3(64), 0, c->stream, A);
  hipLaunchKernelGGL(grim_tables_merge_wave_kernel, dim3(g5), dim3(64), 0, c->stream, A);
  if (start && stop) {
    hipExtLaunchKernelGGL(grim_tables_merge_kernel, dim3(g2), dim3(GRIM_WG), 0, c->stream, nullptr, stop, 0, A);
  } else {
    hipLaunchKernelGGL(grim_tables_merge_kernel, dim3(g2), dim3(GRIM_WG), 0, c->stream, A);
  }
}

// Stage 1 of a run: half-wave kernel, one-wave kernel, general plan-A kernel, finish kernel (state to the pinned
// host copy, device copy reset for the next run).  Default: launched directly.  GRIM_GRAPH=1: captured once per batch and
// replayed as ONE hipGraph launch (no event nodes: they carry no timestamps when replayed on this runtime).
// Timing mode: launched directly, every kernel bracketed by its own start/stop events (hipExtLaunchKernelGGL).
static int enqueue_stage1(grim_batch *b, bool timing) {
  grim_ctx *c = b->ctx;
  DevArgs &A = b->a;
  const uint32_t per_block = GRIM_WG / 32;
  // staging region of the half-wave kernel's rows, at the top of the pool: host-tokenised subjects first, then one slot per
  // line of the device tokenizer
  const uint32_t stage0 = (uint32_t)b->row_limit - (b->n_small + b->n_dev_lines) * b->small_stride;
  const bool both = b->n_small && b->n_dev_lines;  // (timing mode brackets the device-tokenised launch when there are two)
  if (b->n_small) {
    const dim3 grid((b->n_small + per_block - 1) / per_block), block(GRIM_WG);
    if (timing && !both)
      hipExtLaunchKernelGGL(grim_plan_a_small_kernel, grid, block, 0, c->stream, b->ev[3], b->ev[5], 0, A,
                            (const SmallRec *)b->d_small, b->n_small, stage0, b->small_stride);
    else
      hipLaunchKernelGGL(grim_plan_a_small_kernel, grid, block, 0, c->stream, A, (const SmallRec *)b->d_small, b->n_small, stage0,
                         b->small_stride);
    const dim3 cgrid((b->n_small + 63) / 64), cblock(64);
    if (timing && !both)
      hipExtLaunchKernelGGL(grim_small_compact_kernel, cgrid, cblock, 0, c->stream, b->ev[12], b->ev[13], 0, A,
                            (const uint32_t *)b->d_os, (const SmallRec *)nullptr, b->n_small, stage0, b->small_stride);
    else
      hipLaunchKernelGGL(grim_small_compact_kernel, cgrid, cblock, 0, c->stream, A, (const uint32_t *)b->d_os, (const SmallRec *)nullptr,
                         b->n_small, stage0, b->small_stride);
  }
  if (b->n_dev_lines) {
    // GL strings -> half-wave records on the device, then the half-wave kernel over one record slot per line
    DevTok T;
    T.text = b->d_text;
    T.lines = b->d_lines;
    T.lo = b->dev_lo;
    T.hi = b->dev_hi;
    T.dict = b->dict->d;
    T.dsmall = b->d_dsmall;
    T.subj = const_cast<grim_subject *>(A.subj);
    T.tok = const_cast<uint16_t *>(A.tok);
    T.tok_base = (uint32_t)(b->plan.tok_cap ? b->plan.tok_cap : 1);
    T.graph_loci = A.g.n_loci;
    const dim3 tgrid((b->n_dev_lines + (GRIM_WG / 64) * TOK_LINES_PER_WAVE - 1) / ((GRIM_WG / 64) * TOK_LINES_PER_WAVE)), block(GRIM_WG);
    if (timing)
      hipExtLaunchKernelGGL(grim_tokenize_kernel, tgrid, block, 0, c->stream, b->ev[14], b->ev[15], 0, A, T);
    else
      hipLaunchKernelGGL(grim_tokenize_kernel, tgrid, block, 0, c->stream, A, T);
    DevArgs A2 = A;
    A2.small_ctr = A.small_ctr + 2 * b->n_host_small_waves;
    const uint32_t stage_d = stage0 + b->n_small * b->small_stride;
    const dim3 grid((b->n_dev_lines + per_block - 1) / per_block);
    if (timing)
      hipExtLaunchKernelGGL(grim_plan_a_small_kernel, grid, block, 0, c->stream, b->ev[3], b->ev[5], 0, A2,
                            (const SmallRec *)b->d_dsmall, b->n_dev_lines, stage_d, b->small_stride);
    else
      hipLaunchKernelGGL(grim_plan_a_small_kernel, grid, block, 0, c->stream, A2, (const SmallRec *)b->d_dsmall, b->n_dev_lines, stage_d,
                         b->small_stride);
    const dim3 cgrid((b->n_dev_lines + 63) / 64), cblock(64);
    if (timing)
      hipExtLaunchKernelGGL(grim_small_compact_kernel, cgrid, cblock, 0, c->stream, b->ev[12], b->ev[13], 0, A, (const uint32_t *)nullptr,
                            (const SmallRec *)b->d_dsmall, b->n_dev_lines, stage_d, b->small_stride);
    else
      hipLaunchKernelGGL(grim_small_compact_kernel, cgrid, cblock, 0, c->stream, A, (const uint32_t *)nullptr, (const SmallRec *)b->d_dsmall,
                         b->n_dev_lines, stage_d, b->small_stride);
  }
  if (b->n_medium) {
    static const int waves_per_cu = env_int("GRIM_MEDIUM_WAVES", GRIM_MEDIUM_WAVES_PER_CU);
    uint32_t grid = (uint32_t)c->n_cu * (uint32_t)(waves_per_cu > 0 ? waves_per_cu : GRIM_MEDIUM_WAVES_PER_CU);
    if (grid > b->n_medium) grid = b->n_medium;
    if (timing)
      hipExtLaunchKernelGGL(grim_plan_a_medium_kernel, dim3(grid), dim3(64), 0, c->stream, b->ev[0], b->ev[1], 0, A,
                            (const uint32_t *)b->d_om, b->n_medium, A.bail_list);
    else
      hipLaunchKernelGGL(grim_plan_a_medium_kernel, dim3(grid), dim3(64), 0, c->stream, A, (const uint32_t *)b->d_om,
                         b->n_medium, A.bail_list);
  }
  if (A.mid_list && b->n_general + b->n_medium) {
    const uint32_t want = b->n_general + b->n_medium;
    uint32_t grid = (uint32_t)c->n_cu * GRIM_MID_WG_PER_CU;
    if (grid > want) grid = want;
    if (timing)
      hipExtLaunchKernelGGL(grim_plan_a_mid_kernel, dim3(grid), dim3(GRIM_WG), 0, c->stream, b->ev[16], b->ev[17], 0, A);
    else
      hipLaunchKernelGGL(grim_plan_a_mid_kernel, dim3(grid), dim3(GRIM_WG), 0, c->stream, A);
  }
  if (b->n_general + b->n_medium) {
    uint32_t want = b->n_general + b->n_medium;
    const dim3 grid(b->n_slots < want ? b->n_slots : want), block(GRIM_WG);
    if (timing)
      hipExtLaunchKernelGGL(grim_plan_a_kernel, grid, block, 0, c->stream, b->ev[6], b->ev[7], 0, A);
    else
      hipLaunchKernelGGL(grim_plan_a_kernel, grid, block, 0, c->stream, A);
  }
  hipLaunchKernelGGL(grim_finish_kernel, dim3(1), dim3(GRIM_WG), 0, c->stream, A.counters, b->hstate, 0u, 0);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int grim_batch_set_timing(grim_batch *b, int on) {
  if (!b) return -1;
  b->timing = on != 0;
  memset(b->acc_ms, 0, sizeof(b->acc_ms));
  b->n_timed = 0;
  return 0;
}

// A run in two halves, so that a caller can keep the device busy: engine_batch_enqueue launches stage 1 (half-wave, one-wave,
// general kernel, finish kernel) behind whatever the context's stream holds and returns at once; engine_batch_wait waits for
// that stage, launches stage 2 (Plan B / C, the table kernels) when the state block says subjects or accepted pairs are
// waiting, and returns what grim_batch_run returns.  The two may be called from different threads (one after the other).
// has the input copy of the last engine_batch_load arrived?  (any thread; never blocks)
int engine_batch_upload_done(grim_batch *b) {
  if (!b || !b->up_pending) return 1;
  use_device(b->ctx->device);
  if (hipEventQuery(b->ev_up) == hipSuccess) return 1;
  (void)hipGetLastError();
  return 0;
}

int engine_batch_enqueue(grim_batch *b) {
  if (!b) return -1;
  grim_ctx *c = b->ctx;
  use_device(c->device);
  if (((uint64_t)b->n_small + b->n_dev_lines) * b->small_stride > b->row_limit) {
    // the half-wave kernel's rows have fixed places in the staging region at the top of the pool: they must all exist
    b->rows_used = 0;
    set_err(c, "grim_batch_run: output row pool smaller than the half-wave kernel's fixed region");
    return -2;
  }
  b->ms_s = b->ms_a = b->ms_g = b->ms_m = b->ms_t = b->ms_c = b->ms_b = b->ms_k = b->ms_d = 0;
  std::lock_guard<std::mutex> lk(c->run_mu);
  if (bind_scratch(b) != 0) return -1;
  if (b->up_pending) {
    // The batch's input went up on the upload stream.  A wait for its event in the launch stream costs the stream ~9 us
    // (a barrier packet on a signal of another queue: 46 -> 55 us per four 10-us kernels, profiles/r4_notes.md); when the
    // copy is over already -- the stream pipeline launches a chunk after it has sent the NEXT chunk's input up -- the
    // host has seen so and a launch from here on is ordered behind it without any packet.
    b->up_pending = false;
    if (hipEventQuery(b->ev_up) != hipSuccess) {
      (void)hipGetLastError();
      HIPCHK(hipStreamWaitEvent(c->stream, b->ev_up, 0), c, -1);
    }
  }
  if (b->timing) {
    if (enqueue_stage1(b, true) != 0) {
      set_err(c, "grim_batch_run: kernel launch failed");
      return -1;
    }
  } else {
    if (b->graph_state == 0) {
      b->graph_state = -1;
      // measured on MI355X / ROCm 7.2 (tools/step_time.py, 10k-subject batch): replaying the captured stage costs
      // 26.9 us per synchronous run, launching its two kernels directly 21.8 us -- so the replay is opt-in
      static const int use_graph = env_int("GRIM_GRAPH", 0);
      if (use_graph && hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        int rc = enqueue_stage1(b, false);
        hipGraph_t gr = nullptr;
        hipError_t e = hipStreamEndCapture(c->stream, &gr);
        if (rc == 0 && e == hipSuccess && gr && hipGraphInstantiate(&b->gexec, gr, nullptr, nullptr, 0) == hipSuccess)
          b->graph_state = 1;
        if (gr) hipGraphDestroy(gr);
        (void)hipGetLastError();
      }
    }
    if (b->graph_state == 1) {
      HIPCHK(hipGraphLaunch(b->gexec, c->stream), c, -1);
    } else if (enqueue_stage1(b, false) != 0) {
      set_err(c, "grim_batch_run: kernel launch failed");
      return -1;
    }
  }
  HIPCHK(hipEventRecord(b->ev_done, c->stream), c, -1);
  b->enqueued = true;
  return 0;
}

int engine_batch_wait(grim_batch *b) {
  if (!b) return -1;
  grim_ctx *c = b->ctx;
  use_device(c->device);
  DevArgs &A = b->a;
  if (!b->enqueued) {
    set_err(c, "engine_batch_wait: nothing in flight");
    return -1;
  }
  b->enqueued = false;
  HIPCHK(hipEventSynchronize(b->ev_done), c, -1);
  if (b->timing) {
    if (b->n_small || b->n_dev_lines) HIPCHK(hipEventElapsedTime(&b->ms_s, b->ev[3], b->ev[5]), c, -1);
    if (b->n_medium) HIPCHK(hipEventElapsedTime(&b->ms_m, b->ev[0], b->ev[1]), c, -1);
    if (b->n_general + b->n_medium) HIPCHK(hipEventElapsedTime(&b->ms_g, b->ev[6], b->ev[7]), c, -1);
    if (A.mid_list && b->n_general + b->n_medium) HIPCHK(hipEventElapsedTime(&b->ms_d, b->ev[16], b->ev[17]), c, -1);
    if (b->n_small || b->n_dev_lines) HIPCHK(hipEventElapsedTime(&b->ms_c, b->ev[12], b->ev[13]), c, -1);
    if (b->n_dev_lines) HIPCHK(hipEventElapsedTime(&b->ms_k, b->ev[14], b->ev[15]), c, -1);
    b->ms_a = b->ms_s + b->ms_m + b->ms_g + b->ms_d;
  }
  uint32_t head[GRIM_NQ];
  memcpy(head, b->hstate + GRIM_NCTR, 4 * GRIM_NQ);
  // ---- stage 2: Plan B / C when the first stage left subjects for it, then the table kernels ONCE over the accepted
  // pairs both stages queued (a round after each stage cost their tails twice) -------------------------------------
  const bool run_b = A.prm.planb && head[2] + head[6] > 0;
  if (run_b || head[2] + head[6] + head[9] + head[10] > 0) {
    {
      std::lock_guard<std::mutex> lk(c->run_mu);
      if (bind_scratch(b) != 0) return -1;
      if (run_b) {
        uint32_t grid = b->n_slots < head[2] + head[6] ? b->n_slots : head[2] + head[6];
        if (grim_launch_plan_b(A, grid, c->stream, b->timing ? b->ev[4] : nullptr, b->timing ? b->ev[2] : nullptr) != 0) {
          set_err(c, "grim_batch_run: plan-B launch failed");
          return -1;
        }
      }
      enqueue_tables(b, b->timing ? b->ev[10] : nullptr, b->timing ? b->ev[11] : nullptr);
      hipLaunchKernelGGL(grim_finish_kernel, dim3(1), dim3(GRIM_WG), 0, c->stream, A.counters, b->hstate, 0u, 1);
      HIPCHK(hipGetLastError(), c, -1);
      HIPCHK(hipEventRecord(b->ev_done, c->stream), c, -1);
    }
    HIPCHK(hipEventSynchronize(b->ev_done), c, -1);
    if (b->timing) {
      float t2 = 0;
      if (run_b) HIPCHK(hipEventElapsedTime(&b->ms_b, b->ev[4], b->ev[2]), c, -1);
      HIPCHK(hipEventElapsedTime(&t2, b->ev[10], b->ev[11]), c, -1);
      b->ms_t += t2;
    }
    memcpy(head, b->hstate + GRIM_NCTR, 4 * GRIM_NQ);
  }
  if (b->timing) {
    const double v[10] = {(double)b->ms_a + b->ms_b + b->ms_t + b->ms_c + b->ms_k, b->ms_a, b->ms_b, b->ms_s, b->ms_g, b->ms_m, b->ms_t, b->ms_c, b->ms_k, b->ms_d};
    for (int k = 0; k < 10; ++k) b->acc_ms[k] += v[k];
    b->n_timed++;
  }
  static const int dbg_classes = env_int("GRIM_DEBUG_CLASSES", 0);
  if (dbg_classes)
    fprintf(stderr, "grim classes: mid-size kernel took %u, handed %u to the general kernel\n", head[16], head[17]);
  if (dbg_classes)
    fprintf(stderr, "grim classes: small %u medium %u general %u | medium->general %u (+%u heavier), to plan B %u (+%u heavy) | table items %u one-wave, %u bigger (%u work units), %u pair records; workgroup merge: %u items (%u with an overflowed bucket), up to %u pairs; workgroup split: %u items | stage 1 %s\n",
            b->n_small, b->n_medium, b->n_general, head[5], head[7], head[2], head[6], head[9], head[10], head[14], head[8], head[21], head[22], head[23], head[12],
            b->graph_state == 1 ? "replayed as a hipGraph" : "launched directly");
  memcpy(b->counters, b->hstate, 64);
  b->small_ctr_pending = b->n_small + b->n_dev_lines > 0;
  for (int sh = 0; sh < 64; ++sh)
    for (int k = 0; k < 3; ++k) b->counters[k] += b->hstate[8 + 4 * sh + k];
  b->rows_used = head[1];
  b->pool_asked = head[8];
  b->n_irregular = head[GRIM_Q_IRREGULAR];
#ifdef GRIM_STAMPS
  fprintf(stderr, "grim stamps (us):");
  for (int k = 0; k < 16; ++k) fprintf(stderr, " [%d]%.0f", k, b->hstate[GRIM_STAMP_BASE + k] / 100.0);
  fprintf(stderr, "\n");
  fprintf(stderr, "grim mid-size kernel stage us (setup, probes, rows, entries, lists, ladder, first pass, final pass, emit):");
  for (int k = 0; k < 9; ++k) fprintf(stderr, " %.0f", b->hstate[GRIM_MID_BASE + k] / 100.0);
  fprintf(stderr, "\ngrim mid-size kernel hand-overs (branch/candidates, hits, rows, entries, lists, pairs, dedup):");
  for (int k = 0; k < 7; ++k) fprintf(stderr, " %llu", b->hstate[GRIM_MID_BASE + 16 + k]);
  fprintf(stderr, "\n");
  static const char *hname[8] = {"0", "1", "2", "3 workgroup-merge items by pairs", "4 workgroup-split items by pairs",
                                 "5 workgroup-split us by pairs", "6 workgroup-merge items by genotype groups",
                                 "7 workgroup-merge us by pairs"};
  for (int h = 0; h < 8; ++h) {
    fprintf(stderr, "grim hist %s:", hname[h]);
    for (int k = 0; k < 24; ++k) fprintf(stderr, " %llu", b->hstate[GRIM_HIST_BASE + 24 * h + k]);
    fprintf(stderr, "\n");
  }
#endif
  if (b->counters[4] != 0 || head[1] > A.row_cap) {
    if (b->rows_used > A.row_cap) b->rows_used = A.row_cap;
    set_err(c, "grim_batch_run: output row pool exhausted (raise GRIM_ROW_CAP or lower the batch size)");
    return -2;
  }
  return 0;
}

extern "C" int grim_batch_run(grim_batch *b) {
  if (!b) return -1;
  int rc = engine_batch_enqueue(b);
  if (rc == 0) rc = engine_batch_wait(b);
  if (rc == -2 && !b->in_retry && engine_batch_grow_pool(b, 256ull << 20)) {
    // the accepted pairs outgrew the table kernels' pool: size it for what this run asked for and run again (the
    // inputs are still in the pinned arena; the prior matrices stay where they are)
    b->in_retry = true;
    EngineLoad ld = b->last_load;
    rc = engine_batch_load(b, &ld);
    if (rc == 0) rc = grim_batch_run(b);
    b->in_retry = false;
  }
  return rc;
}

extern "C" int grim_batch_run_repeat(grim_batch *b, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) {
    const int rc = grim_batch_run(b);
    if (rc != 0) return rc;
  }
  return 0;
}

extern "C" double grim_batch_kernel_ms(const grim_batch *b, int which) {
  if (!b) return 0.0;
  if (which & 0x10) {  // mean over the timed runs since grim_batch_set_timing(b, 1)
    const int k = which & 0xF;
    return (k < 10 && b->n_timed) ? b->acc_ms[k] / b->n_timed : 0.0;
  }
  if (which == 1) return b->ms_a;
  if (which == 2) return b->ms_b;
  if (which == 3) return b->ms_s;
  if (which == 4) return b->ms_g;
  if (which == 5) return b->ms_m;
  if (which == 6) return b->ms_t;
  if (which == 7) return b->ms_c;
  if (which == 8) return b->ms_k;
  if (which == 9) return b->ms_d;
  return (double)b->ms_a + (double)b->ms_b + (double)b->ms_t + (double)b->ms_c + (double)b->ms_k;
}

extern "C" int grim_batch_counters(const grim_batch *cb, uint64_t out[4]) {
  if (!cb) return -1;
  grim_batch *b = const_cast<grim_batch *>(cb);  // lazily folds the half-wave kernel's per-wave counts in
  if (b->small_ctr_pending) {
    grim_ctx *c = b->ctx;
    use_device(c->device);
    std::vector<uint32_t> h(2 * (size_t)b->n_small_waves);
    HIPCHK(hipMemcpy(h.data(), b->a.small_ctr, 4 * h.size(), hipMemcpyDeviceToHost), c, -1);
    for (size_t i = 0; i < h.size(); i += 2) {
      b->counters[0] += h[i];
      b->counters[2] += h[i + 1];
    }
    b->small_ctr_pending = false;
  }
  out[0] = b->counters[0];
  out[1] = b->counters[1];
  out[2] = b->counters[2];
  out[3] = b->rows_used;
  return 0;
}

extern "C" uint32_t grim_batch_total_rows(const grim_batch *b) { return b ? b->rows_used : 0; }

static int batch_fetch_on(grim_batch *b, uint32_t res_lo, uint32_t res_hi, grim_row *rows_dst, hipStream_t st);
int engine_batch_fetch(grim_batch *b, uint32_t res_lo, uint32_t res_hi, grim_row *rows_dst) {
  return b ? batch_fetch_on(b, res_lo, res_hi, rows_dst, b->ctx->stream) : -1;
}
// the whole batch's results over the context's copy stream: the caller has synchronised the kernels (grim_batch_run
// returned) and may run the NEXT batch's kernels while this copy is in flight; any thread
int engine_batch_fetch_async(grim_batch *b) { return b ? batch_fetch_on(b, 0, b->n_subj, nullptr, b->ctx->copy_stream) : -1; }
// the same in two halves: the copy is queued on the copy stream with an event behind it (issue), another thread waits for
// that event (wait) -- so the thread that watches the kernels is not held up by a PCIe transfer
static int batch_fetch_on(grim_batch *b, uint32_t res_lo, uint32_t res_hi, grim_row *rows_dst, hipStream_t st, bool wait);
// GRIM_DEBUG_COPYTIME (diagnostic): the D2H's own duration by a pair of timing events on the copy stream
static std::atomic<uint64_t> g_ct_ns, g_ct_n;
static hipEvent_t g_ct_ev[2][64];
static std::atomic<uint32_t> g_ct_k;
static int ct_on() {
  static const int on = [] {
    const int v = env_int("GRIM_DEBUG_COPYTIME", 0);
    if (v) {
      for (int i = 0; i < 64; ++i) { (void)hipEventCreate(&g_ct_ev[0][i]); (void)hipEventCreate(&g_ct_ev[1][i]); }
      atexit([] { if (g_ct_n.load()) fprintf(stderr, "grim: D2H on the copy stream: %.1f us average over %llu copies\n", g_ct_ns.load() / 1e3 / (double)g_ct_n.load(), (unsigned long long)g_ct_n.load()); });
    }
    return v;
  }();
  return on;
}
int engine_batch_fetch_issue(grim_batch *b) {
  if (!b) return -1;
  if (ct_on()) {
    const uint32_t k = g_ct_k.fetch_add(1) & 63;
    b->ct_slot = (int)k;
    (void)hipEventRecord(g_ct_ev[0][k], b->ctx->copy_stream);
    if (batch_fetch_on(b, 0, b->n_subj, nullptr, b->ctx->copy_stream, false) != 0) return -1;
    (void)hipEventRecord(g_ct_ev[1][k], b->ctx->copy_stream);
    HIPCHK(hipEventRecord(b->ev_copy, b->ctx->copy_stream), b->ctx, -1);
    return 0;
  }
  if (batch_fetch_on(b, 0, b->n_subj, nullptr, b->ctx->copy_stream, false) != 0) return -1;
  if (!b->fetch_hsa) HIPCHK(hipEventRecord(b->ev_copy, b->ctx->copy_stream), b->ctx, -1);
  return 0;
}
int engine_batch_fetch_wait(grim_batch *b) {
  if (!b) return -1;
  use_device(b->ctx->device);
  if (b->fetch_hsa) {
    b->fetch_hsa = false;
    if (grim_sdma_wait(b->ctx->sdma, b->sdma_job) != 0) {
      set_err(b->ctx, "engine_batch_fetch: the SDMA copy of the results failed");
      return -1;
    }
    return 0;
  }
  HIPCHK(hipEventSynchronize(b->ev_copy), b->ctx, -1);
  if (ct_on() && b->ct_slot >= 0) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, g_ct_ev[0][b->ct_slot], g_ct_ev[1][b->ct_slot]) == hipSuccess) { g_ct_ns += (uint64_t)(ms * 1e6); ++g_ct_n; }
    b->ct_slot = -1;
  }
  return 0;
}
static int batch_fetch_on(grim_batch *b, uint32_t res_lo, uint32_t res_hi, grim_row *rows_dst, hipStream_t st) {
  return batch_fetch_on(b, res_lo, res_hi, rows_dst, st, true);
}
static int batch_fetch_on(grim_batch *b, uint32_t res_lo, uint32_t res_hi, grim_row *rows_dst, hipStream_t st, bool wait) {
  if (!b) return -1;
  grim_ctx *c = b->ctx;
  use_device(c->device);
  b->fetch_hsa = false;
  if (res_hi > b->n_subj) res_hi = b->n_subj;
  // the pinned landing area mirrors the out arena as far as it is used; it grows on demand
  const uint64_t need = b->off_rows + sizeof(grim_row) * (uint64_t)(rows_dst ? 0 : b->rows_used);
  if (need > b->h_out_cap || !b->h_out) {
    uint64_t n = b->h_out_cap ? b->h_out_cap : (1u << 20);
    while (n < need) n *= 2;
    if (n > b->out_cap) n = b->out_cap;
    if (n < need) n = need;
    if (!pin_realloc(b->h_out, n)) {
      b->h_out_cap = 0;
      set_err(c, "engine_batch_fetch: pinned allocation failed");
      return -1;
    }
    b->h_out_cap = n;
  }
  b->h.res = (grim_subject_result *)b->h_out;
  b->h.rows = (grim_row *)(b->h_out + b->off_rows);
  const bool whole = res_lo == 0 && res_hi == b->n_subj && !rows_dst;
  if (whole && b->off_rows - sizeof(grim_subject_result) * (uint64_t)b->n_subj < 4096) {
    // headers and rows are (nearly) back to back: one copy
    const uint64_t bytes = b->rows_used ? b->off_rows + sizeof(grim_row) * (uint64_t)b->rows_used : sizeof(grim_subject_result) * (uint64_t)b->n_subj;
    const uint64_t n16 = (bytes + 15) / 16;  // (both arenas are 256-byte granular: the last sixteen bytes exist)
    if (bytes && !wait && c->export_mode == 2 && st == c->copy_stream) {
      // an SDMA engine of our own (grim_sdma.h); the caller has waited for the kernels that wrote the arena
      if (!b->sdma_job && grim_sdma_job_create(c->sdma, &b->sdma_job) != 0) b->sdma_job = 0;
      if (b->sdma_job && grim_sdma_d2h_issue(c->sdma, b->sdma_job, b->h_out, b->d_out, bytes) == 0) b->fetch_hsa = true;
    }
    const int by_kernel = c->export_mode == 1;
    if (b->fetch_hsa) {
    } else if (bytes && by_kernel && 16 * n16 <= b->h_out_cap && 16 * n16 <= b->out_cap) {
      uint32_t grid = (uint32_t)((n16 + 1023) / 1024);
      if (grid > 32) grid = 32;
      hipLaunchKernelGGL(grim_export_kernel, dim3(grid), dim3(256), 0, st, (uint4 *)b->h_out, (const uint4 *)b->d_out, n16);
      HIPCHK(hipGetLastError(), c, -1);
    } else if (bytes) {
      HIPCHK(hipMemcpyAsync(b->h_out, b->d_out, bytes, hipMemcpyDeviceToHost, st), c, -1);
    }
    g_moved[1] += bytes;
  } else {
    if (res_hi > res_lo) {
      const uint64_t o = sizeof(grim_subject_result) * (uint64_t)res_lo, n = sizeof(grim_subject_result) * (uint64_t)(res_hi - res_lo);
      HIPCHK(hipMemcpyAsync(b->h_out + o, b->d_out + o, n, hipMemcpyDeviceToHost, st), c, -1);
      g_moved[1] += n;
    }
    if (b->rows_used) {
      void *dst = rows_dst ? (void *)rows_dst : (void *)(b->h_out + b->off_rows);
      HIPCHK(hipMemcpyAsync(dst, b->d_out + b->off_rows, sizeof(grim_row) * (uint64_t)b->rows_used, hipMemcpyDeviceToHost, st), c, -1);
      g_moved[1] += sizeof(grim_row) * (uint64_t)b->rows_used;
    }
  }
  if (wait) HIPCHK(hipStreamSynchronize(st), c, -1);
  return 0;
}

extern "C" int grim_batch_results(grim_batch *b, grim_subject_result *res, grim_row *rows) {
  if (!b) return -1;
  grim_ctx *c = b->ctx;
  use_device(c->device);
  if (b->n_subj) HIPCHK(hipMemcpy(res, b->a.res, sizeof(grim_subject_result) * (size_t)b->n_subj, hipMemcpyDeviceToHost), c, -1);
  if (b->rows_used) HIPCHK(hipMemcpy(rows, b->a.rows, sizeof(grim_row) * (size_t)b->rows_used, hipMemcpyDeviceToHost), c, -1);
  return 0;
}

extern "C" void grim_batch_free(grim_batch *b) { batch_destroy(b); }

static void batch_destroy(grim_batch *b) {
  if (!b) return;
  use_device(b->ctx->device);
  hipStreamSynchronize(b->ctx->stream);
  for (int i = 0; i < 18; ++i)
    if (b->ev[i]) hipEventDestroy(b->ev[i]);
  if (b->ev_done) hipEventDestroy(b->ev_done);
  if (b->ev_copy) hipEventDestroy(b->ev_copy);
  if (b->fetch_hsa && b->ctx->sdma) (void)grim_sdma_wait(b->ctx->sdma, b->sdma_job);  // (an abandoned stream: the copy still writes h_out)
  if (b->sdma_job && b->ctx->sdma) grim_sdma_job_destroy(b->ctx->sdma, b->sdma_job);
  if (b->ev_up) hipEventDestroy(b->ev_up);
  if (b->gexec) hipGraphExecDestroy(b->gexec);
  void *dev[] = {b->d_in, b->d_work, b->d_out, b->d_priors, b->d_pool};
  for (void *p : dev)
    if (p) hipFree(p);
  void *pin[] = {b->h_in, b->h_out, b->h_priors, b->hstate};
  for (void *p : pin)
    if (p) hipHostFree(p);
  delete b;
}
