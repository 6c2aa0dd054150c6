// grim_dev.h -- device-side primitives shared by the plan-A and plan-B/C kernels (gfx950, wave64).
//
// Everything here is integer indexing + fp64 scalar arithmetic: no MFMA.  fp64 products are
// written exactly as the reference writes them and the file is compiled with -ffp-contract=off,
// so every probability is bit-identical to CPython's IEEE double arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grim_hip.h"
#include "grim_layout.h"

#define GRIM_WG 256
#ifndef GRIM_WG_PER_CU
#define GRIM_WG_PER_CU 3  // resident workgroups per CU of the general kernel (LDS: 51 KB each)
#endif
#ifndef GRIM_PLANB_WG_PER_CU
#define GRIM_PLANB_WG_PER_CU 2  // the Plan-B kernel needs its 256 VGPRs (measured: 3 per CU spill and run 8 % slower)
#endif
#define GRIM_NWAVE 4
#define GRIM_NONE 0xFFFFFFFFu
#define GRIM_VALID 0x8000000000000000ull
#define GRIM_SIDES (2 * GRIM_MAXPH)
#define GRIM_COMP_CAP 8192  // >= 2 * GRIM_SIDES * GRIM_TOPCAP (64 KB: stays in the L2; a 512 KB table per workgroup did not)
#define GRIM_SAVE_KEEP 10   // entries save_space_mode keeps of either operand of a block join (impute.py:1041)
#define GRIM_SAVE_CAP (GRIM_SAVE_KEEP * GRIM_SAVE_KEEP)
#define GRIM_RTOK_CAP 12288 // u16 tokens per slot: 3 versions of up to 4096 alleles per subject

// ---- graph as the kernels see it --------------------------------------------------------------
// full-label nodes only, with the first population's frequency inline: the half-wave kernel's
// whole look-up is one 16-byte entry (grim_small.h)
struct FullEnt {
  uint64_t key;
  double f0;
};

struct HtEnt {
  uint64_t key;
  uint32_t val;
  uint32_t pad;
};

struct DevGraph {
  const FullEnt *fht;
  uint32_t fht_mask;
  const uint64_t *node_key;
  const uint8_t *node_mask;
  const double *freq;
  const uint32_t *a_start, *a_nbr;
  const uint32_t *b_conn, *b_start, *b_nbr;
  const uint32_t *lab_start, *lab_nodes;
  const uint64_t *lab_key;  // node_key[lab_nodes[i]]: the label scans stream this instead of gathering
  const HtEnt *ht;  // exact-name index: one 16-byte entry per slot, key == 0 marks an empty slot
  uint32_t n_nodes, P, full_mask, ht_mask, n_conn, n_loci;
  uint32_t scan_ok;  // names unique and every top-link row shorter than 2^22: the intersection opening may be used
  uint32_t order_bad;  // bit m: a name over label mask m built from a SUBJECT's alleles (sorted string order, impute.py:271)
                       // is never a graph name (loci_map index order) -- a loci_map that is not alphabetical; 0 as a rule
};

// ---- per-workgroup scratch slot in HBM (offsets in bytes, filled by the host) -------------------
struct SlotLayout {
  uint64_t Tp, Tm, Te;                 // top lists: prob, prefix-min prob, entity   [GRIM_SIDES][GRIM_TOPCAP]
  uint64_t k0, k1, tmin, tgid;         // hash table columns [tab_cap]
  uint64_t Useq, Uprob, Uslot;         // accepted unique pairs [pair_cap]
  uint64_t ska, skb, sva, svb;         // radix ping-pong [pair_cap]
  uint64_t gsum, ghead, gstart, gcnt;  // groups [pair_cap(+1)]
  uint64_t qsum, qfirst;               // population-pair cells [P*P]
  uint64_t bset;                       // plan-B block sets (node ids) [GRIM_NWAVE][GRIM_MAXL][bset_cap]
  uint64_t comp;                       // plan-B/C canonical haplotype table: open-addressing keys [GRIM_COMP_CAP]
  uint64_t proj_k, proj_p;             // plan-B label-scan projections: per-wave hash set [GRIM_NWAVE][proj_cap]
  uint64_t rtok;                       // the subject's allele lists: original + the two reduced versions [GRIM_RTOK_CAP]
  uint64_t save;                       // save_space_mode: per wave two buffers of GRIM_SAVE_CAP joined entries (key + P frequencies)
  uint64_t stride;                     // bytes per slot
};

struct DevArgs {
  DevGraph g;
  grim_params prm;
  const grim_subject *subj;
  const uint16_t *tok;
  const double *priors;
  uint32_t ones_prior;    // index of the all-ones prior matrix (impute.py:1696-1700)
  const uint32_t *order;  // subject indices this launch works on
  uint32_t n_work;
  uint32_t *queue;        // [0] general work counter [1] row head [2] plan-B list length [3] plan-B work counter
                          // [4] one-wave kernel work counter [5] its hand-over count [6] heavy plan-B subjects
                          // [7] its heavier hand-overs [8] pair pool head [9] / [10] work items of the one-wave /
                          // bigger-item table kernels  (GRIM_NQ words; the rest is listed at GRIM_NQ)
  grim_subject_result *res;
  grim_row *rows;
  uint32_t *row_head;
  uint32_t row_cap;
  uint8_t *scratch;
  uint32_t *wctr;               // work counters of the table kernels, GRIM_NSLICE per kernel, a 128-byte line each (slice_next)
  SlotLayout lay;
  uint32_t pair_cap, tab_cap, bset_cap, proj_cap;
  uint32_t *small_ctr;           // half-wave kernel: per wave {probes, frequency vectors}, plain stores; summed on request
  unsigned long long *counters;  // [0] probes [1] nbr ids [2] freq vectors [3] rows [4] overflow flag
  uint32_t *bail_list;           // subjects the one-wave kernel hands to the general kernel: light ones from the front
                                 // (count queue[5]), heavier ones from the back (count queue[7]); n_medium entries
  uint32_t n_medium;
  uint32_t *mid_list;            // subjects the mid-size kernel (grim_mid.h) hands to the general kernel (count queue[17]); null:
                                 // that kernel is switched off and the general kernel takes its own list and the one-wave
                                 // kernel's hand-overs itself
  uint32_t *next_list;           // subjects handed to the next kernel (plan B): light ones from the front
  uint32_t *next_count;          // (count queue[2]), heavy ones from the back (count queue[6]) -- heavy first
  uint32_t next_cap;             // entries in next_list
  uint32_t flags;                // diagnostic switches (environment, read when the batch is created): GRIM_F_*
  PairRec *ppool;                // accepted pairs of subjects whose tables the table kernels build (grim_tables.h)
  uint32_t ppool_cap;            // records
  TabWork *t1_list, *t2_list;    // their work items
  // the three-kernel path of the bigger items: per-item state, bucket / cell starts, work units, pairs dealt into
  // buckets (psort: [3][ppool_cap], genotype / haplotype pair / population-cell order), groups found ([2][ppool_cap]),
  // probabilities in cell order, cell sums
  TabAux *taux;
  uint32_t *tboff;
  CellRec *tcell;
  uint32_t tboff_cap;
  TabUnit *tunits;
  uint32_t tunits_cap;
  uint32_t tstride;              // records per plane of psort / pgrp (the pool's allocated size)
  uint32_t *psort;
  GrpRec *pgrp;
  double *pprob;
};
#define GRIM_F_NO_NODUP 2u       // DevArgs.flags (GRIM_NO_NODUP=1): the pair passes always run their dedup (test switch)
#define GRIM_F_NO_SIDEMASK 4u    // (GRIM_NO_SIDEMASK=1): tiled passes dedup through the slot's hash table as before round 4 (test switch)
#define GRIM_NQ 24               // u32 words of `queue` (the run state block is counters + queue):
                                 // [13] bucket-start slots used [14] work units [15] units done by earlier launches of the run
                                 // [16] work counter of the mid-size kernel [17] its hand-overs to the general kernel
                                 // [18] lines the device tokenizer handed back (GRIM_Q_IRREGULAR, grim_tokdev.h); [11] [19] [20] free
                                 // [12] items the workgroup split kernel took [21] the workgroup merge kernel [22] of those, with an
                                 // overflowed bucket [23] their largest pair count  (diagnostics, GRIM_DEBUG_CLASSES=1)
                                 // (the table kernels' own work counters are DevArgs.wctr)

#ifndef GRIM_HEAVY_LOCI
#define GRIM_HEAVY_LOCI 3
#endif
// hand a subject to the Plan-B kernel.  Subjects with few typed loci (their sides saturate the top lists: tens
// of thousands of pairs) go to the head of its queue so that they do not start last and become the tail.
__device__ __forceinline__ void push_next(const DevArgs &A, uint32_t si, bool heavy) {
  if (heavy)
    A.next_list[A.next_cap - 1u - atomicAdd(A.queue + 6, 1u)] = si;
  else
    A.next_list[atomicAdd(A.next_count, 1u)] = si;
}

// ---- optional stage timers (diagnostic build only: hipcc -DGRIM_STAMPS; never in the shipped .so) ----
#define GRIM_STAMP_BASE (8 + 4 * 64)
#ifdef GRIM_STAMPS
#define GRIM_HIST_BASE (GRIM_STAMP_BASE + 16)  // diagnostic build: log2 histograms, 8 x 24 buckets
#define GRIM_MID_BASE (GRIM_HIST_BASE + 192)   // ... and the mid-size kernel's stage timers [0..15], hand-over reasons [16..31]
#define GRIM_NCTR (8 + 4 * 64 + 16 + 192 + 32)
#else
#define GRIM_NCTR (8 + 4 * 64 + 16)
#endif
#ifdef GRIM_STAMPS
#define STAMP_BEGIN() unsigned long long _t0 = wall_clock64()
#define STAMP(k)                                                           \
  do {                                                                     \
    __syncthreads();                                                       \
    if (threadIdx.x == 0) {                                                \
      unsigned long long _t1 = wall_clock64();                             \
      atomicAdd(&A.counters[GRIM_STAMP_BASE + (k)], _t1 - _t0);            \
      _t0 = _t1;                                                           \
    }                                                                      \
  } while (0)
// HIST(h, v, w): bucket floor(log2(v+1)) of histogram h (0..7) += w
#define HIST(h, v, w)                                                                                  \
  do {                                                                                                 \
    if (threadIdx.x == 0) {                                                                            \
      unsigned int _b = 31u - __clz((unsigned int)(v) + 1u);                                           \
      atomicAdd(&A.counters[GRIM_HIST_BASE + 24 * (h) + (_b < 23u ? _b : 23u)], (unsigned long long)(w)); \
    }                                                                                                  \
  } while (0)
#define STAMP_NOW() wall_clock64()
#else
#define STAMP_BEGIN()
#define STAMP(k)
#define HIST(h, v, w)
#define STAMP_NOW() 0ull
#endif

// ---- small helpers ------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

// hash of a full-haplotype key for the half-wave kernel's table: three full-rate 24-bit multiplies over the key's 24-bit
// pieces (two allele fields each) instead of mix64's two 64-bit ones -- the kernel is instruction-issue bound at scale
__host__ __device__ __forceinline__ uint32_t fht_hash(uint64_t k) {
  const uint32_t a = (uint32_t)k & 0xFFFFFFu, b = (uint32_t)(k >> 24) & 0xFFFFFFu, c = (uint32_t)(k >> 48);
  uint32_t h = (a * 0x9E3779u) ^ (b * 0x85EBCBu) ^ (c * 0xC2B2AFu);
  h ^= h >> 16;
  h *= 0x2C1B3Du;
  h ^= h >> 13;
  return h;
}

// order-preserving map double -> uint64 (bigger double <=> bigger integer)
__device__ __forceinline__ uint64_t f64_ord(double x) {
  uint64_t b = (uint64_t)__double_as_longlong(x);
  return (b & GRIM_VALID) ? ~b : (b | GRIM_VALID);
}

#define ALOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ASTORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// LDS traffic between lanes of ONE wave: DS ops of a wave are processed in order; this only has to
// stop the compiler from moving them.
#define WAVE_SYNC()                                          \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- work lists shared out through SLICED counters ----------------------------------------------------------------------
// An atomic on ONE address costs ~12 ns whoever issues it, so a list of 13 000 items handed out one at a time through a
// single counter to 4 000 waves is 0.2 ms of queueing -- each wave waits for all the others' turns.  The list is cut into
// GRIM_NSLICE stretches with a counter each (own cache line); a wave starts at the stretch blockIdx.x picks and moves on
// to the next one when a stretch is used up (so the tail is still shared by everybody).
#define GRIM_NSLICE 16
#define GRIM_SLICE_STRIDE 32  // u32 words between counters
enum { GRIM_WL_T1 = 0, GRIM_WL_SPLIT_WAVE, GRIM_WL_MERGE_WAVE, GRIM_WL_SPLIT, GRIM_WL_MERGE, GRIM_WL_N };
#define GRIM_WCTR_WORDS (GRIM_WL_N * GRIM_NSLICE * GRIM_SLICE_STRIDE)
struct SliceWalk {
  uint32_t s, tried;
};
// next item of a list of n for this WAVE (all lanes get the same answer), GRIM_NONE: the list is done
__device__ __forceinline__ uint32_t slice_next(uint32_t *wctr, int list, uint32_t n, SliceWalk &sw) {
  const uint32_t per = (n + GRIM_NSLICE - 1) / GRIM_NSLICE;
  uint32_t *ctr = wctr + (uint32_t)list * GRIM_NSLICE * GRIM_SLICE_STRIDE;
  while (sw.tried < GRIM_NSLICE) {
    const uint32_t lo = sw.s * per, hi = lo + per < n ? lo + per : n;
    if (lo < hi) {
      uint32_t i = 0;
      if (lane_id() == 0) i = atomicAdd(&ctr[sw.s * GRIM_SLICE_STRIDE], 1u);
      i = __shfl(i, 0);
      if (i < hi - lo) return lo + i;
    }
    sw.s = (sw.s + 1) % GRIM_NSLICE;
    sw.tried++;
  }
  return GRIM_NONE;
}

// value of lane j (j wave-uniform): v_readlane, not the LDS-crossbar ds_bpermute that __shfl emits
// for a run-time index
__device__ __forceinline__ uint32_t lane_get(uint32_t v, int j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); }
__device__ __forceinline__ uint64_t lane_get(uint64_t v, int j) {
  return ((uint64_t)lane_get((uint32_t)(v >> 32), j) << 32) | lane_get((uint32_t)v, j);
}
__device__ __forceinline__ double lane_get(double v, int j) {
  return __longlong_as_double((long long)lane_get((uint64_t)__double_as_longlong(v), j));
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan over the workgroup; `tmp` is LDS [GRIM_NWAVE]; two barriers
__device__ __forceinline__ uint32_t wg_excl_scan(uint32_t v, uint32_t *tmp, uint32_t &total) {
  uint32_t inc = wave_incl_scan(v);
  if (lane_id() == 63) tmp[wave_id()] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < GRIM_NWAVE; ++w) {
    uint32_t t = tmp[w];
    if (w < wave_id()) base += t;
    tot += t;
  }
  __syncthreads();
  total = tot;
  return base + inc - v;
}

// ---- exact-name lookup: 64-bit key -> node id (networkx_graph.py:260,290,315) ----------------------
__device__ __forceinline__ uint32_t graph_lookup_from(const DevGraph &g, uint64_t key, uint32_t h) {
  for (;;) {
    const HtEnt e = g.ht[h];  // one 16-byte load gives key and node id
    if (e.key == key) return e.val;
    if (e.key == 0) return GRIM_NONE;
    h = (h + 1) & g.ht_mask;
  }
}

__device__ __forceinline__ uint32_t graph_lookup(const DevGraph &g, uint64_t key) {
  return graph_lookup_from(g, key, (uint32_t)mix64(key) & g.ht_mask);
}

// look-up of a name made of a subject's alleles in the subject's (sorted) order over the loci of `mask`: a miss when that
// order is not the graph's for this set of loci (DevGraph::order_bad)
__device__ __forceinline__ bool subject_order_ok(const DevGraph &g, uint32_t mask) { return !((g.order_bad >> mask) & 1u); }
__device__ __forceinline__ uint32_t graph_lookup_subject(const DevGraph &g, uint64_t key, uint32_t mask) {
  return subject_order_ok(g, mask) ? graph_lookup(g, key) : GRIM_NONE;
}

// plan-A neighbour range of a partial node, with the reference's sentinel quirk: the range is
// range(start[i], start[i+1]) and is EMPTY when start[i+1] <= start[i] (networkx_graph.py:195,267-272)
__device__ __forceinline__ uint32_t nbr_count(const uint32_t *start, uint32_t i) {
  uint32_t a = start[i], b = start[i + 1];
  return b > a ? b - a : 0;
}

// ---- per-slot hash table over 64- or 128-bit keys (open addressing, insert-only) -----------------
// k0 == 0 marks an empty slot, so callers OR GRIM_VALID into k0 (and into k1 when the key is 128 bits wide).
// Single 64-bit insert: one CAS claims a slot; there is no second word to publish, hence nothing to wait for.  (128-bit
// keys go through tab_insert_n only: its claim, publish and re-read steps sit in one loop body with one exit, so a lane
// that waits for a sibling lane's second word can never be parked behind it.)
__device__ __forceinline__ uint32_t tab_insert(uint64_t *k0, uint32_t mask, uint64_t a) {
  uint32_t s = (uint32_t)mix64(a) & mask;
  for (;;) {
    uint64_t c0 = ALOAD(&k0[s]);
    if (c0 == 0) {
      const uint64_t old = atomicCAS((unsigned long long *)&k0[s], 0ull, (unsigned long long)a);
      c0 = old == 0 ? a : old;
    }
    if (c0 == a) return s;
    s = (s + 1) & mask;
  }
}

// N independent inserts per thread with their table accesses in flight together: a thread that walks
// thousands of pairs is bound by the ~1-2 us of each device-scope atomic, not by their number.  Same protocol
// and same result as N tab_insert calls (two equal keys of one thread end in the same slot).
template <bool WIDE, int N>
__device__ __forceinline__ void tab_insert_n(uint64_t *k0, uint64_t *k1, uint32_t mask, const uint64_t (&a)[N], const uint64_t (&b)[N],
                                             const bool (&on)[N], uint32_t (&out)[N]) {
  uint32_t s[N];
  bool done[N];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    s[q] = (uint32_t)mix64(a[q] ^ (b[q] * 0x9E3779B97F4A7C15ull)) & mask;
    done[q] = !on[q];
    out[q] = GRIM_NONE;
  }
  for (;;) {
    bool all = true;
#pragma unroll
    for (int q = 0; q < N; ++q) all = all && done[q];
    if (all) return;
    uint64_t c0[N], old[N], c1[N];
    bool claim[N], same[N];
#pragma unroll
    for (int q = 0; q < N; ++q) c0[q] = done[q] ? 1ull : ALOAD(&k0[s[q]]);
#pragma unroll
    for (int q = 0; q < N; ++q) {
      claim[q] = !done[q] && c0[q] == 0;
      old[q] = 0;
      if (claim[q]) old[q] = atomicCAS((unsigned long long *)&k0[s[q]], 0ull, (unsigned long long)a[q]);
    }
#pragma unroll
    for (int q = 0; q < N; ++q)
      if (claim[q]) {
        if (old[q] == 0) {
          if (WIDE) ASTORE(&k1[s[q]], b[q]);
          out[q] = s[q];
          done[q] = true;
        } else {
          c0[q] = old[q];
        }
      }
#pragma unroll
    for (int q = 0; q < N; ++q) {
      same[q] = !done[q] && c0[q] == a[q];
      c1[q] = 0;
      if (WIDE && same[q]) c1[q] = ALOAD(&k1[s[q]]);
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
      if (done[q]) continue;
      if (same[q]) {
        if (!WIDE || c1[q] == b[q]) {
          out[q] = s[q];
          done[q] = true;
          continue;
        }
        if (c1[q] == 0) continue;  // claimed a moment ago, second word not published yet: look again
      }
      s[q] = (s[q] + 1) & mask;
    }
  }
}

// ---- wave-level running top-K with stable ties (impute.py:424-442) ---------------------------------
// Records live in LDS, one set per wave: [0,nrun) is the sorted running list, [nrun,nrun+nbuf) the
// not yet merged newcomers.  Order: bigger sort key first, then smaller `tie` (= stream position).
struct WaveTop {
  uint64_t sk[256];
  uint64_t tie[256];
  double p[256];
  uint64_t aux[256];  // 60-bit haplotype key of the entry (plan B/C: composite haplotypes)
  uint32_t hap[256];
  uint32_t cstart[66];
  uint32_t cnode[64];
  uint64_t caux[64];
  uint16_t toks[GRIM_MAXL * 64];  // this side's alternatives per position (when each list has <= 64)
};

struct TopState {
  int nrun, nbuf, K;
  bool full;
  bool ge;  // entries arrive out of stream order: one that ties the K-th key may still beat it on `tie`
  uint64_t thr;
};

__device__ __forceinline__ void wave_sort(WaveTop &L, int N) {
  int lane = lane_id();
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int q = lane; q < (N >> 1); q += 64) {
        int lo = ((q & ~(j - 1)) << 1) | (q & (j - 1));
        int hi = lo + j;
        bool up = ((lo & k) == 0);
        uint64_t a = L.sk[lo], b = L.sk[hi], ta = L.tie[lo], tb = L.tie[hi];
        bool b_first = (b > a) || (b == a && tb < ta);
        bool a_first = (a > b) || (a == b && ta < tb);
        if (up ? b_first : a_first) {
          L.sk[lo] = b; L.sk[hi] = a;
          L.tie[lo] = tb; L.tie[hi] = ta;
          double pa = L.p[lo]; L.p[lo] = L.p[hi]; L.p[hi] = pa;
          uint32_t ha = L.hap[lo]; L.hap[lo] = L.hap[hi]; L.hap[hi] = ha;
          uint64_t xa = L.aux[lo]; L.aux[lo] = L.aux[hi]; L.aux[hi] = xa;
        }
      }
      WAVE_SYNC();
    }
  }
}

__device__ __forceinline__ void top_flush(WaveTop &L, TopState &st) {
  int lane = lane_id();
  int total = st.nrun + st.nbuf;
  if (st.nbuf > 0 && total > 1) {
    int N = 2;
    while (N < total) N <<= 1;
    for (int r = total + lane; r < N; r += 64) {
      L.sk[r] = 0;
      L.tie[r] = ~0ull;
    }
    WAVE_SYNC();
    wave_sort(L, N);
  }
  st.nrun = total < st.K ? total : st.K;
  st.nbuf = 0;
  WAVE_SYNC();
  if (st.nrun == st.K) {
    st.full = true;
    st.thr = L.sk[st.K - 1];
  }
}

// push one candidate entry per lane (active = this lane has one); wave-uniform control flow
__device__ __forceinline__ void top_push(WaveTop &L, TopState &st, bool active, double p, double key, uint64_t tie,
                                         uint32_t hap, uint64_t aux = 0) {
  uint64_t ord = f64_ord(key);
  bool adm = active && (!st.full || ord > st.thr || (st.ge && ord == st.thr));
  uint64_t m = __ballot(adm);
  if (m == 0) return;
  if (adm) {
    int pos = st.nrun + st.nbuf + __popcll(m & ((1ull << lane_id()) - 1ull));
    L.sk[pos] = ord;
    L.tie[pos] = tie;
    L.p[pos] = p;
    L.hap[pos] = hap;
    L.aux[pos] = aux;
  }
  st.nbuf += __popcll(m);
  WAVE_SYNC();
  if (st.nrun + st.nbuf > 192) top_flush(L, st);
}

// ---- stable LSD radix sort (4 bits/pass) of n (key,value) records by one workgroup ------------------
// Records sit in HBM scratch (L2 resident); each thread owns a contiguous block so that equal
// digits keep their order.  hist: LDS [16*256]; tmp: LDS [GRIM_NWAVE+20].
// Returns 0 if the result is in (ka,va), 1 if in (kb,vb).
__device__ inline int wg_radix_sort(uint64_t *ka, uint32_t *va, uint64_t *kb, uint32_t *vb, uint32_t n, int nbits,
                                    uint32_t *hist, uint32_t *tmp) {
  const int tid = threadIdx.x;
  int cur = 0;
  if (n <= 1) return 0;
  uint32_t per = (n + GRIM_WG - 1) / GRIM_WG;
  uint32_t b0 = tid * per, b1 = b0 + per;
  if (b0 > n) b0 = n;
  if (b1 > n) b1 = n;
  for (int sh = 0; sh < nbits; sh += 4) {
    uint64_t *sk = cur ? kb : ka, *dk = cur ? ka : kb;
    uint32_t *sv = cur ? vb : va, *dv = cur ? va : vb;
    for (int d = 0; d < 16; ++d) hist[d * GRIM_WG + tid] = 0;
    {  // eight keys per step: the loads are in flight together (a thread's block is contiguous)
      uint32_t i = b0;
      for (; i + 8 <= b1; i += 8) {
        uint64_t k[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) k[q] = sk[i + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) hist[((k[q] >> sh) & 15) * GRIM_WG + tid]++;
      }
      for (; i < b1; ++i) hist[((sk[i] >> sh) & 15) * GRIM_WG + tid]++;
    }
    __syncthreads();
    // thread t owns flattened entries [16t,16t+16): digit t/16, threads (t%16)*16..
    uint32_t part = 0;
    for (int e = 0; e < 16; ++e) part += hist[tid * 16 + e];
    // a digit that holds everything makes the pass a no-op
    uint32_t *dtot = tmp + GRIM_NWAVE;  // [16] + flag at [16]
    if (tid < 17) dtot[tid] = 0;
    __syncthreads();
    atomicAdd(&dtot[tid >> 4], part);
    __syncthreads();
    if (tid < 16 && dtot[tid] == n) dtot[16] = 1;
    __syncthreads();
    bool skip = dtot[16] != 0;
    __syncthreads();
    if (skip) continue;
    uint32_t total;
    uint32_t base = wg_excl_scan(part, tmp, total);
    for (int e = 0; e < 16; ++e) {
      uint32_t c = hist[tid * 16 + e];
      hist[tid * 16 + e] = base;
      base += c;
    }
    __syncthreads();
    {
      uint32_t i = b0;
      for (; i + 8 <= b1; i += 8) {
        uint64_t k[8];
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          k[q] = sk[i + q];
          v[q] = sv[i + q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {  // in order: equal digits keep their order
          const uint32_t pos = hist[((k[q] >> sh) & 15) * GRIM_WG + tid]++;
          dk[pos] = k[q];
          dv[pos] = v[q];
        }
      }
      for (; i < b1; ++i) {
        const uint64_t k = sk[i];
        const uint32_t pos = hist[((k >> sh) & 15) * GRIM_WG + tid]++;
        dk[pos] = k;
        dv[pos] = sv[i];
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  return cur;
}

// exclusive scan in place over a u32 array in HBM scratch, result has n+1 entries (arr[n] = total)
__device__ inline void wg_scan_array(uint32_t *arr, uint32_t n, uint32_t *tmp) {
  const int tid = threadIdx.x;
  uint32_t per = (n + GRIM_WG - 1) / GRIM_WG;
  uint32_t b0 = tid * per, b1 = b0 + per;
  if (b0 > n) b0 = n;
  if (b1 > n) b1 = n;
  uint32_t s = 0;
  for (uint32_t i = b0; i < b1; ++i) s += arr[i];
  uint32_t total;
  uint32_t base = wg_excl_scan(s, tmp, total);
  for (uint32_t i = b0; i < b1; ++i) {
    uint32_t c = arr[i];
    arr[i] = base;
    base += c;
  }
  if (tid == 0) arr[n] = total;
  __syncthreads();
}

__device__ __forceinline__ int bits_for(uint32_t n) {  // bits needed to hold values < n
  int b = 1;
  while (b < 32 && (1u << b) < n) ++b;
  return b;
}
