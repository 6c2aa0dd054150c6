// grim_tables.h -- the four output tables of a subject with more than 64 accepted haplotype pairs, as kernels of their own.
//
// The kernels that score pairs (one-wave, general, Plan B/C) leave the accepted pairs of such a subject as 32-byte
// records {haplotype keys, probability, entities} in a pool in HBM, in the reference's pair order, plus a work item;
// the tables -- population pairs, genotypes (.umug), haplotype pairs (.pmug) with their left-to-right sums and stable
// rankings (impute.py:24-99, 497-543) -- are built afterwards by
//   grim_tables_wave_kernel : ONE WAVE per subject with <= 256 pairs, everything in 11 KB of LDS, no workgroup barrier,
//                             14 waves per CU (the mass of the mixed workloads: hundreds of pairs per subject);
//   grim_tables_wg_kernel   : one workgroup per subject with more pairs: the pairs are dealt into buckets of <= 256 by
//                             the hash of their group key (a group never spans buckets), every wave runs the same
//                             LDS-resident grouping on one bucket after the other, the groups are renumbered in
//                             first-seen order and ranked; population pairs (few, huge groups) are partitioned by cell
//                             and summed by register-speed chains, one wave per cell.
// Grouping of <= 256 pairs by a wave: find-or-insert of every pair's key into an LDS hash table whose slot keeps the
// group's first pair (atomicMin = first seen), a bitonic sort of (slot, pair number) so that a group is a run in pair
// order, one lane per run adds the probabilities left to right.  Same groups, same summation order, same tie rule as
// the hash-table + radix-sort path in HBM scratch this replaces for all but pathological inputs (it stays as the
// fallback when a bucket overflows, and GRIM_TABLES_HBM=1 forces it: the tests hold the two together).
#pragma once
#include "grim_pair.h"

#define TAB_N 256        // pairs one wave groups at a time
#define TAB_SH 18        // bits of a pair number inside a sort key (pair_cap = 16 * 128 * 128 < 2^18)
#define TAB_MAXB 4096    // buckets of one table of one work item (pair_cap / 48 fits)
#define GRIM_F_TABLES_HBM 1u  // DevArgs.flags: the workgroup kernel groups through HBM scratch (diagnostic / test switch)

template <int N>
struct WaveTab {
  double prob[N];          // sum of the group local pair i represents
  uint64_t klo[N], khi[N]; // group keys
  uint32_t tab[2 * N];     // hash slots: the local pair that represents the group; afterwards the group sums (double[N])
  uint32_t skey[N];        // first pair (record index) of the group local pair i represents
  double cp[64 + 1];       // the probabilities of the chunk in hand, by lane; [64] = 0.0
  unsigned long long gm[N];  // [rep]: the lanes of the chunk in hand whose pair belongs to the group rep stands for (0 between chunks)
  uint16_t rs[N + 2];      // run j -> its representative
};
struct WaveTabT1 : WaveTab<TAB_N> {
  uint16_t hd[TAB_N];      // (one-wave kernel) head pair of run j
};
#ifndef TAB_NB
#define TAB_NB 256         // pairs per bucket the bucket kernel holds
#endif
#ifndef TAB_DIV
#define TAB_DIV 192u        // a table of n pairs is dealt into the power of two >= n / TAB_DIV buckets (96..192 pairs on average; a
                           // bucket beyond TAB_NB pairs has its table re-dealt into twice as many).  Table kernels per 100 k
                           // config-4 subjects / 2 048 config-5 subjects with round 3's kernels: 5.53 / -- ms at 96, 5.37-5.56 /
                           // 15.6 at 128, 5.23-5.49 / 15.0 at 192, 5.33 / 15.5 at 256 (a unit's fixed cost -- header, pair numbers,
                           // group counter -- against the re-deals of the tables whose buckets overflow)
#endif

struct TabShared {
  uint32_t tmp[GRIM_NWAVE + 24];
  double dtmp[GRIM_NWAVE];
  uint32_t bc[8];
  uint32_t *hist;   // [16 * GRIM_WG]: radix histograms, ranking scratch (a WgArena of the kernel)
  double *qprob;    // [1024]
  uint16_t *qcell;  // [1024]
  uint32_t ng, overflow;
  TabWork work;
  uint32_t bcnt[TAB_MAXB + 1];
  uint16_t *wcnt;   // [GRIM_NWAVE][nb]: pairs of bucket b in wave w's stretch of the item, then its cursor (split kernel)
};

// group key of a pair under `kind`: 0 genotype (impute.py:497-504), 1 unordered haplotype pair (impute.py:24-39),
// 3 unordered population pair (impute.py:535-543).  (kind 2, every pair its own group -- write_best_hap_race_pairs,
// impute.py:79-85 -- needs no grouping.)
__device__ __forceinline__ void tab_key(int kind, int P, const PairRec &r, uint64_t &lo, uint64_t &hi) {
  if (kind == 0) {
    lo = hi = 0;
#pragma unroll
    for (int l = 0; l < GRIM_MAXL; ++l) {
      const uint64_t x = (r.k1 >> (GRIM_ABITS * l)) & 0xFFF, y = (r.k2 >> (GRIM_ABITS * l)) & 0xFFF;
      lo |= (x < y ? x : y) << (GRIM_ABITS * l);
      hi |= (x < y ? y : x) << (GRIM_ABITS * l);
    }
    lo |= GRIM_VALID;
  } else if (kind == 1) {
    const uint32_t h1 = ENT_HAP(r.e1), h2 = ENT_HAP(r.e2);
    lo = (((uint64_t)(h1 < h2 ? h1 : h2)) << 32 | (h1 < h2 ? h2 : h1)) | GRIM_VALID;
    hi = 0;
  } else {
    const uint32_t a = ENT_POP(r.e1), b = ENT_POP(r.e2);
    lo = (uint64_t)((a < b ? a : b) * (uint32_t)P + (a < b ? b : a)) | GRIM_VALID;
    hi = 0;
  }
}
// 64-bit hash of a group key from 32-bit operations (a wave's grouping is instruction bound; mix64's three 64-bit multiplies
// were a tenth of it): bits 0..31 pick the slot inside a bucket, bits 40..51 the bucket
__device__ __forceinline__ uint64_t tab_hash(uint64_t lo, uint64_t hi) {
  uint32_t a = (uint32_t)lo ^ (uint32_t)(hi >> 32) * 0x85EBCA6Bu, b = (uint32_t)(lo >> 32) ^ (uint32_t)hi * 0xC2B2AE35u;
  a ^= b >> 15;
  a *= 0x9E3779B1u;
  b ^= a >> 13;
  b *= 0x85EBCA6Bu;
  a ^= b >> 16;
  return ((uint64_t)b << 32) | a;
}

// ---- one wave groups n <= N pairs --------------------------------------------------------------------------------------
// In: the n <= N <= 256 pairs IN INCREASING ORDER of their numbers (registers, see below).  Out: the number of groups ("runs",
// in no particular order); run j: first pair W.skey[W.rs[j]] & UM (the smallest number in the group = first seen), sum
// ((double *)W.tab)[j] = its probabilities added in pair order (impute.py:497-543: the reference's dict updates).
// 64 pairs at a time, in order: find-or-insert of the pair's key into an LDS hash table whose slot names the group's
// representative (the pair that claimed it); the lanes of a chunk that share a representative are found with ballots over
// the representative's bits, and the first of them adds the chunk's members to the group's sum one after the other -- so a
// sum receives its terms in lane = pair order, one chunk after the other, without sorting anything.  (Until round 2's last
// day this was a bitonic sort of (slot, pair) per bucket followed by run sums: 36 LDS compare-exchange stages for 256 keys.)
// The records come in REGISTERS (R[c], U[c]: record and pair number of local pair 64 c + lane), loaded by the caller in one
// go before anything else happens: a chunk-by-chunk gather inside the loop put a memory round trip in front of every chunk.
template <int N>
__device__ inline uint32_t wave_group_pairs(WaveTab<N> &W, int kind, int P, const PairRec (&R)[N / 64], const uint32_t (&U)[N / 64],
                                            uint32_t n) {
  const int lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  for (int i = lane; i < 2 * N; i += 64) W.tab[i] = GRIM_NONE;
  for (uint32_t i = lane; i < n; i += 64) {
    W.prob[i] = 0.0;          // sum of the group pair i represents
    W.skey[i] = GRIM_NONE;    // ... and its first pair
    W.gm[i] = 0ull;
  }
  if (lane == 0) W.cp[64] = 0.0;
  WAVE_SYNC();
  volatile uint32_t *tab = W.tab;
  uint32_t nruns = 0;
#pragma unroll
  for (int c = 0; c < N / 64; ++c) {
    const uint32_t c0 = 64u * (uint32_t)c;
    if (c0 >= n) break;
    const uint32_t i = c0 + lane;
    const bool act = i < n;
    uint32_t rep = 0, u = 0;
    double p = 0.0;
    uint64_t lo = 0, hi = 0;
    if (act) {
      u = U[c];
      const PairRec r = R[c];
      p = r.prob;
      tab_key(kind, P, r, lo, hi);
      W.klo[i] = lo;
      W.khi[i] = hi;
    }
    WAVE_SYNC();  // a probe compares against the keys of this chunk's representatives too
    bool claimed = false;
    if (act) {
      uint32_t h = (uint32_t)tab_hash(lo, hi) & (2 * N - 1);
      for (;;) {
        uint32_t cur = tab[h];
        if (cur == GRIM_NONE) {
          cur = atomicCAS(&W.tab[h], GRIM_NONE, i);
          if (cur == GRIM_NONE) {  // claimed: this pair stands for the group
            claimed = true;
            rep = i;
            break;
          }
        }
        if (W.klo[cur] == lo && W.khi[cur] == hi) {  // joins the group that pair stands for
          rep = cur;
          break;
        }
        h = (h + 1) & (2 * N - 1);
      }
    }
    // the lanes of this chunk with my representative: every lane sets its bit in the representative's mask and reads the
    // mask back (eight ballots over the representative's bits -- ~80 instructions of a chunk's ~450 -- until round 3's end)
    if (act) atomicOr(&W.gm[rep], 1ull << lane);
    WAVE_SYNC();
    const uint64_t same = act ? W.gm[rep] : 0ull;
    WAVE_SYNC();
    const uint32_t pos = (uint32_t)__popcll(same & lt);
    // The group's members in this chunk join its sum in lane = pair order.  Its FIRST lane does that for all of them, from
    // the chunk's probabilities in LDS, four loads in flight per step: a bucket holds a handful of big groups as a rule (the
    // 16 population pairs x 2 phases of one genotype), and a round per member -- an LDS read-add-write each, as this was
    // until round 3 -- made the chunk's time the size of its biggest group times the LDS latency.
    W.cp[lane] = p;
    WAVE_SYNC();
    if (act && pos == 0) {
      if (W.skey[rep] == GRIM_NONE) W.skey[rep] = u;  // chunks and lanes come in pair order
      double acc = W.prob[rep];
      uint64_t m = same;
      while (m) {
        int b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          b4[q] = m ? __builtin_ctzll(m) : 64;  // [64] holds 0.0: x + 0.0 == x for the positive sums here
          m &= m - 1;                           // (0 stays 0)
        }
        const double p0 = W.cp[b4[0]], p1 = W.cp[b4[1]], p2 = W.cp[b4[2]], p3 = W.cp[b4[3]];
        acc = acc + p0;
        acc = acc + p1;
        acc = acc + p2;
        acc = acc + p3;
      }
      W.prob[rep] = acc;
      W.gm[rep] = 0ull;  // (every lane of the group has read it)
    }
    WAVE_SYNC();
    const uint64_t cm = __ballot(claimed);
    if (claimed) W.rs[nruns + (uint32_t)__popcll(cm & lt)] = (uint16_t)i;
    nruns += (uint32_t)__popcll(cm);
  }
  WAVE_SYNC();
  double *gsum = (double *)W.tab;  // the hash table is spent
  for (uint32_t j = lane; j < nruns; j += 64) gsum[j] = W.prob[W.rs[j]];
  WAVE_SYNC();
  return nruns;
}

// population pairs of n <= N pairs (few groups, many members each: rounds would serialise): one LANE per cell walks the
// pairs in order.  Same outputs as wave_group_pairs.
template <int N>
__device__ inline uint32_t wave_group_cells(WaveTab<N> &W, int P, const PairRec (&R)[N / 64], uint32_t n) {
  const int lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  uint16_t *cellid = (uint16_t *)W.klo;
#pragma unroll
  for (int c = 0; c < N / 64; ++c) {
    const uint32_t i = 64u * (uint32_t)c + lane;
    if (i < n) {
      const uint32_t a = ENT_POP(R[c].e1), b = ENT_POP(R[c].e2);
      cellid[i] = (uint16_t)((a < b ? a : b) * (uint32_t)P + (a < b ? b : a));
      W.prob[i] = R[c].prob;
    }
  }
  WAVE_SYNC();
  double *gsum = (double *)W.tab;
  const uint32_t ncell = (uint32_t)(P * P);
  uint32_t nruns = 0;
  for (uint32_t c0 = 0; c0 < ncell; c0 += 64) {
    const uint32_t cell = c0 + lane;
    double s = 0.0;
    uint32_t first = GRIM_NONE;
    for (uint32_t i = 0; i < n; ++i)
      if (cellid[i] == cell) {
        if (first == GRIM_NONE) first = i;  // (the one-wave kernel's pairs are numbered 0..n-1)
        s = s + W.prob[i];
      }
    const uint64_t m = __ballot(first != GRIM_NONE);
    if (first != GRIM_NONE) {
      const uint32_t j = nruns + (uint32_t)__popcll(m & lt);
      W.rs[j] = (uint16_t)j;
      W.skey[j] = first;
      gsum[j] = s;
    }
    nruns += (uint32_t)__popcll(m);
  }
  WAVE_SYNC();
  return nruns;
}

// ---- the one-wave kernel ----------------------------------------------------------------------------------------------
// rank of run j among nruns: bigger sum first, the group seen first on ties (stable sort by probability)
__device__ __forceinline__ uint32_t wave_run_rank(const WaveTabT1 &W, uint32_t nruns, uint32_t j) {
  const double *gsum = (const double *)W.tab;
  const double s = gsum[j];
  const uint32_t h = W.hd[j];
  uint32_t rank = 0;
  for (uint32_t j2 = 0; j2 < nruns; ++j2) {
    const double s2 = gsum[j2];
    const uint32_t h2 = W.hd[j2];
    rank += (s2 > s || (s2 == s && h2 < h)) ? 1u : 0u;
  }
  return rank;
}

__device__ inline void tables_wave(const DevArgs &A, WaveTabT1 &W, const TabWork &w, RowBlock &rb) {
  const int lane = lane_id();
  const int P = A.g.P;
  const PairRec *rec = A.ppool + w.off;
  const uint32_t n = w.n, mask = w.mask;
  constexpr uint32_t UM = (1u << TAB_SH) - 1u;
  grim_subject_result *out = A.res + w.si;
  const double *gsum = (const double *)W.tab;
  PairRec R[TAB_N / 64];  // the item's records, once for the three groupings
  uint32_t U[TAB_N / 64];
#pragma unroll
  for (int c = 0; c < TAB_N / 64; ++c) {
    U[c] = 64u * (uint32_t)c + lane;
    if (U[c] < n) R[c] = rec[U[c]];
  }
  // ---- population pairs (both pops files share the sums) -------------------------------------------------------------
  {
    const uint32_t nq = wave_group_cells(W, P, R, n);
    for (uint32_t j = lane; j < nq; j += 64) W.hd[j] = (uint16_t)(W.skey[W.rs[j]] & UM);
    WAVE_SYNC();
    for (int t = 0; t < 2; ++t) {
      if (!((mask >> t) & 1u)) continue;
      const int table = t == 0 ? GRIM_T_UMUG_POPS : GRIM_T_PMUG_POPS;
      uint32_t want = nq < A.prm.n_pop_results ? nq : A.prm.n_pop_results;
      if (t == 1 && A.prm.em_mr) want = nq < 1 ? nq : 1;
      if (!(t == 0 ? A.prm.out_muug : A.prm.out_haps)) want = 0;
      const uint32_t off = wave_alloc_rows(A, rb, want);
      if (lane == 0) {
        out->row_off[table] = off == GRIM_NONE ? 0 : off;
        out->n_rows[table] = off == GRIM_NONE ? 0 : want;
      }
      if (off == GRIM_NONE || want == 0) continue;
      for (uint32_t j = lane; j < nq; j += 64) {
        const uint32_t rank = wave_run_rank(W, nq, j);
        if (rank >= want) continue;
        const PairRec r0 = rec[W.hd[j]];
        uint32_t a = ENT_POP(r0.e1), b = ENT_POP(r0.e2);
        if (t == 0 && A.prm.pop_rank[a] > A.prm.pop_rank[b]) {
          const uint32_t x = a;
          a = b;
          b = x;
        }
        grim_row r;
        r.a = a; r.b = b; r.prob = gsum[j]; r.popa = a; r.popb = b;
        A.rows[off + rank] = r;
      }
    }
    WAVE_SYNC();
  }
  // ---- genotypes (.umug) and haplotype pairs (.pmug) -------------------------------------------------------------------
  for (int t = 0; t < 2; ++t) {
    if (!((mask >> t) & 1u)) continue;
    const int table = t == 0 ? GRIM_T_UMUG : GRIM_T_PMUG;
    const bool on = t == 0 ? A.prm.out_muug : A.prm.out_haps;
    uint32_t ng = 0, want = 0;
    const bool own = t == 1 && (A.prm.em_mr || P == 1);  // every pair its own group (one population: the pair pass
                                                         // already made the pairs unique per unordered haplotype pair)
    if (t == 0 || on) {  // the genotype count is reported even when the MUUG file is off
      if (own) {
        ng = n;
      } else {
        ng = wave_group_pairs(W, t == 0 ? 0 : 1, P, R, U, n);
        for (uint32_t j = lane; j < ng; j += 64) W.hd[j] = (uint16_t)(W.skey[W.rs[j]] & UM);
        WAVE_SYNC();
      }
      want = on ? (ng < A.prm.n_results ? ng : A.prm.n_results) : 0;
    }
    const uint32_t off = wave_alloc_rows(A, rb, want);
    if (lane == 0) {
      if (t == 0) out->n_genotypes = ng;
      out->row_off[table] = off == GRIM_NONE ? 0 : off;
      out->n_rows[table] = off == GRIM_NONE ? 0 : want;
    }
    if (off != GRIM_NONE && want > 0) {
      if (own) {  // rank the pairs themselves
        double *pr = (double *)W.tab;
        for (uint32_t i = lane; i < n; i += 64) {
          pr[i] = rec[i].prob;
          W.hd[i] = (uint16_t)i;
        }
        WAVE_SYNC();
      }
      for (uint32_t j = lane; j < ng; j += 64) {
        const uint32_t rank = wave_run_rank(W, ng, j);
        if (rank >= want) continue;
        const PairRec r0 = rec[W.hd[j]];
        grim_row row;
        row.a = r0.k1;
        row.b = r0.k2;
        row.prob = gsum[j];
        row.popa = ENT_POP(r0.e1);
        row.popb = ENT_POP(r0.e2);
        A.rows[off + rank] = row;
      }
    }
    WAVE_SYNC();
  }
}

// one wave = one work item at a time; waves are independent (no __syncthreads)
__global__ __launch_bounds__(64) void grim_tables_wave_kernel(DevArgs A) {
  __shared__ WaveTabT1 W;
  const uint32_t n_items = A.queue[9];  // written by the kernels before this one in the stream
  RowBlock rb = {0, 0, GRIM_ROW_GRAB};
  // one item per visit to the (sliced) work counters: the tail stays one item long
  SliceWalk sw = {blockIdx.x % GRIM_NSLICE, 0};
  for (;;) {
    const uint32_t w = slice_next(A.wctr, GRIM_WL_T1, n_items, sw);
    if (w == GRIM_NONE) break;
    const TabWork item = A.t1_list[w];
    tables_wave(A, W, item, rb);
  }
}

// ---- the workgroup kernel ----------------------------------------------------------------------------------------------
template <typename SH>
__device__ inline uint32_t tab_alloc_rows(const DevArgs &A, SH &sh, uint32_t n) {
  if (threadIdx.x == 0) {
    uint32_t off = n ? atomicAdd(A.row_head, n) : 0;
    if (n && off + n > A.row_cap) {
      atomicExch(&A.counters[4], 1ull);
      off = GRIM_NONE;
    }
    sh.bc[1] = off;
  }
  __syncthreads();
  uint32_t off = sh.bc[1];
  __syncthreads();
  return off;
}

// left-to-right sum of v[0..n), n >= 1, by one wave at register speed: 256 values per step sit in four registers per
// lane (the next 256 are loaded before the current ones are added) and are replayed in order through v_readlane
__device__ inline double wave_chain(const double *v, uint32_t n) {
  const int lane = lane_id();
  double x[4], y[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) x[q] = (uint32_t)(64 * q + lane) < n ? v[64 * q + lane] : 0.0;
  double s = 0.0;
  bool first = true;
  for (uint32_t base = 0; base < n; base += 256) {
#pragma unroll
    for (int q = 0; q < 4; ++q) y[q] = base + 256 + 64 * q + lane < n ? v[base + 256 + 64 * q + lane] : 0.0;
    const uint32_t cnt = n - base < 256 ? n - base : 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int lim = (int)cnt - 64 * q < 64 ? (int)cnt - 64 * q : 64;
      for (int j = 0; j < lim; ++j) {
        const double pj = lane_get(x[q], j);
        s = first ? pj : s + pj;
        first = false;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = y[q];
  }
  return s;
}

// groups of the nU pairs under `kind` through hash tables and a radix sort in HBM scratch (the path of round 1, on
// records): gsum / ghead in first-seen order.  Fallback of tab_group_buckets and the GRIM_TABLES_HBM=1 path.
__device__ inline uint32_t tab_group_hbm(const DevArgs &A, TabShared &sh, const Slot &S, const PairRec *rec, uint32_t nU, int kind) {

  const int tid = threadIdx.x;
  uint32_t ng = 0;
  // one population: U is already unique per unordered haplotype pair (the dedup key of the pair pass is
  // {(hap,pop),(hap,pop)}), so every pair is its own group
  if (kind == 2 || (kind == 1 && A.g.P == 1)) {
    ng = nU;
    for (uint32_t u = tid; u < nU; u += GRIM_WG) {
      S.gsum[u] = rec[u].prob;
      S.ghead[u] = u;
    }
    __syncthreads();
  } else {
    uint32_t cap = 64;
    while (cap < 2 * nU) cap <<= 1;
    if (cap > A.tab_cap) cap = A.tab_cap;
    const uint32_t mask = cap - 1;
    for (uint32_t s = tid; s < cap; s += GRIM_WG) {
      S.k0[s] = 0;
      S.k1[s] = 0;
      S.tmin[s] = GRIM_NONE;
    }
    __syncthreads();
    // GRIM_GROUP_NB pairs per thread and step: their (dependent) key gathers are in flight together, then the inserts
    for (uint32_t u0 = tid; u0 < nU; u0 += GRIM_GROUP_NB * GRIM_WG) {
      uint64_t klo[GRIM_GROUP_NB], khi[GRIM_GROUP_NB];
#pragma unroll
      for (int q = 0; q < GRIM_GROUP_NB; ++q) {
        const uint32_t u = u0 + q * GRIM_WG;
        klo[q] = khi[q] = 0;
        if (u < nU) {
          const PairRec pr = rec[u];
          uint32_t h1 = ENT_HAP(pr.e1), h2 = ENT_HAP(pr.e2);
          if (kind == 1) {
            uint32_t lo = h1 < h2 ? h1 : h2, hi = h1 < h2 ? h2 : h1;
            klo[q] = (((uint64_t)lo << 32) | hi) | GRIM_VALID;
          } else {
            uint64_t a = pr.k1, b = pr.k2;
            uint64_t lo = 0, hi = 0;
#pragma unroll
            for (int l = 0; l < GRIM_MAXL; ++l) {
              uint64_t x = (a >> (GRIM_ABITS * l)) & 0xFFF, y = (b >> (GRIM_ABITS * l)) & 0xFFF;
              lo |= (x < y ? x : y) << (GRIM_ABITS * l);
              hi |= (x < y ? y : x) << (GRIM_ABITS * l);
            }
            klo[q] = lo | GRIM_VALID;
            khi[q] = hi | GRIM_VALID;
          }
        }
      }
      bool on[GRIM_GROUP_NB];
      uint32_t slot[GRIM_GROUP_NB];
#pragma unroll
      for (int q = 0; q < GRIM_GROUP_NB; ++q) on[q] = u0 + q * GRIM_WG < nU;
      if (kind == 1)
        tab_insert_n<false, GRIM_GROUP_NB>(S.k0, S.k1, mask, klo, khi, on, slot);
      else
        tab_insert_n<true, GRIM_GROUP_NB>(S.k0, S.k1, mask, klo, khi, on, slot);
#pragma unroll
      for (int q = 0; q < GRIM_GROUP_NB; ++q) {
        const uint32_t u = u0 + q * GRIM_WG;
        if (on[q]) {
          S.Uslot[u] = slot[q];
          atomicMin(&S.tmin[slot[q]], u);
        }
      }
    }
    __syncthreads();
    // heads in first-seen order -> dense group ids; 4 x 256 pairs per barrier round, their (dependent) slot and
    // tmin reads in flight together
    for (uint32_t u0 = 0; u0 < nU; u0 += 4 * GRIM_WG) {
      uint32_t slot[4];
      bool head[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t u = u0 + q * GRIM_WG + tid;
        slot[q] = u < nU ? S.Uslot[u] : 0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t u = u0 + q * GRIM_WG + tid;
        head[q] = u < nU && ALOAD(&S.tmin[slot[q]]) == u;
      }
      uint64_t m[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        m[q] = __ballot(head[q]);
        if (lane_id() == 0) sh.tmp[q * GRIM_NWAVE + wave_id()] = (uint32_t)__popcll(m[q]);
      }
      __syncthreads();
      uint32_t run = ng, base[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) {
          if (w2 == wave_id()) base[q] = run;
          run += sh.tmp[q * GRIM_NWAVE + w2];
        }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (head[q]) {
          const uint32_t gid = base[q] + (uint32_t)__popcll(m[q] & ((1ull << lane_id()) - 1ull));
          S.tgid[slot[q]] = gid;
          S.ghead[gid] = u0 + q * GRIM_WG + tid;
          S.gcnt[gid] = 0;
        }
      ng = run;
      __syncthreads();
    }
    // stable sort of u by group id, then per-group left-to-right sums
    for (uint32_t u0 = tid; u0 < nU; u0 += 4 * GRIM_WG) {
      uint32_t gid[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t u = u0 + q * GRIM_WG;
        gid[q] = u < nU ? S.tgid[S.Uslot[u]] : 0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t u = u0 + q * GRIM_WG;
        if (u < nU) {
          S.ska[u] = gid[q];
          S.sva[u] = u;
          atomicAdd(&S.gcnt[gid[q]], 1u);
        }
      }
    }
    __syncthreads();
    for (uint32_t g = tid; g < ng; g += GRIM_WG) S.gstart[g] = S.gcnt[g];
    __syncthreads();
    wg_scan_array(S.gstart, ng, sh.tmp);
    int w = wg_radix_sort(S.ska, S.sva, S.skb, S.svb, nU, bits_for(ng), sh.hist, sh.tmp);
    const uint32_t *sorted = w ? S.svb : S.sva;
    for (uint32_t g = tid; g < ng; g += GRIM_WG) {
      uint32_t a = S.gstart[g], b = S.gstart[g + 1];
      double s = rec[sorted[a]].prob;
      uint32_t r = a + 1;
      for (; r + 4 <= b; r += 4) {  // four independent gathers in flight, adds stay in order
        double v0 = rec[sorted[r]].prob, v1 = rec[sorted[r + 1]].prob, v2 = rec[sorted[r + 2]].prob, v3 = rec[sorted[r + 3]].prob;
        s = s + v0;
        s = s + v1;
        s = s + v2;
        s = s + v3;
      }
      for (; r < b; ++r) s = s + rec[sorted[r]].prob;
      S.gsum[g] = s;
    }
    __syncthreads();
  }
  return ng;
}

// ranking of ng groups (ids in first-seen order) by their sums: stable sort by probability, bigger first; only rows
// [0, want) of *order_out are valid when want is small (impute.py:24-76)
__device__ inline void tab_rank(const DevArgs &A, TabShared &sh, const Slot &S, const double *gsum, uint32_t ng, uint32_t want,
                                uint32_t **order_out) {
  const int tid = threadIdx.x;
  // ranking: stable sort by probability, bigger first; input order = first-seen order
  if (ng <= 512) {
    // small: rank by counting, sums staged in LDS (the radix histogram area is free here)
    double *ls = (double *)sh.hist;
    for (uint32_t g = tid; g < ng; g += GRIM_WG) ls[g] = gsum[g];
    __syncthreads();
    for (uint32_t g = tid; g < ng; g += GRIM_WG) {
      double s = ls[g];
      uint32_t rank = 0;
      for (uint32_t g2 = 0; g2 < ng; ++g2) {
        double s2 = ls[g2];
        rank += (s2 > s || (s2 == s && g2 < g)) ? 1u : 0u;
      }
      S.sva[rank] = g;
    }
    __syncthreads();
    *order_out = S.sva;
    return;
  }
  for (uint32_t g = tid; g < ng; g += GRIM_WG) {
    S.ska[g] = ~f64_ord(gsum[g]);
    S.sva[g] = g;
  }
  __syncthreads();
  if (want > 0 && want <= 1024 && want * 4 <= ng) {
    // Only rows [0,want) are written.  First try ONE pass: a 4096-bin histogram over the keys' top 12 bits
    // (sign and exponent of the sum) in LDS locates the bin B that holds the want-th smallest key; when the
    // groups of the bins <= B fit the LDS list they are gathered in id order and ranked by counting.
    {
      uint32_t *h12 = sh.hist;
      for (int i = tid; i < 4096; i += GRIM_WG) h12[i] = 0;
      __syncthreads();
      for (uint32_t g0 = tid; g0 < ng; g0 += 4 * GRIM_WG) {
        uint64_t k[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) k[q] = g0 + q * GRIM_WG < ng ? S.ska[g0 + q * GRIM_WG] : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (g0 + q * GRIM_WG < ng) atomicAdd(&h12[(uint32_t)(k[q] >> 52)], 1u);
      }
      __syncthreads();
      uint32_t part = 0;
      for (int e = 0; e < 16; ++e) part += h12[tid * 16 + e];
      uint32_t total;
      const uint32_t before = wg_excl_scan(part, sh.tmp, total);
      if (before < want && want <= before + part) {
        uint32_t cum = before;
#pragma nounroll
        for (int e = 0; e < 16; ++e) {
          const uint32_t c = h12[tid * 16 + e];
          if (cum + c >= want) {
            sh.bc[4] = (uint32_t)(tid * 16 + e);
            sh.bc[5] = cum + c;
            break;
          }
          cum += c;
        }
      }
      __syncthreads();
      const uint32_t B = sh.bc[4], upto = sh.bc[5];  // groups in bins <= B
      __syncthreads();
      if (upto <= 1024) {
        uint64_t *lk = (uint64_t *)sh.hist;      // [1024] keys (the histogram is spent)
        uint32_t *lg = (uint32_t *)(lk + 1024);  // [1024] group ids
        uint32_t taken = 0;
        for (uint32_t g0 = 0; g0 < ng; g0 += GRIM_WG) {
          const uint32_t g = g0 + tid;
          const uint64_t k = g < ng ? S.ska[g] : ~0ull;
          const bool pick = g < ng && (uint32_t)(k >> 52) <= B;
          const uint64_t mp = __ballot(pick);
          if (lane_id() == 0) sh.tmp[wave_id()] = (uint32_t)__popcll(mp);
          __syncthreads();
          uint32_t pbase = taken, ptot = 0;
          for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) {
            const uint32_t t = sh.tmp[w2];
            if (w2 < wave_id()) pbase += t;
            ptot += t;
          }
          if (pick) {
            const uint32_t pos = pbase + (uint32_t)__popcll(mp & ((1ull << lane_id()) - 1ull));
            lk[pos] = k;
            lg[pos] = g;
          }
          taken += ptot;
          __syncthreads();
        }
        // rank by (key asc, id asc); ids were gathered in ascending order
        for (uint32_t i = tid; i < taken; i += GRIM_WG) {
          const uint64_t k = lk[i];
          uint32_t rank = 0;
          for (uint32_t j = 0; j < taken; ++j) {
            const uint64_t k2 = lk[j];
            rank += (k2 < k || (k2 == k && j < i)) ? 1u : 0u;
          }
          if (rank < want) S.svb[rank] = lg[i];
        }
        __syncthreads();
        *order_out = S.svb;
        return;
      }
    }
    // Otherwise MSD radix select finds the want-th smallest key T; groups
    // with key < T plus the first ties (group id order = first-seen order) are gathered in id
    // order and ranked by counting in LDS.
    uint64_t prefix = 0;
    uint32_t remaining = want;
    uint32_t *bins = sh.tmp + GRIM_NWAVE;  // [16]
    for (int shift = 60; shift >= 0; shift -= 4) {
      if (tid < 16) bins[tid] = 0;
      __syncthreads();
      const uint64_t himask = shift == 60 ? 0ull : (~0ull << (shift + 4));
      uint32_t loc[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) loc[d] = 0;
      for (uint32_t g = tid; g < ng; g += GRIM_WG) {
        uint64_t k = S.ska[g];
        if ((k & himask) == (prefix & himask)) {
          uint32_t dg = (uint32_t)(k >> shift) & 15u;
#pragma unroll
          for (int d = 0; d < 16; ++d) loc[d] += (dg == (uint32_t)d) ? 1u : 0u;
        }
      }
#pragma unroll
      for (int d = 0; d < 16; ++d) {
        uint32_t v = loc[d];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane_id() == 0 && v) atomicAdd(&bins[d], v);
      }
      __syncthreads();
      uint32_t cum = 0, dsel = 15;
      for (uint32_t d = 0; d < 16; ++d) {
        if (cum + bins[d] >= remaining) {
          dsel = d;
          break;
        }
        cum += bins[d];
      }
      remaining -= cum;
      prefix |= (uint64_t)dsel << shift;
      __syncthreads();
    }
    const uint64_t T = prefix;  // `remaining` ties with key == T are taken, the earliest ones
    uint32_t taken = 0, ties = 0;
    uint64_t *lk = (uint64_t *)sh.hist;          // [1024] keys
    uint32_t *lg = (uint32_t *)(lk + 1024);      // [1024] group ids
    for (uint32_t g0 = 0; g0 < ng; g0 += GRIM_WG) {
      uint32_t g = g0 + tid;
      uint64_t k = g < ng ? S.ska[g] : ~0ull;
      bool less = g < ng && k < T, tie = g < ng && k == T;
      // ties in id order
      uint64_t mt = __ballot(tie);
      if (lane_id() == 0) sh.tmp[wave_id()] = (uint32_t)__popcll(mt);
      __syncthreads();
      uint32_t tbase = ties, ttot = 0;
      for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) {
        uint32_t t = sh.tmp[w2];
        if (w2 < wave_id()) tbase += t;
        ttot += t;
      }
      __syncthreads();
      bool pick = less || (tie && tbase + (uint32_t)__popcll(mt & ((1ull << lane_id()) - 1ull)) < remaining);
      ties += ttot;
      uint64_t mp = __ballot(pick);
      if (lane_id() == 0) sh.tmp[wave_id()] = (uint32_t)__popcll(mp);
      __syncthreads();
      uint32_t pbase = taken, ptot = 0;
      for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) {
        uint32_t t = sh.tmp[w2];
        if (w2 < wave_id()) pbase += t;
        ptot += t;
      }
      if (pick) {
        uint32_t pos = pbase + (uint32_t)__popcll(mp & ((1ull << lane_id()) - 1ull));
        if (pos < 1024) {
          lk[pos] = k;
          lg[pos] = g;
        }
      }
      taken += ptot;
      __syncthreads();
    }
    // taken == want; rank by (key asc, id asc) -- ids were gathered in ascending order
    for (uint32_t i = tid; i < taken; i += GRIM_WG) {
      uint64_t k = lk[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < taken; ++j) {
        uint64_t k2 = lk[j];
        rank += (k2 < k || (k2 == k && j < i)) ? 1u : 0u;
      }
      S.svb[rank] = lg[i];
    }
    __syncthreads();
    *order_out = S.svb;
    return;
  }
  int w = wg_radix_sort(S.ska, S.sva, S.skb, S.svb, ng, 64, sh.hist, sh.tmp);
  *order_out = w ? S.svb : S.sva;
  return;
}

// ---- work items with more than GRIM_TAB_T1_MAX pairs: three kernels ---------------------------------------------------
// A subject with thousands of accepted pairs is not one workgroup's job: its pairs are dealt into buckets of <= 256
// by the hash of their group key (a group never spans buckets), the buckets of ALL such subjects become work units of one
// wave each (grim_tables_bucket_kernel, 14 waves per CU, the same LDS grouping as the one-wave kernel), and a third
// kernel puts a subject's groups into first-seen order, ranks them and writes the rows.  Population pairs -- few, huge
// groups -- are partitioned by cell (stable radix sort) in the first kernel; a cell's left-to-right sum is a work unit
// of the second (one wave, register-speed chain).
//   grim_tables_split_kernel  : workgroup per item : bucket / cell partition, work units
//   grim_tables_bucket_kernel : wave per unit      : groups of a bucket (head, sum) / sum of a cell
//   grim_tables_merge_kernel  : workgroup per item : first-seen order, ranking, rows

// first kernel, ONE sweep over the item's records for all three partitions (round 3): cell and the twelve hash bits of the
// genotype / haplotype-pair key of every pair, packed into S.svb[u] = cell << 24 | h(pair key) << 12 | h(genotype key).  The
// partitions then count and deal from these four bytes instead of each reading the 32-byte records and hashing again.
__device__ inline void tab_split_ids(const DevArgs &A, const Slot &S, const TabWork &w, bool split0, bool split1) {
  const int P = A.g.P;
  const PairRec *rec = A.ppool + w.off;
  const uint32_t nU = w.n;
  for (uint32_t ub = threadIdx.x; ub < nU; ub += 4 * GRIM_WG) {  // four records in flight per lane
    PairRec r4[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
      if (ub + GRIM_WG * k4 < nU) r4[k4] = rec[ub + GRIM_WG * k4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const uint32_t u = ub + GRIM_WG * k4;
      if (u >= nU) break;
      const PairRec &r = r4[k4];
      const uint32_t a = ENT_POP(r.e1), b = ENT_POP(r.e2);
      uint32_t id = ((a < b ? a : b) * (uint32_t)P + (a < b ? b : a)) << 24;
      uint64_t lo, hi;
      if (split0) {
        tab_key(0, P, r, lo, hi);
        id |= (uint32_t)(tab_hash(lo, hi) >> 40) & (TAB_MAXB - 1);
      }
      if (split1) {
        tab_key(1, P, r, lo, hi);
        id |= ((uint32_t)(tab_hash(lo, hi) >> 40) & (TAB_MAXB - 1)) << 12;
      }
      S.svb[u] = id;
    }
  }
  __syncthreads();
}

// first kernel, one table: bucket of every pair, bucket starts, the pairs dealt out, one work unit per bucket
// ids_shift >= 0: S.svb[u] holds the packed ids tab_split_ids left (this table's twelve hash bits from bit ids_shift); -1: the
// key sweep is this function's own
__device__ inline void tab_split_table(const DevArgs &A, TabShared &sh, const Slot &S, const TabWork &w, uint32_t item, int t, int kind,
                                       int ids_shift = -1) {
  const int tid = threadIdx.x;
  const int P = A.g.P;
  const PairRec *rec = A.ppool + w.off;
  const uint32_t nU = w.n;
  if (nU > GRIM_NWAVE * 65472u) {  // a wave's stretch would not fit its 16-bit counts: this table goes the HBM way
    if (tid == 0) A.taux[item].nb[t] = 0;
    __syncthreads();
    return;
  }
  uint32_t nb = 1;
  while (nb * TAB_DIV < nU && nb < TAB_MAXB) nb <<= 1;
  // STABLE deal: wave w owns the w-th quarter of the item's pairs and keeps its own count / cursor per bucket, so a
  // bucket receives its pairs in increasing pair number -- the bucket kernel then adds a group's probabilities in the
  // reference's order without sorting.
  const int wv = wave_id(), lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t q = ((nU + GRIM_NWAVE * 64 - 1) / (GRIM_NWAVE * 64)) * 64;  // pairs per wave, whole chunks of 64
  uint16_t *wc = sh.wcnt;
  {
    uint32_t *wz = (uint32_t *)wc;
    for (uint32_t k = tid; k < (GRIM_NWAVE * nb + 1) / 2; k += GRIM_WG) wz[k] = 0;
  }
  __syncthreads();
  const uint32_t u0 = wv * q, u1 = u0 + q < nU ? u0 + q : nU;
  const int sft = ids_shift < 0 ? 0 : ids_shift;
  if (ids_shift >= 0) {  // the keys were hashed once for all tables: count from the ids
#pragma unroll 4
    for (uint32_t u = u0 + lane; u < u1; u += 64) {
      const uint32_t k = wv * nb + ((S.svb[u] >> sft) & (nb - 1));
      atomicAdd((uint32_t *)wc + (k >> 1), 1u << (16 * (k & 1)));
    }
  } else
  for (uint32_t ub = u0 + lane; ub < u1; ub += 4 * 64) {  // four records in flight per lane (the stores below may alias them
    PairRec r4[4];                                         // for all the compiler knows: it would not hoist the loads itself)
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
      if (ub + 64 * k4 < u1) r4[k4] = rec[ub + 64 * k4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const uint32_t u = ub + 64 * k4;
      if (u >= u1) break;
      uint64_t lo, hi;
      tab_key(kind, P, r4[k4], lo, hi);
      const uint32_t h12 = (uint32_t)(tab_hash(lo, hi) >> 40) & (TAB_MAXB - 1);  // high bits: the waves' slot hash uses the low ones
      S.svb[u] = h12;  // the bucket is its low bits, however many buckets there are in the end
      const uint32_t k = wv * nb + (h12 & (nb - 1));
      atomicAdd((uint32_t *)wc + (k >> 1), 1u << (16 * (k & 1)));  // q < 65536: a half never carries into its neighbour
    }
  }
  __syncthreads();
  // bucket sizes; per-wave counts become cursors relative to the bucket's start.  A bucket beyond a wave's arena (TAB_NB
  // pairs: big groups gather -- 16 population pairs x 2 phases of one genotype) would send the whole table down the HBM
  // path of the merge kernel, a millisecond for the items that have them: the deal is repeated with twice the buckets
  // instead, up to three times (the bucket numbers are the low bits of what S.svb holds: no record is read again).
  for (int attempt = 0;; ++attempt) {
    if (tid == 0) sh.bc[6] = 0;
    __syncthreads();
    uint32_t big = 0;
    for (uint32_t b2 = tid; b2 < nb; b2 += GRIM_WG) {
      uint32_t run = 0;
      for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) {
        const uint32_t c = wc[w2 * nb + b2];
        wc[w2 * nb + b2] = (uint16_t)run;
        run += c;
      }
      sh.bcnt[b2] = run;
      big |= run > TAB_NB ? 1u : 0u;
    }
    if (big) sh.bc[6] = 1;
    __syncthreads();
    const bool again = sh.bc[6] != 0 && attempt < 3 && nb < TAB_MAXB;
    __syncthreads();
    if (!again) break;
    nb <<= 1;
    {
      uint32_t *wz = (uint32_t *)wc;
      for (uint32_t k = tid; k < (GRIM_NWAVE * nb + 1) / 2; k += GRIM_WG) wz[k] = 0;
    }
    __syncthreads();
    for (uint32_t u = u0 + lane; u < u1; u += 64) {
      const uint32_t k = wv * nb + ((S.svb[u] >> sft) & (nb - 1));
      atomicAdd((uint32_t *)wc + (k >> 1), 1u << (16 * (k & 1)));
    }
    __syncthreads();
  }
  if (tid == 0) {
    uint32_t base = atomicAdd(A.queue + 13, nb + 1), ubase = atomicAdd(A.queue + 14, nb);
    if (base + nb + 1 > A.tboff_cap || ubase + nb > A.tunits_cap) {
      atomicExch(&A.counters[4], 1ull);  // reported like a row-pool overflow: the caller splits the batch
      base = ubase = GRIM_NONE;
    }
    sh.bc[4] = base;
    sh.bc[5] = ubase;
    TabAux &x = A.taux[item];
    x.boff[t] = base;
    x.nb[t] = base == GRIM_NONE ? 0 : nb;
    x.ng[t] = 0;
    x.overflow[t] = 0;
    sh.bcnt[nb] = 0;
  }
  __syncthreads();
  const uint32_t base = sh.bc[4], ubase = sh.bc[5];
  if (base == GRIM_NONE) return;
  {  // exclusive scan in place: bcnt[b] = first position of bucket b
    const uint32_t per = (nb + GRIM_WG - 1) / GRIM_WG;
    uint32_t b0 = tid * per, b1 = b0 + per;
    if (b0 > nb) b0 = nb;
    if (b1 > nb) b1 = nb;
    uint32_t sum = 0, big = 0;
    for (uint32_t b2 = b0; b2 < b1; ++b2) {
      sum += sh.bcnt[b2];
      big |= sh.bcnt[b2] > TAB_NB ? 1u : 0u;
    }
    uint32_t total;
    uint32_t at = wg_excl_scan(sum, sh.tmp, total);
    for (uint32_t b2 = b0; b2 < b1; ++b2) {
      const uint32_t c = sh.bcnt[b2];
      sh.bcnt[b2] = at;
      A.tboff[base + b2] = at;
      at += c;
    }
    if (tid == 0) {
      A.tboff[base + nb] = nU;
      sh.bcnt[nb] = nU;
    }
    if (big) atomicExch(&A.taux[item].overflow[t], 1u);  // a bucket beyond a wave's arena: the merge kernel takes the HBM path
  }
  __syncthreads();
  uint32_t *dst = A.psort + (uint64_t)t * A.tstride + w.off;
  int nbits = 0;
  while ((1u << nbits) < nb) ++nbits;
  // four chunks of 64 per step: their bucket loads and ballots are in flight together, only the cursor updates are serial
  for (uint32_t c0 = u0; c0 < u1; c0 += 4 * 64) {
    uint32_t bk[4];
    bool act[4];
    uint64_t same[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t u = c0 + k * 64 + lane;
      act[k] = u < u1;
      bk[k] = act[k] ? (S.svb[u] >> sft) & (nb - 1) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) same[k] = __ballot(act[k]);
    for (int bit = 0; bit < nbits; ++bit) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool sbit = (bk[k] >> bit) & 1u;
        const uint64_t m = __ballot(act[k] && sbit);
        same[k] &= sbit ? m : ~m;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (act[k]) {
        const uint32_t rank = (uint32_t)__popcll(same[k] & lt), cnt = (uint32_t)__popcll(same[k]);
        const uint32_t cur = wc[wv * nb + bk[k]];
        const uint32_t pos = sh.bcnt[bk[k]] + cur + rank;
        if (pos < nU) dst[pos] = c0 + k * 64 + lane;
        if (rank + 1 == cnt) wc[wv * nb + bk[k]] = (uint16_t)(cur + cnt);
      }
      WAVE_SYNC();
    }
  }
  __syncthreads();
  for (uint32_t b2 = tid; b2 < nb; b2 += GRIM_WG) {
    TabUnit un;
    un.item = item;
    un.tb = ((uint32_t)t << 28) | b2;
    un.off = w.off;
    un.lo = sh.bcnt[b2];
    un.n = sh.bcnt[b2 + 1] - un.lo;
    un.pad[0] = un.pad[1] = un.pad[2] = 0;
    A.tunits[ubase + b2] = un;
  }
  __syncthreads();
}

// first kernel, population pairs: stable partition by cell, probabilities laid out cell after cell, a unit per cell
// have_ids: S.svb[u] holds the packed ids tab_split_ids left (the cell in bits 24..31)
__device__ inline void tab_split_pops(const DevArgs &A, TabShared &sh, const Slot &S, const TabWork &w, uint32_t item, bool have_ids = false) {
  const int tid = threadIdx.x;
  const int P = A.g.P;
  const int ncell = P * P;
  const PairRec *rec = A.ppool + w.off;
  const uint32_t nU = w.n;
  uint32_t *cnt = ncell <= TAB_MAXB ? sh.bcnt : S.gstart;  // cell starts: LDS unless there are more cells than it holds
  for (int c = tid; c <= ncell; c += GRIM_WG) cnt[c] = 0;
  if (tid == 0) {
    uint32_t base = atomicAdd(A.queue + 13, (uint32_t)ncell + 1);
    if (base + ncell + 1 > A.tboff_cap) {
      atomicExch(&A.counters[4], 1ull);
      base = GRIM_NONE;
    }
    sh.bc[4] = base;
    A.taux[item].cell_base = base;
  }
  __syncthreads();
  const uint32_t base = sh.bc[4];
  if (base == GRIM_NONE) return;
  uint32_t *order = A.psort + 2ull * A.tstride + w.off;
  double *pp = A.pprob + w.off;
  if (ncell > 1 && ncell <= 1024 && sh.wcnt && nU <= GRIM_NWAVE * 65472u) {
    // stable partition by cell with wave-owned stretches and per-wave cursors, as tab_split_table deals buckets: no
    // radix passes through HBM, two sweeps over the pairs
    const int wv = wave_id(), lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t q = ((nU + GRIM_NWAVE * 64 - 1) / (GRIM_NWAVE * 64)) * 64;
    const uint32_t nc = (uint32_t)ncell;
    uint16_t *wc = sh.wcnt;
    {
      uint32_t *wz = (uint32_t *)wc;
      for (uint32_t k = tid; k < (GRIM_NWAVE * nc + 1) / 2; k += GRIM_WG) wz[k] = 0;
    }
    __syncthreads();
    const uint32_t u0 = wv * q, u1 = u0 + q < nU ? u0 + q : nU;
    const int sft = have_ids ? 24 : 0;
    if (have_ids) {
#pragma unroll 4
      for (uint32_t u = u0 + lane; u < u1; u += 64) {
        const uint32_t k = wv * nc + (S.svb[u] >> 24);
        atomicAdd((uint32_t *)wc + (k >> 1), 1u << (16 * (k & 1)));
      }
    } else
    for (uint32_t ub = u0 + lane; ub < u1; ub += 4 * 64) {
      uint32_t ea[4], eb[4];
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4)
        if (ub + 64 * k4 < u1) {
          ea[k4] = rec[ub + 64 * k4].e1;
          eb[k4] = rec[ub + 64 * k4].e2;
        }
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const uint32_t u = ub + 64 * k4;
        if (u >= u1) break;
        const uint32_t a = ENT_POP(ea[k4]), b = ENT_POP(eb[k4]);
        const uint32_t cell = (a < b ? a : b) * (uint32_t)P + (a < b ? b : a);
        S.svb[u] = cell;
        const uint32_t k = wv * nc + cell;
        atomicAdd((uint32_t *)wc + (k >> 1), 1u << (16 * (k & 1)));
      }
    }
    __syncthreads();
    // (a cell holds most of an item's pairs as a rule: the cursors need 32 bits, unlike a bucket's)
    uint32_t *cur32 = (uint32_t *)(wc + GRIM_NWAVE * 1024);  // [GRIM_NWAVE][nc], behind the 16-bit counts
    for (uint32_t c = tid; c < nc; c += GRIM_WG) {
      uint32_t run = 0;
      for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) {
        const uint32_t n = wc[w2 * nc + c];
        cur32[w2 * nc + c] = run;
        run += n;
      }
      cnt[c + 1] = run;
    }
    __syncthreads();
    if (tid == 0) {  // cell starts (the cells are few next to the pairs: a serial scan by one thread is noise)
      uint32_t acc = 0;
      for (int c = 0; c <= ncell; ++c) {
        acc += cnt[c];
        cnt[c] = acc;
      }
    }
    __syncthreads();
    int nbits = 0;
    while ((1u << nbits) < nc) ++nbits;
    for (uint32_t c0 = u0; c0 < u1; c0 += 4 * 64) {
      uint32_t ck[4];
      bool act[4];
      uint64_t same[4];
      double pr[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t u = c0 + k * 64 + lane;
        act[k] = u < u1;
        ck[k] = act[k] ? S.svb[u] >> sft : 0u;
        pr[k] = act[k] ? rec[u].prob : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) same[k] = __ballot(act[k]);
      for (int bit = 0; bit < nbits; ++bit) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool sbit = (ck[k] >> bit) & 1u;
          const uint64_t m = __ballot(act[k] && sbit);
          same[k] &= sbit ? m : ~m;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (act[k]) {
          const uint32_t rank = (uint32_t)__popcll(same[k] & lt), n = (uint32_t)__popcll(same[k]);
          const uint32_t cur = cur32[wv * nc + ck[k]];
          const uint32_t pos = cnt[ck[k]] + cur + rank;
          if (pos < nU) {
            order[pos] = c0 + k * 64 + lane;
            pp[pos] = pr[k];
          }
          if (rank + 1 == n) cur32[wv * nc + ck[k]] = cur + n;
        }
        WAVE_SYNC();
      }
    }
    __syncthreads();
  } else if (ncell > 1) {
    for (uint32_t u = tid; u < nU; u += GRIM_WG) {
      const PairRec r = rec[u];
      const uint32_t a = ENT_POP(r.e1), b = ENT_POP(r.e2);
      const uint32_t cell = (a < b ? a : b) * (uint32_t)P + (a < b ? b : a);
      S.ska[u] = cell;
      S.sva[u] = u;
      atomicAdd(&cnt[cell + 1], 1u);
    }
    __syncthreads();
    const int wch = wg_radix_sort(S.ska, S.sva, S.skb, S.svb, nU, bits_for((uint32_t)ncell), sh.hist, sh.tmp);
    const uint32_t *sorted = wch ? S.svb : S.sva;
    if (tid == 0) {  // cell starts (the cells are few next to the pairs: a serial scan by one thread is noise)
      uint32_t acc = 0;
      for (int c = 0; c <= ncell; ++c) {
        acc += cnt[c];
        cnt[c] = acc;
      }
    }
    __syncthreads();
    for (uint32_t r = tid; r < nU; r += GRIM_WG) {
      const uint32_t u = sorted[r];
      order[r] = u;
      pp[r] = rec[u].prob;
    }
  } else {
    if (tid == 0) {
      cnt[0] = 0;
      cnt[1] = nU;
    }
    for (uint32_t r = tid; r < nU; r += GRIM_WG) {
      order[r] = r;
      pp[r] = rec[r].prob;
    }
    __syncthreads();
  }
  for (int c = tid; c <= ncell; c += GRIM_WG) A.tboff[base + c] = cnt[c];
  for (int c = tid; c < ncell; c += GRIM_WG) {
    CellRec cr;
    cr.sum = 0.0;
    cr.first = GRIM_NONE;
    cr.pad = 0;
    A.tcell[base + c] = cr;
    if (cnt[c + 1] > cnt[c]) {  // a work unit per non-empty cell
      const uint32_t k = atomicAdd(A.queue + 14, 1u);
      if (k < A.tunits_cap) {
        TabUnit un;
        un.item = item;
        un.tb = (2u << 28) | (uint32_t)c;
        un.off = w.off;
        un.lo = cnt[c];
        un.n = cnt[c + 1] - cnt[c];
        un.pad[0] = un.pad[1] = un.pad[2] = 0;
        A.tunits[k] = un;
      } else {
        atomicExch(&A.counters[4], 1ull);
      }
    }
  }
  __syncthreads();
}

// ---- work items of up to TW_MAXN pairs: a WAVE splits them, a wave merges them ----------------------------------------
// Nine in ten of the bigger work items of the mixed workloads hold a few hundred to a few thousand pairs: a workgroup
// spends its time on them in barriers (a dozen per table) with three of its four waves idle in the serial stretches, and a
// CU holds three such workgroups.  One wave per item needs no barrier, the three partitions (cells, genotype buckets,
// haplotype-pair buckets) share ONE sweep that computes keys and one that deals the pairs out, and a CU holds twelve
// items.  Same bucket function, same stable deal, same units as tab_split_table / tab_split_pops -- the bucket kernel
// and either merge kernel cannot tell which split kernel an item went through.
#define TW_MAXN 4096u   // pairs of an item a wave splits
#define TW_MAXB 64u     // ... into at most this many buckets per table (a lane each)
#define TW_MAXC 64u     // population cells (P <= 8)
#define TW_MAXG 1024u   // groups the merge wave holds in registers at a time (16 per lane)
#define TW_MERGE_MAXG 16384u  // groups per table of an item the merge wave takes (in batches of TW_MAXG)
static_assert(TW_MAXN / TAB_DIV <= TW_MAXB && TW_MAXB <= 256, "a wave-split item's buckets; the pairs' 8 hash bits");
static_assert(TW_MAXB <= 64 && TW_MAXC <= 64, "one lane per bucket / cell");

struct SplitWave {
  uint8_t bk[2][TW_MAXN];  // eight hash bits of pair u's key in the genotype / haplotype-pair table: the bucket is the low ones
  uint8_t cell[TW_MAXN];   // its population cell
  uint32_t st[2][TW_MAXB + 1], cur[2][TW_MAXB];  // bucket starts (counts shifted by one before the scan), cursors
  uint32_t cst[TW_MAXC + 1], ccur[TW_MAXC];
};

__device__ __forceinline__ bool tw_split_item(const DevArgs &A, const TabWork &w) {
  return w.n <= TW_MAXN && (uint32_t)(A.g.P * A.g.P) <= TW_MAXC && !(A.flags & GRIM_F_TABLES_HBM);
}
// the merge wave takes an item -- whichever kernel split it -- when every table it has to rank was split, kept to its
// buckets, and has few enough groups and rows
__device__ __forceinline__ bool tw_merge_item(const DevArgs &A, const TabWork &w, const TabAux &x) {
  if ((uint32_t)(A.g.P * A.g.P) > TW_MAXC || (A.flags & GRIM_F_TABLES_HBM) || x.cell_base == GRIM_NONE) return false;
  for (int t = 0; t < 2; ++t) {
    if (!((w.mask >> t) & 1u)) continue;
    const bool on = t == 0 ? A.prm.out_muug : A.prm.out_haps;
    if (!(t == 0 || on)) continue;
    if (x.nb[t] == 0 || x.overflow[t] || x.ng[t] > TW_MERGE_MAXG) return false;
    if (on && (x.ng[t] < A.prm.n_results ? x.ng[t] : A.prm.n_results) > 64u) return false;
  }
  return true;
}

// lanes of the chunk with my id (ids below 2^nbits), as a mask
__device__ __forceinline__ uint64_t tw_same(bool act, uint32_t id, int nbits) {
  uint64_t same = __ballot(act);
  for (int bit = 0; bit < nbits; ++bit) {
    const bool sbit = (id >> bit) & 1u;
    const uint64_t m = __ballot(act && sbit);
    same &= sbit ? m : ~m;
  }
  return same;
}

__device__ inline void split_wave(const DevArgs &A, SplitWave &W, const TabWork &w, uint32_t item) {
  const int lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  const int P = A.g.P;
  const uint32_t ncell = (uint32_t)(P * P);
  const PairRec *rec = A.ppool + w.off;
  const uint32_t nU = w.n;
  TabAux &x = A.taux[item];
  bool split[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int kind = t == 0 ? 0 : (A.prm.em_mr ? 2 : 1);
    const bool active = ((w.mask >> t) & 1u) && !(t == 1 && !A.prm.out_haps);
    split[t] = active && !(kind == 2 || (kind == 1 && P == 1));
    if (active && !split[t] && lane == 0) x.nb[t] = 0;  // every pair its own group: nothing to split (the workgroup merge kernel's)
  }
  uint32_t nb0 = 1;
  while (nb0 * TAB_DIV < nU && nb0 < TW_MAXB) nb0 <<= 1;
  uint32_t nbt[2] = {nb0, nb0};  // buckets per table (a table whose buckets overflow gets more, see below)
  int cbits = 0;
  while ((1u << cbits) < ncell) ++cbits;
  // ---- sweep 1: cell and buckets of every pair, counts ---------------------------------------------------------------
  W.st[0][lane] = W.st[1][lane] = 0;
  W.cst[lane] = 0;
  if (lane == 0) W.cst[TW_MAXC] = W.st[0][TW_MAXB] = W.st[1][TW_MAXB] = 0;
  W.cur[0][lane] = W.cur[1][lane] = 0;
  W.ccur[lane] = 0;
  WAVE_SYNC();
  for (uint32_t ub = lane; ub < nU; ub += 4 * 64) {  // four records in flight per lane
    PairRec r4[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
      if (ub + 64 * k4 < nU) r4[k4] = rec[ub + 64 * k4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const uint32_t u = ub + 64 * k4;
      if (u >= nU) break;
      const PairRec &r = r4[k4];
      const uint32_t a = ENT_POP(r.e1), b = ENT_POP(r.e2);
      const uint32_t cell = (a < b ? a : b) * (uint32_t)P + (a < b ? b : a);
      W.cell[u] = (uint8_t)cell;
      atomicAdd(&W.cst[cell + 1], 1u);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (!split[t]) continue;
        uint64_t lo, hi;
        tab_key(t, P, r, lo, hi);  // kind 0 for the genotype table, kind 1 for the haplotype-pair table
        const uint32_t h8 = (uint32_t)(tab_hash(lo, hi) >> 40) & 255u;
        W.bk[t][u] = (uint8_t)h8;
        atomicAdd(&W.st[t][(h8 & (nb0 - 1)) + 1], 1u);
      }
    }
  }
  WAVE_SYNC();
  // ---- starts; bucket-start slots and work units of the whole item in one go -----------------------------------------
  uint32_t big[2] = {0, 0};
  const uint32_t ccount = W.cst[lane + 1];  // lane = cell
  const uint32_t cincl = wave_incl_scan(ccount);
  uint32_t bcount[2] = {0, 0};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (!split[t]) continue;
    for (;;) {
      bcount[t] = lane < (int)nbt[t] ? W.st[t][lane + 1] : 0u;
      big[t] = __ballot(bcount[t] > TAB_NB) != 0 ? 1u : 0u;
      if (!big[t] || nbt[t] >= TW_MAXB) break;
      // a bucket beyond the bucket kernel's arena (big groups gather): twice the buckets, counted again from the hash bits
      nbt[t] <<= 1;
      WAVE_SYNC();
      W.st[t][lane + 1] = 0;
      WAVE_SYNC();
      for (uint32_t u = lane; u < nU; u += 64) atomicAdd(&W.st[t][(W.bk[t][u] & (nbt[t] - 1)) + 1], 1u);
      WAVE_SYNC();
    }
    const uint32_t incl = wave_incl_scan(bcount[t]);
    WAVE_SYNC();
    if (lane < (int)nbt[t]) W.st[t][lane + 1] = incl;
  }
  WAVE_SYNC();
  W.cst[lane + 1] = cincl;
  WAVE_SYNC();
  const uint64_t cell_m = __ballot(lane < (int)ncell && ccount > 0);
  const uint32_t n_tab = (split[0] ? 1u : 0u) + (split[1] ? 1u : 0u);
  const uint32_t nb_all = (split[0] ? nbt[0] : 0u) + (split[1] ? nbt[1] : 0u);
  const uint32_t need_off = ncell + 1 + nb_all + n_tab, need_units = nb_all + (uint32_t)__popcll(cell_m);
  uint32_t base = 0, ubase = 0;
  if (lane == 0) {
    base = atomicAdd(A.queue + 13, need_off);
    ubase = atomicAdd(A.queue + 14, need_units);
    if (base + need_off > A.tboff_cap || ubase + need_units > A.tunits_cap) {
      atomicExch(&A.counters[4], 1ull);  // reported like a row-pool overflow: the caller splits the batch
      base = GRIM_NONE;
    }
    x.cell_base = base;
  }
  base = __shfl(base, 0);
  ubase = __shfl(ubase, 0);
  if (base == GRIM_NONE) {
    if (lane == 0) {
      if (split[0]) x.nb[0] = 0;
      if (split[1]) x.nb[1] = 0;
    }
    return;
  }
  uint32_t tb_base[2] = {GRIM_NONE, GRIM_NONE}, tu_base[2] = {0, 0};
  {
    uint32_t ob = base + ncell + 1, ou = ubase;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!split[t]) continue;
      tb_base[t] = ob;
      tu_base[t] = ou;
      ob += nbt[t] + 1;
      ou += nbt[t];
      if (lane == 0) {
        x.boff[t] = tb_base[t];
        x.nb[t] = nbt[t];
        x.ng[t] = 0;
        x.overflow[t] = big[t];
        A.tboff[tb_base[t] + nbt[t]] = nU;
      }
      if (lane < (int)nbt[t]) A.tboff[tb_base[t] + lane] = W.st[t][lane];
    }
    // population cells: starts, empty records, a unit per non-empty cell
    if (lane < (int)ncell) {
      A.tboff[base + lane] = W.cst[lane];
      CellRec cr;
      cr.sum = 0.0;
      cr.first = GRIM_NONE;
      cr.pad = 0;
      A.tcell[base + lane] = cr;
    }
    if (lane == 0) A.tboff[base + ncell] = nU;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!split[t] || lane >= (int)nbt[t]) continue;
      TabUnit un;
      un.item = item;
      un.tb = ((uint32_t)t << 28) | (uint32_t)lane;
      un.off = w.off;
      un.lo = W.st[t][lane];
      un.n = bcount[t];
      un.pad[0] = un.pad[1] = un.pad[2] = 0;
      A.tunits[tu_base[t] + lane] = un;
    }
    if ((cell_m >> lane) & 1ull) {
      TabUnit un;
      un.item = item;
      un.tb = (2u << 28) | (uint32_t)lane;
      un.off = w.off;
      un.lo = W.cst[lane];
      un.n = ccount;
      un.pad[0] = un.pad[1] = un.pad[2] = 0;
      A.tunits[ou + (uint32_t)__popcll(cell_m & lt)] = un;
    }
  }
  int nbits[2] = {0, 0};
#pragma unroll
  for (int t = 0; t < 2; ++t)
    while ((1u << nbits[t]) < nbt[t]) ++nbits[t];
  // ---- sweep 2: the stable deal, chunk after chunk in pair order ------------------------------------------------------
  uint32_t *order = A.psort + 2ull * A.tstride + w.off;
  double *pp = A.pprob + w.off;
  uint32_t *dst0 = A.psort + w.off, *dst1 = A.psort + (uint64_t)A.tstride + w.off;
  for (uint32_t cb = 0; cb < nU; cb += 4 * 64) {
   double pr4[4];
#pragma unroll
   for (int k4 = 0; k4 < 4; ++k4) pr4[k4] = cb + 64 * k4 + lane < nU ? rec[cb + 64 * k4 + lane].prob : 0.0;
#pragma unroll
   for (int k4 = 0; k4 < 4; ++k4) {
    const uint32_t c0 = cb + 64 * k4;
    if (c0 >= nU) break;
    const uint32_t u = c0 + lane;
    const bool act = u < nU;
    const double pr = pr4[k4];
    if (ncell > 1) {
      const uint32_t id = act ? W.cell[u] : 0u;
      const uint64_t same = tw_same(act, id, cbits);
      if (act) {
        const uint32_t rank = (uint32_t)__popcll(same & lt), cnt = (uint32_t)__popcll(same);
        const uint32_t cur = W.ccur[id];
        const uint32_t pos = W.cst[id] + cur + rank;
        if (pos < nU) {
          order[pos] = u;
          pp[pos] = pr;
        }
        if (rank + 1 == cnt) W.ccur[id] = cur + cnt;
      }
    } else if (act) {
      order[u] = u;
      pp[u] = pr;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!split[t]) continue;
      const uint32_t id = act ? W.bk[t][u] & (nbt[t] - 1) : 0u;
      const uint64_t same = tw_same(act, id, nbits[t]);
      if (act) {
        const uint32_t rank = (uint32_t)__popcll(same & lt), cnt = (uint32_t)__popcll(same);
        const uint32_t cur = W.cur[t][id];
        const uint32_t pos = W.st[t][id] + cur + rank;
        if (pos < nU) (t == 0 ? dst0 : dst1)[pos] = u;
        if (rank + 1 == cnt) W.cur[t][id] = cur + cnt;
      }
    }
    WAVE_SYNC();
   }
  }
}

__global__ __launch_bounds__(64) void grim_tables_split_wave_kernel(DevArgs A) {
  __shared__ SplitWave W;
  const uint32_t n_items = A.queue[10];
  SliceWalk sw = {blockIdx.x % GRIM_NSLICE, 0};
  for (;;) {
    const uint32_t item = slice_next(A.wctr, GRIM_WL_SPLIT_WAVE, n_items, sw);
    if (item == GRIM_NONE) break;
    const TabWork w = A.t2_list[item];
    if (!tw_split_item(A, w)) continue;  // the workgroup split kernel's
    split_wave(A, W, w, item);
    WAVE_SYNC();
  }
}

// merge of an item the wave split: cells ranked across the lanes; a table's groups (<= TW_MAXG, any order) sit 16 to a
// lane in registers and the rows wanted are picked best first -- round r takes the best group after round r-1's in the
// order (sum descending, first pair ascending), the order tab_rank_grp ranks by.
__device__ inline void merge_wave(const DevArgs &A, const TabWork &w, const TabAux &x, RowBlock &rb) {
  const int lane = lane_id();
  const int P = A.g.P;
  const uint32_t ncell = (uint32_t)(P * P);
  const PairRec *rec = A.ppool + w.off;
  const uint32_t mask = w.mask;
  grim_subject_result *out = A.res + w.si;
  {
    CellRec me;
    me.sum = 0.0;
    me.first = GRIM_NONE;
    if (lane < (int)ncell) me = A.tcell[x.cell_base + lane];
    const bool has = me.first != GRIM_NONE;
    const uint32_t nq = (uint32_t)__popcll(__ballot(has));
    uint32_t rank = 0;
    for (uint32_t c2 = 0; c2 < ncell; ++c2) {
      const double os = __shfl(me.sum, (int)c2);
      const uint32_t of = __shfl(me.first, (int)c2);
      if (of != GRIM_NONE && (int)c2 != lane && (os > me.sum || (os == me.sum && of < me.first))) ++rank;
    }
    const uint32_t nrow = nq < A.prm.n_pop_results ? nq : A.prm.n_pop_results;
    for (int t = 0; t < 2; ++t) {
      if (!((mask >> t) & 1u)) continue;
      const int table = t == 0 ? GRIM_T_UMUG_POPS : GRIM_T_PMUG_POPS;
      const bool on = t == 0 ? A.prm.out_muug : A.prm.out_haps;
      uint32_t want = nrow;
      if (t == 1 && A.prm.em_mr) want = nq < 1 ? nq : 1;
      if (!on) want = 0;
      const uint32_t off = wave_alloc_rows(A, rb, want);
      if (lane == 0) {
        out->row_off[table] = off == GRIM_NONE ? 0 : off;
        out->n_rows[table] = off == GRIM_NONE ? 0 : want;
      }
      if (off == GRIM_NONE || want == 0 || !has || rank >= want) continue;
      const PairRec pr = rec[me.first];
      uint32_t a = ENT_POP(pr.e1), b = ENT_POP(pr.e2);
      if (t == 0 && A.prm.pop_rank[a] > A.prm.pop_rank[b]) {
        const uint32_t y = a;
        a = b;
        b = y;
      }
      grim_row r;
      r.a = a; r.b = b; r.prob = me.sum; r.popa = a; r.popb = b;
      A.rows[off + rank] = r;
    }
  }
  for (int t = 0; t < 2; ++t) {
    if (!((mask >> t) & 1u)) continue;
    const int table = t == 0 ? GRIM_T_UMUG : GRIM_T_PMUG;
    const bool on = t == 0 ? A.prm.out_muug : A.prm.out_haps;
    uint32_t ng = 0, want = 0;
    if (t == 0 || on) {
      ng = x.ng[t];
      want = on ? (ng < A.prm.n_results ? ng : A.prm.n_results) : 0;
    }
    const uint32_t off = wave_alloc_rows(A, rb, want);
    if (lane == 0) {
      if (t == 0) out->n_genotypes = ng;
      out->row_off[table] = off == GRIM_NONE ? 0 : off;
      out->n_rows[table] = off == GRIM_NONE ? 0 : want;
    }
    if (off == GRIM_NONE || want == 0) continue;
    const GrpRec *grp = A.pgrp + (uint64_t)t * A.tstride + w.off;
    constexpr int PER = (int)(TW_MAXG / 64);
    // lane r keeps the r-th best group seen so far (ws, wh); a batch of TW_MAXG groups at a time joins them and the best
    // `want` of the union are picked again
    double ws = -1.0;
    uint32_t wh = GRIM_NONE;
    for (uint32_t g0 = 0; g0 < ng; g0 += TW_MAXG) {
      double s[PER + 1];
      uint32_t h[PER + 1];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const uint32_t g = g0 + (uint32_t)lane + 64u * (uint32_t)k;
        s[k] = -1.0;  // below every sum
        h[k] = GRIM_NONE;
        if (g < ng) {
          const GrpRec r = grp[g];
          s[k] = r.sum;
          h[k] = r.head;
        }
      }
      s[PER] = ws;  // the best of the batches before
      h[PER] = wh;
      double prev_s = 0.0;
      uint32_t prev_h = 0;
      for (uint32_t r = 0; r < want; ++r) {
        double bs = -2.0;
        uint32_t bh = GRIM_NONE;
#pragma unroll
        for (int k = 0; k <= PER; ++k) {
          const bool after = r == 0 || s[k] < prev_s || (s[k] == prev_s && h[k] > prev_h);
          const bool better = s[k] >= 0.0 && after && (s[k] > bs || (s[k] == bs && h[k] < bh));
          bs = better ? s[k] : bs;
          bh = better ? h[k] : bh;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          const double os = __shfl_xor(bs, d);
          const uint32_t oh = __shfl_xor(bh, d);
          const bool take = os > bs || (os == bs && oh < bh);
          bs = take ? os : bs;
          bh = take ? oh : bh;
        }
        prev_s = bs;
        prev_h = bh;
        if (lane == (int)r) {
          ws = bs;
          wh = bh;
        }
      }
    }
    if (lane < (int)want) {
      const PairRec pr = rec[wh];
      grim_row row;
      row.a = pr.k1;
      row.b = pr.k2;
      row.prob = ws;
      row.popa = ENT_POP(pr.e1);
      row.popb = ENT_POP(pr.e2);
      A.rows[off + lane] = row;
    }
  }
}

__global__ __launch_bounds__(64) void grim_tables_merge_wave_kernel(DevArgs A) {
  const uint32_t n_items = A.queue[10];
  RowBlock rb = {0, 0, GRIM_ROW_GRAB};
  SliceWalk sw = {blockIdx.x % GRIM_NSLICE, 0};
  for (;;) {
    const uint32_t item = slice_next(A.wctr, GRIM_WL_MERGE_WAVE, n_items, sw);
    if (item == GRIM_NONE) break;
    const TabWork w = A.t2_list[item];
    const TabAux x = A.taux[item];
    if (!tw_merge_item(A, w, x)) continue;  // the workgroup merge kernel's
    merge_wave(A, w, x, rb);
  }
}

#ifndef GRIM_TAB_WG_PER_CU
#define GRIM_TAB_WG_PER_CU 3
#endif
union SplitArena {  // the cell partition's radix sort and the bucket partition's per-wave cursors never overlap in time
  WgArena a;
  uint16_t wcnt[GRIM_NWAVE * TAB_MAXB];
};
__global__ __launch_bounds__(GRIM_WG, GRIM_TAB_WG_PER_CU) void grim_tables_split_kernel(DevArgs A) {
  __shared__ TabShared sh;
  __shared__ SplitArena arena;
  const int tid = threadIdx.x;
  const uint32_t n_items = A.queue[10];
  Slot S = make_slot(A, blockIdx.x);
  if (tid == 0) {
    sh.hist = arena.a.hist;
    sh.qprob = arena.a.qprob;
    sh.qcell = arena.a.qcell;
    sh.wcnt = arena.wcnt;
  }
  __syncthreads();
  SliceWalk sw = {blockIdx.x % GRIM_NSLICE, 0};  // (wave 0's: the others follow what it finds)
  for (;;) {
    if (tid < 64) {
      const uint32_t nx = slice_next(A.wctr, GRIM_WL_SPLIT, n_items, sw);
      if (tid == 0) sh.bc[3] = nx;
    }
    __syncthreads();
    const uint32_t item = sh.bc[3];
    __syncthreads();
    if (item == GRIM_NONE) break;
    if (tid < (int)(sizeof(TabWork) / 4)) ((uint32_t *)&sh.work)[tid] = ((const uint32_t *)&A.t2_list[item])[tid];
    __syncthreads();
    const TabWork w = sh.work;
    if (tw_split_item(A, w)) continue;  // grim_tables_split_wave_kernel's
    const unsigned long long t_item = STAMP_NOW();
    (void)t_item;
    if (tid == 0) atomicAdd(A.queue + 12, 1u);  // (GRIM_DEBUG_CLASSES=1 prints it)
    const bool hbm_way = (A.flags & GRIM_F_TABLES_HBM) != 0;
    const bool split0 = (w.mask & 1u) && !hbm_way;
    const bool split1 = ((w.mask >> 1) & 1u) && A.prm.out_haps && !(A.prm.em_mr || A.g.P == 1) && !hbm_way;
    // one key sweep for the three partitions when the ids fit the packed word and every partition deals from LDS counts
    const uint32_t ncell_u = (uint32_t)(A.g.P * A.g.P);
    const bool fused = (split0 || split1) && ncell_u > 1 && ncell_u <= 256 && TAB_MAXB <= 4096 && w.n <= GRIM_NWAVE * 65472u;
    if (fused) tab_split_ids(A, S, w, split0, split1);
    tab_split_pops(A, sh, S, w, item, fused);
    for (int t = 0; t < 2; ++t) {
      if (!((w.mask >> t) & 1u)) continue;
      if (t == 1 && !A.prm.out_haps) continue;
      const int kind = t == 0 ? 0 : (A.prm.em_mr ? 2 : 1);
      if (!(t == 0 ? split0 : split1)) {
        if (tid == 0) A.taux[item].nb[t] = 0;  // every pair its own group / HBM path: nothing to split
        continue;
      }
      tab_split_table(A, sh, S, w, item, t, kind, fused ? 12 * t : -1);
    }
    __syncthreads();
    HIST(4, w.n, 1);
    HIST(5, w.n, (STAMP_NOW() - t_item) / 100);
  }
}

// second kernel: one wave = one work unit at a time.  The units are dealt to the waves round robin: a shared work
// counter is ONE address, and an atomic on it costs ~12 ns whoever issues it -- 800 000 units would spend 5 ms there.
// queue[15] = units done by earlier launches of the run (the finish kernel moves it up).
__global__ __launch_bounds__(64) void grim_tables_bucket_kernel(DevArgs A) {
  __shared__ WaveTab<TAB_NB> W;
  const uint32_t n_units = A.queue[14] < A.tunits_cap ? A.queue[14] : A.tunits_cap;
  const int lane = lane_id();
  const int P = A.g.P;
  constexpr uint32_t UM = (1u << TAB_SH) - 1u;
  // XCD-aware: blocks with equal blockIdx % 8 share an XCD (and its L2), and the buckets of one work item -- neighbours
  // in the unit list -- gather from the same records, so each group of blocks takes a contiguous stretch of every round
  // of units: an item's records are then pulled into one L2 instead of eight
  const uint32_t nwg = gridDim.x, q8 = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8;
  const uint32_t vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + blockIdx.x / 8;
  // The chain unit header -> pair numbers -> records -> (grouping) -> group counter -> group records is a row of dependent
  // memory round trips, and a unit is small (~80 pairs): taken one after the other they ARE the kernel's time.  So the loop is
  // a software pipeline over this wave's units i = 0, 1, ...: while unit i is grouped in LDS, the records of unit i+1, the
  // pair numbers of unit i+2 and the header of unit i+3 are on their way, and the groups of unit i-1 are written -- its
  // place in the table's group list (an atomic on the item's counter) was asked for a unit ago.
  constexpr int NC = TAB_NB / 64;
  const uint32_t k0 = A.queue[15] + vid;
  auto load_hdr = [&](uint32_t kk, TabUnit &u) {
    u.item = 0; u.tb = 0; u.off = 0; u.lo = 0; u.n = 0;
    if (kk < n_units) u = A.tunits[kk];
  };
  auto is_bucket = [&](const TabUnit &u) { return (u.tb >> 28) < 2 && u.n > 0 && u.n <= TAB_NB; };
  auto load_idx = [&](const TabUnit &u, uint32_t (&idx)[NC]) {
    if (!is_bucket(u)) return;
    const uint32_t *src = A.psort + (uint64_t)(u.tb >> 28) * A.tstride + u.off + u.lo;
#pragma unroll
    for (int i = 0; i < NC; ++i) idx[i] = (uint32_t)(lane + 64 * i) < u.n ? src[lane + 64 * i] : 0u;
  };
  auto load_rec = [&](const TabUnit &u, const uint32_t (&idx)[NC], PairRec (&R)[NC]) {
    if (!is_bucket(u)) return;
    const PairRec *rec = A.ppool + u.off;
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if ((uint32_t)(lane + 64 * i) < u.n) R[i] = rec[idx[i]];
  };
  TabUnit un, un1, un2, un3;
  uint32_t idx[NC] = {}, idx1[NC] = {}, idx2[NC] = {};
  PairRec R[NC], R1[NC];
  load_hdr(k0, un);
  load_hdr(k0 + nwg, un1);
  load_hdr(k0 + 2 * nwg, un2);
  load_idx(un, idx);
  load_idx(un1, idx1);
  load_rec(un, idx, R);
  // groups of the unit before, waiting for their place: sums and first pairs, 64 j + lane
  double ps[NC];
  uint32_t ph[NC];
  uint32_t p_n = 0, p_g0 = 0;
  GrpRec *p_dst = nullptr;
  for (uint32_t k = k0; k < n_units; k += nwg) {
    load_rec(un1, idx1, R1);
    load_idx(un2, idx2);
    load_hdr(k + 3 * nwg, un3);
    if (p_n) {  // the unit before: its place has arrived
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const uint32_t j = (uint32_t)(lane + 64 * i);
        if (j < p_n) {
          GrpRec gr;
          gr.sum = ps[i];
          gr.head = ph[i];
          gr.pad = 0;
          p_dst[p_g0 + j] = gr;
        }
      }
      p_n = 0;
    }
    TabAux &x = A.taux[un.item];
    const uint32_t t = un.tb >> 28, n = un.n;
    if (n == 0) {
    } else if (t == 2) {  // a population cell: the left-to-right sum of its probabilities
      const double s = wave_chain(A.pprob + un.off + un.lo, n);
      if (lane == 0) {
        CellRec cr;
        cr.sum = s;
        cr.first = A.psort[2ull * A.tstride + un.off + un.lo];  // stable partition: the cell's first pair
        cr.pad = 0;
        A.tcell[x.cell_base + (un.tb & 0x0FFFFFFFu)] = cr;
      }
    } else if (n > TAB_NB) {  // a bucket the arena cannot hold: the merge kernel takes the HBM path for this table
      if (lane == 0) atomicExch(&x.overflow[t], 1u);
    } else {
      const uint32_t nruns = wave_group_pairs(W, t == 0 ? 0 : 1, P, R, idx, n);
      uint32_t g0 = 0;
      if (lane == 0) g0 = atomicAdd(&x.ng[t], nruns);
      const double *gs = (const double *)W.tab;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const uint32_t j = (uint32_t)(lane + 64 * i);
        if (j < nruns) {
          ps[i] = gs[j];
          ph[i] = W.skey[W.rs[j]] & UM;
        }
      }
      WAVE_SYNC();
      p_n = nruns;
      p_g0 = __shfl(g0, 0);
      p_dst = A.pgrp + (uint64_t)t * A.tstride + un.off;
    }
    un = un1;
    un1 = un2;
    un2 = un3;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      idx[i] = idx1[i];
      idx1[i] = idx2[i];
      R[i] = R1[i];
    }
  }
  if (p_n) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const uint32_t j = (uint32_t)(lane + 64 * i);
      if (j < p_n) {
        GrpRec gr;
        gr.sum = ps[i];
        gr.head = ph[i];
        gr.pad = 0;
        p_dst[p_g0 + j] = gr;
      }
    }
  }
}

// third kernel, one table: the groups the bucket kernel found (any order) -> gsum / ghead in first-seen order.
// group id = number of groups whose first pair comes earlier = rank of its first pair among the first pairs (a bitmap
// over the pair numbers and its prefix popcounts)
__device__ inline void tab_merge_groups(const DevArgs &A, TabShared &sh, const Slot &S, const GrpRec *grp, uint32_t ng, uint32_t nU) {
  const int tid = threadIdx.x;
  const uint32_t nw = (nU + 31) / 32;
  uint32_t *bm = S.tgid, *pre = S.gcnt;
  for (uint32_t w = tid; w < nw; w += GRIM_WG) bm[w] = 0;
  __syncthreads();
  for (uint32_t g = tid; g < ng; g += GRIM_WG) {
    const uint32_t h = grp[g].head;
    atomicOr(&bm[h >> 5], 1u << (h & 31));
  }
  __syncthreads();
  {
    const uint32_t per = (nw + GRIM_WG - 1) / GRIM_WG;
    uint32_t w0 = tid * per, w1 = w0 + per;
    if (w0 > nw) w0 = nw;
    if (w1 > nw) w1 = nw;
    uint32_t s = 0;
    for (uint32_t w = w0; w < w1; ++w) s += (uint32_t)__popc(bm[w]);
    uint32_t total;
    uint32_t base = wg_excl_scan(s, sh.tmp, total);
    for (uint32_t w = w0; w < w1; ++w) {
      pre[w] = base;
      base += (uint32_t)__popc(bm[w]);
    }
  }
  __syncthreads();
  for (uint32_t g = tid; g < ng; g += GRIM_WG) {
    const GrpRec gr = grp[g];
    const uint32_t h = gr.head;
    const uint32_t id = pre[h >> 5] + (uint32_t)__popc(bm[h >> 5] & ((1u << (h & 31)) - 1u));
    S.gsum[id] = gr.sum;
    S.ghead[id] = h;
  }
  __syncthreads();
}

// Ranking straight from the bucket kernel's group records (any order): the first-seen order only ever breaks ties, and the
// order of the groups' FIRST PAIRS is that order -- so the rows wanted are the `want` smallest of (key of the sum, head),
// and the renumbering pass (tab_merge_groups) is not needed.  Up to three 12-bit histogram levels narrow the candidates
// down to what the LDS list holds; false: not resolved this way (the caller renumbers and ranks the old way).
__device__ inline bool tab_rank_grp(const DevArgs &A, TabShared &sh, const Slot &S, const GrpRec *grp, uint32_t ng, uint32_t want,
                                    uint32_t **order_out) {
  const int tid = threadIdx.x;
  if (ng <= 512) {
    double *ls = (double *)sh.hist;            // [512]
    uint32_t *lh = (uint32_t *)(ls + 512);     // [512]
    for (uint32_t g = tid; g < ng; g += GRIM_WG) {
      const GrpRec r = grp[g];
      ls[g] = r.sum;
      lh[g] = r.head;
    }
    __syncthreads();
    for (uint32_t g = tid; g < ng; g += GRIM_WG) {
      const double s = ls[g];
      const uint32_t h = lh[g];
      uint32_t rank = 0;
      for (uint32_t g2 = 0; g2 < ng; ++g2) {
        const double s2 = ls[g2];
        rank += (s2 > s || (s2 == s && lh[g2] < h)) ? 1u : 0u;
      }
      if (rank < want) S.sva[rank] = g;
    }
    __syncthreads();
    *order_out = S.sva;
    return true;
  }
  if (!(want > 0 && want <= 512 && want * 4 <= ng)) return false;
  uint32_t *h12 = sh.hist;
  uint64_t prefix = 0, himask = 0, limit = 0;
  uint32_t need = want, below = 0;
  bool resolved = false;
  for (int shift = 52; shift >= 28 && !resolved; shift -= 12) {
    for (int i = tid; i < 4096; i += GRIM_WG) h12[i] = 0;
    __syncthreads();
#pragma unroll 4
    for (uint32_t g = tid; g < ng; g += GRIM_WG) {
      const uint64_t k = ~f64_ord(grp[g].sum);
      if ((k & himask) == prefix) atomicAdd(&h12[(uint32_t)(k >> shift) & 4095u], 1u);
    }
    __syncthreads();
    uint32_t part = 0;
    for (int e = 0; e < 16; ++e) part += h12[tid * 16 + e];
    uint32_t total;
    const uint32_t before = wg_excl_scan(part, sh.tmp, total);
    if (before < need && need <= before + part) {
      uint32_t cum = before;
#pragma nounroll
      for (int e = 0; e < 16; ++e) {
        const uint32_t c = h12[tid * 16 + e];
        if (cum + c >= need) {
          sh.bc[4] = (uint32_t)(tid * 16 + e);
          sh.bc[5] = cum;      // groups of this level in the bins before B
          sh.bc[6] = c;        // groups in bin B
          break;
        }
        cum += c;
      }
    }
    __syncthreads();
    const uint32_t B = sh.bc[4], cum_before = sh.bc[5], in_b = sh.bc[6];
    __syncthreads();
    if (below + cum_before + in_b <= 1024) {
      limit = prefix | ((uint64_t)B << shift) | ((1ull << shift) - 1ull);
      resolved = true;
    } else {
      below += cum_before;
      need -= cum_before;
      prefix |= (uint64_t)B << shift;
      himask |= 4095ull << shift;
    }
  }
  if (!resolved) return false;
  // every group whose key is at most `limit`: at most 1024, any order (the head decides ties, not the position)
  uint64_t *lk = (uint64_t *)sh.hist;      // [1024]
  uint32_t *lh = (uint32_t *)(lk + 1024);  // [1024]
  uint32_t *lg = lh + 1024;                // [1024]
  if (tid == 0) sh.bc[4] = 0;
  __syncthreads();
#pragma unroll 4
  for (uint32_t g = tid; g < ng; g += GRIM_WG) {
    const GrpRec r = grp[g];
    const uint64_t k = ~f64_ord(r.sum);
    if (k <= limit) {
      const uint32_t pos = atomicAdd(&sh.bc[4], 1u);
      if (pos < 1024) {
        lk[pos] = k;
        lh[pos] = r.head;
        lg[pos] = g;
      }
    }
  }
  __syncthreads();
  const uint32_t taken = sh.bc[4] < 1024u ? sh.bc[4] : 1024u;
  for (uint32_t i = tid; i < taken; i += GRIM_WG) {
    const uint64_t k = lk[i];
    const uint32_t h = lh[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < taken; ++j) {
      const uint64_t k2 = lk[j];
      rank += (k2 < k || (k2 == k && lh[j] < h)) ? 1u : 0u;
    }
    if (rank < want) S.svb[rank] = lg[i];
  }
  __syncthreads();
  *order_out = S.svb;
  return true;
}

// third kernel, population pairs: rank the non-empty cells (bigger sum first, earlier first pair on ties), write the rows
__device__ inline void tab_merge_pops(const DevArgs &A, TabShared &sh, const TabWork &w, const TabAux &x, grim_subject_result *out) {
  const int tid = threadIdx.x;
  const int P = A.g.P;
  const int ncell = P * P;
  const PairRec *rec = A.ppool + w.off;
  const uint32_t mask = w.mask;
  if (x.cell_base == GRIM_NONE) return;
  const CellRec *cells = A.tcell + x.cell_base;
  if (tid == 0) sh.bc[2] = 0;
  __syncthreads();
  for (int c = tid; c < ncell; c += GRIM_WG)
    if (cells[c].first != GRIM_NONE) atomicAdd(&sh.bc[2], 1u);
  __syncthreads();
  const uint32_t nq = sh.bc[2];
  __syncthreads();
  const uint32_t nrow = nq < A.prm.n_pop_results ? nq : A.prm.n_pop_results;
  for (int t = 0; t < 2; ++t) {
    if (!((mask >> t) & 1u)) continue;  // this half of the tables belongs to another pass
    const int table = t == 0 ? GRIM_T_UMUG_POPS : GRIM_T_PMUG_POPS;
    const bool on = t == 0 ? A.prm.out_muug : A.prm.out_haps;
    uint32_t want = nrow;
    if (t == 1 && A.prm.em_mr) want = nq < 1 ? nq : 1;  // hap_pop_pair mode: the single best pair (impute.py:2088)
    if (!on) want = 0;
    const uint32_t off = tab_alloc_rows(A, sh, want);
    if (tid == 0) {
      out->row_off[table] = off == GRIM_NONE ? 0 : off;
      out->n_rows[table] = off == GRIM_NONE ? 0 : want;
    }
    if (off == GRIM_NONE || want == 0) continue;
    for (int c = tid; c < ncell; c += GRIM_WG) {
      const CellRec me = cells[c];
      if (me.first == GRIM_NONE) continue;
      uint32_t rank = 0;
      for (int c2 = 0; c2 < ncell; ++c2) {
        const CellRec o = cells[c2];
        if (o.first == GRIM_NONE || c2 == c) continue;
        if (o.sum > me.sum || (o.sum == me.sum && o.first < me.first)) ++rank;
      }
      if (rank >= want) continue;
      const PairRec pr = rec[me.first];
      uint32_t a = ENT_POP(pr.e1), b = ENT_POP(pr.e2);
      if (t == 0 && A.prm.pop_rank[a] > A.prm.pop_rank[b]) {
        const uint32_t y = a;
        a = b;
        b = y;
      }
      grim_row r;
      r.a = a;
      r.b = b;
      r.prob = me.sum;
      r.popa = a;
      r.popb = b;
      A.rows[off + rank] = r;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(GRIM_WG, GRIM_TAB_WG_PER_CU) void grim_tables_merge_kernel(DevArgs A) {
  __shared__ TabShared sh;
  __shared__ WgArena arena;
  const int tid = threadIdx.x;
  const uint32_t n_items = A.queue[10];
  Slot S = make_slot(A, blockIdx.x);
  if (tid == 0) {
    sh.hist = arena.hist;
    sh.qprob = arena.qprob;
    sh.qcell = arena.qcell;
  }
  __syncthreads();
  SliceWalk sw = {blockIdx.x % GRIM_NSLICE, 0};  // (wave 0's: the others follow what it finds)
  for (;;) {
    if (tid < 64) {
      const uint32_t nx = slice_next(A.wctr, GRIM_WL_MERGE, n_items, sw);
      if (tid == 0) sh.bc[3] = nx;
    }
    __syncthreads();
    const uint32_t item = sh.bc[3];
    __syncthreads();
    if (item == GRIM_NONE) break;
    if (tid < (int)(sizeof(TabWork) / 4)) ((uint32_t *)&sh.work)[tid] = ((const uint32_t *)&A.t2_list[item])[tid];
    __syncthreads();
    const TabWork w = sh.work;
    const TabAux x = A.taux[item];
    if (tw_merge_item(A, w, x)) continue;  // grim_tables_merge_wave_kernel's
    const unsigned long long t_item = STAMP_NOW();
    (void)t_item;
    if (tid == 0) {  // (GRIM_DEBUG_CLASSES=1 prints these)
      atomicAdd(A.queue + 21, 1u);
      if (x.overflow[0] | x.overflow[1]) atomicAdd(A.queue + 22, 1u);
      atomicMax(A.queue + 23, w.n);
    }
    const PairRec *rec = A.ppool + w.off;
    const uint32_t nU = w.n;
    grim_subject_result *out = A.res + w.si;
    tab_merge_pops(A, sh, w, x, out);
    for (int t = 0; t < 2; ++t) {
      if (!((w.mask >> t) & 1u)) continue;
      const int table = t == 0 ? GRIM_T_UMUG : GRIM_T_PMUG;
      const bool on = t == 0 ? A.prm.out_muug : A.prm.out_haps;
      uint32_t ng = 0, want = 0;
      uint32_t *order = nullptr;
      const GrpRec *grp = nullptr;  // not null: `order` indexes the bucket kernel's group records
      if (t == 0 || on) {
        const int kind = t == 0 ? 0 : (A.prm.em_mr ? 2 : 1);
        if (x.nb[t] == 0 || x.overflow[t]) {
          ng = tab_group_hbm(A, sh, S, rec, nU, kind);  // own groups, a bucket that overflowed, or GRIM_TABLES_HBM=1
          want = on ? (ng < A.prm.n_results ? ng : A.prm.n_results) : 0;
          if (want) tab_rank(A, sh, S, S.gsum, ng, want, &order);
        } else {
          ng = x.ng[t];
          grp = A.pgrp + (uint64_t)t * A.tstride + w.off;
          want = on ? (ng < A.prm.n_results ? ng : A.prm.n_results) : 0;
          if (want && !tab_rank_grp(A, sh, S, grp, ng, want, &order)) {
            tab_merge_groups(A, sh, S, grp, ng, nU);  // many rows wanted, or too many groups near the cut: first-seen ids, full ranking
            tab_rank(A, sh, S, S.gsum, ng, want, &order);
            grp = nullptr;
          }
        }
      }
      const uint32_t off = tab_alloc_rows(A, sh, want);
      if (tid == 0) {
        if (t == 0) out->n_genotypes = ng;
        out->row_off[table] = off == GRIM_NONE ? 0 : off;
        out->n_rows[table] = off == GRIM_NONE ? 0 : want;
      }
      if (off != GRIM_NONE)
        for (uint32_t r = tid; r < want; r += GRIM_WG) {
          const uint32_t g = order[r];
          const PairRec pr = rec[grp ? grp[g].head : S.ghead[g]];
          grim_row row;
          row.a = pr.k1;
          row.b = pr.k2;
          row.prob = grp ? grp[g].sum : S.gsum[g];
          row.popa = ENT_POP(pr.e1);
          row.popb = ENT_POP(pr.e2);
          A.rows[off + r] = row;
        }
      __syncthreads();
    }
    HIST(3, nU, 1);
    HIST(6, x.ng[0], 1);
    HIST(7, nU, (STAMP_NOW() - t_item) / 100);
  }
}
