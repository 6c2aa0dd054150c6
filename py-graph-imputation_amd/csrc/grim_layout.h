// grim_layout.h -- plain-data layouts and host-side rules shared by the engine (grim_engine.hip), the host
// helpers (grim_host.cpp) and the streaming pipeline (grim_stream.cpp).  No device code, no HIP headers.
#pragma once
#include <stdint.h>

#include "../../include/grim_hip.h"

// what the library keeps in HBM per fast-path subject (every locus typed, no '/' list, one population):
// the whole input of the half-wave kernel (grim_small.h)
struct SmallRec {
  uint16_t tok[2 * GRIM_MAXL];  // position l: side-1 allele, side-2 allele
  uint8_t slot[GRIM_MAXL];
  uint8_t same;                 // positions whose two sides are textually identical
  uint16_t prior_idx;
  uint32_t si;                  // subject index in the batch
};                              // 32 bytes

// one accepted haplotype pair of a subject whose tables are built by the table kernels (grim_tables.h)
struct PairRec {
  uint64_t k1, k2;  // 60-bit haplotype keys
  double prob;
  uint32_t e1, e2;  // entities: haplotype id (24 bits) | population (8 bits)
};                  // 32 bytes

struct TabWork {    // "build the tables of subject si from records [off, off + n) of the pair pool"
  uint32_t si, n, off, mask;
};
#define GRIM_TAB_T1_MAX 256  // work items up to this many pairs go to the one-wave table kernel
struct TabAux {     // per bigger work item: where its buckets and cells live, what the bucket kernel found
  uint32_t boff[2];      // first slot in the bucket-start array, genotype / haplotype-pair table
  uint32_t nb[2];        // buckets (0: the table is not split)
  uint32_t ng[2];        // groups found so far
  uint32_t overflow[2];  // a bucket held more pairs than a wave's arena
  uint32_t cell_base;    // first slot of its population cells (bucket-start array and cell records)
  uint32_t pad[3];
};
struct GrpRec {     // a group the bucket kernel found: sum of its pairs' probabilities in pair order, its first pair
  double sum;
  uint32_t head, pad;
};
struct CellRec {    // a population cell: sum, first pair (GRIM_NONE: empty)
  double sum;
  uint32_t first, pad;
};
struct TabUnit {    // work unit of the bucket kernel: bucket (table 0 / 1) or population cell (table 2) `tb & 0x0FFFFFFF`
  uint32_t item, tb;
  uint32_t off;     // the item's first record in the pair pool
  uint32_t lo, n;   // the bucket's / cell's pairs: [lo, lo + n) of the item's bucket order
  uint32_t pad[3];
};                  // 32 bytes

#define GRIM_SMALL_ROWS_FIXED 1  // ONE row {genotype keys, sum of all pairs, populations 0 0} is the subject's .umug row and both its
                                 // population-pair rows (one population: one genotype, one cell, one sum -- three copies of it
                                 // were a third of the bytes that cross PCIe per subject); then up to 16 .pmug rows

// subject classes (which kernel opens the subject)
enum { GRIM_CLS_SMALL = 0, GRIM_CLS_MEDIUM = 1, GRIM_CLS_GENERAL = 2 };

struct ClassRule {
  bool small_ok;           // one population, options threshold > 1, half-wave kernel not disabled
  bool medium_ok;          // one-wave kernel not disabled
  double medium_max_cost;  // subjects whose grim_cost exceeds it skip the one-wave kernel (0: no limit): it would hand them on
  uint32_t graph_loci;
  uint64_t opt_threshold;
};

// fully typed + unambiguous + one population -> half-wave kernel; all sides opened by the cartesian branch and few
// candidates in total -> one-wave kernel; everything else -> general kernel
static inline double grim_cost(const grim_subject &sj);
static inline int grim_classify(const ClassRule &R, const grim_subject &sj) {
  bool sm = R.small_ok && sj.n_loci == GRIM_MAXL && R.graph_loci == GRIM_MAXL && sj.flags == 0;
  for (int l = 0; l < GRIM_MAXL && sm; ++l) sm = sj.cnt[l][0] == 1 && sj.cnt[l][1] == 1 && sj.wid[l][0] == 1 && sj.wid[l][1] == 1;
  if (sm) return GRIM_CLS_SMALL;
  bool md = R.medium_ok && sj.n_loci >= 1;
  double cand = (double)(1u << sj.n_loci), opts = 1.0;
  for (int l = 0; l < sj.n_loci; ++l) {
    cand *= (double)(sj.cnt[l][0] > sj.cnt[l][1] ? sj.cnt[l][0] : sj.cnt[l][1]);
    opts *= (double)(sj.wid[l][0] > sj.wid[l][1] ? sj.wid[l][0] : sj.wid[l][1]);
  }
  md = md && cand <= 2048.0 && opts < (double)R.opt_threshold;
  if (md && R.medium_max_cost > 0.0 && grim_cost(sj) > R.medium_max_cost) md = false;
  return md ? GRIM_CLS_MEDIUM : GRIM_CLS_GENERAL;
}

// scheduling weight of a general-kernel subject (heaviest first: the work queue's tail stays short)
static inline double grim_cost(const grim_subject &sj) {
  double c = 1.0;
  for (int l = 0; l < sj.n_loci; ++l) c *= (double)(sj.cnt[l][0] > sj.cnt[l][1] ? sj.cnt[l][0] : sj.cnt[l][1]);
  for (int l = sj.n_loci; l < GRIM_MAXL; ++l) c *= 8.0;  // untyped loci multiply the neighbour fan-out
  return c * (double)(1u << (sj.n_loci ? sj.n_loci - 1 : 0));
}

static inline void grim_small_rec(const grim_subject &sj, const uint16_t *tok /* the subject's tokens */, uint32_t si, SmallRec &r) {
  for (int l = 0; l < GRIM_MAXL; ++l) {
    r.tok[2 * l] = tok[2 * l];
    r.tok[2 * l + 1] = tok[2 * l + 1];
    r.slot[l] = sj.slot[l];
  }
  r.same = sj.pad[0];
  r.prior_idx = sj.prior_idx;
  r.si = si;
}
