// grim_pair.h -- haplotype-pair scoring, first-wins dedup, ordered accumulation and ranking for one
// subject by one workgroup.  Restates calc_haps_pairs / calc_haps_pairs_haplotype
// (impute.py:444-548, 550-658), the epsilon ladder of call_comp_phase_prob (impute.py:1658-1693) and
// the writers' ranking (impute.py:24-76) on integer-encoded top lists.
//
// Order fidelity: every accepted pair carries its sequence number f = position in the
// reference's (phase, h, k) loop nest; dedup keeps the smallest f per unordered
// {(hap,pop),(hap,pop)}; sums add probabilities in increasing f; ranking is a stable sort by
// probability, first-seen order breaking ties.
#pragma once
#include "grim_dev.h"

// LDS block shared by the kernels
// one version of one '/'-list of the subject: offset into the slot's token copy, distinct alleles, width
struct ListVer {
  uint32_t off;
  uint16_t cnt, wid;
};

// pairs a thread keeps in flight in the pair pass / in the grouping inserts (register budget: the Plan-B
// kernel must stay within 256 VGPRs for two workgroups per CU)
#ifndef GRIM_PAIR_NB
#define GRIM_PAIR_NB 2
#endif
#ifndef GRIM_GROUP_NB
#define GRIM_GROUP_NB 4
#endif

struct WgShared {
  grim_subject subj;
  uint32_t toff[GRIM_MAXL][2];
  // version 0 = as typed; 1 = only alleles the graph knows (reduce_phase_to_valid_allels,
  // impute.py:864-879); 2 = the 10 most frequent of those (reduce_phase_to_commons_alleles,
  // impute.py:881-912).  side_ver[s] = version phase side s currently uses (0 unless open_phases
  // found no candidate in any phase, impute.py:1620-1627).
  ListVer lv[GRIM_MAXL][2][3];
  uint8_t side_ver[GRIM_SIDES];
  uint32_t ntok;
  // Plan C's most-common-allele reduction (impute.py:881-912 with commons_number=1, planc=True):
  // bestc[s][l] = index inside side s's current list at position l of the allele that stays
  // (0xFFFF: none of the list is known to the graph, the list stays); active while `reduced` is set
  uint16_t bestc[GRIM_SIDES][GRIM_MAXL];
  uint8_t reduced;
  uint8_t nodup;       // no two scored pairs of this subject can be the same unordered entity pair (prepare_lists)
  uint8_t sm_ok;       // the side-mask dedup's entity masks / positions (LDS behind the phase tile) describe the current lists
  uint32_t comp_mask;  // slots - 1 of the composite-haplotype table as cleared for the current pass (plan B / C)
  // abits[l][c]: bit a set = allele id a is in the subject's list of position l, column c (version 0).
  // The intersection opening tests a graph node against a side with one bit per position.
  uint32_t abits[GRIM_MAXL][2][128];
  uint32_t Tn[GRIM_SIDES];
  uint32_t hitn[GRIM_SIDES];  // shared label scan: nodes found per side
  uint32_t hitreq[GRIM_SIDES];  // ... and the membership bits a side requires (bit 2l + c: position l, column c)
  uint8_t cand_any[GRIM_SIDES];
  uint8_t ph_pat[GRIM_MAXPH];
  int nph;
  uint32_t poff[GRIM_MAXPH + 1];
  uint32_t tmp[GRIM_NWAVE + 24];
  double dtmp[GRIM_NWAVE];
  uint32_t bc[8];
  unsigned long long wctr[GRIM_NWAVE][4];
  grim_subject_result out;
  // Work areas of the pair and table stages.  They live in the LDS the waves' top-K work areas (WaveTop) occupy
  // while the phase sides are built -- the two uses never overlap -- which keeps a workgroup at 51 KB of LDS
  // (three workgroups per CU instead of two).  Set once per kernel by wg_arena().
  uint32_t *hist;    // [16 * GRIM_WG] radix histograms, ranking scratch
  double *qprob;     // [1024] staging for the population-cell walk / small-pass keys
  uint16_t *qcell;   // [1024]
};

// the pair pass keeps its dedup table in LDS (over hist / qprob) when a pass accepted at most this many pairs
#define GRIM_PASS_LDS_SLOTS 2048u
#define GRIM_PASS_LDS_MAX 1400u
typedef __attribute__((address_space(3))) uint64_t lds_u64;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

struct WgArena {
  uint32_t hist[16 * GRIM_WG];
  double qprob[1024];
  uint16_t qcell[1024];
};
static_assert(sizeof(WgArena) <= sizeof(WaveTop) * GRIM_NWAVE, "the pair-stage work areas must fit the top-K work areas");

__device__ __forceinline__ void wg_arena(WgShared &sh, WaveTop *wt) {
  if (threadIdx.x == 0) {
    WgArena *a = (WgArena *)wt;
    sh.hist = a->hist;
    sh.qprob = a->qprob;
    sh.qcell = a->qcell;
  }
}

struct Slot {
  double *Tp, *Tm;
  uint32_t *Te;
  uint64_t *k0, *k1;
  uint32_t *tmin, *tgid;
  uint32_t *Useq, *Uslot;
  double *Uprob;
  uint64_t *ska, *skb;
  uint32_t *sva, *svb;
  double *gsum;
  uint32_t *ghead, *gstart, *gcnt;
  double *qsum;
  uint32_t *qfirst;
  uint32_t *bset;
  uint64_t *comp;
  uint64_t *proj_k;
  uint32_t *proj_p;
  uint16_t *rtok;
  double *save;
};

__device__ __forceinline__ Slot make_slot(const DevArgs &A, uint32_t slot_idx) {
  uint8_t *b = A.scratch + (uint64_t)slot_idx * A.lay.stride;
  Slot s;
  s.Tp = (double *)(b + A.lay.Tp);
  s.Tm = (double *)(b + A.lay.Tm);
  s.Te = (uint32_t *)(b + A.lay.Te);
  s.k0 = (uint64_t *)(b + A.lay.k0);
  s.k1 = (uint64_t *)(b + A.lay.k1);
  s.tmin = (uint32_t *)(b + A.lay.tmin);
  s.tgid = (uint32_t *)(b + A.lay.tgid);
  s.Useq = (uint32_t *)(b + A.lay.Useq);
  s.Uslot = (uint32_t *)(b + A.lay.Uslot);
  s.Uprob = (double *)(b + A.lay.Uprob);
  s.ska = (uint64_t *)(b + A.lay.ska);
  s.skb = (uint64_t *)(b + A.lay.skb);
  s.sva = (uint32_t *)(b + A.lay.sva);
  s.svb = (uint32_t *)(b + A.lay.svb);
  s.gsum = (double *)(b + A.lay.gsum);
  s.ghead = (uint32_t *)(b + A.lay.ghead);
  s.gstart = (uint32_t *)(b + A.lay.gstart);
  s.gcnt = (uint32_t *)(b + A.lay.gcnt);
  s.qsum = (double *)(b + A.lay.qsum);
  s.qfirst = (uint32_t *)(b + A.lay.qfirst);
  s.bset = (uint32_t *)(b + A.lay.bset);
  s.comp = (uint64_t *)(b + A.lay.comp);
  s.proj_k = (uint64_t *)(b + A.lay.proj_k);
  s.proj_p = (uint32_t *)(b + A.lay.proj_p);
  s.rtok = (uint16_t *)(b + A.lay.rtok);
  s.save = (double *)(b + A.lay.save);
  return s;
}

// An entity is (hap id : 24 bits | pop : 8 bits).  Hap ids below 2^23 are graph node ids, ids with
// bit 23 set index the slot's composite-haplotype table (plan B/C).
#define ENT_HAP(e) ((e) & 0xFFFFFFu)
#define ENT_POP(e) ((e) >> 24)
#define HAP_COMPOSITE 0x800000u

__device__ __forceinline__ uint64_t hap_key(const DevGraph &g, const Slot &S, uint32_t hap) {
  return (hap & HAP_COMPOSITE) ? (S.comp[hap & 0x7FFFFFu] & ~GRIM_VALID) : g.node_key[hap];
}

struct PairRef {
  double p1, p2, m2;
  uint32_t e1, e2;
};

// pair number f -> (phase, h, k) -> the two top-list entries.  Phase i uses top-list rows 2i, 2i+1.
__device__ __forceinline__ PairRef pair_ref(const WgShared &sh, const Slot &S, uint32_t f) {
  int i = 0;
  while (f >= sh.poff[i + 1]) ++i;
  uint32_t r = f - sh.poff[i];
  uint32_t n2 = sh.Tn[2 * i + 1];
  uint32_t h = r / n2, k = r - h * n2;
  PairRef pr;
  pr.p1 = S.Tp[(2 * i) * GRIM_TOPCAP + h];
  pr.e1 = S.Te[(2 * i) * GRIM_TOPCAP + h];
  pr.p2 = S.Tp[(2 * i + 1) * GRIM_TOPCAP + k];
  pr.m2 = S.Tm[(2 * i + 1) * GRIM_TOPCAP + k];
  pr.e2 = S.Te[(2 * i + 1) * GRIM_TOPCAP + k];
  return pr;
}

// the literal acceptance test of impute.py:457-491: x = eps / P1; loop over k BREAKS at the first
// P2 < x (so k is reachable iff min(P2[0..k]) >= x); prior > 0; prior*P2 >= x (2x when hap1 == hap2)
__device__ __forceinline__ bool pair_accept(double eps, const PairRef &pr, double w) {
  double x = eps / pr.p1;
  if (!(pr.m2 >= x)) return false;
  if (!(w > 0.0)) return false;
  double thr = (ENT_HAP(pr.e1) == ENT_HAP(pr.e2)) ? x * 2.0 : x;
  return w * pr.p2 >= thr;
}

// the same test with the quotient x = eps / P1 at hand: it depends on the pair's FIRST entry only, so a phase tile computes
// it once per first-list entry (128 IEEE divisions) instead of once per scored pair (up to 16 384)
__device__ __forceinline__ bool pair_accept_x(double x, const PairRef &pr, double w) {
  if (!(pr.m2 >= x)) return false;
  if (!(w > 0.0)) return false;
  double thr = (ENT_HAP(pr.e1) == ENT_HAP(pr.e2)) ? x * 2.0 : x;
  return w * pr.p2 >= thr;
}

__device__ __forceinline__ double pair_prob(const PairRef &pr, double w) {
  double prob = pr.p1 * pr.p2 * w;  // (P1*P2)*prior, impute.py:515-521
  if (ENT_HAP(pr.e1) != ENT_HAP(pr.e2)) prob = prob * 2.0;
  return prob;
}

// number of pairs of all phases; fills sh.poff.  All threads call.
__device__ inline uint32_t pair_offsets(WgShared &sh) {
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int i = 0; i < sh.nph; ++i) {
      sh.poff[i] = acc;
      acc += sh.Tn[2 * i] * sh.Tn[2 * i + 1];
    }
    for (int i = sh.nph; i <= GRIM_MAXPH; ++i) sh.poff[i] = acc;
  }
  __syncthreads();
  return sh.poff[GRIM_MAXPH];
}

// ---- the two top lists of ONE phase in LDS (over the radix-histogram area, free during the pair stage) -----------------
// The pair loops below go phase by phase: 2 x <=128 entries are staged once and the phase's <=16 384 pairs read them from
// LDS -- five gathers per pair from per-workgroup HBM scratch were the cost of the pair stage.
#ifndef GRIM_TILE_MIN
#define GRIM_TILE_MIN 8192u  // scored pairs from which the pair stage goes phase by phase through LDS tiles
#endif
struct PairTile {
  double p1[GRIM_TOPCAP], p2[GRIM_TOPCAP], m2[GRIM_TOPCAP];
  uint32_t e1[GRIM_TOPCAP], e2[GRIM_TOPCAP];
  uint16_t d1[GRIM_TOPCAP], d2[GRIM_TOPCAP];    // dense numbers of the entities (pair_dense_ids), when the pass asked for them
  uint64_t bm[GRIM_TOPCAP * GRIM_TOPCAP / 64];  // accepted pairs of the phase, one bit each
  uint64_t wbm[GRIM_TOPCAP * GRIM_TOPCAP / 64]; // ... and the winners among them (side-mask dedup)
  double xq[GRIM_TOPCAP];                        // eps / p1[h] of the pass in hand (tile_quotients)
  double lp[64];                                 // the prior matrix when it has at most 64 cells
};
static_assert(sizeof(PairTile) <= 16 * GRIM_WG * 4, "the phase tile lives in the histogram area");
#define GRIM_SM_LDS_OFF (16u * GRIM_WG * 4u)  // the LDS behind the tile, up to the end of the waves' top-K work areas: free in the pair stage

// all threads; the caller synchronises before reading the tile and again before the next tile_load
__device__ __forceinline__ void tile_load(PairTile &T, const WgShared &sh, const Slot &S, int i, const uint16_t *dense = nullptr) {
  const uint32_t n1 = sh.Tn[2 * i], n2 = sh.Tn[2 * i + 1];
  for (uint32_t t = threadIdx.x; t < n1; t += GRIM_WG) {
    T.p1[t] = S.Tp[(2 * i) * GRIM_TOPCAP + t];
    T.e1[t] = S.Te[(2 * i) * GRIM_TOPCAP + t];
    if (dense) T.d1[t] = dense[(2 * i) * GRIM_TOPCAP + t];
  }
  for (uint32_t t = threadIdx.x; t < n2; t += GRIM_WG) {
    T.p2[t] = S.Tp[(2 * i + 1) * GRIM_TOPCAP + t];
    T.m2[t] = S.Tm[(2 * i + 1) * GRIM_TOPCAP + t];
    T.e2[t] = S.Te[(2 * i + 1) * GRIM_TOPCAP + t];
    if (dense) T.d2[t] = dense[(2 * i + 1) * GRIM_TOPCAP + t];
  }
}

// Dense numbers 0..D-1 (D <= 4096) for the distinct entities of all top lists: a pair's dedup key then fits 24 bits, and
// key and smallest pair number share ONE 64-bit table word -- a dedup insert touches one cache line instead of two (the
// table of a heavy subject does not fit the L2: every line touched is a DRAM round trip).  All threads call; dense[side *
// GRIM_TOPCAP + idx] is valid afterwards.  Uses the group-sum arrays of the slot (free during the pair stage).
#define GRIM_DENSE_SLOTS 8192u
// the table lives in the slot's group-sum arrays (8 and 4 bytes per pair_cap): a tiled pass has np >= GRIM_TILE_MIN <= pair_cap
static_assert(GRIM_TILE_MIN >= GRIM_DENSE_SLOTS, "pair_dense_ids: the dense-id table would overrun gsum / ghead of a small slot");
__device__ inline void pair_dense_ids(const DevArgs &A, WgShared &sh, const Slot &S, uint16_t *dense) {
  const int tid = threadIdx.x;
  uint64_t *ek = (uint64_t *)S.gsum;
  uint32_t *ei = S.ghead;
  const uint32_t n_ent = 2u * (uint32_t)sh.nph * GRIM_TOPCAP;
  for (uint32_t s = tid; s < GRIM_DENSE_SLOTS; s += GRIM_WG) ek[s] = 0;
  if (tid == 0) sh.bc[6] = 0;
  __syncthreads();
  for (int pass = 0; pass < 2; ++pass) {
    for (uint32_t q = tid; q < n_ent; q += GRIM_WG) {
      const uint32_t side = q / GRIM_TOPCAP, idx = q % GRIM_TOPCAP;
      if (idx >= sh.Tn[side]) continue;
      const uint64_t key = (uint64_t)S.Te[q] | GRIM_VALID;
      uint32_t s = (uint32_t)mix64(key) & (GRIM_DENSE_SLOTS - 1);
      for (;;) {
        uint64_t c = ALOAD(&ek[s]);
        if (c == 0 && pass == 0) {
          const uint64_t old = atomicCAS((unsigned long long *)&ek[s], 0ull, (unsigned long long)key);
          if (old == 0) {  // first sight of this entity: it gets the next number
            ASTORE(&ei[s], atomicAdd(&sh.bc[6], 1u));
            c = key;
          } else {
            c = old;
          }
        }
        if (c == key) break;
        s = (s + 1) & (GRIM_DENSE_SLOTS - 1);
      }
      if (pass == 1) dense[q] = (uint16_t)ALOAD(&ei[s]);
    }
    __syncthreads();
  }
}
__device__ __forceinline__ PairRef tile_pair(const PairTile &T, uint32_t r, uint32_t n2, uint32_t magic) {
  const uint32_t h = n2 > 1 ? __umulhi(r, magic) : r, k = r - h * n2;  // exact for r < 2^14, n2 <= 128 (magic = 2^32 / n2 + 1)
  PairRef pr;
  pr.p1 = T.p1[h];
  pr.e1 = T.e1[h];
  pr.p2 = T.p2[k];
  pr.m2 = T.m2[k];
  pr.e2 = T.e2[k];
  return pr;
}
// eps / p1[h] for the rows of the tile in hand; all threads, between the barrier behind tile_load and the pairs
__device__ __forceinline__ void tile_quotients(PairTile &T, uint32_t n1, double eps) {
  for (uint32_t h = threadIdx.x; h < n1; h += GRIM_WG) T.xq[h] = eps / T.p1[h];
}
__device__ __forceinline__ uint32_t tile_row(uint32_t r, uint32_t n2, uint32_t magic) { return n2 > 1 ? __umulhi(r, magic) : r; }
__device__ __forceinline__ uint32_t tile_magic(uint32_t n2) { return n2 > 1 ? (uint32_t)(0x100000000ull / n2) + 1u : 0u; }

// first ladder index at which ANY pair is accepted (n_ladder if none).  Equals the reference's
// "decrease epsilon until the pass returns something" loop (impute.py:1665-1687).
__device__ inline int ladder_first(const DevArgs &A, WgShared &sh, const Slot &S, const double *prior, uint32_t np) {
  const int P = A.g.P;
  int best = A.prm.n_ladder;
  if (np < GRIM_TILE_MIN) {  // few pairs: straight from the slot
    for (uint32_t f = threadIdx.x; f < np && best > 0; f += GRIM_WG) {
      PairRef pr = pair_ref(sh, S, f);
      double w = prior[ENT_POP(pr.e1) * P + ENT_POP(pr.e2)];
      for (int idx = 0; idx < best; ++idx) {
        if (pair_accept(A.prm.ladder[idx], pr, w)) {
          best = idx;
          break;
        }
      }
    }
  } else {
    PairTile &T = *(PairTile *)sh.hist;
    // the quotients ladder[idx] / p1[h] of a phase, once per row and step, in the LDS behind the tile (the waves' top-K work
    // areas are free during the pair stage): a pair then costs compares only, whatever the number of steps it is tried at
    double *X = (double *)((uint8_t *)sh.hist + GRIM_SM_LDS_OFF);
    const int nl = A.prm.n_ladder;
    const bool rowq = (uint32_t)nl * GRIM_TOPCAP * 8u <= (uint32_t)(sizeof(WaveTop) * GRIM_NWAVE) - GRIM_SM_LDS_OFF;
    const bool lds_prior = P * P <= 64;  // (a global load per pair -- an L1 round trip inside the loop -- was most of the ladder's time)
    if (lds_prior)
      for (int c = threadIdx.x; c < P * P; c += GRIM_WG) T.lp[c] = prior[c];
    for (int i = 0; i < sh.nph; ++i) {
      const uint32_t n1 = sh.Tn[2 * i], n2 = sh.Tn[2 * i + 1], npi = n1 * n2;
      if (!npi) continue;
      __syncthreads();
      tile_load(T, sh, S, i);
      __syncthreads();
      if (rowq) {
        for (uint32_t t = threadIdx.x; t < (uint32_t)nl * n1; t += GRIM_WG) {
          const uint32_t idx = t / n1, h = t - idx * n1;
          X[idx * GRIM_TOPCAP + h] = A.prm.ladder[idx] / T.p1[h];
        }
        __syncthreads();
      }
      const uint32_t magic = tile_magic(n2);
      for (uint32_t r = threadIdx.x; r < npi && best > 0; r += GRIM_WG) {
        const PairRef q = tile_pair(T, r, n2, magic);
        const uint32_t cell = ENT_POP(q.e1) * P + ENT_POP(q.e2);
        const double w = lds_prior ? T.lp[cell] : prior[cell];
        const uint32_t h = tile_row(r, n2, magic);
        for (int idx = 0; idx < best; ++idx) {
          if (rowq ? pair_accept_x(X[idx * GRIM_TOPCAP + h], q, w) : pair_accept(A.prm.ladder[idx], q, w)) {
            best = idx;
            break;
          }
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) sh.bc[0] = (uint32_t)A.prm.n_ladder;
  __syncthreads();
  atomicMin(&sh.bc[0], (uint32_t)best);
  __syncthreads();
  int r = (int)sh.bc[0];
  __syncthreads();
  return r;
}

// ---- a tiled pass with the SIDE-MASK dedup (round 4; first in the mid-size kernel, grim_mid.h) ---------------------------
// First-wins dedup of calc_haps_pairs (impute.py:506-511, 603-611) without a table of pairs.  A pair is (entity x of side
// 2i at h, entity y of side 2i+1 at k); the same unordered {x, y} can only come up again where x and y sit in the two lists
// of another phase (either way round) or the other way round in this one.  Every entity (dense number d < D,
// pair_dense_ids) knows the sides it is in -- a 32-bit mask -- and its position in each, so the earlier occurrences of a
// pair are a few bit operations away, and whether one of them was ACCEPTED is a bit of that phase's accept bitmap: a pair
// loses iff an occurrence with a smaller pair number has its bit set.  Two sweeps over the phase tiles: the first leaves
// every phase's accept bits in the slot (32 KB, where the hash table was), the second decides and lists the winners in
// pair order.  What it replaces for a subject with ~40 000 accepted pairs (config 5): 16 bytes of list per accepted pair,
// a 1 MB table cleared per pass and one random 64-byte line per insert -- 15 MB of slot traffic per subject and a quarter
// of the kernel's time.  The masks and positions live in the LDS behind the phase tile (the waves' top-K work areas are
// free during the pair stage).  Returns GRIM_NONE when they do not fit, or a list names an entity twice: the caller takes
// the table path.  All threads call.
__device__ inline uint32_t pair_pass_sidemask(const DevArgs &A, WgShared &sh, const Slot &S, const double *prior, uint32_t np,
                                              double eps, bool emit, double *maxp, bool fresh) {
  const int P = A.g.P;
  const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t nsides = 2u * (uint32_t)sh.nph, n_ent = nsides * GRIM_TOPCAP;
  constexpr uint32_t GW = GRIM_TOPCAP * GRIM_TOPCAP / 64;  // bitmap words per phase
  if (A.tab_cap < GRIM_MAXPH * GW) return GRIM_NONE;
  uint16_t *dense = (uint16_t *)S.gstart;
  // the second pass on the same lists (final epsilon after the MaxProb pass) finds numbers, masks and positions in place
  const bool build = fresh || !sh.sm_ok;
  if (!fresh && !sh.sm_ok) return GRIM_NONE;  // (uniform) the first pass took the table path: so does this one
  if (build) pair_dense_ids(A, sh, S, dense);  // (ends with a barrier) sh.bc[6] = distinct entities
  if (build && threadIdx.x == 0) sh.bc[5] = sh.bc[6];
  __syncthreads();
  const uint32_t D = sh.bc[5];
  uint32_t ltot = 0;
  for (uint32_t sd = 0; sd < nsides; ++sd) ltot += sh.Tn[sd];
  const uint32_t lds_room = (uint32_t)(sizeof(WaveTop) * GRIM_NWAVE) - GRIM_SM_LDS_OFF;
  const uint32_t off_eoff = 4u * D, off_epos = off_eoff + ((2u * (D + 2u) + 3u) & ~3u);
  __syncthreads();
  if (off_epos + ltot > lds_room) {  // (uniform)
    if (tid == 0) sh.sm_ok = 0;
    __syncthreads();
    return GRIM_NONE;
  }
  uint8_t *lds = (uint8_t *)sh.hist + GRIM_SM_LDS_OFF;
  uint32_t *smask = (uint32_t *)lds;
  uint16_t *eoff = (uint16_t *)(lds + off_eoff);
  uint8_t *epos = lds + off_epos;
  if (build) {
  for (uint32_t d = tid; d < D; d += GRIM_WG) smask[d] = 0;
  if (tid == 0) {
    sh.bc[4] = 0;
    sh.sm_ok = 0;
  }
  __syncthreads();
  for (uint32_t q = tid; q < n_ent; q += GRIM_WG) {
    const uint32_t sd = q / GRIM_TOPCAP, idx = q % GRIM_TOPCAP;
    if (idx >= sh.Tn[sd]) continue;
    const uint32_t old = atomicOr(&smask[dense[q]], 1u << sd);
    if ((old >> sd) & 1u) sh.bc[4] = 1;  // an entity twice in one list: not this scheme's case
  }
  __syncthreads();
  if (sh.bc[4]) {
    __syncthreads();
    return GRIM_NONE;
  }
  {
    const uint32_t per = (D + GRIM_WG - 1) / GRIM_WG;
    const uint32_t d0 = tid * per < D ? tid * per : D, d1 = d0 + per < D ? d0 + per : D;
    uint32_t sum = 0;
    for (uint32_t d = d0; d < d1; ++d) sum += (uint32_t)__popc(smask[d]);
    uint32_t total;
    uint32_t at = wg_excl_scan(sum, sh.tmp, total);
    for (uint32_t d = d0; d < d1; ++d) {
      eoff[d] = (uint16_t)at;
      at += (uint32_t)__popc(smask[d]);
    }
  }
  __syncthreads();
  for (uint32_t q = tid; q < n_ent; q += GRIM_WG) {
    const uint32_t sd = q / GRIM_TOPCAP, idx = q % GRIM_TOPCAP;
    if (idx >= sh.Tn[sd]) continue;
    const uint32_t d = dense[q];
    epos[eoff[d] + (uint32_t)__popc(smask[d] & ((1u << sd) - 1u))] = (uint8_t)idx;
  }
  if (tid == 0) sh.sm_ok = 1;
  }  // build
  PairTile &T = *(PairTile *)sh.hist;
  uint64_t *G = (uint64_t *)S.k0;  // [GRIM_MAXPH][GW]: the accept bits of every phase
  const bool lds_prior = P * P <= 64;
  if (lds_prior)
    for (int c = tid; c < P * P; c += GRIM_WG) T.lp[c] = prior[c];
  // ---- sweep 1: accept bits ----------------------------------------------------------------------------------------------
  for (int i = 0; i < sh.nph; ++i) {
    const uint32_t n2 = sh.Tn[2 * i + 1], npi = sh.Tn[2 * i] * n2;
    if (!npi) continue;
    __syncthreads();  // the previous tile is spent
    tile_load(T, sh, S, i);
    __syncthreads();
    tile_quotients(T, sh.Tn[2 * i], eps);
    __syncthreads();
    const uint32_t magic = tile_magic(n2);
    const uint32_t q = ((npi + GRIM_NWAVE * 64 - 1) / (GRIM_NWAVE * 64)) * 64, r0 = wv * q, r1 = r0 + q < npi ? r0 + q : npi;
    for (uint32_t c0 = r0; c0 < r1; c0 += 64) {
      const uint32_t r = c0 + lane;
      bool on = false;
      if (r < r1) {
        const PairRef pr = tile_pair(T, r, n2, magic);
        const uint32_t cell = ENT_POP(pr.e1) * P + ENT_POP(pr.e2);
        on = pair_accept_x(T.xq[tile_row(r, n2, magic)], pr, lds_prior ? T.lp[cell] : prior[cell]);
      }
      const uint64_t m = __ballot(on);
      if (lane == 0) G[(uint32_t)i * GW + (c0 >> 6)] = m;
    }
  }
  __threadfence_block();
  __syncthreads();
  // ---- sweep 2: winners, in pair order -------------------------------------------------------------------------------------
  uint32_t nU = 0;
  double mx = 0.0;
  for (int i = 0; i < sh.nph; ++i) {
    const uint32_t n2 = sh.Tn[2 * i + 1], npi = sh.Tn[2 * i] * n2;
    if (!npi) continue;
    __syncthreads();
    tile_load(T, sh, S, i, dense);
    for (uint32_t w = tid; w < (npi + 63) / 64; w += GRIM_WG) T.bm[w] = G[(uint32_t)i * GW + w];
    __syncthreads();
    const uint32_t magic = tile_magic(n2);
    const uint32_t q = ((npi + GRIM_NWAVE * 64 - 1) / (GRIM_NWAVE * 64)) * 64, r0 = wv * q, r1 = r0 + q < npi ? r0 + q : npi;
    const uint32_t below = (4u << (2 * i)) - 1u;  // sides of the phases up to and including this one
    uint32_t cnt = 0;
    for (uint32_t c0 = r0; c0 < r1; c0 += 64) {
      const uint64_t m = T.bm[c0 >> 6];
      bool win = false;
      if ((m >> lane) & 1ull) {
        const uint32_t r = c0 + lane;
        const uint32_t h = n2 > 1 ? __umulhi(r, magic) : r, k = r - h * n2;
        const uint32_t dx = T.d1[h], dy = T.d2[k];
        const uint32_t mkx = smask[dx], mky = smask[dy], ox = eoff[dx], oy = eoff[dy];
        // bit 2j of `same`: x is in side 2j and y in side 2j+1; of `swp`: y in side 2j and x in side 2j+1
        uint32_t same = mkx & (mky >> 1) & 0x55555555u & below & ~(1u << (2 * i));  // (not this occurrence itself)
        uint32_t swp = mky & (mkx >> 1) & 0x55555555u & below;
        bool lost = false;
        while ((same | swp) && !lost) {
          const bool sw = same == 0;
          const uint32_t bits = sw ? swp : same;
          const int sd = __ffs((int)bits) - 1;  // the even side of the phase, 2j
          if (sw) swp &= swp - 1; else same &= same - 1;
          const int j = sd >> 1;
          const uint32_t m1 = sw ? mky : mkx, m2 = sw ? mkx : mky, o1 = sw ? oy : ox, o2 = sw ? ox : oy;
          const uint32_t h2 = epos[o1 + (uint32_t)__popc(m1 & ((1u << sd) - 1u))];
          const uint32_t k2 = epos[o2 + (uint32_t)__popc(m2 & ((2u << sd) - 1u))];
          const uint32_t b2 = h2 * sh.Tn[2 * j + 1] + k2;
          if (j == i) {
            if (b2 < r && ((T.bm[b2 >> 6] >> (b2 & 63u)) & 1ull)) lost = true;
          } else if ((G[(uint32_t)j * GW + (b2 >> 6)] >> (b2 & 63u)) & 1ull) {
            lost = true;  // (an earlier phase: every pair of it comes first)
          }
        }
        win = !lost;
      }
      const uint64_t mw = __ballot(win);
      if (lane == 0) T.wbm[c0 >> 6] = mw;
      cnt += (uint32_t)__popcll(mw);
    }
    if (lane == 0) sh.tmp[wv] = cnt;
    __syncthreads();
    uint32_t base = nU, total = 0;
    for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) {
      if (w2 < wv) base += sh.tmp[w2];
      total += sh.tmp[w2];
    }
    for (uint32_t c0 = r0; c0 < r1; c0 += 64) {
      const uint64_t mw = T.wbm[c0 >> 6];
      if ((mw >> lane) & 1ull) {
        const uint32_t r = c0 + lane;
        const PairRef pr = tile_pair(T, r, n2, magic);
        const uint32_t cell = ENT_POP(pr.e1) * P + ENT_POP(pr.e2);
        const double prob = pair_prob(pr, lds_prior ? T.lp[cell] : prior[cell]);
        mx = prob > mx ? prob : mx;
        if (emit) {
          const uint32_t pos = base + (uint32_t)__popcll(mw & lt);
          S.Useq[pos] = sh.poff[i] + r;
          S.Uprob[pos] = prob;
        }
      }
      base += (uint32_t)__popcll(mw);
    }
    nU += total;
  }
  __syncthreads();
  for (int d = 32; d > 0; d >>= 1) {
    const double o = __shfl_xor(mx, d);
    if (o > mx) mx = o;
  }
  if (lane == 0) sh.dtmp[wv] = mx;
  __syncthreads();
  mx = sh.dtmp[0];
  for (int w2 = 1; w2 < GRIM_NWAVE; ++w2)
    if (sh.dtmp[w2] > mx) mx = sh.dtmp[w2];
  __syncthreads();
  *maxp = mx;
  return nU;
}

// One full pass at `eps`: dedup (first f wins), MaxProb over winners, optionally the ordered list
// U of winners.  Returns the number of winners; *maxp gets MaxProb.  (impute.py:512-527)
// fresh_lists = false: the caller's previous pair_pass ran on these very top lists (the MaxProb pass before the final one)
__device__ inline uint32_t pair_pass(const DevArgs &A, WgShared &sh, const Slot &S, const double *prior, uint32_t np,
                                     double eps, bool emit, double *maxp, bool fresh_lists = true) {
  const int P = A.g.P;
  const int tid = threadIdx.x;
  // nodup (prepare_lists): every scored pair is a different unordered entity pair, so every accepted pair wins -- no
  // keys, no dedup table, no winner pass; the accepted list IS the winner list
  const bool nodup = sh.nodup && !(A.flags & GRIM_F_NO_NODUP);
  if (np <= GRIM_WG) {
    // few pairs: one thread per pair, first-wins dedup by looking at the earlier pairs' keys in LDS
    uint64_t *pk = (uint64_t *)sh.qprob;  // free outside pop_tables
    bool accd = false;
    uint64_t key = 0;
    double prob = 0.0;
    if ((uint32_t)tid < np) {
      PairRef pr = pair_ref(sh, S, tid);
      double w = prior[ENT_POP(pr.e1) * P + ENT_POP(pr.e2)];
      if (pair_accept(eps, pr, w)) {
        accd = true;
        uint32_t lo = pr.e1 < pr.e2 ? pr.e1 : pr.e2, hi = pr.e1 < pr.e2 ? pr.e2 : pr.e1;
        key = ((uint64_t)lo << 32) | hi | GRIM_VALID;
        prob = pair_prob(pr, w);
      }
    }
    pk[tid] = key;
    __syncthreads();
    bool win = accd;
    if (!nodup)
      for (int f2 = 0; f2 < tid && win; ++f2) win = pk[f2] != key;
    uint64_t m = __ballot(win);
    double mx = win ? prob : 0.0;
    for (int d = 32; d > 0; d >>= 1) {
      double o = __shfl_xor(mx, d);
      if (o > mx) mx = o;
    }
    if (lane_id() == 0) {
      sh.tmp[wave_id()] = (uint32_t)__popcll(m);
      sh.dtmp[wave_id()] = mx;
    }
    __syncthreads();
    uint32_t base = 0, tot = 0;
    mx = sh.dtmp[0];
    for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) {
      uint32_t t = sh.tmp[w2];
      if (w2 < wave_id()) base += t;
      tot += t;
      if (sh.dtmp[w2] > mx) mx = sh.dtmp[w2];
    }
    if (emit && win) {
      uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
      S.Useq[pos] = tid;
      S.Uprob[pos] = prob;
    }
    __syncthreads();
    *maxp = mx;
    return tot;
  }
  // Stage A: the accepted pairs, in pair order, with their dedup key and probability (4 x 256 pairs per barrier
  // round).  Everything after this works on the accepted list only: its length, not the number of scored pairs,
  // sizes the dedup table -- which lives in LDS when the list is short, so that the common subject's dedup
  // causes no HBM traffic at all (clearing a 2 x np-slot table per pass was 8.5 GB of writes per 100 k mixed subjects).
  uint32_t *Af = nodup ? S.Useq : S.sva, *Aslot = S.svb;  // nodup: stage A writes the winners' list itself
  uint64_t *Akey = S.ska;
  double *Aprob = nodup ? S.Uprob : (double *)S.skb;
  const bool keep_list = !nodup || emit;                   // a nodup pass that only wants MaxProb writes nothing
  double amx = 0.0;                                        // nodup: the biggest accepted probability this thread saw
#if defined(GRIM_STAMPS) && !defined(GRIM_NO_PP_STAMPS)
  unsigned long long _pp_t0 = wall_clock64();
#define PP_STAMP(k)                                                                          \
  do {                                                                                       \
    __syncthreads();                                                                         \
    if (threadIdx.x == 0 && emit) {                                                          \
      const unsigned long long _n = wall_clock64();                                           \
      atomicAdd(&A.counters[GRIM_STAMP_BASE + (k)], _n - _pp_t0);                            \
      _pp_t0 = _n;                                                                           \
    }                                                                                        \
  } while (0)
#else
#define PP_STAMP(k)
#endif
  uint32_t nA = 0;
  const bool tiled = np >= GRIM_TILE_MIN;
  if (tiled && !nodup && !(A.flags & GRIM_F_NO_SIDEMASK)) {
    const uint32_t r = pair_pass_sidemask(A, sh, S, prior, np, eps, emit, maxp, fresh_lists);
    if (r != GRIM_NONE) return r;  // (else: the table path below)
  }
  uint32_t *Akey32 = (uint32_t *)S.ska;       // tiled passes: 24-bit keys over dense entity numbers
  uint16_t *dense = (uint16_t *)S.gstart;
  if (tiled) {
    if (!nodup) pair_dense_ids(A, sh, S, dense);
    PairTile &T = *(PairTile *)sh.hist;
    const int lane = lane_id(), wv = wave_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const bool lds_prior = P * P <= 64;
    __syncthreads();
    if (lds_prior)
      for (int c = tid; c < P * P; c += GRIM_WG) T.lp[c] = prior[c];
    for (int i = 0; i < sh.nph; ++i) {
      const uint32_t n2 = sh.Tn[2 * i + 1], npi = sh.Tn[2 * i] * n2;
      if (!npi) continue;
      __syncthreads();  // the previous tile is spent
      tile_load(T, sh, S, i, nodup ? nullptr : dense);
      __syncthreads();
      tile_quotients(T, sh.Tn[2 * i], eps);
      __syncthreads();
      const uint32_t magic = tile_magic(n2);
      // wave w owns the w-th stretch of the phase's pairs (whole chunks of 64): pass 1 marks and counts the accepted ones,
      // pass 2 writes them behind the waves before it -- two barriers per phase instead of two per 1024 pairs
      const uint32_t q = ((npi + GRIM_NWAVE * 64 - 1) / (GRIM_NWAVE * 64)) * 64, r0 = wv * q, r1 = r0 + q < npi ? r0 + q : npi;
      uint32_t cnt = 0;
      for (uint32_t c0 = r0; c0 < r1; c0 += 64) {
        const uint32_t r = c0 + lane;
        bool on = false;
        if (r < r1) {
          const PairRef pr = tile_pair(T, r, n2, magic);
          const uint32_t cell = ENT_POP(pr.e1) * P + ENT_POP(pr.e2);
          on = pair_accept_x(T.xq[tile_row(r, n2, magic)], pr, lds_prior ? T.lp[cell] : prior[cell]);
        }
        const uint64_t m = __ballot(on);
        if (lane == 0) T.bm[c0 >> 6] = m;
        cnt += (uint32_t)__popcll(m);
      }
      if (lane == 0) sh.tmp[wv] = cnt;
      __syncthreads();
      uint32_t base = nA, total = 0;
      for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) {
        if (w2 < wv) base += sh.tmp[w2];
        total += sh.tmp[w2];
      }
      for (uint32_t c0 = r0; c0 < r1; c0 += 64) {
        const uint64_t m = T.bm[c0 >> 6];
        if ((m >> lane) & 1ull) {
          const uint32_t r = c0 + lane;
          const PairRef pr = tile_pair(T, r, n2, magic);
          const uint32_t cell = ENT_POP(pr.e1) * P + ENT_POP(pr.e2);
          const uint32_t pos = base + (uint32_t)__popcll(m & lt);
          const double prob = pair_prob(pr, lds_prior ? T.lp[cell] : prior[cell]);
          if (nodup) {
            amx = prob > amx ? prob : amx;
          } else {
            const uint32_t h = n2 > 1 ? __umulhi(r, magic) : r, k = r - h * n2;
            const uint32_t da = T.d1[h], db = T.d2[k];
            Akey32[pos] = ((da < db ? da : db) << 12) | (da < db ? db : da);
          }
          if (keep_list) {
            Af[pos] = sh.poff[i] + r;
            Aprob[pos] = prob;
          }
        }
        base += (uint32_t)__popcll(m);
      }
      nA += total;
    }
    __syncthreads();
  } else {
    // few pairs: 4 x 256 per barrier round straight from the slot (a tile per phase would cost more barriers than it saves)
    for (uint32_t f0 = 0; f0 < np; f0 += 4 * GRIM_WG) {
      bool on[4];
      uint64_t key[4];
      double prob[4];
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t f = f0 + q * GRIM_WG + tid;
        on[q] = false;
        key[q] = 0;
        prob[q] = 0.0;
        if (f < np) {
          const PairRef pr = pair_ref(sh, S, f);
          const double w = prior[ENT_POP(pr.e1) * P + ENT_POP(pr.e2)];
          if (pair_accept(eps, pr, w)) {
            on[q] = true;
            const uint32_t lo = pr.e1 < pr.e2 ? pr.e1 : pr.e2, hi = pr.e1 < pr.e2 ? pr.e2 : pr.e1;
            key[q] = ((uint64_t)lo << 32) | hi | GRIM_VALID;
            prob[q] = pair_prob(pr, w);
          }
        }
      }
      uint64_t m[4];
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        m[q] = __ballot(on[q]);
        if (lane_id() == 0) sh.tmp[q * GRIM_NWAVE + wave_id()] = (uint32_t)__popcll(m[q]);
      }
      __syncthreads();
      uint32_t run = nA, base[4];
  #pragma unroll
      for (int q = 0; q < 4; ++q)
        for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) {
          if (w2 == wave_id()) base[q] = run;
          run += sh.tmp[q * GRIM_NWAVE + w2];
        }
  #pragma unroll
      for (int q = 0; q < 4; ++q)
        if (on[q]) {
          const uint32_t pos = base[q] + (uint32_t)__popcll(m[q] & ((1ull << lane_id()) - 1ull));
          if (nodup) amx = prob[q] > amx ? prob[q] : amx;
          else Akey[pos] = key[q];
          if (keep_list) {
            Af[pos] = f0 + q * GRIM_WG + tid;
            Aprob[pos] = prob[q];
          }
        }
      nA = run;
      __syncthreads();
    }
  }
  PP_STAMP(0);
  if (nodup) {  // the accepted pairs are the winners, already in pair order where the caller wants them
    for (int d = 32; d > 0; d >>= 1) {
      const double o = __shfl_xor(amx, d);
      if (o > amx) amx = o;
    }
    if (lane_id() == 0) sh.dtmp[wave_id()] = amx;
    __syncthreads();
    double mxn = sh.dtmp[0];
    for (int w2 = 1; w2 < GRIM_NWAVE; ++w2)
      if (sh.dtmp[w2] > mxn) mxn = sh.dtmp[w2];
    __syncthreads();
    *maxp = mxn;
    return nA;
  }
  // Stage B: one slot per unordered entity pair; the smallest position in the accepted list wins it
  const bool in_lds = nA <= GRIM_PASS_LDS_MAX;
  const bool words = tiled && !in_lds;  // one 64-bit word per slot: key << 32 | smallest position (all ones: empty)
  lds_u64 *lk = (lds_u64 *)sh.hist;
  lds_u32 *lm = (lds_u32 *)sh.qprob;
  uint64_t *tw = S.k0;
  uint32_t cap = 64;
  if (in_lds) {
    cap = GRIM_PASS_LDS_SLOTS;
    for (uint32_t s = tid; s < cap; s += GRIM_WG) {
      lk[s] = 0;
      lm[s] = GRIM_NONE;
    }
  } else {
    while (cap < 2 * nA) cap <<= 1;
    if (cap > A.tab_cap) cap = A.tab_cap;
    if (words) {
      for (uint32_t s = tid; s < cap; s += GRIM_WG) tw[s] = ~0ull;
    } else {
      for (uint32_t s = tid; s < cap; s += GRIM_WG) {
        S.k0[s] = 0;
        S.tmin[s] = GRIM_NONE;
      }
    }
  }
  const uint32_t mask = cap - 1;
  __syncthreads();
  if (in_lds) {
    for (uint32_t u = tid; u < nA; u += GRIM_WG) {
      const uint64_t a = tiled ? ((uint64_t)Akey32[u] | GRIM_VALID) : Akey[u];
      uint32_t s = (uint32_t)mix64(a) & mask;
      for (;;) {
        uint64_t c = __hip_atomic_load(&lk[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (c == 0) {
          uint64_t expect = 0;
          c = __hip_atomic_compare_exchange_strong(&lk[s], &expect, a, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                  ? a : expect;
        }
        if (c == a) break;
        s = (s + 1) & mask;
      }
      __hip_atomic_fetch_min(&lm[s], u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      Aslot[u] = s;
    }
  } else if (words) {
    constexpr int NW = 4;  // inserts a thread keeps in flight
    for (uint32_t u0 = tid; u0 < nA; u0 += NW * GRIM_WG) {
      uint32_t key[NW], s[NW];
      uint64_t word[NW];
      bool done[NW];
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const uint32_t u = u0 + q * GRIM_WG;
        done[q] = u >= nA;
        key[q] = done[q] ? 0u : Akey32[u];
        word[q] = ((uint64_t)key[q] << 32) | u;
        s[q] = (key[q] * 0x9E3779B1u ^ (key[q] >> 11) * 0x85EBCA6Bu) & mask;
      }
      for (;;) {
        bool all = true;
#pragma unroll
        for (int q = 0; q < NW; ++q) all = all && done[q];
        if (all) break;
        uint64_t w[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) w[q] = done[q] ? 0ull : ALOAD(&tw[s[q]]);
#pragma unroll
        for (int q = 0; q < NW; ++q)
          if (!done[q] && w[q] == ~0ull) {
            const uint64_t old = atomicCAS((unsigned long long *)&tw[s[q]], ~0ull, (unsigned long long)word[q]);
            w[q] = old == ~0ull ? word[q] : old;  // claimed: my position is in the word already
          }
#pragma unroll
        for (int q = 0; q < NW; ++q)
          if (!done[q]) {
            if ((uint32_t)(w[q] >> 32) == key[q]) {
              if (w[q] != word[q]) atomicMin((unsigned long long *)&tw[s[q]], (unsigned long long)word[q]);
              Aslot[u0 + q * GRIM_WG] = s[q];
              done[q] = true;
            } else {
              s[q] = (s[q] + 1) & mask;
            }
          }
      }
    }
  } else {
    for (uint32_t u0 = tid; u0 < nA; u0 += GRIM_PAIR_NB * GRIM_WG) {
      uint64_t key[GRIM_PAIR_NB], none[GRIM_PAIR_NB];
      bool on[GRIM_PAIR_NB];
      uint32_t slot[GRIM_PAIR_NB];
#pragma unroll
      for (int q = 0; q < GRIM_PAIR_NB; ++q) {
        const uint32_t u = u0 + q * GRIM_WG;
        on[q] = u < nA;
        none[q] = 0;
        key[q] = on[q] ? Akey[u] : 0;
      }
      tab_insert_n<false, GRIM_PAIR_NB>(S.k0, S.k1, mask, key, none, on, slot);
#pragma unroll
      for (int q = 0; q < GRIM_PAIR_NB; ++q) {
        const uint32_t u = u0 + q * GRIM_WG;
        if (on[q]) {
          atomicMin(&S.tmin[slot[q]], u);
          Aslot[u] = slot[q];
        }
      }
    }
  }
  __syncthreads();
  PP_STAMP(1);
  // Stage C: winners in pair order.  Wave w owns the w-th stretch of the accepted list: it counts its winners, and
  // (final pass) writes them behind the waves before it -- one barrier in between instead of two per 1024 pairs.
  uint32_t nU = 0;
  double mx = 0.0;
  {
    const int lane = lane_id(), wv = wave_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t q = ((nA + GRIM_NWAVE * 64 - 1) / (GRIM_NWAVE * 64)) * 64, u0 = wv * q, u1 = u0 + q < nA ? u0 + q : nA;
    auto is_winner = [&](uint32_t u) -> bool {
      const uint32_t slot = Aslot[u];
      if (words) return (uint32_t)ALOAD(&tw[slot]) == u;
      return (in_lds ? (uint32_t)lm[slot] : ALOAD(&S.tmin[slot])) == u;
    };
    // the winners' ballot masks stay in LDS for the writing pass when they fit (the dedup table occupies the histogram
    // area exactly when the list is short enough for the other one)
    uint64_t *wb = in_lds ? (uint64_t *)sh.qcell : (uint64_t *)sh.hist;
    const bool keep = in_lds || nA <= 64u * (16u * GRIM_WG * 4u / 8u);
    uint32_t cnt = 0;
    for (uint32_t c0 = u0; c0 < u1; c0 += 4 * 64) {  // four chunks in flight
      bool w[4];
      double pp[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t u = c0 + k * 64 + lane;
        w[k] = u < u1 && is_winner(u);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) pp[k] = w[k] ? Aprob[c0 + k * 64 + lane] : 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (pp[k] > mx) mx = pp[k];
        const uint64_t m = __ballot(w[k]);
        if (keep && lane == 0 && c0 + k * 64 < u1) wb[(c0 >> 6) + k] = m;
        cnt += (uint32_t)__popcll(m);
      }
    }
    if (lane == 0) sh.tmp[wv] = cnt;
    __syncthreads();
    uint32_t base = 0;
    for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) {
      if (w2 < wv) base += sh.tmp[w2];
      nU += sh.tmp[w2];
    }
    if (emit) {
      for (uint32_t c0 = u0; c0 < u1; c0 += 64) {
        const uint32_t u = c0 + lane;
        const uint64_t m = keep ? wb[c0 >> 6] : __ballot(u < u1 && is_winner(u));
        if ((m >> lane) & 1ull) {
          const uint32_t pos = base + (uint32_t)__popcll(m & lt);
          S.Useq[pos] = Af[u];
          S.Uprob[pos] = Aprob[u];
        }
        base += (uint32_t)__popcll(m);
      }
    }
  }
  PP_STAMP(2);
  // MaxProb
  for (int d = 32; d > 0; d >>= 1) {
    double o = __shfl_xor(mx, d);
    if (o > mx) mx = o;
  }
  if (lane_id() == 0) sh.dtmp[wave_id()] = mx;
  __syncthreads();
  mx = sh.dtmp[0];
  for (int w2 = 1; w2 < GRIM_NWAVE; ++w2)
    if (sh.dtmp[w2] > mx) mx = sh.dtmp[w2];
  __syncthreads();
  *maxp = mx;
  return nU;
}

// ---- <= 64 accepted pairs: the four tables by ONE wave with shuffles (no hash table, no barriers) ----
// Every lane holds one pair; lanes with equal group key form a group, the lowest lane is its
// first-seen member, every member adds the group's probabilities in lane (= sequence) order.
__device__ __forceinline__ void wave_group(uint64_t klo, uint64_t khi, double prob, bool act, int n, int &head, double &sum) {
  const int lane = lane_id();
  head = lane;
  sum = 0.0;
  bool first = true;
  for (int j = 0; j < n; ++j) {
    uint64_t a = lane_get(klo, j), b = lane_get(khi, j);
    double pj = lane_get(prob, j);
    if (act && a == klo && b == khi) {
      if (j < head) head = j;
      sum = first ? pj : sum + pj;
      first = false;
    }
  }
}

__device__ __forceinline__ uint32_t wave_rank(double sum, bool is_head, int n) {
  const int lane = lane_id();
  const uint64_t hm = __ballot(is_head);
  uint32_t rank = 0;
  for (int j = 0; j < n; ++j) {
    double sj = lane_get(sum, j);
    if (((hm >> j) & 1ull) && (sj > sum || (sj == sum && j < lane))) ++rank;
  }
  return rank;
}

// A wave's private piece of the row pool.  The pool head is ONE address: an atomic on it costs ~12 ns whoever
// issues it, which capped the one-wave kernel (four tables per subject) at 20 M subjects/s.  A wave that keeps a
// RowBlock across subjects takes GRIM_ROW_GRAB rows at a time and hands them out locally; the gaps it leaves are
// never referenced by a result header.  {0, 0}: every request goes to the pool (exact size, no gaps).
struct RowBlock {
  uint32_t off, left, grab;
};
#define GRIM_ROW_GRAB 64

__device__ __forceinline__ uint32_t wave_alloc_rows(const DevArgs &A, RowBlock &rb, uint32_t n) {
  if (n == 0) return 0;
  if (n <= rb.left) {  // wave-uniform state
    const uint32_t off = rb.off;
    rb.off += n;
    rb.left -= n;
    return off;
  }
  const uint32_t take = n > rb.grab ? n : rb.grab;
  uint32_t off = 0;
  if (lane_id() == 0) {
    off = atomicAdd(A.row_head, take);
    if (off + take > A.row_cap) {
      atomicExch(&A.counters[4], 1ull);
      off = GRIM_NONE;
    }
  }
  off = __shfl(off, 0);
  if (off == GRIM_NONE) {
    rb.left = 0;
    return GRIM_NONE;
  }
  rb.off = off + n;
  rb.left = take - n;
  return off;
}

// e1/e2/prob/k1/k2: this lane's pair (entities, probability, the two 60-bit haplotype keys)
__device__ inline void emit_small_core(const DevArgs &A, uint32_t nU, uint32_t e1, uint32_t e2, double prob, uint64_t k1,
                                       uint64_t k2, grim_subject_result &out, RowBlock &rb, uint32_t mask = 3) {
  const int lane = lane_id();
  const int n = (int)nU;
  const bool act = lane < n;
  const int P = A.g.P;
  const uint32_t pa = ENT_POP(e1), pb = ENT_POP(e2), h1 = ENT_HAP(e1), h2 = ENT_HAP(e2);
  int head;
  double sum;
  // ---- population pairs (both pops files share the sums) ----------------------------------------
  {
    const uint64_t qk = ((uint64_t)(pa < pb ? pa : pb) * (uint64_t)P + (pa < pb ? pb : pa)) + 1;
    wave_group(qk, 0, prob, act, n, head, sum);
    const bool is_head = act && head == lane;
    const uint32_t nq = (uint32_t)__popcll(__ballot(is_head));
    const uint32_t rank = wave_rank(sum, is_head, n);
    for (int t = 0; t < 2; ++t) {
      if (!((mask >> t) & 1u)) continue;
      const int table = t == 0 ? GRIM_T_UMUG_POPS : GRIM_T_PMUG_POPS;
      uint32_t want = nq < A.prm.n_pop_results ? nq : A.prm.n_pop_results;
      if (t == 1 && A.prm.em_mr) want = nq < 1 ? nq : 1;
      if (!(t == 0 ? A.prm.out_muug : A.prm.out_haps)) want = 0;
      const uint32_t off = wave_alloc_rows(A, rb, want);
      if (lane == 0) {
        out.row_off[table] = off == GRIM_NONE ? 0 : off;
        out.n_rows[table] = off == GRIM_NONE ? 0 : want;
      }
      if (off != GRIM_NONE && is_head && rank < want) {
        uint32_t a = pa, b = pb;
        if (t == 0 && A.prm.pop_rank[a] > A.prm.pop_rank[b]) {
          uint32_t x = a;
          a = b;
          b = x;
        }
        grim_row r;
        r.a = a; r.b = b; r.prob = sum; r.popa = a; r.popb = b;
        A.rows[off + rank] = r;
      }
    }
  }
  // ---- genotypes -----------------------------------------------------------------------------------
  if (mask & 1u) {
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int l = 0; l < GRIM_MAXL; ++l) {
      uint64_t x = (k1 >> (GRIM_ABITS * l)) & 0xFFF, y = (k2 >> (GRIM_ABITS * l)) & 0xFFF;
      lo |= (x < y ? x : y) << (GRIM_ABITS * l);
      hi |= (x < y ? y : x) << (GRIM_ABITS * l);
    }
    wave_group(lo, hi, prob, act, n, head, sum);
    const bool is_head = act && head == lane;
    const uint32_t ng = (uint32_t)__popcll(__ballot(is_head));
    const uint32_t rank = wave_rank(sum, is_head, n);
    const uint32_t want = A.prm.out_muug ? (ng < A.prm.n_results ? ng : A.prm.n_results) : 0;
    const uint32_t off = wave_alloc_rows(A, rb, want);
    if (lane == 0) {
      out.n_genotypes = ng;
      out.row_off[GRIM_T_UMUG] = off == GRIM_NONE ? 0 : off;
      out.n_rows[GRIM_T_UMUG] = off == GRIM_NONE ? 0 : want;
    }
    if (off != GRIM_NONE && is_head && rank < want) {
      grim_row r;
      r.a = k1; r.b = k2; r.prob = sum; r.popa = pa; r.popb = pb;
      A.rows[off + rank] = r;
    }
  }
  // ---- haplotype pairs ------------------------------------------------------------------------------
  if (mask & 2u) {
    uint32_t want = 0, off = 0, rank = 0;
    bool is_head = false;
    if (A.prm.out_haps) {
      const uint64_t hk = A.prm.em_mr ? (uint64_t)(lane + 1) : ((((uint64_t)(h1 < h2 ? h1 : h2)) << 32) | (h1 < h2 ? h2 : h1)) + 1;
      wave_group(hk, 0, prob, act, n, head, sum);
      is_head = act && head == lane;
      const uint32_t ng = (uint32_t)__popcll(__ballot(is_head));
      rank = wave_rank(sum, is_head, n);
      want = ng < A.prm.n_results ? ng : A.prm.n_results;
    }
    off = wave_alloc_rows(A, rb, want);
    if (lane == 0) {
      out.row_off[GRIM_T_PMUG] = off == GRIM_NONE ? 0 : off;
      out.n_rows[GRIM_T_PMUG] = off == GRIM_NONE ? 0 : want;
    }
    if (off != GRIM_NONE && is_head && rank < want) {
      grim_row r;
      r.a = k1; r.b = k2; r.prob = sum; r.popa = pa; r.popb = pb;
      A.rows[off + rank] = r;
    }
  }
}

__device__ inline void emit_small(const DevArgs &A, WgShared &sh, const Slot &S, uint32_t nU, grim_subject_result &out,
                                  uint32_t mask) {
  const int lane = lane_id();
  uint32_t e1 = 0, e2 = 0;
  double prob = 0.0;
  uint64_t k1 = 0, k2 = 0;
  if (lane < (int)nU) {
    PairRef pr = pair_ref(sh, S, S.Useq[lane]);
    e1 = pr.e1;
    e2 = pr.e2;
    prob = S.Uprob[lane];
    k1 = hap_key(A.g, S, ENT_HAP(e1));
    k2 = hap_key(A.g, S, ENT_HAP(e2));
  }
  RowBlock rb = {0, 0, 0};
  emit_small_core(A, nU, e1, e2, prob, k1, k2, out, rb, mask);
}

// ---- more than 64 accepted pairs: the tables are another kernel's work (grim_tables.h) --------------------------------
// The accepted pairs (in the reference's pair order, as the final pass left them in Useq / Uprob) become 32-byte records
// in the batch's pair pool and a work item tells the table kernels where they are and which halves of the tables to
// build.  All threads call.
__device__ inline void queue_tables(const DevArgs &A, WgShared &sh, const Slot &S, uint32_t nU, uint32_t si, uint32_t mask) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    uint32_t off = atomicAdd(A.queue + 8, nU);
    if (off + nU > A.ppool_cap) {
      atomicExch(&A.counters[4], 1ull);  // the run reports the overflow; the caller splits the batch and runs it again
      off = GRIM_NONE;
    }
    sh.bc[1] = off;
  }
  __syncthreads();
  const uint32_t off = sh.bc[1];
  __syncthreads();
  if (off == GRIM_NONE) return;
  for (uint32_t u = tid; u < nU; u += GRIM_WG) {
    const PairRef pr = pair_ref(sh, S, S.Useq[u]);
    PairRec r;
    r.k1 = hap_key(A.g, S, ENT_HAP(pr.e1));
    r.k2 = hap_key(A.g, S, ENT_HAP(pr.e2));
    r.prob = S.Uprob[u];
    r.e1 = pr.e1;
    r.e2 = pr.e2;
    A.ppool[off + u] = r;
  }
  if (tid == 0) {
    TabWork w;
    w.si = si;
    w.n = nU;
    w.off = off;
    w.mask = mask;
    if (nU <= GRIM_TAB_T1_MAX)
      A.t1_list[atomicAdd(A.queue + 9, 1u)] = w;
    else
      A.t2_list[atomicAdd(A.queue + 10, 1u)] = w;
  }
  __syncthreads();
}

// Everything after the final pass: the four output tables of one subject.
// mask: bit 0 = the MUUG half (.umug, .umug.pops), bit 1 = the phased half (.pmug, .pmug.pops).  The two
// halves come from different passes when the MUUG pass ended in Plan C (impute.py:1637-1654).
__device__ inline void emit_tables(const DevArgs &A, WgShared &sh, const Slot &S, uint32_t nU, grim_subject_result &out,
                                   uint32_t si, uint32_t mask = 3) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    sh.bc[2] = 0;
    if (mask & 2u) out.n_pairs = nU;
  }
  __syncthreads();
  if (nU <= 64) {
    if (wave_id() == 0) emit_small(A, sh, S, nU, out, mask);
    __syncthreads();
    return;
  }
  queue_tables(A, sh, S, nU, si, mask);
}
