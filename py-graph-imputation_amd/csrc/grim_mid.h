// grim_mid.h -- Plan A for the mid-size subject by ONE WORKGROUP, entirely in LDS: no per-workgroup HBM scratch, no
// 32 x 128 fixed-stride top lists, no wave-per-side streaming.  Same semantics as grim_plan_a_kernel (grim_plan_a.h /
// grim_pair.h cite the reference lines: open_phases impute.py:914-989, adjs_query networkx_graph.py:253-278,
// convert_list_to_one_dim impute.py:424-442, calc_haps_pairs* impute.py:444-658, the ladder impute.py:1658-1693).
//
// What the registry-shaped workload (BASELINE configs[3]) looks like per subject -- measured on its first 1 500 subjects with
// the oracle: 19 phase sides, 171 cartesian candidates over ALL sides (p99 1 344), 23 of them graph nodes (p99 217), 68
// (haplotype, population) entries (p99 643), 59 entries in the top lists (max 668) -- but 757 scored pairs (p99 15 600) and
// 302 accepted ones (p99 7 000).  The general kernel spends 60 us per subject on the sides whatever their size (32 sides x a
// dependent chain probe -> row start -> neighbour -> frequency -> bitonic top-K, one wave per side) and writes 94 KB of top
// lists and pair lists per subject into its HBM slot.  Here:
//
//   1. the candidates of all sides are flattened over the 256 threads (the one-wave kernel's trick, grim_medium.h): every
//      probe of the subject is in flight at once; hits are compacted in candidate order (ballots);
//   2. the hits' CSR rows, then the rows' haplotypes and their frequency vectors, are gathered flattened too: three memory
//      round trips per SUBJECT instead of per side;
//   3. a side's top-K order comes from counting ranks inside the side's segment (stable: bigger key first, then stream
//      order), the lists -- 12 bytes per entry plus the prefix minimum -- stay in LDS;
//   4. the pair stage reads the lists from LDS; a pass leaves ONE BIT per scored pair (accepted / winner), in pair order;
//      the first-wins dedup table (subjects whose '/' lists overlap) is in LDS;
//   5. the winners go to the tables: <= 64 by one wave's shuffles (emit_small_core), more as records of the pair pool, written
//      in pair order from the bitmap (prefix popcounts), with a work item for the table kernels.
//
// A subject that exceeds a limit below -- or needs a branch this kernel does not have (label-scan sides, reduced lists, a
// loci_map whose order is not the subject's) -- is handed to the general kernel through A.mid_list.
#pragma once
#include "grim_pair.h"

#define MD_C 8192u                   // candidates of all sides together
#define MD_H 768u                    // look-up hits
#define MD_T 2048u                   // haplotypes the hits expand to
#define MD_E 1024u                   // (haplotype, population) entries with p > 0
#define MD_SEG 1024u                 // ... of one side
#define MD_L 704u                    // entries kept in the top lists of all sides
#define MD_NP 32768u                 // scored pairs
#define MD_NPB (MD_NP + GRIM_MAXPH * 64u)  // bits of the pair bitmap (every phase starts a new 64-bit word)
#define MD_ESLOTS 2048u               // slots of the entity index of the dedup (>= 2 * MD_L)
// Four workgroups per CU (39.8 KB of LDS each, 114 VGPRs): the kernel's stages are latency, so its rate is its resident waves.
// Measured on config 4 (gpurun_out/r4m13, r4m14), capacities cut to fit: 3 per CU (1 024 hits / 1 280 entries / 1 024 list
// entries; 298 subjects handed on) 2.18 ms + 0.53 ms of general kernel; 4 per CU (768 / 1 024 / 704; 445 handed on) 1.70 +
// 0.56; 5 per CU (512 / 768 / 512, 96 VGPRs; 6 099 handed on) 1.27 + 1.35.
#ifndef GRIM_MID_WG_PER_CU
#define GRIM_MID_WG_PER_CU 4
#endif

// the arena: [hits | raw entries] while the sides are built, [sort keys | raw entries] while they are ranked,
// [pair bitmap | dedup table] in the pair stage
#define MD_OFF_HNODE 0u
#define MD_OFF_HBASE (MD_OFF_HNODE + 4u * MD_H)
#define MD_OFF_HSTART (MD_OFF_HBASE + 4u * MD_H)
#define MD_OFF_HSIDE (MD_OFF_HSTART + 4u * (MD_H + 4u))
#define MD_OFF_ENT ((MD_OFF_HSIDE + MD_H + 255u) & ~255u)
#define MD_OFF_ENTP MD_OFF_ENT
#define MD_OFF_ENTE (MD_OFF_ENTP + 8u * MD_E)
#define MD_OFF_ENTS (MD_OFF_ENTE + 4u * MD_E)
#define MD_ARENA_A (MD_OFF_ENTS + MD_E)
#define MD_OFF_BM 0u
#define MD_BM_BYTES ((MD_NPB / 8u + 255u) & ~255u)
#define MD_OFF_SMASK (MD_OFF_BM + MD_BM_BYTES)          // entity index: keys while it is built, then the entities' side masks
#define MD_OFF_EOFF (MD_OFF_SMASK + 4u * MD_ESLOTS)     // first occurrence of entity slot r in the position list
#define MD_OFF_ESLOT (MD_OFF_EOFF + ((2u * (MD_ESLOTS + 2u) + 255u) & ~255u))  // slot of every list entry
#define MD_OFF_EPOS (MD_OFF_ESLOT + 2u * MD_L)          // positions, by entity and side
#define MD_OFF_BM2 (MD_OFF_EPOS + MD_L)
#define MD_ARENA_B (MD_OFF_BM2 + MD_BM_BYTES)
#define MD_ARENA ((MD_ARENA_A > MD_ARENA_B ? MD_ARENA_A : MD_ARENA_B) + 15u & ~15u)
static_assert(8u * MD_E <= MD_OFF_ENT, "the sort keys overlay the hit arrays");
static_assert(MD_ESLOTS >= 2u * MD_L && MD_ESLOTS <= 65535u && (MD_ESLOTS & (MD_ESLOTS - 1u)) == 0 && GRIM_TOPCAP <= 256, "entity index: load <= 1/2, 16-bit slots, 8-bit positions");
static_assert(4u * (MD_NPB / 64u + 1u) + 64u * 16u <= 8u * MD_L, "word prefixes and the small-emit staging overlay the prefix minima");

#ifdef GRIM_STAMPS  // diagnostic build: workgroup time per stage, hand-overs by reason
#define MSTAMP(k)                                                          \
  do {                                                                     \
    __syncthreads();                                                       \
    if (threadIdx.x == 0) {                                                \
      const unsigned long long _t1 = wall_clock64();                       \
      atomicAdd(&A.counters[GRIM_MID_BASE + (k)], _t1 - _mt0);             \
      _mt0 = _t1;                                                          \
    }                                                                      \
  } while (0)
#define MBAIL(r)                                                           \
  do {                                                                     \
    if (threadIdx.x == 0) atomicAdd(&A.counters[GRIM_MID_BASE + 16 + (r)], 1ull); \
  } while (0)
#else
#define MSTAMP(k)
#define MBAIL(r)
#endif

struct MidShared {
  grim_subject subj;
  uint32_t toff[GRIM_MAXL][2];
  uint8_t ph_pat[GRIM_MAXPH];
  int nph;
  uint32_t cand_start[GRIM_SIDES + 1];
  uint32_t cnt_side[GRIM_SIDES];
  uint32_t seg_in[GRIM_SIDES + 1];  // raw entries of side s: [seg_in[s], seg_in[s + 1])
  uint32_t seg[GRIM_SIDES + 1];     // its top list: [seg[s], seg[s] + tlen[s])
  uint32_t tlen[GRIM_SIDES];
  uint32_t poff[GRIM_MAXPH + 1];    // pair numbers: phase i owns [poff[i], poff[i + 1])
  uint32_t boff[GRIM_MAXPH + 1];    // ... and bits [boff[i], boff[i] + pairs) of the bitmap (multiples of 64)
  double diag[GRIM_MAXPOP];         // prior[j][j]
  double lp[64];                    // the prior matrix when it has at most 64 cells
  uint32_t tmp[4 * GRIM_NWAVE + 8];
  double dtmp[GRIM_NWAVE];
  uint32_t bc[8];
  unsigned long long wctr[4];
  grim_subject_result out;
  double T_p[MD_L];
  double T_m[MD_L];  // prefix minimum of T_p inside the list; after the final pass: word prefixes + small-emit staging
  uint32_t T_e[MD_L];
  __attribute__((aligned(16))) uint8_t arena[MD_ARENA];
};

struct MidView {  // typed pointers into the arena
  uint32_t *hnode, *hbase, *hstart;
  uint8_t *hside;
  double *entp, *key;
  uint32_t *ente;
  uint8_t *ents;
  uint64_t *bm;
  uint64_t *bm2;
  lds_u32 *smask;
  uint16_t *eoff, *eslot;
  uint8_t *epos;
  uint32_t *wpre;
  uint32_t *se1, *se2;
  double *sprob;
};

__device__ __forceinline__ MidView mid_view(MidShared &M) {
  MidView v;
  v.hnode = (uint32_t *)(M.arena + MD_OFF_HNODE);
  v.hbase = (uint32_t *)(M.arena + MD_OFF_HBASE);
  v.hstart = (uint32_t *)(M.arena + MD_OFF_HSTART);
  v.hside = M.arena + MD_OFF_HSIDE;
  v.entp = (double *)(M.arena + MD_OFF_ENTP);
  v.key = (double *)(M.arena + 0);
  v.ente = (uint32_t *)(M.arena + MD_OFF_ENTE);
  v.ents = M.arena + MD_OFF_ENTS;
  v.bm = (uint64_t *)(M.arena + MD_OFF_BM);
  v.bm2 = (uint64_t *)(M.arena + MD_OFF_BM2);
  v.smask = (lds_u32 *)(M.arena + MD_OFF_SMASK);
  v.eoff = (uint16_t *)(M.arena + MD_OFF_EOFF);
  v.eslot = (uint16_t *)(M.arena + MD_OFF_ESLOT);
  v.epos = M.arena + MD_OFF_EPOS;
  v.wpre = (uint32_t *)M.T_m;
  v.sprob = (double *)((uint8_t *)M.T_m + ((4u * (MD_NPB / 64u + 1u) + 7u) & ~7u));
  v.se1 = (uint32_t *)(v.sprob + 64);
  v.se2 = v.se1 + 64;
  return v;
}

// pair r of phase i (r < n1 * n2 <= 2^14); *x: the pass's epsilon / P1 of the pair's first entry (mid_set_eps)
__device__ __forceinline__ PairRef mid_pair(const MidShared &M, int i, uint32_t r, uint32_t n2, uint32_t magic, double *x = nullptr) {
  const uint32_t h = n2 > 1 ? __umulhi(r, magic) : r, k = r - h * n2;
  const uint32_t a = M.seg[2 * i] + h, b = M.seg[2 * i + 1] + k;
  PairRef pr;
  pr.p1 = M.T_p[a];
  pr.e1 = M.T_e[a];
  pr.p2 = M.T_p[b];
  pr.m2 = M.T_m[b];
  pr.e2 = M.T_e[b];
  if (x) *x = M.T_m[a];
  return pr;
}
// The pair loop's `x = epsilon / P1` (impute.py:457-459) depends on the FIRST list's entry only: one IEEE division per entry
// and pass instead of one per scored pair (a phase scores up to 128 x 128 of them).  The quotients live where a second list
// keeps its prefix minima -- a first list (even side) has no use for those.  All threads call; barrier inside.
__device__ __forceinline__ void mid_set_eps(MidShared &M, double eps) {
  for (int i = 0; i < M.nph; ++i) {
    const uint32_t a0 = M.seg[2 * i], n1 = M.tlen[2 * i];
    for (uint32_t h = threadIdx.x; h < n1; h += GRIM_WG) M.T_m[a0 + h] = eps / M.T_p[a0 + h];
  }
  __syncthreads();
}
// pair_accept (grim_pair.h) with the quotient at hand: the same comparisons on the same values
__device__ __forceinline__ bool mid_accept(double x, const PairRef &pr, double w) {
  if (!(pr.m2 >= x)) return false;
  if (!(w > 0.0)) return false;
  const double thr = (ENT_HAP(pr.e1) == ENT_HAP(pr.e2)) ? x * 2.0 : x;
  return w * pr.p2 >= thr;
}
__device__ __forceinline__ double mid_prior(const MidShared &M, const double *prior, int P, bool lds_prior, uint32_t e1, uint32_t e2) {
  const uint32_t cell = ENT_POP(e1) * (uint32_t)P + ENT_POP(e2);
  return lds_prior ? M.lp[cell] : prior[cell];
}

// One pass of calc_haps_pairs* at eps over all scored pairs (impute.py:444-658): afterwards bit (boff[i] + r) of the bitmap
// says whether pair r of phase i is a WINNER (accepted, and the first of its unordered {(hap, pop), (hap, pop)}).  Returns the
// number of winners, *maxp their largest probability; GRIM_NONE when the pass would need a bigger dedup table than the arena
// holds (the subject is the general kernel's).  All threads call.
__device__ inline uint32_t mid_pass(const DevArgs &A, MidShared &M, const MidView &V, const double *prior, bool lds_prior, bool nodup,
                                    double eps, bool have_x, double *maxp) {
  const int P = A.g.P;
  const int lane = lane_id(), wv = wave_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  uint32_t cnt = 0;
  double amx = 0.0;
  if (!have_x) mid_set_eps(M, eps);  // (the ladder's last sweep left the quotients of its epsilon)
  for (int i = 0; i < M.nph; ++i) {
    const uint32_t n2 = M.tlen[2 * i + 1], npi = M.tlen[2 * i] * n2;
    if (!npi) continue;
    const uint32_t magic = tile_magic(n2), w0 = M.boff[i] >> 6, nw = (npi + 63) >> 6;
    for (uint32_t w = wv; w < nw; w += GRIM_NWAVE) {
      const uint32_t r = (w << 6) + lane;
      bool on = false;
      if (r < npi) {
        double x;
        const PairRef pr = mid_pair(M, i, r, n2, magic, &x);
        const double wgt = mid_prior(M, prior, P, lds_prior, pr.e1, pr.e2);
        on = mid_accept(x, pr, wgt);
        if (on && nodup) {
          const double prob = pair_prob(pr, wgt);
          amx = prob > amx ? prob : amx;
        }
      }
      const uint64_t m = __ballot(on);
      if (lane == 0) V.bm[w0 + w] = m;
      cnt += (uint32_t)__popcll(m);
    }
  }
  if (lane == 0) M.tmp[wv] = cnt;
  __syncthreads();
  uint32_t nA = 0;
  for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) nA += M.tmp[w2];
  __syncthreads();
  if (!nodup && nA > 0) {
    // First-wins dedup (impute.py:506-511, 603-611) WITHOUT a table of pairs.  A pair is (entity x of side 2i at h, entity y
    // of side 2i+1 at k); the same unordered {x, y} can only come up again where x and y sit in the two lists of another
    // phase (either way round) or the other way round in this one.  Every entity knows the sides it is in (a 32-bit mask)
    // and its position in each of them, so the earlier occurrences of a pair are a few bit operations away, and whether
    // one of them was ACCEPTED is a bit of the pass's bitmap: the pair loses iff an occurrence with a smaller bit number
    // has its bit set.  O(list entries) to set up, once per subject; no limit on the pairs a pass accepts.
    if (!M.bc[6]) {
      const int nsides = 2 * M.nph;
      for (uint32_t t = threadIdx.x; t < MD_ESLOTS; t += GRIM_WG) V.smask[t] = 0;
      if (threadIdx.x == 0) M.bc[5] = 0;
      __syncthreads();
      for (int sd = 0; sd < nsides; ++sd) {  // the entity index: slot of (hap, pop); bit 31 marks a used slot
        const uint32_t a0 = M.seg[sd], len = M.tlen[sd];
        for (uint32_t q = threadIdx.x; q < len; q += GRIM_WG) {
          const uint32_t key = M.T_e[a0 + q] | 0x80000000u;
          uint32_t t = ((key * 0x9E3779B1u) >> 16) & (MD_ESLOTS - 1u);
          for (;;) {
            uint32_t c = __hip_atomic_load(&V.smask[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (c == 0) {
              uint32_t expect = 0;
              c = __hip_atomic_compare_exchange_strong(&V.smask[t], &expect, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                      ? key : expect;
            }
            if (c == key) break;
            t = (t + 1u) & (MD_ESLOTS - 1u);
          }
          V.eslot[a0 + q] = (uint16_t)t;
        }
      }
      __syncthreads();
      for (uint32_t t = threadIdx.x; t < MD_ESLOTS; t += GRIM_WG) V.smask[t] = 0;  // the keys are spent: the slots hold side masks now
      __syncthreads();
      for (int sd = 0; sd < nsides; ++sd) {
        const uint32_t a0 = M.seg[sd], len = M.tlen[sd];
        for (uint32_t q = threadIdx.x; q < len; q += GRIM_WG) {
          const uint32_t old = __hip_atomic_fetch_or(&V.smask[V.eslot[a0 + q]], 1u << sd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if ((old >> sd) & 1u) M.bc[5] = 1;  // an entity twice in one list (never, by adjs_query's dict): not this scheme's case
        }
      }
      __syncthreads();
      {  // first occurrence per slot: exclusive prefix of the masks' popcounts (eight slots per thread)
        constexpr uint32_t PER = MD_ESLOTS / GRIM_WG;
        uint32_t sum = 0;
        for (uint32_t q = 0; q < PER; ++q) sum += (uint32_t)__popc((uint32_t)V.smask[threadIdx.x * PER + q]);
        uint32_t total;
        uint32_t at = wg_excl_scan(sum, M.tmp, total);
        for (uint32_t q = 0; q < PER; ++q) {
          V.eoff[threadIdx.x * PER + q] = (uint16_t)at;
          at += (uint32_t)__popc((uint32_t)V.smask[threadIdx.x * PER + q]);
        }
      }
      __syncthreads();
      for (int sd = 0; sd < nsides; ++sd) {  // positions, by entity and (ascending) side
        const uint32_t a0 = M.seg[sd], len = M.tlen[sd];
        for (uint32_t q = threadIdx.x; q < len; q += GRIM_WG) {
          const uint32_t t = V.eslot[a0 + q];
          V.epos[V.eoff[t] + (uint32_t)__popc((uint32_t)V.smask[t] & ((1u << sd) - 1u))] = (uint8_t)q;
        }
      }
      if (threadIdx.x == 0) M.bc[6] = 1;
      __syncthreads();
    }
    if (M.bc[5]) return GRIM_NONE;  // (uniform: read behind a barrier)
    cnt = 0;
    for (int i = 0; i < M.nph; ++i) {
      const uint32_t n2 = M.tlen[2 * i + 1], npi = M.tlen[2 * i] * n2;
      if (!npi) continue;
      const uint32_t magic = tile_magic(n2), w0 = M.boff[i] >> 6, nw = (npi + 63) >> 6;
      const uint32_t below = (4u << (2 * i)) - 1u;  // sides of the phases up to and including this one
      for (uint32_t w = wv; w < nw; w += GRIM_NWAVE) {
        const uint64_t m = V.bm[w0 + w];
        if (m == 0) {
          if (lane == 0) V.bm2[w0 + w] = 0;
          continue;
        }
        bool win = false;
        if ((m >> lane) & 1ull) {
          const uint32_t r = (w << 6) + lane, bit = M.boff[i] + r;
          const uint32_t h = n2 > 1 ? __umulhi(r, magic) : r, k = r - h * n2;
          const uint32_t ea = M.seg[2 * i] + h, eb = M.seg[2 * i + 1] + k;
          const uint32_t tx = V.eslot[ea], ty = V.eslot[eb];
          const uint32_t mx = V.smask[tx], my = V.smask[ty];
          const uint32_t ox = V.eoff[tx], oy = V.eoff[ty];
          // bit 2j of `same`: x is in side 2j and y in side 2j+1; of `swap`: y in side 2j and x in side 2j+1
          uint32_t same = mx & (my >> 1) & 0x55555555u & below & ~(1u << (2 * i));  // (not this occurrence itself)
          uint32_t swp = my & (mx >> 1) & 0x55555555u & below;
          bool lost = false;
          while ((same | swp) && !lost) {
            const bool sw = same == 0;
            const uint32_t bits = sw ? swp : same;
            const int sd = __ffs((int)bits) - 1;  // the even side of the phase, 2j
            if (sw) swp &= swp - 1; else same &= same - 1;
            const int j = sd >> 1;
            // entity of the phase's first list: x (same way round) or y (swapped); its position there, and the other's
            const uint32_t m1 = sw ? my : mx, m2 = sw ? mx : my, o1 = sw ? oy : ox, o2 = sw ? ox : oy;
            const uint32_t h2 = V.epos[o1 + (uint32_t)__popc(m1 & ((1u << sd) - 1u))];
            const uint32_t k2 = V.epos[o2 + (uint32_t)__popc(m2 & ((2u << sd) - 1u))];
            const uint32_t bit2 = M.boff[j] + h2 * M.tlen[2 * j + 1] + k2;
            if (bit2 < bit && ((V.bm[bit2 >> 6] >> (bit2 & 63u)) & 1ull)) lost = true;
          }
          win = !lost;
          if (win) {
            const PairRef pr = mid_pair(M, i, r, n2, magic);
            const double prob = pair_prob(pr, mid_prior(M, prior, P, lds_prior, pr.e1, pr.e2));
            amx = prob > amx ? prob : amx;
          }
        }
        const uint64_t mw = __ballot(win);
        if (lane == 0) V.bm2[w0 + w] = mw;
        cnt += (uint32_t)__popcll(mw);
      }
    }
    __syncthreads();
    {  // the winners become the pass's bitmap
      const uint32_t W = M.boff[GRIM_MAXPH] >> 6;
      for (uint32_t w = threadIdx.x; w < W; w += GRIM_WG) V.bm[w] = V.bm2[w];
    }
    if (lane == 0) M.tmp[wv] = cnt;
    __syncthreads();
    nA = 0;
    for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) nA += M.tmp[w2];
    __syncthreads();
  }
  (void)lt;
  for (int d = 32; d > 0; d >>= 1) {
    const double o = __shfl_xor(amx, d);
    amx = o > amx ? o : amx;
  }
  if (lane == 0) M.dtmp[wv] = amx;
  __syncthreads();
  double mx = M.dtmp[0];
  for (int w2 = 1; w2 < GRIM_NWAVE; ++w2) mx = M.dtmp[w2] > mx ? M.dtmp[w2] : mx;
  __syncthreads();
  *maxp = mx;
  return nA;
}

// returns 0 = done (result header written; tables emitted or queued), 2 = hand over to the general kernel.  All threads
// call; every return is uniform and preceded by a barrier.
__device__ inline int mid_subject(const DevArgs &A, MidShared &M, uint32_t si) {
  const DevGraph &g = A.g;
  const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
  const int P = g.P;
  const MidView V = mid_view(M);
  const uint64_t lt = (1ull << lane) - 1ull;
#ifdef GRIM_STAMPS
  unsigned long long _mt0 = wall_clock64();
#endif
  // ---- subject, phases (gen_phases, impute.py:274-303) ------------------------------------------------------------------
  if (tid < 16) ((uint32_t *)&M.subj)[tid] = ((const uint32_t *)&A.subj[si])[tid];
  if (tid < (int)(sizeof(grim_subject_result) / 4)) ((uint32_t *)&M.out)[tid] = 0;
  if (tid < GRIM_SIDES) M.cnt_side[tid] = 0;
  if (tid == 0) M.bc[6] = 0;  // the dedup's entity index is not built yet
  __syncthreads();
  const grim_subject &sj = M.subj;
  const int n = sj.n_loci;
  const uint16_t *tok = A.tok + sj.tok_off;
  const double *prior = A.priors + (uint64_t)sj.prior_idx * P * P;
  const bool lds_prior = P * P <= 64;
  if (tid == 0) {
    const uint32_t same = sj.pad[0];
    const uint32_t het = ((1u << n) - 1u) & ~same;
    const uint32_t movable = het & ~(uint32_t)sj.flags;  // phase mask: fixed positions never switch
    uint32_t seen = 0;
    int cnt = 0;
    for (uint32_t i = 0; n >= 1 && i < (1u << (n - 1)); ++i) {
      const uint32_t p = i & movable;
      if (!((seen >> p) & 1u)) {
        seen |= (1u << p) | (1u << (p ^ het));
        M.ph_pat[cnt++] = (uint8_t)p;
      }
    }
    M.nph = cnt;
    uint32_t acc = 0;
    for (int l = 0; l < n; ++l)
      for (int s = 0; s < 2; ++s) {
        M.toff[l][s] = acc;
        acc += sj.cnt[l][s];
      }
  }
  if (tid >= 64 && tid < 64 + P) M.diag[tid - 64] = prior[(tid - 64) * P + (tid - 64)];
  if (lds_prior && tid >= 128 && tid < 128 + P * P) M.lp[tid - 128] = prior[tid - 128];
  __syncthreads();
  const int nph = M.nph, nsides = 2 * nph;
  uint32_t typed = 0;
  for (int l = 0; l < n; ++l) typed |= 1u << sj.slot[l];
  const bool full_nodes = typed == g.full_mask;
  // ---- candidates per side (open_phases' cartesian branch, impute.py:926-944) ----------------------------------------------
  if (wv == 0) {
    uint32_t my_nc = 0;
    bool bad = n < 1 || !subject_order_ok(g, typed);
    if (lane < nsides) {
      const uint32_t pat = M.ph_pat[lane >> 1];
      uint64_t options = 1, nc = 1;
      for (int l = 0; l < n; ++l) {
        const int c = (int)((pat >> l) & 1u) ^ (lane & 1);
        nc *= sj.cnt[l][c];
        if (nc > 0xFFFFFFull) nc = 0xFFFFFFull;
        options *= (uint64_t)sj.wid[l][c];
        if (options > 0xFFFFFFFFFFFFull) options = 0xFFFFFFFFFFFFull;
      }
      my_nc = (uint32_t)nc;
      bad = bad || !(options < A.prm.opt_threshold) || nc > MD_C;  // a label-scan side: the general kernel's
    }
    const uint32_t inc = wave_incl_scan(my_nc);
    if (lane < nsides) M.cand_start[lane + 1] = inc;
    if (lane == 0) M.cand_start[0] = 0;
    const uint64_t anybad = __ballot(bad);
    if (lane == 0) M.bc[0] = anybad != 0 ? 1u : 0u;
  }
  __syncthreads();
  const uint32_t C = M.cand_start[nsides];
  if (M.bc[0] || C > MD_C || nsides == 0) {
    __syncthreads();
    MBAIL(0);
    return 2;
  }
  MSTAMP(0);
  // ---- 1. all candidates' look-ups, hits compacted in candidate order ----------------------------------------------------
  uint32_t H = 0;
  {
    constexpr int NQ = 4;
    for (uint32_t c0 = 0; c0 < C; c0 += GRIM_WG * NQ) {
      uint64_t keyq[NQ];
      uint32_t hq[NQ], nodeq[NQ];
      int sideq[NQ];
      HtEnt entq[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const uint32_t gi = c0 + (uint32_t)q * GRIM_WG + tid;
        keyq[q] = 0;
        sideq[q] = 0;
        if (gi < C) {
          int lo = 0, hi = nsides - 1;
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (M.cand_start[mid] <= gi) lo = mid; else hi = mid - 1;
          }
          sideq[q] = lo;
          uint32_t rem = gi - M.cand_start[lo];
          const uint32_t pat = M.ph_pat[lo >> 1];
          uint64_t key = 0;
          for (int l = n - 1; l >= 0; --l) {  // mixed-radix digits, position 0 most significant (cutils.pyx:21-29)
            const int c = (int)((pat >> l) & 1u) ^ (lo & 1);
            const uint32_t cn = sj.cnt[l][c];
            const uint32_t d = rem % cn;
            rem /= cn;
            key |= (uint64_t)(tok[M.toff[l][c] + d] + 1u) << (GRIM_ABITS * sj.slot[l]);
          }
          keyq[q] = key;
        }
        hq[q] = (uint32_t)mix64(keyq[q]) & g.ht_mask;
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (keyq[q]) entq[q] = g.ht[hq[q]];  // the first probes of the thread's candidates are in flight together
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        nodeq[q] = GRIM_NONE;
        if (keyq[q]) {
          if (entq[q].key == keyq[q]) nodeq[q] = entq[q].val;
          else if (entq[q].key != 0) nodeq[q] = graph_lookup_from(g, keyq[q], (hq[q] + 1u) & g.ht_mask);
        }
      }
      uint64_t mq[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        mq[q] = __ballot(nodeq[q] != GRIM_NONE);
        if (lane == 0) M.tmp[q * GRIM_NWAVE + wv] = (uint32_t)__popcll(mq[q]);
      }
      __syncthreads();
      uint32_t run = H, base[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        for (int w2 = 0; w2 < GRIM_NWAVE; ++w2) {
          if (w2 == wv) base[q] = run;
          run += M.tmp[q * GRIM_NWAVE + w2];
        }
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (nodeq[q] != GRIM_NONE) {
          const uint32_t pos = base[q] + (uint32_t)__popcll(mq[q] & lt);
          if (pos < MD_H) {
            V.hnode[pos] = nodeq[q];
            V.hside[pos] = (uint8_t)sideq[q];
          }
        }
      H = run;
      __syncthreads();
    }
  }
  if (H > MD_H) {  // (behind the loop's last barrier; H is uniform)
    MBAIL(1);
    return 2;
  }
  MSTAMP(1);
  // ---- 2. the hits' CSR rows (adjs_query: a full-label node is its own answer, a partial node's top links otherwise) -------
  uint32_t T = 0;
  {
    const uint32_t per = (H + GRIM_WG - 1) / GRIM_WG;  // <= 4, a thread's hits are consecutive
    const uint32_t h0 = tid * per < H ? tid * per : H, h1 = h0 + per < H ? h0 + per : H;
    uint32_t cn[4] = {0, 0, 0, 0}, bs[4] = {0, 0, 0, 0}, sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (h0 + q < h1) {
        const uint32_t node = V.hnode[h0 + q];
        if (full_nodes) {
          cn[q] = 1;
        } else {
          const uint32_t a = g.a_start[node], b = g.a_start[node + 1];
          cn[q] = b > a ? b - a : 0;  // (the reference's sentinel quirk: networkx_graph.py:195, 267-272)
          bs[q] = a;
        }
        sum += cn[q];
      }
    uint32_t total;
    uint32_t at = wg_excl_scan(sum, M.tmp, total);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (h0 + q < h1) {
        V.hstart[h0 + q] = at;
        V.hbase[h0 + q] = bs[q];
        at += cn[q];
      }
    if (tid == 0) V.hstart[H] = total;
    T = total;
    __syncthreads();
  }
  if (T > MD_T) {
    MBAIL(2);
    return 2;
  }
  MSTAMP(2);
  // ---- 3. the rows' haplotypes and their frequency vectors -> entries (hap, pop) with p > 0, in stream order -----------------
  uint32_t E = 0;
  {
    const uint32_t per = (T + GRIM_WG - 1) / GRIM_WG;  // <= 8
    const uint32_t t0 = tid * per < T ? tid * per : T, t1 = t0 + per < T ? t0 + per : T;
    uint32_t hap[8], own[8];
    uint64_t msk[8];
    uint32_t o = 0;
    if (t0 < t1) {  // owner of t0: last hit whose start is <= t0 (hits without a row have start[h] == start[h + 1])
      uint32_t lo = 0, hi = H - 1;
      while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (V.hstart[mid] <= t0) lo = mid; else hi = mid - 1;
      }
      o = lo;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      hap[q] = 0;
      own[q] = 0;
      msk[q] = 0;
      const uint32_t t = t0 + q;
      if (t < t1) {
        while (V.hstart[o + 1] <= t) ++o;
        own[q] = o;
        hap[q] = full_nodes ? V.hnode[o] : g.a_nbr[V.hbase[o] + (t - V.hstart[o])];
      }
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (t0 + q < t1) {
        if (P <= 4) {
          double f[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) f[j] = j < P ? g.freq[(uint64_t)hap[q] * P + j] : 0.0;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (f[j] > 0.0) msk[q] |= 1ull << j;
        } else {
          for (int j = 0; j < P; ++j)
            if (g.freq[(uint64_t)hap[q] * P + j] > 0.0) msk[q] |= 1ull << j;
        }
        cnt += (uint32_t)__popcll(msk[q]);
      }
    uint32_t total;
    uint32_t pos = wg_excl_scan(cnt, M.tmp, total);
    E = total;
    if (E <= MD_E) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (t0 + q < t1 && msk[q]) {
          const uint32_t s = V.hside[own[q]];
          uint64_t m = msk[q];
          atomicAdd(&M.cnt_side[s], (uint32_t)__popcll(m));
          while (m) {
            const int j = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            V.entp[pos] = g.freq[(uint64_t)hap[q] * P + j];
            V.ente[pos] = hap[q] | ((uint32_t)j << 24);
            V.ents[pos] = (uint8_t)s;
            ++pos;
          }
        }
    }
    __syncthreads();
  }
  if (E > MD_E) {
    MBAIL(3);
    return 2;
  }
  MSTAMP(3);
  // ---- 4. segments, ranks, top lists (convert_list_to_one_dim, impute.py:424-442) -------------------------------------------
  const uint32_t K = A.prm.top_n;
  if (wv == 0) {
    const uint32_t sc = lane < nsides ? M.cnt_side[lane] : 0;
    const uint32_t kept = sc < K ? sc : K;
    const uint32_t i1 = wave_incl_scan(sc), i2 = wave_incl_scan(kept);
    const uint32_t kept_total = __shfl(i2, 63);
    if (lane < nsides) {
      M.seg_in[lane] = i1 - sc;
      M.seg[lane] = i2 - kept;
      M.tlen[lane] = kept;
    }
    if (lane == nsides) {
      M.seg_in[lane] = E;
      M.seg[lane] = kept_total;
    }
    const uint64_t big = __ballot(sc > MD_SEG);
    if (lane == 0) M.bc[1] = big != 0 ? 1u : 0u;
  }
  __syncthreads();
  {
    const uint32_t Ltot = M.seg[nsides];
    if (M.bc[1] || Ltot > MD_L) {
      __syncthreads();
      MBAIL(4);
      return 2;
    }
  }
  for (uint32_t e = tid; e < E; e += GRIM_WG) V.key[e] = V.entp[e] * M.diag[V.ente[e] >> 24];  // (the hit arrays are spent)
  __syncthreads();
  for (uint32_t e = tid; e < E; e += GRIM_WG) {
    const uint32_t s = V.ents[e];
    const uint32_t a = M.seg_in[s], b = M.seg_in[s + 1];
    const double k = V.key[e];
    uint32_t rank = 0;
    for (uint32_t e2 = a; e2 < b && rank < K; ++e2) {  // (an entry with K bigger ones in front of it is out: no need for its rank)
      const double k2 = V.key[e2];
      rank += (k2 > k || (k2 == k && e2 < e)) ? 1u : 0u;
    }
    if (rank < K) {
      M.T_p[M.seg[s] + rank] = V.entp[e];
      M.T_e[M.seg[s] + rank] = V.ente[e];
    }
  }
  __syncthreads();
  // prefix minimum of a SECOND list for the pair loop's break (impute.py:463-464, 545-546): a wave per list, shuffles
  for (int s = 2 * wv + 1; s < nsides; s += 2 * GRIM_NWAVE) {
    const uint32_t a = M.seg[s], len = M.tlen[s];
    double carry = __longlong_as_double(0x7FF0000000000000ll);
    for (uint32_t r0 = 0; r0 < len; r0 += 64) {
      const uint32_t r = r0 + lane;
      double v = r < len ? M.T_p[a + r] : __longlong_as_double(0x7FF0000000000000ll);
      for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(v, d);
        if (lane >= d && o < v) v = o;
      }
      if (carry < v) v = carry;
      if (r < len) M.T_m[a + r] = v;
      carry = __shfl(v, 63);
    }
  }
  if (tid == 64) {
    uint32_t acc = 0, bits = 0;
    for (int i = 0; i < nph; ++i) {
      M.poff[i] = acc;
      M.boff[i] = bits;
      const uint32_t npi = M.tlen[2 * i] * M.tlen[2 * i + 1];
      acc += npi;
      bits += (npi + 63u) & ~63u;
    }
    for (int i = nph; i <= GRIM_MAXPH; ++i) {
      M.poff[i] = acc;
      M.boff[i] = bits;
    }
  }
  // every scored pair a different unordered entity pair?  (prepare_lists, grim_plan_a.h: the two lists of every position
  // are the same text or disjoint sets, and some position differs)
  if (wv == 1) {
    bool het = false, clash = false;
    if (lane < n && !((sj.pad[0] >> lane) & 1u)) {
      het = true;
      const uint16_t *l0 = tok + M.toff[lane][0], *l1 = tok + M.toff[lane][1];
      const uint32_t c0 = sj.cnt[lane][0], c1 = sj.cnt[lane][1];
      for (uint32_t a = 0; a < c0 && !clash; ++a)
        for (uint32_t b = 0; b < c1 && !clash; ++b) clash = l0[a] == l1[b];
    }
    const uint64_t hm = __ballot(het), cm = __ballot(clash);
    if (lane == 0) M.bc[2] = (hm != 0 && cm == 0 && !(A.flags & GRIM_F_NO_NODUP)) ? 1u : 0u;
  }
  __syncthreads();
  const uint32_t np = M.poff[GRIM_MAXPH];
  const bool nodup = M.bc[2] != 0;
  if (np > MD_NP) {
    __syncthreads();
    MBAIL(5);
    return 2;
  }
  MSTAMP(4);
  // ---- 5. the ladder (impute.py:1665-1687): first step at which any pair is accepted ---------------------------------------------
  // One sweep per ladder step, the step's quotients precomputed per first-list entry; a sweep ends as soon as any thread has
  // an accepted pair.  (Testing every pair against every step costs a division per pair and step.)
  int e_first = A.prm.n_ladder;
  for (int idx = 0; np > 0 && idx < A.prm.n_ladder; ++idx) {
    if (tid == 0) M.bc[3] = 0;
    mid_set_eps(M, A.prm.ladder[idx]);
    bool found = false;
    for (int i = 0; i < nph && !found; ++i) {
      const uint32_t n2 = M.tlen[2 * i + 1], npi = M.tlen[2 * i] * n2;
      const uint32_t magic = tile_magic(n2);
      for (uint32_t r = tid; r < npi; r += GRIM_WG) {
        double x;
        const PairRef pr = mid_pair(M, i, r, n2, magic, &x);
        if (mid_accept(x, pr, mid_prior(M, prior, P, lds_prior, pr.e1, pr.e2)) || ((r & (7u * GRIM_WG)) == 0 && M.bc[3])) {
          found = true;
          break;
        }
      }
    }
    if (found) M.bc[3] = 1;
    __syncthreads();
    const bool any = M.bc[3] != 0;
    __syncthreads();
    if (any) {
      e_first = idx;
      break;
    }
  }
  MSTAMP(5);
  // ---- 6. the passes: at the first accepting step for MaxProb, then at MaxProb / 100000 (impute.py:1685-1693) ----------------------
  uint32_t nU = 0;
  double mx = 0.0;
  if (np > 0 && e_first < A.prm.n_ladder) {
    double eps = A.prm.ladder[e_first];
    if (eps > 0.0) {
      const uint32_t r1 = mid_pass(A, M, V, prior, lds_prior, nodup, eps, true, &mx);
      if (r1 == GRIM_NONE) {
        MBAIL(6);
        return 2;
      }
      eps = mx / 100000.0;
    }
    MSTAMP(6);
    nU = mid_pass(A, M, V, prior, lds_prior, nodup, eps, eps == A.prm.ladder[e_first], &mx);
    if (nU == GRIM_NONE) {
      MBAIL(6);
      return 2;
    }
  }
  MSTAMP(7);
  // ---- 7. the winners, in pair order -------------------------------------------------------------------------------------------
  uint8_t status = GRIM_ST_MISS, reason = 0;
  if (nU > 0) {
    const uint32_t W = M.boff[GRIM_MAXPH] >> 6;  // words of the bitmap
    {
      const uint32_t per = (W + GRIM_WG - 1) / GRIM_WG;
      const uint32_t w0 = tid * per < W ? tid * per : W, w1 = w0 + per < W ? w0 + per : W;
      uint32_t sum = 0;
      for (uint32_t w = w0; w < w1; ++w) sum += (uint32_t)__popcll(V.bm[w]);
      uint32_t total;
      uint32_t at = wg_excl_scan(sum, M.tmp, total);  // (the prefix minima are spent: V.wpre lives there)
      for (uint32_t w = w0; w < w1; ++w) {
        V.wpre[w] = at;
        at += (uint32_t)__popcll(V.bm[w]);
      }
      if (tid == 0) {
        uint32_t off = 0;
        if (nU > 64) {
          off = atomicAdd(A.queue + 8, nU);
          if (off + nU > A.ppool_cap) {
            atomicExch(&A.counters[4], 1ull);  // the run reports the overflow; the caller grows the pool and runs again
            off = GRIM_NONE;
          }
        }
        M.bc[4] = off;
        M.out.n_pairs = nU;
        M.out.max_prob = mx;
      }
      __syncthreads();
    }
    const uint32_t off = M.bc[4];
    if (off != GRIM_NONE) {
      for (int i = 0; i < nph; ++i) {
        const uint32_t n2 = M.tlen[2 * i + 1], npi = M.tlen[2 * i] * n2;
        if (!npi) continue;
        const uint32_t magic = tile_magic(n2), w0 = M.boff[i] >> 6, nw = (npi + 63) >> 6;
        for (uint32_t w = wv; w < nw; w += GRIM_NWAVE) {
          const uint64_t m = V.bm[w0 + w];
          if (!((m >> lane) & 1ull)) continue;
          const uint32_t pos = V.wpre[w0 + w] + (uint32_t)__popcll(m & lt);
          const PairRef pr = mid_pair(M, i, (w << 6) + lane, n2, magic);
          const double prob = pair_prob(pr, mid_prior(M, prior, P, lds_prior, pr.e1, pr.e2));
          if (nU <= 64) {
            V.se1[pos] = pr.e1;
            V.se2[pos] = pr.e2;
            V.sprob[pos] = prob;
          } else {
            PairRec rec;
            rec.k1 = g.node_key[ENT_HAP(pr.e1)];
            rec.k2 = g.node_key[ENT_HAP(pr.e2)];
            rec.prob = prob;
            rec.e1 = pr.e1;
            rec.e2 = pr.e2;
            A.ppool[off + pos] = rec;
          }
        }
      }
    }
    __syncthreads();
    if (nU <= 64) {
      if (wv == 0) {
        uint32_t e1 = 0, e2 = 0;
        double prob = 0.0;
        uint64_t k1 = 0, k2 = 0;
        if (lane < (int)nU) {
          e1 = V.se1[lane];
          e2 = V.se2[lane];
          prob = V.sprob[lane];
          k1 = g.node_key[ENT_HAP(e1)];
          k2 = g.node_key[ENT_HAP(e2)];
        }
        RowBlock rb = {0, 0, 0};
        emit_small_core(A, nU, e1, e2, prob, k1, k2, M.out, rb, 3);
      }
    } else if (off != GRIM_NONE && tid == 0) {
      TabWork w;
      w.si = si;
      w.n = nU;
      w.off = off;
      w.mask = 3;
      if (nU <= GRIM_TAB_T1_MAX)
        A.t1_list[atomicAdd(A.queue + 9, 1u)] = w;
      else
        A.t2_list[atomicAdd(A.queue + 10, 1u)] = w;
    }
    status = GRIM_ST_OK;
  } else if (A.prm.planb) {
    status = GRIM_ST_UNSUPPORTED;  // replaced by the plan-B kernel's verdict when it runs
    reason = 2;
    if (tid == 0 && A.next_list) push_next(A, si, sj.n_loci <= GRIM_HEAVY_LOCI);
  }
  __syncthreads();
  if (tid == 0) {
    M.out.status = status;
    M.out.reason = reason;
    M.out.plan = 'a';
    A.res[si] = M.out;
    M.wctr[0] += C;  // algorithmic bytes of a subject this kernel completed: probes, CSR ids, frequency vectors
    if (!full_nodes) M.wctr[1] += T;
    M.wctr[2] += T;
  }
  __syncthreads();
  MSTAMP(8);
  return 0;
}

// One workgroup = one subject at a time, pulled from a work counter: first the one-wave kernel's heavier hand-overs (the back of
// its list), then this launch's own subjects (heaviest first), then the light hand-overs -- the order the general kernel took
// them in before this kernel stood in front of it.
__global__ __launch_bounds__(GRIM_WG, GRIM_MID_WG_PER_CU) void grim_plan_a_mid_kernel(DevArgs A) {
  __shared__ MidShared M;
  const int tid = threadIdx.x;
  const uint32_t n_bail = A.bail_list ? A.queue[5] : 0, n_bail_heavy = A.bail_list ? A.queue[7] : 0;
  if (tid < 4) M.wctr[tid] = 0;
  __syncthreads();
  for (;;) {
    if (tid == 0) M.bc[7] = atomicAdd(A.queue + 16, 1u);
    __syncthreads();
    const uint32_t w = M.bc[7];
    __syncthreads();
    if (w >= A.n_work + n_bail + n_bail_heavy) break;
    uint32_t si;
    if (w < n_bail_heavy)
      si = A.bail_list[A.n_medium - 1u - w];
    else if (w - n_bail_heavy < A.n_work)
      si = A.order[w - n_bail_heavy];
    else
      si = A.bail_list[w - n_bail_heavy - A.n_work];
    const int rc = mid_subject(A, M, si);
    if (rc != 0 && tid == 0) A.mid_list[atomicAdd(A.queue + 17, 1u)] = si;
    __syncthreads();
  }
  if (tid == 0) {
    atomicAdd(&A.counters[0], M.wctr[0]);
    atomicAdd(&A.counters[1], M.wctr[1]);
    atomicAdd(&A.counters[2], M.wctr[2]);
  }
}
