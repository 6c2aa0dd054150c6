// grim_medium.h -- Plan A for the typical subject (some '/' ambiguity, some untyped loci) by ONE WAVE,
// without a single workgroup barrier and without HBM scratch.  Same semantics as grim_plan_a_kernel
// (grim_plan_a.h / grim_pair.h cite the reference lines); the differences are purely structural:
//
//   * candidates of ALL phase sides are flattened over the 64 lanes (lane -> (side, candidate) by a
//     binary search in an LDS prefix array), so the hash probes and CSR gathers of up to 32 sides
//     are in flight together instead of one side after the other;
//   * a side's entries are few (<= 64), so its top-K list is produced by counting ranks inside the
//     side's segment; pair dedup looks at the earlier pairs' keys in LDS; the four output tables are
//     the one-wave shuffle code of grim_pair.h (emit_small_core).
//
// Limits (subject is handed to the general kernel when one is exceeded): MW_E entries in total,
// 64 entries per side, MW_NP haplotype pairs, 64 accepted pairs.
#pragma once
#include "grim_pair.h"

#ifndef MW_E
#define MW_E 128
#endif
#define MW_NP 512
// resident one-wave workgroups per CU: bounded by the LDS each needs (sizeof(WaveMed) = 9.2 KB of 160 KB).
// Measured on 92 000 subjects of the mixed workload: 2.35 / 1.44 / 1.18 / 1.16 ms with 5 / 10 / 16 / 20 waves per CU;
// subjects with more than 128 entries nearly always exceed the pair limits anyway (+1 % hand-overs against 384).
#ifndef GRIM_MEDIUM_WAVES_PER_CU
#define GRIM_MEDIUM_WAVES_PER_CU 16
#endif

struct WaveMed {
  // entries in stream order; the area is reused for the pair keys once the lists are built
  union {
    struct {
      double ent_p[MW_E];
      uint32_t ent_e[MW_E];
    } e;
    uint64_t pkeys[MW_NP];
  } u;
  uint8_t ent_side[MW_E];
  double T_p[MW_E], T_m[MW_E];
  uint32_t T_e[MW_E];
  double diag[GRIM_MAXPOP];           // prior[j][j]
  uint16_t seg[GRIM_SIDES + 1];       // first entry of each side
  uint16_t tlen[GRIM_SIDES];          // list length per side
  uint16_t cand_start[GRIM_SIDES + 1];
  uint32_t cstart[65];
  uint32_t cnode[64];
  uint8_t cside[64];
  uint32_t poff[GRIM_MAXPH + 1];
  uint16_t useq[MW_NP];  // accepted pairs in sequence order
  double uprob[64];
  grim_subject subj;
  uint32_t toff[GRIM_MAXL][2];
  uint8_t ph_pat[GRIM_MAXPH];
  int nph;
  uint32_t cnt_side[GRIM_SIDES];
};

__device__ __forceinline__ PairRef med_pair(const WaveMed &M, uint32_t f) {
  int i = 0;
  while (f >= M.poff[i + 1]) ++i;
  uint32_t r = f - M.poff[i];
  uint32_t n2 = M.tlen[2 * i + 1];
  uint32_t h = r / n2, k = r - h * n2;
  uint32_t a = M.seg[2 * i] + h, b = M.seg[2 * i + 1] + k;
  PairRef pr;
  pr.p1 = M.T_p[a];
  pr.e1 = M.T_e[a];
  pr.p2 = M.T_p[b];
  pr.m2 = M.T_m[b];
  pr.e2 = M.T_e[b];
  return pr;
}

// returns 0 = done (result written, or -- more than 64 accepted pairs -- the tables queued for the table kernels),
// 2 = hand over to the general kernel (a size limit was exceeded)
// acc: the wave's algorithmic-byte counts (probes, CSR ids, frequency vectors) of the subjects it completed
__device__ inline int medium_subject(const DevArgs &A, WaveMed &M, uint32_t si, unsigned long long (&acc)[3], RowBlock &rb) {
  const DevGraph &g = A.g;
  const int lane = lane_id();
  const int P = g.P;
  // ---- subject, phases ------------------------------------------------------------------------
  if (lane < 16) ((uint32_t *)&M.subj)[lane] = ((const uint32_t *)&A.subj[si])[lane];
  WAVE_SYNC();
  const grim_subject &sj = M.subj;
  const int n = sj.n_loci;
  const uint16_t *tok = A.tok + sj.tok_off;
  const double *prior = A.priors + (uint64_t)sj.prior_idx * P * P;
  if (lane == 0) {
    uint32_t same = sj.pad[0];
    uint32_t het = ((1u << n) - 1u) & ~same;
    const uint32_t movable = het & ~(uint32_t)sj.flags;  // phase mask: fixed positions never switch
    uint32_t seen = 0;
    int cnt = 0;
    for (uint32_t i = 0; i < (1u << (n - 1)); ++i) {
      uint32_t p = i & movable;
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
  if (lane < P) M.diag[lane] = prior[lane * P + lane];
  if (lane < GRIM_SIDES) M.cnt_side[lane] = 0;
  WAVE_SYNC();
  const int nph = M.nph;
  const int nsides = 2 * nph;
  uint32_t typed = 0;
  for (int l = 0; l < n; ++l) typed |= 1u << sj.slot[l];
  const bool full_nodes = typed == g.full_mask;
  // ---- candidates per side -----------------------------------------------------------------------
  uint32_t my_nc = 0;
  bool bad = false;
  if (lane < nsides) {
    const uint32_t pat = M.ph_pat[lane >> 1];
    uint64_t options = 1;
    my_nc = 1;
    for (int l = 0; l < n; ++l) {
      int c = (int)((pat >> l) & 1u) ^ (lane & 1);
      my_nc *= sj.cnt[l][c];
      options *= (uint64_t)sj.wid[l][c];
      if (options > 0xFFFFFFFFFFFFull) options = 0xFFFFFFFFFFFFull;
    }
    bad = !(options < A.prm.opt_threshold) || my_nc > 4096;
  }
  if (__ballot(bad)) return 2;
  uint32_t inc = wave_incl_scan(my_nc);
  const uint32_t C = __shfl(inc, 63);
  if (C > 4096) return 2;
  if (lane <= nsides && lane < 64) M.cand_start[lane] = (uint16_t)(lane == 0 ? 0 : 0);
  WAVE_SYNC();
  if (lane < nsides) M.cand_start[lane + 1] = (uint16_t)inc;
  if (lane == 0) M.cand_start[0] = 0;
  WAVE_SYNC();
  // ---- look-ups and gathers, all sides together ----------------------------------------------------
  uint32_t E = 0;
  unsigned long long c_probe = 0, c_nbr = 0, c_freq = 0;
  for (uint32_t c0 = 0; c0 < C; c0 += 64) {
    const uint32_t gi = c0 + lane;
    uint32_t node = GRIM_NONE;
    int side = 0;
    if (gi < C) {
      int lo = 0, hi = nsides - 1;
      while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (M.cand_start[mid] <= gi) lo = mid; else hi = mid - 1;
      }
      side = lo;
      uint32_t rem = gi - M.cand_start[side];
      const uint32_t pat = M.ph_pat[side >> 1];
      uint64_t key = 0;
      for (int l = n - 1; l >= 0; --l) {
        int c = (int)((pat >> l) & 1u) ^ (side & 1);
        uint32_t cn = sj.cnt[l][c];
        uint32_t d = rem % cn;
        rem /= cn;
        key |= (uint64_t)(tok[M.toff[l][c] + d] + 1u) << (GRIM_ABITS * sj.slot[l]);
      }
      node = graph_lookup(g, key);
    }
    c_probe += (C - c0) < 64 ? (C - c0) : 64;
    uint32_t cnt = 0;
    if (node != GRIM_NONE) cnt = full_nodes ? 1u : nbr_count(g.a_start, node);
    uint32_t cinc = wave_incl_scan(cnt);
    const uint32_t total = __shfl(cinc, 63);
    if (total == 0) continue;
    M.cstart[lane] = cinc - cnt;
    M.cnode[lane] = node;
    M.cside[lane] = (uint8_t)side;
    if (lane == 0) M.cstart[64] = total;
    WAVE_SYNC();
    for (uint32_t t0 = 0; t0 < total; t0 += 64) {
      const uint32_t t = t0 + lane;
      const bool valid = t < total;
      uint32_t hap = 0;
      int hs = 0;
      uint64_t mask = 0;
      if (valid) {
        int lo = 0, hi = 63;
        while (lo < hi) {
          int mid = (lo + hi + 1) >> 1;
          if (M.cstart[mid] <= t) lo = mid; else hi = mid - 1;
        }
        const uint32_t nd = M.cnode[lo];
        hs = M.cside[lo];
        hap = full_nodes ? nd : g.a_nbr[g.a_start[nd] + (t - M.cstart[lo])];
        for (int j = 0; j < P; ++j)
          if (g.freq[(uint64_t)hap * P + j] > 0.0) mask |= 1ull << j;
      }
      const uint32_t ec = (uint32_t)__popcll(mask);
      uint32_t einc = wave_incl_scan(ec);
      const uint32_t etot = __shfl(einc, 63);
      if (E + etot > MW_E) return 2;
      uint32_t pos = E + einc - ec;
      while (mask) {
        int j = __ffsll((unsigned long long)mask) - 1;
        mask &= mask - 1;
        M.u.e.ent_p[pos] = g.freq[(uint64_t)hap * P + j];
        M.u.e.ent_e[pos] = hap | ((uint32_t)j << 24);
        M.ent_side[pos] = (uint8_t)hs;
        ++pos;
      }
      if (ec) atomicAdd(&M.cnt_side[hs], ec);
      E += etot;
      c_freq += (total - t0) < 64 ? (total - t0) : 64;
    }
    if (!full_nodes) c_nbr += total;
    WAVE_SYNC();
  }
  WAVE_SYNC();
  // ---- segments and per-side lists (convert_list_to_one_dim, impute.py:424-442) ------------------
  const uint32_t K = A.prm.top_n;
  uint32_t sc = lane < nsides ? M.cnt_side[lane] : 0;
  if (__ballot(sc > 64)) return 2;
  uint32_t sinc = wave_incl_scan(sc);
  if (lane < nsides) {
    M.seg[lane] = (uint16_t)(sinc - sc);
    M.tlen[lane] = (uint16_t)(sc < K ? sc : K);
  }
  if (lane == 0) M.seg[nsides] = (uint16_t)E;
  WAVE_SYNC();
  for (uint32_t e = lane; e < E; e += 64) {
    const int s = M.ent_side[e];
    const double p = M.u.e.ent_p[e];
    const uint32_t ent = M.u.e.ent_e[e];
    const double key = p * M.diag[ent >> 24];
    const uint32_t a = M.seg[s], b = M.seg[s + 1];
    uint32_t rank = 0;
    for (uint32_t e2 = a; e2 < b; ++e2) {
      const double k2 = M.u.e.ent_p[e2] * M.diag[M.u.e.ent_e[e2] >> 24];
      rank += (k2 > key || (k2 == key && e2 < e)) ? 1u : 0u;
    }
    if (rank < K) {
      M.T_p[a + rank] = p;
      M.T_e[a + rank] = ent;
    }
  }
  WAVE_SYNC();
  if (lane < nsides) {  // prefix-min for the pair loop's break (impute.py:463-464,545-546)
    const uint32_t a = M.seg[lane];
    double mn = __longlong_as_double(0x7FF0000000000000ll);
    for (uint32_t r = 0; r < M.tlen[lane]; ++r) {
      double v = M.T_p[a + r];
      mn = v < mn ? v : mn;
      M.T_m[a + r] = mn;
    }
  }
  if (lane == 0) {
    uint32_t acc = 0;
    for (int i = 0; i < nph; ++i) {
      M.poff[i] = acc;
      acc += (uint32_t)M.tlen[2 * i] * M.tlen[2 * i + 1];
    }
    for (int i = nph; i <= GRIM_MAXPH; ++i) M.poff[i] = acc;
  }
  WAVE_SYNC();
  const uint32_t np = M.poff[GRIM_MAXPH];
  if (np > MW_NP) return 2;
  // every scored pair a different unordered entity pair (see prepare_lists, grim_plan_a.h): the two lists of every
  // position are the same text or disjoint sets, and some position differs -> the dedup below has nothing to find
  bool nodup = false;
  {
    bool het = false, clash = false;
    if (lane < n && !((sj.pad[0] >> lane) & 1u)) {
      het = true;
      const uint16_t *l0 = tok + M.toff[lane][0], *l1 = tok + M.toff[lane][1];
      const uint32_t c0 = sj.cnt[lane][0], c1 = sj.cnt[lane][1];
      for (uint32_t a = 0; a < c0 && !clash; ++a)
        for (uint32_t b = 0; b < c1 && !clash; ++b) clash = l0[a] == l1[b];
    }
    nodup = __ballot(het) != 0 && __ballot(clash) == 0 && !(A.flags & GRIM_F_NO_NODUP);
  }
  // ---- ladder (impute.py:1665-1687) ----------------------------------------------------------------
  int best = A.prm.n_ladder;
  for (uint32_t f = lane; f < np && best > 0; f += 64) {
    PairRef pr = med_pair(M, f);
    double w = prior[ENT_POP(pr.e1) * P + ENT_POP(pr.e2)];
    for (int idx = 0; idx < best; ++idx)
      if (pair_accept(A.prm.ladder[idx], pr, w)) {
        best = idx;
        break;
      }
  }
  for (int d = 32; d > 0; d >>= 1) {
    int o = __shfl_xor(best, d);
    best = o < best ? o : best;
  }
  uint32_t nU = 0;
  double mx = 0.0;
  if (best < A.prm.n_ladder) {
    double eps = A.prm.ladder[best];
    for (int round = 0; round < 2; ++round) {
      // one pass at eps: keys of accepted pairs, first-wins dedup, MaxProb; the second round (after
      // eps = MaxProb/1e5, impute.py:1685) also lists the winners in sequence order
      const bool emit = (round == 1) || !(eps > 0.0);
      for (uint32_t f = lane; f < np; f += 64) {
        PairRef pr = med_pair(M, f);
        double w = prior[ENT_POP(pr.e1) * P + ENT_POP(pr.e2)];
        uint64_t key = 0;
        if (pair_accept(eps, pr, w)) {
          uint32_t lo = pr.e1 < pr.e2 ? pr.e1 : pr.e2, hi = pr.e1 < pr.e2 ? pr.e2 : pr.e1;
          key = ((uint64_t)lo << 32) | hi | GRIM_VALID;
        }
        M.u.pkeys[f] = key;
      }
      WAVE_SYNC();
      mx = 0.0;
      nU = 0;
      for (uint32_t f0 = 0; f0 < np; f0 += 64) {
        const uint32_t f = f0 + lane;
        bool win = false;
        double prob = 0.0;
        if (f < np) {
          const uint64_t key = M.u.pkeys[f];
          win = key != 0;
          if (!nodup)
            for (uint32_t f2 = 0; f2 < f && win; ++f2) win = M.u.pkeys[f2] != key;
          if (win) {
            PairRef pr = med_pair(M, f);
            prob = pair_prob(pr, prior[ENT_POP(pr.e1) * P + ENT_POP(pr.e2)]);
            mx = prob > mx ? prob : mx;
          }
        }
        const uint64_t m = __ballot(win);
        if (emit && win) {
          uint32_t pos = nU + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          M.useq[pos] = (uint16_t)f;
          if (pos < 64) M.uprob[pos] = prob;
        }
        nU += (uint32_t)__popcll(m);
      }
      for (int d = 32; d > 0; d >>= 1) {
        double o = __shfl_xor(mx, d);
        mx = o > mx ? o : mx;
      }
      if (emit) break;
      eps = mx / 100000.0;
      WAVE_SYNC();
    }
    if (nU > 64) {
      // more pairs than the one-wave shuffle code holds: the tables are the table kernels' work (grim_tables.h).  The
      // accepted pairs become records of the batch's pair pool, in sequence order, and a work item.
      uint32_t off = 0;
      if (lane == 0) {
        off = atomicAdd(A.queue + 8, nU);
        if (off + nU > A.ppool_cap) {
          atomicExch(&A.counters[4], 1ull);
          off = GRIM_NONE;
        }
      }
      off = __shfl(off, 0);
      if (off != GRIM_NONE) {
        for (uint32_t u = lane; u < nU; u += 64) {
          const PairRef pr = med_pair(M, M.useq[u]);
          PairRec r;
          r.k1 = g.node_key[ENT_HAP(pr.e1)];
          r.k2 = g.node_key[ENT_HAP(pr.e2)];
          r.prob = pair_prob(pr, prior[ENT_POP(pr.e1) * P + ENT_POP(pr.e2)]);
          r.e1 = pr.e1;
          r.e2 = pr.e2;
          A.ppool[off + u] = r;
        }
        if (lane == 0) {
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
      }
      grim_subject_result out;
      memset(&out, 0, sizeof(out));
      out.plan = 'a';
      out.n_pairs = nU;
      out.max_prob = mx;
      out.status = GRIM_ST_OK;
      if (lane == 0) A.res[si] = out;
      acc[0] += c_probe;
      acc[1] += c_nbr;
      acc[2] += c_freq;
      return 0;
    }
  }
  WAVE_SYNC();
  // ---- result ----------------------------------------------------------------------------------------
  grim_subject_result out;
  memset(&out, 0, sizeof(out));
  out.plan = 'a';
  if (nU > 0) {
    uint32_t e1 = 0, e2 = 0;
    double prob = 0.0;
    if (lane < (int)nU) {
      PairRef pr = med_pair(M, M.useq[lane]);
      e1 = pr.e1;
      e2 = pr.e2;
      prob = M.uprob[lane];
    }
    out.n_pairs = nU;
    out.max_prob = mx;
    out.status = GRIM_ST_OK;
    emit_small_core(A, nU, e1, e2, prob, lane < (int)nU ? g.node_key[ENT_HAP(e1)] : 0, lane < (int)nU ? g.node_key[ENT_HAP(e2)] : 0, out, rb);
  } else if (A.prm.planb) {
    out.status = GRIM_ST_UNSUPPORTED;  // replaced by the plan-B kernel's verdict
    out.reason = 2;
    if (lane == 0) push_next(A, si, n <= GRIM_HEAVY_LOCI);
  } else {
    out.status = GRIM_ST_MISS;
  }
  if (lane == 0) A.res[si] = out;
  acc[0] += c_probe;
  acc[1] += c_nbr;
  acc[2] += c_freq;
  return 0;
}

// one wave = one subject at a time; waves of a workgroup are independent (no __syncthreads)
__global__ __launch_bounds__(64) void grim_plan_a_medium_kernel(DevArgs A, const uint32_t *order, uint32_t n,
                                                                    uint32_t *bail_list) {
  __shared__ WaveMed M;
  // Work is taken eight subjects at a time: one atomic on a single address costs ~50 ns at the L2 whoever
  // issues it, so a counter bumped once per subject caps the whole kernel at 20 M subjects/s (it did: 4.7 ms
  // per 92 000 subjects whatever the occupancy).  Same for the statistics: one flush per wave, not per subject.
  constexpr uint32_t CH = 8;
  unsigned long long acc[3] = {0, 0, 0};
  RowBlock rb = {0, 0, GRIM_ROW_GRAB};
  for (;;) {
    uint32_t w0 = 0;
    if (lane_id() == 0) w0 = atomicAdd(A.queue + 4, CH);
    w0 = __shfl(w0, 0);
    if (w0 >= n) break;
    const uint32_t w1 = w0 + CH < n ? w0 + CH : n;
    for (uint32_t w = w0; w < w1; ++w) {
      const uint32_t si = order[w];
      const int rc = medium_subject(A, M, si, acc, rb);
      if (rc != 0 && lane_id() == 0) {
        // the heavier hand-overs from the back of the list: the general kernel takes them first
        if (rc == 2)
          bail_list[n - 1u - atomicAdd(A.queue + 7, 1u)] = si;
        else
          bail_list[atomicAdd(A.queue + 5, 1u)] = si;
      }
      WAVE_SYNC();
    }
  }
  if (lane_id() == 0) {
    unsigned long long *c = A.counters + 8 + 4 * (blockIdx.x & 63);
    atomicAdd(&c[0], acc[0]);
    atomicAdd(&c[1], acc[1]);
    atomicAdd(&c[2], acc[2]);
  }
}
