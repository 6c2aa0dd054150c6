// grim_plan_b.h -- Plan B on device: locus-partition fallback for subjects whose Plan-A pass found
// no haplotype pair.  Restates (impute.py lines in brackets):
//   comp_phase_prob_plan_b            [1392-1570]  row loop with per-side "best row" memo + rescue loop
//   comp_hap_prob_plan_b / find_option_freq / create_haplos_string / get_haplo_freqs_pan_b
//                                      [1117-1123, 1072-1115, 1015-1039, 1207-1216]  block look-ups
//   open_option_                       [1041-1069]  block cross product, freq = (f1*f2)*1e-4 per population
//   find_option_freq_missing_data      [1142-1172]  alleles the graph has never seen
//   check_if_alleles_in_data / _of_one_phase_in_data [1224-1258]
//   call_comp_phase_prob levels        [1696-1722]  level 0 = subject prior, level 1 = all-ones prior
// and Graph.adjs_query_by_color / node_probs / haps_with_probs_by_label (networkx_graph.py:238-321).
//
// One workgroup per subject; each wave builds phase sides (block sets -> cross product -> running
// top-K), the workgroup then scores pairs at epsilon 0 (the only value Plan B ever runs with,
// impute.py:1703-1711).  Haplotypes made of blocks are not graph nodes, so every haplotype is
// identified by its 60-bit allele key through the slot's canonical table.
//
// Plan C (independent loci) follows in the same kernel when both Plan-B levels come back empty.
// Not on device (reported GRIM_ST_UNSUPPORTED, never computed elsewhere): Plan B/C for subjects
// whose sides use the label-scan opening (reason 3).
#pragma once
#include "grim_plan_a.h"

#define GRIM_FACTOR_JOIN 0.0001  // impute.py:196

struct SideSpec {
  uint32_t cn[GRIM_MAXL], to[GRIM_MAXL], sl[GRIM_MAXL];
  // bsl[l]: the locus the REFERENCE takes position l for when it cuts a cartesian candidate into Plan-B blocks: the l-th
  // smallest typed index (create_haplos_string counts places by locus index, impute.py:1015-1039) -- the candidate's alleles,
  // though, are in sorted string order (impute.py:271).  The same as sl[l] unless the loci_map is not alphabetical.
  uint32_t bsl[GRIM_MAXL];
  int n;
  uint32_t typed_mask, ncand;
  bool expansion;
};

__device__ __forceinline__ SideSpec side_spec(const DevArgs &A, const WgShared &sh, int ph, int side) {
  const grim_subject &sj = sh.subj;
  SideSpec sp;
  sp.n = sj.n_loci;
  sp.typed_mask = 0;
  sp.ncand = 1;
  uint64_t options = 1;
  const uint32_t pat = sh.ph_pat[ph];
#pragma unroll
  for (int l = 0; l < GRIM_MAXL; ++l) {
    sp.cn[l] = 1; sp.to[l] = 0; sp.sl[l] = 0;
    if (l < sp.n) {
      int c = (int)((pat >> l) & 1u) ^ side;
      const ListVer lv = sh.lv[l][c][sh.side_ver[2 * ph + side]];
      sp.cn[l] = lv.cnt;
      sp.to[l] = lv.off;
      sp.sl[l] = sj.slot[l];
      options *= (uint64_t)lv.wid;
      if (options > 0xFFFFFFFFFFFFull) options = 0xFFFFFFFFFFFFull;
      sp.ncand *= sp.cn[l];
      sp.typed_mask |= 1u << sp.sl[l];
    }
  }
  if (sh.reduced) {  // Plan C's reduction: one allele per list wherever the graph knows any
    options = 1;
    sp.ncand = 1;
#pragma unroll
    for (int l = 0; l < GRIM_MAXL; ++l) {
      if (l < sp.n) {
        const uint16_t b = sh.bestc[2 * ph + side][l];
        if (b != 0xFFFF) {
          sp.to[l] += b;
          sp.cn[l] = 1;
        } else {
          int c = (int)((pat >> l) & 1u) ^ side;
          options *= (uint64_t)sh.lv[l][c][sh.side_ver[2 * ph + side]].wid;
          if (options > 0xFFFFFFFFFFFFull) options = 0xFFFFFFFFFFFFull;
        }
        sp.ncand *= sp.cn[l];
      }
    }
  }
  sp.expansion = options < A.prm.opt_threshold;
#pragma unroll
  for (int l = 0; l < GRIM_MAXL; ++l) sp.bsl[l] = sp.sl[l];
  if (A.g.order_bad) {
#pragma unroll
    for (int l = 0; l < GRIM_MAXL; ++l)
      if (l < sp.n) {
        int rank = 0;
#pragma unroll
        for (int l2 = 0; l2 < GRIM_MAXL; ++l2)
          if (l2 < sp.n && sp.sl[l2] < sp.sl[l]) ++rank;
#pragma unroll
        for (int r = 0; r < GRIM_MAXL; ++r)
          if (r == rank) sp.bsl[r] = sp.sl[l];
      }
  }
  return sp;
}

__device__ __forceinline__ bool allele_known(const DevGraph &g, uint32_t slot, uint32_t id) {
  return graph_lookup(g, (uint64_t)(id + 1u) << (GRIM_ABITS * slot)) != GRIM_NONE;
}

// positions of this phase side none of whose alleles the graph knows (impute.py:1243-1258)
__device__ __forceinline__ uint32_t absent_positions(const DevArgs &A, const uint16_t *tok, const SideSpec &sp) {
  uint32_t m = 0;
#pragma unroll
  for (int l = 0; l < GRIM_MAXL; ++l) {
    if (l < sp.n) {
      bool any = false;
      for (uint32_t t = 0; t < sp.cn[l]; ++t) any |= allele_known(A.g, sp.sl[l], tok[sp.to[l] + t]);
      if (!any) m |= 1u << l;
    }
  }
  return m;
}

// label-scan opening (impute.py:947-981): does node `nd` of the typed-loci label use only alleles of
// this side's alternatives?
__device__ __forceinline__ bool node_passes(const DevGraph &g, const SideSpec &sp, const uint16_t *tok, uint32_t nd) {
  const uint64_t key = g.node_key[nd];
  bool ok = true;
#pragma unroll
  for (int l = 0; l < GRIM_MAXL; ++l) {
    if (l < sp.n && ok) {
      const uint32_t al = (uint32_t)((key >> (GRIM_ABITS * sp.sl[l])) & 0xFFF) - 1u;
      bool hit = false;
      for (uint32_t t = 0; t < sp.cn[l]; ++t) hit |= (tok[sp.to[l] + t] == al);
      ok = hit;
    }
  }
  return ok;
}

// number of candidates the label scan yields for this side (wave-uniform)
__device__ inline uint32_t scan_count(const DevArgs &A, const SideSpec &sp, const uint16_t *tok) {
  const DevGraph &g = A.g;
  const uint32_t a = g.lab_start[sp.typed_mask], b = g.lab_start[sp.typed_mask + 1];
  uint32_t n = 0;
  for (uint32_t i0 = a; i0 < b; i0 += 64) {
    const uint32_t i = i0 + lane_id();
    const bool ok = i < b && node_passes(g, sp, tok, g.lab_nodes[i]);
    n += (uint32_t)__popcll(__ballot(ok));
  }
  return n;
}

// Row whose first block is the full label: plain Plan-A look-up (impute.py:1118-1119).
__device__ inline bool side_lookup_full(const DevArgs &A, WgShared &sh, const Slot &S, const double *prior, WaveTop &L,
                                        const SideSpec &sp, const uint16_t *tok, int row) {
  const DevGraph &g = A.g;
  const int lane = lane_id();
  TopState st;
  st.nrun = 0; st.nbuf = 0; st.K = (int)A.prm.top_n; st.full = false; st.ge = false; st.thr = 0;
  uint64_t item_base = 0, c_nbr = 0, c_freq = 0;
  const bool direct = (sp.typed_mask == g.full_mask);
  // the haplotypes of this list are graph node NAMES (adjs_query's keys), spelled in the graph's locus order; the keys the
  // block joins make are sorted (open_option_) -- different strings, hence different haplotypes, when the two orders differ
  const uint64_t name_flag = g.order_bad ? (1ull << GRIM_KEY_GRAPH_ORDER) : 0ull;
  if (!sp.expansion) {
    // candidates = the nodes the label scan lets through, in node order
    const uint32_t a = g.lab_start[sp.typed_mask], b = g.lab_start[sp.typed_mask + 1];
    for (uint32_t i0 = a; i0 < b; i0 += 64) {
      const uint32_t i = i0 + lane;
      uint32_t node = GRIM_NONE;
      if (i < b) {
        const uint32_t nd = g.lab_nodes[i];
        if (node_passes(g, sp, tok, nd)) node = nd;
      }
      expand_chunk<true>(A, prior, L, st, node, direct, g.a_start, g.a_nbr, 1.0, name_flag, item_base, c_nbr, c_freq);
    }
    store_top<true>(S, sh, L, st, row);
    return st.nrun > 0;
  }
  for (uint32_t c0 = 0; c0 < sp.ncand; c0 += 64) {
    uint32_t c = c0 + lane;
    uint32_t node = GRIM_NONE;
    if (c < sp.ncand) {
      uint64_t key = 0;
      uint32_t rem = c;
#pragma unroll
      for (int l = GRIM_MAXL - 1; l >= 0; --l) {
        if (l < sp.n) {
          uint32_t d = rem % sp.cn[l];
          rem /= sp.cn[l];
          key |= (uint64_t)(tok[sp.to[l] + d] + 1u) << (GRIM_ABITS * sp.sl[l]);
        }
      }
      node = graph_lookup_subject(g, key, sp.typed_mask);
    }
    expand_chunk<true>(A, prior, L, st, node, direct, g.a_start, g.a_nbr, 1.0, name_flag, item_base, c_nbr, c_freq);
  }
  store_top<true>(S, sh, L, st, row);
  return st.nrun > 0;
}

// Alleles the graph has never seen at the positions in `absent` (bit l = position l): look the rest
// of the haplotype up towards the label "all loci but the absent ones", splice the unseen alleles
// back in, scale by factor_missing_data ** (#absent loci)  (impute.py:1142-1172).
__device__ inline bool side_absent(const DevArgs &A, WgShared &sh, const Slot &S, const double *prior, WaveTop &L,
                                   const SideSpec &sp, const uint16_t *tok, uint32_t absent, int row) {
  const DevGraph &g = A.g;
  const int lane = lane_id();
  TopState st;
  st.nrun = 0; st.nbuf = 0; st.K = (int)A.prm.top_n; st.full = false; st.ge = false; st.thr = 0;
  uint32_t src_mask = 0, abs_mask = 0;
#pragma unroll
  for (int l = 0; l < GRIM_MAXL; ++l)
    if (l < sp.n) {
      if ((absent >> l) & 1u) abs_mask |= 1u << sp.sl[l]; else src_mask |= 1u << sp.sl[l];
    }
  const uint32_t target = g.full_mask & ~abs_mask;
  const uint32_t added = target & ~src_mask;
  const int nadd = __popc(added);
  const double scale = A.prm.factor_missing_pow[__popc(abs_mask)];
  uint64_t item_base = 0, c_nbr = 0, c_freq = 0;
  if (sp.expansion && src_mask != 0 && nadd <= 1) {
    const bool direct = (nadd == 0);
    const int add_slot = nadd ? (__ffs(added) - 1) : 0;
    for (uint32_t c0 = 0; c0 < sp.ncand; c0 += 64) {
      uint32_t c = c0 + lane;
      uint32_t src = GRIM_NONE;
      uint64_t outside = 0;
      if (c < sp.ncand) {
        uint64_t key = 0;
        uint32_t rem = c;
#pragma unroll
        for (int l = GRIM_MAXL - 1; l >= 0; --l) {
          if (l < sp.n) {
            uint32_t d = rem % sp.cn[l];
            rem /= sp.cn[l];
            uint64_t a = (uint64_t)(tok[sp.to[l] + d] + 1u) << (GRIM_ABITS * sp.sl[l]);
            if ((absent >> l) & 1u) outside |= a; else key |= a;
          }
        }
        uint32_t node = graph_lookup_subject(g, key, src_mask);
        if (node != GRIM_NONE) src = direct ? node : g.b_conn[(uint64_t)node * GRIM_MAXL + add_slot];
      }
      expand_chunk<true>(A, prior, L, st, src, direct, g.b_start, g.b_nbr, scale, outside, item_base, c_nbr, c_freq);
    }
  }
  store_top<true>(S, sh, L, st, row);
  return st.nrun > 0;
}


// ---- save_space_mode (impute.py:1048-1059) -----------------------------------------------------------------------------
// Before two dicts are joined, either of them that holds more than 10 entries is cut down to 10: the entries are sorted
// by the SUM of their frequency vector, ascending and stable, and deleted from the front until ten are left -- so the ten
// largest stay, of equal sums the later ones, in their original order.  save_select finds them among n entries whose value
// any lane can compute; every lane ends with the same list (ascending entry numbers).
template <typename F>
__device__ inline uint32_t save_select(uint32_t n, F val, uint32_t (&keep)[GRIM_SAVE_KEEP]) {
  const int lane = lane_id();
  if (n <= GRIM_SAVE_KEEP) {
    for (uint32_t k = 0; k < GRIM_SAVE_KEEP; ++k) keep[k] = k;
    return n;
  }
  // pick after pick in the order "larger sum first, of equal sums the later entry first"
  double pv = 0.0;
  uint32_t pi = 0;
  for (uint32_t r = 0; r < GRIM_SAVE_KEEP; ++r) {
    double bv = 0.0;
    uint32_t bi = GRIM_NONE;
    for (uint32_t i = lane; i < n; i += 64) {
      const double v = val(i);
      const bool after_prev = r == 0 || v < pv || (v == pv && i < pi);        // not picked yet
      const bool better = bi == GRIM_NONE || v > bv || (v == bv && i > bi);
      if (after_prev && better) {
        bv = v;
        bi = i;
      }
    }
    for (int d = 32; d > 0; d >>= 1) {
      const double ov = __shfl_xor(bv, d);
      const uint32_t oi = (uint32_t)__shfl_xor((int)bi, d);
      if (oi != GRIM_NONE && (bi == GRIM_NONE || ov > bv || (ov == bv && oi > bi))) {
        bv = ov;
        bi = oi;
      }
    }
    pv = bv;
    pi = bi;
    keep[r] = bi;
  }
  // back into the dict's own order
  for (int a = 1; a < GRIM_SAVE_KEEP; ++a) {
    const uint32_t x = keep[a];
    int b = a;
    while (b > 0 && keep[b - 1] > x) {
      keep[b] = keep[b - 1];
      --b;
    }
    keep[b] = x;
  }
  return GRIM_SAVE_KEEP;
}

// sum(list) of a node's frequency vector as Python adds it: 0 + f[0] + f[1] + ...
__device__ __forceinline__ double freq_sum(const DevGraph &g, uint32_t node) {
  double s = 0.0;
  for (uint32_t j = 0; j < g.P; ++j) s = s + g.freq[(uint64_t)node * g.P + j];
  return s;
}

// The block cross product of a Plan_B_Matrix row under save_space_mode: acc = block 0; for every further block: acc cut to
// 10, the block cut to 10, acc = their join (acc outer, per population (a * b) * 1e-4, all-zero combinations dropped)
// (find_option_freq + open_option_, impute.py:1072-1115, 1041-1069).  The entries of the last join go through the top-K.
__device__ inline void blocks_save(const DevArgs &A, const Slot &S, const double *prior, WaveTop &L, TopState &st, const uint32_t *const *setp,
                                   const uint32_t *setn, int nb) {
  const DevGraph &g = A.g;
  const int lane = lane_id();
  const int P = g.P;
  const uint64_t per = (uint64_t)GRIM_SAVE_CAP * (P + 1);
  double *buf[2] = {S.save + (uint64_t)wave_id() * 2 * per, S.save + (uint64_t)wave_id() * 2 * per + per};
  // an entry of a buffer: P frequencies, then its 60-bit key (as a double-sized word)
  auto key_of = [&](double *b, uint32_t e) -> uint64_t * { return (uint64_t *)(b + (uint64_t)e * (P + 1) + P); };
  auto vec_of = [&](double *b, uint32_t e) -> double * { return b + (uint64_t)e * (P + 1); };
  uint32_t keep[GRIM_SAVE_KEEP];
  // acc := block 0, cut to ten (the cut happens when block 1 is joined: the same thing, nb >= 2)
  uint32_t na = save_select(setn[0], [&](uint32_t i) { return freq_sum(g, setp[0][i]); }, keep);
  int cur = 0;
  for (uint32_t e = lane; e < na; e += 64) {
    const uint32_t nd = setp[0][keep[e]];
    for (int j = 0; j < P; ++j) vec_of(buf[cur], e)[j] = g.freq[(uint64_t)nd * P + j];
    *key_of(buf[cur], e) = g.node_key[nd];
  }
  __threadfence_block();
  for (int b = 1; b < nb; ++b) {
    if (na > GRIM_SAVE_KEEP) {  // the join so far, cut to ten
      double *src = buf[cur];
      uint32_t k2[GRIM_SAVE_KEEP];
      const uint32_t n2 = save_select(na, [&](uint32_t i) {
        double s = 0.0;
        for (int j = 0; j < P; ++j) s = s + vec_of(src, i)[j];
        return s;
      }, k2);
      double *dst = buf[cur ^ 1];
      for (uint32_t e = lane; e < n2; e += 64) {
        for (int j = 0; j < P; ++j) vec_of(dst, e)[j] = vec_of(src, k2[e])[j];
        *key_of(dst, e) = *key_of(src, k2[e]);
      }
      __threadfence_block();
      cur ^= 1;
      na = n2;
    }
    const uint32_t nk = save_select(setn[b], [&](uint32_t i) { return freq_sum(g, setp[b][i]); }, keep);
    double *src = buf[cur], *dst = buf[cur ^ 1];
    uint32_t cnt = 0;
    const uint32_t total = na * nk;  // <= 100
    for (uint32_t c0 = 0; c0 < total; c0 += 64) {
      const uint32_t c = c0 + lane;
      bool on = false;
      uint32_t e1 = 0, nd = 0;
      if (c < total) {
        e1 = c / nk;
        nd = setp[b][keep[c - e1 * nk]];
        for (int j = 0; j < P; ++j) on |= vec_of(src, e1)[j] * g.freq[(uint64_t)nd * P + j] * GRIM_FACTOR_JOIN > 0.0;  // max(list_prob) > 0
      }
      const uint64_t m = __ballot(on);
      if (on) {
        const uint32_t pos = cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        for (int j = 0; j < P; ++j) vec_of(dst, pos)[j] = vec_of(src, e1)[j] * g.freq[(uint64_t)nd * P + j] * GRIM_FACTOR_JOIN;
        *key_of(dst, pos) = *key_of(src, e1) | g.node_key[nd];
      }
      cnt += (uint32_t)__popcll(m);
    }
    __threadfence_block();
    cur ^= 1;
    na = cnt;
    if (na == 0) return;
  }
  double *fin = buf[cur];
  for (uint32_t c0 = 0; c0 < na; c0 += 64) {
    const uint32_t c = c0 + lane;
    const bool valid = c < na;
    const uint64_t key = valid ? *key_of(fin, c) : 0ull;
    for (int j = 0; j < P; ++j) {
      const double p = valid ? vec_of(fin, c)[j] : 0.0;
      const uint64_t tie = (((uint64_t)c * (uint64_t)P + (uint64_t)j) << 8) | (uint64_t)j;
      top_push(L, st, valid && p > 0.0, p, p * prior[j * P + j], tie, 0, key);
    }
  }
}

// A Plan_B_Matrix row with several blocks (impute.py:1072-1115): per block the set of graph nodes
// reachable from the typed part of the candidates (or, for a block without any typed locus that is
// not the first one, every node of the block's label), then the cross product, first block most
// significant, per-population frequency ((f0*f1)*1e-4)*f2)*1e-4 ...
__device__ inline bool side_blocks(const DevArgs &A, WgShared &sh, const Slot &S, const double *prior, WaveTop &L,
                                   const SideSpec &sp, const uint16_t *tok, int mrow, int row) {
  const DevGraph &g = A.g;
  const int lane = lane_id();
  const int P = g.P;
  const int nb = A.prm.planb_nblk[mrow];
  TopState st;
  st.nrun = 0; st.nbuf = 0; st.K = (int)A.prm.top_n; st.full = false; st.ge = false; st.thr = 0;
  uint32_t *wset = S.bset + (uint64_t)wave_id() * GRIM_MAXL * A.bset_cap;
  const uint32_t *setp[GRIM_MAXL];
  uint32_t setn[GRIM_MAXL];
  bool ok = true;
#pragma unroll
  for (int b = 0; b < GRIM_MAXL; ++b) {
    setp[b] = nullptr;
    setn[b] = 1;
    if (b < nb && ok) {
      const uint32_t bm = A.prm.planb_blk[mrow][b];
      const uint32_t tb = bm & sp.typed_mask;
      if (tb == 0) {
        if (b == 0) {
          ok = false;  // an empty first block yields nothing (impute.py:1079-1085)
        } else {
          setp[b] = g.lab_nodes + g.lab_start[bm];  // haps_with_probs_by_label (impute.py:1100-1106)
          setn[b] = g.lab_start[bm + 1] - g.lab_start[bm];
          if (setn[b] == 0) ok = false;
        }
      } else {
        // The positions of a cartesian candidate that make up this block's name, and the loci their alleles really belong
        // to: the same thing (the block's typed loci) unless the loci_map is not alphabetical -- then the reference picks
        // positions by index rank (sp.bsl) out of a candidate that is in sorted string order, the name it builds holds the
        // alleles of the loci `am`, is a graph name only when those come in graph order, and reaches the block's label
        // either as a node of that label (am == bm) or over a connector to it (bm = am + one locus).
        uint32_t selpos = 0, am = 0;
#pragma unroll
        for (int l = 0; l < GRIM_MAXL; ++l)
          if (l < sp.n && ((bm >> (sp.expansion ? sp.bsl[l] : sp.sl[l])) & 1u)) {
            selpos |= 1u << l;
            am |= 1u << sp.sl[l];
          }
        const bool findable = !sp.expansion || (subject_order_ok(g, am) && (am & ~bm) == 0);
        const uint32_t added = bm & ~am;
        const int nadd = __popc(added);
        if (nadd > 1 || !findable) {
          ok = false;  // parents are exactly one locus larger: no connector exists
        } else {
          const int add_slot = nadd ? (__ffs(added) - 1) : 0;
          uint32_t nsub = 1;
#pragma unroll
          for (int l = 0; l < GRIM_MAXL; ++l)
            if (l < sp.n && ((selpos >> l) & 1u)) nsub *= sp.cn[l];
          uint32_t *out = wset + (uint64_t)b * A.bset_cap;
          uint32_t cnt = 0;
          if (!sp.expansion) {
            // candidates come from the label scan: their projections onto the block's typed loci,
            // first occurrences in scan order (create_haplos_string + dict order, impute.py:1015-1039,
            // networkx_graph.py:289-306).  Pass 1 records the first scan position of every projection
            // in a per-wave hash set, pass 2 emits the first occurrences in order.
            uint64_t tbmask = 0;
#pragma unroll
            for (int l = 0; l < GRIM_MAXL; ++l)
              if (l < sp.n && ((tb >> sp.sl[l]) & 1u)) tbmask |= 0xFFFull << (GRIM_ABITS * sp.sl[l]);
            uint64_t *pk = S.proj_k + (uint64_t)wave_id() * A.proj_cap;
            uint32_t *pp = S.proj_p + (uint64_t)wave_id() * A.proj_cap;
            const uint32_t la = g.lab_start[sp.typed_mask], lb = g.lab_start[sp.typed_mask + 1];
            uint32_t pcap = 64;
            while (pcap < 2 * (lb - la) && pcap < A.proj_cap) pcap <<= 1;
            for (uint32_t q = lane; q < pcap; q += 64) {
              pk[q] = 0;
              pp[q] = GRIM_NONE;
            }
            __threadfence_block();
            for (int pass = 0; pass < 2; ++pass) {
              for (uint32_t i0 = la; i0 < lb; i0 += 64) {
                const uint32_t i = i0 + lane;
                uint32_t n_out = 0, node = GRIM_NONE, base = 0;
                if (i < lb) {
                  const uint32_t nd = g.lab_nodes[i];
                  if (node_passes(g, sp, tok, nd)) {
                    const uint64_t key = g.node_key[nd] & tbmask;
                    const uint32_t slot = tab_insert(pk, pcap - 1, key | GRIM_VALID);
                    if (pass == 0) {
                      atomicMin(&pp[slot], i);
                    } else if (ALOAD(&pp[slot]) == i) {
                      node = graph_lookup(g, key);
                      if (node != GRIM_NONE) {
                        if (nadd == 0) {
                          n_out = 1;
                        } else {
                          uint32_t conn = g.b_conn[(uint64_t)node * GRIM_MAXL + add_slot];
                          if (conn != GRIM_NONE) {
                            n_out = nbr_count(g.b_start, conn);
                            base = g.b_start[conn];
                          }
                        }
                      }
                    }
                  }
                }
                if (pass == 1) {
                  uint32_t inc = wave_incl_scan(n_out);
                  uint32_t tot = __shfl(inc, 63);
                  uint32_t off = cnt + inc - n_out;
                  if (off + n_out <= A.bset_cap) {
                    if (nadd == 0) {
                      if (n_out) out[off] = node;
                    } else {
                      for (uint32_t t = 0; t < n_out; ++t) out[off + t] = g.b_nbr[base + t];
                    }
                  }
                  cnt += tot;
                }
              }
              __threadfence_block();
            }
            nsub = 0;  // skip the cartesian enumeration below
          }
          for (uint32_t c0 = 0; c0 < nsub; c0 += 64) {
            uint32_t c = c0 + lane;
            uint32_t n_out = 0, node = GRIM_NONE, base = 0;
            if (c < nsub) {
              uint64_t key = 0;
              uint32_t rem = c;
#pragma unroll
              for (int l = GRIM_MAXL - 1; l >= 0; --l) {
                if (l < sp.n && ((selpos >> l) & 1u)) {
                  uint32_t d = rem % sp.cn[l];
                  rem /= sp.cn[l];
                  key |= (uint64_t)(tok[sp.to[l] + d] + 1u) << (GRIM_ABITS * sp.sl[l]);
                }
              }
              node = graph_lookup(g, key);
              if (node != GRIM_NONE) {
                if (nadd == 0) {
                  n_out = 1;
                } else {
                  uint32_t conn = g.b_conn[(uint64_t)node * GRIM_MAXL + add_slot];
                  if (conn != GRIM_NONE) {
                    n_out = nbr_count(g.b_start, conn);
                    base = g.b_start[conn];
                  }
                }
              }
            }
            uint32_t inc = wave_incl_scan(n_out);
            uint32_t tot = __shfl(inc, 63);
            uint32_t off = cnt + inc - n_out;
            if (off + n_out <= A.bset_cap) {
              if (nadd == 0) {
                if (n_out) out[off] = node;
              } else {
                for (uint32_t t = 0; t < n_out; ++t) out[off + t] = g.b_nbr[base + t];
              }
            }
            cnt += tot;
          }
          if (cnt > A.bset_cap) cnt = A.bset_cap;  // cannot happen: parents of distinct children are disjoint
          setp[b] = out;
          setn[b] = cnt;
          if (cnt == 0) ok = false;
        }
      }
    }
  }
  __threadfence_block();
  if (ok && A.prm.save_mode && nb >= 2) {
    blocks_save(A, S, prior, L, st, setp, setn, nb);
  } else if (ok) {
    uint64_t total = 1;
#pragma unroll
    for (int b = 0; b < GRIM_MAXL; ++b)
      if (b < nb) total *= setn[b];
    for (uint64_t c0 = 0; c0 < total; c0 += 64) {
      uint64_t c = c0 + lane;
      bool valid = c < total;
      uint32_t nd[GRIM_MAXL];
      uint64_t key = 0;
      uint64_t rem = valid ? c : 0;
#pragma unroll
      for (int b = GRIM_MAXL - 1; b >= 0; --b) {
        nd[b] = 0;
        if (b < nb) {
          uint64_t d = rem % setn[b];
          rem /= setn[b];
          nd[b] = setp[b][d];
          key |= g.node_key[nd[b]];
        }
      }
      // four populations per step, every block's frequency loaded before the first product: top_push's LDS fences would
      // otherwise put a memory round trip between one population and the next (the cross product is the bulk of Plan B)
      for (int j0 = 0; j0 < P; j0 += 4) {
        double fv[GRIM_MAXL][4];
#pragma unroll
        for (int b = 0; b < GRIM_MAXL; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) fv[b][q] = (b < nb && j0 + q < P) ? g.freq[(uint64_t)nd[b] * P + j0 + q] : 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = j0 + q;
          if (j >= P) break;
          double acc = fv[0][q];
#pragma unroll
          for (int b = 1; b < GRIM_MAXL; ++b)
            if (b < nb) acc = acc * fv[b][q] * GRIM_FACTOR_JOIN;
          bool act = valid && acc > 0.0;
          uint64_t tie = ((c * (uint64_t)P + (uint64_t)j) << 8) | (uint64_t)j;
          // (a row of ONE block is answered by one look-up: graph names, not joined keys)
          top_push(L, st, act, acc, acc * prior[j * P + j], tie, 0, (nb == 1 && g.order_bad) ? key | (1ull << GRIM_KEY_GRAPH_ORDER) : key);
        }
      }
    }
  }
  store_top<true>(S, sh, L, st, row);
  return st.nrun > 0;
}

__device__ __forceinline__ bool side_row(const DevArgs &A, WgShared &sh, const Slot &S, const double *prior, WaveTop &L,
                                         const SideSpec &sp, const uint16_t *tok, int mrow, int row) {
  if (A.prm.planb_blk[mrow][0] == A.g.full_mask) return side_lookup_full(A, sh, S, prior, L, sp, tok, row);
  return side_blocks(A, sh, S, prior, L, sp, tok, mrow, row);
}

// ---- Plan C (impute.py:1313-1389, 1264-1311): loci treated as independent ------------------------
// reduce_phase_to_commons_alleles(.., commons_number=1, planc=True) (impute.py:881-912) first
// replaces every '/'-list that has at least one allele known to the graph by its single most
// frequent allele (frequency summed over populations under the all-ones prior, first wins ties);
// best[l][c] = token index of that allele inside list (l,c), 0xFFFF if none is known.
__device__ __forceinline__ double pop_sum(const DevGraph &g, uint32_t node) {  // allel_to_SR, impute.py:1260-1262
  double s = 0.0;
  for (uint32_t j = 0; j < g.P; ++j) s = s + g.freq[(uint64_t)node * g.P + j];
  return s;
}

// comp_hap_prob_plan_c: per candidate the product of its alleles' population-summed frequencies
// ((s0*s1)*1e-4)*s2)*1e-4.., unknown alleles kept in the name and paid for with
// factor_missing_data ** count, then joined with every node of the label of the untyped loci.
// one cartesian candidate of a Plan-C side: the product of its alleles' population-summed frequencies (impute.py:1264-1311);
// false: the candidate yields nothing (no allele of it known to the graph, or the product fell to zero)
__device__ __forceinline__ bool plan_c_candidate(const DevArgs &A, const SideSpec &sp, const uint16_t *tok, uint32_t c, double &value, uint64_t &key) {
  const DevGraph &g = A.g;
  uint32_t rem = c;
  uint32_t al[GRIM_MAXL];
#pragma unroll
  for (int l = GRIM_MAXL - 1; l >= 0; --l) {
    al[l] = 0;
    if (l < sp.n) {
      uint32_t d = rem % sp.cn[l];
      rem /= sp.cn[l];
      al[l] = tok[sp.to[l] + d];
    }
  }
  bool have = false, dead = false;
  int n_abs = 0;
  double cur = 0.0;
  key = 0;
#pragma unroll
  for (int l = 0; l < GRIM_MAXL; ++l) {
    if (l < sp.n && !dead) {
      uint64_t k1 = (uint64_t)(al[l] + 1u) << (GRIM_ABITS * sp.sl[l]);
      key |= k1;
      uint32_t node = graph_lookup(g, k1);
      if (node == GRIM_NONE) {
        ++n_abs;
      } else {
        double s = pop_sum(g, node);
        if (!have) {
          cur = s;
          have = true;
        } else {
          cur = cur * s * GRIM_FACTOR_JOIN;
          if (!(cur > 0.0)) dead = true;
        }
      }
    }
  }
  if (!have || dead) return false;
  if (n_abs) cur = cur * A.prm.factor_missing_pow[n_abs];
  value = cur;
  return true;
}

__device__ inline bool side_plan_c(const DevArgs &A, WgShared &sh, const Slot &S, const double *prior, WaveTop &L,
                                   const SideSpec &sp, const uint16_t *tok, int row) {
  const DevGraph &g = A.g;
  const int lane = lane_id();
  const int P = g.P;
  TopState st;
  st.nrun = 0; st.nbuf = 0; st.K = (int)A.prm.top_n; st.full = false; st.ge = false; st.thr = 0;
  const uint32_t miss = g.full_mask & ~sp.typed_mask;
  const uint32_t r0 = miss ? g.lab_start[miss] : 0, rn = miss ? (g.lab_start[miss + 1] - g.lab_start[miss]) : 1;
  if (A.prm.save_mode && miss && rn > 0) {
    // save_space_mode: the join with the nodes of the untyped loci's label goes through open_option_ like every join, so
    // the candidates' dict and the label's dict are both cut to their ten largest entries first (impute.py:1048-1059; a
    // candidate that yields nothing is not in the dict)
    uint32_t kc[GRIM_SAVE_KEEP], kr[GRIM_SAVE_KEEP];
    // (entries = the candidates that yield something, in candidate order: number them first)
    uint32_t n_ok = 0;
    uint32_t *okidx = S.bset + (uint64_t)wave_id() * GRIM_MAXL * A.bset_cap;  // this wave's block-set area: free in Plan C
    const uint32_t okcap = GRIM_MAXL * A.bset_cap;
    for (uint32_t c0 = 0; c0 < sp.ncand; c0 += 64) {
      const uint32_t c = c0 + lane;
      double v;
      uint64_t k;
      const bool on = c < sp.ncand && plan_c_candidate(A, sp, tok, c, v, k);
      const uint64_t m = __ballot(on);
      if (on) {
        const uint32_t pos = n_ok + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < okcap) okidx[pos] = c;
      }
      n_ok += (uint32_t)__popcll(m);
    }
    __threadfence_block();
    if (n_ok > okcap) n_ok = okcap;  // (cannot happen: bset_cap >= the biggest label, candidates of a Plan-C side are few)
    const uint32_t nc = save_select(n_ok, [&](uint32_t i) {
      double v = 0.0;
      uint64_t k;
      plan_c_candidate(A, sp, tok, okidx[i], v, k);
      return v;
    }, kc);
    const uint32_t nr = save_select(rn, [&](uint32_t i) { return pop_sum(g, g.lab_nodes[r0 + i]); }, kr);
    const uint32_t total = nc * nr;
    for (uint32_t i0 = 0; i0 < total; i0 += 64) {
      const uint32_t i = i0 + lane;
      const bool valid = i < total;
      double p = 0.0;
      uint64_t key = 0;
      if (valid) {
        const uint32_t e1 = i / nr, e2 = i - e1 * nr;
        double v = 0.0;
        plan_c_candidate(A, sp, tok, okidx[kc[e1]], v, key);
        const uint32_t rn_node = g.lab_nodes[r0 + kr[e2]];
        p = v * pop_sum(g, rn_node) * GRIM_FACTOR_JOIN;
        key |= g.node_key[rn_node];
      }
      const uint64_t tie = (((uint64_t)i * (uint64_t)P) << 8);
      top_push(L, st, valid && p > 0.0, p, p * prior[0], tie, 0, key);
    }
    store_top<true>(S, sh, L, st, row);
    return st.nrun > 0;
  }
  const uint64_t total = (uint64_t)sp.ncand * rn;
  for (uint64_t i0 = 0; i0 < total; i0 += 64) {
    uint64_t i = i0 + lane;
    bool valid = i < total;
    double p = 0.0;
    uint64_t key = 0;
    if (valid) {
      uint32_t c = (uint32_t)(i / rn), r = (uint32_t)(i % rn);
      double cur = 0.0;
      if (plan_c_candidate(A, sp, tok, c, cur, key)) {
        if (miss) {
          uint32_t rn_node = g.lab_nodes[r0 + r];
          cur = cur * pop_sum(g, rn_node) * GRIM_FACTOR_JOIN;
          key |= g.node_key[rn_node];
        }
        p = cur;
      }
    }
    bool act = valid && p > 0.0;
    uint64_t tie = ((i * (uint64_t)P) << 8);
    top_push(L, st, act, p, p * prior[0], tie, 0, key);
  }
  store_top<true>(S, sh, L, st, row);
  return st.nrun > 0;
}

// ---- the fallback passes; each leaves its accepted pairs in the slot (U) and returns their count --------
// empties the composite-haplotype table for a pass that will hold at most `slots / 2` of them.  All threads call.
__device__ inline void comp_reset(WgShared &sh, const Slot &S, uint32_t slots) {
  __syncthreads();
  if (threadIdx.x == 0) sh.comp_mask = slots - 1;
  for (uint32_t s = threadIdx.x; s < slots; s += GRIM_WG) S.comp[s] = 0;
  __syncthreads();
}
#define GRIM_COMP_SMALL 8192u  // >= 2 * GRIM_SIDES * GRIM_TOPCAP: one set of top lists

struct PbState {
  uint8_t memo[GRIM_SIDES];
  uint8_t have[GRIM_SIDES];  // pb_levels: the matrix row whose top list the slot holds for this side (0xFF none, 0xFE the unseen-allele variant)
  uint8_t side_scan[GRIM_SIDES], side_any[GRIM_SIDES];
  uint32_t absent_side[2];
  uint32_t flag;
};

// candidates per side for the current list versions: which sides the label scan opens, which phases
// survive open_phases (impute.py:987-988), which positions hold only unseen alleles (impute.py:1224-1241)
__device__ inline void pb_open(const DevArgs &A, WgShared &sh, PbState &st, const uint16_t *tok) {
  const int tid = threadIdx.x;
  const int nph = sh.nph;
  if (tid < GRIM_SIDES) {
    st.side_scan[tid] = 0;
    st.side_any[tid] = 0;
  }
  if (tid < 2) st.absent_side[tid] = 0;
  __syncthreads();
  for (int s = wave_id(); s < 2 * nph; s += GRIM_NWAVE) {
    SideSpec sp = side_spec(A, sh, s >> 1, s & 1);
    uint32_t ncand = sp.expansion ? 1u : scan_count(A, sp, tok);
    if (lane_id() == 0) {
      st.side_scan[s] = sp.expansion ? 0 : 1;
      st.side_any[s] = ncand ? 1 : 0;
    }
  }
  __syncthreads();
}

__device__ inline void pb_absent(const DevArgs &A, WgShared &sh, PbState &st, const uint16_t *tok) {
  const int tid = threadIdx.x;
  const int nph = sh.nph;
  if (tid < 2) st.absent_side[tid] = 0;
  __syncthreads();
  if (tid < 2 * GRIM_MAXL) {
    const int side = tid / GRIM_MAXL, l = tid % GRIM_MAXL;
    if (l < sh.subj.n_loci) {
      bool any = false;
      for (int i = 0; i < nph; ++i) {
        if (!(st.side_any[2 * i] && st.side_any[2 * i + 1])) continue;  // phase dropped by open_phases
        if (st.side_scan[2 * i + side]) {
          any = true;  // candidates of a label scan are graph nodes: their alleles are known
          continue;
        }
        SideSpec sp = side_spec(A, sh, i, side);
        for (uint32_t t = 0; t < sp.cn[l]; ++t) any |= allele_known(A.g, sp.sl[l], tok[sp.to[l] + t]);
      }
      if (!any) atomicOr(&st.absent_side[side], 1u << l);
    }
  }
  __syncthreads();
}

__device__ inline uint32_t pb_pairs(const DevArgs &A, WgShared &sh, const Slot &S, const double *prior, double *mx) {
  const uint32_t np = pair_offsets(sh);
  if (np == 0) return 0;
  return pair_pass(A, sh, S, prior, np, 0.0, true, mx);
}

// Plan B proper: two prior levels x (matrix rows + rescue) (impute.py:1696-1722, 1392-1570)
__device__ inline uint32_t pb_levels(const DevArgs &A, WgShared &sh, const Slot &S, WaveTop *wt, PbState &st, const uint16_t *tok,
                                     double *mx) {
  const int tid = threadIdx.x;
  const int P = A.g.P;
  const int nph = sh.nph;
  for (int level = 0; level < 2; ++level) {
    const double *prior = A.priors + (uint64_t)(level == 0 ? sh.subj.prior_idx : A.ones_prior) * P * P;
    if (tid < GRIM_SIDES) {
      st.memo[tid] = 10;
      st.have[tid] = 0xFF;
      sh.Tn[tid] = 0;
    }
    // the composite-haplotype table serves the rows of the level as long as it has room: a side whose best row is known
    // keeps its list from one row to the next (the reference recomputes it for every later row: same arguments, same list)
    comp_reset(sh, S, GRIM_COMP_CAP);
    if (tid == 0) sh.bc[7] = 0;  // entries the table may hold
    __syncthreads();
    for (int m = 0; m < (int)A.prm.planb_rows; ++m) {
      STAMP_BEGIN();
      if (sh.bc[7] > GRIM_COMP_CAP / 4) {  // the next row's lists might not fit: start over with an empty table
        comp_reset(sh, S, GRIM_COMP_CAP);
        if (tid < GRIM_SIDES) {
          st.have[tid] = 0xFF;
          sh.Tn[tid] = 0;
        }
        if (tid == 0) sh.bc[7] = 0;
        __syncthreads();
      }
      for (int s = wave_id(); s < 2 * nph; s += GRIM_NWAVE) {
        if (!(st.side_any[s] && st.side_any[s ^ 1])) continue;  // phase dropped by open_phases
        SideSpec sp = side_spec(A, sh, s >> 1, s & 1);
        const uint32_t ab = st.absent_side[s & 1];
        if (ab == 0) {
          int idx = m < (int)st.memo[s] ? m : (int)st.memo[s];
          if ((int)st.have[s] == idx) continue;  // the list of this row is in place
          bool nonempty = side_row(A, sh, S, prior, wt[wave_id()], sp, tok, idx, s);
          if (lane_id() == 0) {
            st.have[s] = (uint8_t)idx;
            if (nonempty) st.memo[s] = (uint8_t)idx;
            atomicAdd(&sh.bc[7], sh.Tn[s]);
          }
        } else {
          if (st.have[s] == 0xFE) continue;
          side_absent(A, sh, S, prior, wt[wave_id()], sp, tok, ab, s);
          if (lane_id() == 0) {
            st.have[s] = 0xFE;
            atomicAdd(&sh.bc[7], sh.Tn[s]);
          }
        }
      }
      __syncthreads();
      STAMP(1);
      uint32_t nU = pb_pairs(A, sh, S, prior, mx);
      STAMP(2);
      if (nU) return nU;
    }
    // rescue loop (impute.py:1490-1558): its six iterations are identical, one suffices
    comp_reset(sh, S, GRIM_COMP_SMALL);
    if (tid < GRIM_SIDES) sh.Tn[tid] = 0;
    __syncthreads();
    for (int s = wave_id(); s < 2 * nph; s += GRIM_NWAVE) {
      const int mine = st.memo[s], other = st.memo[s ^ 1];
      if (!(st.side_any[s] && st.side_any[s ^ 1])) continue;
      if ((mine == 10) == (other == 10)) continue;  // both unset: skipped; both set: stale lists, no-op
      SideSpec sp = side_spec(A, sh, s >> 1, s & 1);
      if (mine == 10) {
        uint32_t ab = absent_positions(A, tok, sp);
        side_absent(A, sh, S, prior, wt[wave_id()], sp, tok, ab, s);
      } else {
        side_row(A, sh, S, prior, wt[wave_id()], sp, tok, mine, s);
      }
    }
    __syncthreads();
    uint32_t nU = pb_pairs(A, sh, S, prior, mx);
    if (nU) return nU;
  }
  return 0;
}

// Plan A on the current (possibly reduced) specs, inside this kernel: the phased pass after a MUUG pass
// that ended in Plan C starts over with the reduced phases (impute.py:1637-1648)
__device__ inline uint32_t pb_plan_a(const DevArgs &A, WgShared &sh, const Slot &S, WaveTop *wt, PbState &st, const uint16_t *tok,
                                     double *mx) {
  const int tid = threadIdx.x;
  const int P = A.g.P;
  const int nph = sh.nph;
  const double *prior = A.priors + (uint64_t)sh.subj.prior_idx * P * P;
  comp_reset(sh, S, GRIM_COMP_SMALL);
  if (tid < GRIM_SIDES) sh.Tn[tid] = 0;
  __syncthreads();
  for (int s = wave_id(); s < 2 * nph; s += GRIM_NWAVE) {
    if (!(st.side_any[s] && st.side_any[s ^ 1])) continue;
    SideSpec sp = side_spec(A, sh, s >> 1, s & 1);
    side_lookup_full(A, sh, S, prior, wt[wave_id()], sp, tok, s);
  }
  __syncthreads();
  const uint32_t np = pair_offsets(sh);
  if (np == 0) return 0;
  const int e = ladder_first(A, sh, S, prior, np);
  if (e >= A.prm.n_ladder) return 0;
  double eps = A.prm.ladder[e];
  bool first = true;
  if (eps > 0.0) {
    pair_pass(A, sh, S, prior, np, eps, false, mx);
    eps = *mx / 100000.0;
    first = false;
  }
  return pair_pass(A, sh, S, prior, np, eps, true, mx, first);
}

// Plan C's reduction under `prior` (the all-ones matrix left by level 1): sets sh.bestc / sh.reduced
__device__ inline void pb_reduce_for_plan_c(const DevArgs &A, WgShared &sh, const uint16_t *tok, const double *prior) {
  const int tid = threadIdx.x;
  const int P = A.g.P;
  const int nph = sh.nph;
  for (int q = tid; q < 2 * nph * GRIM_MAXL; q += GRIM_WG) {
    const int sd = q / GRIM_MAXL, l = q % GRIM_MAXL;
    uint16_t b = 0xFFFF;
    if (l < sh.subj.n_loci) {
      const int c = (int)((sh.ph_pat[sd >> 1] >> l) & 1u) ^ (sd & 1);
      const ListVer lv = sh.lv[l][c][sh.side_ver[sd]];
      double bs = 0.0;
      for (uint32_t t = 0; t < lv.cnt; ++t) {
        uint32_t node = graph_lookup(A.g, (uint64_t)(tok[lv.off + t] + 1u) << (GRIM_ABITS * sh.subj.slot[l]));
        if (node == GRIM_NONE) continue;
        double sc = 0.0;
        for (int j = 0; j < P; ++j) sc = sc + A.g.freq[(uint64_t)node * P + j] * prior[j * P + j];
        if (b == 0xFFFF || sc > bs) {
          b = (uint16_t)t;
          bs = sc;
        }
      }
    }
    sh.bestc[sd][l] = b;
  }
  __syncthreads();
  if (tid == 0) sh.reduced = 1;
  __syncthreads();
}

__device__ inline uint32_t pb_plan_c(const DevArgs &A, WgShared &sh, const Slot &S, WaveTop *wt, const PbState &st, const uint16_t *tok,
                                     double *mx) {
  const int tid = threadIdx.x;
  const int P = A.g.P;
  const int nph = sh.nph;
  const double *prior = A.priors + (uint64_t)A.ones_prior * P * P;
  comp_reset(sh, S, GRIM_COMP_SMALL);
  if (tid < GRIM_SIDES) sh.Tn[tid] = 0;
  __syncthreads();
  for (int s = wave_id(); s < 2 * nph; s += GRIM_NWAVE) {
    if (!(st.side_any[s] && st.side_any[s ^ 1])) continue;  // phase dropped by open_phases
    SideSpec sp = side_spec(A, sh, s >> 1, s & 1);
    side_plan_c(A, sh, S, prior, wt[wave_id()], sp, tok, s);
  }
  __syncthreads();
  return pb_pairs(A, sh, S, prior, mx);
}

__global__ __launch_bounds__(GRIM_WG, GRIM_PLANB_WG_PER_CU) void grim_plan_b_kernel(DevArgs A) {
  __shared__ WgShared sh;
  __shared__ WaveTop wt[GRIM_NWAVE];
  __shared__ PbState st;
  const int tid = threadIdx.x;
  const int P = A.g.P;
  const uint32_t n_heavy = A.queue[6], n_work = *A.next_count + n_heavy;
  Slot S = make_slot(A, blockIdx.x);
  wg_arena(sh, wt);  // the pair-stage work areas: free whenever pairs are scored, every pass has finished its sides by then
  __syncthreads();
  for (;;) {
    if (tid == 0) sh.bc[3] = atomicAdd(A.queue + 3, 1u);  // own counter: no reset between kernels
    __syncthreads();
    const uint32_t w = sh.bc[3];
    if (w >= n_work) break;
    const uint32_t si = w < n_heavy ? A.next_list[A.next_cap - 1u - w] : A.next_list[w - n_heavy];  // heavy ones first
    if (tid < 16) ((uint32_t *)&sh.subj)[tid] = ((const uint32_t *)&A.subj[si])[tid];
    if (tid < (int)(sizeof(grim_subject_result) / 4)) ((uint32_t *)&sh.out)[tid] = 0;
    if (tid == 0) sh.reduced = 0;
    __syncthreads();
    enumerate_phases(sh);
    const int nph = sh.nph;
    const uint16_t *tok = S.rtok;
    prepare_lists(A, sh, S);  // fits: the plan-A kernel checked
    // same opening (and rewrites of an empty open_phases, impute.py:1619-1627) as the plan-A kernel went through
    for (int stage = 0;; ++stage) {
      pb_open(A, sh, st, tok);
      bool kept = false;
      for (int i = 0; i < nph; ++i) kept |= (st.side_any[2 * i] && st.side_any[2 * i + 1]);
      if (kept || stage == 2) break;
      if (stage == 0) reduce_lists(A, sh, S, A.priors + (uint64_t)sh.subj.prior_idx * P * P);
      apply_stage(A, sh, stage + 1);
    }
    pb_absent(A, sh, st, tok);
    double mx = 0.0;
    uint8_t status = GRIM_ST_MISS, reason = 0, plan = 'b', plan_haps = 0;
    // ---- the pass on the phases as opened: Plan B (Plan A already failed in the first kernels) --------
    STAMP_BEGIN();
    uint32_t nU = pb_levels(A, sh, S, wt, st, tok, &mx);
    STAMP(0);
    if (nU) {
      emit_tables(A, sh, S, nU, sh.out, si, 3);
      STAMP(3);
      status = GRIM_ST_OK;
    } else if (!A.prm.out_muug && A.prm.em) {
      // impute_file(em=True): the phased pass stops after Plan B (impute.py:1649); nothing was found
    } else {
      // ---- Plan C (impute.py:1637-1643 / 1649-1654): lists reduced to their most common allele under the
      // all-ones prior level 1 left behind; every phase opens again
      pb_reduce_for_plan_c(A, sh, tok, A.priors + (uint64_t)A.ones_prior * P * P);
      pb_open(A, sh, st, tok);
      // A side whose remaining (all unseen) lists still multiply to the options threshold is opened by the
      // label scan, which finds nothing -- every allele of a graph node is a seen one -- so open_phases
      // drops its phase (side_any = 0); Plan C runs on the phases that are left, possibly none.
      bool wide = false;
      for (int s = 0; s < 2 * nph; ++s) wide |= st.side_scan[s] != 0 && st.side_any[s] != 0;
      if (wide) {
        status = GRIM_ST_UNSUPPORTED;  // cannot happen by the argument above; never guess
        reason = 3;
      } else {
        plan = 'c';
        const bool two_pass = A.prm.out_muug && A.prm.out_haps;
        // With both outputs on, the phased pass starts over on the REDUCED phases (impute.py:1645-1654): Plan A
        // with the subject's prior, Plan B's two levels, then Plan C again.  The two passes do not depend on
        // each other, so the (cheap) phased Plan A / Plan B attempts run first; when they find nothing the one
        // Plan-C pass below serves both halves instead of being computed twice.
        bool phased_done = false;
        if (two_pass) {
          double mx2 = 0.0;
          pb_absent(A, sh, st, tok);
          plan_haps = 'a';
          uint32_t nH = pb_plan_a(A, sh, S, wt, st, tok, &mx2);
          STAMP(5);
          if (!nH) {
            plan_haps = 'b';
            nH = pb_levels(A, sh, S, wt, st, tok, &mx2);
          }
          STAMP(6);
          if (nH) {
            emit_tables(A, sh, S, nH, sh.out, si, 2u);
            status = GRIM_ST_OK;
            phased_done = true;
          } else if (!A.prm.em) {
            plan_haps = 'c';
          }
          STAMP(7);
        }
        nU = pb_plan_c(A, sh, S, wt, st, tok, &mx);
        STAMP(4);
        if (nU) {
          const uint32_t halves = !two_pass ? 3u : (phased_done || A.prm.em) ? 1u : 3u;
          emit_tables(A, sh, S, nU, sh.out, si, halves);
          status = GRIM_ST_OK;
        }
        STAMP(3);
      }
    }
    __syncthreads();
    if (tid == 0) {
      sh.out.status = status;
      sh.out.reason = reason;
      sh.out.plan = plan;
      sh.out.plan_phased = plan_haps;
      sh.out.max_prob = mx;
      A.res[si] = sh.out;
    }
    __syncthreads();
  }
}

static int grim_launch_plan_b(DevArgs &A, uint32_t n_slots, hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
  // the plan-A kernels leave the list of subjects in next_list/next_count
  if (start && stop)
    hipExtLaunchKernelGGL(grim_plan_b_kernel, dim3(n_slots), dim3(GRIM_WG), 0, stream, start, stop, 0, A);
  else
    hipLaunchKernelGGL(grim_plan_b_kernel, dim3(n_slots), dim3(GRIM_WG), 0, stream, A);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
