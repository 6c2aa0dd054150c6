// grim_plan_a.h -- Plan A for one subject: phase enumeration, candidate enumeration per phase side,
// exact-name lookup + top-link (CSR) neighbour gather, running top-K.
//   gen_phases            impute.py:274-303
//   open_phases           impute.py:914-989  (cutils.pyx:4-51: cartesian opening / label filter)
//   get_haplo_freqs       impute.py:393-397  -> Graph.adjs_query networkx_graph.py:253-278
//   convert_list_to_one_dim impute.py:424-442
#pragma once
#include "grim_pair.h"

// A side is opened by walking its label instead of probing its cartesian product when the product has more
// than half as many candidates as the label has nodes: measured on 512 subjects with w alternatives per locus
// and side on the CAU graph (3 380 full haplotypes, tools/highamb_probe.py), probing costs 0.76 / 1.68 / 3.35 ms
// at w = 5 / 6 / 7 (3 125 / 7 776 / 16 807 candidates per side), walking the label 0.44 - 0.57 ms at any w.

// phases kept by gen_phases: pattern bit l set = position l takes side-2's allele list for H1.
// same_mask bit l set = the two side STRINGS of position l are identical (flipping is a no-op).
__device__ inline void enumerate_phases(WgShared &sh) {
  if (threadIdx.x == 0) {
    int n = sh.subj.n_loci;
    uint32_t same = sh.subj.pad[0];
    uint32_t het = ((1u << n) - 1u) & ~same;
    const uint32_t movable = het & ~(uint32_t)sh.subj.flags;  // phase mask: fixed positions never switch
    uint32_t seen = 0;
    int cnt = 0;
    for (uint32_t i = 0; i < (1u << (n - 1)); ++i) {
      uint32_t p = i & movable;
      if (!((seen >> p) & 1u)) {
        seen |= (1u << p) | (1u << (p ^ het));
        sh.ph_pat[cnt++] = (uint8_t)p;
      }
    }
    sh.nph = cnt;
    uint32_t acc = 0;
    for (int l = 0; l < n; ++l)
      for (int s = 0; s < 2; ++s) {
        sh.toff[l][s] = acc;
        acc += sh.subj.cnt[l][s];
      }
  }
  __syncthreads();
}

// Copy the subject's allele lists into the slot and set up the list versions (all version 0).
// Returns false when the lists do not fit (GRIM_RTOK_CAP).  All threads call.
__device__ inline bool prepare_lists(const DevArgs &A, WgShared &sh, const Slot &S) {
  const grim_subject &sj = sh.subj;
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int l = 0; l < sj.n_loci; ++l)
      for (int c = 0; c < 2; ++c) acc += sj.cnt[l][c];
    sh.ntok = acc;
  }
  if (threadIdx.x < GRIM_SIDES) sh.side_ver[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t ntok = sh.ntok;
  if (3 * ntok > GRIM_RTOK_CAP) return false;
  for (uint32_t t = threadIdx.x; t < ntok; t += GRIM_WG) S.rtok[t] = A.tok[sj.tok_off + t];
  const bool ambiguous = ntok > 2u * sj.n_loci;  // otherwise no side ever has enough candidates to intersect
  if (ambiguous)
    for (int i = threadIdx.x; i < GRIM_MAXL * 2 * 128; i += GRIM_WG) (&sh.abits[0][0][0])[i] = 0;
  if (threadIdx.x < 2 * GRIM_MAXL) {
    const int l = threadIdx.x >> 1, c = threadIdx.x & 1;
    ListVer v;
    v.off = l < sj.n_loci ? sh.toff[l][c] : 0;
    v.cnt = l < sj.n_loci ? sj.cnt[l][c] : 0;
    v.wid = l < sj.n_loci ? sj.wid[l][c] : 0;
    sh.lv[l][c][0] = sh.lv[l][c][1] = sh.lv[l][c][2] = v;
  }
  __syncthreads();
  if (ambiguous && threadIdx.x < 2 * GRIM_MAXL) {
    const int l = threadIdx.x >> 1, c = threadIdx.x & 1;
    const ListVer v = sh.lv[l][c][0];
    for (uint32_t t = 0; t < v.cnt; ++t) {
      const uint32_t tk = A.tok[sj.tok_off + v.off + t];
      sh.abits[l][c][(tk >> 5) & 127u] |= 1u << (tk & 31u);
    }
  }
  __syncthreads();
  // Can two of the subject's pairs ever name the same unordered {(haplotype, population), (haplotype, population)}?  A
  // haplotype of a phase side carries, at every typed position, an allele of that side's list there.  When the two lists of
  // every position are either the same text or DISJOINT sets, and some position differs, any two different sides differ at a
  // position with disjoint lists, so no haplotype belongs to two sides: the two sides of a phase never share an entity and
  // no two kept phases (which are never mirror images) share a side -- every scored pair is unique, and the first-wins
  // dedup of calc_haps_pairs (impute.py:506-511, 603-611) has nothing to do.  (The reduced list versions are subsets.)
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    bool het = false, clash = false;
    if (l < sj.n_loci && !((sj.pad[0] >> l) & 1u)) {
      het = true;
      if (ambiguous) {
        for (int w = 0; w < 128 && !clash; ++w) clash = (sh.abits[l][0][w] & sh.abits[l][1][w]) != 0;
      } else {
        clash = A.tok[sj.tok_off + sh.toff[l][0]] == A.tok[sj.tok_off + sh.toff[l][1]];
      }
    }
    const uint64_t hm = __ballot(het), cm = __ballot(clash);
    if (threadIdx.x == 0) sh.nodup = (hm != 0 && cm == 0) ? 1 : 0;
  }
  __syncthreads();
  return true;
}

// The two rewrites the reference applies when open_phases finds no candidate in any phase
// (impute.py:1620-1627): version 1 of a list keeps the alleles the graph knows (in list order),
// version 2 the 10 with the largest sum_p freq[p]*prior[p][p] (stable, bigger first); a list none of
// whose alleles is known stays as typed.  One thread per list; scores parked in the slot.
__device__ inline void reduce_lists(const DevArgs &A, WgShared &sh, const Slot &S, const double *prior) {
  const DevGraph &g = A.g;
  const grim_subject &sj = sh.subj;
  const int P = g.P;
  if (threadIdx.x < 2 * GRIM_MAXL) {
    const int l = threadIdx.x >> 1, c = threadIdx.x & 1;
    if (l < sj.n_loci) {
      const ListVer v0 = sh.lv[l][c][0];
      const uint32_t off1 = sh.ntok + v0.off, off2 = 2 * sh.ntok + v0.off;
      double *score = S.gsum + v0.off;  // free while sides are being opened
      uint32_t n1 = 0;
      for (uint32_t t = 0; t < v0.cnt; ++t) {
        const uint32_t tk = S.rtok[v0.off + t];
        const uint32_t node = graph_lookup(g, (uint64_t)(tk + 1u) << (GRIM_ABITS * sj.slot[l]));
        if (node == GRIM_NONE) continue;
        double sc = 0.0;
        for (int j = 0; j < P; ++j) sc = sc + g.freq[(uint64_t)node * P + j] * prior[j * P + j];
        S.rtok[off1 + n1] = (uint16_t)tk;
        score[n1] = sc;
        ++n1;
      }
      if (n1 > 0) {
        ListVer v1;
        v1.off = off1; v1.cnt = (uint16_t)n1; v1.wid = (uint16_t)n1;
        sh.lv[l][c][1] = v1;
        const uint32_t keep = n1 < 10 ? n1 : 10;
        for (uint32_t t = 0; t < n1; ++t) {
          uint32_t rank = 0;
          for (uint32_t t2 = 0; t2 < n1; ++t2) rank += (score[t2] > score[t] || (score[t2] == score[t] && t2 < t)) ? 1u : 0u;
          if (rank < keep) S.rtok[off2 + rank] = S.rtok[off1 + t];
        }
        ListVer v2;
        v2.off = off2; v2.cnt = (uint16_t)keep; v2.wid = (uint16_t)keep;
        sh.lv[l][c][2] = v2;
      }
    }
  }
  __syncthreads();
}

// sides whose option count reaches the threshold switch to list version `stage`
// (impute.py:870-873, 889-892: the rewrites only touch such sides).  Returns nothing; all threads call.
__device__ inline void apply_stage(const DevArgs &A, WgShared &sh, int stage) {
  const grim_subject &sj = sh.subj;
  const int s = threadIdx.x;
  if (s < 2 * sh.nph) {
    const uint32_t pat = sh.ph_pat[s >> 1];
    uint64_t options = 1;
    for (int l = 0; l < sj.n_loci; ++l) {
      int c = (int)((pat >> l) & 1u) ^ (s & 1);
      options *= (uint64_t)sh.lv[l][c][sh.side_ver[s]].wid;
      if (options > 0xFFFFFFFFFFFFull) options = 0xFFFFFFFFFFFFull;
    }
    if (!(options < A.prm.opt_threshold)) sh.side_ver[s] = (uint8_t)stage;
  }
  __syncthreads();
}

// Expand the chunk of <=64 look-up hits held one per lane into (hap, pop) entries and push them
// through the running top-K.  DIRECT: `src` is the haplotype node itself; otherwise `src` is a CSR
// row (plan A: a partial node's top links; plan B: a connector's parents) and every neighbour is
// a haplotype.  AUX: entries carry the 60-bit key node_key | L.caux[owner] and p is scaled.
// RANKED: hits arrive in arbitrary order; aux_or carries the hit's position in the reference's candidate
// stream (< 2^28) and `tie` is rebuilt from (that position, neighbour index < 2^22, population < 64), so the
// top-K keeps exactly the entries -- in exactly the order -- the in-order stream would have produced.
template <bool AUX, bool RANKED = false>
__device__ __forceinline__ void expand_chunk(const DevArgs &A, const double *prior, WaveTop &L, TopState &st, uint32_t src,
                                             bool direct, const uint32_t *csr_start, const uint32_t *csr_nbr, double scale,
                                             uint64_t aux_or, uint64_t &item_base, uint64_t &c_nbr, uint64_t &c_freq) {
  const DevGraph &g = A.g;
  const int lane = lane_id();
  const int P = g.P;
  uint32_t cnt = 0;
  if (src != GRIM_NONE) cnt = direct ? 1u : nbr_count(csr_start, src);
  uint32_t inc = wave_incl_scan(cnt);
  uint32_t total = __shfl(inc, 63);
  if (total == 0) return;
  L.cstart[lane] = inc - cnt;
  L.cnode[lane] = src;
  if (AUX || RANKED) L.caux[lane] = aux_or;
  if (lane == 0) L.cstart[64] = total;
  WAVE_SYNC();
  for (uint32_t t0 = 0; t0 < total; t0 += 64) {
    uint32_t t = t0 + lane;
    bool valid = t < total;
    uint32_t hap = 0;
    uint64_t aux = 0, rtie = 0;
    if (valid) {
      int lo = 0, hi = 63;  // owner lane: last l with cstart[l] <= t
      while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (L.cstart[mid] <= t) lo = mid; else hi = mid - 1;
      }
      uint32_t nd = L.cnode[lo];
      hap = direct ? nd : csr_nbr[csr_start[nd] + (t - L.cstart[lo])];
      if (AUX) aux = g.node_key[hap] | L.caux[lo];
      if (RANKED) rtie = ((L.caux[lo] << 22) | (uint64_t)(t - L.cstart[lo])) << 14;
    }
    // four populations' frequencies per step: the loads are issued together (top_push's LDS fences would otherwise put a
    // memory round trip between one population and the next -- half of a side's dependent chain at P = 4)
    for (int j0 = 0; j0 < P; j0 += 4) {
      double pv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) pv[q] = (valid && j0 + q < P) ? g.freq[(uint64_t)hap * P + j0 + q] : 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        if (j >= P) break;
        double p = pv[q];
        if (AUX) p = p * scale;
        bool act = valid && p > 0.0;
        double key = p * prior[j * P + j];
        uint64_t tie = (((item_base + t) * (uint64_t)P + (uint64_t)j) << 8) | (uint64_t)j;
        if (RANKED) tie = rtie | ((uint64_t)j << 8) | (uint64_t)j;
        top_push(L, st, act, p, key, tie, hap, aux);
      }
    }
  }
  item_base += total;
  if (!direct) c_nbr += total;
  c_freq += total;
  WAVE_SYNC();
}

// write the finished list of one side: probabilities, prefix-min, entities.  COMP: haplotypes are
// identified by their 60-bit key through the slot's canonical table (plan B/C) instead of node ids.
template <bool COMP>
__device__ __forceinline__ void store_top(const Slot &S, WgShared &sh, WaveTop &L, TopState &st, int row) {
  const int lane = lane_id();
  top_flush(L, st);
  int n = st.nrun;
  // prefix-min of p over the list order (the pair loop's `break`, impute.py:463-464,545-546)
  double carry = __longlong_as_double(0x7FF0000000000000ll);
  for (int r0 = 0; r0 < n; r0 += 64) {
    int r = r0 + lane;
    double v = r < n ? L.p[r] : __longlong_as_double(0x7FF0000000000000ll);
    for (int d = 1; d < 64; d <<= 1) {
      double o = __shfl_up(v, d);
      if (lane >= d && o < v) v = o;
    }
    if (carry < v) v = carry;
    if (r < n) {
      uint32_t hap = L.hap[r];
      if (COMP) hap = HAP_COMPOSITE | tab_insert(S.comp, sh.comp_mask, L.aux[r] | GRIM_VALID);
      S.Tp[row * GRIM_TOPCAP + r] = L.p[r];
      S.Tm[row * GRIM_TOPCAP + r] = v;
      S.Te[row * GRIM_TOPCAP + r] = hap | ((uint32_t)(L.tie[r] & 0xFF) << 24);
    }
    carry = __shfl(v, 63);
  }
  if (lane == 0) sh.Tn[row] = (uint32_t)n;
  WAVE_SYNC();
}

// ---- one label scan for ALL sides of a subject ----------------------------------------------------------------------
// Every side of a subject opens over the same label (the typed loci) and differs only in which column of the subject's
// '/'-lists it takes per position, so one pass over the label's key stream can serve all <= 32 sides: a node's ten
// membership bits (position x column, sh.abits) decide for every side at once.  That replaces 32 scans of the label
// (sides with >= number_of_options_threshold options, impute.py:947-981) or 32 x candidates random probes of the name index
// (cartesian sides, cutils.pyx:4-31) -- on a WMDA-scale graph (240 000 full haplotypes, 64 MB index) 8-30 ms per subject --
// by ~0.1 ms of streaming.  Hits are parked per side in the slot (label positions) and pushed through the ranked top-K
// with the tie the in-order stream would have given them (cartesian position / label position), so the lists are bit for
// bit what build_side_plan_a produces.
// hits per side the slot holds: S.ska and S.skb (adjacent in the slot, 16 bytes per pair of pair_cap) viewed as u32
__device__ __forceinline__ uint32_t hit_cap(const DevArgs &A) { return (uint32_t)((4ull * A.pair_cap) / GRIM_SIDES); }

// decides (thread 0) whether the shared scan pays: sh.bc[6] = 1 / 0.  All threads call.
__device__ inline bool shared_scan_wanted(const DevArgs &A, WgShared &sh) {
  const DevGraph &g = A.g;
  const grim_subject &sj = sh.subj;
  if (threadIdx.x == 0) {
    uint32_t ok = g.scan_ok && sj.n_loci >= 1 && sh.ntok > 2u * sj.n_loci && hit_cap(A) >= 1024u;
    uint32_t mask = 0;
    for (int l = 0; l < sj.n_loci; ++l) mask |= 1u << sj.slot[l];
    if (!subject_order_ok(g, mask)) ok = 0;  // cartesian sides of such a subject find nothing: the sides are opened one by one
    const uint32_t la = g.lab_start[mask], lb = g.lab_start[mask + 1];
    uint64_t work = 0;  // what the per-side openings would cost, in label nodes / probes
    for (int s = 0; s < 2 * sh.nph && ok; ++s) {
      if (sh.side_ver[s] != 0) ok = 0;  // reduced lists (impute.py:1620-1627): the bits describe the lists as typed
      const uint32_t pat = sh.ph_pat[s >> 1];
      uint64_t options = 1, ncand = 1;
      for (int l = 0; l < sj.n_loci; ++l) {
        const int c = (int)((pat >> l) & 1u) ^ (s & 1);
        const ListVer lv = sh.lv[l][c][0];
        options *= (uint64_t)lv.wid;
        if (options > 0xFFFFFFFFFFFFull) options = 0xFFFFFFFFFFFFull;
        ncand *= lv.cnt;
        if (ncand > 0xFFFFFFFFFFFFull) ncand = 0xFFFFFFFFFFFFull;
        if (options < A.prm.opt_threshold && lv.cnt > 64) ok = 0;  // a cartesian side ranks its hits through LDS lists of <= 64
      }
      if (options < A.prm.opt_threshold) {
        if (ncand >= (1ull << 28)) ok = 0;
        work += ncand;
      } else {
        work += lb - la;
      }
    }
    if (lb - la >= (1u << 28) || work < 2ull * (lb - la)) ok = 0;
    sh.bc[6] = ok;
  }
  __syncthreads();
  const bool r = sh.bc[6] != 0;
  __syncthreads();
  return r;
}

// All sides of the subject by one scan.  Returns false when a side found more nodes than the slot holds (the caller opens
// the sides one by one then).  All threads call.
#ifdef GRIM_STAMPS  // (diagnostic build) workgroup time of the two phases, added to the stage timers 13 / 14
#define SCAN_STAMP(k)                                                                         \
  do {                                                                                        \
    if (threadIdx.x == 0) {                                                                   \
      const unsigned long long _n = wall_clock64();                                            \
      atomicAdd(&A.counters[GRIM_STAMP_BASE + (k)], _n - _scan_t0);                           \
      _scan_t0 = _n;                                                                          \
    }                                                                                         \
  } while (0)
#else
#define SCAN_STAMP(k)
#endif
__device__ inline bool build_sides_shared_scan(const DevArgs &A, WgShared &sh, const Slot &S, const double *prior, WaveTop *wt) {
#ifdef GRIM_STAMPS
  unsigned long long _scan_t0 = wall_clock64();
#endif
  const DevGraph &g = A.g;
  const grim_subject &sj = sh.subj;
  const int tid = threadIdx.x, lane = lane_id();
  const int n = sj.n_loci, nsides = 2 * sh.nph;
  uint32_t mask = 0, sl[GRIM_MAXL];
#pragma unroll
  for (int l = 0; l < GRIM_MAXL; ++l) {
    sl[l] = l < n ? sj.slot[l] : 0;
    if (l < n) mask |= 1u << sl[l];
  }
  const uint32_t la = g.lab_start[mask], lb = g.lab_start[mask + 1];
  uint32_t *hits = (uint32_t *)S.ska;  // [GRIM_SIDES][HCAP] label positions
  const uint32_t HCAP = hit_cap(A);
  if (tid < GRIM_SIDES) sh.hitn[tid] = 0;
  if (tid == 0) sh.bc[7] = 0;
  __syncthreads();
  // ---- phase 1: the label's keys.  Eight keys per thread and step: the eight loads are in flight together (one workgroup
  // per CU and four waves: nothing else hides the memory latency), and a side's requirement is one 10-bit mask
  // (bit 2l + c: position l, column c), so matching a node against all sides is a compare per side.
  if (tid < nsides) {
    const uint32_t pat = sh.ph_pat[tid >> 1];
    uint32_t r = 0;
    for (int l = 0; l < n; ++l) r |= 1u << (2 * l + (int)(((pat >> l) & 1u) ^ (uint32_t)(tid & 1)));
    sh.hitreq[tid] = r;
  }
  __syncthreads();
  // Two LDS tables for the scan, in the arena the top-K work areas use afterwards:
  //   comb[l][a >> 4]: two bits per allele id a -- is it in column 0 / column 1 of position l (one read per position);
  //   smatch[bits]   : which sides a node with these ten membership bits belongs to (a side asks for one column per
  //                    position), so a node that passes costs one read plus a step per side that takes it (0-2 as a
  //                    rule) instead of a compare against every side.
  uint32_t *comb = sh.hist;                    // [GRIM_MAXL][256]
  uint32_t *smatch = sh.hist + GRIM_MAXL * 256;  // [1 << 2n] <= 1024
  for (int k = tid; k < GRIM_MAXL * 256; k += GRIM_WG) {
    const int l = k >> 8, w16 = k & 255;  // alleles 16 * w16 .. 16 * w16 + 15
    uint32_t v = 0;
    if (l < n) {
      const uint32_t h0 = (sh.abits[l][0][w16 >> 1] >> (16 * (w16 & 1))) & 0xFFFFu, h1 = (sh.abits[l][1][w16 >> 1] >> (16 * (w16 & 1))) & 0xFFFFu;
      for (int a = 0; a < 16; ++a) v |= (((h0 >> a) & 1u) | (((h1 >> a) & 1u) << 1)) << (2 * a);
    }
    comb[k] = v;
  }
  for (uint32_t bits = tid; bits < (1u << (2 * n)); bits += GRIM_WG) {
    uint32_t m = 0;
    for (int s = 0; s < nsides; ++s) {
      const uint32_t r = sh.hitreq[s];
      m |= ((bits & r) == r ? 1u : 0u) << s;
    }
    smatch[bits] = m;
  }
  __syncthreads();
  constexpr int NK = 8;
  // (measured round 4 and dropped, neither moved the scan's 0.9 ms per subject: prefetching step t + 1's keys while step t's
  //  are tested; what stayed: a lane stops looking tables up at its first position that matches neither column)
  for (uint32_t i0 = la; i0 < lb; i0 += GRIM_WG * NK) {
    uint64_t key[NK];
#pragma unroll
    for (int q = 0; q < NK; ++q) {
      const uint32_t i = i0 + (uint32_t)q * GRIM_WG + tid;
      key[q] = i < lb ? g.lab_key[i] : 0ull;
    }
#pragma unroll
    for (int q = 0; q < NK; ++q) {
      const uint32_t i = i0 + (uint32_t)q * GRIM_WG + tid;
      if (i >= lb) continue;
      uint32_t bits = 0;  // bit 2l + c: the node's allele at position l is in column c's list
      bool possible = true;
      // a lane drops out at its first position that matches neither column (four nodes in five at the first one)
#pragma unroll
      for (int l = 0; l < GRIM_MAXL; ++l)
        if (l < n && possible) {
          const uint32_t al = ((uint32_t)(key[q] >> (GRIM_ABITS * sl[l])) & 0xFFFu) - 1u;
          const uint32_t two = (comb[l * 256 + ((al >> 4) & 255u)] >> (2 * (al & 15u))) & 3u;
          bits |= two << (2 * l);
          possible = two != 0;
        }
      if (!possible) continue;  // some position matches neither column: no side takes the node (nearly every node)
      for (uint32_t m = smatch[bits]; m; m &= m - 1) {
        const uint32_t s = (uint32_t)__builtin_ctz(m);
        const uint32_t pos = atomicAdd(&sh.hitn[s], 1u);
        if (pos < HCAP)
          hits[s * HCAP + pos] = i;
        else
          sh.bc[7] = 1;
      }
    }
  }
  __syncthreads();
  if (sh.bc[7]) return false;
  SCAN_STAMP(13);
  if (tid == 0) sh.wctr[0][0] += ((uint64_t)(lb - la) * 3) / 4;  // a scanned node costs 12 bytes (key + id), a probe 16
  // ---- phase 2: a wave per side pushes the side's nodes through the ranked top-K ----------------------------------------
  const bool full_nodes = (mask == g.full_mask);
  for (int s = wave_id(); s < nsides; s += GRIM_NWAVE) {
    WaveTop &L = wt[wave_id()];
    const uint32_t pat = sh.ph_pat[s >> 1];
    TopState st;
    st.nrun = 0; st.nbuf = 0; st.K = (int)A.prm.top_n; st.full = false; st.ge = true; st.thr = 0;
    uint32_t cn[GRIM_MAXL];
    uint64_t options = 1;
#pragma unroll
    for (int l = 0; l < GRIM_MAXL; ++l) {
      cn[l] = 1;
      if (l < n) {
        const int c = (int)((pat >> l) & 1u) ^ (s & 1);
        const ListVer lv = sh.lv[l][c][0];
        cn[l] = lv.cnt;
        options = options * (uint64_t)lv.wid;
        if (options > 0xFFFFFFFFFFFFull) options = 0xFFFFFFFFFFFFull;
      }
    }
    const bool cartesian = options < A.prm.opt_threshold;
    if (cartesian) {  // the side's lists in LDS: a hit's position in the cartesian order needs its digits
#pragma unroll
      for (int l = 0; l < GRIM_MAXL; ++l)
        if (l < n) {
          const int c = (int)((pat >> l) & 1u) ^ (s & 1);
          const ListVer lv = sh.lv[l][c][0];
          if ((uint32_t)lane < lv.cnt) L.toks[l * 64 + lane] = S.rtok[lv.off + lane];
        }
      WAVE_SYNC();
    }
    const uint32_t nh = sh.hitn[s];
    uint64_t item_base = 0, c_nbr = 0, c_freq = 0;
    for (uint32_t h0 = 0; h0 < nh; h0 += 64) {
      const uint32_t h = h0 + lane;
      uint32_t node = GRIM_NONE;
      uint64_t rank = 0;
      if (h < nh) {
        const uint32_t i = hits[(uint32_t)s * HCAP + h];
        node = g.lab_nodes[i];
        if (cartesian) {  // mixed-radix position, position 0 most significant (as the expansion counts, cutils.pyx:21-29)
          const uint64_t key = g.lab_key[i];
#pragma unroll
          for (int l = 0; l < GRIM_MAXL; ++l)
            if (l < n) {
              const uint32_t al = ((uint32_t)(key >> (GRIM_ABITS * sl[l])) & 0xFFFu) - 1u;
              uint32_t d = 0;
              while (d < cn[l] && (uint32_t)L.toks[l * 64 + d] != al) ++d;
              rank = rank * cn[l] + d;
            }
        } else {
          rank = i - la;  // the label scan keeps nodes in label order (impute.py:947-981)
        }
      }
      expand_chunk<false, true>(A, prior, L, st, node, full_nodes, g.a_start, g.a_nbr, 1.0, rank, item_base, c_nbr, c_freq);
    }
    store_top<false>(S, sh, L, st, s);
    if (lane == 0) {
      sh.cand_any[s] = (cartesian || nh > 0) ? 1 : 0;
      sh.wctr[wave_id()][1] += c_nbr;
      sh.wctr[wave_id()][2] += c_freq;
    }
  }
  __syncthreads();
  SCAN_STAMP(14);
  return true;
}

// Build the top list of phase `ph` (index into sh.ph_pat), side `side` into list row `row`.
__device__ inline void build_side_plan_a(const DevArgs &A, WgShared &sh, const Slot &S, const double *prior, WaveTop &L,
                                         int ph, int side, int row) {
  const DevGraph &g = A.g;
  const grim_subject &sj = sh.subj;
  const uint16_t *tok = S.rtok;
  const int lane = lane_id();
  const int n = sj.n_loci;
  const uint32_t pat = sh.ph_pat[ph];
  const int ver = sh.side_ver[row];
  TopState st;
  st.nrun = 0; st.nbuf = 0; st.K = (int)A.prm.top_n; st.full = false; st.ge = false; st.thr = 0;
  uint32_t cn[GRIM_MAXL], to[GRIM_MAXL], sl[GRIM_MAXL];
  uint64_t options = 1;
  uint32_t ncand = 1, mask = 0;
#pragma unroll
  for (int l = 0; l < GRIM_MAXL; ++l) {
    cn[l] = 1; to[l] = 0; sl[l] = 0;
    if (l < n) {
      int c = (int)((pat >> l) & 1u) ^ side;
      const ListVer lv = sh.lv[l][c][ver];
      cn[l] = lv.cnt;
      to[l] = lv.off;
      sl[l] = sj.slot[l];
      uint64_t w = lv.wid;
      options = options * w;
      if (options > 0xFFFFFFFFFFFFull) options = 0xFFFFFFFFFFFFull;
      ncand *= cn[l];
      mask |= 1u << sl[l];
    }
  }
  uint64_t item_base = 0, c_probe = 0, c_nbr = 0, c_freq = 0;
  bool any_cand = false;
  if (options < A.prm.opt_threshold) {
    // cartesian opening, position 0 most significant (cutils.pyx:21-29 applied locus by locus)
    any_cand = true;
    const bool full_nodes = (mask == g.full_mask);
    // Four chunks of 64 candidates per step: the four first probes are independent loads in flight
    // together; hits are expanded chunk by chunk, in order.  The candidate's mixed-radix digits are
    // advanced by +64 with carries instead of five 32-bit divisions per candidate (a lone wave is
    // instruction-issue bound: the divisions were most of the loop).
    uint32_t dg[GRIM_MAXL], inc64[GRIM_MAXL];
    {
      uint32_t rem = (uint32_t)lane, r64 = 64;
#pragma unroll
      for (int l = GRIM_MAXL - 1; l >= 0; --l) {
        dg[l] = 0;
        inc64[l] = 0;
        if (l < n) {
          dg[l] = rem % cn[l];
          rem /= cn[l];
          inc64[l] = r64 % cn[l];
          r64 /= cn[l];
        }
      }
    }
    // the side's allele lists in LDS (64 cycles instead of an L1/L2 round trip per digit)
    bool in_lds = true;
#pragma unroll
    for (int l = 0; l < GRIM_MAXL; ++l)
      if (l < n && cn[l] > 64) in_lds = false;
    if (in_lds) {
#pragma unroll
      for (int l = 0; l < GRIM_MAXL; ++l)
        if (l < n && (uint32_t)lane < cn[l]) L.toks[l * 64 + lane] = tok[to[l] + lane];
      WAVE_SYNC();
    }
    // Intersection opening: when the cartesian product dwarfs the label, walk the label's nodes instead
    // (coalesced key stream, one LDS bit per position) and give each hit its position in the cartesian
    // order; the ranked top-K then yields the list the in-order probe stream would.
    const uint32_t la = g.lab_start[mask], lb = g.lab_start[mask + 1];
    // a loci_map whose order is not the subject's for these loci: every candidate NAME misses (the candidates still count
    // as candidates: open_phases keeps the phase, impute.py:932-944)
    const bool findable = subject_order_ok(g, mask);
    if (!findable) ncand = 0;
    const bool intersect = findable && g.scan_ok && in_lds && ncand >= 1024u && ncand < (1u << 28) && 2ull * ncand > (uint64_t)(lb - la);
    if (intersect) {
      st.ge = true;
      for (uint32_t i0 = la; i0 < lb; i0 += 64) {
        const uint32_t i = i0 + lane;
        bool ok = i < lb;
        uint64_t key = ok ? g.lab_key[i] : 0;
#pragma unroll
        for (int l = 0; l < GRIM_MAXL; ++l)
          if (l < n) {
            const uint32_t al = ((uint32_t)(key >> (GRIM_ABITS * sl[l])) & 0xFFFu) - 1u;
            const int c = (int)((pat >> l) & 1u) ^ side;
            ok = ok && ((sh.abits[l][c][(al >> 5) & 127u] >> (al & 31u)) & 1u);
          }
        if (__ballot(ok) == 0) continue;
        uint64_t rank = 0;  // mixed-radix position, position 0 most significant (as the expansion counts)
#pragma unroll
        for (int l = 0; l < GRIM_MAXL; ++l)
          if (l < n && ok) {
            const uint32_t al = ((uint32_t)(key >> (GRIM_ABITS * sl[l])) & 0xFFFu) - 1u;
            uint32_t d = 0;
            while (d < cn[l] && (uint32_t)L.toks[l * 64 + d] != al) ++d;
            ok = d < cn[l];  // the bit came from version 0 of the list; a reduced version may lack the allele
            rank = rank * cn[l] + d;
          }
        const uint32_t node = ok ? g.lab_nodes[i] : GRIM_NONE;
        if (__ballot(ok) == 0) continue;
        expand_chunk<false, true>(A, prior, L, st, node, full_nodes, g.a_start, g.a_nbr, 1.0, rank, item_base, c_nbr, c_freq);
      }
      c_probe += ((uint64_t)(lb - la) * 3) / 4;  // a scanned node costs 12 bytes (key + id), a probe 16
    }
    constexpr int NQ = 8;  // chunks of 64 candidates per step: NQ independent first probes in flight
    for (uint32_t c0 = 0; !intersect && c0 < ncand; c0 += 64u * NQ) {
      uint64_t keyq[NQ];
      HtEnt entq[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        uint64_t key = 0;
#pragma unroll
        for (int l = 0; l < GRIM_MAXL; ++l)
          if (l < n) {
            const uint32_t t = in_lds ? (uint32_t)L.toks[l * 64 + dg[l]] : (uint32_t)tok[to[l] + dg[l]];
            key |= (uint64_t)(t + 1u) << (GRIM_ABITS * sl[l]);
          }
        keyq[q] = key;
        uint32_t carry = 0;  // digits += 64 (indices past ncand wrap harmlessly; `c < ncand` guards their use)
#pragma unroll
        for (int l = GRIM_MAXL - 1; l >= 0; --l) {
          if (l < n) {
            uint32_t v = dg[l] + inc64[l] + carry;
            carry = v >= cn[l] ? 1u : 0u;
            dg[l] = carry ? v - cn[l] : v;
          }
        }
      }
      uint32_t hq[NQ], nodeq[NQ];
      bool pend[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        hq[q] = (uint32_t)mix64(keyq[q]) & g.ht_mask;
        entq[q] = g.ht[hq[q]];
      }
      // Resolve the NQ chunks together: a lane whose slot holds another key walks on (linear probing), and with
      // 64 lanes per chunk some lane almost always does.  The follow-up probes of ALL chunks are issued before
      // any is waited for -- one memory round trip per probing round instead of one per round and chunk.
      bool any_pending = false;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const bool valid = c0 + 64u * q + lane < ncand;
        nodeq[q] = (valid && entq[q].key == keyq[q]) ? entq[q].val : GRIM_NONE;
        pend[q] = valid && entq[q].key != keyq[q] && entq[q].key != 0;
        any_pending |= pend[q];
      }
      while (__ballot(any_pending)) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (pend[q]) {
            hq[q] = (hq[q] + 1) & g.ht_mask;
            entq[q] = g.ht[hq[q]];
          }
        any_pending = false;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (pend[q]) {
            if (entq[q].key == keyq[q]) {
              nodeq[q] = entq[q].val;
              pend[q] = false;
            } else if (entq[q].key == 0) {
              pend[q] = false;
            }
            any_pending |= pend[q];
          }
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const uint32_t cq = c0 + 64u * q;
        if (cq >= ncand) break;
        c_probe += (ncand - cq) < 64 ? (ncand - cq) : 64;
        if (__ballot(nodeq[q] != GRIM_NONE) == 0) continue;  // nothing found in this chunk (the usual case)
        expand_chunk<false>(A, prior, L, st, nodeq[q], full_nodes, g.a_start, g.a_nbr, 1.0, 0, item_base, c_nbr, c_freq);
      }
    }
  } else {
    // label scan: every node of the typed-loci label whose alleles all belong to this side's
    // alternatives, in node-id order (impute.py:947-981, cutils.pyx:33-51).  The label's keys are streamed (lab_key);
    // membership is one bit per position while the lists are as typed (sh.abits), a walk over the list otherwise.
    const bool full_nodes = (mask == g.full_mask);
    const bool by_bits = ver == 0 && sh.ntok > 2u * (uint32_t)n;  // the bitsets exist and describe this version
    uint32_t a = g.lab_start[mask], b = g.lab_start[mask + 1];
    for (uint32_t j0 = a; j0 < b; j0 += 64 * 4) {
      uint64_t kq[4];  // four chunks of 64 keys in flight
#pragma unroll
      for (int q = 0; q < 4; ++q) kq[q] = j0 + 64u * q + lane < b ? g.lab_key[j0 + 64u * q + lane] : 0ull;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
      const uint32_t i0 = j0 + 64u * q;
      if (i0 >= b) break;
      uint32_t i = i0 + lane;
      bool ok = i < b;
      if (ok) {
        const uint64_t key = kq[q];
#pragma unroll
        for (int l = 0; l < GRIM_MAXL; ++l) {
          if (l < n && ok) {
            const uint32_t al = (uint32_t)((key >> (GRIM_ABITS * sl[l])) & 0xFFF) - 1u;
            if (by_bits) {
              const int c = (int)((pat >> l) & 1u) ^ side;
              ok = (sh.abits[l][c][(al >> 5) & 127u] >> (al & 31u)) & 1u;
            } else {
              bool hit = false;
              for (uint32_t t = 0; t < cn[l]; ++t) hit |= (tok[to[l] + t] == al);
              ok = hit;
            }
          }
        }
      }
      if (__ballot(ok) == 0) continue;
      any_cand = true;
      const uint32_t node = ok ? g.lab_nodes[i] : GRIM_NONE;
      expand_chunk<false>(A, prior, L, st, node, full_nodes, g.a_start, g.a_nbr, 1.0, 0, item_base, c_nbr, c_freq);
      }
    }
  }
  store_top<false>(S, sh, L, st, row);
  if (lane == 0) {
    sh.cand_any[row] = any_cand ? 1 : 0;
    sh.wctr[wave_id()][0] += c_probe;
    sh.wctr[wave_id()][1] += c_nbr;
    sh.wctr[wave_id()][2] += c_freq;
  }
}
