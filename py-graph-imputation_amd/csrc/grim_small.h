// grim_small.h -- Plan A for the simplest and by far most common subject class: every locus typed,
// no '/' ambiguity, one population (BASELINE configs 2 and 3).  Same semantics as
// grim_plan_a_kernel, specialised so that ONE HALF WAVE handles a subject entirely in registers:
//
//   * the 2^5 = 32 ways of picking, per locus, the allele of side 1 or side 2 are exactly the 32
//     candidate haplotypes of all 16 phases x 2 sides (gen_phases, impute.py:274-303), so lane c of
//     the half wave owns combination c: key -> hash probe -> node -> frequency (coalesced across
//     the two subjects of the wave, no LDS);
//   * phase i pairs combination i with its complement i^31 (a lane shuffle); with one population a
//     top list has one entry, so there is one haplotype pair per phase and the reference's
//     (phase,h,k) order is the lane order;
//   * ladder, MaxProb, final epsilon, left-to-right sums and the stable ranking are wave
//     reductions / shuffles over those <=16 lanes.
//
// Rows go to a pre-assigned fixed-stride region of the row pool (no atomics on the hot path).
#pragma once
#include "grim_pair.h"


__device__ __forceinline__ double half_shfl_d(double v, int src_in_half) {
  return __shfl(v, (threadIdx.x & 32) | src_in_half);
}
__device__ __forceinline__ uint32_t half_shfl_u(uint32_t v, int src_in_half) {
  return __shfl(v, (threadIdx.x & 32) | src_in_half);
}

// max over the 16 lanes of a DPP row (lanes 0-15 / 16-31 / 32-47 / 48-63), left in every lane of the row:
// four row rotations, no LDS crossbar
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)((unsigned long long)b >> 32), CTRL, 0xF, 0xF, false);
  return __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo));
}
__device__ __forceinline__ double row16_max(double v) {
  double o;
  o = dpp_d<0x121>(v); v = o > v ? o : v;  // row_ror:1
  o = dpp_d<0x122>(v); v = o > v ? o : v;  // row_ror:2
  o = dpp_d<0x124>(v); v = o > v ? o : v;  // row_ror:4
  o = dpp_d<0x128>(v); v = o > v ? o : v;  // row_ror:8
  return v;
}

// SmallRec (grim_layout.h): what the library keeps in HBM per fast-path subject, 32 bytes
__global__ __launch_bounds__(GRIM_WG) void grim_plan_a_small_kernel(DevArgs A, const SmallRec *recs, uint32_t n,
                                                                   uint32_t row_base, uint32_t row_stride) {
  const DevGraph &g = A.g;
  const int hl = threadIdx.x & 31;            // lane inside the half wave = combination
  const uint32_t w = blockIdx.x * (GRIM_WG / 32) + (threadIdx.x >> 5);
  bool live = w < n;
  const int n_ladder = A.prm.n_ladder;
  double lad[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) lad[i] = A.prm.ladder[i];
  // ---- candidate of this lane: one 32-byte record (same address for the half wave), one probe ----
  bool hit = false;  // this lane's combination is a haplotype of the graph
  uint64_t mykey = 0;
  double f = 0.0;
  uint32_t same = 0, si = 0;
  double w_prior = 0.0;
  if (live) {
    const uint4 *rp = (const uint4 *)(recs + w);
    const uint4 r0 = rp[0], r1 = rp[1];
    const uint32_t words[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    // tok[k] = half k&1 of word k>>1; slot[l] = byte l of bytes 20..24; same = byte 25; prior = bytes 26..27
    same = (words[6] >> 8) & 0xFFu;
    si = words[7];
    live = si != GRIM_NONE;  // a record slot of the device tokenizer that holds no subject (grim_tokdev.h)
    w_prior = live ? A.priors[(uint64_t)(words[6] >> 16)] : 0.0;  // P == 1: the matrix is one number
#pragma unroll
    for (int l = 0; l < GRIM_MAXL; ++l) {
      const uint32_t tw = words[l];                       // tok[2l] | tok[2l+1] << 16
      const uint32_t t = ((hl >> l) & 1) ? (tw >> 16) : (tw & 0xFFFFu);
      const uint32_t slot = (l < 4 ? (words[5] >> (8 * l)) : words[6]) & 0xFFu;
      mykey |= (uint64_t)(t + 1u) << (GRIM_ABITS * slot);
    }
    uint32_t h = fht_hash(mykey) & g.fht_mask;
    for (; live;) {
      const FullEnt e = g.fht[h];
      if (e.key == mykey) { hit = true; f = e.f0; break; }
      if (e.key == 0) break;
      h = (h + 1) & g.fht_mask;
    }
  }
  // ---- phases: lane i < 16 is phase i, its partner is the complementary combination ---------
  const bool hit2 = half_shfl_u(hit ? 1u : 0u, hl ^ 31) != 0;
  const double f2 = half_shfl_d(f, hl ^ 31);
  const uint64_t key2 = ((uint64_t)half_shfl_u((uint32_t)(mykey >> 32), hl ^ 31) << 32) | half_shfl_u((uint32_t)mykey, hl ^ 31);
  const uint32_t het4 = 15u & ~same;
  bool kept = hl < 16 && ((uint32_t)hl & ~het4 & 15u) == 0 && (!((same >> 4) & 1u) || ((uint32_t)hl ^ het4) >= (uint32_t)hl);
  // entries need p > 0 (impute.py:430); a missing node has f == 0
  const bool pair_ok = live && kept && f > 0.0 && f2 > 0.0;
  const bool same_hap = mykey == key2;  // names are unique: same key, same haplotype
  PairRef pr;
  pr.p1 = f; pr.p2 = f2; pr.m2 = f2;
  pr.e1 = same_hap ? 0u : 1u; pr.e2 = 0u;  // only equality of the two haplotypes matters below
  // ---- ladder: first step that accepts any pair (impute.py:1665-1687).  Acceptance is monotone in
  // epsilon, so the subject's first step is the first at which ANY of its lanes accepts: the loop ends
  // as soon as both halves have one (scalar ballots, no per-lane search to the lane's own first step).
  const bool hi_half = (threadIdx.x & 32) != 0;
  double eps = 0.0;
  bool found = false;
  bool acc = false;  // this lane's verdict at the step that decided its half
  {
    const uint64_t okb = __ballot(pair_ok);
    bool need_lo = (okb & 0xFFFFull) != 0, need_hi = ((okb >> 32) & 0xFFFFull) != 0;
    bool got_lo = false, got_hi = false;
    double eps_lo = 0.0, eps_hi = 0.0;
    auto step = [&](double e) {
      const bool a = pair_ok && pair_accept(e, pr, w_prior);
      const uint64_t b = __ballot(a);
      const bool take_lo = need_lo && (b & 0xFFFFull), take_hi = need_hi && ((b >> 32) & 0xFFFFull);
      if (take_lo) { need_lo = false; got_lo = true; eps_lo = e; }
      if (take_hi) { need_hi = false; got_hi = true; eps_hi = e; }
      if (hi_half ? take_hi : take_lo) acc = a;
    };
    // the first eight steps sit in scalar registers (loaded at kernel start, under the probe's latency)
#pragma unroll
    for (int idx = 0; idx < 8; ++idx)
      if (idx < n_ladder && (need_lo || need_hi)) step(lad[idx]);
    for (int idx = 8; idx < n_ladder && (need_lo || need_hi); ++idx) step(A.prm.ladder[idx]);
    eps = hi_half ? eps_hi : eps_lo;
    found = hi_half ? got_hi : got_lo;
  }
  double prob = acc ? pair_prob(pr, w_prior) : 0.0;
  double mx = row16_max(prob);  // the 16 phase lanes of a half are one DPP row
  if (found && eps > 0.0) {
    eps = mx / 100000.0;  // impute.py:1685, then the last round
    acc = pair_ok && pair_accept(eps, pr, w_prior);
    prob = acc ? pair_prob(pr, w_prior) : 0.0;
    mx = row16_max(prob);
  }
  const uint64_t bal = __ballot(acc);
  const uint32_t accmask = (uint32_t)(hi_half ? (bal >> 32) : bal) & 0xFFFFu;
  const uint32_t nU = __popc(accmask);
  // ---- sums in phase order; one genotype and one population cell hold every pair.  Stable ranking of
  // the phased pairs: bigger first, earlier phase first on ties.  Only phases some half accepted are
  // visited (1-4 of the 16, typically); their values come over v_readlane.
  double total = 0.0;
  bool first = true;
  uint32_t rank = 0;
  for (uint32_t um = ((uint32_t)bal | (uint32_t)(bal >> 32)) & 0xFFFFu; um; um &= um - 1) {
    const int i = __builtin_ctz(um);
    const double p_lo = lane_get(prob, i), p_hi = lane_get(prob, 32 + i);
    const double pi = hi_half ? p_hi : p_lo;
    if ((accmask >> i) & 1u) {
      total = first ? pi : total + pi;
      first = false;
      if (pi > prob || (pi == prob && i < hl)) ++rank;
    }
  }
  // algorithmic byte counters: 32 probes, one frequency vector per hit.  One plain store per wave into
  // its own cell (atomics on shared cells serialised at the L2 and cost a third of the kernel).
  {
    // a kept phase looks up its two sides; combinations of dropped (duplicate) phases do not count
    const uint64_t need = __ballot(live && kept), needp = __ballot(live && kept && hit),
                   needq = __ballot(live && kept && hit2);
    if ((threadIdx.x & 63) == 0) {
      const uint32_t gw = blockIdx.x * (GRIM_WG / 64) + (threadIdx.x >> 6);
      *(uint2 *)(A.small_ctr + 2 * gw) = make_uint2(2u * (uint32_t)__popcll(need), (uint32_t)(__popcll(needp) + __popcll(needq)));
    }
  }
  if (!live) return;
  // ---- output, by lane role: one row store and one header store serve the whole half ------------------------------
  //   hl  0..15  phase lanes       -> their .pmug row (rank-th of the subject's block)
  //   hl 16                        -> the fixed row: .umug, .umug.pops and .pmug.pops in one (GRIM_SMALL_ROWS_FIXED)
  //   hl 19..25                    -> the seven 8-byte pieces of the 56-byte result header
  const uint32_t off = row_base + w * row_stride;
  const bool ok = nU > 0;
  const int first_lane = ok ? (__ffs(accmask) - 1) : 0;
  const uint32_t n_pm = (ok && A.prm.out_haps) ? (nU < A.prm.n_results ? nU : A.prm.n_results) : 0;
  // the first accepted pair names the (one) genotype
  const uint64_t ka = ((uint64_t)half_shfl_u((uint32_t)(mykey >> 32), first_lane) << 32) | half_shfl_u((uint32_t)mykey, first_lane);
  const uint64_t kb = ((uint64_t)half_shfl_u((uint32_t)(key2 >> 32), first_lane) << 32) | half_shfl_u((uint32_t)key2, first_lane);
  {
    const int f = hl - 16;
    const bool phase = hl < 16;
    const bool wr = ok && (phase ? (acc && rank < n_pm) : f == 0);
    if (wr) {
      grim_row r;
      r.a = phase ? mykey : ka;  // a found node's key is the key that found it
      r.b = phase ? key2 : kb;
      r.prob = phase ? prob : total;
      r.popa = 0;
      r.popb = 0;
      A.rows[off + (phase ? GRIM_SMALL_ROWS_FIXED + rank : (uint32_t)f)] = r;
    }
  }
  if (hl >= 19 && hl < 26) {
    const int j = hl - 19;
    const uint8_t status = ok ? GRIM_ST_OK : (A.prm.planb ? GRIM_ST_UNSUPPORTED : GRIM_ST_MISS);  // UNSUPPORTED: the plan-B kernel overwrites it
    const uint32_t head = (uint32_t)status | ((uint32_t)'a' << 8) | ((!ok && A.prm.planb) ? (2u << 16) : 0u);
    const uint32_t n_um = (ok && A.prm.out_muug && A.prm.n_results) ? 1u : 0u;
    const uint32_t n_up = (ok && A.prm.out_muug && A.prm.n_pop_results) ? 1u : 0u;
    const uint32_t n_pp = (ok && A.prm.out_haps && A.prm.n_pop_results) ? 1u : 0u;
    const uint64_t mxb = ok ? (uint64_t)__double_as_longlong(mx) : 0ull;
    uint32_t lo, hi;
    // dwords: 0 status/plan/reason/plan_phased  1 n_pairs  2 n_genotypes  3-6 row_off  7-10 n_rows  11 pad  12-13 max_prob
    switch (j) {
      case 0: lo = head; hi = ok ? nU : 0u; break;
      case 1: lo = ok ? 1u : 0u; hi = ok ? off + 0 : 0u; break;
      case 2: lo = ok ? off + 0 : 0u; hi = ok ? off + GRIM_SMALL_ROWS_FIXED : 0u; break;
      case 3: lo = ok ? off + 0 : 0u; hi = n_um; break;
      case 4: lo = n_up; hi = n_pm; break;
      case 5: lo = n_pp; hi = 0u; break;
      default: lo = (uint32_t)mxb; hi = (uint32_t)(mxb >> 32); break;
    }
    ((uint2 *)(A.res + si))[j] = make_uint2(lo, hi);
  }
  if (!ok && hl == 0 && A.prm.planb) push_next(A, si, false);  // every locus typed: never a heavy one
}
