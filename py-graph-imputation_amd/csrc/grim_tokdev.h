// grim_tokdev.h -- GL strings -> subject records ON THE DEVICE, for the lines the half-wave kernel will take: every locus
// typed, one allele per side (no '/' list).  Restates, for that kind of line, clean_up_gl + gl2haps (impute.py:105-118,
// 246-272) exactly as the host's one-pass tokenizer does (tokenise_gl_fast, grim_host.cpp): a line it accepts gets the same
// subject record, tokens and half-wave record the host would have written; a line it does not accept is flagged and the
// host's general tokenizer takes it afterwards (grim_stream.cpp) -- never a different answer, at worst a second pass.
//
// Layout: TEN LANES PER LINE, lane a = the a-th allele of the GL string (position a / 2, side a % 2), six lines per wave.
// All ten lanes of a line walk the GL field together (16-byte loads of the same addresses: one fetch serves the ten) and
// check its shape -- nine separators, '+' and '^' alternating, no '/', no 'g' / 'L' for clean_up_gl to delete; each lane
// then packs ITS allele's bytes, finds the locus (the bytes before '*') and probes that locus's dictionary table (32-byte
// slots, name inline: one probe = one slot = two 16-byte loads), so the ten probes of a line are in flight together.  Pair
// checks go over lane shuffles: the two alleles of a position name the same locus, positions are in sorted order
// (gl2haps sorts each side's entries), no locus twice.
#pragma once
#include "grim_dev.h"
#include "grim_tok.h"

struct DevTok {
  const uint8_t *text;   // the chunk's text (padded: 16-byte loads may run past the last line)
  const LineRec *lines;  // the chunk's lines
  uint32_t lo, hi;       // lines [lo, hi) are tokenised (the whole chunk, or a part of it that is run on its own)
  DevDict dict;
  SmallRec *dsmall;      // [hi - lo] the half-wave kernel's records, slot k = line lo + k; si == GRIM_NONE: no device subject there
  grim_subject *subj;    // subject records (by line) for the kernels a subject may move on to (Plan B / C)
  uint16_t *tok;         // tokens of device line j: tok[tok_base + 10 j ...]
  uint32_t tok_base;
  uint32_t graph_loci;
};
#define GRIM_REASON_HOST_TOKENIZER 7  // grim_subject_result.reason of a line the device tokenizer hands back to the host
#define GRIM_Q_IRREGULAR 18           // queue word: lines handed back

#define TOK_LANES (2 * GRIM_MAXL)
#define TOK_LINES_PER_WAVE (64 / TOK_LANES)

__global__ __launch_bounds__(GRIM_WG) void grim_tokenize_kernel(DevArgs A, DevTok T) {
  const int lane = lane_id();
  const int a = lane % TOK_LANES, lw = lane / TOK_LANES;
  const uint32_t idx = (blockIdx.x * (GRIM_WG / 64) + wave_id()) * TOK_LINES_PER_WAVE + lw;  // record slot: line lo + idx
  const uint32_t line = T.lo + idx;
  const bool in_grid = lw < TOK_LINES_PER_WAVE && line < T.hi;
  LineRec lr;
  lr.gl_off = 0; lr.gl_len = 0; lr.prior_idx = 0;
  if (in_grid) lr = T.lines[line];
  const bool active = in_grid && lr.gl_len != 0;
  bool bad = active && lr.gl_len > GRIM_TOK_MAXGL;
  const uint32_t n = (active && !bad) ? lr.gl_len : 0;
  // ---- the GL field into LDS: the ten lanes of a line fetch 16 bytes each (one coalesced 160-byte window that starts at the
  // 16-byte boundary before the field), and every later step reads bytes from there ---------------------------------------
  __shared__ uint4 lbuf[GRIM_WG / 64][TOK_LINES_PER_WAVE + 1][TOK_LANES];
  uint4 *mybuf = lbuf[wave_id()][lw < TOK_LINES_PER_WAVE ? lw : TOK_LINES_PER_WAVE];
  const uint32_t a0 = lr.gl_off & ~15u, skip = lr.gl_off - a0, tot = n ? skip + n : 0;  // tot <= 15 + GRIM_TOK_MAXGL = 159
  uint4 q = make_uint4(0, 0, 0, 0);
  if (16u * (uint32_t)a < tot) q = ((const uint4 *)(T.text + a0))[a];
  mybuf[a] = q;
  // this lane's 16 bytes: where the separators are, which of them are '+', whether a character rules the line out
  uint32_t m_sep = 0, m_plus = 0;
  {
    const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int bb = 0; bb < 16; ++bb) {
      const uint32_t pos = 16u * (uint32_t)a + (uint32_t)bb;
      const uint32_t c = (wd[bb >> 2] >> (8 * (bb & 3))) & 0xFFu;
      const bool in = pos >= skip && pos < tot;
      if (in && c == '+') m_plus |= 1u << bb;
      if (in && (c == '+' || c == '^')) m_sep |= 1u << bb;
      if (in && (c == '/' || c == 'g' || c == 'L')) bad = true;  // a '/' list (not the half-wave kernel's subject) / what clean_up_gl deletes
    }
  }
  WAVE_SYNC();
  // separator number a - 1 opens this lane's allele, separator number a closes it ("x+y" per position, '^' between
  // positions: separator k must be '+' for even k, '^' for odd k; nine in all)
  uint32_t nsep = 0, start = 0, end = n;
  {
    const int base0 = lw * TOK_LANES;
#pragma unroll
    for (int v = 0; v < TOK_LANES; ++v) {
      const int src = base0 + v < 64 ? base0 + v : 0;
      uint32_t mv = (uint32_t)__shfl((int)m_sep, src);
      const uint32_t pv = (uint32_t)__shfl((int)m_plus, src);
      const uint32_t c = (uint32_t)__popc(mv);
      if (a > 0 && nsep < (uint32_t)a && nsep + c >= (uint32_t)a) {  // separator a - 1 is in this vector
        uint32_t m2 = mv;
        for (uint32_t k = nsep; k + 1 < (uint32_t)a; ++k) m2 &= m2 - 1;
        start = 16u * (uint32_t)v + (uint32_t)__builtin_ctz(m2) + 1u - skip;
      }
      if (nsep <= (uint32_t)a && nsep + c > (uint32_t)a) {  // separator a
        uint32_t m2 = mv;
        for (uint32_t k = nsep; k < (uint32_t)a; ++k) m2 &= m2 - 1;
        const uint32_t bit = (uint32_t)__builtin_ctz(m2);
        end = 16u * (uint32_t)v + bit - skip;
        if (((pv >> bit) & 1u) != ((a & 1) == 0 ? 1u : 0u)) bad = true;
      }
      nsep += c;
    }
    if (n && nsep != TOK_LANES - 1) bad = true;  // fewer or more than five positions of two sides
  }
  // ---- this lane's allele: packed name, locus ------------------------------------------------------------------------
  const uint32_t len = end > start ? end - start : 0;
  if (active && (len == 0 || len > GRIM_TOKNAME)) bad = true;
  uint32_t w[6] = {0, 0, 0, 0, 0, 0};
  uint32_t star = len;
  uint64_t locus = 0;
  if (active && !bad) {
    const uint8_t *p = (const uint8_t *)mybuf + skip + start;
    uint32_t cb[GRIM_TOKNAME];
#pragma unroll
    for (int k = 0; k < GRIM_TOKNAME; ++k) cb[k] = (uint32_t)k < len ? (uint32_t)p[k] : 0u;  // independent LDS byte reads
#pragma unroll
    for (int k = GRIM_TOKNAME - 1; k >= 0; --k)
      if ((uint32_t)k < len && cb[k] == '*') star = (uint32_t)k;  // the first '*'
#pragma unroll
    for (int k = 0; k < GRIM_TOKNAME; ++k) w[k >> 2] |= cb[k] << (24 - 8 * (k & 3));
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if ((uint32_t)k < star) locus |= (uint64_t)cb[k] << (56 - 8 * k);
    if (star > 8) bad = true;  // a locus name longer than the table holds (or no '*' in a long name)
    // an entry that starts or ends with 'U' ("A*UUUU"): clean_up_gl drops the position (impute.py:110-117)
    if (((a & 1) == 0 && cb[0] == 'U') || ((a & 1) == 1 && p[len - 1] == 'U')) bad = true;
  }
  int slot = -1;
  if (active && !bad) {
#pragma unroll
    for (int s = 0; s < GRIM_MAXL; ++s)
      if (s < (int)T.dict.n_loci && T.dict.locus_len[s] == star && T.dict.locus[s] == locus) slot = s;
    if (slot < 0) bad = true;
  }
  // ---- the dictionary probe (all lanes' first slots are in flight together) ----------------------------------------------
  uint32_t id = 0;
  if (active && !bad) {
    const uint32_t h = tokname_hash(w, len);
    const DictEnt *tab = T.dict.tab[slot];
    const uint32_t mask = T.dict.mask[slot];
    uint32_t k = h & mask;
    for (;;) {
      const uint4 e0 = ((const uint4 *)(tab + k))[0], e1 = ((const uint4 *)(tab + k))[1];
      if (!(e1.z >> 31)) {  // empty slot: the graph has never seen this allele -- the host numbers it (overlay ids)
        bad = true;
        break;
      }
      if (e1.w == h && ((e1.z >> 16) & 0xFFu) == len && e0.x == w[0] && e0.y == w[1] && e0.z == w[2] && e0.w == w[3] && e1.x == w[4] &&
          e1.y == w[5]) {
        id = e1.z & 0xFFFFu;
        break;
      }
      k = (k + 1) & mask;
    }
  }
  // ---- checks between the lanes of a line ------------------------------------------------------------------------------
  const int base = lw * TOK_LANES;
  uint32_t same = 0, used = 0, slots5 = 0;
  {
    // the partner of a position (the other side): same locus; identical text -> the position is homozygous as typed
    const int partner = lane ^ 1;
    const int pslot = __shfl(slot, partner);
    bool eq = __shfl((int)len, partner) == (int)len;
#pragma unroll
    for (int i = 0; i < 6; ++i) eq = eq && __shfl((int)w[i], partner) == (int)w[i];
    if (active && !bad && pslot != slot) bad = true;
    // sorted(): side s of position k must sort before side s of position k + 1 (gl2haps sorts each side's entries;
    // impute.py:271) -- big-endian packing makes the word-wise comparison the string comparison
    const int nxt = lane + 2;
    bool less = false, decided = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const uint32_t o = (uint32_t)__shfl((int)w[i], nxt < 64 ? nxt : lane);
      if (!decided && o != w[i]) {
        less = w[i] < o;
        decided = true;
      }
    }
    if (active && !bad && a + 2 < TOK_LANES && !(decided && less)) bad = true;
    // the five positions: which are homozygous as typed, which slots they name
#pragma unroll
    for (int k = 0; k < GRIM_MAXL; ++k) {
      const int src = base + 2 * k;
      const int sk = __shfl(slot, src < 64 ? src : 0);
      const int ek = __shfl(eq ? 1 : 0, src < 64 ? src : 0);
      if (sk >= 0) {
        used |= 1u << sk;
        slots5 |= (uint32_t)sk << (4 * k);
      }
      same |= (uint32_t)(ek & 1) << k;
    }
  }
  // one verdict per line: every lane fine, five different loci, the graph's five
  const uint64_t badm = __ballot(bad);
  const uint32_t line_bad = (uint32_t)(badm >> base) & ((1u << TOK_LANES) - 1u);
  const bool ok = active && line_bad == 0 && __popc(used) == GRIM_MAXL && T.graph_loci == GRIM_MAXL && T.dict.n_loci == GRIM_MAXL;
  // ---- outputs -------------------------------------------------------------------------------------------------------------
  if (!in_grid) return;
  SmallRec *rec = T.dsmall + idx;
  if (ok) {
    rec->tok[a] = (uint16_t)id;  // tok[2 l + side]: lane order
    T.tok[(uint64_t)T.tok_base + (uint64_t)TOK_LANES * line + a] = (uint16_t)id;
    if ((a & 1) == 0) rec->slot[a >> 1] = (uint8_t)slot;
  }
  if (a == 0) {
    if (ok) {
      rec->same = (uint8_t)same;
      rec->prior_idx = lr.prior_idx;
      rec->si = line;
      grim_subject sj;
      sj.tok_off = T.tok_base + TOK_LANES * line;
      sj.prior_idx = lr.prior_idx;
      sj.n_loci = GRIM_MAXL;
      sj.flags = 0;
#pragma unroll
      for (int k = 0; k < GRIM_MAXL; ++k) {
        sj.slot[k] = (uint8_t)((slots5 >> (4 * k)) & 0xFu);
        sj.cnt[k][0] = sj.cnt[k][1] = 1;
        sj.wid[k][0] = sj.wid[k][1] = 1;
      }
      sj.pad[0] = (uint8_t)same;
      sj.pad[1] = sj.pad[2] = 0;
      sj.reserved[0] = sj.reserved[1] = 0;
      T.subj[line] = sj;
    } else {
      rec->si = GRIM_NONE;
      if (active) {  // handed back to the host's tokenizer
        uint32_t *r = (uint32_t *)(A.res + line);
        r[0] = (uint32_t)GRIM_ST_UNSUPPORTED | ((uint32_t)'a' << 8) | ((uint32_t)GRIM_REASON_HOST_TOKENIZER << 16);
        atomicAdd(A.queue + GRIM_Q_IRREGULAR, 1u);
      }
    }
  }
}
