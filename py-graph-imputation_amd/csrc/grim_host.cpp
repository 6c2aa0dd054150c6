// grim_host.cpp -- host side of the boundary, in C++: allele dictionary, GL tokenizer, result formatter, prior matrices.
//
// Replaces, for whole files at a time and on all host cores, what the reference does per line in
// Python (it is the end-to-end bottleneck once the kernels are fast):
//   impute_file line handling      impute.py:2022-2036   (rstrip, ',' or '%' split, id / GL / races)
//   clean_up_gl                    impute.py:105-118
//   gl2haps                        impute.py:246-272
//   calc_priority_matrix           impute.py:1844-1924
//   write_best_prob*, .miss/.problem rules, str(float)   impute.py:24-99, 2061-2118
// No GPU code here.  The cores (tokenize_range / format_range) work on a byte range of whole lines and are what the
// streaming pipeline (grim_stream.cpp) runs on its worker threads; grim_tokenize / grim_format are the whole-block
// entry points built on the same cores.  Same outcomes as grim/imputation/impute.py::_tokenise / _write_rows (the
// Python versions stay as the cross-check in tests/).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/grim_hip.h"

#include "grim_host_internal.h"

extern "C" grim_dict *grim_dict_create(uint32_t n_loci) {
  if (n_loci == 0 || n_loci > GRIM_MAXL) return nullptr;
  grim_dict *d = new grim_dict();
  d->n_loci = n_loci;
  d->locus_name.resize(n_loci);
  d->ids.resize(n_loci);
  d->names.resize(n_loci);
  return d;
}

extern "C" void grim_dict_free(grim_dict *d) { delete d; }

extern "C" int grim_dict_set_locus(grim_dict *d, uint32_t slot, const char *name) {
  if (!d || slot >= d->n_loci) return -1;
  d->locus_name[slot] = name;
  d->locus_slot[name] = slot;
  return 0;
}

int32_t dict_intern(grim_dict *d, uint32_t slot, sv a) {
  auto &m = d->ids[slot];
  std::string key(a);
  auto it = m.find(key);
  if (it != m.end()) return (int32_t)it->second;
  uint32_t id = (uint32_t)d->names[slot].size();
  if (id >= (1u << GRIM_ABITS) - 2) return -1;
  m.emplace(key, id);
  d->names[slot].push_back(std::move(key));
  return (int32_t)id;
}

extern "C" int32_t grim_dict_intern(grim_dict *d, uint32_t slot, const char *allele) {
  if (!d || slot >= d->n_loci) return -1;
  return dict_intern(d, slot, sv(allele));
}

extern "C" int32_t grim_dict_find(const grim_dict *d, uint32_t slot, const char *allele) {
  if (!d || slot >= d->n_loci) return -1;
  auto it = d->ids[slot].find(allele);
  return it == d->ids[slot].end() ? -1 : (int32_t)it->second;
}

extern "C" const char *grim_dict_name(const grim_dict *d, uint32_t slot, uint32_t id) {
  if (!d || slot >= d->n_loci || id >= d->names[slot].size()) return nullptr;
  return d->names[slot][id].c_str();
}

extern "C" uint32_t grim_dict_count(const grim_dict *d, uint32_t slot) {
  return (d && slot < d->n_loci) ? (uint32_t)d->names[slot].size() : 0;
}

void dict_snapshot(const grim_dict *d, DictSnap &S) {
  S = DictSnap();
  S.n_loci = d->n_loci;
  size_t total = 0;
  for (uint32_t s = 0; s < d->n_loci; ++s)
    for (const std::string &n : d->names[s]) total += n.size();
  S.pool.reserve(total + 16);
  for (uint32_t s = 0; s < d->n_loci; ++s) {
    const auto &nm = d->names[s];
    S.base[s] = (uint32_t)nm.size();
    S.name_off[s].resize(nm.size() + 1);
    uint32_t cap = 16;
    while (cap < 2 * nm.size() + 2) cap <<= 1;
    S.tab[s].assign(cap, DictSnap::Ent{0, 0, 0, 0, 0});
    S.mask[s] = cap - 1;
    for (size_t i = 0; i < nm.size(); ++i) {
      S.name_off[s][i] = (uint32_t)S.pool.size();
      S.pool += nm[i];
    }
    S.name_off[s][nm.size()] = (uint32_t)S.pool.size();
  }
  S.pool.append(16, '\0');  // 8-byte loads of the last names stay inside the buffer
  for (uint32_t s = 0; s < d->n_loci; ++s)
    for (size_t i = 0; i < d->names[s].size(); ++i) {
      const sv n = S.name(s, (uint32_t)i);
      const uint64_t h = DictSnap::hash(n);
      uint32_t k = (uint32_t)h & S.mask[s];
      while (S.tab[s][k].used) k = (k + 1) & S.mask[s];
      S.tab[s][k] = DictSnap::Ent{h, (uint32_t)i, S.name_off[s][i], (uint32_t)n.size(), 1};
    }
  bool alnum = true;
  for (const auto &kv : d->locus_slot) {
    S.loci.push_back({kv.first, kv.second});
    for (char c : kv.first) alnum = alnum && ((c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'));
  }
  // sorted() of a haplotype's allele names: names of different loci differ inside "<locus>*", and '*' sorts before
  // every alphanumeric character, so the order is the order of the strings "<locus>*" -- the same for every haplotype
  std::vector<std::pair<std::string, uint32_t>> by;
  for (uint32_t s = 0; s < d->n_loci; ++s) by.emplace_back(d->locus_name[s] + "*", s);
  std::sort(by.begin(), by.end());
  for (uint32_t s = 0; s < d->n_loci; ++s) S.order[s] = (uint8_t)by[s].second;
  S.fixed_order = alnum;
}

// ------------------------------------------------------------------------------------------------
// tokenizer
// ------------------------------------------------------------------------------------------------
static inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

namespace {

struct Scratch {  // per thread, reused line after line
  std::string clean;
  std::vector<sv> parts, side1, side2;
  std::string last_r1s, last_r2s;
  uint32_t last_race = 0;
  bool have_last = false;
  // race pairs this range has seen: the shared table (a mutex) is asked once per distinct pair and range
  std::unordered_map<std::string, uint32_t> race_cache;
  std::string race_key;
};

}  // namespace

static inline void insertion_sort(std::vector<sv> &v) {
  for (size_t i = 1; i < v.size(); ++i) {
    sv x = v[i];
    size_t j = i;
    while (j > 0 && x < v[j - 1]) {
      v[j] = v[j - 1];
      --j;
    }
    v[j] = x;
  }
}

// one GL string -> kind; on K_DEV the subject record is in `sj` and its tokens were appended at tok[n_tok...]
static int tokenise_gl(const DictSnap &D, sv gl, bool planb, Scratch &sc, grim_subject &sj, uint16_t *tok, uint64_t tok_room,
                       uint64_t &n_tok, std::vector<OvEnt> &ov, std::string &ov_pool, uint32_t line_no, bool &tok_overflow) {
  if (gl.empty()) return K_PROBLEM_ID;
  // clean_up_gl: drop every 'g' and 'L', then the entries that start or end with 'U'
  if (memchr(gl.data(), 'g', gl.size()) || memchr(gl.data(), 'L', gl.size())) {
    sc.clean.clear();
    for (char c : gl)
      if (c != 'g' && c != 'L') sc.clean.push_back(c);
    gl = sv(sc.clean);
  }
  std::vector<sv> &parts = sc.parts;
  parts.clear();
  {
    size_t a = 0;
    for (;;) {
      const void *q = a < gl.size() ? memchr(gl.data() + a, '^', gl.size() - a) : nullptr;
      size_t b = q ? (size_t)((const char *)q - gl.data()) : gl.size();
      sv e = gl.substr(a, b - a);
      const bool drop = !e.empty() && (e.front() == 'U' || e.back() == 'U');
      if (!drop) parts.push_back(e);
      if (!q) break;
      a = b + 1;
    }
  }
  if (parts.empty()) return K_PROBLEM_ID;  // cleaned == ""
  if (parts.size() == 1 && (parts[0].empty() || parts[0] == " ")) return K_PROBLEM_ID;
  std::vector<sv> &side1 = sc.side1, &side2 = sc.side2;
  side1.clear();
  side2.clear();
  size_t blanks = 0;
  for (sv p : parts) {
    if (p.empty()) return K_PROBLEM_RAW;  // p[0] -> IndexError in the reference
    if (p[0] == '+') p = p.substr(1);
    const void *q = p.empty() ? nullptr : memchr(p.data(), '+', p.size());
    if (!q) {
      if (p.empty()) {
        ++blanks;
        continue;
      }
      return K_PROBLEM_ID;
    }
    const size_t c = (size_t)((const char *)q - p.data());
    side1.push_back(p.substr(0, c));
    sv rest = p.substr(c + 1);
    const void *q2 = rest.empty() ? nullptr : memchr(rest.data(), '+', rest.size());
    side2.push_back(q2 ? rest.substr(0, (size_t)((const char *)q2 - rest.data())) : rest);
  }
  const size_t n = parts.size() - blanks;
  if (n != side1.size() || n < 1) return K_PROBLEM_RAW;
  if (n > D.n_loci) return K_UNSUPPORTED_GL;  // more entries than the graph has loci (a locus named twice among them, or a locus the
                                              // graph does not have): the reference opens 2^(n-1) phases over them all the same
  insertion_sort(side1);
  insertion_sort(side2);
  memset(&sj, 0, sizeof(sj));
  const uint64_t tok_mark = n_tok;
  const size_t ov_mark = ov.size(), pool_mark = ov_pool.size();
  bool unknown_locus = false;
  uint32_t used = 0, npos = 0;
  uint32_t n_over[GRIM_MAXL] = {0, 0, 0, 0, 0};
  for (size_t k = 0; k < n; ++k) {
    const sv ent[2] = {side1[k], side2[k]};
    // every alternative of both sides must name the same locus
    sv locus;
    bool first = true, mixed = false;
    for (int s = 0; s < 2 && !mixed; ++s) {
      size_t a = 0;
      for (;;) {
        const void *q = a < ent[s].size() ? memchr(ent[s].data() + a, '/', ent[s].size() - a) : nullptr;
        const size_t b = q ? (size_t)((const char *)q - ent[s].data()) : ent[s].size();
        const sv al = ent[s].substr(a, b - a);
        const void *st = al.empty() ? nullptr : memchr(al.data(), '*', al.size());
        const sv l = st ? al.substr(0, (size_t)((const char *)st - al.data())) : al;
        if (first) {
          locus = l;
          first = false;
        } else if (l != locus) {
          mixed = true;
          break;
        }
        if (!q) break;
        a = b + 1;
      }
    }
    if (mixed) goto unsupported_gl;
    {
      const int32_t slot_i = D.find_locus(locus);
      if (slot_i < 0) {
        unknown_locus = true;
        continue;
      }
      const uint32_t slot = (uint32_t)slot_i;
      if ((used >> slot) & 1u) goto unsupported_gl;
      used |= 1u << slot;
      sj.slot[npos] = (uint8_t)slot;
      if (ent[0] == ent[1]) sj.pad[0] |= (uint8_t)(1u << k);
      for (int s = 0; s < 2; ++s) {
        const uint64_t start = n_tok;
        uint32_t cnt = 0, wid = 0;
        size_t a = 0;
        for (;;) {
          const void *q = a < ent[s].size() ? memchr(ent[s].data() + a, '/', ent[s].size() - a) : nullptr;
          const size_t b = q ? (size_t)((const char *)q - ent[s].data()) : ent[s].size();
          const sv al = ent[s].substr(a, b - a);
          ++wid;
          int32_t id = D.find(slot, al);
          if (id < 0) {
            // not a dictionary allele: the subject's own id for it (same text, same id)
            for (size_t o = ov_mark; o < ov.size(); ++o) {
              const OvEnt &e = ov[o];
              if (e.slot == slot && e.len == al.size() && memcmp(ov_pool.data() + e.off, al.data(), al.size()) == 0) {
                id = e.id;
                break;
              }
            }
            if (id < 0) {
              const uint32_t nid = D.base[slot] + n_over[slot];
              if (nid >= (1u << GRIM_ABITS) - 2) {
                // more distinct alleles at one locus than a key field holds: reported as unsupported (reason 5)
                n_tok = tok_mark;
                ov.resize(ov_mark);
                ov_pool.resize(pool_mark);
                return K_UNSUPPORTED;
              } else {
                ++n_over[slot];
                id = (int32_t)nid;
                ov.push_back(OvEnt{line_no, (uint16_t)nid, (uint8_t)slot, 0, (uint32_t)al.size(), (uint64_t)ov_pool.size()});
                ov_pool.append(al.data(), al.size());
              }
            }
          }
          bool dup = false;
          for (uint64_t t = start; t < n_tok && !dup; ++t) dup = tok[t] == (uint16_t)id;
          if (!dup) {
            if (n_tok >= tok_room) {
              tok_overflow = true;
              goto irregular;
            }
            tok[n_tok++] = (uint16_t)id;
            ++cnt;
          }
          if (!q) break;
          a = b + 1;
        }
        sj.cnt[npos][s] = (uint16_t)(cnt > 65535 ? 65535 : cnt);
        sj.wid[npos][s] = (uint16_t)(wid > 65535 ? 65535 : wid);
      }
      ++npos;
    }
  }
  if (unknown_locus) {
    n_tok = tok_mark;
    ov.resize(ov_mark);
    ov_pool.resize(pool_mark);
    return planb ? K_PROBLEM_RAW : K_MISS_NO_DEVICE;  // KeyError in Plan B / plain miss (see _tokenise)
  }
  sj.n_loci = (uint8_t)npos;
  return K_DEV;
irregular:
  n_tok = tok_mark;
  ov.resize(ov_mark);
  ov_pool.resize(pool_mark);
  return K_PROBLEM_RAW;
unsupported_gl:
  // a locus named twice, or two loci in one entry (after the per-side sort): the reference has no locus check and goes on
  // with what gl2haps paired by index (impute.py:246-272) -- rows or a raw line, depending on the plan that answers
  n_tok = tok_mark;
  ov.resize(ov_mark);
  ov_pool.resize(pool_mark);
  return K_UNSUPPORTED_GL;
}

// The regular line -- "L*x+L*y^..." with '/' lists, every locus and allele known, parts already in sorted order, no 'g' /
// 'L' / 'U' clean-up needed -- in ONE pass over the bytes (the general path below costs ~50 memchr calls on 10-byte
// ranges per line).  false: not that kind of line, nothing was changed, tokenise_gl decides.  Same outputs as
// tokenise_gl on every line it accepts (tests/test_host_cpp.py holds both against the Python twin).
struct GlClass {
  uint8_t c[256];
  GlClass() {
    memset(c, 0, sizeof(c));
    c[(uint8_t)'/'] = c[(uint8_t)'+'] = c[(uint8_t)'^'] = 1;
    c[(uint8_t)'g'] = c[(uint8_t)'L'] = 2;
    c[(uint8_t)'*'] = 3;
  }
};
static const GlClass kGlClass;
static bool tokenise_gl_fast(const DictSnap &D, sv gl, grim_subject &sj, uint16_t *tok, uint64_t tok_room, uint64_t &n_tok) {
  const char *p = gl.data();
  const size_t n = gl.size();
  if (n == 0 || n > 60000) return false;
  struct Al {
    uint16_t off, len, star;
  };
  struct Side {
    uint16_t off, len, a0, na;
  };
  Al al[96];
  Side side[2 * GRIM_MAXL];
  uint32_t n_al = 0, nparts = 0;
  size_t i = 0;
  for (;;) {
    if (nparts >= GRIM_MAXL || nparts >= D.n_loci) return false;
    const size_t ps = i;
    for (int sd = 0; sd < 2; ++sd) {
      Side &S = side[2 * nparts + sd];
      S.off = (uint16_t)i;
      S.a0 = (uint16_t)n_al;
      S.na = 0;
      for (;;) {
        const size_t as = i;
        size_t star = (size_t)-1;
        while (i < n) {
          const uint8_t k = kGlClass.c[(uint8_t)p[i]];  // 0: part of a name
          if (k) {
            if (k == 1) break;          // '/', '+', '^'
            if (k == 2) return false;   // 'g', 'L': clean_up_gl has work to do
            if (star == (size_t)-1) star = i - as;  // '*'
          }
          ++i;
        }
        if (i == as || n_al >= 96) return false;
        al[n_al].off = (uint16_t)as;
        al[n_al].len = (uint16_t)(i - as);
        al[n_al].star = (uint16_t)(star == (size_t)-1 ? i - as : star);
        ++n_al;
        ++S.na;
        if (i < n && p[i] == '/') {
          ++i;
          continue;
        }
        break;
      }
      S.len = (uint16_t)(i - S.off);
      if (sd == 0) {
        if (i >= n || p[i] != '+') return false;
        ++i;
      } else if (i < n && p[i] == '+') {
        return false;
      }
    }
    if (p[ps] == 'U' || p[i - 1] == 'U') return false;
    ++nparts;
    if (i >= n) break;
    ++i;  // '^'
    if (i >= n) return false;
  }
  // sorted(): the parts must already be in the order sorting each side's strings gives
  for (uint32_t k = 0; k + 1 < nparts; ++k)
    for (int sd = 0; sd < 2; ++sd) {
      const Side &a = side[2 * k + sd], &b = side[2 * (k + 1) + sd];
      if (p[a.off] < p[b.off]) continue;  // (the usual case: another locus, another first letter)
      if (p[a.off] > p[b.off] || !(sv(p + a.off, a.len) < sv(p + b.off, b.len))) return false;
    }
  memset(&sj, 0, sizeof(sj));
  const uint64_t tok_mark = n_tok;
  uint32_t used = 0;
  for (uint32_t k = 0; k < nparts; ++k) {
    const Side &s1 = side[2 * k], &s2 = side[2 * k + 1];
    const Al &f = al[s1.a0];
    const sv locus(p + f.off, f.star);
    for (uint32_t q = s1.a0; q < (uint32_t)s2.a0 + s2.na; ++q)
      if (al[q].star != f.star || !DictSnap::same_bytes(p + al[q].off, locus.data(), f.star)) return n_tok = tok_mark, false;
    const int32_t slot_i = D.find_locus(locus);
    if (slot_i < 0 || ((used >> slot_i) & 1u)) return n_tok = tok_mark, false;
    const uint32_t slot = (uint32_t)slot_i;
    used |= 1u << slot;
    sj.slot[k] = (uint8_t)slot;
    if (s1.len == s2.len && DictSnap::same_bytes(p + s1.off, p + s2.off, s1.len)) sj.pad[0] |= (uint8_t)(1u << k);
    for (int sd = 0; sd < 2; ++sd) {
      const Side &S = sd ? s2 : s1;
      const uint64_t start = n_tok;
      uint32_t cnt = 0;
      for (uint32_t q = S.a0; q < (uint32_t)S.a0 + S.na; ++q) {
        const int32_t id = D.find(slot, sv(p + al[q].off, al[q].len));
        if (id < 0) return n_tok = tok_mark, false;  // an allele the graph does not know: the general path numbers it
        bool dup = false;
        for (uint64_t t = start; t < n_tok && !dup; ++t) dup = tok[t] == (uint16_t)id;
        if (!dup) {
          if (n_tok >= tok_room) return n_tok = tok_mark, false;
          tok[n_tok++] = (uint16_t)id;
          ++cnt;
        }
      }
      sj.cnt[k][sd] = (uint16_t)cnt;
      sj.wid[k][sd] = S.na;
    }
  }
  sj.n_loci = (uint8_t)nparts;
  return true;
}

void tokenize_range(const TokParams &prm, const char *text, uint64_t lo, uint64_t hi, TokRange &R) {
  const DictSnap &D = *prm.snap;
  Scratch sc;
  const bool fast = !getenv("GRIM_NO_FAST_TOKENIZER");  // tests: every line through the general path
  std::vector<uint16_t> tmp_tok;  // dense mode: the current subject's tokens
  uint32_t line_no = 0;
  uint64_t a = lo;
  while (a < hi) {
    const char *nl = (const char *)memchr(text + a, '\n', hi - a);
    uint64_t b = nl ? (uint64_t)(nl - text) : hi;
    const uint64_t next = b + 1;
    while (b > a && is_space(text[b - 1])) --b;  // rstrip
    const sv line(text + a, b - a);
    LineInfo li{a, (uint32_t)(b - a), 0};
    int kind;
    // ',' when the line has one, else '%' (impute.py:2024-2027); fields: id, GL[, race1, race2, ...]
    const char sep = memchr(line.data(), ',', line.size()) ? ',' : '%';
    size_t s[4];
    int ns = 0;
    {
      size_t p = 0;
      while (ns < 4) {
        const void *q = p < line.size() ? memchr(line.data() + p, sep, line.size() - p) : nullptr;
        if (!q) break;
        s[ns++] = (size_t)((const char *)q - line.data());
        p = s[ns - 1] + 1;
      }
    }
    uint32_t race = 0;
    grim_subject sj;
    uint64_t tok_before = R.n_tok;
    bool tok_overflow = false, dev_line = false;
    if (prm.line_dst) prm.line_dst[line_no] = LineRec{0, 0, 0};
    if (ns == 0 || ns == 2) {  // fewer than two fields, or exactly three: IndexError in the reference -> raw line
      kind = K_PROBLEM_RAW;
      li.id_len = (uint32_t)(ns ? s[0] : line.size());
    } else {
      li.id_len = (uint32_t)s[0];
      sv r1, r2;
      if (ns >= 3) {
        r1 = line.substr(s[1] + 1, s[2] - s[1] - 1);
        r2 = ns >= 4 ? line.substr(s[2] + 1, s[3] - s[2] - 1) : line.substr(s[2] + 1);
      }
      if (sc.have_last && r1 == sv(sc.last_r1s) && r2 == sv(sc.last_r2s)) {
        race = sc.last_race;
      } else {
        sc.race_key.assign(r1);
        sc.race_key.push_back('\x01');
        sc.race_key.append(r2);
        auto rc = sc.race_cache.find(sc.race_key);
        if (rc != sc.race_cache.end()) {
          race = rc->second;
        } else {
          race = prm.races->resolve(r1, r2);
          sc.race_cache.emplace(sc.race_key, race);
        }
        if (race >= 0xFFFFu) {
          R.race_overflow = true;
          race = 0;
        }
        sc.last_r1s.assign(r1);
        sc.last_r2s.assign(r2);
        sc.last_race = race;
        sc.have_last = true;
      }
      const sv gl = ns >= 2 ? line.substr(s[0] + 1, s[1] - s[0] - 1) : line.substr(s[0] + 1);
      if (prm.line_dst && !R.dense && !gl.empty() && gl.size() <= GRIM_TOK_MAXGL && !memchr(gl.data(), '/', gl.size()) && race < 0xFFFFu) {
        // no '/' list: the device tokenizer's line (grim_tokdev.h); what it cannot take comes back through tokenize_lines
        prm.line_dst[line_no] = LineRec{(uint32_t)(gl.data() - text), (uint16_t)gl.size(), (uint16_t)race};
        kind = K_DEV;
        dev_line = true;
      } else if (R.dense) {
        // worst case one token per two bytes
        if (tmp_tok.size() < gl.size() / 2 + 8) tmp_tok.resize(gl.size() / 2 + 8);
        uint64_t nt = 0;
        if (fast && tokenise_gl_fast(D, gl, sj, tmp_tok.data(), tmp_tok.size(), nt))
          kind = K_DEV;
        else
          kind = tokenise_gl(D, gl, prm.planb, sc, sj, tmp_tok.data(), tmp_tok.size(), nt, R.ov, R.ov_pool, line_no, tok_overflow);
        if (kind == K_DEV) {
          sj.tok_off = (uint32_t)(R.tok_base + R.tok.size());
          R.tok.insert(R.tok.end(), tmp_tok.begin(), tmp_tok.begin() + nt);
          R.n_tok = R.tok.size();
        }
      } else {
        if (fast && tokenise_gl_fast(D, gl, sj, R.tok_dst, R.tok_cap, R.n_tok))
          kind = K_DEV;
        else
          kind = tokenise_gl(D, gl, prm.planb, sc, sj, R.tok_dst, R.tok_cap, R.n_tok, R.ov, R.ov_pool, line_no, tok_overflow);
        if (kind == K_DEV) sj.tok_off = (uint32_t)(R.tok_base + tok_before);
      }
    }
    if (kind == K_DEV || kind == K_MISS_NO_DEVICE || kind == K_PROBLEM_ID) {
      if (prm.masks && !dev_line) {
        auto it = prm.masks->fixed.find(std::string(line.substr(0, li.id_len)));
        if (it == prm.masks->fixed.end()) {
          if (kind == K_DEV) {
            R.n_tok = tok_before;
            if (R.dense) R.tok.resize(tok_before);
            while (!R.ov.empty() && R.ov.back().line == line_no) R.ov.pop_back();
          }
          kind = K_PROBLEM_RAW;
        } else if (kind == K_DEV) {
          sj.flags = it->second;
        }
      }
    }
    if (dev_line) {
      ++R.n_subj;
      ++R.n_devtok;
    } else if (kind == K_DEV) {
      sj.prior_idx = (uint16_t)race;
      const uint32_t subject_no = R.dense ? R.n_subj : line_no;
      if (R.dense) {
        R.subj.push_back(sj);
        R.dev.push_back((int32_t)R.n_subj);
      } else {
        R.subj_dst[line_no] = sj;
      }
      if (prm.classify) {
        const uint32_t si = R.first_subject + subject_no;
        const int cls = grim_classify(*prm.classify, sj);
        if (cls == GRIM_CLS_SMALL) {
          SmallRec rec;
          const uint16_t *tk = R.dense ? R.tok.data() + (sj.tok_off - R.tok_base) : R.tok_dst + (sj.tok_off - R.tok_base);
          grim_small_rec(sj, tk, si, rec);
          R.small.push_back(rec);
          R.os.push_back(si);
        } else if (cls == GRIM_CLS_MEDIUM) {
          R.om.push_back(si);
        } else {
          R.og.push_back(si);
        }
      }
      ++R.n_subj;
    } else if (R.dense) {
      R.dev.push_back(-1);
    }
    R.kind.push_back((uint8_t)kind);
    R.line.push_back(li);
    ++line_no;
    a = next;
  }
}

void tokenize_lines(const TokParams &prm, const char *text, TokRange &R, const std::vector<uint32_t> &lines, const LineRec *lrec,
                    std::vector<uint32_t> &os, std::vector<uint32_t> &om, std::vector<uint32_t> &og, std::vector<SmallRec> &small) {
  const DictSnap &D = *prm.snap;
  Scratch sc;
  for (uint32_t j : lines) {
    if (j >= R.kind.size() || R.kind[j] != K_DEV) continue;
    const sv gl(text + lrec[j].gl_off, lrec[j].gl_len);
    grim_subject sj;
    bool tok_overflow = false;
    const uint64_t before = R.n_tok;
    int kind;
    if (tokenise_gl_fast(D, gl, sj, R.tok_dst, R.tok_cap, R.n_tok))
      kind = K_DEV;
    else
      kind = tokenise_gl(D, gl, prm.planb, sc, sj, R.tok_dst, R.tok_cap, R.n_tok, R.ov, R.ov_pool, j, tok_overflow);
    R.kind[j] = (uint8_t)kind;
    if (kind != K_DEV) {
      if (R.n_subj) --R.n_subj;
      continue;
    }
    sj.tok_off = (uint32_t)(R.tok_base + before);
    sj.prior_idx = lrec[j].prior_idx;
    R.subj_dst[j] = sj;
    const uint32_t si = R.first_subject + j;
    const int cls = prm.classify ? grim_classify(*prm.classify, sj) : GRIM_CLS_GENERAL;
    if (cls == GRIM_CLS_SMALL) {
      SmallRec rec;
      grim_small_rec(sj, R.tok_dst + (sj.tok_off - R.tok_base), si, rec);
      small.push_back(rec);
      os.push_back(si);
    } else if (cls == GRIM_CLS_MEDIUM) {
      om.push_back(si);
    } else {
      og.push_back(si);
    }
  }
  // the formatter finds a line's overlay alleles by binary search over the line numbers
  std::stable_sort(R.ov.begin(), R.ov.end(), [](const OvEnt &a, const OvEnt &b) { return a.line < b.line; });
}

// ------------------------------------------------------------------------------------------------
// prior matrices: calc_priority_matrix (impute.py:1844-1924), one per distinct race pair.  Every
// operation is an IEEE double operation in the order numpy performs it (the file is compiled with
// -ffp-contract=off), so the matrices are bit-identical to the reference's.
// ------------------------------------------------------------------------------------------------
static void split_races(sv r, const PriorSpec &ps, std::vector<int> &out, bool &known) {
  out.clear();
  size_t a = 0;
  for (;;) {
    size_t b = r.find(';', a);
    sv e = r.substr(a, b == sv::npos ? sv::npos : b - a);
    int idx = -1;
    for (size_t p = 0; p < ps.pops.size(); ++p)
      if (sv(ps.pops[p]) == e) {
        idx = (int)p;
        break;
      }
    if (idx >= 0) known = true;
    out.push_back(idx);  // -1: a name that is not a population ("" in the reference)
    if (b == sv::npos) break;
    a = b + 1;
  }
}

void prior_matrix(const PriorSpec &ps, sv race1, sv race2, double *out) {
  const size_t P = ps.pops.size();
  auto base = [&]() {
    for (size_t i = 0; i < P; ++i)
      for (size_t j = 0; j < P; ++j) out[i * P + j] = (ps.unk_mr || i == j) ? 1.0 : 0.0;
  };
  if (race1.empty() && race2.empty()) return base();
  std::vector<int> l1, l2;
  bool known = false;
  split_races(race1, ps, l1, known);
  split_races(race2, ps, l2, known);
  if (!known) return base();
  std::vector<double> acc(P * P, 0.0), t(P * P), u(P * P);
  const double g = ps.gamma, al = ps.alpha, de = ps.delta;
  for (int ra : l1)
    for (int rb : l2) {
      if (ra < 0 && rb < 0) continue;
      std::fill(t.begin(), t.end(), 0.0);
      if (ra < 0 || rb < 0) {
        const size_t r = (size_t)(ra < 0 ? rb : ra);
        const double g2 = g * 2;
        for (size_t j = 0; j < P; ++j) t[r * P + j] = t[r * P + j] + g2;
        for (size_t i = 0; i < P; ++i)
          for (size_t j = 0; j < P; ++j) u[i * P + j] = t[i * P + j] + t[j * P + i];
        t.swap(u);
        t[r * P + r] -= g2;
      } else {
        const size_t a = (size_t)ra, b = (size_t)rb;
        for (size_t i = 0; i < P; ++i) {  // row and column may overlap at (a,b): scalar order
          t[a * P + i] = t[a * P + i] + g;
          t[i * P + b] = t[i * P + b] + g;
        }
        t[a * P + b] -= g;
        t[a * P + b] = t[a * P + b] + al;
        if (a != b) {
          for (size_t i = 0; i < P; ++i)
            for (size_t j = 0; j < P; ++j) u[i * P + j] = t[i * P + j] + t[j * P + i];
          t.swap(u);
          t[a * P + a] -= g;
          t[b * P + b] -= g;
        }
        t[a * P + a] += de;
        if (a != b) t[b * P + b] += de;
      }
      for (size_t i = 0; i < P; ++i)
        for (size_t j = 0; j < P; ++j) {
          const double v = (ps.eta * 1.0 + t[i * P + j]) + ps.beta * (i == j ? 1.0 : 0.0);
          acc[i * P + j] += v;
        }
    }
  double total = 0.0;
  bool first = true;
  for (size_t i = 0; i < P; ++i)
    for (size_t j = 0; j < P; ++j) {
      acc[i * P + j] = acc[i * P + j] * ps.count_by_prob[i] * ps.count_by_prob[j];
      total = first ? acc[i * P + j] : total + acc[i * P + j];  // Python: 0 (an int) + x
      first = false;
    }
  for (size_t k = 0; k < P * P; ++k) out[k] = acc[k] / total;
}

extern "C" int grim_prior_matrix(const grim_prior_spec *spec, const char *const *pop_names, uint32_t n_pops, const char *race1,
                                 const char *race2, double *out) {
  if (!spec || !pop_names || !out || n_pops == 0) return -1;
  PriorSpec ps;
  ps.alpha = spec->alpha; ps.eta = spec->eta; ps.beta = spec->beta; ps.gamma = spec->gamma; ps.delta = spec->delta;
  ps.unk_mr = spec->unk_mr != 0;
  for (uint32_t i = 0; i < n_pops; ++i) {
    ps.pops.emplace_back(pop_names[i]);
    ps.count_by_prob.push_back(spec->count_by_prob ? spec->count_by_prob[i] : 1.0);
  }
  prior_matrix(ps, sv(race1 ? race1 : ""), sv(race2 ? race2 : ""), out);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// formatter
// ------------------------------------------------------------------------------------------------
void OutBuf::grow(size_t k) {
  size_t nc = cap ? cap * 2 : 4096;
  while (nc < n + k) nc *= 2;
  p = (char *)realloc(p, nc);
  cap = nc;
}

// str(float) of CPython (repr style 'r': shortest digits; exponent form when exp10 < -4 or >= 16)
char *py_float_to(double x, char *out) {
  if (x == 0.0) {
    const char *z = std::signbit(x) ? "-0.0" : "0.0";
    const size_t n = strlen(z);
    memcpy(out, z, n);
    return out + n;
  }
  char buf[40];
  auto r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);
  char *s = buf, *end = r.ptr;
  if (*s == '-') {
    *out++ = '-';
    ++s;
  }
  char *e = (char *)memchr(s, 'e', (size_t)(end - s));
  if (!e) {  // inf / nan
    memcpy(out, s, (size_t)(end - s));
    return out + (end - s);
  }
  int exp10 = 0;
  {
    const char *q = e + 1;
    bool neg = false;
    if (*q == '-') { neg = true; ++q; } else if (*q == '+') ++q;
    for (; q < end; ++q) exp10 = exp10 * 10 + (*q - '0');
    if (neg) exp10 = -exp10;
  }
  if (exp10 < -4 || exp10 >= 16) {
    // CPython prints "d.ddde-XX" with at least two exponent digits: exactly to_chars' scientific form
    memcpy(out, s, (size_t)(end - s));
    return out + (end - s);
  }
  // digits without the point
  char dg[24];
  int nd = 0;
  for (char *q = s; q < e; ++q)
    if (*q != '.') dg[nd++] = *q;
  if (exp10 < 0) {
    *out++ = '0';
    *out++ = '.';
    for (int i = 0; i < -exp10 - 1; ++i) *out++ = '0';
    memcpy(out, dg, (size_t)nd);
    return out + nd;
  }
  const int ip = exp10 + 1;
  if (nd <= ip) {
    memcpy(out, dg, (size_t)nd);
    out += nd;
    for (int i = nd; i < ip; ++i) *out++ = '0';
    *out++ = '.';
    *out++ = '0';
    return out;
  }
  memcpy(out, dg, (size_t)ip);
  out += ip;
  *out++ = '.';
  memcpy(out, dg + ip, (size_t)(nd - ip));
  return out + (nd - ip);
}

void py_float(double x, std::string &out) {
  char b[48];
  char *e = py_float_to(x, b);
  out.append(b, (size_t)(e - b));
}

static inline char *put_u64(uint64_t v, char *out) {
  char tmp[24];
  int n = 0;
  do {
    tmp[n++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) *out++ = tmp[--n];
  return out;
}

namespace {
struct NameLookup {  // allele id of a key field -> text: dictionary, or the subject's own overlay
  const DictSnap &D;
  const TokRange &R;
  size_t ov_lo = 0, ov_hi = 0;  // overlay entries of the current line
  void set_line(uint32_t j) {
    ov_lo = ov_hi = 0;
    if (R.ov.empty()) return;
    auto it = std::lower_bound(R.ov.begin(), R.ov.end(), j, [](const OvEnt &e, uint32_t v) { return e.line < v; });
    ov_lo = (size_t)(it - R.ov.begin());
    ov_hi = ov_lo;
    while (ov_hi < R.ov.size() && R.ov[ov_hi].line == j) ++ov_hi;
  }
  inline sv get(uint32_t slot, uint32_t id) const {
    if (id < D.base[slot]) return D.name(slot, id);
    for (size_t o = ov_lo; o < ov_hi; ++o)
      if (R.ov[o].slot == slot && R.ov[o].id == id) return sv(R.ov_pool.data() + R.ov[o].off, R.ov[o].len);
    return sv();
  }
  // the allele names of a haplotype key in sorted() order -- or, graph_order, in the order of a graph node's name (loci_map
  // index order = slot order: the two differ only under a loci_map that is not alphabetical); returns how many
  inline int alleles(uint64_t key, sv *out, bool graph_order = false) const {
    int n = 0;
    for (uint32_t q = 0; q < D.n_loci; ++q) {
      const uint32_t s = graph_order ? q : D.order[q];
      const uint32_t a = (uint32_t)((key >> (GRIM_ABITS * s)) & 0xFFF);
      if (!a) continue;
      const sv nm = get(s, a - 1);
      if (nm.data()) out[n++] = nm;
    }
    if (!D.fixed_order && !graph_order) std::sort(out, out + n);
    return n;
  }
};
}  // namespace

void format_range(const FmtParams &fp, const char *text, const TokRange &tr, const grim_subject_result *res, const grim_row *rows,
                  uint64_t first_line, const uint8_t *skip, FmtRange &o) {
  const DictSnap &D = *fp.snap;
  const grim_params *prm = fp.prm;
  NameLookup names{D, tr};
  sv a[GRIM_MAXL], b[GRIM_MAXL];
  const size_t nl = tr.kind.size();
  auto pop_name = [&](uint32_t idx, int plan) -> sv { return plan == 'c' ? sv("all_pops") : sv(fp.pops[idx < fp.pops.size() ? idx : 0]); };
  // A phased row prints the haplotype STRINGS the reference holds (impute.py:24-58): names of graph nodes as the graph
  // spells them (Plan A, and Plan-B rows answered by one look-up: bit GRIM_KEY_GRAPH_ORDER of the key), keys of joined
  // blocks with their alleles sorted (open_option_, impute.py:1041-1069).  Only a loci_map that is not alphabetical can tell.
  auto hap_name = [&](uint64_t key, OutBuf &out, int plan) {
    const int n = names.alleles(key, a, plan == 'a' || ((key >> GRIM_KEY_GRAPH_ORDER) & 1ull));
    size_t need = (size_t)n;
    for (int i = 0; i < n; ++i) need += a[i].size();
    char *q = out.room(need);
    for (int i = 0; i < n; ++i) {
      if (i) *q++ = '~';
      memcpy(q, a[i].data(), a[i].size());
      q += a[i].size();
    }
    out.n = (size_t)(q - out.p);
  };
  auto raw_line = [&](size_t j) {
    o.t[5].put(sv(text + tr.line[j].off, tr.line[j].len));
    o.t[5].put('\n');
  };
  auto idx_id = [&](OutBuf &out, uint64_t i, sv sid) {
    char *q = out.room(24 + sid.size());
    q = put_u64(i, q);
    *q++ = ',';
    memcpy(q, sid.data(), sid.size());
    q += sid.size();
    *q++ = '\n';
    out.n = (size_t)(q - out.p);
  };
  auto log_line = [&](uint64_t i, sv sid, uint64_t count) {  // "{i} Subject: {id} {n} haplotypes"
    OutBuf &L = o.t[6];
    char *q = L.room(64 + sid.size());
    q = put_u64(i, q);
    memcpy(q, " Subject: ", 10);
    q += 10;
    memcpy(q, sid.data(), sid.size());
    q += sid.size();
    *q++ = ' ';
    q = put_u64(count, q);
    memcpy(q, " haplotypes\n", 12);
    q += 12;
    L.n = (size_t)(q - L.p);
  };
  auto log_exception = [&](uint64_t i, sv sid) {  // f"{i} Subject: {id} - Exception"
    OutBuf &L = o.t[6];
    char *q = L.room(64 + sid.size());
    q = put_u64(i, q);
    memcpy(q, " Subject: ", 10);
    q += 10;
    memcpy(q, sid.data(), sid.size());
    q += sid.size();
    memcpy(q, " - Exception\n", 13);
    q += 13;
    L.n = (size_t)(q - L.p);
  };
  for (size_t j = 0; j < nl; ++j) {
    if (skip && skip[j]) continue;
    const uint64_t i = first_line + j;
    const sv sid(text + tr.line[j].off, tr.line[j].id_len);
    const int kind = tr.kind[j];
    if (kind == K_UNSUPPORTED || kind == K_UNSUPPORTED_GL) {
      o.unsupported.push_back((uint32_t)j);
      continue;
    }
    if (kind == K_PROBLEM_RAW) {
      raw_line(j);
      if (fp.want_log) log_exception(i, sid);
      continue;
    }
    if (kind == K_PROBLEM_ID) {
      idx_id(o.t[5], i, sid);
      continue;
    }
    const grim_subject_result *r = nullptr;
    if (kind == K_DEV) r = &res[tr.dense ? (uint32_t)tr.dev[j] + tr.first_subject : tr.first_subject + (uint32_t)j];
    if (r && r->status == GRIM_ST_UNSUPPORTED) {
      o.unsupported.push_back((uint32_t)j);
      continue;
    }
    if (r && r->status == GRIM_ST_NOPHASE) {
      // no phase could be opened: the reference's placeholder result raises in the phased writer
      // (impute.py:1607-1609, 2090-2097) -> raw line; with haplotype output off nothing is written
      if (prm->out_haps) {
        raw_line(j);
        if (fp.want_log) {
          log_line(i, sid, 3);  // len("Nan")
          log_exception(i, sid);
        }
      } else if (fp.want_log) {
        log_line(i, sid, 0);
      }
      continue;
    }
    if (prm->eps_nonpositive && kind != K_MISS_NO_DEVICE) {
      // epsilon <= 0 (impute.py:1663-1665): no pass ever runs, both results are the {"Haps": "NaN"} sentinel without a
      // "Pops" entry; .miss sees an empty MUUG result only when the MUUG pass is off, then a writer raises KeyError
      if (prm->out_haps && !prm->out_muug) idx_id(o.t[4], i, sid);
      raw_line(j);
      if (fp.want_log) log_exception(i, sid);
      continue;
    }
    const uint32_t n_pairs = (r && prm->out_haps) ? r->n_pairs : 0, n_geno = (r && prm->out_muug) ? r->n_genotypes : 0;
    if (prm->out_haps && n_pairs == 0 && n_geno == 0) idx_id(o.t[4], i, sid);  // impute.py:2065-2068 (never when haplotype output is off)
    if (fp.want_log) {
      if (prm->out_haps) log_line(i, sid, r ? r->n_pairs : 0);
      if (prm->out_muug) log_line(i, sid, r ? r->n_genotypes : 0);
      OutBuf &L = o.t[6];
      char *q = L.room(40);
      q = py_float_to(fp.per_subject_s, q);
      *q++ = '\n';
      L.n = (size_t)(q - L.p);
    }
    if (!r) continue;
    names.set_line((uint32_t)j);
    for (int pass = 0; pass < 4; ++pass) {
      // the reference writes phased rows, phased pops, MUUG rows, MUUG pops (impute.py:2070-2118)
      static const int order[4] = {GRIM_T_PMUG, GRIM_T_PMUG_POPS, GRIM_T_UMUG, GRIM_T_UMUG_POPS};
      const int table = order[pass];
      const bool on = (table == GRIM_T_UMUG || table == GRIM_T_UMUG_POPS) ? prm->out_muug : prm->out_haps;
      if (!on) continue;
      const bool phased = table == GRIM_T_PMUG || table == GRIM_T_PMUG_POPS;
      const int plan = phased && r->plan_phased ? r->plan_phased : r->plan;
      OutBuf &out = o.t[table];
      for (uint32_t k = 0; k < r->n_rows[table]; ++k) {
        const grim_row &row = rows[r->row_off[table] + k];
        out.put(sid);
        out.put(',');
        if (table == GRIM_T_UMUG) {  // impute.py:497-504
          const int na = names.alleles(row.a, a), nb = names.alleles(row.b, b);
          const int n = na < nb ? na : nb;
          size_t need = 2 * (size_t)n;
          for (int z = 0; z < n; ++z) need += a[z].size() + b[z].size();
          char *q = out.room(need);
          for (int z = 0; z < n; ++z) {
            if (z) *q++ = '^';
            sv x = a[z], y = b[z];
            if (y < x) std::swap(x, y);
            memcpy(q, x.data(), x.size());
            q += x.size();
            *q++ = '+';
            memcpy(q, y.data(), y.size());
            q += y.size();
          }
          out.n = (size_t)(q - out.p);
        } else if (table == GRIM_T_PMUG) {
          if (prm->em_mr) {  // impute.py:79-99
            hap_name(row.a, out, plan);
            out.put(';');
            out.put(pop_name(row.popa, plan));
            out.put(',');
            hap_name(row.b, out, plan);
            out.put(';');
            out.put(pop_name(row.popb, plan));
          } else {
            hap_name(row.a, out, plan);
            out.put('+');
            hap_name(row.b, out, plan);
          }
        } else {  // a population-pair row names its pair in popa / popb (a / b repeat it, except in the ONE row that is a
                  // half-wave subject's genotype row and both its population rows at once: grim_small.h)
          out.put(pop_name(row.popa, plan));
          out.put(',');
          out.put(pop_name(row.popb, plan));
        }
        char *q = out.room(64);
        *q++ = ',';
        q = py_float_to(row.prob, q);
        *q++ = ',';
        q = put_u64(k, q);
        *q++ = '\n';
        out.n = (size_t)(q - out.p);
      }
      if (table == GRIM_T_UMUG_POPS && plan == 'c' && r->n_rows[table] == 0) {  // impute.py:1375-1378
        out.put(sid);
        out.put(sv(",all_pops,all_pops,0,0\n"));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// whole-block entry points (grim_tokenize / grim_format) on top of the cores
// ------------------------------------------------------------------------------------------------
namespace {
struct LocalRaces : RaceResolver {  // one table per grim_tokenize call
  std::mutex mu;
  std::unordered_map<std::string, uint32_t> idx;
  std::vector<std::pair<std::string, std::string>> pairs;
  uint32_t resolve(sv r1, sv r2) override {
    std::string key;
    key.reserve(r1.size() + r2.size() + 1);
    key.append(r1);
    key.push_back('\x01');
    key.append(r2);
    std::lock_guard<std::mutex> lk(mu);
    auto it = idx.find(key);
    if (it != idx.end()) return it->second;
    const uint32_t k = (uint32_t)pairs.size();
    idx.emplace(std::move(key), k);
    pairs.emplace_back(std::string(r1), std::string(r2));
    return k;
  }
};
}  // namespace

struct grim_parsed {
  std::string text;
  DictSnap snap;
  TokRange all;  // dense: every line of the block
  std::vector<std::pair<std::string, std::string>> races;
  std::vector<uint64_t> ov_key;  // unused
};

struct grim_text {
  std::string t[7];
  std::vector<uint32_t> unsupported;
};

extern "C" grim_parsed *grim_tokenize(grim_dict *d, const char *text, uint64_t len, int planb, int n_threads) {
  if (!d) return nullptr;
  grim_parsed *P = new grim_parsed();
  P->text.assign(text, len);
  dict_snapshot(d, P->snap);
  if (n_threads < 1) n_threads = 1;
  // byte ranges of whole lines
  size_t nt = std::min<size_t>((size_t)n_threads, std::max<size_t>(1, len / 65536));
  std::vector<uint64_t> cut(nt + 1, len);
  cut[0] = 0;
  for (size_t t = 1; t < nt; ++t) {
    uint64_t p = len * t / nt;
    if (p < cut[t - 1]) p = cut[t - 1];
    const char *nl = p < len ? (const char *)memchr(P->text.data() + p, '\n', len - p) : nullptr;
    cut[t] = nl ? (uint64_t)(nl - P->text.data()) + 1 : len;
  }
  LocalRaces races;
  TokParams tp{&P->snap, planb != 0, &races, nullptr, nullptr};
  std::vector<TokRange> parts(nt);
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; ++t) {
    if (nt == 1)
      tokenize_range(tp, P->text.data(), cut[t], cut[t + 1], parts[t]);
    else
      th.emplace_back([&, t]() { tokenize_range(tp, P->text.data(), cut[t], cut[t + 1], parts[t]); });
  }
  for (auto &x : th) x.join();
  // merge in line order (the race indices are already global)
  TokRange &A = P->all;
  A.dense = true;
  for (TokRange &R : parts) {
    const uint32_t sbase = (uint32_t)A.subj.size(), lbase = (uint32_t)A.kind.size();
    const uint64_t tbase = A.tok.size(), pbase = A.ov_pool.size();
    for (grim_subject sj : R.subj) {
      sj.tok_off += (uint32_t)tbase;
      A.subj.push_back(sj);
    }
    A.tok.insert(A.tok.end(), R.tok.begin(), R.tok.end());
    A.kind.insert(A.kind.end(), R.kind.begin(), R.kind.end());
    A.line.insert(A.line.end(), R.line.begin(), R.line.end());
    for (int32_t dv : R.dev) A.dev.push_back(dv < 0 ? -1 : dv + (int32_t)sbase);
    for (OvEnt e : R.ov) {
      e.line += lbase;
      e.off += pbase;
      A.ov.push_back(e);
    }
    A.ov_pool += R.ov_pool;
    A.race_overflow = A.race_overflow || R.race_overflow;
  }
  A.n_subj = (uint32_t)A.subj.size();
  A.n_tok = A.tok.size();
  if (A.tok.empty()) A.tok.push_back(0);
  P->races = races.pairs;
  if (A.race_overflow) {
    delete P;
    return nullptr;
  }
  return P;
}

extern "C" void grim_parsed_free(grim_parsed *p) { delete p; }
extern "C" uint32_t grim_parsed_lines(const grim_parsed *p) { return (uint32_t)p->all.kind.size(); }
extern "C" uint32_t grim_parsed_subjects(const grim_parsed *p) { return (uint32_t)p->all.subj.size(); }
extern "C" const grim_subject *grim_parsed_subject_array(const grim_parsed *p) { return p->all.subj.data(); }
extern "C" const uint16_t *grim_parsed_tokens(const grim_parsed *p, uint64_t *n) {
  if (n) *n = p->all.tok.size();
  return p->all.tok.data();
}
extern "C" const uint8_t *grim_parsed_kinds(const grim_parsed *p) { return p->all.kind.data(); }
extern "C" const int32_t *grim_parsed_dev_index(const grim_parsed *p) { return p->all.dev.data(); }
extern "C" uint32_t grim_parsed_n_races(const grim_parsed *p) { return (uint32_t)p->races.size(); }
extern "C" const char *grim_parsed_race(const grim_parsed *p, uint32_t i, int which) {
  if (i >= p->races.size()) return nullptr;
  return which ? p->races[i].second.c_str() : p->races[i].first.c_str();
}
// overrides used by the host language for things only it knows (bin_imputation_in_file phase masks,
// impute.py:2001-2005,2030-2032): force a line's outcome kind / set a subject's fixed-position mask
extern "C" int grim_parsed_set_kind(grim_parsed *p, uint32_t line, uint8_t kind) {
  if (!p || line >= p->all.kind.size()) return -1;
  p->all.kind[line] = kind;
  return 0;
}
extern "C" int grim_parsed_set_flags(grim_parsed *p, uint32_t line, uint8_t flags) {
  if (!p || line >= p->all.kind.size() || p->all.dev[line] < 0) return -1;
  p->all.subj[p->all.dev[line]].flags = flags;
  return 0;
}
// subject id text of line i (not NUL terminated)
extern "C" const char *grim_parsed_id(const grim_parsed *p, uint32_t i, uint32_t *len) {
  if (i >= p->all.kind.size()) return nullptr;
  if (len) *len = p->all.line[i].id_len;
  return p->text.data() + p->all.line[i].off;
}
// text of allele `id` at locus slot `slot` as line `line` uses it: a dictionary allele, or one of the line's own
// (ids from the dictionary's size upwards are private to a subject); NULL when there is none
extern "C" const char *grim_parsed_allele(const grim_parsed *p, uint32_t line, uint32_t slot, uint32_t id, uint32_t *len) {
  if (!p || slot >= p->snap.n_loci) return nullptr;
  NameLookup nm{p->snap, p->all};
  nm.set_line(line);
  const sv s = nm.get(slot, id);
  if (len) *len = (uint32_t)s.size();
  return s.data();
}

extern "C" grim_text *grim_format(const grim_dict *d, const grim_parsed *P, const grim_params *prm, const char *const *pop_names,
                                  uint32_t n_pops, const grim_subject_result *res, const grim_row *rows, uint64_t line_offset,
                                  const uint8_t *skip, int n_threads) {
  if (!d || !P || !prm) return nullptr;
  FmtParams fp;
  fp.snap = &P->snap;
  fp.prm = prm;
  for (uint32_t i = 0; i < n_pops; ++i) fp.pops.emplace_back(pop_names[i]);
  if (fp.pops.empty()) fp.pops.emplace_back("");
  (void)n_threads;  // the block API formats on the calling thread; the streaming pipeline is the parallel path
  FmtRange out;
  format_range(fp, P->text.data(), P->all, res, rows, line_offset, skip, out);
  grim_text *T = new grim_text();
  for (int k = 0; k < 7; ++k) T->t[k].assign(out.t[k].p ? out.t[k].p : "", out.t[k].n);
  T->unsupported = out.unsupported;
  return T;
}

extern "C" const char *grim_text_get(const grim_text *t, int which, uint64_t *len) {
  if (!t || which < 0 || which > 6) return nullptr;
  if (len) *len = t->t[which].size();
  return t->t[which].data();
}

extern "C" void grim_text_free(grim_text *t) { delete t; }

// str(float) exposed for tests
extern "C" int grim_format_double(double x, char *buf, int cap) {
  char b[48];
  char *e = py_float_to(x, b);
  const int n = (int)(e - b);
  if (n + 1 > cap) return -1;
  memcpy(buf, b, (size_t)n);
  buf[n] = 0;
  return n;
}
