// grim_host.cpp -- host side of the boundary, in C++: allele dictionary, GL tokenizer, result formatter.
//
// Replaces, for whole files at a time and on all host cores, what the reference does per line in
// Python (it is the end-to-end bottleneck once the kernels are fast):
//   impute_file line handling      impute.py:2022-2036   (rstrip, ',' or '%' split, id / GL / races)
//   clean_up_gl                    impute.py:105-118
//   gl2haps                        impute.py:246-272
//   write_best_prob*, .miss/.problem rules, str(float)   impute.py:24-99, 2061-2118
// No GPU code here.  Same outcomes as grim/imputation/impute.py::_tokenise / _write_rows (the Python
// versions stay as the single-subject path and as the cross-check in tests/).
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <charconv>
#include <cmath>
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

// ------------------------------------------------------------------------------------------------
// tokenizer
// ------------------------------------------------------------------------------------------------
enum { K_DEV = 0, K_PROBLEM_ID = 1, K_PROBLEM_RAW = 2, K_MISS_NO_DEVICE = 3 };

struct Pending {  // an allele the dictionary does not know yet: interned serially after the parallel pass
  uint64_t tok_index;
  uint32_t slot;
  std::string name;
};

struct Chunk {
  std::vector<uint8_t> kind;
  std::vector<int32_t> dev;
  std::vector<uint64_t> line_off, id_off;
  std::vector<uint32_t> line_len, id_len;
  std::vector<grim_subject> subj;
  std::vector<uint16_t> tok;
  std::vector<Pending> pending;
  std::vector<std::pair<std::string, std::string>> races;  // local unique pairs
  std::unordered_map<std::string, uint32_t> race_idx;
  std::vector<uint32_t> subj_race;                          // per local subject: local race index
  std::vector<uint32_t> line_race;
};

struct grim_parsed {
  std::string text;
  std::vector<uint8_t> kind;
  std::vector<int32_t> dev;
  std::vector<uint64_t> line_off, id_off;
  std::vector<uint32_t> line_len, id_len;
  std::vector<grim_subject> subj;
  std::vector<uint16_t> tok;
  std::vector<std::pair<std::string, std::string>> races;
};

static inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

static void split(sv s, char sep, std::vector<sv> &out) {
  out.clear();
  size_t a = 0;
  for (;;) {
    size_t b = s.find(sep, a);
    if (b == sv::npos) {
      out.push_back(s.substr(a));
      return;
    }
    out.push_back(s.substr(a, b - a));
    a = b + 1;
  }
}

// one GL string -> kind (+ subject record and tokens appended to the chunk when K_DEV)
static int tokenise_gl(const grim_dict *d, sv gl, bool planb, Chunk &ck, std::string &scratch, std::vector<sv> &parts,
                       std::vector<sv> &tmp1, std::vector<sv> &tmp2) {
  if (gl.empty()) return K_PROBLEM_ID;
  // clean_up_gl: drop every 'g' and 'L', then the entries that start or end with 'U'
  scratch.clear();
  for (char c : gl)
    if (c != 'g' && c != 'L') scratch.push_back(c);
  split(sv(scratch), '^', parts);
  {
    size_t w = 0;
    for (size_t i = 0; i < parts.size(); ++i) {
      sv e = parts[i];
      bool drop = !e.empty() && (e.front() == 'U' || e.back() == 'U');
      if (!drop) parts[w++] = e;
    }
    parts.resize(w);
  }
  if (parts.empty()) return K_PROBLEM_ID;                       // cleaned == ""
  if (parts.size() == 1 && (parts[0].empty() || parts[0] == " ")) return K_PROBLEM_ID;
  std::vector<sv> &side1 = tmp1, &side2 = tmp2;
  side1.clear();
  side2.clear();
  size_t blanks = 0;
  std::vector<sv> two;
  for (sv p : parts) {
    if (p.empty()) return K_PROBLEM_RAW;  // p[0] -> IndexError in the reference
    if (p[0] == '+') p = p.substr(1);
    split(p, '+', two);
    if (two.size() == 1) {
      if (two[0].empty()) {
        ++blanks;
        continue;
      }
      return K_PROBLEM_ID;
    }
    side1.push_back(two[0]);
    side2.push_back(two[1]);
  }
  size_t n = parts.size() - blanks;
  std::sort(side1.begin(), side1.end());
  std::sort(side2.begin(), side2.end());
  if (n != side1.size() || n < 1 || n > d->n_loci) return K_PROBLEM_RAW;
  grim_subject sj;
  memset(&sj, 0, sizeof(sj));
  sj.tok_off = (uint32_t)ck.tok.size();
  const size_t tok_mark = ck.tok.size(), pend_mark = ck.pending.size();
  bool unknown_locus = false;
  uint32_t used = 0, npos = 0;
  std::vector<sv> alts[2];
  for (size_t k = 0; k < n; ++k) {
    split(side1[k], '/', alts[0]);
    split(side2[k], '/', alts[1]);
    sv locus;
    bool first = true, mixed = false;
    for (int s = 0; s < 2; ++s)
      for (sv a : alts[s]) {
        sv l = a.substr(0, a.find('*'));
        if (first) {
          locus = l;
          first = false;
        } else if (l != locus) {
          mixed = true;
        }
      }
    if (mixed) goto irregular;
    {
      auto it = d->locus_slot.find(std::string(locus));
      if (it == d->locus_slot.end()) {
        unknown_locus = true;
        continue;
      }
      uint32_t slot = it->second;
      if ((used >> slot) & 1u) goto irregular;
      used |= 1u << slot;
      sj.slot[npos] = (uint8_t)slot;
      if (side1[k] == side2[k]) sj.pad[0] |= (uint8_t)(1u << k);
      for (int s = 0; s < 2; ++s) {
        size_t start = ck.tok.size();
        uint32_t cnt = 0;
        for (size_t i = 0; i < alts[s].size(); ++i) {
          sv a = alts[s][i];
          bool dup = false;
          for (size_t j = 0; j < i && !dup; ++j) dup = (alts[s][j] == a);
          if (dup) continue;
          auto f = d->ids[slot].find(std::string(a));
          if (f != d->ids[slot].end()) {
            ck.tok.push_back((uint16_t)f->second);
          } else {
            ck.pending.push_back({(uint64_t)ck.tok.size(), slot, std::string(a)});
            ck.tok.push_back(0xFFFF);
          }
          ++cnt;
        }
        (void)start;
        sj.cnt[npos][s] = (uint16_t)(cnt > 65535 ? 65535 : cnt);
        sj.wid[npos][s] = (uint16_t)(alts[s].size() > 65535 ? 65535 : alts[s].size());
      }
      ++npos;
    }
  }
  if (unknown_locus) {
    ck.tok.resize(tok_mark);
    ck.pending.resize(pend_mark);
    return planb ? K_PROBLEM_RAW : K_MISS_NO_DEVICE;  // KeyError in Plan B / plain miss (see _tokenise)
  }
  sj.n_loci = (uint8_t)npos;
  ck.subj.push_back(sj);
  return K_DEV;
irregular:
  ck.tok.resize(tok_mark);
  ck.pending.resize(pend_mark);
  return K_PROBLEM_RAW;
}

static void parse_range(const grim_dict *d, const std::string &text, const std::vector<std::pair<uint64_t, uint64_t>> &lines,
                        size_t lo, size_t hi, bool planb, Chunk &ck) {
  std::string scratch;
  std::vector<sv> fields, parts, t1, t2;
  for (size_t li = lo; li < hi; ++li) {
    uint64_t a = lines[li].first, b = lines[li].second;
    while (b > a && is_space(text[b - 1])) --b;  // rstrip
    sv line(text.data() + a, b - a);
    ck.line_off.push_back(a);
    ck.line_len.push_back((uint32_t)(b - a));
    char sep = line.find(',') != sv::npos ? ',' : '%';
    split(line, sep, fields);
    int kind;
    uint64_t id_off = a;
    uint32_t id_len = 0, race = 0;
    if (fields.size() < 2 || fields.size() == 3) {
      kind = K_PROBLEM_RAW;
      if (!fields.empty()) id_len = (uint32_t)fields[0].size();
    } else {
      id_len = (uint32_t)fields[0].size();
      std::string r1, r2;
      if (fields.size() > 2) {
        r1 = std::string(fields[2]);
        r2 = std::string(fields[3]);
      }
      std::string key = r1 + '\x01' + r2;
      auto it = ck.race_idx.find(key);
      if (it == ck.race_idx.end()) {
        race = (uint32_t)ck.races.size();
        ck.race_idx.emplace(key, race);
        ck.races.emplace_back(r1, r2);
      } else {
        race = it->second;
      }
      kind = tokenise_gl(d, fields[1], planb, ck, scratch, parts, t1, t2);
    }
    ck.kind.push_back((uint8_t)kind);
    ck.id_off.push_back(id_off);
    ck.id_len.push_back(id_len);
    if (kind == K_DEV) {
      ck.dev.push_back((int32_t)ck.subj.size() - 1);
      ck.subj_race.push_back(race);
    } else {
      ck.dev.push_back(-1);
    }
  }
}

extern "C" grim_parsed *grim_tokenize(grim_dict *d, const char *text, uint64_t len, int planb, int n_threads) {
  if (!d) return nullptr;
  grim_parsed *P = new grim_parsed();
  P->text.assign(text, len);
  // line table ('\n' separated; a final line without '\n' counts; no empty line after a trailing '\n')
  std::vector<std::pair<uint64_t, uint64_t>> lines;
  uint64_t a = 0;
  while (a < len) {
    const char *nl = (const char *)memchr(P->text.data() + a, '\n', len - a);
    uint64_t b = nl ? (uint64_t)(nl - P->text.data()) : len;
    lines.emplace_back(a, b);
    a = b + 1;
  }
  size_t nl = lines.size();
  if (n_threads < 1) n_threads = 1;
  size_t nt = std::min<size_t>((size_t)n_threads, std::max<size_t>(1, nl / 512));
  std::vector<Chunk> chunks(nt);
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; ++t) {
    size_t lo = nl * t / nt, hi = nl * (t + 1) / nt;
    if (nt == 1) {
      parse_range(d, P->text, lines, lo, hi, planb != 0, chunks[t]);
    } else {
      th.emplace_back(parse_range, d, std::cref(P->text), std::cref(lines), lo, hi, planb != 0, std::ref(chunks[t]));
    }
  }
  for (auto &x : th) x.join();
  // merge in line order
  std::unordered_map<std::string, uint32_t> race_idx;
  for (Chunk &ck : chunks) {
    const uint32_t sbase = (uint32_t)P->subj.size();
    const uint64_t tbase = P->tok.size();
    for (Pending &p : ck.pending) {
      int32_t id = dict_intern(d, p.slot, p.name);
      ck.tok[p.tok_index] = (uint16_t)(id < 0 ? 0xFFFE : id);
    }
    std::vector<uint32_t> rmap(ck.races.size());
    for (size_t r = 0; r < ck.races.size(); ++r) {
      std::string key = ck.races[r].first + '\x01' + ck.races[r].second;
      auto it = race_idx.find(key);
      if (it == race_idx.end()) {
        rmap[r] = (uint32_t)P->races.size();
        race_idx.emplace(key, rmap[r]);
        P->races.push_back(ck.races[r]);
      } else {
        rmap[r] = it->second;
      }
    }
    for (size_t s = 0; s < ck.subj.size(); ++s) {
      grim_subject sj = ck.subj[s];
      sj.tok_off += (uint32_t)tbase;
      sj.prior_idx = (uint16_t)rmap[ck.subj_race[s]];
      P->subj.push_back(sj);
    }
    P->tok.insert(P->tok.end(), ck.tok.begin(), ck.tok.end());
    for (size_t i = 0; i < ck.kind.size(); ++i) {
      P->kind.push_back(ck.kind[i]);
      P->dev.push_back(ck.dev[i] < 0 ? -1 : ck.dev[i] + (int32_t)sbase);
      P->line_off.push_back(ck.line_off[i]);
      P->line_len.push_back(ck.line_len[i]);
      P->id_off.push_back(ck.id_off[i]);
      P->id_len.push_back(ck.id_len[i]);
    }
  }
  if (P->tok.empty()) P->tok.push_back(0);
  return P;
}

extern "C" void grim_parsed_free(grim_parsed *p) { delete p; }
extern "C" uint32_t grim_parsed_lines(const grim_parsed *p) { return (uint32_t)p->kind.size(); }
extern "C" uint32_t grim_parsed_subjects(const grim_parsed *p) { return (uint32_t)p->subj.size(); }
extern "C" const grim_subject *grim_parsed_subject_array(const grim_parsed *p) { return p->subj.data(); }
extern "C" const uint16_t *grim_parsed_tokens(const grim_parsed *p, uint64_t *n) {
  if (n) *n = p->tok.size();
  return p->tok.data();
}
extern "C" const uint8_t *grim_parsed_kinds(const grim_parsed *p) { return p->kind.data(); }
extern "C" const int32_t *grim_parsed_dev_index(const grim_parsed *p) { return p->dev.data(); }
extern "C" uint32_t grim_parsed_n_races(const grim_parsed *p) { return (uint32_t)p->races.size(); }
extern "C" const char *grim_parsed_race(const grim_parsed *p, uint32_t i, int which) {
  if (i >= p->races.size()) return nullptr;
  return which ? p->races[i].second.c_str() : p->races[i].first.c_str();
}
// overrides used by the host language for things only it knows (bin_imputation_in_file phase masks,
// impute.py:2001-2005,2030-2032): force a line's outcome kind / set a subject's fixed-position mask
extern "C" int grim_parsed_set_kind(grim_parsed *p, uint32_t line, uint8_t kind) {
  if (!p || line >= p->kind.size()) return -1;
  p->kind[line] = kind;
  return 0;
}
extern "C" int grim_parsed_set_flags(grim_parsed *p, uint32_t line, uint8_t flags) {
  if (!p || line >= p->kind.size() || p->dev[line] < 0) return -1;
  p->subj[p->dev[line]].flags = flags;
  return 0;
}
// subject id text of line i (not NUL terminated)
extern "C" const char *grim_parsed_id(const grim_parsed *p, uint32_t i, uint32_t *len) {
  if (i >= p->kind.size()) return nullptr;
  if (len) *len = p->id_len[i];
  return p->text.data() + p->id_off[i];
}

// ------------------------------------------------------------------------------------------------
// formatter
// ------------------------------------------------------------------------------------------------
// str(float) of CPython (repr style 'r': shortest digits; exponent form when exp10 < -4 or >= 16)
void py_float(double x, std::string &out) {
  if (x == 0.0) {
    out += (std::signbit(x) ? "-0.0" : "0.0");
    return;
  }
  char buf[40];
  auto r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);
  sv s(buf, r.ptr - buf);
  if (s.front() == '-') {
    out.push_back('-');
    s.remove_prefix(1);
  }
  size_t e = s.find('e');
  if (e == sv::npos) {  // inf / nan
    out += s;
    return;
  }
  std::string digits;
  for (char c : s.substr(0, e))
    if (c != '.') digits.push_back(c);
  int exp10 = 0;
  std::from_chars(s.data() + e + 1 + (s[e + 1] == '+' ? 1 : 0), s.data() + s.size(), exp10);
  if (exp10 < -4 || exp10 >= 16) {
    out.push_back(digits[0]);
    if (digits.size() > 1) {
      out.push_back('.');
      out.append(digits, 1, std::string::npos);
    }
    out.push_back('e');
    out.push_back(exp10 < 0 ? '-' : '+');
    int ae = exp10 < 0 ? -exp10 : exp10;
    if (ae < 10) out.push_back('0');
    out += std::to_string(ae);
  } else if (exp10 < 0) {
    out += "0.";
    out.append((size_t)(-exp10 - 1), '0');
    out += digits;
  } else {
    size_t ip = (size_t)exp10 + 1;
    if (digits.size() <= ip) {
      out += digits;
      out.append(ip - digits.size(), '0');
      out += ".0";
    } else {
      out.append(digits, 0, ip);
      out.push_back('.');
      out.append(digits, ip, std::string::npos);
    }
  }
}

static void key_alleles(const grim_dict *d, uint64_t key, std::vector<sv> &out) {
  out.clear();
  for (uint32_t s = 0; s < d->n_loci; ++s) {
    uint32_t a = (uint32_t)((key >> (GRIM_ABITS * s)) & 0xFFF);
    if (a && a - 1 < d->names[s].size()) out.push_back(sv(d->names[s][a - 1]));
  }
  std::sort(out.begin(), out.end());
}

struct FmtOut {
  std::string t[6];  // umug, umug_pops, pmug, pmug_pops, miss, problem
};

struct grim_text {
  std::string t[6];
};

static void format_range(const grim_dict *d, const grim_parsed *P, const grim_params *prm, const std::vector<std::string> &pops,
                         const grim_subject_result *res, const grim_row *rows, uint64_t line_offset, const uint8_t *skip,
                         size_t lo, size_t hi, FmtOut &o) {
  std::vector<sv> a, b;
  auto pop_name = [&](uint32_t idx, int plan) -> sv { return plan == 'c' ? sv("all_pops") : sv(pops[idx < pops.size() ? idx : 0]); };
  auto hap_name = [&](uint64_t key, std::string &out, std::vector<sv> &tmp) {
    key_alleles(d, key, tmp);
    for (size_t i = 0; i < tmp.size(); ++i) {
      if (i) out.push_back('~');
      out += tmp[i];
    }
  };
  for (size_t j = lo; j < hi; ++j) {
    if (skip && skip[j]) continue;
    const uint64_t i = line_offset + j;
    sv sid(P->text.data() + P->id_off[j], P->id_len[j]);
    const int kind = P->kind[j];
    if (kind == K_PROBLEM_RAW) {
      o.t[5].append(P->text.data() + P->line_off[j], P->line_len[j]);
      o.t[5].push_back('\n');
      continue;
    }
    if (kind == K_PROBLEM_ID) {
      o.t[5] += std::to_string(i);
      o.t[5].push_back(',');
      o.t[5] += sid;
      o.t[5].push_back('\n');
      continue;
    }
    const grim_subject_result *r = kind == K_DEV ? &res[P->dev[j]] : nullptr;
    if (r && r->status == GRIM_ST_NOPHASE) {
      // no phase could be opened: the reference's placeholder result raises in the phased writer
      // (impute.py:1607-1609, 2090-2097) -> raw line; with haplotype output off nothing is written
      if (prm->out_haps) {
        o.t[5].append(P->text.data() + P->line_off[j], P->line_len[j]);
        o.t[5].push_back('\n');
      }
      continue;
    }
    const uint32_t n_pairs = (r && prm->out_haps) ? r->n_pairs : 0, n_geno = (r && prm->out_muug) ? r->n_genotypes : 0;
    if (prm->out_haps && n_pairs == 0 && n_geno == 0) {  // impute.py:2065-2068 (never when haplotype output is off)
      o.t[4] += std::to_string(i);
      o.t[4].push_back(',');
      o.t[4] += sid;
      o.t[4].push_back('\n');
    }
    if (!r) continue;
    for (int pass = 0; pass < 4; ++pass) {
      // the reference writes phased rows, phased pops, MUUG rows, MUUG pops (impute.py:2070-2118)
      static const int order[4] = {GRIM_T_PMUG, GRIM_T_PMUG_POPS, GRIM_T_UMUG, GRIM_T_UMUG_POPS};
      const int table = order[pass];
      const bool on = (table == GRIM_T_UMUG || table == GRIM_T_UMUG_POPS) ? prm->out_muug : prm->out_haps;
      if (!on) continue;
      const bool phased = table == GRIM_T_PMUG || table == GRIM_T_PMUG_POPS;
      const int plan = phased && r->plan_phased ? r->plan_phased : r->plan;
      std::string &out = o.t[table];
      for (uint32_t k = 0; k < r->n_rows[table]; ++k) {
        const grim_row &row = rows[r->row_off[table] + k];
        out += sid;
        out.push_back(',');
        if (table == GRIM_T_UMUG) {  // impute.py:497-504
          key_alleles(d, row.a, a);
          key_alleles(d, row.b, b);
          size_t n = std::min(a.size(), b.size());
          for (size_t z = 0; z < n; ++z) {
            if (z) out.push_back('^');
            sv x = a[z], y = b[z];
            if (y < x) std::swap(x, y);
            out += x;
            out.push_back('+');
            out += y;
          }
        } else if (table == GRIM_T_PMUG) {
          if (prm->em_mr) {  // impute.py:79-99
            hap_name(row.a, out, a);
            out.push_back(';');
            out += pop_name(row.popa, plan);
            out.push_back(',');
            hap_name(row.b, out, a);
            out.push_back(';');
            out += pop_name(row.popb, plan);
          } else {
            hap_name(row.a, out, a);
            out.push_back('+');
            hap_name(row.b, out, a);
          }
        } else {
          out += pop_name((uint32_t)row.a, plan);
          out.push_back(',');
          out += pop_name((uint32_t)row.b, plan);
        }
        out.push_back(',');
        py_float(row.prob, out);
        out.push_back(',');
        out += std::to_string(k);
        out.push_back('\n');
      }
      if (table == GRIM_T_UMUG_POPS && plan == 'c' && r->n_rows[table] == 0) {  // impute.py:1375-1378
        out += sid;
        out += ",all_pops,all_pops,0,0\n";
      }
    }
  }
}

extern "C" grim_text *grim_format(const grim_dict *d, const grim_parsed *P, const grim_params *prm, const char *const *pop_names,
                                  uint32_t n_pops, const grim_subject_result *res, const grim_row *rows, uint64_t line_offset,
                                  const uint8_t *skip, int n_threads) {
  if (!d || !P || !prm) return nullptr;
  std::vector<std::string> pops;
  for (uint32_t i = 0; i < n_pops; ++i) pops.emplace_back(pop_names[i]);
  if (pops.empty()) pops.emplace_back("");
  size_t nl = P->kind.size();
  if (n_threads < 1) n_threads = 1;
  size_t nt = std::min<size_t>((size_t)n_threads, std::max<size_t>(1, nl / 512));
  std::vector<FmtOut> outs(nt);
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; ++t) {
    size_t lo = nl * t / nt, hi = nl * (t + 1) / nt;
    if (nt == 1)
      format_range(d, P, prm, pops, res, rows, line_offset, skip, lo, hi, outs[t]);
    else
      th.emplace_back(format_range, d, P, prm, std::cref(pops), res, rows, line_offset, skip, lo, hi, std::ref(outs[t]));
  }
  for (auto &x : th) x.join();
  grim_text *T = new grim_text();
  for (int k = 0; k < 6; ++k) {
    size_t tot = 0;
    for (auto &o : outs) tot += o.t[k].size();
    T->t[k].reserve(tot);
    for (auto &o : outs) T->t[k] += o.t[k];
  }
  return T;
}

extern "C" const char *grim_text_get(const grim_text *t, int which, uint64_t *len) {
  if (!t || which < 0 || which > 5) return nullptr;
  if (len) *len = t->t[which].size();
  return t->t[which].data();
}

extern "C" void grim_text_free(grim_text *t) { delete t; }

// str(float) exposed for tests
extern "C" int grim_format_double(double x, char *buf, int cap) {
  std::string s;
  py_float(x, s);
  if ((int)s.size() + 1 > cap) return -1;
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}
