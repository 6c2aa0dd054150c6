// grim_host_internal.h -- shared by the host-side translation units of the library (no GPU code)
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/grim_hip.h"
#include "grim_layout.h"
#include "grim_tok.h"

using sv = std::string_view;

// allele dictionary: per locus slot, allele string <-> dense id
struct grim_dict {
  uint32_t n_loci;
  std::vector<std::string> locus_name;
  std::unordered_map<std::string, uint32_t> locus_slot;
  std::vector<std::unordered_map<std::string, uint32_t>> ids;
  std::vector<std::vector<std::string>> names;
};

int32_t dict_intern(grim_dict *d, uint32_t slot, sv a);  // -1 when the locus has run out of ids
void py_float(double x, std::string &out);               // CPython str(float)
char *py_float_to(double x, char *out);                  // same, into a buffer with >= 32 bytes of room; returns the end

// ---- frozen view of a dictionary: what the tokenizer and the formatter threads read (never mutated) --------------
// Alleles a subject brings that the dictionary does not know are NOT added to it (the dictionary belongs to the
// graph and lives as long as the graph does): they get ids base[slot], base[slot]+1, ... that are private to that
// one subject (an "overlay"), which is all the device needs -- it never compares alleles of different subjects.
struct DictSnap {
  struct Ent {
    uint64_t h;
    uint32_t id, off, len, used;
  };
  struct Locus {
    std::string name;
    uint32_t slot;
  };
  uint32_t n_loci = 0;
  uint32_t base[GRIM_MAXL] = {0, 0, 0, 0, 0};
  std::string pool;
  std::vector<uint32_t> name_off[GRIM_MAXL];  // [base + 1]
  std::vector<Ent> tab[GRIM_MAXL];
  uint32_t mask[GRIM_MAXL] = {0, 0, 0, 0, 0};
  std::vector<Locus> loci;
  uint8_t order[GRIM_MAXL] = {0, 1, 2, 3, 4};  // slots in the order sorted() puts a haplotype's allele names
  bool fixed_order = false;                    // ... which is the same for every haplotype (locus names are alphanumeric)

  static inline uint64_t hash(sv s) {
    const char *p = s.data();
    const size_t n = s.size();
    uint64_t a, b;
    if (n >= 8) {
      memcpy(&a, p, 8);
      memcpy(&b, p + n - 8, 8);
      for (size_t i = 8; i + 8 < n; i += 8) {
        uint64_t c;
        memcpy(&c, p + i, 8);
        a = (a ^ c) * 0x9E3779B97F4A7C15ull;
        a ^= a >> 29;
      }
    } else if (n >= 4) {
      uint32_t x, y;
      memcpy(&x, p, 4);
      memcpy(&y, p + n - 4, 4);
      a = x;
      b = y;
    } else if (n > 0) {
      a = (uint8_t)p[0] | ((uint64_t)(uint8_t)p[n >> 1] << 8) | ((uint64_t)(uint8_t)p[n - 1] << 16);
      b = 0;
    } else {
      a = b = 0;
    }
    uint64_t h = (a ^ 0xff51afd7ed558ccdULL) * (b ^ 0xc4ceb9fe1a85ec53ULL ^ (uint64_t)n);
    h ^= h >> 32;
    h *= 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    return h;
  }
  // n equal bytes?  (inline: the names are ~10 bytes and a libc call per comparison was a third of the tokenizer)
  static inline bool same_bytes(const char *x, const char *y, size_t n) {
    if (n >= 8) {
      size_t i = 0;
      for (; i + 8 <= n; i += 8) {
        uint64_t a, b;
        memcpy(&a, x + i, 8);
        memcpy(&b, y + i, 8);
        if (a != b) return false;
      }
      if (i < n) {  // the last eight bytes, overlapping what was compared
        uint64_t a, b;
        memcpy(&a, x + n - 8, 8);
        memcpy(&b, y + n - 8, 8);
        return a == b;
      }
      return true;
    }
    for (size_t i = 0; i < n; ++i)
      if (x[i] != y[i]) return false;
    return true;
  }
  inline int32_t find(uint32_t slot, sv a) const {
    const uint64_t h = hash(a);
    const uint32_t m = mask[slot];
    const Ent *t = tab[slot].data();
    for (uint32_t i = (uint32_t)h & m;; i = (i + 1) & m) {
      const Ent &e = t[i];
      if (!e.used) return -1;
      if (e.h == h && e.len == a.size() && same_bytes(pool.data() + e.off, a.data(), a.size())) return (int32_t)e.id;
    }
  }
  inline sv name(uint32_t slot, uint32_t id) const {
    const uint32_t o = name_off[slot][id];
    return sv(pool.data() + o, name_off[slot][id + 1] - o);
  }
  inline int32_t find_locus(sv l) const {
    for (const Locus &x : loci)
      if (x.name.size() == l.size() && same_bytes(x.name.data(), l.data(), l.size())) return (int32_t)x.slot;
    return -1;
  }
};
void dict_snapshot(const grim_dict *d, DictSnap &out);

// ---- tokenizer core: a byte range of whole lines -> per-line outcome, subject records, tokens -----------------------
// K_UNSUPPORTED: reason 5 (more alleles than a key field holds); K_UNSUPPORTED_GL: reason 8 (a GL string that names a locus
// twice or mixes loci in one entry: the reference pairs the entries by index after a per-side string sort, impute.py:246-272,
// and carries on with haplotypes over repeated loci -- not representable in a key, so reported, never silently different)
enum { K_DEV = 0, K_PROBLEM_ID = 1, K_PROBLEM_RAW = 2, K_MISS_NO_DEVICE = 3, K_UNSUPPORTED = 4, K_UNSUPPORTED_GL = 5 };

struct LineInfo {
  uint64_t off;     // start of the line in the text
  uint32_t len;     // without trailing white space
  uint32_t id_len;  // the subject id is the first id_len bytes
};

struct OvEnt {  // an allele outside the dictionary: (line, slot, id) -> its text (a copy in the range's ov_pool)
  uint32_t line;  // line number inside the range
  uint16_t id;
  uint8_t slot;
  uint8_t pad;
  uint32_t len;
  uint64_t off;
};

// resolves a (race1, race2) pair to the index of its prior matrix; must be callable from several threads
struct RaceResolver {
  virtual ~RaceResolver() {}
  virtual uint32_t resolve(sv r1, sv r2) = 0;
};

// per-id phase masks (bin_imputation_in_file, impute.py:2001-2005, 2030-2032): id -> bitmask of positions that keep
// their side in every phase; an id that is not in the table sends its line to .problem (KeyError in the reference)
struct MaskTable {
  std::unordered_map<std::string, uint8_t> fixed;
};

struct TokParams {
  const DictSnap *snap;
  bool planb;
  RaceResolver *races;
  const MaskTable *masks;     // or null
  const ClassRule *classify;  // or null: no class lists
  // device tokenizer (slab mode only): line_dst[k] = the k-th line of the range.  A line whose GL field has no '/' list
  // gets its record filled in and is left to the device (kind K_DEV, no subject record, no tokens, no class list entry);
  // every other line gets gl_len = 0 and is tokenised here as usual.  null: everything is tokenised here.
  LineRec *line_dst = nullptr;
};

struct TokRange {
  // per line of the range
  std::vector<uint8_t> kind;
  std::vector<LineInfo> line;
  std::vector<int32_t> dev;  // dense mode: subject number inside the range, or -1
  // subjects and tokens: dense mode appends to the vectors; slab mode writes subject k at subj_dst[line number] and
  // tokens into tok_dst[0 .. tok_cap) (tok_off = tok_base + position)
  bool dense = true;
  std::vector<grim_subject> subj;
  std::vector<uint16_t> tok;
  grim_subject *subj_dst = nullptr;
  uint16_t *tok_dst = nullptr;
  uint64_t tok_cap = 0, tok_base = 0;
  uint64_t n_tok = 0;
  uint32_t n_subj = 0;
  uint32_t n_devtok = 0;  // lines left to the device tokenizer
  std::vector<OvEnt> ov;
  std::string ov_pool;
  bool race_overflow = false;  // more than 65534 distinct race pairs
  // class lists (subject numbers as the device will see them: first_subject + ...), when TokParams.classify is set
  uint32_t first_subject = 0;
  std::vector<uint32_t> os, om, og;
  std::vector<SmallRec> small;
  void clear() {
    kind.clear(); line.clear(); dev.clear(); subj.clear(); tok.clear(); ov.clear(); ov_pool.clear(); os.clear(); om.clear(); og.clear();
    small.clear();
    n_tok = 0; n_subj = 0; n_devtok = 0; race_overflow = false;
  }
};

// text[lo, hi) must consist of whole lines ('\n' terminated, except possibly the last one)
void tokenize_range(const TokParams &prm, const char *text, uint64_t lo, uint64_t hi, TokRange &out);
// Lines the device tokenizer handed back (numbers inside the range, ascending; slab mode): tokenised here after all, with
// the range's own state -- subject records at subj_dst[line], tokens behind the range's, overlay alleles, kinds -- and
// appended to the class lists given.  lrec: the range's line records (GL extent, prior index).
void tokenize_lines(const TokParams &prm, const char *text, TokRange &R, const std::vector<uint32_t> &lines, const LineRec *lrec,
                    std::vector<uint32_t> &os, std::vector<uint32_t> &om, std::vector<uint32_t> &og, std::vector<SmallRec> &small);

// ---- formatter core ---------------------------------------------------------------------------------------------------
struct OutBuf {
  char *p = nullptr;
  size_t n = 0, cap = 0;
  ~OutBuf() { free(p); }
  OutBuf() {}
  OutBuf(const OutBuf &) = delete;
  OutBuf &operator=(const OutBuf &) = delete;
  inline char *room(size_t k) {
    if (n + k > cap) grow(k);
    return p + n;
  }
  void grow(size_t k);
  inline void put(sv s) {
    char *q = room(s.size());
    memcpy(q, s.data(), s.size());
    n += s.size();
  }
  inline void put(char c) {
    *room(1) = c;
    ++n;
  }
  void clear() { n = 0; }
};

struct FmtParams {
  const DictSnap *snap;
  const grim_params *prm;
  std::vector<std::string> pops;
  bool want_log = false;     // also the per-subject stdout lines of impute_file (impute.py:2074-2078, 2108-2112, 2138, 2142)
  double per_subject_s = 0;  // the number printed after a subject's lines
};

struct FmtRange {
  OutBuf t[7];  // umug, umug_pops, pmug, pmug_pops, miss, problem, log
  std::vector<uint32_t> unsupported;  // line numbers (inside the range) of GRIM_ST_UNSUPPORTED subjects, left out of every text
};

// lines [0, n) of a tokenized range; res/rows as the device returned them; subject number of line j = dev ? dev[j]
// : first_subject + j; first_line = global index of line 0 (the i of "i,id" in .miss/.problem); skip: optional per line
void format_range(const FmtParams &fp, const char *text, const TokRange &tr, const grim_subject_result *res, const grim_row *rows,
                  uint64_t first_line, const uint8_t *skip, FmtRange &out);

// calc_priority_matrix (impute.py:1844-1924) for one race pair, in the reference's operation order; out[P*P]
struct PriorSpec {
  double alpha, eta, beta, gamma, delta;
  bool unk_mr;                         // UNK_priors == "MR": all-ones base matrix, else identity
  std::vector<double> count_by_prob;   // [P]
  std::vector<std::string> pops;
};
void prior_matrix(const PriorSpec &ps, sv race1, sv race2, double *out);
