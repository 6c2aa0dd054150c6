// grim_graph_host.cpp -- the graph's host side in C++: no GPU code.
//
//   grim_graphgen_csv        hpf.csv -> nodes.csv, edges.csv, top_links.csv, info_node.csv
//                            (graph_generation/generate_neo4j_multi_hpf.py:209-486: same rows, same order,
//                            same number text; top_links.csv rows ascending per node, see the Python twin)
//   grim_hostgraph_load_csv  the three CSVs -> the integer arrays grim_graph_upload takes
//                            (Graph.build_graph, networkx_graph.py:42-213, quirks of :157-198 included)
//
// Python twins (kept as the cross-check in tests/): graph_generation/generate_neo4j_multi_hpf.py and
// grim/imputation/networkx_graph.py::Graph._build_graph_python.
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <charconv>
#include <memory>
#include <numeric>
#include <string>

#include "grim_host_internal.h"

namespace {

void set_err(char *err, uint64_t cap, const std::string &msg) {
  if (!err || cap == 0) return;
  size_t n = std::min<size_t>(msg.size(), (size_t)cap - 1);
  memcpy(err, msg.data(), n);
  err[n] = 0;
}

bool read_file(const char *path, std::string &out) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize((size_t)(n > 0 ? n : 0));
  size_t got = n > 0 ? fread(&out[0], 1, (size_t)n, f) : 0;
  fclose(f);
  return got == out.size();
}

bool write_file(const char *path, const std::string &s) {
  FILE *f = fopen(path, "wb");
  if (!f) return false;
  size_t put = s.empty() ? 0 : fwrite(s.data(), 1, s.size(), f);
  return fclose(f) == 0 && put == s.size();
}

// next line of a text blob without its terminator ("\n" or "\r\n"); false at the end
bool next_line(const std::string &text, size_t &pos, sv &line) {
  if (pos >= text.size()) return false;
  size_t e = text.find('\n', pos);
  if (e == std::string::npos) e = text.size();
  size_t len = e - pos;
  if (len && text[pos + len - 1] == '\r') --len;
  line = sv(text.data() + pos, len);
  pos = e + 1;
  return true;
}

void split_sv(sv s, char sep, std::vector<sv> &out) {
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

bool parse_double(sv s, double &v) {  // float(str): surrounding blanks allowed
  while (!s.empty() && (s.front() == ' ' || s.front() == '\t')) s.remove_prefix(1);
  while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r' || s.back() == '\n')) s.remove_suffix(1);
  if (!s.empty() && s.front() == '+') s.remove_prefix(1);
  auto r = std::from_chars(s.data(), s.data() + s.size(), v);
  return r.ec == std::errc() && r.ptr == s.data() + s.size();
}

// row starts of a CSR over `n_vertices` from the sorted source column, as networkx_graph.py:157-198 builds
// them: a vertex without out-edges copies the previous start, the closing sentinel is the VERTEX count
bool row_starts(const std::vector<uint32_t> &src_sorted, uint32_t n_vertices, std::vector<uint32_t> &starts, std::string &err) {
  starts.assign((size_t)n_vertices + 1, 0);
  if (src_sorted.empty() || src_sorted.back() != n_vertices - 1) {
    err = "graph: the highest-numbered vertex has no out-edges (reference cannot load this graph either)";
    return false;
  }
  std::vector<int64_t> first((size_t)n_vertices, -1);
  for (size_t i = src_sorted.size(); i-- > 0;) first[src_sorted[i]] = (int64_t)i;
  uint32_t prev = 0;
  for (uint32_t v = 0; v < n_vertices; ++v) {
    if (first[v] >= 0) prev = (uint32_t)first[v];
    starts[v] = prev;
  }
  starts[n_vertices] = n_vertices;  // QUIRK: sentinel = len(Vertices)
  return true;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// loader
// ------------------------------------------------------------------------------------------------
struct grim_hostgraph {
  uint32_t n_nodes = 0, n_pops = 0, n_loci = 0, full_mask = 0, n_conn = 0;
  std::vector<uint64_t> node_key;
  std::vector<uint8_t> node_mask;
  std::vector<double> freq;
  std::vector<uint32_t> a_start, a_nbr, b_conn, b_start, b_nbr, lab_start, lab_nodes;
};

// ids_are_rows: every node id is the decimal text of its row number (what the generator writes), so an id is
// parsed instead of looked up
static bool read_pairs(const std::string &text, const char *path, const std::unordered_map<std::string, uint32_t> &id_to_row,
                       bool ids_are_rows, uint32_t n_nodes, std::vector<uint32_t> &a, std::vector<uint32_t> &b, std::string &err) {
  size_t pos = 0;
  sv line;
  next_line(text, pos, line);  // header
  std::string k;
  while (next_line(text, pos, line)) {
    if (line.empty()) continue;
    size_t c1 = line.find(',');
    if (c1 == sv::npos) {
      err = std::string("malformed row in ") + path;
      return false;
    }
    size_t c2 = line.find(',', c1 + 1);
    sv f0 = line.substr(0, c1), f1 = line.substr(c1 + 1, c2 == sv::npos ? sv::npos : c2 - c1 - 1);
    if (ids_are_rows) {
      uint32_t x = 0, y = 0;
      auto r0 = std::from_chars(f0.data(), f0.data() + f0.size(), x);
      auto r1 = std::from_chars(f1.data(), f1.data() + f1.size(), y);
      if (r0.ec == std::errc() && r0.ptr == f0.data() + f0.size() && r1.ec == std::errc() && r1.ptr == f1.data() + f1.size() &&
          x < n_nodes && y < n_nodes && (f0.size() == 1 || f0[0] != '0') && (f1.size() == 1 || f1[0] != '0')) {
        a.push_back(x);
        b.push_back(y);
        continue;
      }
    }
    k.assign(f0);
    auto i0 = id_to_row.find(k);
    k.assign(f1);
    auto i1 = id_to_row.find(k);
    if (i0 == id_to_row.end() || i1 == id_to_row.end()) {
      err = std::string("unknown node id in ") + path;
      return false;
    }
    a.push_back(i0->second);
    b.push_back(i1->second);
  }
  return true;
}

// the three CSV texts -> arrays (the texts come from files, or straight from the generator: grim_hostgraph_from_hpf)
static grim_hostgraph *load_texts(grim_dict *d, const char *full_loci, const std::string &nodes_text, const std::string &top_text,
                                  const std::string &edges_text, const char *nodes_csv, const char *top_links_csv,
                                  const char *edges_csv, char *err, uint64_t err_cap) {
  std::string e;
  auto fail = [&](const std::string &m) -> grim_hostgraph * {
    set_err(err, err_cap, m);
    return nullptr;
  };
  const uint32_t nl = (uint32_t)strlen(full_loci);
  if (nl == 0 || nl > GRIM_MAXL || nl != d->n_loci) return fail("grim_hostgraph_load_csv: locus count mismatch");
  grim_hostgraph *h = new grim_hostgraph();
  std::unique_ptr<grim_hostgraph> guard(h);
  h->n_loci = nl;
  h->full_mask = (1u << nl) - 1u;
  // ---- nodes.csv: id, name, label, freq;freq;... (networkx_graph.py:48-66) -------------------------
  std::unordered_map<std::string, uint32_t> id_to_row;
  bool ids_are_rows = true;
  {
    const std::string &text = nodes_text;
    size_t pos = 0;
    sv line;
    next_line(text, pos, line);
    std::vector<sv> f, parts;
    while (next_line(text, pos, line)) {
      if (line.empty()) continue;
      split_sv(line, ',', f);
      if (f.size() < 4) return fail("nodes.csv: row with fewer than 4 fields");
      uint64_t key = 0;
      uint32_t mask = 0;
      split_sv(f[1], '~', parts);
      for (sv al : parts) {
        size_t star = al.find('*');
        auto it = d->locus_slot.find(std::string(al.substr(0, star)));
        if (it == d->locus_slot.end()) return fail("nodes.csv: allele of an unknown locus: " + std::string(al));
        const uint32_t s = it->second;
        const int32_t id = dict_intern(d, s, al);
        if (id < 0) return fail("more than 4093 alleles at one locus");
        key |= (uint64_t)(id + 1) << (GRIM_ABITS * s);
        mask |= 1u << s;
      }
      uint32_t lab_mask = 0;
      for (char ch : f[2]) {
        const char *p = strchr(full_loci, ch);
        if (!p) return fail("nodes.csv: label character outside FULL_LOCI");
        lab_mask |= 1u << (uint32_t)(p - full_loci);
      }
      if (mask != lab_mask) return fail("node " + std::string(f[1]) + ": alleles do not match its label " + std::string(f[2]));
      split_sv(f[3], ';', parts);
      if (h->n_pops == 0) h->n_pops = (uint32_t)parts.size();
      if (parts.size() != h->n_pops) return fail("nodes.csv: rows with different numbers of frequencies");
      for (sv x : parts) {
        double v;
        if (!parse_double(x, v)) return fail("nodes.csv: bad frequency '" + std::string(x) + "'");
        h->freq.push_back(v);
      }
      {
        uint32_t idv = 0;
        auto r = std::from_chars(f[0].data(), f[0].data() + f[0].size(), idv);
        if (!(r.ec == std::errc() && r.ptr == f[0].data() + f[0].size() && idv == h->node_key.size() &&
              (f[0].size() == 1 || f[0][0] != '0')))
          ids_are_rows = false;
      }
      id_to_row[std::string(f[0])] = (uint32_t)h->node_key.size();
      h->node_key.push_back(key);
      h->node_mask.push_back((uint8_t)mask);
    }
  }
  const uint32_t V = (uint32_t)h->node_key.size();
  h->n_nodes = V;
  if (V == 0) return fail("nodes.csv holds no node");
  auto nbits = [](uint32_t m) { return (uint32_t)__builtin_popcount(m); };
  // ---- plan A: partial -> full (networkx_graph.py:71-88, 136-207) --------------------------------------
  {
    std::vector<uint32_t> n1, n2;
    if (!read_pairs(top_text, top_links_csv, id_to_row, ids_are_rows, V, n1, n2, e)) return fail(e);
    std::vector<uint64_t> pr(n1.size());
    for (size_t i = 0; i < n1.size(); ++i) {
      const bool flip = h->node_mask[n1[i]] == h->full_mask;
      const uint32_t src = flip ? n2[i] : n1[i], dst = flip ? n1[i] : n2[i];
      pr[i] = ((uint64_t)src << 32) | dst;
    }
    std::sort(pr.begin(), pr.end());
    std::vector<uint32_t> src(pr.size());
    h->a_nbr.resize(pr.size());
    for (size_t i = 0; i < pr.size(); ++i) {
      src[i] = (uint32_t)(pr[i] >> 32);
      h->a_nbr[i] = (uint32_t)pr[i];
    }
    if (!row_starts(src, V, h->a_start, e)) return fail(e);
  }
  // ---- plan B: child -> connector(parent label, child) -> parents (networkx_graph.py:91-130) ------------
  {
    std::vector<uint32_t> n1, n2;
    if (!read_pairs(edges_text, edges_csv, id_to_row, ids_are_rows, V, n1, n2, e)) return fail(e);
    const size_t E = n1.size();
    std::vector<uint32_t> child(E), parent(E), conn(E);
    std::unordered_map<uint64_t, uint32_t> conn_id;  // (child, parent label) -> id, in order of first appearance
    conn_id.reserve(E);
    h->b_conn.assign((size_t)V * GRIM_MAXL, 0xFFFFFFFFu);
    for (size_t i = 0; i < E; ++i) {
      const bool first_is_child = nbits(h->node_mask[n1[i]]) < nbits(h->node_mask[n2[i]]);
      child[i] = first_is_child ? n1[i] : n2[i];
      parent[i] = first_is_child ? n2[i] : n1[i];
      const uint32_t pmask = h->node_mask[parent[i]];
      const uint64_t ck = (uint64_t)child[i] * 64u + pmask;
      auto it = conn_id.find(ck);
      if (it == conn_id.end()) it = conn_id.emplace(ck, (uint32_t)conn_id.size()).first;
      conn[i] = it->second;
      const uint32_t added = pmask & ~(uint32_t)h->node_mask[child[i]];
      if (nbits(added) != 1) return fail("edges.csv holds a parent that is not exactly one locus larger than its child");
      h->b_conn[(size_t)child[i] * GRIM_MAXL + (uint32_t)__builtin_ctz(added)] = conn[i];
    }
    const uint32_t ncon = (uint32_t)conn_id.size();
    h->n_conn = ncon;
    std::vector<uint64_t> w(2 * E);
    for (size_t i = 0; i < E; ++i) {
      w[i] = ((uint64_t)child[i] << 32) | (V + conn[i]);
      w[E + i] = ((uint64_t)(V + conn[i]) << 32) | parent[i];
    }
    std::sort(w.begin(), w.end());
    w.erase(std::unique(w.begin(), w.end()), w.end());  // drop_duplicates + lexsort (:142-150)
    std::vector<uint32_t> src(w.size()), w_start;
    h->b_nbr.resize(w.size());
    for (size_t i = 0; i < w.size(); ++i) {
      src[i] = (uint32_t)(w[i] >> 32);
      h->b_nbr[i] = (uint32_t)w[i];
    }
    if (!row_starts(src, V + ncon, w_start, e)) return fail(e);
    h->b_start.assign(w_start.begin() + V, w_start.end());
  }
  // ---- nodes grouped by label, in id order (haps_by_label, networkx_graph.py:215-236) --------------------
  {
    h->lab_start.assign((1u << GRIM_MAXL) + 1, 0);
    for (uint32_t i = 0; i < V; ++i) h->lab_start[h->node_mask[i] + 1]++;
    for (uint32_t m = 0; m < (1u << GRIM_MAXL); ++m) h->lab_start[m + 1] += h->lab_start[m];
    std::vector<uint32_t> cur(h->lab_start.begin(), h->lab_start.end() - 1);
    h->lab_nodes.resize(V);
    for (uint32_t i = 0; i < V; ++i) h->lab_nodes[cur[h->node_mask[i]]++] = i;
  }
  guard.release();
  return h;
}

extern "C" grim_hostgraph *grim_hostgraph_load_csv(grim_dict *d, const char *full_loci, const char *nodes_csv,
                                                   const char *top_links_csv, const char *edges_csv, char *err, uint64_t err_cap) {
  if (!d || !full_loci || !nodes_csv || !top_links_csv || !edges_csv) {
    set_err(err, err_cap, "grim_hostgraph_load_csv: null argument");
    return nullptr;
  }
  std::string nodes, top, edges;
  for (auto pr : {std::make_pair(nodes_csv, &nodes), std::make_pair(top_links_csv, &top), std::make_pair(edges_csv, &edges)})
    if (!read_file(pr.first, *pr.second)) {
      set_err(err, err_cap, std::string("cannot read ") + pr.first);
      return nullptr;
    }
  return load_texts(d, full_loci, nodes, top, edges, nodes_csv, top_links_csv, edges_csv, err, err_cap);
}

extern "C" int grim_hostgraph_desc(const grim_hostgraph *h, grim_graph_desc *out) {
  if (!h || !out) return -1;
  out->n_nodes = h->n_nodes;
  out->n_pops = h->n_pops;
  out->n_loci = h->n_loci;
  out->full_mask = h->full_mask;
  out->node_key = h->node_key.data();
  out->node_mask = h->node_mask.data();
  out->freq = h->freq.data();
  out->a_start = h->a_start.data();
  out->a_nbr = h->a_nbr.data();
  out->n_a_nbr = h->a_nbr.size();
  out->b_conn = h->b_conn.data();
  out->b_start = h->b_start.data();
  out->b_nbr = h->b_nbr.data();
  out->n_conn = h->n_conn;
  out->n_b_nbr = h->b_nbr.size();
  out->lab_start = h->lab_start.data();
  out->lab_nodes = h->lab_nodes.data();
  out->label_order_bad = 0;  // (the caller knows the loci_map: grim/imputation/networkx_graph.py sets it)
  out->reserved = 0;
  return 0;
}

extern "C" void grim_hostgraph_free(grim_hostgraph *h) { delete h; }

// ------------------------------------------------------------------------------------------------
// generator
// ------------------------------------------------------------------------------------------------
namespace {

struct Num {  // a Python number that is either the int 0 (never touched by a float) or a float
  double v = 0.0;
  bool is_float = false;
};

void put_num(const Num &n, std::string &out) {
  if (n.is_float)
    py_float(n.v, out);
  else
    out.push_back('0');
}

}  // namespace

// generator core.  Each of the four CSVs goes to its file when a path is given and stays in keep[k] when keep is not null
// (k: 0 nodes, 1 edges, 2 top links, 3 info) -- the direct hpf -> arrays route hands the texts to the loader without files.
static int graphgen_core(const char *hpf_csv, const char *const *pops, const double *cutoff, uint32_t n_pops,
                         const char *const *locus_names, const uint32_t *locus_index, uint32_t n_locus_names,
                         const char *nodes_csv, const char *edges_csv, const char *top_links_csv, const char *info_csv,
                         std::string *keep, char *err, uint64_t err_cap) {
  auto fail = [&](const std::string &m) {
    set_err(err, err_cap, m);
    return -1;
  };
  auto emit = [&](int k, const char *path, std::string &text) -> bool {
    if (path && !write_file(path, text)) return false;
    if (keep) keep[k].swap(text);
    return true;
  };
  if (!hpf_csv || !pops || !cutoff || !locus_names || !locus_index) return fail("grim_graphgen_csv: null argument");
  // label characters: the sorted set of str(index) (generate_neo4j_multi_hpf.py:101-110)
  std::vector<std::string> chars;
  for (uint32_t i = 0; i < n_locus_names; ++i) chars.push_back(std::to_string(locus_index[i]));
  std::sort(chars.begin(), chars.end());
  chars.erase(std::unique(chars.begin(), chars.end()), chars.end());
  std::string full;
  for (auto &c : chars) full += c;
  const uint32_t nloc = (uint32_t)full.size();
  if (nloc == 0 || nloc > GRIM_MAXL || nloc != chars.size()) return fail("grim_graphgen_csv: 1..5 single-digit locus indexes expected");
  std::unordered_map<std::string, uint32_t> loc_pos;  // locus name -> position index-1 inside a canonical name
  for (uint32_t i = 0; i < n_locus_names; ++i) {
    if (locus_index[i] == 0 || locus_index[i] > nloc) return fail("grim_graphgen_csv: locus index outside 1..n");
    loc_pos[locus_names[i]] = locus_index[i] - 1;
  }
  std::unordered_map<std::string, uint32_t> pop_idx;
  for (uint32_t p = 0; p < n_pops; ++p) pop_idx[pops[p]] = p;

  // ---- full haplotypes, first-seen order (:289-339) ---------------------------------------------------
  std::string text;
  if (!read_file(hpf_csv, text)) return fail(std::string("cannot read ") + hpf_csv);
  std::vector<std::unordered_map<std::string, uint32_t>> al_id(nloc);  // per position: allele text -> local id (0 = absent)
  std::vector<std::vector<std::string>> al_name(nloc, std::vector<std::string>(1, "0"));
  std::unordered_map<std::string, uint32_t> full_index;  // canonical name -> full haplotype number
  std::vector<std::string> full_name;
  std::vector<std::array<uint32_t, GRIM_MAXL>> full_slots;
  std::vector<Num> full_freq;  // [n_full][n_pops]
  {
    size_t pos = 0;
    sv line;
    std::vector<sv> f, parts;
    std::string name;
    while (next_line(text, pos, line)) {
      split_sv(line, ',', f);
      if (f.size() != 3) return fail("hpf.csv: a row does not have 3 fields");
      if (f[0] == "hap") continue;
      double fr;
      if (!parse_double(f[2], fr)) return fail("hpf.csv: bad frequency '" + std::string(f[2]) + "'");
      auto pi = pop_idx.find(std::string(f[1]));
      if (pi == pop_idx.end()) return fail("hpf.csv: population '" + std::string(f[1]) + "' is not configured");  // KeyError there
      if (fr == 0.0 || fr < cutoff[pi->second]) continue;
      std::array<sv, GRIM_MAXL> slot_txt;
      for (uint32_t i = 0; i < nloc; ++i) slot_txt[i] = sv("0");
      split_sv(f[0], '~', parts);
      for (sv a : parts) {
        if (!a.empty() && a.back() == 'g') a.remove_suffix(1);
        auto lp = loc_pos.find(std::string(a.substr(0, a.find('*'))));
        if (lp == loc_pos.end()) return fail("hpf.csv: allele of an unknown locus: " + std::string(a));
        slot_txt[lp->second] = a;
      }
      name.clear();
      for (uint32_t i = 0; i < nloc; ++i) {
        if (i) name.push_back('~');
        name += slot_txt[i];
      }
      auto it = full_index.find(name);
      uint32_t fi;
      if (it == full_index.end()) {
        fi = (uint32_t)full_name.size();
        full_index.emplace(name, fi);
        full_name.push_back(name);
        std::array<uint32_t, GRIM_MAXL> ids{};
        for (uint32_t i = 0; i < nloc; ++i) {
          if (slot_txt[i] == "0") continue;
          std::string k(slot_txt[i]);
          auto ai = al_id[i].find(k);
          if (ai == al_id[i].end()) {
            ai = al_id[i].emplace(k, (uint32_t)al_name[i].size()).first;
            al_name[i].push_back(k);
          }
          ids[i] = ai->second;
        }
        full_slots.push_back(ids);
        full_freq.resize(full_freq.size() + n_pops);
      } else {
        fi = it->second;
      }
      Num &cell = full_freq[(size_t)fi * n_pops + pi->second];
      cell.v = fr;  // a repeated (population, haplotype) row keeps the later value (dict assignment)
      cell.is_float = true;
    }
  }
  const uint32_t n_full = (uint32_t)full_name.size();
  for (uint32_t i = 0; i < nloc; ++i)
    if (al_name[i].size() > 0xFFFu) return fail("more than 4094 alleles at one locus");
  // a haplotype without an allele at some locus would make the partial labels ambiguous; the reference's
  // names then hold "0" placeholders -- keep them as allele id 0 (text "0")
  // ---- labels: full first, then smaller subsets, larger first, combinations() order (:101-110) -------------
  std::vector<uint32_t> labels;  // position bitmasks
  labels.push_back((1u << nloc) - 1u);
  for (uint32_t r = nloc - 1; r >= 1; --r) {
    // combinations(range(nloc), r) in lexicographic order
    std::vector<uint32_t> idx(r);
    std::iota(idx.begin(), idx.end(), 0u);
    for (;;) {
      uint32_t m = 0;
      for (uint32_t i : idx) m |= 1u << i;
      labels.push_back(m);
      int k = (int)r - 1;
      while (k >= 0 && idx[k] == nloc - r + (uint32_t)k) --k;
      if (k < 0) break;
      ++idx[k];
      for (uint32_t j = (uint32_t)k + 1; j < r; ++j) idx[j] = idx[j - 1] + 1;
    }
  }
  auto label_text = [&](uint32_t m) {
    std::string s;
    for (uint32_t i = 0; i < nloc; ++i)
      if ((m >> i) & 1u) s.push_back(full[i]);
    return s;
  };
  auto proj_name = [&](uint32_t f, uint32_t m, std::string &out) {
    bool first = true;
    for (uint32_t i = 0; i < nloc; ++i)
      if ((m >> i) & 1u) {
        if (!first) out.push_back('~');
        first = false;
        out += al_name[i][full_slots[f][i]];
      }
  };
  // ---- partial nodes (:364-415): ids in creation order, label by label ------------------------------------
  struct Part {
    uint32_t mask;
    uint32_t first_id;                 // id of its first node
    std::vector<uint32_t> node_of;     // [n_full] -> local node
    std::vector<uint32_t> rep;         // local node -> a representative full haplotype (the first)
    std::vector<uint32_t> start, mem;  // CSR local node -> contributing full haplotypes, ascending
    std::vector<Num> freq;             // [n_nodes][n_pops]
  };
  std::vector<Part> part(labels.size());
  std::vector<int> label_slot(1u << nloc, -1);
  uint32_t next_id = n_full;
  for (size_t li = 1; li < labels.size(); ++li) {
    Part &pt = part[li];
    pt.mask = labels[li];
    pt.first_id = next_id;
    label_slot[pt.mask] = (int)li;
    pt.node_of.resize(n_full);
    std::unordered_map<uint64_t, uint32_t> seen;
    seen.reserve(n_full);
    std::vector<uint32_t> cnt;
    for (uint32_t f = 0; f < n_full; ++f) {
      uint64_t key = 0;  // 12 bits per position (allele counts checked above)
      for (uint32_t i = 0; i < nloc; ++i)
        if ((pt.mask >> i) & 1u) key |= (uint64_t)(full_slots[f][i] & 0xFFFu) << (12 * i);
      auto it = seen.find(key);
      if (it == seen.end()) {
        it = seen.emplace(key, (uint32_t)pt.rep.size()).first;
        pt.rep.push_back(f);
        cnt.push_back(0);
      }
      pt.node_of[f] = it->second;
      cnt[it->second]++;
    }
    const uint32_t nn = (uint32_t)pt.rep.size();
    next_id += nn;
    pt.start.assign(nn + 1, 0);
    for (uint32_t k = 0; k < nn; ++k) pt.start[k + 1] = pt.start[k] + cnt[k];
    pt.mem.resize(n_full);
    std::vector<uint32_t> cur(pt.start.begin(), pt.start.end() - 1);
    for (uint32_t f = 0; f < n_full; ++f) pt.mem[cur[pt.node_of[f]]++] = f;
    pt.freq.assign((size_t)nn * n_pops, Num());
    for (uint32_t f = 0; f < n_full; ++f) {  // running sums in full-haplotype order (:398-405)
      Num *dst = &pt.freq[(size_t)pt.node_of[f] * n_pops];
      const Num *src = &full_freq[(size_t)f * n_pops];
      for (uint32_t p = 0; p < n_pops; ++p) {
        if (src[p].is_float) {
          dst[p].v = dst[p].is_float ? dst[p].v + src[p].v : 0.0 + src[p].v;
          dst[p].is_float = true;
        }
      }
    }
  }
  // ---- nodes.csv (:341-358, 419-430) -------------------------------------------------------------------------
  std::string out;
  out.reserve((size_t)next_id * 64);
  out += "haplotypeId:ID(HAPLOTYPE),name,loci:LABEL,frequency:DOUBLE[]\r\n";
  for (uint32_t f = 0; f < n_full; ++f) {
    out += std::to_string(f);
    out.push_back(',');
    out += full_name[f];
    out.push_back(',');
    out += full;
    out.push_back(',');
    for (uint32_t p = 0; p < n_pops; ++p) {
      if (p) out.push_back(';');
      put_num(full_freq[(size_t)f * n_pops + p], out);
    }
    out += "\r\n";
  }
  for (size_t li = 1; li < labels.size(); ++li) {
    const Part &pt = part[li];
    const std::string lt = label_text(pt.mask);
    for (uint32_t k = 0; k < pt.rep.size(); ++k) {
      out += std::to_string(pt.first_id + k);
      out.push_back(',');
      proj_name(pt.rep[k], pt.mask, out);
      out.push_back(',');
      out += lt;
      out.push_back(',');
      for (uint32_t p = 0; p < n_pops; ++p) {
        if (p) out.push_back(';');
        put_num(pt.freq[(size_t)k * n_pops + p], out);
      }
      out += "\r\n";
    }
  }
  if (!emit(0, nodes_csv, out)) return fail(std::string("cannot write ") + nodes_csv);
  // ---- edges.csv (:82-97, 434-455): per child node, per contributing haplotype, per added locus -----------------
  out.clear();
  out += ":START_ID(HAPLOTYPE),:END_ID(HAPLOTYPE),CP:DOUBLE[],:TYPE\r\n";
  for (size_t li = 1; li < labels.size(); ++li) {
    const Part &pt = part[li];
    for (uint32_t k = 0; k < pt.rep.size(); ++k) {
      const Num *cf = &pt.freq[(size_t)k * n_pops];
      for (uint32_t q = pt.start[k]; q < pt.start[k + 1]; ++q) {
        const uint32_t f = pt.mem[q];
        for (uint32_t i = 0; i < nloc; ++i) {
          if ((pt.mask >> i) & 1u) continue;
          const uint32_t pm = pt.mask | (1u << i);
          uint32_t pid;
          if (pm == labels[0])
            pid = f;
          else {
            const Part &pp = part[(size_t)label_slot[pm]];
            pid = pp.first_id + pp.node_of[f];
          }
          out += std::to_string(pt.first_id + k);
          out.push_back(',');
          out += std::to_string(pid);
          out.push_back(',');
          for (uint32_t p = 0; p < n_pops; ++p) {
            if (p) out.push_back(';');
            const Num &a = full_freq[(size_t)f * n_pops + p];
            // 0 if c == 0 else a / c   (c == 0 only as the untouched int 0)
            if (!cf[p].is_float || cf[p].v == 0.0)
              out.push_back('0');
            else
              py_float((a.is_float ? a.v : 0.0) / cf[p].v, out);
          }
          out += ",CP\r\n";
        }
      }
    }
  }
  if (!emit(1, edges_csv, out)) return fail(std::string("cannot write ") + edges_csv);
  // ---- top_links.csv (:305, 392-394, 470): ascending full-haplotype id per node ----------------------------------
  out.clear();
  out += ":START_ID(HAPLOTYPE),:END_ID(HAPLOTYPE),:TYPE\r\n";
  for (size_t li = 1; li < labels.size(); ++li) {
    const Part &pt = part[li];
    for (uint32_t k = 0; k < pt.rep.size(); ++k)
      for (uint32_t q = pt.start[k]; q < pt.start[k + 1]; ++q) {
        out += std::to_string(pt.first_id + k);
        out.push_back(',');
        out += std::to_string(pt.mem[q]);
        out += ",TOP\r\n";
      }
  }
  if (!emit(2, top_links_csv, out)) return fail(std::string("cannot write ") + top_links_csv);
  // ---- info_node.csv (:476-484) ----------------------------------------------------------------------------------
  out.clear();
  out += "INFO_NODE_ID:ID(INFO_NODE),populations:STRING[],INFO_NODE:LABEL\r\n1,";
  for (uint32_t p = 0; p < n_pops; ++p) {
    if (p) out.push_back(';');
    out += pops[p];
  }
  out += ",INFO_NODE\r\n";
  if (!emit(3, info_csv, out)) return fail(std::string("cannot write ") + info_csv);
  return 0;
}

extern "C" int grim_graphgen_csv(const char *hpf_csv, const char *const *pops, const double *cutoff, uint32_t n_pops,
                                 const char *const *locus_names, const uint32_t *locus_index, uint32_t n_locus_names,
                                 const char *nodes_csv, const char *edges_csv, const char *top_links_csv, const char *info_csv,
                                 char *err, uint64_t err_cap) {
  if (!nodes_csv || !edges_csv || !top_links_csv || !info_csv) {
    set_err(err, err_cap, "grim_graphgen_csv: null argument");
    return -1;
  }
  return graphgen_core(hpf_csv, pops, cutoff, n_pops, locus_names, locus_index, n_locus_names, nodes_csv, edges_csv, top_links_csv,
                       info_csv, nullptr, err, err_cap);
}

// hpf.csv -> the loaded graph in one call: generate_graph (generate_neo4j_multi_hpf.py:209-486) feeding Graph.build_graph
// (networkx_graph.py:42-213) without the four files in between.  The CSV paths may each be null; one that is given is
// written as grim_graphgen_csv would (drop-in: other tools read them).
extern "C" grim_hostgraph *grim_hostgraph_from_hpf(grim_dict *d, const char *full_loci, const char *hpf_csv, const char *const *pops,
                                                   const double *cutoff, uint32_t n_pops, const char *const *locus_names,
                                                   const uint32_t *locus_index, uint32_t n_locus_names, const char *nodes_csv,
                                                   const char *edges_csv, const char *top_links_csv, const char *info_csv,
                                                   char *err, uint64_t err_cap) {
  if (!d || !full_loci) {
    set_err(err, err_cap, "grim_hostgraph_from_hpf: null argument");
    return nullptr;
  }
  std::string text[4];
  if (graphgen_core(hpf_csv, pops, cutoff, n_pops, locus_names, locus_index, n_locus_names, nodes_csv, edges_csv, top_links_csv,
                    info_csv, text, err, err_cap) != 0)
    return nullptr;
  return load_texts(d, full_loci, text[0], text[2], text[1], "nodes.csv (in memory)", "top_links.csv (in memory)",
                    "edges.csv (in memory)", err, err_cap);
}
