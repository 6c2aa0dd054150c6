"""
CPU ORACLE for the grim.impute hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This module is an independent, string-level restatement (plain Python + a
little numpy) of the algorithm in nmdp-bioinformatics/py-graph-imputation's
`grim/imputation/impute.py` and `grim/imputation/networkx_graph.py`.  It exists
only so that tests, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` have something to check / time the HIP path against on machines where
the reference itself is not present.  Nothing under `py-graph-imputation_amd/`
imports it.

Parity pin: `tests/golden/` holds inputs + the six output files produced by the
REAL reference (imported from /root/reference in the build container by
`tools/make_golden.py`); `tests/test_oracle_golden.py` checks this restatement
against every one of them byte for byte.

Every function cites the reference lines (`impute.py:a-b`, `networkx_graph.py:a-b`)
whose behaviour it restates.  Quirks of the reference that influence results
are reproduced on purpose and are marked QUIRK.
"""

from __future__ import annotations

import csv
import json
import os
from collections import OrderedDict

import numpy as np

FACTOR_JOIN = 0.0001  # impute.py:196  (self.factor)


# --------------------------------------------------------------------------
# configuration  (run_impute_def.py:63-129, 188-192)
# --------------------------------------------------------------------------
DEFAULT_PLANB_MATRIX = [
    [[1, 2, 3, 4, 5]],
    [[1, 2, 3], [4, 5]],
    [[1], [2, 3], [4, 5]],
    [[1, 2, 3], [4], [5]],
    [[1], [2, 3], [4], [5]],
    [[1], [2], [3], [4], [5]],
]


def config_from_json(js, base_graph="", base_in=""):
    """Normalise a conf JSON dict the way run_impute_def.py:54-129,188-192 does."""
    gpath = js.get("graph_files_path")
    if gpath[-1] != "/":
        gpath += "/"
    out_dir = js.get("imputation_out_path", "output")
    if out_dir[-1] != "/":
        out_dir += "/"

    def outp(key):
        name = js.get(key)
        head, tail = os.path.split(name)
        return os.path.join(head, out_dir.rstrip("/"), tail)

    loci_map = dict(js.get("loci_map", {"A": 1, "B": 3, "C": 2, "DQB1": 4, "DRB1": 5}))
    cfg = {
        "planb": js.get("planb", True),
        "pops": js.get("populations"),
        "priority": js.get("priority"),
        "epsilon": js.get("epsilon", 1e-3),
        "number_of_results": js.get("number_of_results", 1000),
        "number_of_pop_results": js.get("number_of_pop_results", 100),
        "output_MUUG": js.get("output_MUUG", True),
        "output_haplotypes": js.get("output_haplotypes", False),
        "node_file": base_graph + gpath + js.get("node_csv_file"),
        "top_links_file": base_graph + gpath + js.get("top_links_csv_file"),
        "edges_file": base_graph + gpath + js.get("edges_csv_file"),
        "imputation_input_file": base_in + js.get("imputation_in_file"),
        "imputation_out_umug_freq_file": outp("imputation_out_umug_freq_filename"),
        "imputation_out_umug_pops_file": outp("imputation_out_umug_pops_filename"),
        "imputation_out_hap_freq_file": outp("imputation_out_hap_freq_filename"),
        "imputation_out_hap_pops_file": outp("imputation_out_hap_pops_filename"),
        "imputation_out_miss_file": outp("imputation_out_miss_filename"),
        "imputation_out_problem_file": outp("imputation_out_problem_filename"),
        "factor_missing_data": js.get("factor_missing_data", 0.01),
        "loci_map": loci_map,
        "matrix_planb": js.get("Plan_B_Matrix", DEFAULT_PLANB_MATRIX),
        "pops_count_file": base_graph + js.get("pops_count_file", ""),
        "use_pops_count_file": js.get("pops_count_file", False),
        "number_of_options_threshold": js.get("number_of_options_threshold", 100000),
        "max_haplotypes_number_in_phase": js.get("max_haplotypes_number_in_phase", 100),
        "bin_imputation_input_file": base_in + js.get("bin_imputation_in_file", "None"),
        "nodes_for_plan_A": js.get("Plan_A_Matrix", []),
        "save_mode": js.get("save_space_mode", False),
        "UNK_priors": js.get("UNK_priors", "MR"),
        "out_dir": out_dir,
    }
    cfg["full_loci"] = "".join(sorted({str(v) for v in loci_map.values()}))
    return cfg


# --------------------------------------------------------------------------
# graph store   (networkx_graph.py:14-321)
# --------------------------------------------------------------------------
def _row_starts(src_sorted, n_vertices):
    """Row-start array with the reference's exact construction, incl. its QUIRKs
    (networkx_graph.py:157-198): rows without out-edges copy the previous start and
    the closing sentinel is the VERTEX count, not the edge count."""
    uniq, first = np.unique(src_sorted, return_index=True)
    starts = []
    j = 0
    for i in range(n_vertices):
        if int(uniq[j]) == i:  # IndexError if trailing vertices have no edges (as in ref)
            starts.append(int(first[j]))
            j += 1
        else:
            starts.append(starts[-1] if starts else 0)
    starts.append(int(n_vertices))  # QUIRK: sentinel = len(Vertices)
    return np.array(starts, dtype=np.uint32)


class OGraph:
    """String-keyed graph store.  `attr[name] = (label, [freq per pop], id)`;
    connector pseudo-nodes of the plan-B graph map name -> int id."""

    def __init__(self, full_loci):
        self.full_loci = full_loci
        self.names = []  # plan-A vertices (Vertices)
        self.attr = {}
        self.w_names = []  # plan-B vertices incl. connectors (Whole_Vertices)
        self.w_attr = {}
        self._by_label = {}

    def load(self, nodes_csv, top_links_csv, edges_csv):
        id2name = {}
        with open(nodes_csv) as fh:  # networkx_graph.py:45-68
            rd = csv.reader(fh)
            next(rd)
            for row in rd:
                if not row:
                    continue
                freqs = [float(x) for x in row[3].split(";")]
                self.names.append(row[1])
                self.attr[row[1]] = (row[2], freqs, len(self.names) - 1)
                self.w_names.append(row[1])
                self.w_attr[row[1]] = (row[2], list(freqs), len(self.w_names) - 1)
                id2name[row[0]] = row[1]
        a_edges = []
        with open(top_links_csv) as fh:  # networkx_graph.py:71-88
            rd = csv.reader(fh)
            next(rd)
            for row in rd:
                if not row:
                    continue
                n1, n2 = id2name[row[0]], id2name[row[1]]
                if n1 in self.attr and n2 in self.attr:
                    if self.attr[n1][0] == self.full_loci:
                        a_edges.append((int(row[1]), int(row[0])))
                    else:
                        a_edges.append((int(row[0]), int(row[1])))
        w_edges = []
        with open(edges_csv) as fh:  # networkx_graph.py:91-130
            rd = csv.reader(fh)
            next(rd)
            for row in rd:
                if not row:
                    continue
                n1, n2 = id2name[row[0]], id2name[row[1]]
                l1, l2 = self.w_attr[n1][0], self.w_attr[n2][0]
                if len(l1) < len(l2):
                    child_id, child, parent_id, plabel = row[0], n1, row[1], l2
                    conn = plabel + child
                    if conn not in self.w_attr:
                        self.w_names.append(conn)
                        self.w_attr[conn] = len(self.w_names) - 1
                        w_edges.append((int(child_id), self.w_attr[conn]))
                    w_edges.append((self.w_attr[conn], int(parent_id)))
                else:
                    conn = l1 + n2
                    if conn not in self.w_attr:
                        self.w_names.append(conn)
                        self.w_attr[conn] = len(self.w_names) - 1
                    w_edges.append((int(row[1]), self.w_attr[conn]))
                    w_edges.append((self.w_attr[conn], int(row[0])))
        a = np.array(a_edges, dtype=np.uint32).reshape(-1, 2)
        w = np.array(w_edges, dtype=np.uint32).reshape(-1, 2)
        w = np.unique(w, axis=0)  # drop_duplicates + lexsort (networkx_graph.py:142-152)
        a = a[np.lexsort((a[:, 1], a[:, 0]))]
        self.nbr_start = _row_starts(a[:, 0], len(self.names))
        self.nbr = a[:, 1].copy()
        self.w_start = _row_starts(w[:, 0], len(self.w_names))
        self.w_nbr = w[:, 1].copy()
        return self

    # networkx_graph.py:215-236
    def haps_by_label(self, label):
        if label not in self._by_label:
            self._by_label[label] = [n for n, a in self.attr.items() if a[0] == label]
        return self._by_label[label]

    # networkx_graph.py:238-251
    def haps_with_probs_by_label(self, label):
        return {n: self.attr[n][1] for n in self.haps_by_label(label)}

    # networkx_graph.py:253-278
    def adjs_query(self, names):
        out = {}
        for nm in names:
            a = self.attr.get(nm)
            if a is None:
                continue
            if a[0] == self.full_loci:
                out[nm] = a[1]
            else:
                i = a[2]
                for e in range(int(self.nbr_start[i]), int(self.nbr_start[i + 1])):
                    adj = self.names[self.nbr[e]]
                    out[adj] = self.attr[adj][1]
        return out

    # networkx_graph.py:280-321
    def adjs_query_by_color(self, names, label_a, label_b):
        out = {}
        if label_a == label_b:
            for nm in names:
                if nm in self.w_attr:
                    out[nm] = self.w_attr[nm][1]
            return out
        for nm in names:
            if nm in self.w_attr:
                conn = label_b + nm
                if conn in self.w_attr:
                    ci = self.w_attr[conn]
                    for e in range(int(self.w_start[ci]), int(self.w_start[ci + 1])):
                        adj = self.w_names[self.w_nbr[e]]
                        out[adj] = self.w_attr[adj][1]
        return out


# --------------------------------------------------------------------------
# GL string handling
# --------------------------------------------------------------------------
def clean_gl(gl):
    """impute.py:105-118: delete every 'g' and 'L', drop loci that start/end with 'U'."""
    gl = gl.replace("g", "").replace("L", "")
    return "^".join(x for x in gl.split("^") if x.strip("U") == x)


def gl_to_genotype(gl):
    """impute.py:246-272.  Returns None for an unusable GL, else (side lists, n_loci)."""
    if gl == "" or gl == " ":
        return None
    side1, side2, blanks = [], [], 0
    parts = gl.split("^")
    n = len(parts)
    for p in parts:
        if p[0] == "+":  # IndexError on an empty locus entry, as in the reference
            p = p[1:]
        two = p.split("+")
        if len(two) == 1:
            if two == [""]:
                blanks += 1
                continue
            return None
        side1.append(two[0])
        side2.append(two[1])
    n -= blanks
    return [sorted(side1), sorted(side2)], n


def phases_of(gen, n_loci, b_phases=None):
    """impute.py:274-303.  b_phases (bin_imputation_in_file): a position may only take side 2's
    allele when its entry is 1."""
    out, seen = [], set()
    allowed = None if b_phases is None else [i for i, e in enumerate(b_phases) if e == 1]
    for i in range(2 ** (n_loci - 1)):
        pick = [(i >> k) & 1 for k in range(n_loci)]
        if allowed is not None:
            pick = [b if k in allowed else 0 for k, b in enumerate(pick)]
        h1 = [gen[pick[k]][k] for k in range(n_loci)]
        h2 = [gen[1 - pick[k]][k] for k in range(n_loci)]
        g1 = "~".join(h1) + "^" + "~".join(h2)
        g2 = "~".join(h2) + "^" + "~".join(h1)
        if g1 not in seen or g2 not in seen:
            seen.add(g1)
            seen.add(g2)
            out.append([h1, h2])
    return out


# --------------------------------------------------------------------------
# the engine
# --------------------------------------------------------------------------
class OracleImputer:
    def __init__(self, graph, cfg, count_by_prob=None):
        self.g = graph
        self.cfg = cfg
        self.pops = cfg["pops"]
        P = len(self.pops)
        self.loci_index = dict(cfg["loci_map"])  # impute.py:182-189 (ints as given)
        self.loci_order = list(cfg["loci_map"].keys())
        self.loci_label = {k: str(v) for k, v in cfg["loci_map"].items()}  # cypher_query.py:20-21
        self.full_loci = cfg["full_loci"]
        self.matrix = cfg["matrix_planb"]
        self.f_missing = cfg["factor_missing_data"]
        self.threshold = cfg["number_of_options_threshold"]
        self.top_n = cfg["max_haplotypes_number_in_phase"]
        self.save_mode = cfg["save_mode"]
        self.unk = cfg["UNK_priors"]
        if count_by_prob is None:  # impute.py:205-212
            self.count_by_prob = np.ones(P)
            if cfg["use_pops_count_file"]:
                with open(cfg["pops_count_file"]) as fh:
                    for i, line in enumerate(fh):
                        self.count_by_prob[i] = float(line.strip().split(",")[2])
        else:
            self.count_by_prob = count_by_prob
        self.prior = np.ones((P, P))
        self.plan = "a"
        self.log = []

    # ---- labels -----------------------------------------------------------
    def _label_of_indexes(self, idxs):  # cypher_plan_b.py:13-32
        return "".join(str(i) for i in sorted(idxs))

    def _label_of_name(self, name):  # cypher_plan_b.py:34-42
        return "".join(sorted(self.loci_label[a.split("*")[0]] for a in name.split("~")))

    # ---- prior matrix  (impute.py:1844-1924) -------------------------------
    def _priority_matrix(self, races1, races2, pr):
        P = len(self.pops)
        T = np.zeros((P, P))
        eye = np.identity(P)
        for r1 in races1:
            for r2 in races2:
                if r1 == "" and r2 == "":
                    continue
                t = np.zeros((P, P))
                if r1 == "" or r2 == "":
                    r = self.pops.index(r2) if r1 == "" else self.pops.index(r1)
                    for i in range(P):
                        t[r, i] = t[r, i] + pr["gamma"] * 2
                    t = t + t.transpose()
                    t[r, r] -= pr["gamma"] * 2
                else:
                    a, b = self.pops.index(r1), self.pops.index(r2)
                    for i in range(P):
                        t[a, i] = t[a, i] + pr["gamma"]
                        t[i, b] = t[i, b] + pr["gamma"]
                    t[a, b] -= pr["gamma"]
                    t[a, b] = t[a, b] + pr["alpha"]
                    if a != b:
                        t = t + t.transpose()
                        t[a, a] -= pr["gamma"]
                        t[b, b] -= pr["gamma"]
                    t[a, a] += pr["delta"]
                    if a != b:
                        t[b, b] += pr["delta"]
                t = pr["eta"] * np.ones((P, P)) + t + pr["beta"] * eye
                T += t
        total = 0
        for i in range(P):
            for j in range(P):
                T[i][j] = T[i][j] * self.count_by_prob[i] * self.count_by_prob[j]
                total += T[i][j]
        self.prior = T / total

    # ---- candidate enumeration (impute.py:914-989, cutils.pyx:4-51) ----------
    def _open(self, pmags, n_loci):
        out = []
        for h in pmags:
            sides = []
            label_nodes = None
            for k in range(2):
                splits = [tuple(a.split("/")) for a in h[k]]
                options = 1
                for i in range(n_loci):
                    options *= len(splits[i])
                if options < self.threshold:
                    cand = [list(h[k])]
                    for i in range(n_loci):  # cartesian, locus 0 most significant
                        if len(splits[i]) > 1:
                            nxt = []
                            for c in cand:
                                for alt in splits[i]:
                                    c2 = list(c)
                                    c2[i] = alt
                                    nxt.append(c2)
                            cand = nxt
                else:
                    if label_nodes is None or len(label_nodes) == 0:
                        present = []
                        for loc, lab in self.loci_label.items():
                            if any(a.split("*", 1)[0] == loc for a in h[k]):
                                present.append(lab)
                        label_nodes = self.g.haps_by_label("".join(sorted(present)))
                    allowed = {alt for s in splits for alt in s}
                    cand = []
                    for nm in label_nodes:
                        parts = nm.split("~")
                        cnt = 0
                        for p in parts:
                            if p not in allowed:
                                break
                            cnt += 1
                        if cnt == n_loci:
                            cand.append(parts)
                sides.append([cand])
            if sides[0][0] and sides[1][0]:
                out.append(sides)
        return out

    # ---- single-locus existence test (impute.py:1207-1222) ------------------
    def _lookup_colored(self, names, division):
        if len(names) == 0:
            return {}
        target = self._label_of_indexes(division)
        src = self._label_of_name(names[0])
        return self.g.adjs_query_by_color(names, src, target)

    def _alleles_exist(self, alleles):
        idx = self.loci_index[alleles[0].split("*")[0]]
        return self._lookup_colored(alleles, [idx])

    # impute.py:864-912
    def _reduce_valid(self, pmags, n_loci, planc=False):
        for h in pmags:
            for k in range(2):
                options = 1
                for i in range(n_loci):
                    options *= len(h[k][i].split("/"))
                if options >= self.threshold or planc:
                    for i, g in enumerate(h[k]):
                        found = self._alleles_exist(g.split("/"))
                        if found != {}:
                            h[k][i] = "/".join(found.keys())

    def _reduce_common(self, pmags, n_loci, keep=1, planc=False):
        for h in pmags:
            for k in range(2):
                options = 1
                for i in range(n_loci):
                    options *= len(h[k][i].split("/"))
                if options >= self.threshold or planc:
                    for i, g in enumerate(h[k]):
                        found = self._alleles_exist(g.split("/"))
                        if found != {}:
                            score = {}
                            for al, fr in found.items():
                                s = 0
                                for p, v in enumerate(fr):
                                    s += v * self.prior[p, p]
                                score[al] = s
                            best = sorted(score.items(), key=lambda kv: kv[1], reverse=True)[:keep]
                            h[k][i] = "/".join(al for al, _ in best)

    # ---- top-N flattening (impute.py:424-442) -------------------------------
    def _top(self, probs):
        flat = []
        for k in range(len(probs)):
            row = probs[k]
            for j in range(len(row)):
                if row[j] > 0:
                    flat.append((row[j] * self.prior[j][j], row[j], k, j))
        flat.sort(key=lambda t: t[0], reverse=True)
        return [(t[1], t[2], t[3]) for t in flat[: self.top_n]]

    # ---- pair scoring (impute.py:444-548, 550-658) ---------------------------
    def _score(self, haps1, haps2, top1, top2, eps, acc):
        prior = self.prior
        for p1, k1, j1 in top1:
            x = eps / p1
            x2 = x * 2
            for p2, k2, j2 in top2:
                if not (p2 >= x):
                    break
                w = prior[j1][j2]
                if not (w > 0):
                    continue
                a, b = haps1[k1], haps2[k2]
                if (a != b and w * p2 >= x) or (a == b and w * p2 >= x2):
                    ra, rb = self.pops[j1], self.pops[j2]
                    ident = "-".join(sorted([a + "," + ra, b + "," + rb]))
                    if ident in acc["seen"]:
                        continue
                    acc["seen"].add(ident)
                    prob = p1 * p2 * w
                    if a != b:
                        prob = prob * 2
                    if prob > acc["max"]:
                        acc["max"] = prob
                    if acc["muug"]:
                        geno = "^".join(
                            "+".join(sorted(z)) for z in zip(sorted(a.split("~")), sorted(b.split("~")))
                        )
                        acc["geno"][geno] = acc["geno"][geno] + prob if geno in acc["geno"] else prob
                        rr = sorted([ra, rb])
                        rr = rr[0] + "," + rr[1]
                        acc["pops"][rr] = acc["pops"][rr] + prob if rr in acc["pops"] else prob
                    else:
                        acc["pairs"].append([a, b])
                        acc["pair_pops"].append([ra, rb])
                        acc["pair_probs"].append(prob)

    @staticmethod
    def _acc(muug):
        return {"seen": set(), "max": 0, "muug": muug, "geno": {}, "pops": {},
                "pairs": [], "pair_pops": [], "pair_probs": []}

    @staticmethod
    def _result(acc):
        if acc["muug"]:
            return {"MaxProb": acc["max"], "Haps": acc["geno"], "Pops": acc["pops"]}
        return {"MaxProb": acc["max"], "Haps": acc["pairs"], "Probs": acc["pair_probs"],
                "Pops": acc["pair_pops"]}

    @staticmethod
    def _split(d):  # impute.py:353-360 / 1008-1013
        if not d:
            return "", ""
        return list(d.keys()), list(d.values())

    # ---- plan A (impute.py:660-754, 393-397) ---------------------------------
    def _plan_a(self, phases, eps, muug):
        acc = self._acc(muug)
        haps2, probs2 = [], []
        for ph in phases:
            haps1, probs1 = self._split(self.g.adjs_query(["~".join(c) for c in ph[0][0]]))
            if len(probs1) > 0:
                haps2, probs2 = self._split(self.g.adjs_query(["~".join(c) for c in ph[1][0]]))
            self._score(haps1, haps2, self._top(probs1), self._top(probs2), eps, acc)
        return self._result(acc)

    # ---- plan B helpers --------------------------------------------------------
    def _missing_loci(self, phases):  # impute.py:994-1006, 1193-1200
        first = phases[0][0][0][0]
        if len(first) >= len(self.full_loci):
            return []
        have = [self.loci_index[a.split("*")[0]] for a in first]
        out = []
        for loc in self.loci_order:
            v = self.loci_index[loc]
            if v not in have and v not in out:
                out.append(v)
        return out

    def _block_strings(self, cands, division, missing):  # impute.py:1015-1039
        out = []
        for hap in cands:
            s = ""
            for d in division:
                if d in missing:
                    continue
                place = d - sum(1 for m in missing if d > m)
                s = str(hap[place - 1]) if s == "" else s + "~" + str(hap[place - 1])
            if s != "":
                out.append(s)
        return out

    def _join(self, inner, outer, planc=False, keep=10):  # impute.py:1041-1069
        size = 1 if planc else len(self.pops)
        if self.save_mode:
            for d in (outer, inner):
                if len(d) > keep:
                    tot = sorted(((h, sum(v)) for h, v in d.items()), key=lambda kv: kv[1])
                    while len(d) > keep:
                        del d[tot[0][0]]
                        del tot[0]
        res = {}
        for k1, f1 in outer.items():
            for k2, f2 in inner.items():
                fr = [f1[i] * f2[i] * FACTOR_JOIN for i in range(size)]
                if max(fr) > 0:
                    res["~".join(sorted(k1.split("~") + k2.split("~")))] = fr
        return res

    def _row_freqs(self, row, side, missing):  # impute.py:1072-1123
        if row[0] == list(set(self.loci_index.values())):
            return self.g.adjs_query(["~".join(c) for c in side[0]])
        cur = self._lookup_colored(self._block_strings(side[0], row[0], missing), row[0])
        if cur != {}:
            for division in row[1:]:
                part = self._lookup_colored(self._block_strings(side[0], division, missing), division)
                if part == {}:
                    if all(d in missing for d in division):
                        part = self.g.haps_with_probs_by_label(self._label_of_indexes(division))
                    else:
                        cur = {}
                        break
                cur = self._join(part, cur)
        return cur

    def _row_freqs_absent(self, side, absent):  # impute.py:1125-1172
        keep = [x for x in set(self.loci_index.values()) if x not in list(set(absent))]
        n_abs = len(list(set(absent)))
        res = {}
        for hap in side[0]:
            inside, outside = "", []
            for al in hap:
                if self.loci_index[al.split("*")[0]] in absent:
                    outside.append(al)
                else:
                    inside += "~" + str(al)
            inside = inside[1:]
            outside = list(set(outside))
            if inside != "":
                for key, fr in self._lookup_colored([inside], keep).items():
                    parts = key.split("~")
                    parts = parts[: absent[0] - 1] + outside + parts[absent[0] - 1:]
                    res["~".join(sorted(parts))] = [x * (self.f_missing ** n_abs) for x in fr]
        return res

    def _absent_loci(self, phases, side):  # impute.py:1224-1241
        out = []
        width = len(phases[0][0][0][0])
        for t in range(width):
            al = []
            for ph in phases:
                for cands in ph[side]:
                    for hap in cands:
                        al.append(hap[t])
            al = list(set(al))
            if self._alleles_exist(al) == {}:
                out.append(self.loci_index[al[0].split("*")[0]])
        return out

    def _absent_loci_one(self, side):  # impute.py:1243-1258
        out = []
        for t in range(len(side[0][0])):
            al = list(set(hap[t] for hap in side[0]))
            if self._alleles_exist(al) == {}:
                out.append(self.loci_index[al[0].split("*")[0]])
        return out

    def _matrix_row(self, i):  # impute.py:1182-1191
        return self.matrix[i] if len(self.matrix) > i else []

    # ---- plan B (impute.py:1392-1570) ---------------------------------------------
    def _plan_b(self, phases, muug):
        eps = 0.0  # call site always arrives with 0.0 (impute.py:1703-1711)
        acc = self._acc(muug)
        haps2, probs2 = [], []
        absent1 = self._absent_loci(phases, 0)
        absent2 = self._absent_loci(phases, 1)
        for ph in phases:
            ph[0].append(10)
            ph[1].append(10)
        row_i = 0
        missing = []
        r1 = r2 = None

        # `hap_total == {}` (impute.py:1414,1492): in both output modes that dict gains an
        # entry exactly when a pair is accepted, i.e. when `seen` grows.
        def no_result():
            return len(acc["seen"]) == 0

        while no_result():
            row = self._matrix_row(row_i)
            if row == []:
                break
            missing = self._missing_loci(phases)
            for ph in phases:
                if absent1 == []:
                    idx = min(row_i, ph[0][1])
                    row = self._matrix_row(idx)
                    r1 = self._split(self._row_freqs(row, ph[0], missing))
                    if len(r1[0]):
                        ph[0][1] = idx
                else:
                    r1 = self._split(self._row_freqs_absent(ph[0], absent1))
                haps1, probs1 = r1
                if absent2 == []:
                    idx = min(row_i, ph[1][1])
                    row = self._matrix_row(idx)
                    r2 = self._split(self._row_freqs(row, ph[1], missing))
                    if len(r2[0]):
                        ph[1][1] = idx
                    haps2, probs2 = r2
                elif len(probs1) > 0:
                    r2 = self._split(self._row_freqs_absent(ph[1], absent2))
                    haps2, probs2 = r2
                self._score_b(haps1, haps2, probs1, probs2, eps, acc)
            row_i += 1

        cur = 0
        while no_result() and cur < 6:
            for ph in phases:
                i1, i2 = min(10, ph[0][1]), min(10, ph[1][1])
                if not (i1 == 10 and i2 == 10):
                    if i1 == 10 and len(ph[0][0]) > 0:
                        absent1 = self._absent_loci_one(ph[0])
                        r1 = self._split(self._row_freqs_absent(ph[0], absent1))
                        r2 = self._split(self._row_freqs(self._matrix_row(i2), ph[1], missing))
                    if i2 == 10 and len(ph[1][0]) > 0:
                        r1 = self._split(self._row_freqs(self._matrix_row(i1), ph[0], missing))
                        absent2 = self._absent_loci_one(ph[1])
                        r2 = self._split(self._row_freqs_absent(ph[1], absent2))
                    # QUIRK: r1/r2 may be stale from an earlier phase (impute.py:1496-1524)
                    self._score_b(r1[0], r2[0], r1[1], r2[1], eps, acc)
            cur += 1
        return self._result(acc)

    def _score_b(self, haps1, haps2, probs1, probs2, eps, acc):
        self._score(haps1, haps2, self._top(probs1), self._top(probs2), eps, acc)

    # ---- plan C (impute.py:1264-1389) ------------------------------------------------
    def _single_locus_product(self, cands, missing):
        res = {}
        for hap in cands:
            cur, absent = {}, []
            for al in hap:
                part = self._lookup_colored([al], [self.loci_index[al.split("*")[0]]])
                part = {k: [sum(v)] for k, v in part.items()}
                if part == {}:
                    absent.append(al)
                elif cur == {}:
                    cur = part
                else:
                    cur = self._join(part, cur, True)
                    if not cur:
                        break
            if absent:
                for key, fr in cur.items():
                    res["~".join(sorted(key.split("~") + absent))] = [
                        x * (self.f_missing ** len(absent)) for x in fr
                    ]
            else:
                res.update(cur)
        rest = {k: [sum(v)] for k, v in
                self.g.haps_with_probs_by_label(self._label_of_indexes(missing)).items()}
        if res:
            if rest:
                res = self._join(rest, res, True)
            else:
                for m in missing:
                    one = {k: [sum(v)] for k, v in
                           self.g.haps_with_probs_by_label(self._label_of_indexes([m])).items()}
                    if one:
                        res = self._join(one, res, True)
        return res

    def _plan_c(self, phases, muug):
        acc = self._acc(muug)
        haps2, probs2 = [], []
        missing = self._missing_loci(phases)
        for ph in phases:
            haps1, probs1 = self._split(self._single_locus_product(ph[0][0], missing))
            if len(probs1) > 0:
                haps2, probs2 = self._split(self._single_locus_product(ph[1][0], missing))
            self._score(haps1, haps2, self._top(probs1), self._top(probs2), 0, acc)
        res = self._result(acc)
        if muug:
            res["Pops"] = {"all_pops,all_pops": sum(acc["pops"].values())}
        else:
            res["Pops"] = [["all_pops", "all_pops"] for _ in acc["pair_pops"]]
        return res

    # ---- ladder + fallbacks (impute.py:1658-1724) -------------------------------------
    def _ladder(self, eps, phases, muug, planb):
        res = {"Haps": "NaN", "Probs": 0}
        last = False
        while eps > 0:
            eps /= 10
            if eps < 1.0e-9:
                eps = 0.0
            res = self._plan_a(phases, eps, muug)
            if len(res["Haps"]) > 0 and eps > 0:
                eps = res["MaxProb"] / 100000
                last = True
                break
        if last:
            res = self._plan_a(phases, eps, muug)
        P = len(self.pops)
        for level in range(2):
            if level == 1:
                self.prior = np.ones((P, P))  # QUIRK: unconditional reset (impute.py:1696-1700)
            if planb and len(res["Haps"]) == 0:
                self.plan = "b"
                res = self._plan_b(_deep(phases), muug)
        return res

    # ---- per subject (impute.py:1584-1656, 1940-1983) ------------------------------------
    def impute_one(self, gl, race1, race2, muug_out=None, haps_out=None, planb=None, b_phases=None):
        cfg = self.cfg
        muug_out = cfg["output_MUUG"] if muug_out is None else muug_out
        haps_out = cfg["output_haplotypes"] if haps_out is None else haps_out
        planb = cfg["planb"] if planb is None else planb
        P = len(self.pops)
        cleaned = clean_gl(gl)
        self.prior = np.ones((P, P)) if self.unk == "MR" else np.identity(P)
        if race1 or race2:
            r1 = race1.split(";")
            r2 = race2.split(";")
            known = False
            for lst in (r1, r2):
                for i, r in enumerate(lst):
                    if r not in self.pops:
                        lst[i] = ""
                    else:
                        known = True
            if known:
                self._priority_matrix(r1, r2, cfg["priority"])
        if not gl:
            return None, None
        parsed = gl_to_genotype(cleaned)
        if parsed is None:
            return None, None
        gen, n_loci = parsed
        pmags = phases_of(gen, n_loci, b_phases)
        if pmags == []:
            return None, None
        res_m = {"MaxProb": 0, "Haps": {}, "Pops": {}}
        res_h = {"Haps": "Nan", "Probs": 0, "Pops": {}}
        phases = self._open(pmags, n_loci)
        if not phases:
            self._reduce_valid(pmags, n_loci)
            phases = self._open(pmags, n_loci)
        if not phases:
            self._reduce_common(pmags, n_loci, keep=10)
            phases = self._open(pmags, n_loci)
        if phases:
            eps = cfg["epsilon"]
            if muug_out:
                saved = np.array(self.prior, order="K", copy=True)
                res_m = self._ladder(eps, phases, True, planb)
                if planb and len(res_m["Haps"]) == 0:
                    self.plan = "c"
                    self._reduce_common(pmags, n_loci, 1, True)
                    phases = self._open(pmags, n_loci)
                    res_m = self._plan_c(phases, True)
                self.prior = saved
            if haps_out:
                res_h = self._ladder(eps, phases, False, planb)
                if planb and len(res_h["Haps"]) == 0 and not getattr(self, "em", False):  # impute.py:1649
                    self._reduce_common(pmags, n_loci, 1, True)
                    phases = self._open(pmags, n_loci)
                    res_h = self._plan_c(phases, False)
        return res_m, res_h

    # ---- file driver (impute.py:1985-2155) and writers (impute.py:24-99) ---------------------
    def impute_lines(self, lines, em_mr=False, em=False):
        """Returns dict of the six output texts keyed 'umug','umug_pops','pmug','pmug_pops',
        'miss','problem'.  self.log collects the per-subject stdout lines."""
        cfg = self.cfg
        out = {k: [] for k in ("umug", "umug_pops", "pmug", "pmug_pops", "miss", "problem")}
        self.em = em  # impute_file(em=True), impute.py:1985
        n_res, n_pop = cfg["number_of_results"], cfg["number_of_pop_results"]
        sid = None
        f_bin = None
        if os.path.isfile(cfg["bin_imputation_input_file"]):  # impute.py:2001-2005
            with open(cfg["bin_imputation_input_file"]) as fh:
                f_bin = json.load(fh)
        for i, line in enumerate(lines):
            try:
                line = line.rstrip()
                parts = line.split(",") if "," in line else line.split("%")
                sid, gl = parts[0], parts[1]
                race1 = race2 = None
                if len(parts) > 2:
                    race1, race2 = parts[2], parts[3]
                self.plan = "a"
                b_phases = [1] * (len(self.full_loci) - 1)
                if f_bin is not None:
                    b_phases = f_bin[sid]  # KeyError -> the reference's bare except
                res_m, res_h = self.impute_one(gl, race1, race2, b_phases=b_phases)
                if res_m is None:
                    out["problem"].append(f"{i},{sid}\n")
                    continue
                if (len(res_h["Haps"]) == 0 or res_h["Haps"] == "NaN") and len(res_m["Haps"]) == 0:
                    out["miss"].append(f"{i},{sid}\n")
                if cfg["output_haplotypes"]:
                    haps, probs, pops = res_h["Haps"], res_h["Probs"], res_h["Pops"]
                    self.log.append(f"{i} Subject: {sid} {len(haps)} haplotypes")
                    if em_mr:
                        _write_pairs_with_races(sid, haps, pops, probs, n_res, out["pmug"])
                        _write_merged(sid, pops, probs, 1, out["pmug_pops"], ",")
                    else:
                        _write_merged(sid, haps, probs, n_res, out["pmug"], "+")
                        _write_merged(sid, pops, probs, n_pop, out["pmug_pops"], ",")
                if cfg["output_MUUG"]:
                    m_haps, m_pops = res_m["Haps"], res_m["Pops"]  # both are read before the print (impute.py:2103-2104)
                    self.log.append(f"{i} Subject: {sid} {len(m_haps)} haplotypes")
                    _write_sorted(sid, m_haps, n_res, out["umug"])
                    _write_sorted(sid, m_pops, n_pop, out["umug_pops"])
            except Exception:  # reference: bare except (impute.py:2141-2144)
                self.log.append(f"{i} Subject: {sid} - Exception")
                out["problem"].append(str(line) + "\n")
        return {k: "".join(v) for k, v in out.items()}

    def impute_file(self, em_mr=False):
        cfg = self.cfg
        with open(cfg["imputation_input_file"]) as fh:
            texts = self.impute_lines(fh, em_mr=em_mr)
        names = {"umug": "imputation_out_umug_freq_file", "umug_pops": "imputation_out_umug_pops_file",
                 "pmug": "imputation_out_hap_freq_file", "pmug_pops": "imputation_out_hap_pops_file",
                 "miss": "imputation_out_miss_file", "problem": "imputation_out_problem_file"}
        for k, key in names.items():
            if k.startswith("umug") and not cfg["output_MUUG"]:
                continue
            if k.startswith("pmug") and not cfg["output_haplotypes"]:
                continue
            with open(cfg[key], "w") as fh:
                fh.write(texts[k])
        return texts


def _deep(x):  # cutils.pyx:53-65
    return [_deep(e) if isinstance(e, list) else e for e in x]


def _write_sorted(sid, table, limit, sink):  # impute.py:61-76
    rows = sorted(table.items(), key=lambda kv: kv[1], reverse=True)
    for k in range(min(limit, len(rows))):
        sink.append(f"{sid},{rows[k][0]},{rows[k][1]},{k}\n")


def _write_merged(sid, pairs, probs, limit, sink, sign):  # impute.py:24-58
    table = OrderedDict()
    for k in range(len(pairs)):
        fwd = pairs[k][0] + sign + pairs[k][1]
        if fwd in table:
            table[fwd] = probs[k] + table[fwd]
        else:
            rev = pairs[k][1] + sign + pairs[k][0]
            if rev in table:
                table[rev] = probs[k] + table[rev]
            else:
                table[fwd] = probs[k]
    rows = sorted(table.items(), key=lambda kv: kv[1], reverse=True)
    for k in range(min(limit, len(rows))):
        sink.append(f"{sid},{rows[k][0]},{rows[k][1]},{k}\n")


def _write_pairs_with_races(sid, haps, pops, probs, limit, sink):  # impute.py:79-99
    rows = [(probs[i], haps[i][0] + ";" + pops[i][0] + "," + haps[i][1] + ";" + pops[i][1])
            for i in range(len(probs))]
    rows.sort(key=lambda r: r[0], reverse=True)
    for k in range(min(limit, len(rows))):
        sink.append(f"{sid},{rows[k][1]},{rows[k][0]},{k}\n")


def build_from_conf(conf_path, base_graph="", base_in=""):
    with open(conf_path) as fh:
        cfg = config_from_json(json.load(fh), base_graph, base_in)
    g = OGraph(cfg["full_loci"]).load(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
    return OracleImputer(g, cfg), cfg
