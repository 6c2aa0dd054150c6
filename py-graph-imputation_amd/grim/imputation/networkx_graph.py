"""
Graph store for the MI355X engine: nodes.csv / top_links.csv / edges.csv -> integer-encoded CSR
arrays that live in HBM.

Same public surface as the reference's grim/imputation/networkx_graph.py (`Graph(config)`,
`build_graph(nodes, top_links, edges)`), different representation:

  reference (networkx_graph.py)                          here
  ---------------------------------------------------   -----------------------------------------
  Vertices_attributes[name] -> (label, freqs, id) :53   64-bit key = sum((allele_id+1) << 12*slot),
                                                         open-addressing index built by the library
  Edges / Neighbors_start (plan A)             :136-207  a_nbr / a_start   (uint32 CSR)
  Whole_* with "label+name" connector nodes    :91-130   b_conn[node][added slot] -> connector,
                                                         b_start / b_nbr   (uint32 CSR)
  haps_by_label (O(V) scan, memoised)          :215-236  lab_start / lab_nodes (nodes grouped by label)

The reference's row-start construction is reproduced exactly, including its two quirks
(:157-198): vertices without out-edges copy the previous row start, and the closing sentinel is
the vertex count instead of the edge count (so the highest-numbered vertex loses its neighbours).
"""

import csv

import numpy as np

from .. import _native as nat


def _row_starts(src_sorted, n_vertices):
    """Vectorised restatement of networkx_graph.py:157-198 (see module docstring)."""
    uniq, first = np.unique(src_sorted, return_index=True)
    starts = np.zeros(n_vertices + 1, dtype=np.int64)
    if len(uniq) == 0 or int(uniq[-1]) != n_vertices - 1:
        # the reference indexes past the end of `unique_values` here and dies with IndexError
        raise IndexError("graph: the highest-numbered vertex has no out-edges (reference cannot load this graph either)")
    has = np.zeros(n_vertices, dtype=bool)
    has[uniq] = True
    val = np.zeros(n_vertices, dtype=np.int64)
    val[uniq] = first
    # forward fill: a vertex without out-edges takes the previous vertex's start (0 at the front)
    idx = np.where(has, np.arange(n_vertices), -1)
    idx = np.maximum.accumulate(idx)
    starts[:n_vertices] = np.where(idx >= 0, val[np.maximum(idx, 0)], 0)
    starts[n_vertices] = n_vertices  # QUIRK: sentinel = len(Vertices)
    return starts.astype(np.uint32)


class Graph(object):
    def __init__(self, config):
        self.full_loci = config["full_loci"]
        if config.get("nodes_for_plan_A"):
            raise NotImplementedError("Plan_A_Matrix (reduced-label graphs) is not supported by this build")
        loci_map = config["loci_map"]
        # locus name -> slot; slot s <-> label character full_loci[s]
        self.locus_slot = {}
        for name, val in loci_map.items():
            self.locus_slot[name] = self.full_loci.index(str(val))
        self.slot_locus = [None] * len(self.full_loci)
        for name, s in self.locus_slot.items():
            self.slot_locus[s] = name
        if len(self.full_loci) > nat.MAXL:
            raise NotImplementedError("more than %d loci" % nat.MAXL)
        # Candidate names are built from the subject's alleles in sorted STRING order (impute.py:271) while graph names are
        # in loci_map index order (generate_neo4j_multi_hpf.py:59-68).  Where the two orders differ for a set of loci -- a
        # loci_map that is not alphabetical, such as the reference's own default A 1, B 3, C 2 -- a name built from a
        # subject's alleles over that set never equals a graph name, and the reference's look-ups miss; so do the device's
        # (bit m of label_order_bad: label mask m is such a set; csrc/grim_dev.h graph_lookup_subject).
        self.label_order_bad = 0
        for m in range(1, 1 << len(self.full_loci)):
            slots = [s for s in range(len(self.full_loci)) if (m >> s) & 1]
            names = [self.slot_locus[s] for s in slots]
            if None in names:
                continue
            if sorted(names, key=lambda n: n + "*") != names:
                self.label_order_bad |= 1 << m
        # one allele dictionary (C++ side of the library) shared by loader, tokenizer and formatter
        self.adict = nat.AlleleDict(self.slot_locus)
        self.arrays = None
        self._dev = {}

    # ------------------------------------------------------------------------------------------
    def allele_id(self, slot, allele, create=True):
        return self.adict.intern(slot, allele)

    def key_alleles(self, key):
        out = []
        for s in range(len(self.full_loci)):
            a = (int(key) >> (nat.ABITS * s)) & 0xFFF
            out.append(self.adict.name(s, a - 1) if a else None)
        return out

    def key_to_name(self, key):
        return "~".join(x for x in self.key_alleles(key) if x)

    # ------------------------------------------------------------------------------------------
    def build_graph(self, nodesFile, edgesFile, allEdgesFile):
        """nodesFile = nodes.csv, edgesFile = top_links.csv, allEdgesFile = edges.csv
        (argument names as in networkx_graph.py:42).  Parsed and indexed by the library's C++ loader
        (grim_hostgraph_load_csv); `_build_graph_python` is the same thing in numpy, kept as the
        cross-check in tests/."""
        self._check_labels(nodesFile)
        self.arrays = nat.load_graph_csv(self.adict, self.full_loci, nodesFile, edgesFile, allEdgesFile)
        self.arrays["label_order_bad"] = self.label_order_bad
        self.n_graph_alleles = [self.adict.count(s) for s in range(len(self.full_loci))]
        self._dev = {}
        return self

    def _check_labels(self, nodesFile, rows=2000):
        """The label column of nodes.csv is written with the loci_map the graph was GENERATED with
        (generate_neo4j_multi_hpf.py:419-430); this build derives labels from the alleles under the loci_map of the
        configuration at hand.  When the two maps differ (a graph made with A 1, B 2, C 3 read under the reference's default
        A 1, B 3, C 2, say) the reference keeps the file's labels next to its own index arithmetic and answers something in
        between; that mixture is not reproduced -- refuse instead of answering differently."""
        with open(nodesFile) as fh:
            rd = csv.reader(fh)
            next(rd, None)
            for k, row in enumerate(rd):
                if k >= rows:
                    break
                if len(row) < 3:
                    continue
                try:
                    want = "".join(sorted(self.full_loci[self.locus_slot[a.split("*")[0]]] for a in row[1].split("~")))
                except KeyError:
                    continue
                if want != row[2]:
                    raise NotImplementedError(
                        "nodes.csv labels haplotype %r as %r, the configuration's loci_map makes it %r: the graph was generated with "
                        "another loci_map than the one in use" % (row[1], row[2], want))

    def build_graph_from_hpf(self, hpf_file, populations, cutoffs, loci_map, csv_paths=None):
        """hpf.csv -> this graph, generator and loader back to back inside the library (grim_hostgraph_from_hpf):
        what graph_freqs() + build_graph() produce, without the four CSVs in between.  csv_paths = optional
        (nodes, edges, top_links, info_node) to write them as well."""
        self.arrays = nat.graph_from_hpf(self.adict, self.full_loci, hpf_file, populations, cutoffs, loci_map, csv_paths)
        self.arrays["label_order_bad"] = self.label_order_bad
        self.n_graph_alleles = [self.adict.count(s) for s in range(len(self.full_loci))]
        self._dev = {}
        return self

    def _build_graph_python(self, nodesFile, edgesFile, allEdgesFile):
        nl = len(self.full_loci)
        ids, keys, masks, freqs = [], [], [], []
        id_to_row = {}
        with open(nodesFile) as fh:
            rd = csv.reader(fh)
            next(rd)
            for row in rd:
                if not row:
                    continue
                key = 0
                mask = 0
                for al in row[1].split("~"):
                    s = self.locus_slot[al.split("*")[0]]
                    key |= (self.allele_id(s, al) + 1) << (nat.ABITS * s)
                lab_mask = 0
                for ch in row[2]:
                    lab_mask |= 1 << self.full_loci.index(ch)
                for s in range(nl):
                    if (key >> (nat.ABITS * s)) & 0xFFF:
                        mask |= 1 << s
                if mask != lab_mask:
                    raise ValueError("node %s: alleles do not match its label %s" % (row[1], row[2]))
                id_to_row[row[0]] = len(keys)
                keys.append(key)
                masks.append(mask)
                freqs.append([float(x) for x in row[3].split(";")])
        V = len(keys)
        node_key = np.array(keys, dtype=np.uint64)
        node_mask = np.array(masks, dtype=np.uint8)
        freq = np.ascontiguousarray(np.array(freqs, dtype=np.float64))
        P = freq.shape[1]
        full_mask = (1 << nl) - 1
        nbits = np.array([bin(m).count("1") for m in range(1 << nl)], dtype=np.int64)

        def read_pairs(path):
            a, b = [], []
            with open(path) as fh:
                rd = csv.reader(fh)
                next(rd)
                for row in rd:
                    if row:
                        a.append(id_to_row[row[0]])
                        b.append(id_to_row[row[1]])
            return np.array(a, dtype=np.int64), np.array(b, dtype=np.int64)

        # plan A: partial -> full (networkx_graph.py:71-88)
        n1, n2 = read_pairs(edgesFile)
        flip = node_mask[n1] == full_mask
        src = np.where(flip, n2, n1)
        dst = np.where(flip, n1, n2)
        order = np.lexsort((dst, src))
        src, dst = src[order], dst[order]
        a_start = _row_starts(src, V)
        a_nbr = dst.astype(np.uint32)

        # plan B: child -> connector(parent label, child) -> parents (networkx_graph.py:91-130)
        n1, n2 = read_pairs(allEdgesFile)
        first_is_child = nbits[node_mask[n1]] < nbits[node_mask[n2]]
        child = np.where(first_is_child, n1, n2)
        parent = np.where(first_is_child, n2, n1)
        pmask = node_mask[parent].astype(np.int64)
        conn_key = child * 64 + pmask
        # connector ids in order of first appearance, after the V real vertices
        uniq, first_pos, inv = np.unique(conn_key, return_index=True, return_inverse=True)
        rank = np.empty(len(uniq), dtype=np.int64)
        rank[np.argsort(first_pos, kind="stable")] = np.arange(len(uniq))
        conn = rank[inv]
        ncon = len(uniq)
        w_src = np.concatenate([child, V + conn])
        w_dst = np.concatenate([V + conn, parent])
        w = np.unique(np.stack([w_src, w_dst], axis=1), axis=0)  # drop_duplicates + lexsort
        w_start = _row_starts(w[:, 0], V + ncon)
        b_start = np.ascontiguousarray(w_start[V:])
        b_nbr = w[:, 1].astype(np.uint32)
        added = pmask & ~node_mask[child].astype(np.int64)
        if np.any(nbits[added] != 1):
            raise NotImplementedError("edges.csv holds a parent that is not exactly one locus larger than its child")
        slot_added = np.log2(added).astype(np.int64)
        b_conn = np.full(V * nat.MAXL, 0xFFFFFFFF, dtype=np.uint32)
        b_conn[child * nat.MAXL + slot_added] = conn.astype(np.uint32)

        lab_order = np.argsort(node_mask, kind="stable").astype(np.uint32)
        counts = np.bincount(node_mask, minlength=1 << nat.MAXL)[: 1 << nat.MAXL]
        lab_start = np.zeros((1 << nat.MAXL) + 1, dtype=np.uint32)
        lab_start[1:] = np.cumsum(counts)

        self.arrays = {
            "n_nodes": V, "n_pops": P, "n_loci": nl, "full_mask": full_mask,
            "node_key": node_key, "node_mask": node_mask, "freq": freq,
            "a_start": a_start, "a_nbr": a_nbr,
            "b_conn": b_conn, "b_start": b_start, "b_nbr": b_nbr,
            "lab_start": lab_start, "lab_nodes": lab_order, "label_order_bad": self.label_order_bad,
        }
        self.n_graph_alleles = [self.adict.count(s) for s in range(nl)]
        self._dev = {}
        return self

    # ---- the reference's look-ups over the loaded arrays (networkx_graph.py:215-321) ----------------
    # The kernels do these on the device with integer keys; the methods below answer the same questions on the
    # host, from the same arrays, for callers that use the Graph object directly (names in, names out, frequencies
    # as lists of floats, dict insertion order as in the reference).
    def _label_mask(self, label):
        m = 0
        for ch in str(label):
            if ch not in self.full_loci:
                return None
            m |= 1 << self.full_loci.index(ch)
        return m

    def _mask_label(self, mask):
        return "".join(self.full_loci[s] for s in range(len(self.full_loci)) if (int(mask) >> s) & 1)

    def _node_index(self):
        idx = getattr(self, "_name_index", None)
        if idx is None:
            idx = {}
            keys = self.arrays["node_key"]
            for i in range(len(keys)):  # a repeated name keeps the later row, like dict assignment (networkx_graph.py:53)
                idx[int(keys[i])] = i
            self._name_index = idx
        return idx

    def node_of_name(self, name):
        """node id of a haplotype NAME as the graph files spell it (alleles joined by '~' in loci_map index order), or None"""
        key = 0
        for al in str(name).split("~"):
            slot = self.locus_slot.get(al.split("*")[0])
            if slot is None:
                return None
            i = nat.host_lib().grim_dict_find(self.adict.h, slot, al.encode())
            if i < 0 or i >= self.n_graph_alleles[slot] or (key >> (nat.ABITS * slot)) & 0xFFF:
                return None
            key |= (i + 1) << (nat.ABITS * slot)
        node = self._node_index().get(key)
        if node is None or self.key_to_name(key) != name:  # the reference's dicts are keyed by the exact string
            return None
        return node

    def node_name(self, node):
        return self.key_to_name(int(self.arrays["node_key"][node]))

    def _freqs(self, node):
        return [float(x) for x in self.arrays["freq"][node]]

    def haps_by_label(self, label):
        """networkx_graph.py:215-236: names of every node of `label` (e.g. "125"), in node order."""
        m = self._label_mask(label)
        if m is None:
            return []
        a, b = int(self.arrays["lab_start"][m]), int(self.arrays["lab_start"][m + 1])
        return [self.node_name(int(i)) for i in self.arrays["lab_nodes"][a:b]]

    def haps_with_probs_by_label(self, label):
        """networkx_graph.py:238-251"""
        m = self._label_mask(label)
        if m is None:
            return {}
        a, b = int(self.arrays["lab_start"][m]), int(self.arrays["lab_start"][m + 1])
        return {self.node_name(int(i)): self._freqs(int(i)) for i in self.arrays["lab_nodes"][a:b]}

    def adjs_query(self, alleleList):
        """networkx_graph.py:253-278: full haplotypes (and their frequencies) each name stands for -- itself when it is a
        full haplotype, else its top-link neighbours (row-start quirks of the loader included)."""
        out = {}
        full = self.arrays["full_mask"]
        a_start, a_nbr = self.arrays["a_start"], self.arrays["a_nbr"]
        for name in alleleList:
            i = self.node_of_name(name)
            if i is None:
                continue
            if int(self.arrays["node_mask"][i]) == full:
                out[name] = self._freqs(i)
            else:
                for e in range(int(a_start[i]), int(a_start[i + 1])):  # empty when start[i+1] <= start[i] (the quirk)
                    j = int(a_nbr[e])
                    out[self.node_name(j)] = self._freqs(j)
        return out

    def node_probs(self, nodes, label):
        """networkx_graph.py:309-321"""
        out = {}
        for name in nodes:
            i = self.node_of_name(name)
            if i is not None:
                out[name] = self._freqs(i)
        return out

    def adjs_query_by_color(self, alleleList, labelA, labelB):
        """networkx_graph.py:280-307: nodes of label `labelB` that contain each name (one locus more: the plan-B graph's
        connector of (parent label, child))."""
        if labelA == labelB:
            return self.node_probs(alleleList, labelA)
        out = {}
        mb = self._label_mask(labelB)
        b_conn, b_start, b_nbr = self.arrays["b_conn"], self.arrays["b_start"], self.arrays["b_nbr"]
        for name in alleleList:
            i = self.node_of_name(name)
            if i is None or mb is None:
                continue
            added = mb & ~int(self.arrays["node_mask"][i])
            if added == 0 or added & (added - 1) or (mb & int(self.arrays["node_mask"][i])) != int(self.arrays["node_mask"][i]):
                continue  # no connector "labelB + name": the parent label must be the child's plus exactly one locus
            c = int(b_conn[i * nat.MAXL + added.bit_length() - 1])
            if c == 0xFFFFFFFF:
                continue
            for e in range(int(b_start[c]), int(b_start[c + 1])):
                j = int(b_nbr[e])
                out[self.node_name(j)] = self._freqs(j)
        return out

    # ------------------------------------------------------------------------------------------
    def device(self, ctx):
        """Upload once per context; the handle keeps the HBM copy alive."""
        if self.arrays is None:
            raise RuntimeError("Graph.build_graph has not been called")
        dg = self._dev.get(id(ctx))
        if dg is None:
            dg = nat.DeviceGraph(ctx, self.arrays)
            self._dev[id(ctx)] = dg
        return dg

    def host_bytes(self):
        return sum(v.nbytes for v in self.arrays.values() if isinstance(v, np.ndarray))
