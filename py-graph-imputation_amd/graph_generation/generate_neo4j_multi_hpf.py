"""
generate_graph(config_file, em_pop, em, use_default_path): hpf.csv -> nodes.csv, edges.csv,
top_links.csv, info_node.csv.

Drop-in for the reference's graph_generation/generate_neo4j_multi_hpf.py:209-486 (same conf
keys, same CSV dialect and row order; nodes.csv / edges.csv / info_node.csv are byte
identical, top_links.csv has the same rows -- the reference emits each node's links in
Python-set order, here they are ascending by full-haplotype id; the loader sorts them).

Layout facts the hot-path loader relies on (reference lines in brackets):
  * labels: the full label first, then every smaller locus subset, larger subsets first,
    `itertools.combinations` order  [:101-110]
  * node id == creation order == row order in nodes.csv  [:341-358, 364-415, 419-430]
  * a partial node's frequency vector is the running left-to-right sum, in full-haplotype
    order, of the full haplotypes that project onto it  [:398-405]
  * edges.csv has one row per (child node, full haplotype, added locus); CP = full-haplotype
    freq / child freq per population (0 when the child freq is 0)  [:82-97, 434-455]

Not supported (raises): `Plan_A_Matrix` reduced-label graphs (reference :113-192) -- SURVEY §8f.4.
"""

import json
import os
import pathlib
from itertools import combinations


def _label_list(full):
    labels = [full]
    for r in range(len(full) - 1, 0, -1):
        labels.extend("".join(c) for c in combinations(full, r))
    return labels


def _canonical(hap, locus_index, n):
    """alleles placed by locus index, trailing 'g' removed, '0' where a locus is absent [:59-68]."""
    slots = ["0"] * n
    for a in hap.split("~"):
        if a[-1] == "g":
            a = a[:-1]
        slots[locus_index[a.split("*")[0]] - 1] = a
    return slots


def _num(x):
    return str(x)


def generator_inputs(config_file, em_pop=None, em=False, use_default_path=False):
    """-> (hpf file, populations, per-population cut-offs, loci_map) exactly as generate_graph derives them"""
    base = os.path.dirname(os.path.realpath(__file__)) + "/" if use_default_path else ""
    with open(config_file) as fh:
        conf = json.load(fh)
    if conf.get("Plan_A_Matrix", []):
        raise NotImplementedError("Plan_A_Matrix (reduced-label graphs) is not supported by this build")
    pops = em_pop if em_pop else conf.get("populations")
    trim = conf.get("freq_trim_threshold")
    counts_file = pathlib.Path(base + conf.get("pops_count_file", ""))
    cutoff = {}
    if em or not counts_file.is_file():
        for p in pops:
            cutoff[p] = trim
    else:
        with open(counts_file) as fh:
            for line in fh:
                p, cnt, _ratio = line.strip().split(",")
                cutoff[p] = trim / float(cnt)
    return base + conf.get("freq_file"), pops, [cutoff[p] for p in pops], dict(conf.get("loci_map"))


def generate_graph(config_file="../conf/minimal-configuration-script.json", em_pop=None, em=False,
                   use_default_path=False, quiet=False, python_twin=False):
    base = ""
    if use_default_path:
        base = os.path.dirname(os.path.realpath(__file__)) + "/"
    with open(config_file) as fh:
        conf = json.load(fh)
    if conf.get("Plan_A_Matrix", []):
        raise NotImplementedError("Plan_A_Matrix (reduced-label graphs) is not supported by this build")

    csvdir = conf.get("graph_files_path")
    pathlib.Path(csvdir).mkdir(parents=True, exist_ok=True)
    if csvdir[-1] != "/":
        csvdir += "/"
    pops = em_pop if em_pop else conf.get("populations")
    trim = conf.get("freq_trim_threshold")
    freq_file = base + conf.get("freq_file")
    counts_file = pathlib.Path(base + conf.get("pops_count_file", ""))

    cutoff = {}
    if em or not counts_file.is_file():
        for p in pops:
            cutoff[p] = trim
    else:
        with open(counts_file) as fh:
            for line in fh:
                p, cnt, _ratio = line.strip().split(",")
                cutoff[p] = trim / float(cnt)

    if not quiet:
        bar = "*" * 100
        print(bar)
        print("Performing graph generation based on following configuration:")
        print("\tPopulation: {}".format(pops))
        print("\tFreq File: {}".format(freq_file))
        print("\tFreq Trim Threshold: {}".format(trim))
        print(bar)

    locus_index = dict(conf.get("loci_map"))
    if not python_twin:
        # the library's C++ generator (grim_graphgen_csv): same four files, byte for byte
        from grim import _native as nat

        nat.graphgen_csv(freq_file, pops, [cutoff[p] for p in pops], locus_index, csvdir + conf.get("node_csv_file"),
                         csvdir + conf.get("edges_csv_file"), csvdir + conf.get("top_links_csv_file"),
                         csvdir + conf.get("info_node_csv_file"))
        return
    full = "".join(sorted({str(v) for v in locus_index.values()}))
    nloc = len(full)
    labels = _label_list(full)

    # ---- full haplotypes --------------------------------------------------------------
    by_pop = {}  # "POP-name" -> freq
    order = {}  # name -> allele slots, first-seen order
    with open(freq_file) as fh:
        for line in fh:
            if not line:
                continue
            hap, pop, freq = line.split(",")
            if hap == "hap":
                continue
            freq = float(freq)
            if freq == 0.0 or freq < cutoff[pop]:
                continue
            slots = _canonical(hap, locus_index, nloc)
            name = "~".join(slots)
            order[name] = slots
            by_pop[pop + "-" + name] = freq

    full_names = list(order.keys())
    full_slots = [order[n] for n in full_names]
    full_freq = [[by_pop.get(p + "-" + n, 0) for p in pops] for n in full_names]
    n_full = len(full_names)
    next_id = n_full

    # ---- partial labels ----------------------------------------------------------------
    # per label: name -> [id, freq vector, parents[(label, name, full index)], top links[full ids]]
    part = {}
    for lab in labels[1:]:
        idx = [full.index(c) for c in lab]
        rest = [i for i in range(nloc) if i not in idx]
        grown = []
        for i in rest:
            bigger = sorted(idx + [i])
            grown.append(("".join(full[j] for j in bigger), bigger))
        nodes = {}
        for f in range(n_full):
            slots = full_slots[f]
            name = "~".join(slots[i] for i in idx)
            node = nodes.get(name)
            if node is None:
                node = [next_id, [0] * len(pops), [], []]
                next_id += 1
                nodes[name] = node
            for plab, pidx in grown:
                node[2].append((plab, "~".join(slots[j] for j in pidx), f))
            node[3].append(f)
            fv = full_freq[f]
            node[1] = [a + b for a, b in zip(node[1], fv)]
        part[lab] = nodes

    def node_id(lab, name):
        return full_names_index[name] if lab == full else part[lab][name][0]

    full_names_index = {n: i for i, n in enumerate(full_names)}

    # ---- nodes.csv ----------------------------------------------------------------------
    with open(csvdir + conf.get("node_csv_file"), "w", newline="") as fh:
        fh.write("haplotypeId:ID(HAPLOTYPE),name,loci:LABEL,frequency:DOUBLE[]\r\n")
        for i, n in enumerate(full_names):
            fh.write("%d,%s,%s,%s\r\n" % (i, n, full, ";".join(map(_num, full_freq[i]))))
        for lab in labels[1:]:
            for name, node in part[lab].items():
                fh.write("%d,%s,%s,%s\r\n" % (node[0], name, lab, ";".join(map(_num, node[1]))))

    # ---- edges.csv ----------------------------------------------------------------------
    with open(csvdir + conf.get("edges_csv_file"), "w", newline="") as fh:
        fh.write(":START_ID(HAPLOTYPE),:END_ID(HAPLOTYPE),CP:DOUBLE[],:TYPE\r\n")
        for lab in labels[1:]:
            for name, node in part[lab].items():
                child_freq = node[1]
                for plab, pname, f in node[2]:
                    cp = [0 if c == 0 else a / c for a, c in zip(full_freq[f], child_freq)]
                    fh.write("%d,%d,%s,CP\r\n" % (node[0], node_id(plab, pname), ";".join(map(_num, cp))))

    # ---- top_links.csv --------------------------------------------------------------------
    with open(csvdir + conf.get("top_links_csv_file"), "w", newline="") as fh:
        fh.write(":START_ID(HAPLOTYPE),:END_ID(HAPLOTYPE),:TYPE\r\n")
        for lab in labels[1:]:
            for name, node in part[lab].items():
                for f in node[3]:
                    fh.write("%d,%d,TOP\r\n" % (node[0], f))

    # ---- info_node.csv ----------------------------------------------------------------------
    with open(csvdir + conf.get("info_node_csv_file"), "w", newline="") as fh:
        fh.write("INFO_NODE_ID:ID(INFO_NODE),populations:STRING[],INFO_NODE:LABEL\r\n")
        fh.write("1,%s,INFO_NODE\r\n" % ";".join(pops))
