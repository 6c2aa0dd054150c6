#!/usr/bin/env python3
"""
BASELINE config 5 stand-in: a synthetic WMDA-scale MULTI-POPULATION graph (the real WMDA set needs a network fetch,
SURVEY 8c) and high-ambiguity subjects on it.

Graph: N_HAPS five-locus haplotypes built by recombining CAU class-I (A~B~C) and class-II (DQB1~DRB1) blocks (plus
class-I blocks with single alleles swapped) with Zipf-like frequencies; three populations, each a random 65 % of the
haplotypes with its own log-normal frequency scaling (the recipe of the committed 4-population set).
300 000 haplotypes (287 000 after the populations' subsets) give ~1.1 M nodes, 8.6 M top links and 21.5 M edge rows -- the size quoted at
graph_generation/generate_neo4j_multi_hpf.py:30-33 -- built by the PRODUCT generator and loader (C++).

Subjects: every locus typed, 8 or 16 alternatives per locus and side (seeded), races mixed over the populations;
number_of_options_threshold = 1e6, max_haplotypes_number_in_phase = 100, UNK_priors = MR.  With 8 alternatives a side has
32 768 candidates (cartesian / intersection opening); with 16 it has 2^20 >= the threshold (label scan).

    ensure(n_haps) -> work directory     conf() -> conf dict     subjects(n, seed) -> input lines
    python tools/wmda_scale.py [n_subjects=64]        (GPU box: builds the graph, runs the subjects, checks an oracle
                                                       sample on the small variant, prints kernel times)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import harness
import synth

sys.path.insert(0, harness.ROOT)

N_HAPS = int(os.environ.get("GRIM_WMDA_HAPS", "300000"))
POPS = ["WMA", "WMB", "WMC"]


def name_of(n_haps=None):
    return "wmda%d" % (n_haps or N_HAPS)


def make_haplotypes(n_haps, seed=7):
    """-> list of (haplotype string in the freqs files' locus order, "1", freq)"""
    rows = synth.read_freqs(synth.CAU_FREQS)
    rng = np.random.default_rng(seed)
    haps = [synth.hap_alleles(h) for h, _, _ in rows]
    c1, c2 = {}, {}
    for h, (_, _, f) in zip(haps, rows):
        c1[(h["A"], h["B"], h["C"])] = c1.get((h["A"], h["B"], h["C"]), 0) + f
        c2[(h["DQB1"], h["DRB1"])] = c2.get((h["DQB1"], h["DRB1"]), 0) + f
    k1, k2 = list(c1), list(c2)
    p1 = np.array([c1[k] for k in k1])
    p1 /= p1.sum()
    p2 = np.array([c2[k] for k in k2])
    p2 /= p2.sum()
    extra = set()  # more class-I blocks: single alleles swapped between existing blocks
    while len(k1) + len(extra) < max(len(k1), (n_haps * 3) // max(1, len(k2))):
        a, b = k1[int(rng.integers(len(k1)))], k1[int(rng.integers(len(k1)))]
        j = int(rng.integers(3))
        nb = tuple(b[i] if i == j else a[i] for i in range(3))
        if nb not in c1:
            extra.add(nb)
    k1x = k1 + sorted(extra)
    p1x = np.concatenate([p1, np.full(len(extra), p1.min() * 0.1)])
    p1x /= p1x.sum()
    seen = {}
    while len(seen) < n_haps:
        i = rng.choice(len(k1x), size=n_haps, p=p1x)
        j = rng.choice(len(k2), size=n_haps, p=p2)
        for a, b in zip(i, j):
            if len(seen) >= n_haps:
                break
            seen.setdefault((int(a), int(b)), None)
    keys = list(seen)
    z = 1.0 / np.arange(1, len(keys) + 1) ** 0.9
    z = z / z.sum()
    rng.shuffle(z)
    out = []
    for (a, b), f in zip(keys, z):
        A, B, C = k1x[a]
        Q, R = k2[b]
        out.append(("%s~%s~%s~%s~%s" % (A, C, B, R, Q), "1", float("%.6g" % f)))
    return out


def ensure(n_haps=None):
    """work directory of the graph (built on first use: freqs files, hpf.csv, the four graph CSVs)"""
    n_haps = n_haps or N_HAPS
    name = name_of(n_haps)
    harness.POPS[name] = list(POPS)
    work = os.path.join(harness.WORK, name)
    if os.path.exists(os.path.join(work, "output", "csv", "info_node.csv")):
        return work
    from graph_generation.generate_hpf import produce_hpf
    from graph_generation.generate_neo4j_multi_hpf import generate_graph

    base = make_haplotypes(n_haps)
    rng = np.random.default_rng(11)
    os.makedirs(os.path.join(work, "data", "freqs"), exist_ok=True)
    os.makedirs(os.path.join(work, "data", "subjects"), exist_ok=True)
    for p in POPS:
        synth.write_freqs(os.path.join(work, "data", "freqs", p + ".freqs.gz"), synth.synth_population(base, rng, drop=0.35))
    c = conf()
    with open(os.path.join(work, "graph_conf.json"), "w") as fh:
        json.dump(c, fh)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        produce_hpf("graph_conf.json", quiet=True)
        generate_graph("graph_conf.json", quiet=True)
    finally:
        os.chdir(cwd)
    return work


def conf():
    c = harness.base_conf(POPS)
    c["freq_trim_threshold"] = 1e-12
    c["UNK_priors"] = "MR"
    c["number_of_options_threshold"] = 1000000
    c["max_haplotypes_number_in_phase"] = 100
    return c


_rows_cache = {}


def union_rows(n_haps=None):
    """the graph's haplotypes with a frequency (population WMA's, else the first population that has it)"""
    n_haps = n_haps or N_HAPS
    if n_haps not in _rows_cache:
        work = ensure(n_haps)
        seen = {}
        for p in POPS:
            for h, c, f in synth.read_freqs(os.path.join(work, "data", "freqs", p + ".freqs.gz")):
                seen.setdefault(h, (h, c, f))
        _rows_cache[n_haps] = list(seen.values())
    return _rows_cache[n_haps]


def subjects(n, seed=5, n_haps=None, widths=(8, 16)):
    """n high-ambiguity subjects: alternating widths, races drawn over the populations"""
    gen = synth.SubjectGen(union_rows(n_haps), seed, pops=POPS)
    out = []
    for i in range(n):
        w = widths[i % len(widths)]
        line = gen.high_ambiguity(1, width=w, prefix="H%d_" % i)[0]
        sid, gl = line.split(",")[:2]
        r1, r2 = gen.races()
        out.append("%s,%s,%s,%s" % (sid, gl, r1, r2))
    return out


def main():
    n_subj = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    import __graft_entry__ as ge

    ge.build()
    os.environ["GRIM_QUIET"] = "1"
    os.environ.setdefault("GRIM_TIMING", "1")
    t0 = time.time()
    ensure()
    name = name_of()
    print("graph csv ready in %.1f s" % (time.time() - t0), flush=True)
    from grim import _native as nat

    lines = subjects(n_subj)
    for rep in range(2):
        t1 = time.time()
        got, log, imp = harness.run_product(name, conf(), lines, tag="w5", on_unsupported="skip", quiet=True)
        dt = time.time() - t1
    st = imp.last_stats
    g = imp.netGraph
    a = g.arrays
    print("graph: %d nodes, %d top links, %d plan-B edges, %d populations, host arrays %.1f MB, device %.1f MB" % (
        a["n_nodes"], len(a["a_nbr"]), len(a["b_nbr"]), a["n_pops"], g.host_bytes() / 1e6,
        g.device(nat.default_context()).device_bytes() / 1e6), flush=True)
    ctr = st["counters"]
    algo = 16 * ctr[0] + 4 * ctr[1] + 8 * a["n_pops"] * ctr[2]
    print("high ambiguity n=%d  wall %.2f s  device thread %.4f s  kernels %.3f ms (A %.3f, B/C %.3f)  probes %d nbr %d freq %d  "
          "graph-side algorithmic %.1f MB -> %.1f GB/s  unsupported %d  reruns %d" % (
              len(lines), dt, st["device_s"], st["kernel_ms"], st["kernel_a_ms"], st["kernel_b_ms"], ctr[0], ctr[1], ctr[2],
              algo / 1e6, algo / max(st["kernel_ms"], 1e-9) / 1e6, len(imp.unsupported), st.get("reruns", 0)), flush=True)


if __name__ == "__main__":
    main()
