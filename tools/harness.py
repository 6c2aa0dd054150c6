"""
Shared test / bench / smoke plumbing: build graphs from the committed frequency files with the
PRODUCT generator, run a scenario through the product (HIP) path and through the oracle.
Nothing here reads /root/reference.
"""

import contextlib
import io
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "py-graph-imputation_amd")
GOLD = os.path.join(ROOT, "tests", "golden")
WORK = os.environ.get("GRIM_WORK", os.path.join(ROOT, "tests", "_work"))
for p in (PKG, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

POPS = {"cau": ["CAU"], "pop4": ["CAU", "AFA", "HIS", "API"],
        # nine populations: CAU plus eight synthesised on the fly (seed 9); oracle-checked only, no golden files
        "pop9": ["CAU", "AFA", "HIS", "API", "NAM", "MENA", "SAS", "EAS", "OCE"]}
# graphs whose configuration differs from the minimal conf's: a loci_map whose index order is not the alphabetical locus order
# (the reference's own default map, B = 3 and C = 2)
GRAPH_OVERRIDES = {"cau_bc": {"loci_map": {"A": 1, "B": 3, "C": 2, "DQB1": 4, "DRB1": 5}},
                   "pop4_bc": {"loci_map": {"A": 1, "B": 3, "C": 2, "DQB1": 4, "DRB1": 5}}}
POPS["cau_bc"] = POPS["cau"]
POPS["pop4_bc"] = POPS["pop4"]
OUT_FILES = {"umug": "don.umug", "umug_pops": "don.umug.pops", "pmug": "don.pmug", "pmug_pops": "don.pmug.pops",
             "miss": "don.miss", "problem": "don.problem"}


def base_conf(pops):
    with open(os.path.join(GOLD, "cau_min", "conf.json")) as fh:
        conf = json.load(fh)
    conf["populations"] = list(pops)
    return conf


def ensure_graph(name):
    """-> work directory holding data/freqs, output/hpf.csv, output/csv/*.csv for graph `name`."""
    from graph_generation.generate_hpf import produce_hpf
    from graph_generation.generate_neo4j_multi_hpf import generate_graph

    work = os.path.join(WORK, name)
    marker = os.path.join(work, "output", "csv", "info_node.csv")
    if os.path.exists(marker):
        return work
    os.makedirs(os.path.join(work, "data", "freqs"), exist_ok=True)
    os.makedirs(os.path.join(work, "data", "subjects"), exist_ok=True)
    extra_rng = None
    for p in POPS[name]:
        src = os.path.join(GOLD, "data", "freqs", p + ".freqs.gz")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(work, "data", "freqs"))
        else:  # a population without a committed file: synthesised from CAU like the committed ones
            import numpy as np
            import synth

            if extra_rng is None:
                extra_rng = np.random.default_rng(9)
            synth.write_freqs(os.path.join(work, "data", "freqs", p + ".freqs.gz"),
                              synth.synth_population(synth.read_freqs(synth.CAU_FREQS), extra_rng, drop=0.5))
    conf = base_conf(POPS[name])
    conf.update(GRAPH_OVERRIDES.get(name, {}))
    with open(os.path.join(work, "graph_conf.json"), "w") as fh:
        json.dump(conf, fh)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        produce_hpf("graph_conf.json", quiet=True)
        generate_graph("graph_conf.json", quiet=True)
    finally:
        os.chdir(cwd)
    return work


def _write_inputs(work, conf, lines, tag):
    conf = dict(conf)
    if conf.get("bin_imputation_in_file") and "_bin_src" in conf:
        shutil.copy(conf.pop("_bin_src"), os.path.join(work, conf["bin_imputation_in_file"]))
    conf.pop("_bin_src", None)
    conf.pop("_em", None)
    conf["imputation_in_file"] = "data/subjects/%s.csv" % tag
    conf["imputation_out_path"] = "output_" + tag
    with open(os.path.join(work, conf["imputation_in_file"]), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    for f in OUT_FILES.values():  # a run that switches an output off must not see an older run's file
        stale = os.path.join(work, "output_" + tag, f)
        if os.path.exists(stale):
            os.remove(stale)
    cpath = os.path.join(work, "conf_%s.json" % tag)
    with open(cpath, "w") as fh:
        json.dump(conf, fh)
    return conf, cpath


def read_outputs(work, tag):
    out = {}
    for k, f in OUT_FILES.items():
        p = os.path.join(work, "output_" + tag, f)
        out[k] = open(p).read() if os.path.exists(p) else ""
    return out


_graph_cache = {}


def run_product(graph_name, conf, lines, tag="prod", em_mr=False, on_unsupported="raise", quiet=False):
    """Run the HIP path through the reference-shaped API.  -> (texts, log lines, Imputation)."""
    from grim.imputation.impute import Imputation
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    work = ensure_graph(graph_name)
    em = bool(conf.get("_em"))  # impute_file(em=True)
    conf, cpath = _write_inputs(work, conf, lines, tag)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg, out_dir = load_config(cpath)
        g = _graph_cache.get(graph_name)
        if g is None:
            g = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
            _graph_cache[graph_name] = g
        imp = Imputation(g, cfg)
        imp.on_unsupported = on_unsupported
        imp.quiet = quiet
        os.makedirs(out_dir, exist_ok=True)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            imp.impute_file(cfg, em_mr=em_mr, em=em)
    finally:
        os.chdir(cwd)
    log = [l for l in buf.getvalue().splitlines() if "Subject:" in l]
    return read_outputs(work, tag), log, imp


_ograph_cache = {}


def run_oracle(graph_name, conf, lines, tag="orc", em_mr=False):
    """Run the CPU oracle.  -> (texts, log lines)."""
    import grim_oracle as go

    work = ensure_graph(graph_name)
    em = bool(conf.get("_em"))
    conf, cpath = _write_inputs(work, conf, lines, tag)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg = go.config_from_json(conf)
        g = _ograph_cache.get(graph_name)
        if g is None:
            g = go.OGraph(cfg["full_loci"]).load(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
            _ograph_cache[graph_name] = g
        imp = go.OracleImputer(g, cfg)
        texts = imp.impute_lines(lines, em_mr=em_mr, em=em)
    finally:
        os.chdir(cwd)
    return texts, list(imp.log)


def golden(scenario):
    """-> (graph name, conf dict, input lines, expected texts, expected log, hap_pop_pair)"""
    d = os.path.join(GOLD, scenario)
    meta = json.load(open(os.path.join(d, "meta.json")))
    conf = json.load(open(os.path.join(d, "conf.json")))
    lines = [l.rstrip("\n") for l in open(os.path.join(d, "input.csv"))]
    exp = {}
    for k, f in OUT_FILES.items():
        p = os.path.join(d, f)
        exp[k] = open(p).read() if os.path.exists(p) else ""
    log = [l for l in open(os.path.join(d, "log.txt")).read().splitlines() if "Subject:" in l]
    if meta.get("em"):
        conf["_em"] = True
    if os.path.exists(os.path.join(d, "bin.json")):
        conf["_bin_src"] = os.path.join(d, "bin.json")
    return meta["graph"], conf, lines, exp, log, meta["hap_pop_pair"]


def scenarios():
    return sorted(s for s in os.listdir(GOLD) if os.path.exists(os.path.join(GOLD, s, "conf.json")))


def drop_subjects(texts, ids):
    """remove every output line of the given subject ids (used while a path is unsupported)"""
    ids = set(ids)
    out = {}
    for k, t in texts.items():
        keep = []
        for line in t.splitlines(keepends=True):
            first = line.split(",")[0]
            second = line.split(",")[1].strip() if k in ("miss", "problem") and "," in line else None
            if first in ids or (second in ids):
                continue
            keep.append(line)
        out[k] = "".join(keep)
    return out
