#!/usr/bin/env python3
"""hpf.csv -> CSVs -> arrays: C++ (library) against the Python twins, at a chosen scale.
    python tools/graph_build_time.py [n_haplotypes=100000] [--twin]"""
import json, os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import harness, synth, wmda_scale
sys.path.insert(0, harness.ROOT)
import numpy as np
n_haps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100000
twin = "--twin" in sys.argv
work = os.path.join(harness.WORK, "gbt%d" % n_haps)
os.makedirs(os.path.join(work, "data", "freqs"), exist_ok=True)
conf = harness.base_conf(["WMD"]); conf["freq_trim_threshold"] = 1e-12
cpath = os.path.join(work, "graph_conf.json")
if not os.path.exists(os.path.join(work, "output", "hpf.csv")):
    synth.write_freqs(os.path.join(work, "data", "freqs", "WMD.freqs.gz"), wmda_scale.make_freqs(n_haps))
    json.dump(conf, open(cpath, "w"))
    from graph_generation.generate_hpf import produce_hpf
    os.chdir(work); produce_hpf("graph_conf.json", quiet=True)
os.chdir(work)
from graph_generation.generate_neo4j_multi_hpf import generate_graph
from grim.run_impute_def import load_config
from grim.imputation.networkx_graph import Graph
def md5s():
    return {f: hashlib.md5(open("output/csv/" + f, "rb").read()).hexdigest() for f in ("nodes.csv", "edges.csv", "top_links.csv")}
t = time.time(); generate_graph("graph_conf.json", quiet=True); tg = time.time() - t; m1 = md5s()
cfg, _ = load_config("graph_conf.json")
t = time.time(); g1 = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"]); tl = time.time() - t
rows = {f: sum(1 for _ in open("output/csv/" + f)) - 1 for f in ("nodes.csv", "edges.csv", "top_links.csv")}
print("haplotypes %d  rows %s" % (n_haps, rows))
print("C++   : generate_graph %.2f s, build_graph %.2f s" % (tg, tl))
if twin:
    t = time.time(); generate_graph("graph_conf.json", quiet=True, python_twin=True); tg2 = time.time() - t; m2 = md5s()
    t = time.time(); g2 = Graph(cfg)._build_graph_python(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"]); tl2 = time.time() - t
    same = all(np.array_equal(g1.arrays[k], g2.arrays[k]) if isinstance(g2.arrays[k], np.ndarray) else g1.arrays[k] == g2.arrays[k] for k in g2.arrays)
    print("Python: generate_graph %.2f s, build_graph %.2f s   CSVs identical: %s   arrays identical: %s" % (tg2, tl2, m1 == m2, same))
