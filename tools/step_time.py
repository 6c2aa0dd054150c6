#!/usr/bin/env python3
"""wall time of one synchronous batch run (config 2, 10k subjects): hipGraph replay vs direct launches vs timing mode"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import harness, synth
sys.path.insert(0, harness.ROOT)
os.environ["GRIM_QUIET"] = "1"
import numpy as np
from grim import _native as nat
from grim.imputation.impute import Imputation
from grim.imputation.networkx_graph import Graph
from grim.run_impute_def import load_config
work = harness.ensure_graph("cau"); os.chdir(work)
cfg, _ = load_config("graph_conf.json")
g = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
imp = Imputation(g, cfg)
rows = synth.read_freqs(synth.CAU_FREQS)
text = ("\n".join(synth.SubjectGen(rows, 0).full(10000)) + "\n").encode()
parsed = nat.Parsed(g.adict, text, cfg["planb"])
priors = np.stack([imp._prior_matrix(r1, r2, cfg["priority"]) for r1, r2 in parsed.races()])
params = imp._params(cfg, cfg["planb"], False)
ctx = nat.default_context(0)
for mode in ("graph", "direct", "timing"):
    os.environ["GRIM_GRAPH"] = "1" if mode == "graph" else "0"
    batch = nat.DeviceBatch(ctx, g.device(ctx), params, parsed.subjects(), parsed.tokens(), priors)
    batch.set_timing(mode == "timing")
    for _ in range(20):
        batch.run()
    t = time.perf_counter()
    for _ in range(200):
        batch.run()
    dt = (time.perf_counter() - t) / 200
    print("%-7s %.1f us per run   small kernel %.2f us" % (mode, dt * 1e6, batch.kernel_ms(3) * 1e3))
    batch.close()
