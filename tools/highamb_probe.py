#!/usr/bin/env python3
"""kernel time for config-5 style subjects (8 alternatives per locus and side, threshold 1e6) on the CAU graph"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import harness, synth
sys.path.insert(0, harness.ROOT)
os.environ["GRIM_QUIET"] = "1"
os.environ.setdefault("GRIM_TIMING", "1")  # per-kernel events
rows = synth.read_freqs(synth.CAU_FREQS)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
width = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lines = synth.SubjectGen(rows, 50).high_ambiguity(n, width=width)
conf = dict(harness.base_conf(["CAU"]), number_of_options_threshold=1000000)
for _ in range(2):
    got, log, imp = harness.run_product("cau", conf, lines, tag="ha")
st = imp.last_stats
print("width=%d" % width, end=" "); print("n=%d kernels %.3f ms (A %.3f B %.3f) probes %d -> %.2f G probes/s" % (n, st["kernel_ms"], st["kernel_a_ms"], st["kernel_b_ms"], st["counters"][0], st["counters"][0] / st["kernel_ms"] / 1e6))
