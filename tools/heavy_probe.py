#!/usr/bin/env python3
"""time the heaviest subjects of the mixed workload one by one (use with the GRIM_STAMPS build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import harness, synth
sys.path.insert(0, harness.ROOT)
import numpy as np
os.environ["GRIM_QUIET"] = "1"
os.environ.setdefault("GRIM_TIMING", "1")  # per-kernel events
import grim.imputation.impute as I
rows = synth.read_freqs(synth.CAU_FREQS)
gen = synth.SubjectGen(rows, 5, pops=["CAU"])
lines = gen.mixed(10000)
conf = harness.base_conf(["CAU"])
keep = {}
orig = I.Imputation._run_arrays
def spy(self, subj, tokens, priors, params):
    res, rws = orig(self, subj, tokens, priors, params); keep["res"] = res; return res, rws
I.Imputation._run_arrays = spy
harness.run_product("cau", conf, lines, tag="hv")
res = keep["res"]
npair = res['n_pairs'].astype(np.int64)
edges = [0, 1, 65, 129, 257, 513, 1025, 2049, 4097, 16385, 1 << 30]
print('n_pairs histogram:', [(edges[i], int(((npair >= edges[i]) & (npair < edges[i + 1])).sum())) for i in range(len(edges) - 1)])
print('plans:', {chr(p): int((res['plan'] == p).sum()) for p in np.unique(res['plan'])})
for plan in (ord('a'), ord('b'), ord('c')):
    idx = np.nonzero(res["plan"] == plan)[0]
    top = idx[np.argsort(-res["n_pairs"][idx].astype(np.int64))[:3]]
    for i in top:
        print("---- plan %s subject %d nU=%d nG=%d: %s" % (chr(plan), i, res["n_pairs"][i], res["n_genotypes"][i], lines[i][:110]), flush=True)
        got, log, imp = harness.run_product("cau", conf, [lines[i]], tag="hv1")
        print("     kernel ms total %.3f A %.3f B %.3f" % (imp.last_stats["kernel_ms"], imp.last_stats["kernel_a_ms"], imp.last_stats["kernel_b_ms"]), flush=True)
