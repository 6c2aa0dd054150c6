#!/usr/bin/env python3
"""time every plan-B subject of the mixed workload alone; print the slowest"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import harness, synth
sys.path.insert(0, harness.ROOT)
import numpy as np
os.environ["GRIM_QUIET"] = "1"
import grim.imputation.impute as I
rows = synth.read_freqs(synth.CAU_FREQS)
lines = synth.SubjectGen(rows, 5, pops=["CAU"]).mixed(10000)
conf = harness.base_conf(["CAU"])
keep = {}
orig = I.Imputation._run_arrays
def spy(self, subj, tokens, priors, params):
    res, rws = orig(self, subj, tokens, priors, params); keep["res"] = res; return res, rws
I.Imputation._run_arrays = spy
harness.run_product("cau", conf, lines, tag="pb")
res = keep["res"]
idx = np.nonzero(res["plan"] != ord('a'))[0]
out = []
for i in idx:
    got, log, imp = harness.run_product("cau", conf, [lines[i]], tag="pb1")
    out.append((imp.last_stats["kernel_b_ms"], int(i), int(res["n_pairs"][i]), chr(res["plan"][i])))
out.sort(reverse=True)
print("n plan B/C subjects:", len(out), " sum ms:", sum(o[0] for o in out))
for ms, i, nu, plan in out[:8]:
    print("%.3f ms  plan %s nU=%d  %s" % (ms, plan, nu, lines[i][:150]))
