#!/usr/bin/env python3
"""how many subjects of a label-scan-heavy workload (low number_of_options_threshold) hit the branches
that are not on the device (reasons 1 and 3)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import harness, synth
sys.path.insert(0, harness.ROOT)
os.environ["GRIM_QUIET"] = "1"
rows = synth.read_freqs(synth.CAU_FREQS)
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lines = synth.SubjectGen(rows, 61).mixed(600, amb=0.7, miss=0.1, recomb=0.5)
conf = dict(harness.base_conf(["CAU"]), number_of_options_threshold=thr)
got, log, imp = harness.run_product("cau", conf, lines, tag="unsup", on_unsupported="skip")
c = collections.Counter(r for _, _, r in imp.unsupported)
print("threshold", thr, "subjects", len(lines), "unsupported by reason:", dict(c))
exp, elog = harness.run_oracle("cau", conf, lines, tag="unsup_orc")
skipped = [sid for _, sid, _ in imp.unsupported]
exp2 = harness.drop_subjects(exp, skipped)
print("supported part identical to oracle:", all(exp2[k] == got[k] for k in exp2))
for i, sid, r in imp.unsupported[:6]:
    print(r, lines[i][:160])
