#!/usr/bin/env python3
"""stage timers of a -DGRIM_STAMPS build on the config-4 workload (GRIM_LIB=<stamps .so>):  python tools/stamps4.py [n=100000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GRIM_QUIET"] = "1"
import harness, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pops = harness.POPS["pop4"]
lines = synth.SubjectGen(synth.read_freqs(synth.CAU_FREQS), 3, pops=pops).mixed(n)
conf = harness.base_conf(pops)
conf["UNK_priors"] = "MR"
os.environ["GRIM_CHUNK_LINES"] = str(n)
for rep in range(2):
    got, log, imp = harness.run_product("pop4", conf, lines, tag="s4", on_unsupported="skip", quiet=True)
print(imp.last_stats)
