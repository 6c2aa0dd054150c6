#!/usr/bin/env python3
"""One worker of bench.py's CPU baseline: the oracle (oracle/grim_oracle.py) over a slice of a subject file, the way
scripts/runfile_mp.py of the reference runs one process per chunk.  Never touches the GPU.  Prints the seconds of the
imputation itself ("s": input lines in memory -> output texts in memory) and of the graph load before it ("load_s").
    python tools/cpu_baseline_worker.py <graph name> <conf json> <subject file> <first line> <last line>"""
import json, os, sys, time
os.environ["HIP_VISIBLE_DEVICES"] = ""
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import harness

gname, conf_path, subj_path, lo, hi = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
if gname.startswith("wmda"):  # the synthetic WMDA-scale graph of tools/wmda_scale.py (built there on first use)
    import wmda_scale
    wmda_scale.ensure(int(gname[4:]))
conf = json.load(open(conf_path))
lines = [l.rstrip("\n") for l in open(subj_path)][lo:hi]
t0 = time.perf_counter()
harness.run_oracle(gname, conf, lines[:1], tag="cpu_base_%d" % lo)   # loads the graph (cached in the process)
t = time.perf_counter()
harness.run_oracle(gname, conf, lines, tag="cpu_base_%d" % lo)
print(json.dumps({"n": len(lines), "s": time.perf_counter() - t, "load_s": t - t0}))
