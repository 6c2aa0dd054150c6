#!/usr/bin/env python3
"""file -> file timing of the drop-in API: Imputation.impute_file = the library's streaming pipeline (tokenizer threads,
H2D, kernels, D2H, formatter threads, ordered pwrite).   python tools/e2e_time.py [n] [full|mixed] [graph]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import harness, synth
sys.path.insert(0, harness.ROOT)
import __graft_entry__ as ge

def main():
    ge.build()
    os.environ["GRIM_QUIET"] = "1"
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    kind = sys.argv[2] if len(sys.argv) > 2 else "full"
    gname = sys.argv[3] if len(sys.argv) > 3 else "cau"
    rows = synth.read_freqs(synth.CAU_FREQS)
    gen = synth.SubjectGen(rows, 0 if gname == "cau" else 3, pops=harness.POPS[gname])
    lines = gen.full_fast(n) if kind == "full" else gen.mixed(n)
    conf = harness.base_conf(harness.POPS[gname])
    if gname != "cau":
        conf["UNK_priors"] = "MR"
    harness.run_product(gname, conf, lines[:100], tag="e2e_warm")      # graph build + upload + first touch
    got, log, imp = harness.run_product(gname, conf, lines, tag="e2e", quiet=True)   # writes the input file, warms the page cache
    cfg = dict(imp.config)
    work = harness.ensure_graph(gname)
    os.chdir(work)
    cfg["imputation_input_file"] = os.path.join(work, "data", "subjects", "e2e.csv")
    imp.quiet = True
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        imp.impute_file(cfg)                                                 # input file -> six output files
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, dict(imp.last_stats))
    dt, st = best
    print("   stream:", json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items() if k not in ("host_s",)}))
    print("   host cpu-seconds:", {k: round(v, 4) for k, v in st.get("host_s", {}).items()})
    print("e2e %s/%s n=%d: %.3f s  -> %.0f subjects/s  (device thread busy %.4f s, kernels %.3f ms)" % (
        gname, kind, n, dt, n / dt, st["device_s"], st["kernel_ms"]))

if __name__ == "__main__":
    main()
