#!/usr/bin/env python3
"""file -> file timing of the drop-in API (host tokeniser + upload + kernels + download + formatter)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import harness, synth
sys.path.insert(0, harness.ROOT)
import __graft_entry__ as ge

def main():
    ge.build()
    os.environ["GRIM_QUIET"] = "1"
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    kind = sys.argv[2] if len(sys.argv) > 2 else "full"
    rows = synth.read_freqs(synth.CAU_FREQS)
    gen = synth.SubjectGen(rows, 0)
    lines = gen.full(n) if kind == "full" else gen.mixed(n)
    conf = harness.base_conf(["CAU"])
    harness.run_product("cau", conf, lines[:100], tag="e2e_warm")      # graph build + upload + first touch
    from grim.imputation import impute as I
    got, log, imp = harness.run_product("cau", conf, lines, tag="e2e")   # writes the input file, warms the page cache
    cfg = dict(imp.config)
    work = harness.ensure_graph("cau")
    os.chdir(work)
    cfg["imputation_input_file"] = os.path.join(work, "data", "subjects", "e2e.csv")
    imp.quiet = True
    t0 = time.perf_counter()
    imp.impute_file(cfg)                                                 # input file -> six output files
    dt = time.perf_counter() - t0
    st = imp.last_stats
    print("   host phases:", {k: round(v, 4) for k, v in st.get("host_s", {}).items()}, "upload", round(st.get("upload_s", 0), 4), "download", round(st.get("download_s", 0), 4))
    print("e2e %s n=%d: %.3f s  -> %.0f subjects/s  (device run %.4f s, kernels %.3f ms)" % (kind, n, dt, n / dt, st["run_s"], st["kernel_ms"]))

if __name__ == "__main__":
    main()
