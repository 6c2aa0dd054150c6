#!/usr/bin/env python3
"""The kernel-only loop of bench.py ALONE (leg B: tokenised subjects resident in HBM, results left in HBM), for a clean
rocprofv3 pass: no stream path beside it, so the per-kernel averages of the trace are the undisturbed launch durations that
bench.py's `roofline.avg_launch_ms` (HIP events) has to agree with.

    rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/kernel_only.py --workload config2 --runs 300

Prints one JSON line: the HIP-event means of the same runs, the algorithmic bytes per launch (SURVEY 8d formula) and the
roofline fraction computed from them."""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
os.environ["GRIM_QUIET"] = "1"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--subjects", type=int, default=0)
    ap.add_argument("--runs", type=int, default=300)
    ap.add_argument("--no-timing", action="store_true", help="plain launches (no per-kernel HIP events): what rocprofv3 should see")
    args = ap.parse_args()
    import __graft_entry__ as ge

    ge.build()
    import numpy as np

    import bench
    import harness
    import synth
    from grim import _native as nat
    from grim.imputation.impute import Imputation
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    gname, desc, dflt_n, scaling = bench.WORKLOADS[args.workload]
    n = args.subjects or dflt_n
    if gname == "wmda":
        import wmda_scale

        work = wmda_scale.ensure()
        conf = wmda_scale.conf()
    else:
        work = harness.ensure_graph(gname)
        conf = harness.base_conf(harness.POPS[gname])
        if gname == "pop4":
            conf["UNK_priors"] = "MR"
    cpath = os.path.join(work, "conf_kernel_only.json")
    json.dump(conf, open(cpath, "w"))
    os.chdir(work)
    cfg, _ = load_config(cpath)
    graph = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = bench.make_lines(args.workload, n, 0, 1, rows)
    text = ("\n".join(lines) + "\n").encode()
    imp = Imputation(graph, cfg, device=0)
    P = len(cfg["pops"])
    params = imp._params(cfg, cfg["planb"], False)
    ps, keep = nat.prior_spec(cfg["priority"], imp.unk_priors, imp.count_by_prob)
    ctx = nat.default_context(0)
    dgraph = graph.device(ctx)
    parsed = nat.Parsed(graph.adict, text, cfg["planb"])
    subj, toks = parsed.subjects(), parsed.tokens()
    n_tok = int(subj["cnt"].sum())
    priors = np.stack([nat.prior_matrix(ps, cfg["pops"], r1, r2) for r1, r2 in parsed.races()])
    batch = nat.DeviceBatch(ctx, dgraph, params, subj, toks, priors)
    batch.run()
    res, _ = batch.results()
    for _ in range(3):
        batch.run()
    batch.set_timing(not args.no_timing)
    for _ in range(args.runs):
        batch.run()
    ctr = batch.counters()
    ctr[3] = int(res["n_rows"].sum())
    algo = (8 * len(subj) + 2 * n_tok) + 16 * ctr[0] + 4 * ctr[1] + 8 * P * ctr[2] + 24 * ctr[3]
    names = {3: "grim_plan_a_small_kernel", 7: "grim_small_compact_kernel", 5: "grim_plan_a_medium_kernel", 9: "grim_plan_a_mid_kernel",
             4: "grim_plan_a_kernel", 2: "grim_plan_b_kernel", 6: "grim_tables_*"}
    ms = {v: batch.kernel_ms(0x10 | k) for k, v in names.items()} if not args.no_timing else {}
    out = {"workload": desc, "subjects": len(subj), "runs": args.runs, "algorithmic_bytes_per_launch": int(algo),
           "hip_event_mean_ms": ms}
    dom = max(ms, key=ms.get) if ms else None
    if dom and ms[dom] > 0:
        out["dominant"] = dom
        out["achieved_GBs_dominant"] = algo / (ms[dom] * 1e-3) / 1e9
        out["frac_of_8TBs"] = out["achieved_GBs_dominant"] / 8000.0
    print(json.dumps(out))
    batch.close()
    parsed.close()


if __name__ == "__main__":
    main()
