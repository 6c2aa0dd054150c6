#!/usr/bin/env python3
"""
bench.py -- imputed subjects/sec on the CAU 5-locus graph (BASELINE.json metric).

A step = one pass of the grim.impute hot path (plan A kernel, plan B/C kernel when subjects need
it) over one batch of synthetic subjects that is already resident in HBM.  N=1 workload:
BASELINE.json configs[1] -- 10k fully typed synthetic subjects, seed 0.  N>1: weak scaling, every
rank owns its own 10k-subject batch (seed = rank), graph replicated per GPU, no data-path
collective; the timed region is bracketed by barrier + device sync and the max over ranks counts.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--subjects", type=int, default=10000, help="subjects per GPU (config 2: 10000)")
    ap.add_argument("--workload", default="full", choices=["full", "mixed"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-workers", type=int, default=8, help="processes of the CPU baseline (cpu_baseline.cores)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ["GRIM_QUIET"] = "1"

    import __graft_entry__ as ge

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        n_dev = torch.cuda.device_count()  # does not initialise the GPU
        backend = os.environ.get("GRIM_BENCH_BACKEND") or ("nccl" if n_dev > 0 else "gloo")
        if n_dev > 0:
            torch.cuda.set_device(local_rank % n_dev)  # ranks > devices only in rehearsals (gloo)
        dist.init_process_group(backend=backend)
        if rank == 0:
            ge.build()  # one rank (re)builds the library if it is stale; the others wait
        dist.barrier()
    ge.build()
    import harness
    import synth
    from grim import _native as nat
    from grim.imputation.impute import Imputation
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    # ---- graph + subjects (host work, outside the timed region) -----------------------------------
    if rank == 0:
        work = harness.ensure_graph("cau")
    if dist is not None:
        dist.barrier()
    work = harness.ensure_graph("cau")
    conf = harness.base_conf(["CAU"])
    cpath = os.path.join(work, "conf_bench_%d.json" % rank)
    json.dump(conf, open(cpath, "w"))
    os.chdir(work)
    cfg, _ = load_config(cpath)
    graph = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
    rows = synth.read_freqs(synth.CAU_FREQS)
    gen = synth.SubjectGen(rows, rank)  # seed 0 on rank 0 = config 2
    lines = gen.full(args.subjects) if args.workload == "full" else gen.mixed(args.subjects)
    n_dev_all = max(1, nat.lib().grim_device_count())
    imp = Imputation(graph, cfg, device=local_rank % n_dev_all)
    import numpy as np
    parsed = nat.Parsed(graph.adict, ("\n".join(lines) + "\n").encode(), cfg["planb"])  # C++ tokenizer of the library
    subj, toks = parsed.subjects(), parsed.tokens()
    assert len(subj) == len(lines)
    records = subj
    n_tok = int(subj["cnt"].sum())
    priors = np.stack([imp._prior_matrix(r1, r2, cfg["priority"]) for r1, r2 in parsed.races()])
    params = imp._params(cfg, cfg["planb"], False)
    ctx = nat.default_context(local_rank % n_dev_all)
    batch = nat.DeviceBatch(ctx, graph.device(ctx), params, subj, toks, priors)  # upload: subjects now resident in HBM
    batch.run()
    res, rows_out = batch.results()
    n_ok = int((res["status"] == nat.ST_OK).sum())

    def sync_barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        batch.run()
    # every timed step: each kernel bracketed by its own start/stop HIP events on the launch stream
    # (hipExtLaunchKernelGGL); the library keeps the per-kernel sums, read once after the loop
    batch.set_timing(True)
    sync_barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run()  # one complete synchronous run: kernels, stream synchronise, state check
    sync_barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ctr = batch.counters()  # probes, CSR neighbour ids, frequency vectors gathered, (row pool use)
    ctr[3] = int(res["n_rows"].sum())  # rows actually produced
    P = len(cfg["pops"])
    # B_subj = B_in + sum_sides(16 q + 4 nbr + 8 P c) + 24 B_rows ; B_in = 4 + 2/token + 4
    algo_bytes = (8 * len(records) + 2 * n_tok) + 16 * ctr[0] + 4 * ctr[1] + 8 * P * ctr[2] + 24 * ctr[3]
    # HIP-event time of each kernel, averaged over the timed steps; the roofline is quoted for the
    # dominant one (config 2: every subject takes the half-wave kernel)
    names = ("grim_plan_a_small_kernel", "grim_plan_a_kernel", "grim_plan_b_kernel", "grim_plan_a_medium_kernel")
    per_kernel = [batch.kernel_ms(0x10 | w) for w in (3, 4, 2, 5)]  # means over the timed steps
    dom = max(range(3), key=lambda i: per_kernel[i])
    avg_ms = per_kernel[dom]
    achieved = algo_bytes / (avg_ms * 1e-3) / 1e9

    # HBM traffic per launch of the dominant kernel: PMC numbers cannot be collected from inside this
    # process; they come from the committed rocprofv3 --pmc passes over this same command
    traffic, traffic_src = None, None
    pmc_file = "r1g_pmc_traffic.json"  # newest committed PMC passes (tools/profile_round.sh)
    pmc_path = os.path.join(ROOT, "profiles", pmc_file)
    if os.path.exists(pmc_path) and args.workload == "full" and args.subjects == 10000:
        pmc = json.load(open(pmc_path)).get(names[dom])
        if pmc and "hbm_bytes_raw" in pmc:
            traffic = pmc["hbm_bytes_raw"]
            traffic_src = "profiles/" + pmc_file + ": (FETCH_SIZE + WRITE_SIZE) KB x 1024 per launch, separate --pmc passes; " \
                          "with FETCH_SIZE doubled (gfx950 wide-read correction): %d" % pmc["hbm_bytes_fetch_doubled"]

    out = None
    if rank == 0:
        out = {
            "metric": "imputed subjects/sec (whole node), 5-locus CAU graph",
            "value": world * len(records) * args.steps / elapsed,
            "unit": "subjects/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "CAU 5-locus graph, %d synthetic %s subjects per GPU, seed=rank (BASELINE configs[1])"
                            % (args.subjects, "fully-typed" if args.workload == "full" else "mixed"),
                "subjects_per_gpu": len(records), "subjects_with_results": n_ok,
                "graph_nodes": int(graph.arrays["n_nodes"]), "populations": P,
                "inputs": "tokenised subjects resident in HBM; results left in HBM",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": names[dom], "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms": dict(zip(names, per_kernel)),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            # bounded sample, about 15 s of wall time: the batch itself plus more of the same generator, split over
            # worker processes the way the reference's scripts/runfile_mp.py splits a file (one oracle per chunk,
            # each loading the graph itself); the workers never touch the GPU
            import subprocess

            workers = max(1, min(args.cpu_workers, os.cpu_count() or 1))
            if args.workload == "full":
                sample = lines[:10000] + synth.SubjectGen(rows, 1000).full(50000 * workers)
            else:
                sample = lines[: min(len(lines), 8000 * workers)]
            spath = os.path.join(work, "data", "subjects", "bench_cpu.csv")
            with open(spath, "w") as fh:
                fh.write("\n".join(sample) + "\n")
            cjson = os.path.join(work, "conf_bench_cpu.json")
            json.dump(conf, open(cjson, "w"))
            t1 = time.perf_counter()
            procs = []
            for k in range(workers):
                lo, hi = len(sample) * k // workers, len(sample) * (k + 1) // workers
                procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_worker.py"), "cau", cjson, spath,
                                               str(lo), str(hi)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL))
            done = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in procs]
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {
                "value": len(sample) / dt, "unit": "subjects/s", "cores": workers, "kind": "port",
                "sample": "%d subjects of the same generator through oracle/grim_oracle.py (Python restatement of the reference), "
                          "%d worker processes with a contiguous chunk each as in scripts/runfile_mp.py, %.1f s wall including each "
                          "worker's graph load; slowest worker %.1f s of imputation" % (len(sample), workers, dt, max(d["s"] for d in done)),
            }
        print(json.dumps(out))
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
