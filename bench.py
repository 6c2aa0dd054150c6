#!/usr/bin/env python3
"""
bench.py -- imputed subjects/sec on the CAU 5-locus graph (BASELINE.json metric).

A step = one pass of the grim.impute hot path over one batch of synthetic subjects.  N=1 workload (default):
BASELINE.json configs[1] -- 10k fully typed synthetic subjects, seed 0.

What is timed (SURVEY 8d protocol), all in one run:
  value / ms_per_step   HEADLINE: host GL strings in memory -> result records in host memory.  Every step hands the batch's
                        text to the library's streaming pipeline (grim_stream, records mode): tokenizer threads -> pinned
                        staging -> H2D -> kernels -> D2H -> records in pinned host memory, consumed by this process.
                        Steps are pipelined (depth 4) the way a file's chunks are; the graph is resident in HBM.
  kernel_only/roofline  the same batch resident in HBM, kernels only (per-kernel HIP events on the launch stream):
                        the input of the roofline fraction (algorithmic bytes / kernel time / 8 TB/s).
  file_to_file          Imputation.impute_file: subject file on disk -> the six output files on disk (adds the formatter
                        threads and the ordered pwrites).
  cpu_baseline          the oracle (Python restatement of the reference) on the box's host cores, strings in memory ->
                        result texts in memory, graph load excluded -- the same scope as the headline.
N>1: one process per GPU, weak scaling (every rank owns its own batch, seed = rank), graph replicated, no data-path
collective; barrier + device sync around the timed region, max over ranks.  `--workload config3` is the strong-scaling
variant (1 M subjects in total, split over the ranks).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MAX_CHUNK = 131072

WORKLOADS = {
    # name: (graph, description, default subjects per step on ONE GPU, scaling)
    "config2": ("cau", "BASELINE configs[1]: CAU 5-locus graph, 10k synthetic fully-typed subjects", 10000, "weak"),
    "config3": ("cau", "BASELINE configs[2]: CAU 5-locus graph, 1M synthetic fully-typed subjects (seed 1) split over the GPUs", 1000000, "strong"),
    "config4": ("pop4", "BASELINE configs[3]: 4-population 5-locus graph, 100k subjects with missing loci / ambiguity / recombinants (seed 3), MR priors", 100000, "strong"),
    "config5": ("wmda", "BASELINE configs[4] (synthetic stand-in): WMDA-scale multi-population graph, high-ambiguity subjects, options threshold 1e6, 100 haplotypes in phase", 2048, "strong"),
    "mixed": ("cau", "CAU 5-locus graph, mixed subjects (ambiguity, missing loci, recombinants); not a BASELINE config", 10000, "weak"),
}


PROFILE_TAG = "r4"  # profiles/<tag>_*: the committed rocprofv3 evidence bench.py cites (tools/profile_round.sh + collect_profiles.py)


def source_digest():
    """sha256 over the kernel / host sources and the C header: identifies the BUILD a committed profile was taken of"""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "py-graph-imputation_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc)) + [os.path.join(ROOT, "include", "grim_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def committed_profile(workload):
    """what profiles/ holds for this workload -- NOT measured in this run: rocprofv3 averages / minima of every kernel of the
    stream path, the kernel-only pass (config 2), the PMC traffic; with the digest of the sources they were taken of and
    whether that is the build being measured now"""
    import csv

    out = {"note": "committed profile (tools/profile_round.sh on a gpurun box), not this run"}
    meta_path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_meta.json")
    if not os.path.exists(meta_path):
        return None
    meta = json.load(open(meta_path))
    out["profile_commit"] = meta.get("commit")
    out["profile_source_digest"] = meta.get("source_digest")
    out["this_build_source_digest"] = source_digest()
    out["same_build"] = out["profile_source_digest"] == out["this_build_source_digest"]

    def stats(path):
        avg, mn, calls = {}, {}, {}
        for r in csv.DictReader(open(path)):
            k = r["Name"].split("(")[0]
            if k.startswith("grim_"):
                avg[k] = round(float(r["AverageNs"]) / 1e3, 2)
                mn[k] = round(float(r["MinNs"]) / 1e3, 2)
                calls[k] = int(r["Calls"])
        return {"avg_us": avg, "min_us": mn, "calls": calls}

    sp = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (PROFILE_TAG, workload))
    if os.path.exists(sp):
        out["stream_path"] = dict(stats(sp), source="profiles/" + os.path.basename(sp) + " (bench.py under rocprofv3: copies and the other "
                                  "kernels of the stream run beside each launch and stretch the averages)")
    kp = os.path.join(ROOT, "profiles", "%s_%s_kernel_only_stats.csv" % (PROFILE_TAG, workload))
    if os.path.exists(kp):
        out["kernel_only"] = dict(stats(kp), source="profiles/" + os.path.basename(kp) + " (tools/kernel_only.py under rocprofv3: the "
                                  "resident-batch loop alone -- the averages avg_launch_ms has to agree with)")
    tp = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_traffic.json")
    if os.path.exists(tp):
        out["pmc_traffic_file"] = "profiles/" + os.path.basename(tp)
    return out


def make_lines(workload, n, rank, world, rows):
    import harness
    import synth

    if workload == "config2":
        return synth.SubjectGen(rows, rank).full_fast(n)  # seed 0 on rank 0 = config 2
    if workload == "config3":
        total = n * world
        return synth.SubjectGen(rows, 1).full_fast(total)[rank * n:(rank + 1) * n]
    if workload == "config4":
        total = n * world
        return synth.SubjectGen(rows, 3, pops=harness.POPS["pop4"]).mixed(total)[rank * n:(rank + 1) * n]
    if workload == "config5":
        import wmda_scale
        return wmda_scale.subjects(n * world)[rank * n:(rank + 1) * n]
    return synth.SubjectGen(rows, rank).mixed(n)


def launch_ranks(n):
    """`python bench.py --gpus N` outside a torchrun job: run `python -m torch.distributed.run --nproc-per-node N bench.py
    <same arguments>` as a child (one rank per GPU), stdout / stderr inherited so rank 0's JSON line comes through, and exit
    with its code.  (What scripts/runfile_mp.py:109-148 does for the reference: the script forks its own workers.)"""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS) + ["full"])
    ap.add_argument("--subjects", type=int, default=0, help="subjects per step and GPU (default: the workload's size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--copy-input", action="store_true", help="the stream copies the GL strings it is given (grim_stream_write) instead of reading "
                    "them where they are (grim_stream_write_borrowed: the default here -- the strings are in memory for the whole run)")
    ap.add_argument("--no-file", action="store_true", help="skip the file -> file leg")
    ap.add_argument("--file-subjects", type=int, default=1000000, help="lines of the file -> file leg (config 2/3: 1M = config 3's file)")
    ap.add_argument("--cpu-workers", type=int, default=8, help="processes of the CPU baseline (cpu_baseline.cores)")
    ap.add_argument("--kernel-steps", type=int, default=50, help="runs of the resident-batch kernel loop (roofline)")
    ap.add_argument("--chunk-lines", type=int, default=0, help="lines per device batch of the stream (default: the whole step, at most %d)" % MAX_CHUNK)
    ap.add_argument("--depth", type=int, default=8, help="device batches the stream keeps in flight (a step's latency through tokenizer, H2D, kernels, "
                    "D2H and the consumer is several steps long: the depth, not the slowest stage, bounds the rate when it is too small)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="the timed region of --steps steps is repeated until it has run this long in total; "
                    "ms_per_step is the MEDIAN region (min / max beside it)")
    ap.add_argument("--max-repeats", type=int, default=400)
    ap.add_argument("--sharded", action="store_true", help="time the product's multi-GPU driver instead: grim.shard.impute_sharded, subject file on disk -> "
                    "the six merged output files on disk, chunks pulled from the job's store (a step = one whole file)")
    args = ap.parse_args()
    if args.workload == "full":
        args.workload = "config2"
    if args.workload == "config5":
        # a step is ONE device batch here and each of the stream's four batch slots sizes its (multi-GB) pair pool on first
        # use: the untimed steps must touch every slot
        args.warmup = max(args.warmup, 4)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: start the N ranks as CHILD processes (nothing in this process has touched the
        # GPU, and it never will), hand their output and exit code on
        return launch_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the job has WORLD_SIZE=%d ranks (start it as `python bench.py --gpus N`, or with "
                 "torchrun --nproc-per-node N bench.py --gpus N)" % (args.gpus, world))
    os.environ["GRIM_QUIET"] = "1"

    import __graft_entry__ as ge

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The path has no data-path collective (subjects shard, the graph is replicated): the process group is the control
        # plane only -- barriers around the timed regions and the max over ranks of their times -- so it is gloo, on CPU
        # tensors, and torch never opens the GPU beside libgrim_hip.so.  GRIM_BENCH_BACKEND=nccl puts the same calls on RCCL.
        n_dev = torch.cuda.device_count()  # does not initialise the GPU
        backend = os.environ.get("GRIM_BENCH_BACKEND") or "gloo"
        if backend == "nccl" and n_dev > 0:
            torch.cuda.set_device(local_rank % n_dev)
        dist.init_process_group(backend=backend)
        if rank == 0:
            ge.build()  # one rank (re)builds the library if it is stale; the others wait
        dist.barrier()
    ge.build()
    import numpy as np

    import harness
    import synth
    from grim import _native as nat
    from grim.imputation.impute import Imputation
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    gname, desc, dflt_n, scaling = WORKLOADS[args.workload]
    n_step = args.subjects or (dflt_n if scaling == "weak" else max(1, dflt_n // world))
    # chunk size of the stream: a divisor of the step so that every step is a whole number of device batches
    max_chunk = args.chunk_lines if args.chunk_lines > 0 else MAX_CHUNK
    m = (n_step + max_chunk - 1) // max_chunk
    while n_step % m:
        m += 1
    chunk_lines = n_step // m

    # ---- graph + subjects (host work, outside every timed region) ------------------------------------
    if gname == "wmda":
        import wmda_scale
        if rank == 0:
            wmda_scale.ensure()
        if dist is not None:
            dist.barrier()
        work = wmda_scale.ensure()
        conf = wmda_scale.conf()
        gname = wmda_scale.name_of()
    else:
        if rank == 0:
            harness.ensure_graph(gname)
        if dist is not None:
            dist.barrier()
        work = harness.ensure_graph(gname)
        conf = harness.base_conf(harness.POPS[gname])
        if gname == "pop4":
            conf["UNK_priors"] = "MR"
    cpath = os.path.join(work, "conf_bench_%d.json" % rank)
    json.dump(conf, open(cpath, "w"))
    os.chdir(work)
    cfg, _ = load_config(cpath)
    graph = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = make_lines(args.workload, n_step, rank, world, rows)
    text = ("\n".join(lines) + "\n").encode()
    n_dev_all = max(1, nat.lib().grim_device_count())
    imp = Imputation(graph, cfg, device=local_rank % n_dev_all)
    imp.quiet = True
    P = len(cfg["pops"])
    params = imp._params(cfg, cfg["planb"], False)
    ps, keep = nat.prior_spec(cfg["priority"], imp.unk_priors, imp.count_by_prob)
    ctx = nat.default_context(local_rank % n_dev_all)
    dgraph = graph.device(ctx)

    def sync_barrier():
        if dist is not None:
            dist.barrier()

    ranks_seen = [{"rank": rank, "device": local_rank % n_dev_all}]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks_seen[0])
        ranks_seen = gathered

    if args.sharded:
        return sharded_leg(args, rank, world, dist, work, conf, cfg, graph, lines, gname, desc, n_step, scaling, ranks_seen)

    # ---- A. headline: host strings -> host records, K steps through one stream ------------------------
    # One timed REGION = exactly --steps steps between two barriers.  A region of config 2 lasts a few milliseconds, far too
    # short to measure on a shared box, so the region is repeated (same stream, pipeline drained in between) until the regions
    # add up to --min-seconds; the reported step time is the median region's, with the fastest and slowest beside it.
    st = nat.Stream(ctx, dgraph, graph.adict, params, ps, cfg["pops"], want_text=False, want_records=True,
                    chunk_lines=chunk_lines, depth=args.depth)
    chunks_per_step = n_step // chunk_lines
    import queue
    cmds = queue.Queue()
    feed_err = []

    def feed():
        try:
            while True:
                n = cmds.get()
                if n is None:
                    break
                # one call per region (the way a file comes in: big blocks): the library cuts the text into steps of
                # chunk_lines lines itself, and this thread does not compete for the interpreter with the consumer
                st.write(region_text if n == args.steps else text * n, borrowed=not args.copy_input)
            st.finish()
        except Exception as e:  # pragma: no cover
            feed_err.append(e)

    region_text = text * args.steps
    th = threading.Thread(target=feed)
    th.start()
    seen = {"lines": 0, "ok": 0}

    def drain(n_chunks):
        for _ in range(n_chunks):
            rec = st.next_records()
            if rec is None:
                raise RuntimeError("stream ended early: %r" % feed_err)
            first_line, kinds, res, rows_addr, handle = rec
            seen["lines"] += len(kinds)
            # the consumer reads every record's status (a byte view: the structured-field version of this line cost 80 us
            # per 10 000 records and had become the slowest stage of the pipeline)
            status = res.view(np.uint8).reshape(len(res), nat.RESULT_DT.itemsize)[:, 0]
            seen["ok"] += int(np.count_nonzero((status == nat.ST_OK) & (kinds == nat.K_DEVICE)))
            st.release(handle)

    def agree_max(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def region():
        sync_barrier()
        t0 = time.perf_counter()
        cmds.put(args.steps)
        drain(args.steps * chunks_per_step)
        sync_barrier()
        return agree_max(time.perf_counter() - t0)  # max over ranks

    cmds.put(args.warmup)
    drain(args.warmup * chunks_per_step)
    seen.update(lines=0, ok=0)
    regions = [region()]
    n_ok_per_step = seen["ok"] // max(1, args.steps)
    repeats = int(min(args.max_repeats, max(3, -(-args.min_seconds // max(regions[0], 1e-9)))))  # the same on every rank
    while len(regions) < repeats:
        regions.append(region())
    cmds.put(None)
    th.join()
    if feed_err:
        raise feed_err[0]
    assert st.next_records() is None
    sstats = st.stats()
    st.close()
    regions.sort()
    elapsed = regions[len(regions) // 2]  # the median region: exactly --steps steps

    # ---- B. kernels only: the same batch resident in HBM (roofline input) ----------------------------------
    parsed = nat.Parsed(graph.adict, text, cfg["planb"])
    subj, toks = parsed.subjects(), parsed.tokens()
    n_tok = int(subj["cnt"].sum())
    priors = np.stack([nat.prior_matrix(ps, cfg["pops"], r1, r2) for r1, r2 in parsed.races()])
    batch = nat.DeviceBatch(ctx, dgraph, params, subj, toks, priors)
    batch.run()
    res, rows_out = batch.results()
    for _ in range(3):
        batch.run()
    batch.set_timing(True)  # every kernel bracketed by its own start/stop HIP events on the launch stream
    tk0 = time.perf_counter()
    for _ in range(args.kernel_steps):
        batch.run()
    kernel_loop_s = time.perf_counter() - tk0
    ctr = batch.counters()  # probes, CSR neighbour ids, frequency vectors gathered, (row pool use)
    ctr[3] = int(res["n_rows"].sum())  # rows actually produced
    # B_subj = B_in + sum_sides(16 q + 4 nbr + 8 P c) + 24 B_rows ; B_in = 4 + 2/token + 4
    algo_bytes = (8 * len(subj) + 2 * n_tok) + 16 * ctr[0] + 4 * ctr[1] + 8 * P * ctr[2] + 24 * ctr[3]
    names = ("grim_plan_a_small_kernel", "grim_plan_a_kernel", "grim_plan_b_kernel", "grim_plan_a_medium_kernel",
             "grim_tables_wave_kernel+grim_tables_split_kernel+grim_tables_bucket_kernel+grim_tables_merge_kernel",
             "grim_small_compact_kernel", "grim_plan_a_mid_kernel")
    per_kernel = [batch.kernel_ms(0x10 | w) for w in (3, 4, 2, 5, 6, 7, 9)]  # means over the timed runs
    dom = max(range(len(per_kernel)), key=lambda i: per_kernel[i])
    avg_ms = per_kernel[dom]
    all_ms = sum(per_kernel)
    # the roofline is quoted for the dominant kernel against the bytes of the whole pass when that kernel moves all of
    # them (config 2/3: every subject is the half-wave kernel's; the row compaction that follows it re-packs rows for
    # the D2H copy and has no algorithmic bytes of its own -- its time is listed in kernel_ms and counted in
    # kernel_only), else for the sum of the kernels
    whole = dom == 0 and (per_kernel[0] + per_kernel[5]) >= 0.95 * all_ms
    roof_ms = avg_ms if whole else all_ms
    achieved = algo_bytes / (roof_ms * 1e-3) / 1e9 if roof_ms > 0 else 0.0
    batch.close()
    parsed.close()

    # HBM traffic per launch of the dominant kernel: PMC numbers cannot be collected from inside this process; they come from
    # the committed rocprofv3 --pmc passes over this same command (tools/profile_round.sh) and are labelled as such
    prof = committed_profile(args.workload) if n_step == dflt_n else None
    traffic, traffic_src = None, None
    if prof and prof.get("pmc_traffic_file"):
        pmc = json.load(open(os.path.join(ROOT, prof["pmc_traffic_file"]))).get(args.workload, {}).get(names[dom])
        if pmc and "hbm_bytes_raw" in pmc:
            traffic = pmc["hbm_bytes_raw"]
            traffic_src = prof["pmc_traffic_file"] + ": (FETCH_SIZE + WRITE_SIZE) KB x 1024 per launch, separate --pmc passes (committed " \
                          "profile of source digest %s, this build %s); with FETCH_SIZE doubled (gfx950 wide-read correction): %d" % (
                              prof["profile_source_digest"], prof["this_build_source_digest"], pmc["hbm_bytes_fetch_doubled"])
    rocprof = prof
    frac_committed = None
    if prof and whole and prof.get("kernel_only", {}).get("avg_us", {}).get(names[dom]):
        # the roofline fraction recomputed from the COMMITTED kernel-only average: what a reader can check without a GPU
        frac_committed = algo_bytes / (prof["kernel_only"]["avg_us"][names[dom]] * 1e-6) / 1e9 / HBM_PEAK_GBS

    out = None
    if rank == 0:
        total_subjects = world * n_step * args.steps  # per region
        out = {
            "metric": "imputed subjects/sec (whole node), 5-locus CAU graph",
            "value": total_subjects / elapsed,
            "unit": "subjects/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_min": 1e3 * regions[0] / args.steps, "ms_per_step_max": 1e3 * regions[-1] / args.steps,
            "repeats": len(regions), "timed_s": sum(regions),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": desc + ("; %d subjects per step and GPU, seed=rank" % n_step if scaling == "weak" else
                                    "; %d subjects per step and GPU" % n_step),
                "subjects_per_step_per_gpu": n_step, "subjects_with_results_per_step": n_ok_per_step, "ranks": ranks_seen,
                "graph_nodes": int(graph.arrays["n_nodes"]), "populations": P,
                "input": "copied by the stream (grim_stream_write)" if args.copy_input else "lent to the stream (grim_stream_write_borrowed): the tokenizer threads read the strings where they lie",
                "timed_region": "host GL strings in memory -> grim_stream (tokenizer threads, pinned staging, H2D, kernels, D2H) -> "
                                "result records in pinned host memory, read by the caller; %d device batch(es) of %d lines per step, "
                                "%d in flight; value = subjects of one region of --steps steps / the MEDIAN region time over `repeats` "
                                "back-to-back regions (barrier + drained pipeline on both sides of each)" % (chunks_per_step, chunk_lines, args.depth),
                "host_threads": int(os.environ.get("GRIM_HOST_THREADS", "0")) or int(nat.host_lib().grim_default_threads()),
                "stream_cpu_s": {"tokenize": sstats.tokenize_cpu_s, "device_thread_busy": sstats.device_s},
                "results_download": (lambda e: "SDMA engine 0x%x named through ROCr (csrc/grim_sdma.h); uploads on the HIP runtime's engine" % e
                                     if e > 1 else "copy kernel (grim_export_kernel)" if e == 0 else "hipMemcpyAsync")(ctx.export_engine()),
            },
            "kernel_only": {
                "subjects_per_s": len(subj) / (all_ms * 1e-3) if all_ms > 0 else None,
                "kernel_ms_per_step": all_ms,
                "synchronous_run_ms": 1e3 * kernel_loop_s / args.kernel_steps,
                "inputs": "tokenised subjects resident in HBM; results left in HBM (round 1's headline)",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": names[dom] if whole else "+".join(n for n, v in zip(names, per_kernel) if v > 0),
                "avg_launch_ms": roof_ms, "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms": dict(zip(names, per_kernel)),
                "frac_from_committed_kernel_only_profile": frac_committed,
                "rocprof": rocprof,
            },
        }
        if world == 1 and not args.no_file:
            out["file_to_file"] = file_leg(args, work, conf, cfg, imp, lines, rows, n_step)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_leg(args, work, conf, lines, rows, gname)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def sharded_leg(args, rank, world, dist, work, conf, cfg, graph, lines, gname, desc, n_step, scaling, ranks_seen):
    """--sharded: the product's multi-GPU driver, file -> files.  Every rank calls grim.shard.impute_sharded on the SAME
    conf; chunks of --chunk-lines lines (default 65536) are pulled from the job's store, every rank streams its chunks
    through one long-lived grim_stream and writes its pieces straight into the six shared output files at offsets that follow
    from the chunk sizes published on the store.  A step = the whole file (world x n_step subjects); --steps steps timed one
    by one, median reported."""
    from grim import shard

    all_lines = lines
    if dist is not None:  # every rank generated its slice of the strong-scaling workload: the file is the whole of it
        gathered = [None] * world
        dist.all_gather_object(gathered, lines)
        all_lines = [l for part in gathered for l in part]
    path = os.path.join(work, "data", "subjects", "bench_sharded.csv")
    out_dir = os.path.join(work, "output_bench_sharded")
    c2 = dict(conf)
    c2["imputation_in_file"] = "data/subjects/bench_sharded.csv"
    c2["imputation_out_path"] = "output_bench_sharded"
    cpath = os.path.join(work, "conf_bench_sharded.json")
    if rank == 0:
        with open(path, "w") as fh:
            fh.write("\n".join(all_lines) + "\n")
        json.dump(c2, open(cpath, "w"))
        os.makedirs(out_dir, exist_ok=True)
    if dist is not None:
        dist.barrier()
    chunk = args.chunk_lines or 65536
    times = []
    for k in range(args.warmup + args.steps):
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        shard.impute_sharded(cpath, graph=graph, chunk_lines=chunk)  # ends with the job's barrier
        dt = time.perf_counter() - t0
        if dist is not None:
            import torch
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if k >= args.warmup:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    if rank == 0:
        sizes = {k: os.path.getsize(os.path.join(out_dir, f)) for k, f in (("umug", "don.umug"), ("pmug", "don.pmug")) if os.path.exists(os.path.join(out_dir, f))}
        print(json.dumps({
            "metric": "imputed subjects/sec (whole node), 5-locus CAU graph", "value": len(all_lines) / med, "unit": "subjects/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * med,
            "ms_per_step_min": 1e3 * times[0], "ms_per_step_max": 1e3 * times[-1], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc + "; --sharded: grim.shard.impute_sharded, subject file on disk -> six merged output files on disk "
                                          "(page cache, no fsync), %d subjects in the file, chunks of %d lines pulled from the job's store" % (len(all_lines), chunk),
                       "subjects_per_step": len(all_lines), "host_threads_per_rank": int(nat_threads()), "output_bytes": sizes, "ranks": ranks_seen},
        }))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def nat_threads():
    from grim import _native as nat
    return nat.host_lib().grim_default_threads()


def file_leg(args, work, conf, cfg, imp, lines, rows, n_step):
    """Imputation.impute_file, subject file on disk -> six output files on disk; best of 3 after one warm run."""
    import synth

    n_file = n_step
    flines = lines
    if args.workload in ("config2", "config3") and args.file_subjects > n_step:
        n_file = args.file_subjects  # config 3's file: 1M subjects, seed 1
        flines = synth.SubjectGen(rows, 1).full_fast(n_file)
    path = os.path.join(work, "data", "subjects", "bench_file.csv")
    with open(path, "w") as fh:
        fh.write("\n".join(flines) + "\n")
    c2 = dict(cfg)
    c2["imputation_input_file"] = path
    out_dir = os.path.join(work, "output_bench_file")
    os.makedirs(out_dir, exist_ok=True)
    for key, path_key, flag in imp._OUT_FILES:
        c2[path_key] = os.path.join(out_dir, key + ".txt")
    imp.impute_file(c2)
    best, stats = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        imp.impute_file(c2)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, stats = dt, dict(imp.last_stats)
    return {
        "subjects": n_file, "seconds": best, "subjects_per_s": n_file / best,
        "what": "Imputation.impute_file: subject file on disk -> .umug/.umug.pops/.pmug/.pmug.pops/.miss/.problem on disk (best of 3; "
                "page cache, no fsync -- the reference does not sync either)",
        "pipeline_wall_s": stats.get("wall_s"), "device_thread_busy_s": stats.get("device_s"),
        "cpu_s": stats.get("host_s"), "output_bytes": sum(stats.get("text_bytes", [])[:6]), "chunks": stats.get("chunks"),
    }


def cpu_leg(args, work, conf, lines, rows, gname):
    """bounded sample, about 15 s of wall time: more subjects of the same generator, split over worker processes the way
    the reference's scripts/runfile_mp.py splits a file (one oracle per chunk, each loading the graph itself; the graph
    load is NOT counted: the GPU side's graph is resident too); the workers never touch the GPU"""
    import subprocess

    import synth

    workers = max(1, min(args.cpu_workers, os.cpu_count() or 1))
    if args.workload in ("config2", "config3"):
        sample = synth.SubjectGen(rows, 1000).full_fast(20000 * workers)
    elif args.workload == "config5":
        workers = min(workers, 2)  # every worker loads the 1.1 M-node graph into Python dicts (minutes, gigabytes)
        sample = lines[: 2 * workers]
    else:
        sample = lines[: min(len(lines), 2500 * workers)]
    spath = os.path.join(work, "data", "subjects", "bench_cpu.csv")
    with open(spath, "w") as fh:
        fh.write("\n".join(sample) + "\n")
    cjson = os.path.join(work, "conf_bench_cpu.json")
    json.dump(conf, open(cjson, "w"))
    t1 = time.perf_counter()
    procs = []
    for k in range(workers):
        lo, hi = len(sample) * k // workers, len(sample) * (k + 1) // workers
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_worker.py"), gname, cjson, spath,
                                       str(lo), str(hi)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL))
    done = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in procs]
    dt = time.perf_counter() - t1
    slowest = max(d["s"] for d in done)
    return {
        "value": len(sample) / slowest, "unit": "subjects/s", "cores": workers, "kind": "port",
        "sample": "%d subjects of the same generator through oracle/grim_oracle.py (Python restatement of the reference), %d worker "
                  "processes with a contiguous chunk each as in scripts/runfile_mp.py; input lines in memory -> output texts in memory, "
                  "slowest worker %.1f s (graph load excluded, as on the GPU side; %.1f s wall with it)" % (len(sample), workers, slowest, dt),
        "with_graph_load_subjects_per_s": len(sample) / dt,
    }


if __name__ == "__main__":
    main()
