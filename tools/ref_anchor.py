#!/usr/bin/env python3
"""
Reference anchor: time the REAL reference (py-graph-imputation @ /root/reference)
in this build container on the exact seeded inputs of BASELINE configs 2 and 4, with the split-and-fork pattern
of the reference's scripts/runfile_mp.py:109-148 (N contiguous chunks of the input, one forked process per
chunk, each calling Imputation.impute_file on its chunk).  Build-container only: the reference never travels,
only the numbers (profiles/r2_ref_anchor.json) are committed.

    python tools/ref_anchor.py [config2|config4|all] [--procs 1,8] [--n2 10000] [--n4 4000]
"""

import contextlib
import io
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (scratch copy of the reference + its one Cython module)
import synth  # noqa: E402


def _chunk_worker(work, conf_name, chunk_idx, q):
    """one process of runfile_mp.py: graph load + impute_file over one chunk (file -> files)"""
    os.chdir(work)
    sys.argv = ["x"]
    from grim import grim

    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        grim.impute(conf_name)
    q.put((chunk_idx, time.perf_counter() - t0))


def time_reference(work, pops, lines, procs, overrides, tag):
    conf = dict(mg.BASE_CONF, populations=list(pops))
    conf.update(overrides or {})
    n = len(lines)
    jobs = []
    for k in range(procs):
        lo, hi = n * k // procs, n * (k + 1) // procs
        c = dict(conf)
        c["imputation_in_file"] = "data/subjects/%s_%d.csv" % (tag, k)
        c["imputation_out_path"] = "output_%s_%d" % (tag, k)
        with open(os.path.join(work, c["imputation_in_file"]), "w") as fh:
            fh.write("\n".join(lines[lo:hi]) + "\n")
        name = "conf_%s_%d.json" % (tag, k)
        with open(os.path.join(work, name), "w") as fh:
            json.dump(c, fh)
        jobs.append(name)
    q = mp.Queue()
    t0 = time.perf_counter()
    ps = [mp.Process(target=_chunk_worker, args=(work, name, k, q)) for k, name in enumerate(jobs)]
    for p in ps:
        p.start()
    per = sorted(q.get() for _ in ps)
    for p in ps:
        p.join()
    wall = time.perf_counter() - t0
    return {"subjects": n, "procs": procs, "wall_s": round(wall, 3), "subjects_per_s": round(n / wall, 1),
            "slowest_chunk_s": round(max(t for _, t in per), 3)}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "all"
    procs = [1, 8]
    n2, n4 = 10000, 4000
    for i, a in enumerate(sys.argv):
        if a == "--procs":
            procs = [int(x) for x in sys.argv[i + 1].split(",")]
        if a == "--n2":
            n2 = int(sys.argv[i + 1])
        if a == "--n4":
            n4 = int(sys.argv[i + 1])
    os.environ.setdefault("PYTHONHASHSEED", "0")
    mg.prepare_reference()
    cau = synth.read_freqs(synth.CAU_FREQS)
    out = {"host": "build container, %d vCPU" % (os.cpu_count() or 0), "reference": "py-graph-imputation v0.1.1 @ /root/reference",
           "protocol": "split-and-fork as scripts/runfile_mp.py:109-148: N contiguous chunks, one process per chunk running "
                       "grim.impute(conf) (graph load + impute_file, file -> six files); wall = slowest process incl. fork", "runs": []}
    if which in ("config2", "all"):
        w1 = mg.build_graph("cau", ["CAU"]) if not os.path.exists(os.path.join(mg.SCRATCH, "work", "cau", "output", "csv", "nodes.csv")) \
            else os.path.join(mg.SCRATCH, "work", "cau")
        lines = synth.SubjectGen(cau, 0).full(n2)  # BASELINE configs[1]: 10k fully typed, seed 0
        for p in procs:
            r = time_reference(w1, ["CAU"], lines, p, None, "anchor2")
            r["config"] = "config 2: CAU 5-locus, %d fully typed subjects, synth.SubjectGen(seed 0).full" % n2
            print(json.dumps(r), flush=True)
            out["runs"].append(r)
    if which in ("config4", "all"):
        w4 = mg.build_graph("pop4", mg.POP4) if not os.path.exists(os.path.join(mg.SCRATCH, "work", "pop4", "output", "csv", "nodes.csv")) \
            else os.path.join(mg.SCRATCH, "work", "pop4")
        lines = synth.SubjectGen(cau, 3, pops=mg.POP4).mixed(n4)  # BASELINE configs[3] recipe, first n4 of seed 3
        for p in procs:
            r = time_reference(w4, mg.POP4, lines, p, {"UNK_priors": "MR"}, "anchor4")
            r["config"] = "config 4: 4-pop 5-locus, first %d subjects of synth.SubjectGen(seed 3, pop4).mixed (amb .3, miss .15, recomb .3), MR priors" % n4
            print(json.dumps(r), flush=True)
            out["runs"].append(r)
    path = os.path.join(ROOT, "profiles", "r2_ref_anchor.json")
    if os.path.exists(path):  # keep the runs of earlier invocations (other workloads / sizes / process counts)
        try:
            old = json.load(open(path)).get("runs", [])
        except ValueError:
            old = []
        key = lambda r: (r.get("config"), r.get("procs"), r.get("subjects"))
        new = {key(r) for r in out["runs"]}
        out["runs"] = [r for r in old if key(r) not in new] + out["runs"]
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
