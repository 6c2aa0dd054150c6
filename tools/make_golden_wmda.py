#!/usr/bin/env python3
"""
tests/golden/wmda_full/: the REAL reference (/root/reference, build container only) on the FULL-SIZE config-5 stand-in --
the 300 000-haplotype, three-population graph of tools/wmda_scale.py (1.06 M nodes) -- for a few high-ambiguity subjects.
The graph CSVs come from the product generator (byte-identical to the reference generator's on the graphs where both were
run: tests/golden/graphs/*/graph_info.json); the reference LOADS them with its own Graph.build_graph and imputes with its
own Imputation.  Only inputs and outputs (data) are committed.

    python tools/make_golden_wmda.py [n_subjects=6]
"""
import contextlib
import io
import json
import os
import shutil
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
os.environ.setdefault("PYTHONHASHSEED", "0")
import make_golden as mg  # noqa: E402
import wmda_scale  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "wmda_full")
FILES = ("don.umug", "don.umug.pops", "don.pmug", "don.pmug.pops", "don.miss", "don.problem")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    t0 = time.time()
    work = wmda_scale.ensure()  # product generator (C++), no GPU needed
    print("graph ready (%.0f s): %s" % (time.time() - t0, work), flush=True)
    lines = wmda_scale.subjects(n)
    mg.prepare_reference()
    ref_work = os.path.join(mg.SCRATCH, "work", wmda_scale.name_of())
    os.makedirs(os.path.join(ref_work, "data", "subjects"), exist_ok=True)
    os.makedirs(os.path.join(ref_work, "output"), exist_ok=True)
    if not os.path.exists(os.path.join(ref_work, "output", "csv")):
        os.symlink(os.path.join(work, "output", "csv"), os.path.join(ref_work, "output", "csv"))
    shutil.copy(os.path.join(work, "output", "pop_counts_file.txt"), os.path.join(ref_work, "output"))
    conf = wmda_scale.conf()
    with open(os.path.join(ref_work, "data", "subjects", "input.csv"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    conf["imputation_in_file"] = "data/subjects/input.csv"
    conf["imputation_out_path"] = "output"
    with open(os.path.join(ref_work, "conf.json"), "w") as fh:
        json.dump(conf, fh, indent=1)
    for f in FILES:
        p = os.path.join(ref_work, "output", f)
        if os.path.exists(p):
            os.remove(p)
    # the reference runs in a child process of its own: this one has imported the PRODUCT's `grim` package (wmda_scale ->
    # harness), and the two packages share their name
    child = (
        "import sys, os, io, contextlib\n"
        "sys.path.insert(0, %r)\n"
        "os.chdir(%r)\n"
        "sys.argv = ['x']\n"
        "from grim import grim\n"
        "assert grim.__file__.startswith(%r), grim.__file__\n"
        "buf = io.StringIO()\n"
        "with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):\n"
        "    grim.impute('conf.json')\n"
        "open('ref_stdout.txt', 'w').write(buf.getvalue())\n" % (mg.SCRATCH, ref_work, mg.SCRATCH))
    import subprocess
    t1 = time.time()
    env = dict(os.environ, PYTHONHASHSEED="0")
    env.pop("PYTHONPATH", None)
    subprocess.check_call([sys.executable, "-c", child], env=env)
    buf = io.StringIO(open(os.path.join(ref_work, "ref_stdout.txt")).read())
    print("reference: graph load + %d subjects in %.0f s" % (n, time.time() - t1), flush=True)
    os.makedirs(OUT, exist_ok=True)
    shutil.copy(os.path.join(ref_work, "conf.json"), os.path.join(OUT, "reference_conf.json"))  # not "conf.json": harness.scenarios() lists directories holding one
    shutil.copy(os.path.join(ref_work, "data", "subjects", "input.csv"), os.path.join(OUT, "input.csv"))
    for f in FILES:
        p = os.path.join(ref_work, "output", f)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(OUT, f))
    log = [l for l in buf.getvalue().splitlines() if "Subject:" in l]
    with open(os.path.join(OUT, "log.txt"), "w") as fh:
        fh.write("\n".join(log) + "\n")
    with open(os.path.join(OUT, "meta.json"), "w") as fh:
        json.dump({"graph": wmda_scale.name_of(), "n_haps": wmda_scale.N_HAPS, "hap_pop_pair": False, "n_subjects": n,
                   "generator": "tools/make_golden_wmda.py (reference run in the build container)"}, fh)
    print("\n".join(log))


if __name__ == "__main__":
    main()
