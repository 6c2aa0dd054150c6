#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (what tools/profile_round.sh left) -> profiles/<tag>_*: the per-workload kernel stats as rocprofv3
wrote them, one summary line per (kernel, counter) of every --pmc pass, and the traffic JSON.
    python tools/collect_profiles.py [tag=r2]"""
import collections, csv, glob, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
for w in ("config2", "config3", "config4", "config5"):
    st = os.path.join(src, w, "r_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, w)))
    for kind in ("fetch", "write", "sq"):
        d = os.path.join(src, "%s_%s" % (w, kind))
        files = glob.glob(os.path.join(d, "*counter_collection.csv"))
        if not files:
            continue
        acc = collections.defaultdict(list)
        for f in files:
            for r in csv.DictReader(open(f)):
                acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        with open(os.path.join(dst, "%s_%s_pmc_%s_summary.csv" % (tag, w, kind)), "w") as out:
            unit = "KB" if kind != "sq" else "count"
            out.write("kernel,counter,launches,mean_%s,min_%s,max_%s\n" % (unit, unit, unit))
            for (k, c), v in sorted(acc.items()):
                out.write("%s,%s,%d,%.1f,%.1f,%.1f\n" % (k, c, len(v), sum(v) / len(v), min(v), max(v)))
ko = os.path.join(src, "config2_kernel_only", "r_kernel_stats.csv")
if os.path.exists(ko):
    shutil.copy(ko, os.path.join(dst, "%s_config2_kernel_only_stats.csv" % tag))
    log = os.path.join(src, "config2_kernel_only.log")
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            open(os.path.join(dst, "%s_config2_kernel_only.json" % tag), "w").write(lines[-1])
for w in ("config2", "config2_steps20", "config3", "config4", "config5"):
    bj = os.path.join(src, "bench_%s.json" % w)
    if os.path.exists(bj):
        lines = [l for l in open(bj) if l.startswith("{")]
        if lines:
            open(os.path.join(dst, "%s_bench_%s.json" % (tag, w)), "w").write(lines[-1])
tj = os.path.join(src, "pmc_traffic.json")
if os.path.exists(tj):
    shutil.copy(tj, os.path.join(dst, "%s_pmc_traffic.json" % tag))
# which build the profiles are of: a digest of the kernel / host sources (bench.py computes the same one at run time and says
# whether the committed profile it cites is of the build it is measuring) and the commit they were collected at
import hashlib, json, subprocess
sys.path.insert(0, ROOT)
import bench
meta = {"source_digest": bench.source_digest(), "tag": tag}
try:
    meta["commit"] = subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=ROOT).decode().strip()
    meta["dirty"] = bool(subprocess.check_output(["git", "status", "--porcelain", "--", "py-graph-imputation_amd/csrc", "include"], cwd=ROOT).decode().strip())
except Exception:
    pass
json.dump(meta, open(os.path.join(dst, "%s_meta.json" % tag), "w"), indent=1)
print("profiles/%s_* written" % tag)
