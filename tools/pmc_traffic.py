#!/usr/bin/env python3
"""profiles/<round>_pmc_traffic.json from the rocprofv3 counter_collection CSVs of tools/profile_round.sh (one --pmc FETCH_SIZE
pass and one --pmc WRITE_SIZE pass per workload, as MI355X_MICROARCH's HBM section prescribes):
    python tools/pmc_traffic.py <prof dir with W_fetch/ and W_write/> out.json
-> {workload: {kernel: {FETCH_SIZE_KB_per_launch, WRITE_SIZE_KB_per_launch, hbm_bytes_raw, hbm_bytes_fetch_doubled, launches}}}"""
import collections, csv, glob, json, os, sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(path, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


root, out_path = sys.argv[1], sys.argv[2]
out = {}
for w in ("config2", "config3", "config4", "config5"):
    fd, wd = os.path.join(root, w + "_fetch"), os.path.join(root, w + "_write")
    if not (os.path.isdir(fd) and os.path.isdir(wd)):
        continue
    fetch, write = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        wv, nw = write.get(k, (0.0, 0))
        res[k] = {"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": wv, "hbm_bytes_raw": (f + wv) * 1024,
                  "hbm_bytes_fetch_doubled": (2 * f + wv) * 1024, "launches": max(nf, nw)}
    out[w] = res
out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --workload W` (tools/profile_round.sh); "
               "KB units, means over the launches of a kernel (warm-up, timed and kernel-only loops alike); MI355X_MICROARCH HBM section: "
               "FETCH_SIZE counts half of wide coalesced reads on gfx950 (doubled variant given), other access widths uncalibrated; "
               "graphs of configs 2-4 are cache resident")
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
