#!/usr/bin/env python3
"""profiles/<round>_pmc_traffic.json from two rocprofv3 counter_collection CSVs (one --pmc FETCH_SIZE pass,
one --pmc WRITE_SIZE pass, as MI355X_MICROARCH's HBM section prescribes):
    python tools/pmc_traffic.py fetch.csv write.csv out.json"""
import collections, csv, json, sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = {"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "hbm_bytes_raw": (f + w) * 1024,
              "hbm_bytes_fetch_doubled": (2 * f + w) * 1024}
out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 10 --warmup 2 "
               "--no-cpu-baseline` (config 2, 10k subjects); KB units; MI355X_MICROARCH HBM section: FETCH_SIZE counts half of wide "
               "coalesced reads on gfx950 (doubled variant given), other access widths uncalibrated; the graph is cache resident")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
