#!/bin/bash
# The sharded file -> file path at 1 / 2 / 4 ranks on ONE GPU box (ranks > devices: every rank on the box's one GPU, gloo control
# plane) -- a rehearsal of `bench.py --gpus N --sharded`, not a scaling curve: what it shows is whether the multi-rank
# machinery (chunk counter, size exchange, placed writes) costs anything next to one rank doing the same file.
#   bash tools/rehearse_sharded.sh <out dir>      -> <out>/sharded_N.json, <out>/sharded_rehearsal.json
O=${1:-gpurun_out/rehearse}
mkdir -p $O
for n in 1 2 4; do
  timeout -k 10 400 python bench.py --gpus $n --workload ${WORKLOAD:-config3} --sharded --steps ${STEPS:-5} --warmup 2 > $O/sharded_$n.json 2> $O/sharded_$n.err
  echo "ranks $n rc=$?"
done
python - <<PY
import json
out = {}
for n in (1, 2, 4):
    try:
        d = json.loads([l for l in open("$O/sharded_%d.json" % n) if l.startswith("{")][-1])
        out[str(n)] = {k: d[k] for k in ("value", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "n_gpus")}
        out[str(n)]["ranks"] = d["config"].get("ranks")
        out[str(n)]["output_bytes"] = d["config"].get("output_bytes")
    except Exception as e:
        out[str(n)] = {"error": str(e)}
json.dump(out, open("$O/sharded_rehearsal.json", "w"), indent=1)
print(json.dumps(out))
PY
