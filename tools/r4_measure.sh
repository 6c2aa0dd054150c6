#!/bin/bash
# round-4 measurement pass on a gpurun box: full GPU tests, then config-4 / config-2 bench lines (A/B switches in the environment)
set -o pipefail
out=gpurun_out/${1:-r4m}
mkdir -p $out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1; echo "rc=$?" >> $out/gputest.log
  tail -3 $out/gputest.log
fi
B="--steps 5 --warmup 2 --no-file --no-cpu-baseline --min-seconds 0 --kernel-steps 10"
GRIM_DEBUG_CLASSES=1 timeout -k 10 300 python bench.py --workload config4 $B > $out/c4.json 2> $out/c4.err; echo "c4 rc=$?"
GRIM_NO_MID=1 timeout -k 10 300 python bench.py --workload config4 $B > $out/c4_nomid.json 2> $out/c4_nomid.err; echo "c4 nomid rc=$?"
python - <<PY
import json
for f in ("c4","c4_nomid"):
    try:
        d=json.loads([l for l in open("$out/%s.json"%f) if l.startswith("{")][-1])
        print(f, "ms_per_step %.3f"%d["ms_per_step"], {k:round(v,3) for k,v in d["roofline"]["kernel_ms"].items()}, "sum %.3f"%d["kernel_only"]["kernel_ms_per_step"])
    except Exception as e:
        print(f, "failed", e)
PY
grep "grim classes" $out/c4.err | tail -2
