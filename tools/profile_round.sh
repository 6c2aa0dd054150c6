#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on a GPU box:  bash tools/profile_round.sh <tag>
# (kernel-trace stats for config 2 and the mixed workload; FETCH_SIZE / WRITE_SIZE / SQ counters in separate passes)
set -e
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/full -o r -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/full.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mixed -o r -- python3 $R/bench.py --workload mixed --steps 10 --warmup 2 --no-cpu-baseline > $O/mixed.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/sq.log 2>&1
cd $R
python tools/pmc_traffic.py $O/fetch/r_counter_collection.csv $O/write/r_counter_collection.csv $O/pmc_traffic.json > /dev/null
head -n 8 $O/full/r_kernel_stats.csv $O/mixed/r_kernel_stats.csv
