#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on a GPU box:  bash tools/profile_round.sh <tag>
# Per workload (config2 = BASELINE configs[1], config3 = 1 M subjects, config4 = 4 populations x 100k, config5 = WMDA-scale
# graph x 256 high-ambiguity subjects): kernel-trace stats of `bench.py --workload W`, and FETCH_SIZE / WRITE_SIZE in
# SEPARATE --pmc passes (MI355X_MICROARCH.md, HBM section); SQ counters for config 2.  Summaries land in gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-file --min-seconds 0"
run() {  # name, rocprof options, bench options
  rocprofv3 $2 --output-format csv -d $O/$1 -o r -- python3 $R/bench.py $3 > $O/$1.log 2>&1
}
# the kernel-only loop ALONE (no stream path beside it): the averages bench.py's roofline.avg_launch_ms has to agree with
rocprofv3 --kernel-trace --stats --output-format csv -d $O/config2_kernel_only -o r -- python3 $R/tools/kernel_only.py --workload config2 --runs 300 > $O/config2_kernel_only.log 2>&1
[ "${ONLY_KERNEL_ONLY:-0}" = "1" ] && exit 0
# the bench lines of the same build (full JSON with roofline), unprofiled
for w in config2 config3 config4 config5; do python3 $R/bench.py --workload $w --no-file --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err || true; done
# ... and the driver's own invocation (20 steps per timed region: the pipeline's fill and drain weigh more)
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-file --no-cpu-baseline > $O/bench_config2_steps20.json 2> $O/bench_config2_steps20.err || true
run config2        "--kernel-trace --stats" "--steps 50 --warmup 5 $B"
run config2_fetch  "--pmc FETCH_SIZE --kernel-trace" "--steps 10 --warmup 2 --kernel-steps 10 $B"
run config2_write  "--pmc WRITE_SIZE --kernel-trace" "--steps 10 --warmup 2 --kernel-steps 10 $B"
run config2_sq     "--pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace" "--steps 10 --warmup 2 --kernel-steps 10 $B"
run config3        "--kernel-trace --stats" "--workload config3 --steps 3 --warmup 1 --kernel-steps 10 $B"
run config3_fetch  "--pmc FETCH_SIZE --kernel-trace" "--workload config3 --steps 1 --warmup 1 --kernel-steps 3 $B"
run config3_write  "--pmc WRITE_SIZE --kernel-trace" "--workload config3 --steps 1 --warmup 1 --kernel-steps 3 $B"
run config4        "--kernel-trace --stats" "--workload config4 --steps 3 --warmup 1 --kernel-steps 5 $B"
run config4_fetch  "--pmc FETCH_SIZE --kernel-trace" "--workload config4 --steps 2 --warmup 1 --kernel-steps 3 $B"
run config4_write  "--pmc WRITE_SIZE --kernel-trace" "--workload config4 --steps 2 --warmup 1 --kernel-steps 3 $B"
run config5        "--kernel-trace --stats" "--workload config5 --steps 3 --warmup 2 --kernel-steps 5 $B"
run config5_fetch  "--pmc FETCH_SIZE --kernel-trace" "--workload config5 --steps 2 --warmup 1 --kernel-steps 3 $B"
run config5_write  "--pmc WRITE_SIZE --kernel-trace" "--workload config5 --steps 2 --warmup 1 --kernel-steps 3 $B"
cd $R
python tools/pmc_traffic.py $O $O/pmc_traffic.json > /dev/null
for w in config2 config3 config4 config5; do echo "== $w"; head -n 9 $O/$w/r_kernel_stats.csv | cut -c1-120; done
