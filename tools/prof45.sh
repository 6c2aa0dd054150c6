set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_x
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in config4 config5; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o r -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --kernel-steps 5 --no-cpu-baseline --no-file > $O/$w.log 2>&1
echo "== $w"; head -n 9 $O/$w/r_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-150
done
