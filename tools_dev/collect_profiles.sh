#!/bin/bash
# Everything under profiles/ for one round, at one commit (run on the GPU box through gpurun, from the repo root):
#   <prefix>_bench.json                the bench line (python bench.py, default arguments)
#   <prefix>_bench_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command (kernel timing sections off)
#   <prefix>_step_timeline.txt         every launch of one step of that trace: start / end / queue / kernel / grid
#   <prefix>_{conv_pmc,mfma_util,match_pmc}.json   PMC passes (tools_dev/collect_pmc.sh)
# usage: tools_dev/collect_profiles.sh <commit-hash> <prefix, e.g. r03>
set -e
COMMIT=$1; P=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_$P
rm -rf $O; mkdir -p $O
python3 bench.py > $O/${P}_bench.json 2> $O/bench.err
echo "bench done" ; head -c 300 $O/${P}_bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/${P}_bench_profiled.json 2> $O/trace.err
cp $(ls $O/trace/*kernel_stats.csv $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1) $O/${P}_bench_kernel_stats.csv
python3 tools_dev/step_timeline.py $O/trace 0 3 > $O/${P}_step_timeline.txt
rm -f $O/trace/*.db $O/trace/*/*.db $O/trace/*kernel_trace.csv $O/trace/*/*kernel_trace.csv
bash tools_dev/collect_pmc.sh $COMMIT $O/$P > $O/pmc.log 2>&1
tail -2 $O/pmc.log
rm -rf gpurun_out/pmc
ls -la $O
