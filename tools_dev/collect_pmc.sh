#!/bin/bash
# Collect the PMC summaries bench.py reports next to its rooflines (run on the GPU box through gpurun, from the repo root):
#   conv:  FETCH_SIZE and WRITE_SIZE of every launch of one train step (single stream), separate passes
#   mfma:  SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES per convolution kernel
#   match: FETCH_SIZE / WRITE_SIZE of the matching kernels at batch 64
# Counters are collected with --kernel-trace only (never with the trace domains gpurun refuses next to --pmc).
# usage: tools_dev/collect_pmc.sh <commit-hash> <out-prefix, e.g. profiles/r03>
set -e
COMMIT=$1; OUT=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/pmc
rm -rf $D; mkdir -p $D
export SSD_OVERLAP_HEADS=0
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/conv_fetch -o p -- python tools_dev/time_step.py 64 > $D/conv_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/conv_write -o p -- python tools_dev/time_step.py 64 > $D/conv_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $D/mfma -o p -- python tools_dev/time_step.py 64 > $D/mfma.log 2>&1
unset SSD_OVERLAP_HEADS
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/match_fetch -o p -- python tools_dev/time_match.py 64:mix > $D/match_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/match_write -o p -- python tools_dev/time_match.py 64:mix > $D/match_write.log 2>&1
python tools_dev/pmc_report.py $D $COMMIT $OUT
