#!/bin/bash
# LDS counters of one kernel family (dev): usage: pmc_lds.sh "<ab_conv_multi spec>" <kernel substring>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_lds; rm -rf $O; mkdir -p $O
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$n -o p -- python3 tools_dev/ab_conv_multi.py "$1" > $O/$n.log 2>&1
  f=$(ls $O/$n/*counter_collection.csv $O/$n/*/*counter_collection.csv 2>/dev/null | head -1)
  python3 - "$f" "$2" <<'PY'
import csv, sys, collections
f, sub = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if sub in r['Kernel_Name']:
        a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in agg.items(): print("%-28s launches %4d  mean %.4g" % (k, n, v / n))
PY
done
