#!/bin/bash
# SQ counters of k_conv_chain (dev): where do its waves spend their cycles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_chain; rm -rf $O; mkdir -p $O
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_IFETCH" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$n -o p -- python3 tools_dev/time_chain.py > $O/$n.log 2>&1
  f=$(ls $O/$n/*counter_collection.csv $O/$n/*/*counter_collection.csv 2>/dev/null | head -1)
  python3 - "$f" k_conv_chain <<'PY'
import csv, sys, collections
f, sub = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if sub in r['Kernel_Name']:
        a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in agg.items(): print("%-28s launches %4d  mean %.4g" % (k, n, v / n))
PY
done
rm -rf $O/*/
