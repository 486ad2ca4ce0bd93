#!/bin/bash
# SQ counters of k_conv3x3_patch32 vs k_conv3x3_p512 on one layer (run on the GPU box via gpurun, repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/pmc_p512
rm -rf $D; mkdir -p $D
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --kernel-trace --output-format csv -d $D/a -o p -- python tools_dev/ab_conv_multi.py "fwd 64 75 256 256 3 1 SSD_CONV_P512 0,1" "fwd 64 38 512 512 3 1 SSD_CONV_P512 0,1" > $D/a.log 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
for f in glob.glob("gpurun_out/pmc_p512/a/**/*counter_collection.csv", recursive=True):
    for x in csv.DictReader(open(f)):
        k = x["Kernel_Name"]
        if "patch32" not in k and "p512" not in k: continue
        key = (k.split("(")[0][-40:], x["Grid_Size"])
        acc[key][x["Counter_Name"]] += float(x["Counter_Value"])
        if x["Dispatch_Id"] not in seen:
            seen.add(x["Dispatch_Id"]); n[key] += 1; acc[key]["ns"] += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
for k, a in acc.items():
    c = n[k]
    busy = a["SQ_BUSY_CYCLES"] / 32
    print(k, "launches", c, "us %.1f" % (a["ns"] / c / 1e3), "clock %.2f GHz" % (busy / a["ns"]),
          "mfma_busy %.3f" % (a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / busy),
          "wait_any %.3f wait_inst %.3f wait_lds %.3f active %.3f (of wave cycles)" % tuple(a[q] / a["SQ_WAVE_CYCLES"] for q in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY")),
          "wave_cycles/SIMD/launch %.0f" % (a["SQ_WAVE_CYCLES"] * 4 / 1024 / c), "mfma/launch %.0f" % (a["SQ_INSTS_MFMA"] / c))
PY
