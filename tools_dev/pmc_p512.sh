#!/bin/bash
# SQ counters of k_conv3x3_patch32 / k_conv3x3_p512 on block3_conv2's forward, with and without their memory traffic
# (tools_dev/p512_ablate.py: 8 configurations x 51 launches in a fixed order).  Run on the GPU box via gpurun, repo root.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/pmc_p512
rm -rf $D; mkdir -p $D
P512_LAYERS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/a -o p -- python tools_dev/p512_ablate.py > $D/a.log 2>&1
python - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for f in glob.glob("gpurun_out/pmc_p512/a/**/*counter_collection.csv", recursive=True):
    for x in csv.DictReader(open(f)):
        k = x["Kernel_Name"]
        if "patch32" not in k and "p512" not in k: continue
        d = rows.setdefault(int(x["Dispatch_Id"]), {"name": "p512" if "p512" in k else "p32", "ns": int(x["End_Timestamp"]) - int(x["Start_Timestamp"])})
        d[x["Counter_Name"]] = d.get(x["Counter_Name"], 0.0) + float(x["Counter_Value"])
ids = sorted(rows)
# first launch is the untimed warm call of the script, then 8 configurations of 51 launches
seq = ids[1:]
names = ["p32 full", "p32 no weights", "p32 no patch", "p32 no traffic", "p512 full", "p512 no weights", "p512 no patch", "p512 no traffic"]
for i, nm in enumerate(names):
    grp = [rows[j] for j in seq[i * 51:(i + 1) * 51]]
    if not grp: break
    a = collections.defaultdict(float)
    for r in grp:
        for k, v in r.items():
            if k != "name": a[k] += v
    busy = a["SQ_BUSY_CYCLES"] / 32
    print("%-16s %s us %.1f  clock(SQ_BUSY) %.2f GHz  clock(GRBM) %.2f GHz  mfma_busy %.3f  wait_any %.3f wait_inst %.3f active %.3f" % (
        nm, grp[0]["name"], a["ns"] / len(grp) / 1e3, busy / a["ns"], a["GRBM_GUI_ACTIVE"] / 8 / a["ns"],
        a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / busy, a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"],
        a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"]))
PY
