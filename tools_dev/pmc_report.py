"""Summarise the rocprofv3 --pmc passes of tools_dev/collect_pmc.sh into the small JSON files bench.py reads:
pmc_report.py <pmc dir> <commit> <out prefix>   ->  <prefix>_conv_pmc.json, <prefix>_mfma_util.json, <prefix>_match_pmc.json"""
import collections, csv, glob, json, sys

d, commit, out = sys.argv[1:4]


def rows(sub, counters=None):
    r = []
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, sub), recursive=True):
        r += [x for x in csv.DictReader(open(f)) if counters is None or x["Counter_Name"] in counters]
    r.sort(key=lambda x: int(x["Dispatch_Id"]))
    return r


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def last_step(rs):
    """Rows of the last complete train step of tools_dev/time_step.py (a step ends with its k_adam dispatch)."""
    adam = sorted({int(x["Dispatch_Id"]) for x in rs if "k_adam" in x["Kernel_Name"]})
    return [x for x in rs if adam[-2] < int(x["Dispatch_Id"]) <= adam[-1]]


conv = lambda n: n.startswith(("k_conv", "k_wgrad", "k_igemm", "k_hz_", "k_hw_"))      # incl. the sparse head backward
DOMINANT = "k_conv3x3_wgrad_patch<16, 2>"                                             # bench.py's roofline.kernel

# ---- convolution traffic of one step
fetch, write = collections.OrderedDict(), collections.OrderedDict()
for x in last_step(rows("conv_fetch", {"FETCH_SIZE"})):
    fetch[short(x["Kernel_Name"])] = fetch.get(short(x["Kernel_Name"]), 0.0) + float(x["Counter_Value"])
for x in last_step(rows("conv_write", {"WRITE_SIZE"})):
    write[short(x["Kernel_Name"])] = write.get(short(x["Kernel_Name"]), 0.0) + float(x["Counter_Value"])
c = {"commit": commit, "workload": "one batch-64 train step, single stream (tools_dev/time_step.py under rocprofv3 --pmc, separate passes)",
     "unit": "KB as reported by rocprofv3 (FETCH_SIZE counts 128-byte requests as 64 B on gfx950: doubled below, MI355X_MICROARCH.md)",
     "conv_fetch_kb": sum(v for k, v in fetch.items() if conv(k)), "conv_write_kb": sum(v for k, v in write.items() if conv(k)),
     "per_kernel_fetch_kb": {k: round(v, 1) for k, v in fetch.items() if conv(k)},
     "per_kernel_write_kb": {k: round(v, 1) for k, v in write.items() if conv(k)}}
c["conv_hbm_bytes_per_step"] = int((2 * c["conv_fetch_kb"] + c["conv_write_kb"]) * 1024)
n_dom = len({x["Dispatch_Id"] for x in last_step(rows("conv_fetch", {"FETCH_SIZE"})) if short(x["Kernel_Name"]) == DOMINANT})
if n_dom:
    c["dominant_kernel"] = DOMINANT
    c["dominant_kernel_launches_per_step"] = n_dom
    c["dominant_kernel_hbm_bytes_per_step"] = int((2 * fetch.get(DOMINANT, 0.0) + write.get(DOMINANT, 0.0)) * 1024)
json.dump(c, open(out + "_conv_pmc.json", "w"), indent=1)

# ---- MFMA utilisation per convolution kernel (one step)
acc = collections.OrderedDict()
seen = set()
for x in last_step(rows("mfma")):
    k = short(x["Kernel_Name"])
    if conv(k):
        a = acc.setdefault(k, collections.defaultdict(float))
        a[x["Counter_Name"]] += float(x["Counter_Value"])
        if x["Dispatch_Id"] not in seen:
            seen.add(x["Dispatch_Id"])
            a["ns"] += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
SIMDS, SES, CLK = 1024, 32, 2.4                      # MI355X: 256 CUs x 4 SIMDs, 32 shader engines, 2.4 GHz maximum clock
m = {"commit": commit, "workload": c["workload"],
     "note": "per convolution kernel, summed over its launches of one step.  SQ_VALU_MFMA_BUSY_CYCLES is per SIMD (summed over "
             "1024 SIMDs), SQ_BUSY_CYCLES per shader engine (32): mfma_busy = (MFMA_BUSY / 1024) / (SQ_BUSY / 32), the fraction of the "
             "kernel's busy time its matrix pipes were occupied; mfma_busy_vs_2.4GHz = MFMA_BUSY / (1024 x duration x 2.4 GHz), a "
             "lower bound that ignores the clock the chip actually held (profiled passes run 1.9-2.3 GHz).  valu_per_mfma = "
             "SQ_INSTS_VALU / SQ_INSTS_MFMA; wait_inst_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.", "kernels": {}}
tot_b = tot_m = tot_ns = 0.0
for k, v in acc.items():
    busy, mf, ns = v.get("SQ_BUSY_CYCLES", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("ns", 0.0)
    if v.get("SQ_INSTS_MFMA"):
        tot_b += busy; tot_m += mf; tot_ns += ns
    m["kernels"][k] = {"mfma_busy": round((mf / SIMDS) / (busy / SES), 4) if busy else None,
                       "mfma_busy_vs_2.4GHz": round(mf / (SIMDS * ns * CLK), 4) if ns else None,
                       "valu_per_mfma": round(v.get("SQ_INSTS_VALU", 0.0) / v["SQ_INSTS_MFMA"], 3) if v.get("SQ_INSTS_MFMA") else None,
                       "wait_inst_frac": round(v.get("SQ_WAIT_INST_ANY", 0.0) / v["SQ_WAVE_CYCLES"], 4) if v.get("SQ_WAVE_CYCLES") else None,
                       "us_per_step_under_pmc": round(ns / 1e3, 1), "SQ_BUSY_CYCLES": busy, "SQ_VALU_MFMA_BUSY_CYCLES": mf}
m["all_mfma_kernels_mfma_busy"] = round((tot_m / SIMDS) / (tot_b / SES), 4) if tot_b else None
m["all_mfma_kernels_mfma_busy_vs_2.4GHz"] = round(tot_m / (SIMDS * tot_ns * CLK), 4) if tot_ns else None
json.dump(m, open(out + "_mfma_util.json", "w"), indent=1)

# ---- matching kernels (mean per launch)
mt = {"commit": commit, "workload": "ssd_match_encode, batch 64, COCO-shaped boxes (tools_dev/time_match.py 64:mix)", "unit": "KB per launch (FETCH_SIZE as reported: double it)"}
for sub, ctr in (("match_fetch", "FETCH_SIZE"), ("match_write", "WRITE_SIZE")):
    per = collections.defaultdict(list)
    for x in rows(sub, {ctr}):
        k = short(x["Kernel_Name"])
        if k.startswith("k_match"):
            per[k].append(float(x["Counter_Value"]))
    for k, v in per.items():
        mt["%s.%s" % (k, ctr)] = round(sum(v) / len(v), 3)
json.dump(mt, open(out + "_match_pmc.json", "w"), indent=1)
print(json.dumps({"conv_hbm_bytes_per_step": c["conv_hbm_bytes_per_step"], "mfma_busy": m["all_mfma_kernels_mfma_busy"],
                  "mfma_busy_vs_2.4GHz": m["all_mfma_kernels_mfma_busy_vs_2.4GHz"],
                  "match": {k: v for k, v in mt.items() if "." in k}}))
