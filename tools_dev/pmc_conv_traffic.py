"""Sum FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports them) over the convolution launches of ONE step of
tools_dev/time_step.py:  pmc_conv_traffic.py <fetch_dir> <write_dir> <out.json>
(collected in separate --pmc passes, single stream: SSD_OVERLAP_HEADS=0)."""
import csv, glob, json, sys, collections
def per_kernel(d, counter):
    acc = collections.OrderedDict()
    rows = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    # steps are delimited by k_adam: keep the last complete step
    adam = [i for i, r in enumerate(rows) if 'k_adam' in r['Kernel_Name']]
    seg = rows[adam[-2] + 1:adam[-1] + 1]
    for r in seg:
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[n] = acc.get(n, 0.0) + float(r['Counter_Value'])
    return acc
fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
conv = lambda n: n.startswith('k_conv') or n.startswith('k_wgrad') or n.startswith('k_igemm')
out = {"unit": "KB as reported by rocprofv3 (FETCH_SIZE counts 128-byte requests as 64 B on gfx950: double it)",
       "conv_fetch_kb": sum(v for k, v in fetch.items() if conv(k)), "conv_write_kb": sum(v for k, v in write.items() if conv(k)),
       "per_kernel_fetch_kb": {k: round(v, 1) for k, v in fetch.items() if conv(k)},
       "per_kernel_write_kb": {k: round(v, 1) for k, v in write.items() if conv(k)}}
out["conv_hbm_bytes_per_step"] = int((2 * out["conv_fetch_kb"] + out["conv_write_kb"]) * 1024)
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps({k: out[k] for k in ("conv_fetch_kb", "conv_write_kb", "conv_hbm_bytes_per_step")}))
