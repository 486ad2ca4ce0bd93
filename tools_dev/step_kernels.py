"""Per-kernel time of ONE step from a rocprofv3 kernel trace of tools_dev/time_step.py (or bench.py): the last `nsteps`
steps' launches, grouped by kernel name (+ grid when asked), microseconds per step.  usage: step_kernels.py <dir> <steps in the trace> [steps to average]"""
import csv, glob, sys, collections, re
d, total_steps = sys.argv[1], int(sys.argv[2])
use = int(sys.argv[3]) if len(sys.argv) > 3 else 1
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# a step starts at its k_image_prep launch
starts = [i for i, r in enumerate(rows) if 'k_image_prep' in r['Kernel_Name']]
assert len(starts) >= use, "no k_image_prep launches in the trace"
rows = rows[starts[-use]:]
agg = collections.OrderedDict()
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', n)
for r in rows:
    k = short(r['Kernel_Name'])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
print("launches/step %d, sum of kernel time %.1f us/step, span %.1f us/step" % (len(rows) // use, tot / use, (t1 - t0) / 1e3 / use))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-58s n=%3d  %9.1f us  %5.1f %%" % (k[:58], v[0] // use, v[1] / use, 100 * v[1] / tot))
