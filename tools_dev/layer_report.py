"""Per-launch conv timing from a rocprof kernel trace of tools_dev/time_step.py (one fwd+bwd)."""
import csv, glob, sys
sys.path.insert(0, '.')
from importlib import import_module
d = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'k_conv' in r['Kernel_Name'] or 'k_wgrad' in r['Kernel_Name']]
# engine plan
import ssd_object_detection_amd.engine as E
import ssd_object_detection_amd.ops as ops
nodes = []; s = 300
fm = []
for kind, cin, cout, k, stride, mode, feat in E.SSD300_TRUNK:
    ho = ops.same_pad(s, k, stride)[0] if mode == 'same' else ops.valid_out(s, k, stride)
    nodes.append((kind, cin, cout, k, s, ho)); s = ho
    if feat: fm.append((ho, cout))
convs = [n for n in nodes if n[0] == 'conv']
heads = [(h, c, n * 85) for (h, c), n in zip(fm, E.SSD300_NUM_PRIORS)]
fwd = [("fwd c%d %dx%d %d->%d k%d" % (i, n[5], n[5], n[1], n[2], n[3]), 2.0 * B * n[5] * n[5] * n[2] * n[3] * n[3] * n[1]) for i, n in enumerate(convs)]
fwd += [("fwd head%d %dx%d %d->%d" % (i, h, h, c, no), 2.0 * B * h * h * no * 9 * c) for i, (h, c, no) in enumerate(heads)]
per_step = len([r for r in rows]) // 3
# find first step's rows: take the last third (steady state)
rows = rows[-per_step:]
names = [r['Kernel_Name'] for r in rows]
def dur(r): return (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
i = 0
print("---- forward")
tot = 0
for name, fl in fwd:
    r = rows[i]; i += 1
    print("%-34s %8.1f us  %7.1f TF/s  grid %s" % (name, dur(r), fl / dur(r) / 1e6, r['Grid_Size_X']+'x'+r['Grid_Size_Y']+'x'+r['Grid_Size_Z']))
    tot += dur(r)
print("forward conv total %.2f ms" % (tot / 1e3))
print("---- backward (in launch order)")
bt = {}
for r in rows[i:]:
    key = r['Kernel_Name'].split('(')[0][-40:]
    bt.setdefault(key, 0.0); bt[key] += dur(r)
for k, v in bt.items(): print("%-44s %.2f ms" % (k, v / 1e3))
# backward detail: per layer "wgrad [reduce] dgrad [finalize]" in engine.backward order
j = i
print("---- backward detail")
def kind(r):
    n = r['Kernel_Name']
    if 'reduce' in n: return 'reduce'
    if 'wgrad' in n: return 'wgrad'
    if 'finalize' in n: return 'finalize'
    return 'dgrad'
def grid(r): return r['Grid_Size_X'] + 'x' + r['Grid_Size_Y'] + 'x' + r['Grid_Size_Z']
def short(r): return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
def layer(label, fl, has_dgrad):
    global j
    w = rows[j]; j += 1
    assert kind(w) == 'wgrad', (label, w['Kernel_Name'])
    tw = dur(w)
    while j < len(rows) and kind(rows[j]) == 'reduce':
        tw += dur(rows[j]); j += 1
    line = "%-30s wgrad %7.1f us %7.1f TF/s %-26s %-16s" % (label, tw, fl / tw / 1e6, short(w)[:26], grid(w))
    t = tw
    if has_dgrad:
        dg = rows[j]; j += 1
        assert kind(dg) == 'dgrad', (label, dg['Kernel_Name'])
        td = dur(dg)
        while j < len(rows) and kind(rows[j]) == 'finalize':
            td += dur(rows[j]); j += 1
        line += " | dgrad %7.1f us %7.1f TF/s %-30s %s" % (td, fl / td / 1e6, short(dg)[:30], grid(dg))
        t += td
    print(line)
    return t
tb = 0
for lvl, (h, c, no) in enumerate(heads):
    tb += layer("head%d %dx%d %d->%d" % (lvl, h, h, c, no), 2.0 * B * h * h * no * 9 * c, True)
for ci in range(len(convs) - 1, -1, -1):
    n = convs[ci]
    fl = 2.0 * B * n[5] * n[5] * n[2] * n[3] * n[3] * n[1]
    tb += layer("c%-2d %dx%d %d->%d k%d" % (ci, n[5], n[5], n[1], n[2], n[3]), fl, ci > 0)
print("backward conv total %.2f ms" % (tb / 1e3))
