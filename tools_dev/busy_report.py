"""GPU busy fraction from a rocprofv3 kernel trace: busy_report.py <dir>  (union of kernel intervals vs wall, per step)"""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv') + glob.glob(sys.argv[1] + '/*_kernel_trace.csv'))[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
# a step starts with the first launch of the anchor matching (k_match_rows)
first = [i for i, r in enumerate(rows) if 'k_match_rows' in r[2]]
first = [i for j, i in enumerate(first) if j == 0 or i - first[j - 1] > 20]
for a, b in zip(first[-4:-1], first[-3:]):
    seg = rows[a:b]
    t0, t1 = seg[0][0], seg[-1][1]
    busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, _ in seg[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e - s for s, e, _ in seg)
    gaps = sorted(((seg[i + 1][0] - max(x[1] for x in seg[:i + 1][-8:]), seg[i + 1][2][:50]) for i in range(len(seg) - 1)), reverse=True)[:5]
    print("step wall %.2f ms  busy(union) %.2f ms  sum %.2f ms  launches %d" % ((t1 - t0) / 1e6, busy / 1e6, tot / 1e6, len(seg)))
    print("   largest gaps (us):", [(round(g / 1e3, 1), n) for g, n in gaps])
