"""Timeline of the last step in a rocprofv3 kernel trace (bench.py or tools_dev/time_step.py): start, end, duration, queue,
kernel, grid -- from the step's k_image_prep launch on.  usage: step_timeline.py <dir> [min start us]"""
import csv, glob, re, sys
d = sys.argv[1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'k_image_prep' in r['Kernel_Name']]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2          # which step from the end
rows = rows[starts[-k]:starts[-k + 1] if k > 1 else len(rows)]
t0 = int(rows[0]['Start_Timestamp'])
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', n)[:36]
for r in rows:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    if s >= lo:
        print('%8.1f %8.1f %7.1f  q%-3s %-38s grid %s' % (s, e, e - s, r.get('Queue_Id', '?'), short(r['Kernel_Name']), r['Grid_Size_X']))
