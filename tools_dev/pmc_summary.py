"""mean counter value per launch for kernels whose name contains a filter: pmc_summary.py <dir> <filter>"""
import csv, glob, sys, collections
d, filt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if filt in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print("%-36s n=%3d mean %.4g" % (k, len(v), sum(v) / len(v)))
