import csv, glob, collections, sys
d=sys.argv[1]; nseg=int(sys.argv[2]); names=sys.argv[3:]
f=glob.glob(d+'/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
dd=collections.defaultdict(list)
for r in rows:
    dd[r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:28]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in dd.items():
    if len(v) >= nseg*10 and len(v)%nseg==0:
        n=len(v)//nseg
        print('%-28s'%k, '  '.join('%s: med %.1f min %.1f'%(names[i] if i<len(names) else i, sorted(v[i*n:(i+1)*n])[n//2], min(v[i*n:(i+1)*n])) for i in range(nseg)))
