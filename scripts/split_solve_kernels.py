#!/usr/bin/env python3
"""Mean duration of the solve's kernels in a trace of `bench.py --only-c4`, the one-handle launches (512^3, 'big') and
the per-rank launches ('small') apart."""
import csv,sys,collections
f=sys.argv[1]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if any(t in n for t in ('fft_x','fft_columns','gradient')):
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        key=n.split('(')[0][:60]
        agg[key].append(d)
tot_big=tot_small=0
for k,v in sorted(agg.items()):
    v.sort()
    # split at geometric midpoint between min and max
    thr=(v[0]*v[-1])**0.5
    small=[x for x in v if x<thr]; big=[x for x in v if x>=thr]
    if v[-1]/v[0]<2: small=v; big=[]
    ms=sum(small)/len(small) if small else 0; mb=sum(big)/len(big) if big else 0
    print('%-62s small n=%4d %7.1f us   big n=%3d %7.1f us'%(k,len(small),ms,len(big),mb))
    tot_small+=ms; tot_big+=mb
print('sum small %.1f us, sum big %.1f us'%(tot_small,tot_big))
