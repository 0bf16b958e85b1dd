#!/usr/bin/env python3
"""As per_rank_kernels.py for the full-EM decomposition (`--only-c4 --c4-solver yee`): the launches between two E updates
of a rank's planes."""
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# decomposed part: em_update_e launches on nk = nzl planes are short; find the first short one
ue=[(i,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for i,r in enumerate(rows) if 'em_update_e_kernel' in r['Kernel_Name']]
mx=max(d for _,d in ue)
mn=min(d for _,d in ue)
short=[i for i,d in ue if d<mx/3] if mx>2*mn else [i for i,_ in ue]   # (--c4-skip-single: ranks only)
i0,i1=short[8],short[-1]
n=len(short)-9
agg=collections.Counter(); tot=0
for r in rows[i0+1:i1+1]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    agg[r['Kernel_Name'].split('(')[0][:70]]+=d; tot+=d
print('EM: per rank and sub-step over %d rank-substeps: %.1f us of kernels'%(n,tot/n))
for k,v in agg.most_common(14): print('   %-72s %8.1f us'%(k,v/n))
big=collections.Counter(); nb=0
for r in rows[:short[0]]:
    pass
