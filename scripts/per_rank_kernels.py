#!/usr/bin/env python3
"""Kernel time per rank and sub-step of the decomposed part of `bench.py --only-c4` from rocprofv3's kernel trace
(scripts/prof_bench.sh <dir> --only-c4; python scripts/per_rank_kernels.py <dir>/k_kernel_trace.csv).  The ranks share
one GPU and run one after the other: the launches between two x-inverse passes of the decomposed solve are one rank's
sub-step.  __amd_rocclr_copyBuffer is the in-process transport (RCCL's job on real links)."""
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
dur=lambda r:(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
xi=[(i,dur(r)) for i,r in enumerate(rows) if 'fft_x_inverse_kernel' in r['Kernel_Name']]
mx=max(d for _,d in xi)
mn=min(d for _,d in xi)
small=[i for i,d in xi if d<mx/3] if mx>2*mn else [i for i,_ in xi]   # (--c4-skip-single: ranks only)
i0,i1=small[16],small[-1]
n=len(small)-17
agg=collections.Counter(); tot=0
for r in rows[i0+1:i1+1]:
    d=dur(r); agg[r['Kernel_Name'].split('(')[0][:70]]+=d; tot+=d
print('per rank and sub-step over %d rank-substeps: %.1f us of kernels (%.1f without the in-process copies)'%(n,tot/n,(tot-agg.get('__amd_rocclr_copyBuffer',0))/n))
for k,v in agg.most_common(22): print('   %-72s %8.1f us'%(k,v/n))
