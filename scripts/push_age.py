import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'push_tiles_kernel<float, true' in r['Kernel_Name'] and r['Kernel_Name'].count('false, true>')]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
age=None; by={}
for r in rows:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    if 'true, true, false, true' in r['Kernel_Name']: age=0; by.setdefault('R',[]).append(d)
    elif age is not None: age+=1; by.setdefault(age,[]).append(d)
print({k:round(sum(v)/len(v),3) for k,v in by.items()})
