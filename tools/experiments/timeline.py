import csv,glob,sys
d=sys.argv[1]; span=float(sys.argv[2]) if len(sys.argv)>2 else 3.8
k=list(csv.DictReader(open(glob.glob(d+'/*/*kernel_trace.csv')[0])))
c=list(csv.DictReader(open(glob.glob(d+'/*/*memory_copy_trace.csv')[0])))
ev=[]
for r in k:
    name=r['Kernel_Name']
    short='PULL' if 'pull_' in name else 'prepass' if 'prepass' in name else 'FILTER' if 'window_filter' in name else 'combine' if 'combine' in name else name[:20]
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),short+' wg=%d'%(int(r.get('Grid_Size_X','0'))//int(r.get('Workgroup_Size_X','1') or 1))))
for r in c:
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'copy '+('up' if 'HOST_TO' in r['Direction'] else 'DOWN')))
ev.sort()
tend=max(e[1] for e in ev)
last=[e for e in ev if e[0]>tend-span*1e6]
t0=last[0][0]
for s,e,n in last:
    print("%8.3f %8.3f  %6.3f  %s"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,n))
