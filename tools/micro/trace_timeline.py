import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
print(rows[0].keys())
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:40],r.get('Queue_Id'),r.get('Stream_Id')) for r in rows]
ks.sort()
t0=ks[len(ks)*2//3][0]
for s,e,n,q,st in ks[len(ks)*2//3:len(ks)*2//3+150]:
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} q{q} s{st} {n}")
