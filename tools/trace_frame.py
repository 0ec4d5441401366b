import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# take the last ~3000 kernels (steady-state graph replays); find period by voxelize kernel "vox_insert"
ks=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in rows]
idx=[i for i,k in enumerate(ks) if k[2].startswith("vox_insert") or "vox_insert" in k[2]]
a,b=idx[-12],idx[-11]
fr=ks[a:b]
t0=fr[0][0]
print("frame kernels",len(fr),"period us",(ks[b][0]-t0)/1000)
busy=0; last_end=t0
# union of intervals
iv=sorted((s,e) for s,e,_ in fr)
cur_s,cur_e=iv[0]
tot=0
for s,e in iv[1:]:
    if s>cur_e: tot+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
tot+=cur_e-cur_s
print("union busy us",tot/1000)
for s,e,n in fr:
    print(f"{(s-t0)/1000:8.1f} {(e-s)/1000:7.1f}  {n[:70]}")
