import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "cv_tc_kernel" in n or "cv_ss_kernel" in n or "cv_tn_kernel" in n:
        key=n[n.index("<"):n.index(">")+1]+" grid="+r["Grid_Size_X"] if "Grid_Size_X" in r else n[:60]
        d[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000)
for k,v in sorted(d.items()):
    v=sorted(v); print(f"{k:45s} n={len(v):4d} min {v[0]:6.1f} med {v[len(v)//2]:6.1f}")
