import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "rnvp_few" in r["Kernel_Name"]]
seq=[(r["Kernel_Name"][10:33], int(r["End_Timestamp"])-int(r["Start_Timestamp"]), r["VGPR_Count"], r["Scratch_Size"]) for r in sorted(rows,key=lambda r:int(r["Start_Timestamp"]))]
prev=None; grp=[]
for s in seq+[("end",0,0,0)]:
    if s[0]!=prev and grp:
        v=sorted(g[1] for g in grp); print(prev, len(v), "median", v[len(v)//2], "min", v[0], "vgpr", grp[0][2], "scratch", grp[0][3])
        grp=[]
    prev=s[0]; grp.append(s)
