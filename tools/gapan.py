import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
rows=[r for r in rows if r["Kernel_Name"].startswith(("k_xspec13","k_os13","k_scale"))][-12:]
prev=None
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(r["Kernel_Name"][:10], "dur %.1f"%((e-s)/1000), "gap", None if prev is None else (s-prev)/1000)
    prev=e
