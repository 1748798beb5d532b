import csv,glob,sys
import os
f=max(glob.glob("gpurun_out/prof_build_%s/*/*_kernel_trace.csv"%sys.argv[1]), key=os.path.getmtime)   # the newest run
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[(r["Kernel_Name"][:58], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, int(r["Start_Timestamp"])) for r in rows]
idx=[i for i,(n,_,_) in enumerate(names) if "k_ingest" in n]
start=idx[-2]; t0=names[start][2]
for n,d,t in names[start:idx[-1]]:
    print("%8.1f us  +%7.1f  %s"%((t-t0)/1e3,d,n))
