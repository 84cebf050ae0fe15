import collections, csv, re, sys
acc=collections.defaultdict(lambda: collections.defaultdict(float)); ln=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); n=re.sub(r"^void ","",n); n=re.sub(r"_ZN12_GLOBAL__N_1\d+","",n)[:48]
    acc[n][r["Counter_Name"]]+=float(r["Counter_Value"]); ln[n].add(r["Dispatch_Id"])
names=sorted({c for v in acc.values() for c in v})
print(names)
rows=sorted(acc.items(), key=lambda kv:-kv[1].get("SQ_BUSY_CYCLES",0))[:16]
for n,c in rows:
    w=c.get("SQ_WAVE_CYCLES",1) or 1; b=c.get("SQ_BUSY_CYCLES",1) or 1
    print("%-48s"%n, " ".join("%s=%.3f"%(k.replace("SQ_","")[:18], c[k]/w) for k in names if k not in("SQ_BUSY_CYCLES","SQ_WAVE_CYCLES")), "wave/busy=%.1f"%(w/b))
