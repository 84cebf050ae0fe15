import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take the last ~10 steps: find adam_kernel occurrences as step markers
idx=[i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
# each step has 2 adam launches (two groups?) -> use every 2nd
marks=idx[1::2]
a,b=marks[-8],marks[-2]
seg=rows[a+1:b+1]
steps=6
t0=int(seg[0]['Start_Timestamp']); t1=int(seg[-1]['End_Timestamp'])
busy=0; cur_end=t0; gaps=[]
for r in seg:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if s>cur_end:
        gaps.append((s-cur_end, r['Kernel_Name'][:50]))
    if e>cur_end:
        busy+= e-max(s,cur_end); cur_end=e
print("wall/step %.1f us, union-busy/step %.1f us, idle/step %.1f us, launches/step %.1f"%((t1-t0)/steps/1e3, busy/steps/1e3, (t1-t0-busy)/steps/1e3, len(seg)/steps))
gaps.sort(reverse=True)
agg=collections.defaultdict(lambda:[0,0])
for g,n in gaps:
    agg[n][0]+=g; agg[n][1]+=1
for n,(g,c) in sorted(agg.items(), key=lambda kv:-kv[1][0])[:10]:
    print("  idle before %-50s %.1f us/step (%d per step)"%(n, g/steps/1e3, c/steps))
names=collections.Counter(r['Kernel_Name'][:60] for r in seg)
for n,c in names.items():
    if 'ccl' in n.lower() or 'copyBuffer' in n or 'fill' in n.lower(): print("  ", n, c/steps)
