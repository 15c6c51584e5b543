import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the second half (steady state)
rows = rows[len(rows)//2:]
busy = 0; gap = 0; prev=None
byname = collections.Counter(); cnt = collections.Counter()
gapafter = collections.Counter()
for r in rows:
    s,e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("rgx::(anonymous namespace)::","").replace("void ","").split("(")[0][:40]
    busy += e-s; byname[nm] += e-s; cnt[nm]+=1
    if prev is not None and s>prev: gap += s-prev; gapafter[nm] += s-prev
    prev = max(prev or 0, e)
tot = busy+gap
print("span %.1f ms busy %.1f ms gaps %.1f ms (%.1f %%)" % (tot/1e6, busy/1e6, gap/1e6, 100*gap/tot))
for nm,v in byname.most_common(14):
    print("  %-40s n=%5d busy %8.2f ms  gap-in-front %7.2f ms" % (nm, cnt[nm], v/1e6, gapafter[nm]/1e6))
