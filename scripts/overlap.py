"""How much do the launches of one kernel overlap?  python scripts/overlap.py <rocprofv3 out dir> <kernel substring>
Prints, per burst of launches (gaps > 1 ms split bursts): launches, summed duration, wall span, mean concurrency."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f)) if sys.argv[2] in r['Kernel_Name'])
bursts, cur = [], []
for s, e in rows:
    if cur and s - max(x[1] for x in cur) > 1_000_000:
        bursts.append(cur); cur = []
    cur.append((s, e))
if cur: bursts.append(cur)
for b in bursts[:12]:
    tot = sum(e - s for s, e in b); span = max(e for s, e in b) - min(s for s, e in b)
    print(f"{len(b):4d} launches: sum {tot/1e6:7.3f} ms, span {span/1e6:7.3f} ms, concurrency {tot/max(span,1):.2f}")
