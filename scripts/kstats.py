"""Summarize a rocprofv3 kernel_stats.csv: python scripts/kstats.py <dir> [filter]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for r in csv.DictReader(open(f)):
    if flt and flt not in r['Name']:
        continue
    n = r['Name'].split('(')[0][-44:]
    print(f"{n:46s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} tot_ms={float(r['TotalDurationNs'])/1e6:9.1f} pct={r['Percentage']}")
