"""Per-kernel timeline of the last query batch in a rocprofv3 kernel trace:
python scripts/timeline.py <dir-with-*_kernel_trace.csv>"""
import csv
import glob
import sys

f = (glob.glob(sys.argv[1] + '/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if ('build_tables<true, 4>' in r['Kernel_Name'] or 'bound_tables<' in r['Kernel_Name'])]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
prev = t0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].replace('gulon::', '').replace('(anonymous namespace)::', '').replace('void ', '')[:44]
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} gap {(s - prev) / 1e3:6.1f}  {nm:44s} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}")
    prev = e
print(f"step total {(prev - t0) / 1e3:.1f} us")
