"""Per-kernel statistics from a rocprofv3 results database:  python scripts/kstats_db.py <dir> [name filter]"""
import glob
import re
import sqlite3
import subprocess
import sys

db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
sym = [t for t in tabs if 'info_kernel_symbol' in t][0]
rows = list(c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from {kd} d "
                      f"join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 4 desc"))
names = subprocess.run(['c++filt'], input="\n".join(r[0].replace('.kd', '') for r in rows), capture_output=True,
                       text=True).stdout.split("\n")
for (name, n, avg, tot), dn in zip(rows, names):
    if flt and not re.search(flt, dn):
        continue
    m = re.search(r'([A-Za-z_]\w*(?:<[^()]*>)?)\s*\(', dn)
    print(f"{(m.group(1) if m else dn)[-56:]:56s} calls={n:5d} avg_us={avg:10.1f} tot_ms={tot:9.1f}")
