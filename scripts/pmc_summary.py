"""Mean per-dispatch PMC values of one kernel from rocprofv3 counter_collection csv files:
python scripts/pmc_summary.py <kernel-substring> <dir> [<dir> ...]"""
import csv
import glob
import sys
from collections import defaultdict

kern = sys.argv[1]
acc = defaultdict(list)
for d in sys.argv[2:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
print("counter,dispatches,mean_per_dispatch")
for k, v in acc.items():
    print(f"{k},{len(v)},{sum(v) / len(v):.3f}")
