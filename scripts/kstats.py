"""Summarize a rocprofv3 kernel_stats.csv: python scripts/kstats.py <dir> [filter]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ''


def short(name):
    """gulon::kernel<template args> without the argument list."""
    m = re.search(r'(?:gulon::)?(?:\(anonymous namespace\)::)?([A-Za-z_]\w*(?:<[^()]*>)?)\s*\(', name)
    return (m.group(1) if m else name.split('(')[0])[-60:]


for r in csv.DictReader(open(f)):
    if flt and flt not in r['Name']:
        continue
    print(f"{short(r['Name']):56s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} "
          f"tot_ms={float(r['TotalDurationNs'])/1e6:9.1f} pct={r['Percentage']}")
