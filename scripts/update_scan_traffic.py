"""Refresh the filter-kernel entries of profiles/scan_traffic.json from the counter summaries of a collection run:
    python scripts/update_scan_traffic.py gpurun_out/r03      (after scripts/collect_profiles_r03.sh pmc forms)
Every entry is stamped with the hash of the kernel sources of THIS tree (bench.py reports entries of other sources as
stale), so run it on the tree the counters were taken from."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha16

src = sys.argv[1]
tf = os.path.join(ROOT, "profiles", "scan_traffic.json")
db = json.load(open(tf))
for tag, key, name in (("filter", "filter_kernel_n10000000_m16_B1024", "filter_kernel_pmc.csv"),
                       ("c5", "filter_kernel_n10000000_m64_B1024", "c5_kernel_pmc.csv"),
                       ("m25", "filter_kernel_n1000000_m25_B1024", "m25_kernel_pmc.csv")):
    path = os.path.join(src, f"{tag}_kernel_pmc.csv")
    if not os.path.exists(path):
        continue
    c = {r["counter"]: float(r["mean_per_dispatch"]) for r in csv.DictReader(open(path))}
    rec = db.get(key, {})
    rec.update({
        "hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
        "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
        "grbm_gui_active": c["GRBM_GUI_ACTIVE"], "lds_idx_active": c["SQ_LDS_IDX_ACTIVE"],
        "lds_bank_conflict": c["SQ_LDS_BANK_CONFLICT"], "cus": 256,
        # vector-ALU busy: issue cycles (4 per instruction and SIMD) over the kernel's cycles
        "valu_busy": c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8.0),
        "source": f"profiles/r03/{name}", "source_sha16": kernel_source_sha16("filter_kernel")})
    db[key] = rec
    print(key, "hbm GB/launch", round(rec["hbm_bytes_per_launch"] / 1e9, 3), "valu", round(rec["valu_busy"], 4),
          "lds busy", round(c["SQ_LDS_IDX_ACTIVE"] / 256 / (c["GRBM_GUI_ACTIVE"] / 8.0), 4))
json.dump(db, open(tf, "w"), indent=1)
