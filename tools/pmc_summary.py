"""Summarise a rocprofv3 --pmc counter_collection.csv for one kernel: mean counter values per dispatch, mean duration
from the dispatch timestamps, and the clock GRBM_GUI_ACTIVE implies.  usage: pmc_summary.py CSV KERNEL_SUBSTRING"""
import csv, sys, collections

def main():
    path, key = sys.argv[1], sys.argv[2]
    vals = collections.defaultdict(list)
    dur = {}
    for r in csv.DictReader(open(path)):
        if key not in r["Kernel_Name"]:
            continue
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    if not dur:
        print("no dispatch of", key)
        return
    d = sorted(dur.values())
    ms = d[len(d) // 2]
    out = {k: sum(v) / len(v) for k, v in vals.items()}
    print(f"{key}: {len(dur)} dispatches, median {ms:.3f} ms")
    for k, v in sorted(out.items()):
        print(f"  {k:32s} {v:16.0f}")
    if "GRBM_GUI_ACTIVE" in out:
        print(f"  clock {out['GRBM_GUI_ACTIVE'] / ms * 1e-6:.2f} GHz")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "GRBM_GUI_ACTIVE" in out:
        # summed over SIMDs (4 per CU, 256 CUs) in units of 4 cycles on gfx950 (MI355X_MICROARCH.md, counters)
        print(f"  MFMA busy / (1024 SIMD x cycles) = {out['SQ_VALU_MFMA_BUSY_CYCLES'] * 4 / 1024 / out['GRBM_GUI_ACTIVE']:.3f}  (if the x4 unit applies)")

main()
