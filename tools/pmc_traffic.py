#!/usr/bin/env python3
"""Turns `rocprofv3 --pmc` CSVs into profiles/pmc_traffic.json (HBM bytes per launch, per kernel) and
an MFMA-utilisation summary.

Counters are collected in SEPARATE passes (FETCH_SIZE uses 3 of the 4 TCC slots, WRITE_SIZE 2), one run
of the same command each, without any trace option:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_mfma ...

Corrections applied, as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
  * FETCH_SIZE and WRITE_SIZE are in KiB -> x 1024;
  * FETCH_SIZE reports half the bytes of a wide coalesced streaming read -> x 2 (kernels whose reads are
    narrower are uncalibrated; the raw figure is kept next to the corrected one);
  * WRITE_SIZE is exact for streaming stores.

    python tools/pmc_traffic.py --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write [--mfma gpurun_out/pmc_mfma] --frames T
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def base_name(kernel):
    k = kernel.strip()
    m = re.match(r"^_Z(\d+)", k)     # a mangled name rocprofv3 left as it was (templates over _Float16 vectors)
    if m:
        n = int(m.group(1))
        return k[m.end():m.end() + n]
    k = re.sub(r"^void\s+", "", k)
    return re.split(r"[<(]", k, maxsplit=1)[0].strip()


def read_counters(directory):
    """-> {kernel base name: {counter: [value per dispatch, ...]}}"""
    out = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = base_name(row["Kernel_Name"])
                # one row per (dispatch, counter[, dimension instance]): sum the instances of a dispatch
                out[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: {c: list(d.values()) for c, d in cs.items()} for k, cs in out.items()}


def mean(v):
    return sum(v) / len(v) if v else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--mfma")
    ap.add_argument("--frames", type=int, required=True, help="frames per launch of the profiled workload")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    fetch, write = read_counters(args.fetch), read_counters(args.write)
    mf = read_counters(args.mfma) if args.mfma else {}
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        fkb = mean(fetch.get(k, {}).get("FETCH_SIZE", []))
        wkb = mean(write.get(k, {}).get("WRITE_SIZE", []))
        rec = {
            "frames": args.frames,
            "launches_seen": len(fetch.get(k, {}).get("FETCH_SIZE", [])),
            "fetch_size_kib_raw": fkb, "write_size_kib": wkb,
            "hbm_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0,
            "hbm_bytes_per_launch_uncorrected": (fkb + wkb) * 1024.0,
            "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count) + WRITE_SIZE, x1024 (KiB)",
            "source": args.tag,
        }
        m = mf.get(k)
        if m and "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            busy, act = mean(m["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(m["GRBM_GUI_ACTIVE"])
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy over 256 CUs x 4 SIMDs
            rec["mfma_busy_frac"] = busy / (4 * 256 * (act / 8.0)) if act else None
        res[k] = rec
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for k, r in res.items():
        print("%-16s fetch %.1f MiB(raw) write %.1f MiB -> %.1f MiB/launch  %s" % (
            k, r["fetch_size_kib_raw"] / 1024, r["write_size_kib"] / 1024, r["hbm_bytes_per_launch"] / 2 ** 20,
            ("mfma busy %.3f" % r["mfma_busy_frac"]) if r.get("mfma_busy_frac") else ""))


if __name__ == "__main__":
    main()
