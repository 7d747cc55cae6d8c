#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>/ (written by tools/profile_round.sh) into profiles/<tag>_*.

FETCH_SIZE / WRITE_SIZE are reported in KiB summed over the 8 XCDs; on gfx950 FETCH_SIZE counts a 16 B/lane
coalesced stream at half its bytes (MI355X_MICROARCH.md, HBM section), so hbm_read = raw_kb * 1024 * 2; WRITE_SIZE is
taken as is.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for key in ("ipcr_filter", "ipcr_index", "pack_kernel", "verify_kernel", "filter_generic", "filter_index",
                "lcg_fill", "fill_pad"):
        if key in name:
            return key
    return None


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: only the latest run of a directory counts"""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


def pmc(dirname, counter):
    per = {}
    for f in newest(os.path.join(dirname, "**", "*counter_collection.csv")):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = short(row["Kernel_Name"])
                if k:
                    per.setdefault(k, []).append(float(row["Counter_Value"]))
    return {k: {"launches": len(v), "avg_raw_kb": sum(v) / len(v)} for k, v in per.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
    bench = json.loads(line)
    with open(os.path.join(dst, f"{tag}_bench.json"), "w") as fh:
        fh.write(line + "\n")
    serial = os.path.join(src, "bench_serial.json")
    if os.path.exists(serial):
        with open(os.path.join(dst, f"{tag}_bench_serial.json"), "w") as fh:
            fh.write(open(serial).read().strip().splitlines()[-1] + "\n")
    stats = newest(os.path.join(src, "stats", "**", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    fetch = pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write = pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    kernel = "ipcr_filter"
    out = {
        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py "
                "--steps 4 --warmup 1 --no-cpu-baseline`; FETCH_SIZE is reported in KiB and, on gfx950, at half the "
                "bytes of a 16 B/lane coalesced stream (MI355X_MICROARCH.md, HBM): hbm_read = raw_kb*1024*2; "
                "WRITE_SIZE exact.",
        "FETCH_SIZE": fetch, "WRITE_SIZE": write, "kernel": kernel,
    }
    if kernel in fetch and kernel in write:
        rd = int(fetch[kernel]["avg_raw_kb"] * 1024 * 2)
        wr = int(write[kernel]["avg_raw_kb"] * 1024)
        out.update(hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr,
                   algorithmic_bytes_per_launch=int(bench["roofline"].get("algorithmic_bytes_per_launch", 0)) or None)
    with open(os.path.join(dst, f"{tag}_filter_pmc.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: out.get(k) for k in ("hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch")}))
    if stats:
        with open(stats[0], newline="") as fh:
            for row in csv.DictReader(fh):
                if short(row.get("Name", "")):
                    print(row["Name"][:40], row.get("Calls"), row.get("AverageNs"))


if __name__ == "__main__":
    main()
