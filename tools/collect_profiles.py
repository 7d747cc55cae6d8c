#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>/ (written by tools/profile_round.sh) into profiles/<tag>_*.

FETCH_SIZE / WRITE_SIZE are reported in KiB summed over the 8 XCDs; on gfx950 FETCH_SIZE counts a 16 B/lane
coalesced stream at half its bytes (MI355X_MICROARCH.md, HBM section), so hbm_read = raw_kb * 1024 * 2; WRITE_SIZE is
taken as is.  SQ_* counters are summed over the chip; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = {"c2": "ipcr_filter", "c2n": "ipcr_filter", "c3": "ipcr_filter", "c4": "ipcr_index_filter", "c4n": "ipcr_index_filter",
          "c2g": "filter_generic_quad_kernel"}   # c2g: C2 through the table-driven kernel (IPCR_SPECIALIZE=0)
ALG_BYTES = 1_125_000_000   # 3.0e9 bases x 0.375 B: what one sweep of the benchmark genome reads


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: only the latest run of a directory counts"""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


def counters(dirname, kernel):
    """average per launch of every counter collected in `dirname` for dispatches of `kernel`"""
    acc = {}
    for f in newest(os.path.join(dirname, "**", "*counter_collection.csv")):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if kernel in row["Kernel_Name"] and not (kernel == "ipcr_filter" and "index" in row["Kernel_Name"]):
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: {"launches": len(v), "avg": sum(v) / len(v)} for k, v in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    if os.path.exists(os.path.join(src, "bench.json")):
        line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
        with open(os.path.join(dst, f"{tag}_bench.json"), "w") as fh:
            fh.write(line + "\n")
    else:   # only the prof stage was run this time (profile_round.sh <tag> prof "..."): the committed line stays
        line = open(os.path.join(dst, f"{tag}_bench.json")).read().strip().splitlines()[-1]
    bench = json.loads(line)
    serial = os.path.join(src, "bench_serial.json")
    if os.path.exists(serial):
        with open(os.path.join(dst, f"{tag}_bench_serial.json"), "w") as fh:
            fh.write(open(serial).read().strip().splitlines()[-1] + "\n")
    for w in ("c2n", "c3", "c4"):   # the other workloads as the main line (their own cpu_baseline)
        f = os.path.join(src, f"bench_{w}.json")
        if os.path.exists(f):
            with open(os.path.join(dst, f"{tag}_bench_{w}.json"), "w") as fh:
                fh.write(open(f).read().strip().splitlines()[-1] + "\n")
    alg = int(bench["roofline"].get("algorithmic_bytes_per_launch", 0)) or None
    for w, kernel in KERNEL.items():
        stats = newest(os.path.join(src, f"stats_{w}", "**", "*kernel_stats.csv"))
        if stats:
            shutil.copy(stats[0], os.path.join(dst, f"{tag}_{w}_kernel_stats.csv"))
            with open(stats[0], newline="") as fh:
                for row in csv.DictReader(fh):
                    if kernel in row.get("Name", ""):
                        print(w, row["Name"][:40], row.get("Calls"), row.get("AverageNs"))
        out = {"workload": w, "kernel": kernel,
               "note": "rocprofv3 --kernel-trace --pmc <set> (one set per pass, nothing else traced) over `python3 bench.py "
                       "--workload %s --no-cpu-baseline --no-others --steps 3..4 --warmup 1`; averages per launch of %s. "
                       "FETCH_SIZE is reported in KiB and, on gfx950, at half the bytes of a 16 B/lane coalesced stream "
                       "(MI355X_MICROARCH.md, HBM): hbm_read = raw_kb*1024*2; WRITE_SIZE exact." % (w, kernel)}
        found = 0
        for name in ("fetch", "write", "sq", "lds", "grbm"):
            for k, v in counters(os.path.join(src, f"pmc_{name}_{w}"), kernel).items():
                out[k] = v
                found += 1
        if not found:   # this workload was not profiled in this collection (profile_round.sh <tag> prof "c4"): its files stay
            continue
        if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
            rd = int(out["FETCH_SIZE"]["avg"] * 1024 * 2)
            wr = int(out["WRITE_SIZE"]["avg"] * 1024)
            out.update(hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr,
                       algorithmic_bytes_per_launch=alg)
        if w in ("c2", "c2n", "c3") and alg and "SQ_INSTS_VALU" in out and "GRBM_GUI_ACTIVE" in out:
            # the specialised filter: one wave per block of 64 columns x 32 strands x 128 bases, 128 + 19 row steps each
            blocks = -(-int(alg / 0.375) // 262144)
            cu_cycles = 256.0 * out["GRBM_GUI_ACTIVE"]["avg"] / 8.0
            out["derived"] = {
                "blocks": blocks,
                "row_steps_per_launch": blocks * 147,
                "valu_instructions_per_row_step": round(out["SQ_INSTS_VALU"]["avg"] / (blocks * 147), 1),
                "valu_issue_busy_frac_upper_bound": round(out["SQ_INSTS_VALU"]["avg"] * 4.0 / (4.0 * cu_cycles), 3),
                "hbm_bytes_over_algorithmic": round(out["hbm_bytes_per_launch"] / alg, 4) if "hbm_bytes_per_launch" in out else None,
            }
        if w in ("c4", "c4n") and all(k in out for k in ("SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")):
            # the index kernel is not HBM-bound: say what binds it, from the counters themselves.  One wave-level base step =
            # 64 bases (lane = strand, no tail rows): the genome of the profiled run / 64 (its size is in the bench line's
            # algorithmic bytes: 0.375 bytes per base)
            genome_bases = (alg or 0) / 0.375
            steps = genome_bases / 64.0
            cu_cycles = 256.0 * out["GRBM_GUI_ACTIVE"]["avg"] / 8.0          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            out["derived"] = {
                "genome_bases": int(genome_bases),
                "wave_base_steps_per_launch": round(steps),
                "valu_instructions_per_base_step": round(out["SQ_INSTS_VALU"]["avg"] / steps, 1),
                "lds_instructions_per_base_step": round(out["SQ_INSTS_LDS"]["avg"] / steps, 2) if "SQ_INSTS_LDS" in out else None,
                # an upper bound: 4 cycles per wave instruction on 4 SIMDs per CU -- v_and / v_or / v_xor / v_lshrrev / v_add /
                # v_bitop3 issue in 2 (tools/ubench/valu_rates.hip), so the SIMDs are less busy than this says
                "valu_issue_busy_frac_upper_bound": round(out["SQ_INSTS_VALU"]["avg"] * 4.0 / (4.0 * cu_cycles), 3),
                "lds_busy_frac": round(out["SQ_LDS_IDX_ACTIVE"]["avg"] / cu_cycles, 3),
                "lds_bank_conflict_frac_of_busy": round(out["SQ_LDS_BANK_CONFLICT"]["avg"] / out["SQ_LDS_IDX_ACTIVE"]["avg"], 3),
                "lds_cycles_per_lds_instruction": round(out["SQ_LDS_IDX_ACTIVE"]["avg"] / out["SQ_INSTS_LDS"]["avg"], 2) if "SQ_INSTS_LDS" in out else None,
            }
            out["note"] += (" CAUTION for this kernel: the x2 rule is calibrated for 16 B/lane coalesced streams; the index filter loads "
                            "its tiles one dword per lane (the 32 lanes of a half wave take the 32 rows of a chunk of one column: "
                            "eight 16-byte pieces per wave instruction), an access width the guide calls uncalibrated, so hbm_read "
                            "is an upper bound and the raw counter a lower one.  Every tile word is loaded once per sweep, plus the "
                            "last chunk of a strand a second time as the next strand's history.  The kernel is bound by the LDS "
                            "(random byte lookups, most of their cycles bank conflicts), see `derived`.")
        name = "filter" if w == "c2" else w
        with open(os.path.join(dst, f"{tag}_{name}_pmc.json"), "w") as fh:
            json.dump(out, fh, indent=1)
        print(w, {k: out.get(k) for k in ("hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch")},
              {k: round(v["avg"]) for k, v in out.items() if isinstance(v, dict) and k.startswith("SQ_")})


# ------------------------------------------------------------------------------------------------ profiles/README.md
def _num(x, nd=4):
    return ("%." + str(nd) + "g") % x


def kernel_stats_row(path):
    """(kernel, calls, average us) of the dominant ipcr kernel in a rocprofv3 --stats CSV"""
    best = None
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name", "")
            if "ipcr_filter" in name or "ipcr_index_filter" in name or ("_c2g_" in path and "filter_generic_quad_kernel" in name):
                tot = float(row.get("TotalDurationNs", 0) or 0)
                if best is None or tot > best[3]:
                    best = (name.split("(")[0], int(row["Calls"]), float(row["AverageNs"]) / 1e3, tot)
    return best[:3] if best else None


def readme_text():
    """profiles/README.md, every number read from the file its row describes (tests/test_host_logic.py checks that the
    committed README equals this text: the prose cannot drift from the evidence)"""
    d = os.path.join(ROOT, "profiles")
    files = sorted(os.listdir(d))
    tags = sorted({f.split("_")[0] for f in files if f[:1] == "r" and f[1:3].isdigit()}, reverse=True)
    out = ["# profiles/", "",
           "All from `gpurun` on one MI355X: `tools/profile_round.sh <tag> bench|prof` on the GPU box, then",
           "`python tools/collect_profiles.py <tag>` here; **this file is written by `python tools/collect_profiles.py readme`**,",
           "every figure below is read from the file named in its row (`tests/test_host_logic.py::test_profiles_readme_is_generated`",
           "fails when the two differ).  Fractions are of the 8 TB/s HBM3E spec; one sweep of the benchmark genome reads",
           "1.125 GB (3.0e9 bases x 0.375 B).  What the numbers mean and how they came about: DESIGN.md sections 4 and 5.", ""]
    for tag in tags:
        out += ["## " + tag, "", "| file | what it holds |", "|---|---|"]
        for f in [x for x in files if x.startswith(tag + "_")]:
            path = os.path.join(d, f)
            row = None
            if f.endswith("kernel_stats.csv"):
                ks = kernel_stats_row(path)
                if ks:
                    gbs = ALG_BYTES / (ks[2] * 1e-6) / 1e9
                    row = "`rocprofv3 --kernel-trace --stats`: `%s` %d calls, average %s us -> %s GB/s = %s of 8 TB/s" % (
                        ks[0], ks[1], _num(ks[2]), _num(gbs), _num(gbs / 8000.0, 3))
            elif f.endswith("_pmc.json"):
                j = json.load(open(path))
                parts = ["`rocprofv3 --pmc` passes, averages per launch of `%s`" % j.get("kernel", "?")]
                if j.get("hbm_bytes_per_launch") and j.get("algorithmic_bytes_per_launch"):
                    parts.append("HBM traffic %d B = %s x the algorithmic %d B" % (
                        j["hbm_bytes_per_launch"], _num(j["hbm_bytes_per_launch"] / j["algorithmic_bytes_per_launch"]), j["algorithmic_bytes_per_launch"]))
                dv = j.get("derived") or {}
                for k, label in (("valu_instructions_per_row_step", "VALU instructions per row step"),
                                 ("valu_instructions_per_base_step", "VALU instructions per 64-base step"),
                                 ("lds_instructions_per_base_step", "LDS instructions per step"),
                                 ("lds_busy_frac", "LDS busy fraction of the sweep"),
                                 ("lds_bank_conflict_frac_of_busy", "bank conflicts' share of the LDS time"),
                                 ("valu_issue_busy_frac_upper_bound", "VALU issue (upper bound)")):
                    if dv.get(k) is not None:
                        parts.append("%s %s" % (label, _num(dv[k])))
                row = "; ".join(parts)
            elif f.endswith(".json") and "_bench" in f:
                try:
                    j = json.loads(open(path).read().strip().splitlines()[-1])
                except Exception:  # noqa: BLE001
                    j = None
                if j and "roofline" in j:
                    r = j["roofline"]
                    parts = ["`bench.py` line: %s %s %s, %s ms per step; `%s` %s ms per launch = **%s** of 8 TB/s" % (
                        _num(j["value"], 5), j.get("unit", ""), "(" + j["config"]["workload"].split(":")[0] + ")" if "config" in j and "workload" in j["config"] else "",
                        _num(j["ms_per_step"]), r.get("kernel", "?"), _num(r.get("avg_launch_ms", 0)), _num(r.get("frac", 0)))]
                    if r.get("traffic_over_algorithmic"):
                        parts.append("traffic measured in the run %s x algorithmic" % _num(r["traffic_over_algorithmic"]))
                    ow = (j.get("config") or {}).get("other_workloads") or {}
                    for k in ("c2n", "c3", "c4", "c5"):
                        if k in ow and "sweep_ms" in ow[k]:
                            parts.append("%s sweep %s ms = %s, step %s ms" % (k, _num(ow[k]["sweep_ms"]), _num(ow[k]["roofline_frac"], 3), _num(ow[k]["ms_per_step"])))
                    sc = ow.get("scan_chunk") or {}
                    if sc:
                        parts.append("scan_chunk " + " / ".join("%s %s" % (k.replace("gbases_per_s_", "").replace("_", " "), _num(v)) for k, v in sc.items() if k.startswith("gbases_per_s")) + " Gbases/s")
                    cp = ow.get("scan_chunk_c4_panel") or {}
                    if cp:
                        parts.append("scan_chunk under the 1024-row panel, 16 workers: clean chunks %s, chunks with N %s Gbases/s" % (
                            _num(cp["gbases_per_s_clean"]), _num(cp["gbases_per_s_with_n"])))
                    c5 = ow.get("c5_chunk") or {}
                    if c5:
                        parts.append("c5_chunk " + "; ".join("%s %s" % (k, _num(v)) for k, v in c5.items() if k.startswith("gbases_per_s") or k.startswith("probe_best_hit_us") or k.startswith("probe_rescan_ms")))
                    ft = ow.get("fasta_to_tsv") or {}
                    if ft:
                        parts.append("FASTA -> TSV %s Gbases/s" % _num(ft["gbases_per_s"]))
                    cb = j.get("cpu_baseline")
                    if cb:
                        parts.append("cpu_baseline %s Gbases/s on %d threads (%s)" % (_num(cb["value"]), cb["cores"], cb["kind"]))
                    row = "; ".join(parts)
            if row is None:
                row = "(see the file)"
            out.append("| `%s` | %s |" % (f, row))
        out.append("")
    return "\n".join(out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "readme":
        with open(os.path.join(ROOT, "profiles", "README.md"), "w") as fh:
            fh.write(readme_text())
        sys.exit(0)
    main()
    with open(os.path.join(ROOT, "profiles", "README.md"), "w") as fh:
        fh.write(readme_text())
