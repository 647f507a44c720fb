#!/usr/bin/env python3
"""Condenses rocprofv3 output (kernel_stats.csv of a --kernel-trace --stats run, counter_collection.csv
of --pmc runs) into the small tracked summaries under profiles/.

usage: python scripts/summarize_profile.py <gpurun_out/prof_dir> <profiles/out_prefix> [note]
expects <dir>/stats/**/**_kernel_stats.csv, optionally <dir>/fetch/**/**_counter_collection.csv and
<dir>/write/**/**_counter_collection.csv (separate PMC passes, as MI355X_MICROARCH.md prescribes), and
<dir>/sq_a, <dir>/sq_b (SQ counter passes of the same command: scripts/profile_bench.sh).

Writes <prefix>.md, <prefix>_kernel_stats.csv, <prefix>_traffic.json (HBM bytes per launch AND the profile's average
kernel duration, so that bench.py can state whether the run it describes matches the profile) and <prefix>_sq.json
(SQ counters per launch of the nufft kernels, from which bench.py derives `roofline.binding_resource`).
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("void nufft::", "nufft::")
    return name[:110]


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    out = []
    out.append(f"# rocprofv3 summary ({os.path.basename(prefix)})\n")
    if note:
        out.append(note + "\n")
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        out.append("## rocprofv3 --kernel-trace --stats (per-kernel time)\n")
        out.append("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|")
        for r in csv.DictReader(open(stats[0])):
            if float(r["Percentage"]) < 0.05:
                continue
            out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
                       f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
        out.append("")
    pmc = {}
    for key in ("fetch", "write"):
        files = glob.glob(os.path.join(src, key, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        pmc[key] = {k: sum(v) / len(v) for k, v in agg.items()}
    if pmc:
        out.append("## HBM traffic per launch from PMC counters (separate --pmc passes)\n")
        out.append("FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide "
                   "coalesced streaming read (MI355X_MICROARCH.md §HBM): the `read GB (x2)` column applies that correction; "
                   "WRITE_SIZE is exact for 16-B stores and float atomics.\n")
        out.append("| kernel | FETCH_SIZE raw GB | read GB (x2) | WRITE_SIZE GB | traffic GB (corrected) |\n|---|---|---|---|---|")
        names = sorted(set(pmc.get("fetch", {})) | set(pmc.get("write", {})))
        for n in names:
            f = pmc.get("fetch", {}).get(n, 0.0) * 1024 / 1e9
            w = pmc.get("write", {}).get(n, 0.0) * 1024 / 1e9
            if f + w < 0.01:
                continue
            out.append(f"| `{short(n)}` | {f:.3f} | {2 * f:.3f} | {w:.3f} | {2 * f + w:.3f} |")
        out.append("")
    avg_us = {}
    if stats:
        for r in csv.DictReader(open(stats[0])):
            avg_us[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    if pmc:
        import json
        traffic = {}
        for n in sorted(set(pmc.get("fetch", {})) | set(pmc.get("write", {}))):
            f = pmc.get("fetch", {}).get(n, 0.0) * 1024
            w = pmc.get("write", {}).get(n, 0.0) * 1024
            traffic[short(n)] = {"fetch_size_bytes_raw": f, "read_bytes_corrected": 2 * f, "write_bytes": w,
                                 "hbm_bytes_per_launch": 2 * f + w}
            if short(n) in avg_us:
                traffic[short(n)]["kernel_avg_us"] = avg_us[short(n)][0]
                traffic[short(n)]["kernel_calls"] = avg_us[short(n)][1]
        with open(prefix + "_traffic.json", "w") as fh:
            json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB -> bytes, "
                                 "FETCH_SIZE x2 gfx950 correction (MI355X_MICROARCH.md)", "kernels": traffic}, fh, indent=1)
    # SQ counters per launch (averages over the launches of a kernel) of the library's own kernels
    sq = collections.defaultdict(dict)
    for key in ("sq_a", "sq_b", "sq_c"):
        for f in glob.glob(os.path.join(src, key, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                if "nufft::" in r["Kernel_Name"]:
                    agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                for c, vals in v.items():
                    sq[k][c] = sum(vals) / len(vals)
    if sq:
        import json
        big = {k: v for k, v in sq.items() if v.get("SQ_BUSY_CYCLES", 0) > 1e6 or v.get("SQ_WAVE_CYCLES", 0) > 1e8 or v.get("SQ_WAIT_ANY", 0) > 1e7}
        out.append("## SQ counters per launch (separate --pmc passes of the same command)\n")
        cols = sorted({c for v in big.values() for c in v})
        out.append("| kernel | " + " | ".join(cols) + " |\n|---|" + "---|" * len(cols))
        for k, v in sorted(big.items()):
            out.append(f"| `{k[:70]}` | " + " | ".join(f"{v.get(c, float('nan')):.4g}" for c in cols) + " |")
        out.append("")
        with open(prefix + "_sq.json", "w") as fh:
            json.dump({"source": "rocprofv3 --kernel-trace --pmc <SQ counters> (separate passes; averages per launch)", "kernels": big}, fh, indent=1)
    with open(prefix + ".md", "w") as fh:
        fh.write("\n".join(out) + "\n")
    if stats:
        with open(prefix + "_kernel_stats.csv", "w") as fh:
            fh.write(open(stats[0]).read())
    print("wrote", prefix + ".md")


if __name__ == "__main__":
    main()
