#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md) of bench.py into
profiles/pmc_conv_tangent_<precision>.json: average HBM-side bytes per launch of the dominant kernel.

  python tools/pmc_summary.py <fetch_dir> <write_dir> <kernel substring> <out.json> [command string]
"""
import csv, glob, json, os, sqlite3, sys


def avg(d, counter, kernel):
    vals = []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):      # rocprofv3's default rocpd output
        db = sqlite3.connect(path)
        vals += [float(v) for (v,) in db.execute("select value from counters_collection where counter_name = ? and "
                                                 "instr(kernel_name, ?) > 0", (counter, kernel))]
    if not vals:
        raise SystemExit(f"no {counter} rows for '{kernel}' under {d}")
    return sum(vals) / len(vals) * 1024.0, len(vals)          # the counters are in KiB


fetch_dir, write_dir, kernel, out = sys.argv[1:5]
fetch, n = avg(fetch_dir, "FETCH_SIZE", kernel)
write, _ = avg(write_dir, "WRITE_SIZE", kernel)
json.dump({
    "kernel": kernel,
    "command": sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 bench.py ... (separate passes)",
    "launches": n,
    "fetch_size_bytes_raw_avg": fetch,
    "write_size_bytes_avg": write,
    "correction": "FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports 1/2 of a wide coalesced 16 B/lane read)",
    "traffic_bytes_per_launch": 2.0 * fetch + write,
}, open(out, "w"), indent=1)
print(open(out).read())
