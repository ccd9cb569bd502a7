#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes per kernel: average counter values per launch of every kernel whose name contains one of
the given substrings, over any number of output directories (one counter group per pass: MI355X_MICROARCH.md, PMC slots).

  python tools/pmc_kernel.py --out profiles/r02_pmc_gram_cholesky.json --kernel gram_chol --note "..." DIR [DIR ...]

Derived figures written next to the raw averages (when the counters are present):
  traffic_bytes_per_launch   2 x FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH doubled: the guide's gfx950 correction)
  mfma_busy_frac             SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x GRBM_GUI_ACTIVE / 8): share of the kernel's cycles in which a
                             SIMD's matrix pipe is busy, at the clock the chip actually held (GRBM_GUI_ACTIVE sums the 8 XCDs)
"""
import argparse, csv, glob, json, os, sqlite3
from collections import defaultdict

SIMDS = 256 * 4


def rows(d):
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            yield r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])
    for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(path)
        try:
            for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
                yield k, c, float(v)
        except sqlite3.Error:
            pass


ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--kernel", action="append", required=True)
ap.add_argument("--out", required=True)
ap.add_argument("--note", default="")
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--source", default=None, help="kernel source file: its SHA-256 (first 16 hex digits) is stamped into the summary")
a = ap.parse_args()
acc = defaultdict(lambda: defaultdict(list))
for d in a.dirs:
    for k, c, v in rows(d):
        for sub in a.kernel:
            if sub in k:
                acc[sub][c].append(v)
out = {"command": a.note, "batch": a.batch, "kernels": {}}
if a.source:
    import hashlib
    out["kernel_source"] = os.path.relpath(a.source)
    out["kernel_source_sha16"] = hashlib.sha256(open(a.source, "rb").read()).hexdigest()[:16]
for sub, cs in acc.items():
    e = {"launches": max(len(v) for v in cs.values()), "counters_avg_per_launch": {c: sum(v) / len(v) for c, v in sorted(cs.items())}}
    m = e["counters_avg_per_launch"]
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["fetch_size_bytes_raw_avg"], e["write_size_bytes_avg"] = m["FETCH_SIZE"] * 1024, m["WRITE_SIZE"] * 1024
        e["traffic_bytes_per_launch"] = 2 * 1024 * m["FETCH_SIZE"] + 1024 * m["WRITE_SIZE"]
        e["correction"] = "FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports 1/2 of a wide coalesced 16 B/lane read)"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        e["kernel_cycles"] = m["GRBM_GUI_ACTIVE"] / 8
        e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * m["GRBM_GUI_ACTIVE"] / 8)
    out["kernels"][sub] = e
if len(out["kernels"]) == 1:                       # bench.py reads these two keys at top level
    only = next(iter(out["kernels"].values()))
    for key in ("traffic_bytes_per_launch",):
        if key in only:
            out[key] = only[key]
json.dump(out, open(a.out, "w"), indent=1)
print(open(a.out).read())
