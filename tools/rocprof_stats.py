#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats writes a rocpd sqlite database: dump its top_kernels view as the CSV kept under profiles/.

  python tools/rocprof_stats.py <rocprof output dir> <out.csv> [command string for the header]   (every kernel, no truncation)
"""
import glob, sqlite3, sys, os
d, out = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --cpu-batch 0 --no-kernel-timer --no-f32-exact --no-legs  (MI355X, C3 B=512; 3 steps in the trace; durations in us)"
rows = []
for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
    db = sqlite3.connect(path)
    try:
        rows += list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    except Exception as e:
        cols = [r[1] for r in db.execute("pragma table_info(top_kernels)")]
        print("columns:", cols, e)
        raise
with open(out, "w") as f:
    f.write(f'"# {cmd}"\n')
    f.write("Name,Calls,TotalDuration_us,Average_us,Percentage\n")
    for n, c, t, a, p in sorted(rows, key=lambda r: -r[2]):
        f.write(f'"{n}",{c},{t:.1f},{a:.2f},{p:.3f}\n')
print(open(out).read())
