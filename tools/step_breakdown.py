"""Per-step kernel accounting from a rocprofv3 kernel trace (results .db): the LAST `nsteps` optimizer steps are cut out of the
trace (a step ends with its adamw_kernel launch) and summarised per kernel (GEMMs per template + grid), with the device idle time
between consecutive kernels.  Usage: python tools/step_breakdown.py results.db [nsteps] [marker-substring] [skip]
(skip = number of trailing marker launches to leave out: bench.py runs 6 plan-building + `--steps` shot-mix steps behind the
headline loop, so `--steps 10` -> skip 16 selects headline steps.)  A final "seq" argument prints one step's launch sequence."""
import sqlite3
import sys

db = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
marker = sys.argv[3] if len(sys.argv) > 3 else "adamw_kernel"
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if db.endswith(".csv"):      # rocprofv3 --output-format csv: <name>_kernel_trace.csv
    import csv
    rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]),
             int(r["Grid_Size_Z"])) for r in csv.DictReader(open(db))]
    rows.sort(key=lambda r: r[1])
else:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    if "kernels" not in tabs:
        raise SystemExit("no kernels view in %s: %s" % (db, tabs))
    rows = list(c.execute("select name, start, end, grid_x, workgroup_x, grid_z from kernels order by start"))
ends = [i for i, r in enumerate(rows) if marker in r[0]]
if skip:
    ends = ends[:-skip]
if len(ends) < nsteps + 1:
    raise SystemExit("only %d marker launches" % len(ends))
lo, hi = ends[-nsteps - 1] + 1, ends[-1] + 1
sel = rows[lo:hi]
wall = (sel[-1][2] - rows[lo - 1][2]) / nsteps / 1e3
busy, idle, agg = 0.0, 0.0, {}
prev_end = rows[lo - 1][2]
for name, s, e, gx, wx, gz in sel:
    key = name.replace("(anonymous namespace)::", "").replace("void ", "")
    key = key.split("(")[0]
    if "gemm_kernel" in key or "lin_kernel" in key:
        key = "%s wgs=%d z=%d" % (key.replace("unsigned short", "bf16"), gx // max(wx, 1), gz)
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    busy += (e - s) / 1e3
    idle += max(0, s - prev_end) / 1e3
    prev_end = max(prev_end, e)
print("steps %d: wall %.1f us/step, kernels %.1f us/step, idle between kernels %.1f us/step, %d launches/step"
      % (nsteps, wall, busy / nsteps, idle / nsteps, len(sel) // nsteps))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-86s %6.1f /step %8.1f us avg %8.1f us/step %5.1f%%" % (k[:86], n / nsteps, t / n, t / nsteps, 100 * t / busy))
if sys.argv[-1] == "seq":
    a, b = ends[-2] + 1, ends[-1] + 1
    prev = t0 = rows[a - 1][2]
    for name, s, e, gx, wx, gz in rows[a:b]:     # start / end offsets from the step's start show which launches overlap (side lanes)
        n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        print("%-72s wgs=%5d z=%2d %7.1f us  gap %5.1f  [%8.1f .. %8.1f]" % (n, gx // max(wx, 1), gz, (e - s) / 1e3, (s - prev) / 1e3,
                                                                           (s - t0) / 1e3, (e - t0) / 1e3))
        prev = max(prev, e)
