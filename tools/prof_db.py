"""Summarise a rocprofv3 results .db (kernel-trace): per-kernel totals, optionally GEMM launches grouped by grid."""
import sqlite3, sys
db, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
mode = sys.argv[3] if len(sys.argv) > 3 else "kernels"
c = sqlite3.connect(db)
if mode == "kernels":
    rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print("total %.2f ms  per step %.3f ms" % (tot / 1e6, tot / 1e6 / steps))
    for r in rows[:40]:
        print("%-100s %6d %9.3f ms/step %8.1f us %5.1f%%" % (r[0][:100], r[1], r[2] / 1e6 / steps, r[3] / 1e3, 100 * r[2] / tot))
else:
    rows = list(c.execute("select name, grid_x, grid_z, workgroup_x, count(*), sum(end-start), avg(end-start) from kernels "
                          "where name like '%gemm_kernel%' group by name, grid_x, grid_z order by 6 desc"))
    for r in rows[:40]:
        print("%-30s wgs=%6d z=%3d n=%5d %8.3f ms/step avg=%7.1f us" % (r[0].split('gemm_kernel')[1][:28], r[1] // r[3], r[2], r[4], r[5] / 1e6 / steps, r[6] / 1e3))
