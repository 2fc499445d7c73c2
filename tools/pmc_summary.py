"""Summarise a rocprofv3 --pmc results db: per kernel name, mean of each counter."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
q = """select k.kernel_name, c.name, avg(p.value), count(*) from rocpd_pmc_event p
 join rocpd_info_pmc c on p.pmc_id = c.id join rocpd_kernel_dispatch d on p.event_id = d.event_id
 join rocpd_info_kernel_symbol k on d.kernel_id = k.id group by k.kernel_name, c.name"""
# launches of one kernel at different problem sizes (bench.py also times the attention kernel at B = 32) are kept apart by grid size
qg = """select k.kernel_name || ' grid=' || d.grid_size_x, c.name, avg(p.value), count(*) from rocpd_pmc_event p
 join rocpd_info_pmc c on p.pmc_id = c.id join rocpd_kernel_dispatch d on p.event_id = d.event_id
 join rocpd_info_kernel_symbol k on d.kernel_id = k.id group by k.kernel_name, d.grid_size_x, c.name"""
try:
    try:
        rows = list(cur.execute(qg))
    except Exception:
        rows = list(cur.execute(q))
except Exception as e:
    print("query failed:", e); print(tabs); sys.exit(0)
by = collections.defaultdict(dict)
for kn, cn, v, n in rows: by[kn][cn] = (v, n)
for kn, d in by.items():
    print(kn[:120])
    for cn, (v, n) in sorted(d.items()): print("    %-32s %16.1f  (n=%d)" % (cn, v, n))
