#!/usr/bin/env python3
"""print per-kernel count / avg / min / max (ns) from a rocprofv3 results.db:  python tools/prof_db.py <db> [...]"""
import sqlite3, sys
for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    print(path)
    for r in cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                         f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"):
        print("  %-70s n=%-5d avg=%9.0f min=%8d max=%8d" % r)
