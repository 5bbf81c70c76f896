#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd sqlite database: python scripts/prof_db.py <results.db> [steps]."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("total kernel time %.3f ms (%.3f ms/step over %g steps)" % (tot / 1e6, tot / 1e6 / steps, steps))
for name, n, t, avg in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)[:70]
    print("%-70s n %6d  tot %9.3f ms  %6.2f%%  avg %8.1f us  /step %7.3f ms" % (name, n, t / 1e6, 100 * t / tot, avg / 1e3, t / 1e6 / steps))
