"""Condense a rocprofv3 results .db (rocpd) into a short per-kernel table: python tools/prof_db.py <db> [top]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(con.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by 3 desc"))
total = sum(r[2] for r in rows)
span = con.execute("select max(end) - min(start) from kernels").fetchone()[0]
print(f"kernel time {total / 1e6:.2f} ms over a span of {span / 1e6:.2f} ms, {sum(r[1] for r in rows)} launches")
print(f"{'calls':>6} {'total ms':>10} {'avg us':>10} {'%':>6}  name")
for name, calls, tot, avg in rows[:top]:
    print(f"{calls:6d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {100 * tot / total:6.2f}  {name[:110]}")
