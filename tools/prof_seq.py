"""Launch sequence from a rocprofv3 results .db (rocpd): name, duration and idle gap before each kernel, for the
kernels whose name contains a given substring and everything between them:
python tools/prof_seq.py <db> <substring> [max rows]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
needle = sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(con.execute("select name, start, end from kernels order by start"))
hits = [i for i, r in enumerate(rows) if needle in r[0]]
if not hits:
    sys.exit(f"no kernel matches {needle}")
lo = hits[max(0, len(hits) - 4)]
prev = rows[lo - 1][2] if lo > 0 else rows[lo][1]
for name, s, e in rows[lo:hits[-1] + 1][:limit]:
    print(f"gap {(s - prev) / 1e3:8.2f} us   run {(e - s) / 1e3:9.2f} us   {name[:90]}")
    prev = e
