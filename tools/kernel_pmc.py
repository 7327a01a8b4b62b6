"""Per-kernel averages of rocprofv3 PMC passes.
    python tools/kernel_pmc.py <out.json> <name pattern>[,<pattern>...] <pass results.db> [<results.db> ...]
Every .db is one `rocprofv3 --pmc <counters>` pass over the same program.  For each kernel whose name contains one of
the patterns: launches per pass and, per counter, the average value per launch (rows of one dispatch are summed:
rocprofv3 reports one row per instance for some blocks).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts each 128-byte read request of a wide coalesced stream as 64 bytes, so `hbm_bytes` =
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 when both were collected."""
import json
import sqlite3
import sys

out_path, patterns, paths = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
summary = {}
for path in paths:
    con = sqlite3.connect(path)
    for pat in patterns:
        rows = con.execute("select dispatch_id, counter_name, sum(value), max(duration) from counters_collection "
                           "where kernel_name like ? group by dispatch_id, counter_name", (f"%{pat}%",)).fetchall()
        if not rows:
            continue
        entry = summary.setdefault(pat, {"launches_per_pass": None, "avg_per_launch": {}, "avg_duration_us_under_pmc": {}})
        by_counter = {}
        for _, counter, value, duration in rows:
            by_counter.setdefault(counter, []).append((value, duration))
        for counter, vals in by_counter.items():
            entry["avg_per_launch"][counter] = sum(v for v, _ in vals) / len(vals)
            entry["launches_per_pass"] = len(vals)
            entry["avg_duration_us_under_pmc"][counter] = sum(d for _, d in vals) / len(vals) / 1e3
for entry in summary.values():
    a = entry["avg_per_launch"]
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        entry["hbm_bytes"] = (2 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024
json.dump({"method": __doc__, "kernels": summary}, open(out_path, "w"), indent=1)
print(json.dumps(summary, indent=1))
