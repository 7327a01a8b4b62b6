"""Condense a rocprofv3 kernel_stats.csv into a short table: python tools/summarize_prof.py <csv> [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
total = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"# total kernel time {total/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
print(f"{'kernel':70s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}")
for r in rows[:top]:
    name = r["Name"]
    name = name if len(name) <= 70 else name[:67] + "..."
    print(f"{name:70s} {int(r['Calls']):6d} {int(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
