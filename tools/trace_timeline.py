"""Per-queue timeline of one bench step from a rocprofv3 kernel trace: python tools/trace_timeline.py <kernel_trace.csv>
Prints, for the last step (after the last big gap), each queue's busy span and its top kernels."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# last step = from the last syrk launch on C=14336 (the longest kernel) onwards
syrk = [r for r in rows if "syrk_kernel" in r["Kernel_Name"] and r["e"] - r["s"] > 30e6]
t0 = syrk[-1]["s"]
step = [r for r in rows if r["s"] >= t0 - 5e6]
tend = max(r["e"] for r in step)
print(f"step span {(tend - t0) / 1e6:.1f} ms, {len(step)} kernels")
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: kv[1][0]["s"]):
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"queue {q}: first {(rs[0]['s'] - t0) / 1e6:7.2f} ms  last end {(max(r['e'] for r in rs) - t0) / 1e6:7.2f} ms  "
          f"kernel time {busy / 1e6:7.2f} ms  launches {len(rs)}")
    agg = collections.Counter()
    for r in rs:
        agg[r["Kernel_Name"][:60]] += r["e"] - r["s"]
    for name, t in agg.most_common(5):
        print(f"      {t / 1e6:7.2f} ms  {name}")
