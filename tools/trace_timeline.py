"""Per-stream timeline of the last bench step from a rocprofv3 kernel trace (rocpd .db):
    python tools/trace_timeline.py <results.db>
For each stream: busy span, kernel time, and the sequence of phases (runs of launches of the same kernel family)."""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = [dict(name=n, s=s, e=e, stream=st) for n, s, e, st in
        con.execute("select name, start, end, stream_id from kernels order by start")]
# A Hessian is a run of back-to-back syrk_kernel launches (one per round of workgroups); the big one (C = 14336) is the
# run that lasts more than 30 ms.  A step starts at its big Hessian.
runs = []
for r in rows:
    if "syrk_kernel" not in r["name"]:
        continue
    if runs and r["s"] - runs[-1]["e"] < 0.5e6:
        runs[-1]["e"] = r["e"]
    else:
        runs.append(dict(s=r["s"], e=r["e"]))
syrk = [g for g in runs if g["e"] - g["s"] > 30e6]
# the last big Hessian that starts a whole step (bench.py ends with a pass of Hessians alone: skip those)
ends = [r["s"] for r in syrk[1:]] + [rows[-1]["e"] + 1]
starts = [(r, sum(1 for x in rows if r["s"] <= x["s"] < nxt and "sweep" in x["name"])) for r, nxt in zip(syrk, ends)]
t0 = [r for r, n in starts if n > 0][-1]["s"]
nxt = min([r["s"] for r in syrk if r["s"] > t0] + [rows[-1]["e"] + 1])
step = [r for r in rows if t0 - 5e6 <= r["s"] < nxt - 5e6]
tend = max(r["e"] for r in step)
print(f"step span {(tend - t0) / 1e6:.1f} ms, {len(step)} kernels")


def family(name: str) -> str:
    for key in ("syrk", "transpose_pad", "range_search", "panel", "gemm_f32", "trsm", "sweep", "fused_int", "lazy",
                "symmetr", "damp", "diag", "permute", "gather"):
        if key in name:
            return key
    return name.split("(")[0][-40:]


by = collections.defaultdict(list)
for r in step:
    by[r["stream"]].append(r)
for st, rs in sorted(by.items(), key=lambda kv: kv[1][0]["s"]):
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"stream {st}: first {(rs[0]['s'] - t0) / 1e6:7.2f}  last end {(max(r['e'] for r in rs) - t0) / 1e6:7.2f} ms  "
          f"kernel time {busy / 1e6:7.2f} ms  launches {len(rs)}")
    # phases: merge consecutive launches whose families were seen within the same 2 ms window
    seg = None
    segs = []
    for r in rs:
        f = family(r["name"])
        if seg is not None and (f in seg["fams"] or r["s"] - seg["e"] < 0.3e6 and len(seg["fams"]) < 6):
            seg["fams"][f] += r["e"] - r["s"]
            seg["e"] = max(seg["e"], r["e"])
            seg["n"] += 1
        else:
            seg = dict(s=r["s"], e=r["e"], n=1, fams=collections.Counter({f: r["e"] - r["s"]}))
            segs.append(seg)
    for g in segs:
        if g["e"] - g["s"] < 0.5e6:
            continue
        top = ", ".join(f"{k} {v / 1e6:.1f}" for k, v in g["fams"].most_common(4))
        print(f"      {(g['s'] - t0) / 1e6:7.2f} -> {(g['e'] - t0) / 1e6:7.2f} ms  {g['n']:5d} launches  busy "
              f"{sum(g['fams'].values()) / 1e6:6.1f} ms  [{top}]")
