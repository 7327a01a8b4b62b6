set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fql
rm -rf gpurun_out/fql/prof
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --stats -d gpurun_out/fql/prof -o f -- python3 tools/fq_variants_rate.py uint4_post_g128_f16s > gpurun_out/fql/rate.log 2> gpurun_out/fql/prof.err
cat gpurun_out/fql/rate.log
python3 tools/prof_db.py $(find gpurun_out/fql/prof -name "*.db" | head -1) 12 | cut -c1-140
python3 - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/fql/prof/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
view = [n for n in names if n.startswith("kernels")] or [n for n in names if "kernel_dispatch" in n]
print(view[:3])
try:
    rows = con.execute(f"select start, end, name from {view[0]} order by start").fetchall()
    fk = [r for r in rows if "fused_int" in r[2]]
    gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(fk, fk[1:])]
    print("fused_int launches", len(fk), "gaps us:", [round(g, 1) for g in gaps[-12:]])
    print("durations us:", [round((r[1] - r[0]) / 1e3, 1) for r in fk[-8:]])
except Exception as e:
    print("query failed", e)
PY
rm -rf gpurun_out/fql/prof
