set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/lintrace
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof -o l --output-format csv -- python3 tools/quick_linalg.py inv 14336 > $O/run.log 2>&1 || { tail -20 $O/run.log; exit 1; }
tail -1 $O/run.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/lintrace/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last factorisation: from the last flip_pad kernel on
last = max(i for i, r in enumerate(rows) if "flip_pad" in r["Kernel_Name"])
run = rows[last:]
t0 = int(run[0]["Start_Timestamp"])
big = [(r["Kernel_Name"].split("(")[0][-40:], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "")) for r in run if "gemm_split" in r["Kernel_Name"]]
for b in big:
    print(f"{b[0]:42s} at {b[1]:9.1f} us  dur {b[2]:8.1f} us  grid {b[3]}")
PY
rm -rf $O/prof
