"""HBM-side bytes of the SYRK launches from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over
`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline`:
    python tools/syrk_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
FETCH_SIZE counts kilobytes with every 128-byte read request counted as 64 bytes on gfx950, hence doubled."""
import csv, json, sys


def per_launch(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if "syrk_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    by = {}
    for r in rows:
        by[int(r["Dispatch_Id"])] = by.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
assert len(fetch) == len(write) and fetch, (len(fetch), len(write))
launches = [dict(fetch_size_kb=f, write_size_kb=w, hbm_bytes=(2 * f + w) * 1024) for f, w in zip(fetch, write)]
out = dict(kernel="syrk_kernel (bench.py step: 1 launch C=14336, 3 launches C=4096, T=262144)",
           method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 1 "
                  "--warmup 0 --no-cpu-baseline`; FETCH_SIZE (KB) doubled (gfx950 counts 128-B read requests as 64 B), "
                  "WRITE_SIZE (KB) as is",
           launches=launches, avg_hbm_bytes_per_launch=sum(l["hbm_bytes"] for l in launches) / len(launches))
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out)[:600])
