cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r04_ckpmc
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r04_ckpmc/f -o p --output-format csv -- python3 tools/experiments/checksum_bench.py > gpurun_out/r04_ckpmc/f.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r04_ckpmc/w -o p --output-format csv -- python3 tools/experiments/checksum_bench.py > gpurun_out/r04_ckpmc/w.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in ("f", "w"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r04_ckpmc/%s/**/*counter_collection.csv" % tag, recursive=True):
        per = collections.defaultdict(float); names = {}
        for r in csv.DictReader(open(f)):
            per[r["Dispatch_Id"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = (r["Kernel_Name"], r["Counter_Name"])
        for d, v in per.items():
            agg[names[d]].append(v)
    for (k, c), v in agg.items():
        if "checksum" in k:
            print(tag, k[:50], c, "launches", len(v), "min %.4g max %.4g (KB units: x1024 = %.4g GB max)" % (min(v), max(v), max(v) * 1024 / 1e9))
PY
