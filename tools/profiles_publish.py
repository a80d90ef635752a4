"""Copy the evidence of one tools/final_collect.sh call (gpurun_out/<tag>*) into profiles/<round>/ under
the names bench.py and the READMEs refer to.   usage: tools/profiles_publish.py <tag> <round-dir> <prefix>"""
import glob, json, os, shutil, sys

tag, rnd, pre = sys.argv[1], sys.argv[2], sys.argv[3]
src = os.path.join("gpurun_out", tag)
dst = os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


for name, out in (("bench.json", pre + "_bench.json"), ("bench_under_rocprof.json", pre + "_bench_under_rocprof.json")):
    json.dump(last_json(os.path.join(src, name)), open(os.path.join(dst, out), "w"), indent=1)
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, pre + "_kernel_stats.csv"))
for f in glob.glob(os.path.join(src, "stats_inf16k", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, pre + "_inflate16k_kernel_stats.csv"))
for sub, out in (("stats_c5", pre + "_config5_inflate_kernel_stats.csv"), ("stats_c3", pre + "_config3_kernel_stats.csv")):
    for f in glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, out))
for name, out in (("config5_under_rocprof.json", pre + "_config5_under_rocprof.json"), ("config3_under_rocprof.json", pre + "_config3_under_rocprof.json")):
    if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)) > 10:
        json.dump(last_json(os.path.join(src, name)), open(os.path.join(dst, out), "w"), indent=1)
if os.path.exists(os.path.join(src, "inflate16k_under_rocprof.json")):
    json.dump(last_json(os.path.join(src, "inflate16k_under_rocprof.json")),
              open(os.path.join(dst, pre + "_inflate16k_under_rocprof.json"), "w"), indent=1)
for name, out in (("soak.txt", pre + "_soak.txt"), ("inflate_crossover.txt", "inflate_crossover.txt"),
                  ("gpu_tests.txt", pre + "_gpu_tests.txt")):
    if os.path.exists(os.path.join(src, name)):
        lines = [ln for ln in open(os.path.join(src, name)).read().splitlines() if "amdgpu.ids" not in ln]
        open(os.path.join(dst, out), "w").write("\n".join(lines[-12:] if name != "inflate_crossover.txt" else lines) + "\n")
lz = {}
for cfg, key in (("c2", "lz77_default_16384x65536_text"), ("c3", "lz77_default_4096x262144_text")):
    p = os.path.join(src, "traffic_%s.json" % cfg)
    if os.path.exists(p) and os.path.getsize(p) > 10:
        txt = open(p).read()
        lz[key] = json.loads(txt[txt.index("{"):])  # (the collector prints the queue split in front)
if lz:
    json.dump(lz, open(os.path.join(dst, "lz77_traffic.json"), "w"), indent=1)
inf = {}
for fname, n, label, rule in (("traffic_inflate.json", 131072, "8 GiB", "raw"), ("traffic_inflate_16k.json", 16384, "1 GiB", "x2")):
    p = os.path.join(src, fname)
    if not (os.path.exists(p) and os.path.getsize(p) > 10):
        continue
    raw = json.load(open(p))
    kern = {k: v for k, v in raw.items() if isinstance(v, dict)}
    name = max(kern, key=lambda k: kern[k]["FETCH_SIZE_bytes_raw"]) if kern else None
    if not name:
        continue
    b = last_json(os.path.join(src, "bench.json"))
    leg = b.get("extra", {}).get("config5_inflate_8GiB" if n == 131072 else "inflate_1GiB_16384_streams", {})
    algo = n * 65536 + int(leg.get("config", {}).get("compressed_bytes_per_gpu", 0))
    fetch = kern[name]["FETCH_SIZE_bytes_raw"] * (2 if rule == "x2" else 1)
    tot = fetch + kern[name]["WRITE_SIZE_bytes"]
    inf["inflate_%dx65536_text" % n] = {
        "workload": "%d x 65536 B S-text streams (%s out)" % (n, label), "kernel": name,
        "build_id": raw.get("build_id"), "git_head": raw.get("git_head"), "algorithmic_bytes": algo,
        "FETCH_SIZE_bytes_raw": kern[name]["FETCH_SIZE_bytes_raw"], "WRITE_SIZE_bytes": kern[name]["WRITE_SIZE_bytes"],
        "read_rule": "raw: single 64-B sector requests (one per history fetch of a lane)" if rule == "raw" else
                     "x2 (upper bound): the compressed input is read in coalesced 256-B rows (gfx950 FETCH_SIZE reports half for "
                     "those), the far-history bytes in single sectors (reported in full); the mix is not separable, so all of it is doubled",
        "hbm_bytes_per_launch": tot, "over_algorithmic": round(tot / algo, 2) if algo else None,
        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/inflate_traffic.sh)"}
if inf:
    json.dump(inf, open(os.path.join(dst, "inflate_traffic.json"), "w"), indent=1)
print("published", sorted(os.listdir(dst)))
