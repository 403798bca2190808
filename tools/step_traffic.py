"""HBM bytes of one training iteration from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short bench.py run:
sum over all kernels / number of iterations, plus the per-kernel split.   step_traffic.py <fetch dir> <write dir> <iterations>"""
import collections, csv, glob, json, re, sys


def per_kernel(root, counter):
    d = collections.defaultdict(float)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
                d[name] += float(r["Counter_Value"])
    return d


fetch, write, n = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
names = sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0)))
rows = {k: {"read_MB": round(2 * fetch.get(k, 0) * 1024 / n / 1e6, 1), "write_MB": round(write.get(k, 0) * 1024 / n / 1e6, 1)} for k in names}
total = sum(v["read_MB"] + v["write_MB"] for v in rows.values())
print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python bench.py --steps 10 --warmup 2; "
                            "KiB counters, FETCH_SIZE doubled per MI355X_MICROARCH.md; divided by the iterations run (setup kernels included)",
                  "iterations": n, "HBM_MB_per_iteration": round(total, 1), "kernels": {k: v for k, v in list(rows.items())[:24]}}, indent=1))
