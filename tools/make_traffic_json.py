"""profiles/r01_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/prof_hashgrid.py P.
Usage: make_traffic_json.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> > profiles/r01_pmc_traffic.json"""
import collections, csv, glob, json, sys


def avg(root, counter):
    d = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                for k in ("hashgrid_bwd_aggregate", "hashgrid_bwd_owner", "hashgrid_fwd"):
                    if k in r["Kernel_Name"]:
                        d[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


fetch, write = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
import os
commit = os.environ.get("NESVOR_COMMIT", "unknown")  # the gpurun snapshot carries no .git: the caller passes the commit
out = {
    "commit": commit,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python tools/prof_hashgrid.py P ; "
              "N=2^20, L=16, F=2, T=2^19, PSF-cloud points, input grad on",
    "unit": "bytes per launch",
    "note": "FETCH_SIZE, WRITE_SIZE are reported in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
            "requests at 64 B); WRITE_SIZE matches known byte counts exactly (hashgrid_fwd writes 131072 KiB = N*32*4 B)",
}
for k in fetch:
    out[k] = {"FETCH_SIZE_KiB": round(fetch[k], 1), "WRITE_SIZE_KiB": round(write.get(k, 0.0), 1),
              "traffic_bytes": int(2 * fetch[k] * 1024 + write.get(k, 0.0) * 1024)}
if "hashgrid_bwd_aggregate" in out and "hashgrid_bwd_owner" in out:
    out["hashgrid_bwd"] = {"traffic_bytes": out["hashgrid_bwd_aggregate"]["traffic_bytes"] + out["hashgrid_bwd_owner"]["traffic_bytes"]}
print(json.dumps(out, indent=1))
