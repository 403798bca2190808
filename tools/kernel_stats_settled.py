"""Per-kernel duration statistics from a rocprofv3 --kernel-trace csv that a reader can check the bench line against.

rocprofv3's own --stats summary averages over EVERY dispatch of a run, including the settle iterations in which the
hash-grid backward's record queues still grow (an aggregation launch of 10 ms among 300-us ones: round 3's tracked csv read
409 us on average where the bench's HIP events said 311).  Here every kernel name gets calls / median / p10 / p90 / mean of
the middle 80 % / min / max in microseconds - the median does not move with a dozen outliers - and, at the end, the roofline
fraction of the hash-grid launches recomputed from the medians alone (2328 B/point x 2^20 points over forward +
aggregation + owner pass WITHOUT the table's AdamW step - the launch configuration bench.py's `roofline` block times).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline ...
    python tools/kernel_stats_settled.py gpurun_out/kt > profiles/r04_bench_n1_kernel_stats.csv
"""
import collections, csv, glob, sys

root = sys.argv[1]
points = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
d = collections.defaultdict(list)
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "")


rows = []
for name, v in d.items():
    v.sort()
    n = len(v)
    mid = v[n // 10 : n - n // 10] if n >= 10 else v
    rows.append((sum(v), short(name), n, v[n // 2], v[n // 10], v[min(n - 1, (9 * n) // 10)], sum(mid) / len(mid), v[0], v[-1]))
rows.sort(reverse=True)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "median_us", "p10_us", "p90_us", "mean_mid80_us", "min_us", "max_us", "total_ms"])
for tot, name, n, med, p10, p90, mm, mn, mx in rows:
    w.writerow([name, n, f"{med:.2f}", f"{p10:.2f}", f"{p90:.2f}", f"{mm:.2f}", f"{mn:.2f}", f"{mx:.2f}", f"{tot / 1e3:.3f}"])


def med_of(pred):
    c = [r for r in rows if pred(r[1])]
    return max(c, key=lambda r: r[2])[3] if c else None  # the instantiation launched most often


fwd = med_of(lambda k: k.startswith("hashgrid_fwd_cloud"))
agg = med_of(lambda k: k.startswith("hashgrid_bwd_aggregate"))
own = med_of(lambda k: k.startswith("hashgrid_bwd_owner") and k.split(">")[0].replace(" ", "").endswith("false"))  # <F, COALESCED, ADAM = false>
own_adam = med_of(lambda k: k.startswith("hashgrid_bwd_owner") and k.split(">")[0].replace(" ", "").endswith("true"))
table_params = int(sys.argv[3]) if len(sys.argv) > 3 else 7854240  # the headline table (L=16, F=2, T=2^19): AdamW moves 28 B per parameter
if fwd and agg and (own or own_adam):
    gb = 2328 * points / 1e9
    if own_adam:  # the PRODUCT launches (round 5: what bench.py's `roofline` times): the owner pass takes the table's AdamW step
        us = fwd + agg + own_adam
        gb_a = gb + 28 * table_params / 1e9
        print(f"# hash-grid roofline from the medians above, product launches: ({fwd:.1f} + {agg:.1f} + {own_adam:.1f}) us = {us:.1f} us; "
              f"2328 B/point x {points} points + 28 B x {table_params} table parameters (AdamW inside the owner launch) = {gb_a:.3f} GB "
              f"-> {gb_a / (us * 1e-6) / 1e3:.3f} TB/s = frac {gb_a / (us * 1e-6) / 8e3:.4f} of 8 TB/s; with the 2328 B/point alone "
              f"({gb:.3f} GB): frac_8d_bytes_only {gb / (us * 1e-6) / 8e3:.4f}")
    if own:
        us = fwd + agg + own
        print(f"# ... with the owner pass WITHOUT the optimizer ({own:.1f} us; the Python-issued variant): ({fwd:.1f} + {agg:.1f} + {own:.1f}) us "
              f"-> frac {gb / (us * 1e-6) / 8e3:.4f}")
