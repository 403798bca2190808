"""Print the kernel timeline of the last few training steps from a rocprofv3 --kernel-trace csv:
per kernel the start offset, duration and the idle gap before it.  Usage:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --steps 6 --warmup 3 --no-cpu
  python tools/step_timeline.py gpurun_out/tl [anchor-substring]"""
import csv
import glob
import sys

root = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "psf_transform_fwd"
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if anchor in r[2]]
# the timed region: take the last-but-one anchor .. last anchor
a, b = starts[-2], starts[-1]
t0 = rows[a][0]
prev_end = rows[a][0]
busy = 0
for s, e, name in rows[a:b]:
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    short = short.replace("rocprim::ROCPRIM_400001_NS::detail::", "rocprim::")[:90]
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short))
    busy += e - s
    prev_end = max(prev_end, e)
print("step span %.1f us, busy %.1f us, kernels %d" % ((rows[b][0] - t0) / 1e3, busy / 1e3, b - a))
