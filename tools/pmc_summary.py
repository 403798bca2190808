"""Average PMC counter values per kernel from rocprofv3 --pmc csv output dirs.  Usage: pmc_summary.py <substr> <dir>..."""
import collections, csv, glob, sys
sub = sys.argv[1]
for root in sys.argv[2:]:
    d = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(d):
        print("%-28s %16.0f  (n=%d)" % (k, sum(d[k]) / len(d[k]), len(d[k])))
