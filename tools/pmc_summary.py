"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (averages per dispatch)."""
import collections, csv, glob, sys
pat = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(pat):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.Counter())
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:90]
        if only and only not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    for k, d in agg.items():
        print(k)
        for c, v in sorted(d.items()):
            print("    %-30s %14.5g  (avg of %d dispatches)" % (c, v / cnt[k][c], cnt[k][c]))
