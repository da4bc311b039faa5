"""Summarise a rocprofv3 --pmc pass with SQ counters per kernel (averages per launch):

    python tools/pmc_sq.py <counter_collection.csv> [substring of the kernel names to show]

Units per /opt/skills/guides/MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over
waves, SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles (summed over SIMDs / shader engines as the counter defines)."""
import collections, csv, sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(rows, key=lambda k: -rows[k].get("SQ_WAVE_CYCLES", 0)):
    if pat not in k:
        continue
    print(k[:150])
    for c, v in sorted(rows[k].items()):
        print(f"    {c:32s} {v / cnt[k][c]:16.0f}   x{cnt[k][c]}")
