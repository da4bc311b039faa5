"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into per-kernel HBM traffic per launch.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
WRITE_SIZE is exact for streaming stores and float atomics; on gfx950 FETCH_SIZE reports exactly half of a wide
(16 B/lane) coalesced streaming read and is uncalibrated for other access widths, so both the raw and the doubled
read figure are recorded and `traffic` uses the doubled one only for kernels whose loads are 16 B per lane."""
import collections, csv, json, sys

WIDE = ("gemm_kernel", "bn_relu_apply", "bn_bwd_reduce", "narrow_", "three_interpolate_kernel", "tig_reduce")


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f_kib, nf = fetch.get(k, (0.0, 0))
    w_kib, nw = write.get(k, (0.0, 0))
    wide = any(t in k for t in WIDE)
    rd = f_kib * 1024 * (2 if wide else 1)
    out[k] = {"launches": max(nf, nw), "fetch_size_bytes_raw": f_kib * 1024, "write_size_bytes": w_kib * 1024,
              "read_correction": 2 if wide else 1, "traffic_bytes_per_launch": rd + w_kib * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"])[:12]:
    print(f"{k[:70]:70s} x{v['launches']:4d}  read {v['fetch_size_bytes_raw'] * v['read_correction'] / 1e6:9.2f} MB  write {v['write_size_bytes'] / 1e6:9.2f} MB")
