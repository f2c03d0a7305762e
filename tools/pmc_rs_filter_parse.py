"""tools/pmc_rs_filter.sh: per-dispatch FETCH_SIZE / WRITE_SIZE (KiB) of the filter's and the softmax stream's kernels -> bytes per call over the
algorithmic bytes of one read of the logits (1 984 x 152 064 x 2).  FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced stream)."""
import csv
import glob
import sys

ALG = 1984 * 152064 * 2


def per_kernel(dirname, counter):
    files = glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {dirname}")
    acc = {}
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
        acc.setdefault(name, {}).setdefault(int(row["Dispatch_Id"]), 0.0)
        acc[name][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    return {k: [v[d] for d in sorted(v)] for k, v in acc.items()}


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
print(f"## {sys.argv[3]}   (algorithmic bytes of one read of the logits: {ALG / 1e6:.1f} MB)")
for name in sorted(f):
    if not ("rs_filter" in name or "rs_probs" in name):
        continue
    fs, ws_ = f[name][1:], w.get(name, [0.0] * len(f[name]))[1:]          # (the first call warms up)
    if not fs:
        continue
    fb = 2.0 * 1024.0 * sum(fs) / len(fs)
    wb = 1024.0 * sum(ws_) / max(len(ws_), 1)
    print(f"   {name:28s} {len(fs)} calls   fetched {fb / 1e6:8.1f} MB = {fb / ALG:5.3f} x   written {wb / 1e6:7.2f} MB")
