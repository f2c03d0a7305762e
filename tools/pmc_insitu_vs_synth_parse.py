"""Per-dispatch counters of mb_verify_kernel from tools/pmc_insitu_vs_synth.sh, in situ (bench.py's decode step) against
synthetic (tools/verify_trace.py: the same launch over logits nobody has just written, from an idle stream)."""
import csv
import glob
import sys
from collections import defaultdict


def load(dirname):
    cnt = defaultdict(dict)          # dispatch -> counter -> value
    for f in glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "mb_verify_kernel" in row["Kernel_Name"]:
                d = int(row["Dispatch_Id"])
                cnt[d][row["Counter_Name"]] = cnt[d].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    dur = {}
    for f in glob.glob(f"{dirname}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "mb_verify_kernel" in row["Kernel_Name"]:
                dur[int(row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), int(row.get("Grid_Size", row.get("Grid_Size_X", 0)) or 0))
    return cnt, dur


def main():
    out = sys.argv[1]
    for i in range(1, 5):
        for mode in ("insitu", "synth"):
            cnt, dur = load(f"{out}/{mode}_{i}")
            ds = sorted(d for d in cnt if d in dur)
            if not ds:
                print(f"pass {i} {mode}: no dispatches found")
                continue
            # the largest launches of the run (64 prompts: the last third by grid/duration are the steady-state ones)
            ds = sorted(ds, key=lambda d: dur[d][0])[len(ds) // 2:]
            names = sorted({k for d in ds for k in cnt[d]})
            t_us = sum(dur[d][0] for d in ds) / len(ds) / 1e3
            line = f"pass {i} {mode:7s} {len(ds):3d} launches, mean duration {t_us:7.1f} us (profiled)"
            for nme in names:
                v = sum(cnt[d].get(nme, 0.0) for d in ds) / len(ds)
                line += f"  {nme} {v:.4g}"
                if nme == "GRBM_GUI_ACTIVE":
                    line += f" (= {v / (t_us * 1e3) :.3f} cycles per ns: the clock the launch ran at, GHz)"
                else:
                    line += f" ({v / t_us:.4g} per us)"
            print(line)


if __name__ == "__main__":
    main()
