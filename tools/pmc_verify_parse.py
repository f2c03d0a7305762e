"""Turn the two PMC passes of tools/pmc_verify.sh into profiles/pmc_verify_latest.json.

Per verify launch:  HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE  (both counters are in KiB; FETCH_SIZE is doubled because
gfx950's rocprofv3 reports half of a wide coalesced streaming read, MI355X_MICROARCH.md "HBM").  The launches of the
decode loop are the mb_verify_kernel dispatches of the command (every one of them belongs to the decode loop); their algorithmic bytes
are valid_rows * V * element_size as dumped by bench.py (JF_DUMP_LAUNCHES)."""
import csv
import glob
import json
import sys
from pathlib import Path


def per_dispatch(dirname: str, counter: str):
    files = glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {dirname}")
    acc = {}
    for row in csv.DictReader(open(files[0])):
        if "mb_verify_kernel" not in row["Kernel_Name"] or row["Counter_Name"] != counter:
            continue
        k = int(row["Dispatch_Id"])
        acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"])
    return [acc[k] for k in sorted(acc)]


def main():
    out = Path(sys.argv[1])
    fetch = per_dispatch(str(out / "FETCH_SIZE"), "FETCH_SIZE")
    write = per_dispatch(str(out / "WRITE_SIZE"), "WRITE_SIZE")
    la = json.loads((out / "launches_FETCH_SIZE.json").read_text())
    lb = json.loads((out / "launches_WRITE_SIZE.json").read_text())
    assert la["valid"] == lb["valid"], "the two passes must see the same launches (same seed, same steps)"
    n = len(la["valid"])
    fetch, write = fetch[-n:], write[-n:]
    assert len(fetch) == n and len(write) == n, (len(fetch), len(write), n)
    alg = [v * la["V"] * la["esz"] for v in la["valid"]]
    hbm = [2.0 * f * 1024.0 + w * 1024.0 for f, w in zip(fetch, write)]
    res = dict(traffic_over_algorithmic=sum(hbm) / sum(alg), launches=n, algorithmic_bytes=sum(alg), hbm_bytes=sum(hbm),
               fetch_kib_raw=sum(fetch), write_kib_raw=sum(write),
               rows_valid_mean=sum(la["valid"]) / n, rows_launched_mean=sum(la["rows"]) / n,
               method="tools/pmc_verify.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                      "`bench.py --steps 12 --warmup 2 --no-scripted --no-shapes --no-sections --cpu-baseline-seconds 0`; per launch HBM bytes = "
                      "(2*FETCH_SIZE + WRITE_SIZE) KiB (FETCH_SIZE doubled: gfx950 reports half of a wide coalesced stream, "
                      "MI355X_MICROARCH.md HBM section); the decode loop's launches = the mb_verify_kernel dispatches; "
                      "algorithmic bytes = draft-carrying rows * V * 2")
    (out / "pmc_verify.json").write_text(json.dumps(res, indent=1))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
