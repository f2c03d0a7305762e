#!/usr/bin/env python3
"""oracle/verify_bench.py — TEST INFRASTRUCTURE / CPU BASELINE ONLY (never imported by the product).

Times oracle/verify_ref.c's ``ref_argmax_rows`` (the verify body's HBM-heavy op, MB:476) over bf16 logits of a given launch
shape, as its own process so that the OpenMP runtime starts with an explicit placement: bench.py runs it with
``OMP_PROC_BIND=close OMP_PLACES=cores`` — every thread pinned to one core, consecutive threads on consecutive cores, i.e.
the static row partition of the fill and of the scan lands each row's pages (first touch) on the NUMA node of the core that
scans it.  Prints one JSON line.

    python oracle/verify_bench.py ROWS V SECONDS
"""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np


def main():
    rows, V, budget_s = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    lib = C.CDLL(str(Path(__file__).resolve().parent / "_build" / "libjf_oracle.so"))
    lib.ref_argmax_rows.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    lib.ref_num_threads.restype = C.c_int
    lib.ref_fill_rows_bf16.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint32]
    x = np.empty((rows, V), dtype=np.uint16)                  # untouched pages: placed by the threads that fill (and later scan) them
    lib.ref_fill_rows_bf16(x.ctypes.data, rows, V, V, 1234)
    out = np.zeros(rows, dtype=np.int64)
    lib.ref_argmax_rows(x.ctypes.data, 1, rows, V, V, out.ctypes.data)      # warm-up
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        lib.ref_argmax_rows(x.ctypes.data, 1, rows, V, V, out.ctypes.data)
        reps += 1
    dt = (time.perf_counter() - t0) / max(reps, 1)
    nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")]) if os.path.isdir("/sys/devices/system/node") else None
    print(json.dumps(dict(us_per_call=dt * 1e6, gbs=rows * V * 2 / dt / 1e9, rows=rows, threads=int(lib.ref_num_threads()), reps=reps,
                          seconds=budget_s, numa_nodes=nodes,
                          placement=f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND', 'unset')} OMP_PLACES={os.environ.get('OMP_PLACES', 'unset')} "
                                    f"OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS', 'unset')}: "
                                    + ("threads pinned in order, " if os.environ.get("OMP_PROC_BIND") else "the runtime's default placement, ")
                                    + "rows first-touched by the thread that scans them (static schedule in fill and scan), one warm-up call "
                                      "before the timed repetitions",
                          what="oracle/verify_ref.c ref_argmax_rows (C + OpenMP) over bf16 logits of the bench's launch shape")))


if __name__ == "__main__":
    main()
