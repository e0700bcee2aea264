#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) into
per-kernel-instance HBM traffic per launch.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Units/corrections (guide, section HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly 1/2 of the
bytes of a wide (16 B/lane) coalesced stream -- every kernel here loads 16 B per lane -- so fetch bytes = 2 * FETCH_SIZE
* 1024; WRITE_SIZE is exact for 16 B/lane stores and float atomics."""
import collections
import csv
import glob
import json
import os
import sys


def load(d):
    f = (glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv")))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]))].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    fe, wr = load(fetch), load(write)
    res = []
    for k, v in fe.items():
        w = wr.get(k, [])
        res.append({"kernel": k[0], "grid_threads": k[1], "workgroup": k[2], "launches": len(v),
                    "fetch_bytes_per_launch": round(2.0 * 1024.0 * sum(v) / len(v)),
                    "write_bytes_per_launch": round(1024.0 * sum(w) / len(w)) if w else None,
                    "fetch_size_raw_kib": round(sum(v) / len(v), 2)})
    res.sort(key=lambda r: -(r["fetch_bytes_per_launch"] * r["launches"]))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({"commit": os.environ.get("IMMTSF_PMC_COMMIT"),            # tree the passes were taken at (the GPU box has no .git)
               "csrc_sha": bench.csrc_sha(),                             # content hash of the kernel sources: bench.py trusts the profile only for this build
               "windows_per_gpu": int(os.environ.get("IMMTSF_PMC_WINDOWS", "64")), "config": os.environ.get("IMMTSF_PMC_CONFIG", "cfg2"),
               "ms_per_step": float(os.environ["IMMTSF_PMC_MS"]) if os.environ.get("IMMTSF_PMC_MS") else None,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 6 "
                         "--warmup 2 --no-cpu-baseline --no-roofline --no-graph",
               "correction": "fetch bytes = 2 * FETCH_SIZE KiB * 1024 (gfx950 half-count for 16 B/lane streams); write exact",
               "kernels": res[:60]}, open(out, "w"), indent=1)
    for r in res[:12]:
        print(r["kernel"][:70], r["grid_threads"], r["fetch_bytes_per_launch"] / 1e6, "MB fetch",
              (r["write_bytes_per_launch"] or 0) / 1e6, "MB write")


if __name__ == "__main__":
    main()
