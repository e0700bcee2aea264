#!/usr/bin/env python3
"""Summarise rocprofv3 SQ counter passes (tools/sq_pass_r05.sh) into per-kernel averages per launch.

    python tools/sq_summary.py <pass A dir> <pass B dir> profiles/r05_sq_w64.json

Derived figures (formulas stated because ROCm 7.2 ships no gfx950 derived-counter section -- MI355X_MICROARCH.md, rocprofv3 PMC slots):
  kernel_cycles   = GRBM_GUI_ACTIVE / 8                      (rocprofv3 sums the 8 XCDs)
  mfma_busy       = SQ_VALU_MFMA_BUSY_CYCLES / (kernel_cycles * 256 CUs * 4 SIMDs)      (the gfx94x MfmaUtil formula)
  mfma_flops      = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512       (the counter's unit is 512 operations)
  wave-cycle split: WAIT_ANY (parked on s_waitcnt / barrier), WAIT_INST_ANY (issue stall), ACTIVE_INST_ANY, each / SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import json
import os
import sys


def load(d):
    f = (glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv")))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    a_dir, b_dir, out = sys.argv[1:4]
    A, B = load(a_dir), load(b_dir)
    res = []
    for k, ca in A.items():
        cb = B.get(k, {})
        avg = lambda c, n: (sum(c[n]) / len(c[n])) if c.get(n) else None      # noqa: E731
        gui, busy, mops = avg(ca, "GRBM_GUI_ACTIVE"), avg(ca, "SQ_VALU_MFMA_BUSY_CYCLES"), avg(ca, "SQ_INSTS_VALU_MFMA_MOPS_BF16")
        wave, wany, winst, act = (avg(cb, n) for n in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"))
        cyc = gui / 8.0 if gui else None
        r = {"kernel": k[0], "grid_threads": k[1], "workgroup": k[2], "launches": len(next(iter(ca.values()))),
             "GRBM_GUI_ACTIVE": gui, "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CYCLES": avg(ca, "SQ_BUSY_CYCLES"),
             "SQ_INSTS_VALU_MFMA_MOPS_BF16": mops, "SQ_WAVE_CYCLES": wave, "SQ_WAIT_ANY": wany, "SQ_WAIT_INST_ANY": winst,
             "SQ_ACTIVE_INST_ANY": act,
             "kernel_cycles": cyc,
             "mfma_busy": (busy / (cyc * 256 * 4)) if (busy is not None and cyc) else None,
             "mfma_flops": mops * 512 if mops is not None else None,
             "wait_any_frac": (wany / wave) if (wany is not None and wave) else None,
             "wait_inst_frac": (winst / wave) if (winst is not None and wave) else None,
             "active_inst_frac": (act / wave) if (act is not None and wave) else None}
        res.append(r)
    res.sort(key=lambda r: -((r["SQ_VALU_MFMA_BUSY_CYCLES"] or 0) * r["launches"]))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({"commit": os.environ.get("IMMTSF_PMC_COMMIT"), "csrc_sha": bench.csrc_sha(),
               "windows_per_gpu": int(os.environ.get("IMMTSF_PMC_WINDOWS", "64")), "config": "cfg2",
               "source": "rocprofv3 --pmc <group> (one run per group: A = SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 "
                         "GRBM_GUI_ACTIVE, B = SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY) -- python3 bench.py --steps 4 "
                         "--warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph",
               "formulas": __doc__.split("Derived figures", 1)[1], "kernels": res[:40]}, open(out, "w"), indent=1)
    for r in res[:10]:
        f = lambda x: "-" if x is None else "%.3f" % x      # noqa: E731
        print(r["kernel"][:64], r["grid_threads"], "mfma_busy", f(r["mfma_busy"]), "GF", f((r["mfma_flops"] or 0) / 1e9),
              "wait_any", f(r["wait_any_frac"]), "wait_inst", f(r["wait_inst_frac"]), "active", f(r["active_inst_frac"]))


if __name__ == "__main__":
    main()
