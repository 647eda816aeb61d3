#!/usr/bin/env python3
"""Turn the rocprofv3 counter-collection CSVs of the SQ passes (tools/profile_round.sh) into
  profiles/<label>_sq_counters.csv   one row per kernel, per-launch averages of every collected counter
  profiles/sq_counters.json           the same for the two blend kernels, keyed by the library build hash (read by bench.py for
                                      `roofline_valu`)
Units (MI355X_MICROARCH.md, 's_memtime tick vs SQ PMC units'): SQ_INSTS_* count wave-instructions; SQ_WAVE_CYCLES, SQ_WAIT_*,
SQ_ACTIVE_INST_* count QUAD-cycles (4 shader cycles) summed over waves; SQ_BUSY_CYCLES quad... per shader engine; GRBM_GUI_ACTIVE
counts shader cycles summed over the 8 XCDs (effective clock = GRBM_GUI_ACTIVE / 8 / kernel time).
usage: summarize_sq.py LABEL CONFIG OUTDIR counter_csv [counter_csv ...]"""
import csv
import hashlib
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name).strip()


def main():
    label, config, outdir = sys.argv[1:4]
    acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for path in sys.argv[4:]:
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
    counters = sorted({c for k in acc for c in acc[k]})
    keep = [k for k in acc if not k.startswith("at::") and "rocclr" not in k]
    keep.sort(key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0.0) / max(1, cnt[k].get("SQ_WAVE_CYCLES", 1)))
    with open(os.path.join(outdir, f"{label}_sq_counters.csv"), "w") as f:
        f.write("kernel," + ",".join(counters) + "\n")
        for k in keep:
            f.write('"' + k + '",' + ",".join(f"{acc[k][c] / cnt[k][c]:.6g}" if cnt[k].get(c) else "" for c in counters) + "\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.environ.get("GSR_LIB", os.path.join(root, "3dgs-native_amd", "libgsr_hip.so"))
    with open(lib, "rb") as f:
        build = hashlib.sha256(f.read()).hexdigest()[:16]
    path = os.path.join(outdir, "sq_counters.json")
    d = json.load(open(path)) if os.path.exists(path) else {}
    if d.get("_build") != build:
        d = {}
    d["_build"] = build
    d["_round"] = f"profiles/{label}_sq_counters.csv"
    d["_how"] = ("rocprofv3 --pmc passes (kernel-trace only, 8 SQ counters each) of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline "
                 "--no-stage-events`; per-launch averages; SQ_INSTS_* = wave-instructions, SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* = quad-cycles")
    sel = {}
    for k in keep:
        stage = "blend_bwd" if k.startswith("blend_backward_splat_kernel") else "blend_fwd" if k.startswith("blend_forward_kernel") else None
        if stage:
            sel[stage] = {c: acc[k][c] / cnt[k][c] for c in counters if cnt[k].get(c)}
            sel[stage]["kernel"] = k
    d[config] = sel
    json.dump(d, open(path, "w"), indent=1)
    for st, v in sel.items():
        print(st, {c: f"{x:.4g}" for c, x in v.items() if c != "kernel"})


if __name__ == "__main__":
    main()
