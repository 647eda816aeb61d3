#!/usr/bin/env python3
"""Turn two rocprofv3 counter-collection CSVs (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass) into
profiles/<label>_pmc_fetch_write_summary.csv and profiles/pmc_traffic.json.
HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and on gfx950 FETCH_SIZE reports half the bytes of
the wide reads these kernels issue (/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section).
usage: summarize_pmc.py LABEL CONFIG fetch.csv write.csv OUTDIR"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

STAGE_OF = [("preprocess_kernel", "preprocess"), ("blend_forward_kernel", "blend_fwd"), ("blend_backward_splat_kernel", "blend_bwd"),
            ("geom_backward_kernel", "geom_bwd"), ("expand_kernel", "expand"), ("ranges_kernel", "ranges"), ("scan_reduce_kernel<0>", "scan"),
            ("scan_final_kernel<0>", "scan"), ("scan_reduce_kernel<2>", "depth_scan"), ("scan_final_kernel<2>", "depth_scan"),
            ("pack_records_kernel", "bwd_prep"), ("fillBufferAligned", "bwd_prep"), ("expand_blocks_kernel", "expand"),
            ("scan_ctl_hist_kernel", "scan"), ("depth_block_offsets_kernel", "depth_scan"), ("ranges_fixup_kernel", "tile_sort")]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name).strip()


def read(path, counter):
    tot, calls = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            k = short(row["Kernel_Name"])
            tot[k] += float(row["Counter_Value"])
            calls[k] += 1
    return tot, calls


def radix_stage(kernel):
    """radix_scatter_kernel<ITEMS, BITS, type, CARRY, FINAL, DEPTH, THREADS> / radix_hist_kernel<ITEMS, BITS, type, DEPTH>: the DEPTH
    flag tells the N-item depth passes from the D-item tile passes (chunk sizes and thread counts changed in round 4)."""
    args = [a.strip() for a in kernel[kernel.index("<") + 1:kernel.rindex(">")].split(",")] if "<" in kernel else []
    flag = args[5] if kernel.startswith("radix_scatter") and len(args) > 5 else args[3] if kernel.startswith("radix_hist") and len(args) > 3 else ""
    return "depth_sort" if flag in ("true", "1") else "tile_sort"


def main():
    label, config, fetch_csv, write_csv, outdir = sys.argv[1:6]
    ft, fc = read(fetch_csv, "FETCH_SIZE")
    wt, wc = read(write_csv, "WRITE_SIZE")
    steps = fc.get("preprocess_kernel", 0)
    if not steps:
        raise SystemExit("no preprocess_kernel dispatches in the FETCH_SIZE pass")
    rows, stage = [], defaultdict(float)
    for k in sorted(ft, key=lambda k: -(2 * ft[k] + wt.get(k, 0.0))):
        f_avg, w_avg = ft[k] / fc[k], wt.get(k, 0.0) / max(1, wc.get(k, 0))
        rows.append((k, fc[k] / steps, f_avg, w_avg, (2 * f_avg + w_avg) * 1024))
        per_step = (2 * ft[k] + wt.get(k, 0.0)) * 1024 / steps
        st = next((s for pat, s in STAGE_OF if pat in k), None)
        if st is None and k.startswith("radix_"):
            st = radix_stage(k)
        if st:
            stage[st] += per_step
    with open(os.path.join(outdir, f"{label}_pmc_fetch_write_summary.csv"), "w") as f:
        f.write("kernel,launches_per_step,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_bytes_corrected_per_launch\n")
        for k, n, fa, wa, b in rows:
            f.write(f"\"{k}\",{n:g},{fa:.0f},{wa:.0f},{b:.0f}\n")
    path = os.path.join(outdir, "pmc_traffic.json")
    d = json.load(open(path)) if os.path.exists(path) else {}
    d["_how"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python bench.py --steps 3 --warmup 1 "
                 "--no-cpu-baseline --no-stage-events` on MI355X; bytes per step = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the "
                 "stage's kernels (gfx950 FETCH_SIZE reports half the bytes of wide reads: MI355X_MICROARCH.md, HBM); tools/summarize_pmc.py")
    d["_round"] = f"profiles/{label}_*"
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.environ.get("GSR_LIB", os.path.join(root, "3dgs-native_amd", "libgsr_hip.so"))
    with open(lib, "rb") as f:      # bench.py quotes these bytes only for the build they were measured on
        d["_build"] = hashlib.sha256(f.read()).hexdigest()[:16]
    d[config] = {k: int(v) for k, v in sorted(stage.items())}
    json.dump(d, open(path, "w"), indent=1)
    print(json.dumps(d[config]))


if __name__ == "__main__":
    main()
