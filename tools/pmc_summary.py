#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter CSVs per kernel (per dispatch) for the encoder kernel."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{out}/*/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            short = ("encoder_typed_x3" if "encoder_typed" in k and ", true>" in k else
                     "encoder_typed" if "encoder_typed" in k else "encoder_fused" if "encoder_fused" in k else
                     "plan_stats" if "plan_stats" in k else "plan_chunks" if "plan_chunks" in k else
                     "wide_update" if "wide_update" in k else "wide_message" if "wide_message" in k else
                     "wide_reduce" if "wide_reduce" in k else
                     "gated_update_bwd_wide16" if "gated_update_bwd_wide16" in k else
                     "gated_update_wide16" if "gated_update_wide16" in k else
                     "bmm_message_typed_bwd_mfma" if "bmm_message_typed_bwd_mfma" in k else
                     "bmm_message_typed_seg_mfma" if "bmm_message_typed_seg_mfma" in k else
                     "strided_gemm_splitk_big" if "strided_gemm_splitk_big" in k else
                     "reduce_scatter" if "reduce_scatter_kernel" in k else None)
            if short:
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kern, cs in acc.items():
    print(f"== {kern}")
    for name in sorted(cs):
        v = cs[name]
        print(f"  {name:32s} mean/dispatch {sum(v) / len(v):16.1f}  (n={len(v)})")

import json
res = {}
for kern, cs in acc.items():
    mean = {k: sum(v) / len(v) for k, v in cs.items()}
    entry = {"counters_mean_per_dispatch": mean}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 128-B requests at 64 B: double it
        entry["hbm_bytes_per_launch"] = (2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
        entry["mfma_pipe_busy_frac"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (mean["GRBM_GUI_ACTIVE"] / 8.0)
    res[kern] = entry
with open(f"{out}/pmc.json", "w") as fh:
    json.dump(res, fh, indent=1)
