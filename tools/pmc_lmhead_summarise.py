#!/usr/bin/env python
"""gpurun_out/pmcL1 (FETCH_SIZE pass) + gpurun_out/pmcL2 (WRITE_SIZE pass) of tools/pmc_lmhead.sh ->
gpurun_out/<tag>_lmhead_traffic.json (copied to profiles/): HBM bytes per launch of the LM-head logits GEMM, corrected with the calibration
kernel of known byte count that runs in the same passes (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {}
for d, cn in [("pmcL1", "FETCH_SIZE"), ("pmcL2", "WRITE_SIZE")]:
    f = glob.glob(os.path.join(root, "gpurun_out", d, "*counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != cn:
            continue
        k = "gemm" if any(x in r["Kernel_Name"] for x in ("gemm_glds", "gemm_a16", "gemm_p16", "gemm_b16")) else ("cast" if "cast_f32_bf16" in r["Kernel_Name"] else None)
        if k:
            agg[k].append(float(r["Counter_Value"]))
    res[cn] = {k: sum(v) / len(v) for k, v in agg.items()}
cast_read, cast_write = 256 * 1024 * 1024 * 4, 256 * 1024 * 1024 * 2
fcorr = cast_read / (res["FETCH_SIZE"]["cast"] * 1024)
wcorr = cast_write / (res["WRITE_SIZE"]["cast"] * 1024)
fetch = res["FETCH_SIZE"]["gemm"] * 1024 * fcorr
write = res["WRITE_SIZE"]["gemm"] * 1024 * wcorr
M = int(sys.argv[2]) if len(sys.argv) > 2 else 22784      # rows of the probed launch (tools/lmhead_probe.py)
V, VP, K = 52305, 52480, 768
out = {
    "kernel": "gemm_p16_kernel<A k-contig, B k-contig, bf16 out> (LM head logits, N = Vpad; round 6: the two-waves-per-SIMD loop)",
    "shape_MNK": [M, V, K], "computed_columns": VP,
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python3 "
               "tools/lmhead_probe.py 5   (tools/pmc_lmhead.sh, summarised by tools/pmc_lmhead_summarise.py)",
    "raw_KiB_per_launch": {"FETCH_SIZE": res["FETCH_SIZE"]["gemm"], "WRITE_SIZE": res["WRITE_SIZE"]["gemm"]},
    "calibration": {"kernel": "cast_f32_bf16_kernel over 256 Mi elements (1 GiB read, 0.5 GiB written)",
                    "raw_KiB": {"FETCH_SIZE": res["FETCH_SIZE"]["cast"], "WRITE_SIZE": res["WRITE_SIZE"]["cast"]},
                    "fetch_correction": round(fcorr, 4), "write_correction": round(wcorr, 4),
                    "note": "FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950; WRITE_SIZE is exact"},
    "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
    "traffic_bytes_per_launch": round(fetch + write),
    "algorithmic_bytes_per_launch": M * K * 2 + VP * K * 2 + M * VP * 2,
}
json.dump(out, open(os.path.join(root, "gpurun_out", f"{tag}_lmhead_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
