#!/usr/bin/env python
"""gpurun_out/pmc_<tag>_{mix,text}_{F,W,Q}/summary.json (tools/pmc_step.sh) + the kernel-trace statistics of the same
round (gpurun_out/<tag>_m{mix,text}_kernel_stats.txt durations) -> profiles/<tag>_m{mix,text}_counters.txt:
per kernel HBM-side bytes per launch (FETCH_SIZE x 2 on wide reads as calibrated in r01_lmhead_traffic.json, WRITE_SIZE
exact), achieved GB/s against the 8 TB/s peak, and MFMA-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel
cycles) with kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs."""
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
FETCH_CORR = 1.9936      # profiles/r01_lmhead_traffic.json calibration (wide coalesced reads under-reported 2x on gfx950)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


for w in ("mix", "text"):
    per = {}
    for p in "FWQ":
        f = os.path.join(root, "gpurun_out", f"pmc_{tag}_{w}_{p}", "summary.json")
        if not os.path.exists(f):
            continue
        for k, cs in json.load(open(f)).items():
            per.setdefault(k, {}).update({c: v[1] / max(v[0], 1) for c, v in cs.items()})
            per[k]["calls"] = max(per[k].get("calls", 0), max(v[0] for v in cs.values()))
    rows = []
    for k, c in per.items():
        name, grid, wg = k.split("|")
        fetch = c.get("FETCH_SIZE", 0.0) * 1024 * FETCH_CORR
        write = c.get("WRITE_SIZE", 0.0) * 1024
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                 # per-XCD busy cycles ~ kernel duration in cycles
        mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc if cyc else 0.0
        wave = c.get("SQ_WAVE_CYCLES", 0.0)
        ns = c.get("DURATION_NS", 0.0)
        rows.append((c["calls"] * (fetch + write), short(name), int(grid) // max(int(wg), 1), c["calls"], fetch, write, cyc,
                     mfma, c.get("SQ_WAIT_ANY", 0.0) / wave if wave else 0.0,
                     c.get("SQ_ACTIVE_INST_VALU", 0.0) / wave if wave else 0.0, (cyc / ns) if ns else 0.0))
    rows.sort(reverse=True)
    out = os.path.join(root, "profiles", f"{tag}_m{w}_counters.txt")
    with open(out, "w") as fh:
        fh.write(f"# rocprofv3 --pmc passes over bench.py --workload m-{w} (5 steps incl. warm-up), tools/pmc_step.sh {tag}\n"
                 "# bytes = HBM-side (L2 fabric) traffic per launch: FETCH_SIZE x 1.99 (gfx950 wide-read correction), WRITE_SIZE exact\n"
                 "# cycles = GRBM_GUI_ACTIVE / 8 XCDs under the profiler (clock ~1.9-2.0 GHz); GB/s = bytes / (cycles / 2.0 GHz)\n"
                 "# mfma = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles); wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES; valu = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES\n"
                 "# GHz = kernel cycles / the wall time of the same dispatches in the same (SQ counter) pass: the clock the kernel ran at inside the step\n")
        fh.write(f"{'kernel':70s} {'blocks':>7s} {'calls':>5s} {'fetch MB':>9s} {'write MB':>9s} {'kcycles':>8s} {'GB/s':>7s} {'of 8TB/s':>8s} {'mfma':>6s} {'wait':>6s} {'valu':>6s} {'GHz':>5s}\n")
        for _, name, blocks, calls, fetch, write, cyc, mfma, wait, valu, ghz in rows[:40]:
            gbs = (fetch + write) / (cyc / ((ghz or 2.0) * 1e9)) / 1e9 if cyc else 0.0
            fh.write(f"{name:70s} {blocks:7d} {calls:5d} {fetch / 1e6:9.1f} {write / 1e6:9.1f} {cyc / 1e3:8.1f} {gbs:7.0f} {gbs / 8000:8.1%} {mfma:6.1%} {wait:6.1%} {valu:6.1%} {ghz:5.2f}\n")
    print(open(out).read())
