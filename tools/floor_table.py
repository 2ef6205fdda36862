#!/usr/bin/env python
"""Floor table of the m-mix step: per kernel family and per GEMM launch class the executed FLOPs, the
algorithmic HBM bytes, the L2 -> LDS DMA bytes of the tiling, the time each of those takes on the chip, the measured time and the ratios.

    python tools/floor_table.py profiles/r05_mmix_kernel_stats.txt profiles/r05_mmix_counters.txt > profiles/r05_floor_table.md

Shapes: the bench default (768d x 6L x 24H, 64 x 1024 positions per step, 22784 loss rows after padding, 52480 computed vocabulary
columns, 25288 image patches = 21 caption images of 256 patches + 22 Atari examples of 26 frames x 36 patches).
Measured times: the rocprofv3 kernel table of the step on one stream (sum of a family's kernels divided by the steps in the table;
bench.py's own roofline legs are taken out by launch counts).
Clocks: MEASURED per kernel inside the step (counters file: GRBM_GUI_ACTIVE / 8 XCDs divided by the wall time of the same dispatches
in the same rocprofv3 pass), not constants -- VERDICT r04 weak 5.  Two matrix floors are printed: at the nominal 2.5 PFLOP/s (2.4 GHz)
and at the clock the kernel ran at.
DMA floor (round 5): the k-loops of the LDS-DMA GEMMs move (tile_M + tile_N) x 64 B per 32-k tile and workgroup through the CU's
L2 -> LDS path, which sustains ~DMA_RATE bytes per clock and CU (profiles/r05_gemm_loop_ablation.txt: the loops with their MFMAs
replaced by s_nop); floor = bytes / (256 CUs x DMA_RATE x clock).  A 256 x 256 tile needs 32 B per clock at the full matrix rate.
Vector floor (attention and the patch kernels are VALU work): executed vector instructions x 64 lanes / (1024 SIMDs x 16 lanes per clock
x clock); attention: SQ_INSTS_VALU per launch (profiles/r04_attn_counters.txt), ResidualBlock: the ISA's instruction counts per patch and wave."""
import re
import sys

M, D, L, T, B, H = 65536, 768, 6, 1024, 64, 24
ROWS_LM, VPAD, V = 22784, 52480, 52305
PATCHES = 21 * 256 + 22 * 26 * 36
PEAK, HBM = 2.5e15, 6.3e12
DMA_RATE = 55.0                       # bytes per clock and CU the L2 -> LDS path delivers for whole-line requests (round 6: tools/probe/dma_rate_probe.hip;
                                      # round 5 priced 28 here, which was the one-wave-per-SIMD loops' issue rate, not the path's)
VALU_LANES_PER_CLK = 1024 * 16        # SIMDs x lanes issued per clock
ATTN_VALU_WAVE_INSTS = (1.322e8 + 1.677e8) * 6          # per step: forward + backward launch, 6 layers
PATCH_VALU_WAVE_INSTS = (2601 + 4310) * 4 * float(PATCHES)

# GEMM launch classes of the step: (label, kernel substring, blocks, launches per step, FLOPs per launch, tile_M, tile_N, M, N, K-sum)
# "K-sum": contraction length summed over the launches of the class that share the row (the N = 768 dgrads have K = 768 / 2304 / 3072)
# Round 6: every class with a k-contiguous A operand runs on gemm_p16_kernel<A_KC, B_KC, F> (F = epilogue feature mask, gemm_epi.h);
# the round-5 kernel of each class is listed behind it so that older tables still parse.
GEMMS = [
    ("c_attn fwd (bias, bf16)", ("gemm_p16_kernelILb1ELb0ELj257E", "gemm_a16_kernelILb1ELb0"), 2304, 6, 256, 256, M, 3 * D, D),
    ("c_fc fwd + GELU + gelu' (2 x bf16 out)", ("gemm_p16_kernelILb1ELb0ELj2311E", "gemm_glds64_kernelILb0"), 3072, 6, 256, 256, M, 4 * D, D),
    ("attn + MLP c_proj fwd (bias, dropout, fp32 residual in / out)", ("gemm_p16_kernelILb1ELb0ELj113E",), 768, 12, 256, 256, M, D, (D + 4 * D) / 2.0),
    ("attn + MLP c_proj fwd (bias, dropout, fp32 residual in / out)", ("gemm_b16_kernel",), 1536, 12, 128, 256, M, D, (D + 4 * D) / 2.0),
    ("dgrad MLP c_proj x gelu' (+ c_fc bias gradient)", ("gemm_p16_kernelILb1ELb1ELj5384E", "gemm_glds64_kernelILb1"), 3072, 6, 256, 256, M, 4 * D, D),
    ("N = 768 dgrads (c_fc, attn c_proj, c_attn: K = 3072 / 768 / 2304)", ("gemm_p16_kernelILb1ELb1ELj256E", "gemm_a16_kernelILb1ELb1"), 768, 18, 256, 256, M, D, (4 * D + D + 3 * D) / 3.0),
    ("wgrads c_fc / MLP c_proj (split-K)", ("gemm_a16_kernelILb0ELb0",), 252, 12, 256, 256, 4 * D, D, M),
    ("wgrads c_attn / attn c_proj (split-K)", ("gemm_a16_kernelILb0ELb0",), 243, 12, 256, 256, 2 * D, D, M),
    ("LM-head logits", ("gemm_p16_kernelILb1ELb1ELj256E", "gemm_a16_kernelILb1ELb1"), 18245, 1, 256, 256, ROWS_LM, VPAD, D),
    ("LM-head dH (split-K)", ("gemm_p16_kernelILb1ELb0ELj576E", "gemm_a16_kernelILb1ELb0"), 1869, 1, 256, 256, ROWS_LM, D, VPAD),
    ("LM-head dW (split-K)", ("gemm_a16_kernelILb0ELb0",), 1230, 1, 256, 256, VPAD, D, ROWS_LM),
]


def fam_of(name, blocks):
    if "gemm_" in name or "gemv" in name:
        if blocks[0] in (18245, 1869, 1230):
            return "LM-head GEMMs (logits, dH, dW)"
        return "block GEMMs (c_attn, c_proj, c_fc, mlp c_proj: fwd, dgrad, wgrad) + patch projection"
    if "attn_" in name:
        return "attention (hd = 32: fwd + two-kernel bwd)"
    if "ln_" in name:
        return "LayerNorm fwd / bwd (+ parameter reductions)"
    if "ce_bf16" in name or "ce_fwd" in name:
        return "cross-entropy (bf16 logits -> dlogits in place)"
    if "resblock" in name or "patch_pos" in name:
        return "image patch kernels (ResidualBlock fwd / bwd, position add)"
    if "adamw" in name or "sqnorm" in name or "adam_step" in name:
        return "clip + AdamW"
    if "splitk_reduce" in name or "colsum" in name:
        return "split-K / column-sum reductions"
    if "pack_embed" in name or "scatter_rows" in name or "gather_rows" in name or "dropout_f32" in name or "cast_f32" in name or "mask_bias" in name or "segsum" in name:
        return "packing / embedding / row gathers"
    return "other (fills, copies, torch glue)"


def parse_stats(path):
    rows = []
    for ln in open(path):
        m = re.match(r"\s*([0-9.]+)\s+([0-9.]+)\s+(\d+)\s+([0-9.]+)\s+\((\d+), (\d+), (\d+)\)\s+\(.*?\)\s+(\S+)", ln)
        if m:
            rows.append((float(m.group(2)), int(m.group(3)), float(m.group(4)), (int(m.group(5)), int(m.group(6)), int(m.group(7))), m.group(8)))
    return rows


def parse_clocks(path):
    """counters file of tools/pmc_step_summarise.py -> [(kernel text, blocks, GHz, mfma busy)]"""
    out = []
    if not path:
        return out
    for ln in open(path):
        m = re.match(r"(.{70}) +(\d+) +(\d+) +[0-9.]+ +[0-9.]+ +[0-9.]+ +\d+ +[0-9.]+% +([0-9.]+)% +[0-9.]+% +[0-9.]+%(?: +([0-9.]+))?", ln)
        if m:
            out.append((m.group(1).strip(), int(m.group(2)), float(m.group(5) or 0.0), float(m.group(4)) / 100))
    return out


def clock_of(clocks, key, blocks, default):
    """key: substring of the DEMANGLED kernel text of the counters file"""
    for name, b, ghz, _ in clocks:
        if key in name and b == blocks and ghz > 0:
            return min(ghz, 2.4)       # (sub-60-us kernels read above the 2.4 GHz maximum: GRBM_GUI_ACTIVE counts the dispatch around them)
    return default


DEMANGLED = {"gemm_a16_kernelILb1ELb0": "gemm_a16_kernel<true, false>", "gemm_a16_kernelILb1ELb1": "gemm_a16_kernel<true, true>",
             "gemm_a16_kernelILb0ELb0": "gemm_a16_kernel<false, false>", "gemm_glds_kernelILb1ELb0": "gemm_glds_kernel<true, false",
             "gemm_glds_kernelILb1ELb1": "gemm_glds_kernel<true, true", "gemm_b16_kernel": "gemm_b16_kernel",
             "gemm_glds64_kernelILb0": "gemm_glds64_kernel<false>", "gemm_glds64_kernelILb1": "gemm_glds64_kernel<true>",
             "gemm_p16_kernelILb1ELb0ELj257E": "gemm_p16_kernel<true, false, 257", "gemm_p16_kernelILb1ELb0ELj2311E": "gemm_p16_kernel<true, false, 2311",
             "gemm_p16_kernelILb1ELb0ELj113E": "gemm_p16_kernel<true, false, 113", "gemm_p16_kernelILb1ELb1ELj5384E": "gemm_p16_kernel<true, true, 5384",
             "gemm_p16_kernelILb1ELb1ELj256E": "gemm_p16_kernel<true, true, 256", "gemm_p16_kernelILb1ELb0ELj576E": "gemm_p16_kernel<true, false, 576"}


def main():
    path = sys.argv[1]
    clocks = parse_clocks(sys.argv[2] if len(sys.argv) > 2 else None)
    rows = parse_stats(path)
    steps = max(c for (_, c, _, _, n) in rows if "ce_bf16" in n)
    meas, fam_ms_cyc = {}, {}
    for tot_ms, calls, avg, blocks, name in rows:
        f = fam_of(name, blocks)
        per_step = calls // steps
        if per_step == 0:
            continue
        meas[f] = meas.get(f, 0.0) + avg * per_step / 1e3          # ms per step
        ghz = 0.0
        for cname, b, g, _ in clocks:            # time-weighted clock of the family
            if b == blocks[0] and g > 0 and any(tok in cname for tok in re.findall(r"[a-z][a-z_0-9]*_kernel", name)):
                ghz = g
                break
        if ghz:
            a = fam_ms_cyc.setdefault(f, [0.0, 0.0])
            a[0] += avg * per_step
            a[1] += avg * per_step * ghz
    # (short kernels read above the 2.4 GHz maximum: GRBM_GUI_ACTIVE also counts the dispatch around a 10-us kernel -- capped)
    fam_clock = {f: min(v[1] / v[0], 2.4) for f, v in fam_ms_cyc.items() if v[0] > 0}

    # ---- per GEMM launch class -------------------------------------------------------------------------------------------------
    print(f"### GEMM launch classes inside the step (`{path}`; clocks: `{sys.argv[2] if len(sys.argv) > 2 else '-'}`)\n")
    print("| launch class | launches / step | us / launch | TFLOP/s | of 2.5 PF | GHz in the step | of the matrix peak at that clock | DMA GB / launch | DMA floor us (55 B/clk/CU) | measured / max(matrix at clock, DMA) |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    seen = set()
    for label, keys, blocks, per_step, tm, tn, m, n, k in GEMMS:
        hit, key = [], None
        for key in keys:
            hit = [(avg, calls) for (_, calls, avg, b, name) in rows if key in name and b[0] == blocks]
            if hit:
                break
        if not hit or label in seen:
            continue
        seen.add(label)
        label = f"{label} [{key.split('_kernel')[0]}]"
        us = sum(a * c for a, c in hit) / sum(c for _, c in hit)
        fl = 2.0 * m * n * k
        ghz = clock_of(clocks, DEMANGLED[key], blocks, 2.0)
        dma = (m / tm) * (n / tn) * (k / 32.0) * (tm + tn) * 64.0
        t_mat = fl / (PEAK * ghz / 2.4) * 1e6
        t_dma = dma / (256 * DMA_RATE * ghz * 1e9) * 1e6
        print(f"| {label} | {per_step} | {us:.1f} | {fl / us / 1e6:.0f} | {fl / us / 1e6 / 2500:.2f} | {ghz:.2f} | {fl / us / 1e6 / (2500 * ghz / 2.4):.2f} | "
              f"{dma / 1e9:.2f} | {t_dma:.0f} | {us / max(t_mat, t_dma):.2f} |")
    print()

    # ---- model: executed FLOPs and algorithmic bytes per step ------------------------------------------------------------------
    blk_fwd = 2.0 * M * D * (3 * D + D + 4 * D + 4 * D)
    gemm_flops = 3 * L * blk_fwd + 3 * 2.0 * PATCHES * 768 * D
    gemm_bytes = L * (  # forward
        M * D * 2 + M * 3 * D * 2 + M * D * 2 + 2 * M * D * 4 + M * D * 2 + 2 * M * 4 * D * 2 + M * 4 * D * 2 + 2 * M * D * 4
        # dgrads: read upstream + (factor), write
        + M * D * 2 + M * 4 * D * 2 * 2 + M * 4 * D * 2 + M * D * 2 + M * D * 2 + M * D * 2 + M * 3 * D * 2 + M * D * 2
        # wgrads: read both operands
        + (M * 4 * D * 2 + M * D * 2) * 2 + M * D * 2 * 2 + (M * 3 * D * 2 + M * D * 2))
    live = [g for g in GEMMS if any(any(key in name and b[0] == g[2] for key in g[1]) for (_, _, _, b, name) in rows)]
    uniq = {}
    for g in live:
        uniq.setdefault(g[0], g)
    gemm_dma = sum(per * (m / tm) * (n / tn) * (k / 32.0) * (tm + tn) * 64.0 for (_, _, blocks, per, tm, tn, m, n, k) in uniq.values() if blocks not in (18245, 1869, 1230))
    lm_dma = sum(per * (m / tm) * (n / tn) * (k / 32.0) * (tm + tn) * 64.0 for (_, _, blocks, per, tm, tn, m, n, k) in uniq.values() if blocks in (18245, 1869, 1230))
    lm_flops = 3 * 2.0 * ROWS_LM * D * VPAD
    lm_bytes = ROWS_LM * VPAD * 2 * 3.0 + VPAD * D * 2 * 3
    attn_flops = L * (2.0 * B * T * T * D) * 3.5                       # useful causal: fwd 4 T^2/2 hd per head, bwd 2.5x
    attn_bytes = L * (M * 3 * D * 2 + M * D * 2 + (M * 3 * D * 2 + 2 * M * D * 2 + M * 3 * D * 2) + 2 * B * H * (T // 32) ** 2 * 32 * 4 / 2)
    ln_bytes = (2 * L + 1) * (M * D * 4 + M * D * 2) + 2 * L * (M * D * 2 + 3 * M * D * 4 + M * D * 2) + (M * D * 4 * 2 + ROWS_LM * D * 4)
    ce_bytes = 2.0 * ROWS_LM * VPAD * 2
    adam_bytes = 124.4e6 * (4 * 4 + 3 * 4 + 2)
    patch_flops = PATCHES * (2 * 27 * 128 * 256 * 2) * 4.0
    fams = [
        ("block GEMMs (c_attn, c_proj, c_fc, mlp c_proj: fwd, dgrad, wgrad) + patch projection", gemm_flops, gemm_bytes, 0.0, gemm_dma),
        ("LM-head GEMMs (logits, dH, dW)", lm_flops, lm_bytes, 0.0, lm_dma),
        ("attention (hd = 32: fwd + two-kernel bwd)", attn_flops, attn_bytes, ATTN_VALU_WAVE_INSTS, 0.0),
        ("LayerNorm fwd / bwd (+ parameter reductions)", 0.0, ln_bytes, 0.0, 0.0),
        ("cross-entropy (bf16 logits -> dlogits in place)", 0.0, ce_bytes, 0.0, 0.0),
        ("image patch kernels (ResidualBlock fwd / bwd, position add)", patch_flops, PATCHES * 768 * (4 + 2 + 4 + 4) * 2.0, PATCH_VALU_WAVE_INSTS, 0.0),
        ("clip + AdamW", 0.0, adam_bytes, 0.0, 0.0),
        ("split-K / column-sum reductions", 0.0, 0.0, 0.0, 0.0),
        ("packing / embedding / row gathers", 0.0, M * D * 4 * 4.0, 0.0, 0.0),
        ("other (fills, copies, torch glue)", 0.0, 0.0, 0.0, 0.0),
    ]
    print(f"### Kernel families (m-mix, 64 x 1024 per step; `{path}`)\n")
    print("| kernel family | executed TFLOP | algorithmic GB | L2 -> LDS DMA GB | vector G lane-ops | GHz in the step | floor ms at nominal 2.5 PF / 6.3 TB/s | "
          "floor ms at the measured clock | bound | measured ms | measured / floor at clock |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    tf = tn_ = tm = 0.0
    for name, fl, by, vinst, dma in fams:
        ghz = fam_clock.get(name, 2.0)
        vi = vinst * 64
        parts = {"matrix": fl / (PEAK * ghz / 2.4), "hbm": by / HBM, "dma": dma / (256 * DMA_RATE * ghz * 1e9), "vector": vi / (VALU_LANES_PER_CLK * ghz * 1e9)}
        nominal = max(fl / PEAK, by / HBM, vi / (VALU_LANES_PER_CLK * 2.4e9)) * 1e3
        bound = max(parts, key=parts.get)
        floor = parts[bound] * 1e3
        m = meas.get(name, 0.0)
        tf += floor
        tn_ += nominal
        tm += m
        ratio = f"{m / floor:.2f}" if floor > 0.02 else "-"
        print(f"| {name} | {fl / 1e12:.2f} | {by / 1e9:.2f} | {dma / 1e9:.1f} | {vi / 1e9:.1f} | {ghz:.2f} | {nominal:.2f} | {floor:.2f} | {bound if floor > 0.02 else '-'} | {m:.2f} | {ratio} |")
    print(f"| **sum** | | | | | | **{tn_:.1f}** | **{tf:.1f}** | | **{tm:.1f}** | {tm / tf:.2f} |")


if __name__ == "__main__":
    main()
