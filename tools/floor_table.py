#!/usr/bin/env python
"""Floor table of the m-mix step: per kernel family the executed FLOPs, the algorithmic HBM bytes, the time
that work takes at min(matrix pipe at the SUSTAINED clock, 6.3 TB/s), the measured time and their ratio.

    python tools/floor_table.py profiles/r04_mmix_kernel_stats.txt [steps_in_table] > profiles/r04_floor_table.md

Shapes: the bench default (768d x 6L x 24H, 64 x 1024 positions per step, 22784 loss rows after padding, 52480 computed vocabulary
columns, ~12289 image patches).  Measured times: the rocprofv3 kernel table of the same step (sum of the family's kernels divided by
the steps in the table; the table's launches of bench.py's own roofline legs are taken out by their grid sizes).
Vector floor (attention and the patch kernels are VALU work, not matrix work): executed vector instructions x 64 lanes / (1024 SIMDs x 16
lanes per clock x clock).  Attention: SQ_INSTS_VALU per launch of profiles/r04_attn_counters.txt (forward 1.322e8, one-pass backward
1.677e8 wave-instructions at B = 64; 10.5 per score in the forward, 4.3 of them the dropout decision); ResidualBlock: the ISA's vector
instructions per patch and wave (forward 2601, backward 4310; exact-erf GELU on 128 channels x 256 pixels) x 4 waves x 12289 patches.
Sustained clock: what the counters of this round show under each kind of load (profiles/r04_gemm_counters.txt: 8192^3 on the
hand-placed loop runs at 1.55 GHz, the K = 768 shapes and attention around 2.0-2.1 GHz); the matrix peak scales with it from
2.5 PFLOP/s at 2.4 GHz."""
import re
import sys

M, D, L, T, B, H = 65536, 768, 6, 1024, 64, 24
ROWS_LM, VPAD, V = 22784, 52480, 52305
PATCHES = 12289
PEAK, HBM = 2.5e15, 6.3e12
VALU_LANES_PER_CLK = 1024 * 16        # SIMDs x lanes issued per clock
ATTN_VALU_WAVE_INSTS = (1.322e8 + 1.677e8) * 6          # per step: forward + backward launch, 6 layers
PATCH_VALU_WAVE_INSTS = (2601 + 4310) * 4 * 12289.0     # per step


def fam_of(name, blocks):
    if "gemm_" in name or "gemv" in name:
        if blocks[0] in (18245, 1869, 1230, 615 * 2, 615) or "18245" in str(blocks):
            return "LM-head GEMMs (logits, dH, dW)"
        return "block GEMMs (c_attn, c_proj, c_fc, mlp c_proj: fwd, dgrad, wgrad) + patch projection"
    if "attn_" in name:
        return "attention (hd = 32: fwd + one-pass bwd)"
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
    if "pack_embed" in name or "scatter_rows" in name or "gather_rows" in name or "dropout_f32" in name or "cast_f32" in name or "mask_bias" in name:
        return "packing / embedding / row gathers"
    return "other (fills, copies, torch glue)"


def main():
    path = sys.argv[1]
    rows = []
    for ln in open(path):
        m = re.match(r"\s*([0-9.]+)\s+([0-9.]+)\s+(\d+)\s+([0-9.]+)\s+\((\d+), (\d+), (\d+)\)\s+\(.*?\)\s+(\S+)", ln)
        if m:
            rows.append((float(m.group(2)), int(m.group(3)), float(m.group(4)), (int(m.group(5)), int(m.group(6)), int(m.group(7))), m.group(8)))
    # steps in the table = launches of the cross-entropy kernel (one per step)
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else max(c for (_, c, _, _, n) in rows if "ce_bf16" in n)
    meas = {}
    for tot_ms, calls, avg, blocks, name in rows:
        f = fam_of(name, blocks)
        # bench.py's roofline legs: 12 extra logits launches (18245 blocks) beyond one per step, and the roofline_more kernels (few calls):
        # keep per-step launches only by capping calls at a multiple of steps
        per_step = calls // steps
        if per_step == 0:
            continue
        meas[f] = meas.get(f, 0.0) + avg * per_step / 1e3          # ms per step
    # ---- model: executed FLOPs and algorithmic bytes per step ------------------------------------------------------------------
    blk_fwd = 2.0 * M * D * (3 * D + D + 4 * D + 4 * D)
    gemm_flops = 3 * L * blk_fwd + 3 * 2.0 * PATCHES * 768 * D
    # per layer and pass the GEMMs read their activation operand and write their output once (bf16 unless residual fp32)
    gemm_bytes = L * (  # forward
        M * D * 2 + M * 3 * D * 2 + M * D * 2 + 2 * M * D * 4 + M * D * 2 + 2 * M * 4 * D * 2 + M * 4 * D * 2 + 2 * M * D * 4
        # dgrads: read upstream + (factor), write
        + M * D * 2 + M * 4 * D * 2 * 2 + M * 4 * D * 2 + M * D * 2 + M * D * 2 + M * D * 2 + M * 3 * D * 2 + M * D * 2
        # wgrads: read both operands
        + (M * 4 * D * 2 + M * D * 2) * 2 + M * D * 2 * 2 + (M * 3 * D * 2 + M * D * 2))
    lm_flops = 3 * 2.0 * ROWS_LM * D * VPAD
    lm_bytes = ROWS_LM * VPAD * 2 * 3.0 + VPAD * D * 2 * 3
    attn_flops = L * (2.0 * B * T * T * D) * 3.5                       # useful causal: fwd 4 T^2/2 hd per head, bwd 2.5x
    attn_bytes = L * (M * 3 * D * 2 + M * D * 2 + (M * 3 * D * 2 + 2 * M * D * 2 + M * 3 * D * 2) + 2 * B * H * (T // 32) ** 2 * 32 * 4 / 2)
    ln_bytes = (2 * L + 1) * (M * D * 4 + M * D * 2) + 2 * L * (M * D * 2 + 3 * M * D * 4 + M * D * 2) + (M * D * 4 * 3)
    ce_bytes = 2.0 * ROWS_LM * VPAD * 2
    adam_bytes = 124.4e6 * (4 * 4 + 3 * 4 + 2)
    patch_flops = PATCHES * (2 * 27 * 128 * 256 * 2) * 4.0
    fams = [
        ("block GEMMs (c_attn, c_proj, c_fc, mlp c_proj: fwd, dgrad, wgrad) + patch projection", gemm_flops, gemm_bytes, 1.85),
        ("LM-head GEMMs (logits, dH, dW)", lm_flops, lm_bytes, 1.7),
        ("attention (hd = 32: fwd + one-pass bwd)", attn_flops, attn_bytes, 2.1, ATTN_VALU_WAVE_INSTS),
        ("LayerNorm fwd / bwd (+ parameter reductions)", 0.0, ln_bytes, 2.1),
        ("cross-entropy (bf16 logits -> dlogits in place)", 0.0, ce_bytes, 2.1),
        ("image patch kernels (ResidualBlock fwd / bwd, position add)", patch_flops, PATCHES * 768 * (4 + 2 + 4 + 4) * 2.0, 2.1, PATCH_VALU_WAVE_INSTS),
        ("clip + AdamW", 0.0, adam_bytes, 2.1),
        ("split-K / column-sum reductions", 0.0, 0.0, 2.1),
        ("packing / embedding / row gathers", 0.0, M * D * 4 * 4.0, 2.1),
        ("other (fills, copies, torch glue)", 0.0, 0.0, 2.1),
    ]
    print(f"| kernel family (m-mix, 64 x 1024 per step; `{path}`) | executed TFLOP | algorithmic GB | vector G lane-ops | sustained GHz | floor ms = max(FLOP / (2.5 PF x GHz / 2.4), B / 6.3 TB/s, lane-ops / (16384 x GHz)) | bound | measured ms | measured / floor |")
    print("|---|---|---|---|---|---|---|---|---|")
    tf = tm = 0.0
    for fam in fams:
        name, fl, by, ghz = fam[:4]
        vi = fam[4] * 64 if len(fam) > 4 else 0.0
        parts = {"matrix": fl / (PEAK * ghz / 2.4), "hbm": by / HBM, "vector": vi / (VALU_LANES_PER_CLK * ghz * 1e9)}
        bound = max(parts, key=parts.get)
        floor = parts[bound] * 1e3
        m = meas.get(name, 0.0)
        tf += floor
        tm += m
        ratio = f"{m / floor:.2f}" if floor > 0.02 else "-"
        print(f"| {name} | {fl / 1e12:.2f} | {by / 1e9:.2f} | {vi / 1e9:.1f} | {ghz:.2f} | {floor:.2f} | {bound if floor > 0.02 else '-'} | {m:.2f} | {ratio} |")
    print(f"| **sum** | | | | | **{tf:.1f}** | | **{tm:.1f}** | {tm / tf:.2f} |")


if __name__ == "__main__":
    main()
