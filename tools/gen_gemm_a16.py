#!/usr/bin/env python
"""Writes neko_amd/csrc/gemm_a16_loop.inc: the hand-placed main loop of gemm_a16.hip, one instruction stream per operand-layout pair.

    python tools/gen_gemm_a16.py            # regenerate (the .inc is committed; build.py does not run this)

Why a generator: the loop is written instruction by instruction (registers, waits and the position of every LDS read and DMA piece
between the MFMAs are chosen here, not by hipcc), and the four layout variants x 4 unrolled k-tiles differ only in operand-read
forms and literal offsets.  Everything below is gfx950 assembly text; nothing is translated from another source.

Geometry (one workgroup = 4 waves = one 256 x 256 output tile, one wave per SIMD, 512 registers per lane):
  wave (wm, wn) owns 128 x 128: 8 x 8 blocks of 16 x 16, accumulators a[0:255] (block (ti, tj) = a[4(8 ti + tj) : +3]),
  v_mfma_f32_16x16x32_bf16 with SWAPPED operands (srcA = B fragment, srcB = A fragment) so that a lane holds row l&15 and the
  four consecutive columns 4(l>>4)..+3 of a block (one 16-B piece of an output row).
  k-tile = 32: 64 MFMAs per wave between block barriers, one fragment set (8 A + 8 B fragments = 64 VGPRs) per k-tile, two sets
  (v[128:191], v[192:255]) alternating: the next tile's fragments are read from LDS between this tile's MFMAs.
LDS ring: 4 stages; A stages at s * 16 KB, B stages at 64 KB + s * 16 KB (every read offset fits the 16-bit immediate).
  k-contiguous operand tile [256 rows][32 k] (64-B rows): 16-B chunk c of row r sits at slot c ^ g[(r>>2)&3], g = [0,2,3,1]:
      a 16x16x32 fragment is ONE ds_read_b128 per lane (row l&15, chunk l>>4) and all four 16-lane groups of the instruction
      spread over the 16 slots of a 256-B bank row.
  k-strided operand tile [32 k][256 cols] (512-B rows): 16-B piece p of k-row k sits at piece p ^ (2 hh(k)),
      hh(k) = (k&3) | ((k>>3)&1)<<2: a fragment is two ds_read_b64_tr_b16 (k-rows 8(l>>4) + (c>>2) and +4) and the eight k-rows
      a 32-lane half touches land in eight different 32-B bank segments.
DMA: global_load_lds_dwordx4, 8 pieces (1 KiB each) per wave per k-tile, three tiles ahead; M0 rewritten per piece (s_add_u32 from
  the wave's LDS base), the piece's source = 64-bit SGPR base (advanced per tile, frozen at the last tile: the three surplus
  requests at the end re-fetch the last tile into slots nobody reads) + a loop-invariant per-lane VGPR offset.
Per tile:  s_waitcnt vmcnt(8) [tile kt+1 landed: this wave's pieces] ; s_waitcnt lgkmcnt(0) [fragment set of tile kt complete] ;
  s_barrier [tile kt+1 visible from every wave; every wave is past the reads of tile kt-1, whose slot tile kt+3 refills] ;
  64 MFMAs with the reads of tile kt+1 and the DMA pieces of tile kt+3 placed in fixed gaps.
"""
import os
import sys

NSTAGE = 4
A_ALL = 0
B_ALL = 65536
STAGE = 16384          # per operand per stage
NPIECE = 4             # DMA pieces per wave per operand per k-tile

# fixed registers (all listed as clobbers in gemm_a16.hip)
S_GA, S_GB, S_KT, S_CNT, S_M0, S_TA, S_TB = 84, 86, 88, 89, 90, 91, 92
V_KSA, V_KSB = 96, 104          # 8 address registers each (k-strided operand, one per 16-column / 16-row block)
V_FRAG = 128                    # two fragment sets of 64


def frag(setp, which, t):
    """first VGPR of fragment t (0..7) of operand which ('a'/'b') in set setp"""
    return V_FRAG + 64 * setp + (0 if which == "a" else 32) + 4 * t


def reads_for(which, kc, stage, setp):
    """LDS reads of the 8 fragments of one operand of the tile in ring stage `stage` into set setp"""
    out = []
    for t in range(8):
        r = frag(setp, which, t)
        if kc:
            out.append(f"ds_read_b128 v[{r}:{r + 3}], %[r{which}] offset:{stage * STAGE + t * 1024}")
        else:
            base = (V_KSA if which == "a" else V_KSB) + t
            out.append(f"ds_read_b64_tr_b16 v[{r}:{r + 1}], v{base} offset:{stage * STAGE}")
            out.append(f"ds_read_b64_tr_b16 v[{r + 2}:{r + 3}], v{base} offset:{stage * STAGE + 2048}")
    return out


def dma_piece(which, pc, stage):
    """(set M0, issue) of one DMA piece; the two must not be adjacent (SALU write of M0 -> LDS-DMA needs one wait state)"""
    region = A_ALL if which == "a" else B_ALL
    sg = S_GA if which == "a" else S_GB
    return (f"s_add_u32 m0, %[ldsw], {region + stage * STAGE + pc * 1024}",
            f"global_load_lds_dwordx4 %[vo{which}{pc}], s[{sg}:{sg + 1}]")


def advance():
    """next DMA tile: the bases move on only while tile index + 1 < nkt (afterwards the last tile is re-fetched)"""
    return [f"s_add_u32 s{S_KT}, s{S_KT}, 1",
            f"s_cmp_lt_u32 s{S_KT}, %[nkt]",
            f"s_cselect_b32 s{S_TA}, %[sa], 0",
            f"s_cselect_b32 s{S_TB}, %[sb], 0",
            f"s_add_u32 s{S_GA}, s{S_GA}, s{S_TA}",
            f"s_addc_u32 s{S_GA + 1}, s{S_GA + 1}, 0",
            f"s_add_u32 s{S_GB}, s{S_GB}, s{S_TB}",
            f"s_addc_u32 s{S_GB + 1}, s{S_GB + 1}, 0"]


def mfma(ti, tj, setp):
    acc = 4 * (8 * ti + tj)
    a, b = frag(setp, "a", ti), frag(setp, "b", tj)
    return f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc + 3}], v[{b}:{b + 3}], v[{a}:{a + 3}], a[{acc}:{acc + 3}]"


def tile_body(u, a_kc, b_kc, sched):
    """k-tile with ring stage u (of the 4 unrolled): MFMAs on set u&1, reads of stage u+1 into the other set, DMA into stage u+3"""
    setp, nxt, dst = u & 1, (u + 1) % NSTAGE, (u + 3) % NSTAGE
    lines = ["s_waitcnt vmcnt(8)", "s_waitcnt lgkmcnt(0)", "s_barrier"]
    reads = reads_for("b", b_kc, nxt, setp ^ 1) + reads_for("a", a_kc, nxt, setp ^ 1)
    pieces = [dma_piece("a", pc, dst) for pc in range(NPIECE)] + [dma_piece("b", pc, dst) for pc in range(NPIECE)]
    fill = {m: [] for m in range(64)}
    # fragment reads: evenly over MFMAs [0, read_span)
    span = sched["read_span"]
    for i, r in enumerate(reads):
        fill[(i * span) // len(reads)].append(r)
    # DMA pieces: M0 write after MFMA m, the request after MFMA m+1
    first, step = sched["dma_first"], sched["dma_step"]
    for i, (setm0, issue) in enumerate(pieces):
        m = first + i * step
        fill[m].append(setm0)
        fill[m + 1].append(issue)
    adv = advance()
    for i, ins in enumerate(adv):
        fill[min(63, first + len(pieces) * step + i)].append(ins)
    order = [(ti, tj) for ti in range(8) for tj in (range(8) if ti % 2 == 0 or not sched["snake"] else range(7, -1, -1))]
    for m, (ti, tj) in enumerate(order):
        lines.append(mfma(ti, tj, setp))
        lines += fill[m]
    return lines


def stream(a_kc, b_kc, sched):
    L = []
    L += ["s_nop 4",
          f"s_mov_b32 s{S_M0}, m0",
          f"s_mov_b32 s{S_GA}, %[galo]", f"s_mov_b32 s{S_GA + 1}, %[gahi]",
          f"s_mov_b32 s{S_GB}, %[gblo]", f"s_mov_b32 s{S_GB + 1}, %[gbhi]",
          f"s_mov_b32 s{S_KT}, 0",
          f"s_lshr_b32 s{S_CNT}, %[nkt], 2"]
    # per-block LDS addresses of a k-strided operand: base + ((t ^ hh) << 5)
    for which, kc, vb in (("a", a_kc, V_KSA), ("b", b_kc, V_KSB)):
        if not kc:
            for t in range(8):
                L += [f"v_xor_b32 v{vb + t}, {t}, %[h{which}]", f"v_lshl_add_u32 v{vb + t}, v{vb + t}, 5, %[r{which}]"]
    # prologue: tiles 0..2 requested, accumulators zeroed underneath
    zero = [f"v_accvgpr_write_b32 a{i}, 0" for i in range(256)]
    zi = 0
    for t in range(NSTAGE - 1):
        for which in ("a", "b"):
            for pc in range(NPIECE):
                setm0, issue = dma_piece(which, pc, t)
                L += [setm0, zero[zi], issue] + zero[zi + 1:zi + 8]
                zi += 8
        L += advance()
    L += zero[zi:]
    # tile 0 landed and visible; its fragments into set 0
    L += ["s_waitcnt vmcnt(16)", "s_barrier"]
    L += reads_for("b", b_kc, 0, 0) + reads_for("a", a_kc, 0, 0)
    L += ["LOOP_%=:"]
    for u in range(NSTAGE):
        L += tile_body(u, a_kc, b_kc, sched)
    L += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", "s_cbranch_scc1 LOOP_%="]
    # surplus DMA landed, surplus fragment reads returned (their VGPRs go back to the compiler), accumulators readable
    L += ["s_waitcnt vmcnt(0)", "s_waitcnt lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_barrier", f"s_mov_b32 m0, s{S_M0}"]
    return L


def clobbers():
    c = ['"memory"', '"vcc"', '"scc"']
    c += [f'"v{i}"' for i in range(V_KSA, 256)]
    c += [f'"s{i}"' for i in range(S_GA, S_TB + 1)]
    return ", ".join(c)


SCHED = {"read_span": 40, "dma_first": 6, "dma_step": 6, "snake": True}


def main():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neko_amd", "csrc", "gemm_a16_loop.inc")
    txt = ["// GENERATED by tools/gen_gemm_a16.py -- do not edit; the generator is the source (design notes in its docstring).", ""]
    for a_kc in (True, False):
        for b_kc in (True, False):
            name = f"NEKO_A16_LOOP_{'KC' if a_kc else 'KS'}_{'KC' if b_kc else 'KS'}"
            L = stream(a_kc, b_kc, SCHED)
            txt.append(f"#define {name} \\")
            txt += [f'  "{ins}\\n\\t" \\' for ins in L[:-1]]
            txt.append(f'  "{L[-1]}"')
            txt.append("")
    txt.append(f"#define NEKO_A16_CLOBBERS {clobbers()}")
    txt.append("")
    open(out, "w").write("\n".join(txt))
    print(out, sum(len(t) for t in txt), "bytes")


if __name__ == "__main__":
    sys.exit(main())
