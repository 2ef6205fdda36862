#!/usr/bin/env python
"""Writes neko_amd/csrc/gemm_a16_loop.inc: the hand-placed main loop of gemm_a16.hip, one instruction stream per operand-layout pair.

    python tools/gen_gemm_a16.py [--kc 32|64]           # regenerate (the .inc is committed; build.py does not run this)
    python tools/gen_gemm_a16.py --geom b16             # the 128 x 256 / two-workgroups-per-CU variant (gemm_b16_loop.inc), see GEOMS

Why a generator: the loop is written instruction by instruction (registers, waits and the position of every LDS read and DMA piece
between the MFMAs are chosen here, not by hipcc), and the four layout variants x 4 unrolled k-tiles differ only in operand-read
forms and literal offsets.  Everything below is gfx950 assembly text; nothing is translated from another source.

Geometry (one workgroup = 4 waves = one 256 x 256 output tile, one wave per SIMD, 512 registers per lane):
  wave (wm, wn) owns 128 x 128: 8 x 8 blocks of 16 x 16, accumulators a[0:255] (block (ti, tj) = a[4(8 ti + tj) : +3]),
  v_mfma_f32_16x16x32_bf16 with SWAPPED operands (srcA = B fragment, srcB = A fragment) so that a lane holds row l&15 and the
  four consecutive columns 4(l>>4)..+3 of a block (one 16-B piece of an output row).
  k-tile = 32: 64 MFMAs per wave between block barriers, one fragment set (8 A + 8 B fragments = 64 VGPRs) per k-tile, two sets
  (v[128:191], v[192:255]) alternating: the next tile's fragments are read from LDS between this tile's MFMAs.
LDS: 64 KB per operand (A at 0, B at 64 KB), organised per operand layout:
  k-strided operand ("ks", rows of the source are k): ring of 4 stages x [32 k][256 cols] (512-B rows); 16-B piece p of k-row k
      sits at piece p ^ (2 hh(k)), hh(k) = (k&3) | ((k>>3)&1)<<2: a fragment is two ds_read_b64_tr_b16 (k-rows 8(l>>4) + (c>>2)
      and +4) and the eight k-rows a 32-lane half touches land in eight different 32-B bank segments.  DMA piece = 2 k-rows.
  k-contiguous operand, --kc 64 ("kc64", the default): TWO slots x [256 rows][64 k] (128-B rows = whole cache lines: with 64-B
      rows every line of the operand was requested twice, by two consecutive k-tiles, and the L2 -> LDS traffic of the NT form
      ran near the L2's limit); 16-B chunk c (0..7) of row r sits at position c ^ ((r>>1)&7); a fragment of the k-tile in half h
      of the slot is ONE ds_read_b128 per lane (row l&15, chunk (l>>4) + 4h) and every 16-lane group of the instruction covers
      all 16 slots of a 256-B bank row.  DMA piece = 8 rows; a slot (two k-tiles) is requested in one go during an odd tile.
  k-contiguous operand, --kc 32 ("kc32", the first version, kept for A/B runs): ring of 4 stages x [256 rows][32 k] (64-B rows),
      chunk c at slot c ^ g[(r>>2)&3], g = [0,2,3,1].  DMA piece = 16 rows.
DMA: global_load_lds_dwordx4 (1 KiB per wave-instruction); M0 rewritten per piece (s_add_u32 from the wave's LDS base), the
  piece's source = 64-bit SGPR base (advanced per stage / slot, frozen at the last one: surplus requests at the end re-fetch the
  last stage into LDS nobody reads any more) + a loop-invariant per-lane VGPR offset.  ks / kc32: tile t+3 is requested during tile
  t; kc64: the slot of tiles (t+3, t+4) during odd tile t (its previous content, tiles (t-1, t), was last read during tile t-1).
Per tile:  s_waitcnt vmcnt(N) [tile t+1's data landed: this wave's pieces; N = pieces issued after the last one it needs, computed
  below by replaying the issue order] ; s_waitcnt lgkmcnt(0) [fragment set of tile t complete] ; s_barrier [tile t+1 visible from
  every wave; every wave is past the reads of tile t] ; 64 MFMAs with the reads of tile t+1 and the DMA pieces in fixed gaps.
"""
import argparse
import os
import sys

# fixed registers (all listed as clobbers in gemm_a16.hip / gemm_b16.hip)
S_G = {"a": 84, "b": 86}        # 64-bit DMA source bases
S_NEXT = {"a": 88, "b": 89}     # index of the next stage / slot to request
S_LIM = {"a": 90, "b": 91}      # number of stages / slots
S_CNT, S_M0, S_T = 92, 93, 94
S_LAST = 95


class Geom:
    """One workgroup geometry.  a16: 256 x 256 per workgroup, 128 x 128 per wave (8 x 8 blocks, 256 accumulators, 512 registers, one
    workgroup per CU, 128 KB of LDS: 64 KB per operand, 4-stage 32-k rings / two 64-k slots).  b16 (round 5): 128 x 256 per workgroup,
    64 x 128 per wave (4 x 8 blocks, 128 accumulators, <= 256 registers, 80 KB of LDS) so that TWO workgroups share a CU and one's
    output phase (epilogue arithmetic + stores) runs under the other's main loop: A (k-contiguous only) in two 64-k slots of 128 rows =
    32 KB at 0, B in a 3-stage 32-k ring = 48 KB at 32 KB (a slot that was read during tile t - 1 is refilled with tile t + 3 during
    tile t, so three stages carry the same three tiles of lookahead as a16's four).  The slot period (4 tiles) and the ring period (3)
    give a loop body of 12 k-tiles: the contraction range must be a multiple of 384."""

    def __init__(self, name, fa, fb, ring, region, slot64, npiece64, v_ks, v_frag, boff, trip, nslot=2, npiece_ring=4, nsets=2):
        self.name, self.fa, self.fb, self.ring, self.region, self.slot64, self.npiece64 = name, fa, fb, ring, region, slot64, npiece64
        self.v_ks, self.v_frag, self.boff, self.trip = v_ks, v_frag, boff, trip
        self.nslot, self.npiece_ring, self.nsets = nslot, npiece_ring, nsets      # 64-k slots of a kc64 operand; pieces per wave per ring stage
        self.set = 4 * (fa + fb)                 # registers of one fragment set
        self.nm = fa * fb                        # MFMAs per k-tile
        self.vlo = min(list(v_ks.values()) + [v_frag])
        self.vhi = v_frag + nsets * self.set     # clobbered VGPRs: [vlo, vhi)


GEOMS = {
    "a16": Geom("A16", 8, 8, 4, {"a": 0, "b": 65536}, {"a": 32768, "b": 32768}, {"a": 8, "b": 8}, {"a": 96, "b": 104}, 128, 32, 4),
    "b16": Geom("B16", 4, 8, 3, {"a": 0, "b": 32768}, {"a": 16384}, {"a": 4}, {"b": 24}, 32, 16, 12),
    # p16 (round 6): 256 x 256 per workgroup of EIGHT waves (two per SIMD), 128 x 64 per wave; see stream_p16
    "p16": Geom("P16", 8, 4, 4, {"a": 0, "b": 98304}, {"a": 32768}, {"a": 4}, {"a": 48, "b": 56}, 64, 32, 12, nslot={"a": 3, "b": 2}, npiece_ring=2, nsets=1),
}
G = GEOMS["a16"]


class Op:
    """one operand: layout mode, LDS addressing, DMA pieces"""

    def __init__(self, which, mode):
        self.w, self.mode = which, mode
        self.nslot = G.nslot[which] if isinstance(G.nslot, dict) else G.nslot      # 64-k slots (kc64)
        self.npiece = G.npiece64[which] if mode == "kc64" else G.npiece_ring     # per wave per request unit (stage or slot)
        self.nfrag = G.fa if which == "a" else G.fb
        assert mode != "ks" or which in G.v_ks, f"geometry {G.name}: operand {which} cannot be k-strided"

    def frag(self, setp, t):
        return G.v_frag + G.set * setp + (0 if self.w == "a" else G.boff) + 4 * t

    def reads(self, tile, setp):
        """LDS reads of the fragments of k-tile `tile` (absolute index) into fragment set setp"""
        out = []
        for t in range(self.nfrag):
            r = self.frag(setp, t)
            if self.mode == "kc64":
                slot, half = (tile >> 1) % self.nslot, tile & 1
                # (the DS offset field has 16 bits: the third slot of p16 is addressed from a second pair of base registers)
                hi = 2 if slot >= 2 else 0
                out.append(f"ds_read_b128 v[{r}:{r + 3}], %[r{self.w}{half + hi}] offset:{(slot - hi) * G.slot64[self.w] + t * 2048}")
            elif self.mode == "kc32":
                out.append(f"ds_read_b128 v[{r}:{r + 3}], %[r{self.w}0] offset:{(tile % G.ring) * 16384 + t * 1024}")
            else:
                base = G.v_ks[self.w] + t
                out.append(f"ds_read_b64_tr_b16 v[{r}:{r + 1}], v{base} offset:{(tile % G.ring) * 16384}")
                out.append(f"ds_read_b64_tr_b16 v[{r + 2}:{r + 3}], v{base} offset:{(tile % G.ring) * 16384 + 2048}")
        return out

    def unit_of_tile(self, t):
        """request unit (slot index for kc64, stage index otherwise) holding tile t"""
        return t >> 1 if self.mode == "kc64" else t

    def pieces(self, unit):
        """[(set M0, issue)] of request unit `unit` (absolute index; its LDS place is unit mod 2 / mod 4)"""
        out = []
        for pc in range(self.npiece):
            if self.mode == "kc64":
                dst = G.region[self.w] + (unit % self.nslot) * G.slot64[self.w] + pc * 1024
            else:
                dst = G.region[self.w] + (unit % G.ring) * 16384 + pc * 1024
            sg = S_G[self.w]
            out.append((f"s_add_u32 m0, %[ldsw{self.w}], {dst}", f"global_load_lds_dwordx4 %[vo{self.w}{pc}], s[{sg}:{sg + 1}]"))
        return out

    def advance(self):
        """the base moves on only while unit index + 1 < number of units (afterwards the last unit is re-fetched)"""
        sg, nx, lim = S_G[self.w], S_NEXT[self.w], S_LIM[self.w]
        return [f"s_add_u32 s{nx}, s{nx}, 1",
                f"s_cmp_lt_u32 s{nx}, s{lim}",
                f"s_cselect_b32 s{S_T}, %[s{self.w}], 0",
                f"s_add_u32 s{sg}, s{sg}, s{S_T}",
                f"s_addc_u32 s{sg + 1}, s{sg + 1}, 0"]

    def units_requested_in_tile(self, t):
        """absolute unit indices requested during tile t of the steady state"""
        if self.mode == "kc64":
            return [(t + 3) >> 1] if t % 2 == 1 else []
        return [t + RING_LEAD]


def mfma(A, B, ti, tj, setp):
    acc = 4 * (G.fb * ti + tj)
    a, b = A.frag(setp, ti), B.frag(setp, tj)
    return f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc + 3}], v[{b}:{b + 3}], v[{a}:{a + 3}], a[{acc}:{acc + 3}]"


class Issue:
    """replays the DMA issue order to derive the s_waitcnt vmcnt counts"""

    def __init__(self):
        self.log = []

    def add(self, tag):
        self.log.append(tag)

    def wait_count(self, needed):
        """pieces issued after the last piece carrying any of the `needed` (operand, unit) tags"""
        last = max((i for i, tag in enumerate(self.log) if tag in needed), default=-1)
        return len(self.log) - 1 - last


def tile_body(A, B, t, sched, issue):
    """k-tile t (absolute, steady state): MFMAs on set t&1, reads of tile t+1 into the other set, DMA requests, wait count"""
    setp = t & 1
    need = {("a", A.unit_of_tile(t + 1)), ("b", B.unit_of_tile(t + 1))}
    n = issue.wait_count(need)
    lines = [f"s_waitcnt vmcnt({n})", "s_waitcnt lgkmcnt(0)", "s_barrier"]
    reads = B.reads(t + 1, setp ^ 1) + A.reads(t + 1, setp ^ 1)
    pieces, adv_after = [], {}
    for op in (A, B):
        for unit in op.units_requested_in_tile(t):
            ps = op.pieces(unit)
            for p in ps:
                pieces.append((p, (op.w, unit)))
            adv_after[len(pieces) - 1] = op.advance()
    NM = G.nm
    fill = {m: [] for m in range(NM)}
    span = sched["read_span"]
    for i, r in enumerate(reads):
        fill[(i * span) // len(reads)].append(r)
    if pieces:
        first = sched["dma_first"]
        step = min(sched["dma_step"], max(2, (NM - 4 - first) // len(pieces)))
        for i, ((setm0, req), tag) in enumerate(pieces):
            m = first + i * step
            fill[m].append(setm0)
            fill[m + 1].append(("DMA", req, tag))
            if i in adv_after:
                for k, ins in enumerate(adv_after[i]):
                    fill[min(NM - 1, m + 2 + k // 3)].append(ins)
    order = [(ti, tj) for ti in range(G.fa) for tj in (range(G.fb) if ti % 2 == 0 or not sched["snake"] else range(G.fb - 1, -1, -1))]
    for m, (ti, tj) in enumerate(order):
        lines.append(mfma(A, B, ti, tj, setp))
        for f in fill[m]:
            if isinstance(f, tuple):
                issue.add(f[2])
                lines.append(f[1])
            else:
                lines.append(f)
    return lines, n


def stream(a_mode, b_mode, sched):
    A, B = Op("a", a_mode), Op("b", b_mode)
    issue = Issue()
    L = ["s_nop 4", f"s_mov_b32 s{S_M0}, m0"]
    for op in (A, B):
        sg = S_G[op.w]
        L += [f"s_mov_b32 s{sg}, %[g{op.w}lo]", f"s_mov_b32 s{sg + 1}, %[g{op.w}hi]", f"s_mov_b32 s{S_NEXT[op.w]}, 0",
              f"s_lshr_b32 s{S_LIM[op.w]}, %[nkt], 1" if op.mode == "kc64" else f"s_mov_b32 s{S_LIM[op.w]}, %[nkt]"]
    L += [f"s_lshr_b32 s{S_CNT}, %[nkt], 2"] if G.trip == 4 else [f"s_mov_b32 s{S_CNT}, %[ntrips]"]
    for op in (A, B):
        if op.mode == "ks":                      # per-block LDS addresses: base + ((t ^ hh) << 5)
            vb = G.v_ks[op.w]
            for t in range(op.nfrag):
                L += [f"v_xor_b32 v{vb + t}, {t}, %[h{op.w}]", f"v_lshl_add_u32 v{vb + t}, v{vb + t}, 5, %[r{op.w}0]"]
    # prologue requests (tile 0's data first), accumulators zeroed underneath
    zero = [f"v_accvgpr_write_b32 a{i}, 0" for i in range(4 * G.nm)]
    zi = 0
    # in the order the steady state would have issued them during tiles -3, -2, -1 (the loop body's wait counts assume it)
    units = [(op, u) for t in range(-max(3, RING_LEAD), 0) for op in (A, B) for u in op.units_requested_in_tile(t) if u >= 0]
    for op, u in units:
        for setm0, req in op.pieces(u):
            L += [setm0, zero[zi], req] + zero[zi + 1:zi + 4]
            zi += 4
            issue.add((op.w, u))
        L += op.advance()
    L += zero[zi:]
    n0 = issue.wait_count({("a", 0), ("b", 0)})
    L += [f"s_waitcnt vmcnt({n0})", "s_barrier"]
    L += B.reads(0, 0) + A.reads(0, 0)
    # two trips replayed: the second one's counts are the steady state (the first trip follows the prologue's issue order, which
    # must give the same counts for the loop to be one body)
    trips = []
    for trip in range(3):
        body, counts = [], []
        for u in range(G.trip):
            lines, n = tile_body(A, B, G.trip * trip + u, sched, issue)
            body += lines
            counts.append(n)
        trips.append((body, counts))
    assert trips[0][1] == trips[1][1] == trips[2][1], f"wait counts differ between trips: {[c for _, c in trips]}"
    L += ["LOOP_%=:"] + trips[1][0]
    L += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", "s_cbranch_scc1 LOOP_%="]
    # surplus DMA landed, surplus fragment reads returned (their VGPRs go back to the compiler), accumulators readable
    L += ["s_waitcnt vmcnt(0)", "s_waitcnt lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_barrier", f"s_mov_b32 m0, s{S_M0}"]
    check_scc(L)
    return ablate(L), trips[1][1], n0


ABLATE = ""
RING_LEAD = 3        # a ring stage is requested this many k-tiles ahead (--ring-lead 4: the 4-stage rings of a16 allow one more, see main)


def ablate(L):
    """timing-only variants of a finished stream (see --ablate)"""
    if not ABLATE:
        return L
    out = []
    in_loop = False
    for ins in L:
        if ins.startswith("LOOP_"):
            in_loop = True
        if ABLATE == "mfma" and ins.startswith("v_mfma"):
            out.append("s_nop 0")
        elif ABLATE == "dma" and in_loop and ins.startswith("global_load_lds"):
            continue
        elif ABLATE == "reads" and in_loop and ins.startswith("ds_read"):
            continue
        elif ABLATE == "barrier" and in_loop and ins == "s_barrier" and not out[-1].startswith("s_nop"):
            continue                       # (the barrier behind the loop's closing s_nops stays: the epilogue reuses the ring)
        else:
            out.append(ins)
    return out


def check_scc(L):
    """SCC producer / consumer pairs must be adjacent (s_add_u32 m0 and the other scalar adds all write SCC)"""
    for i, ins in enumerate(L):
        if ins.startswith("s_addc_u32") or ins.startswith("s_cselect_b32") or ins.startswith("s_cbranch_scc"):
            prev = L[i - 1]
            ok = prev.startswith(("s_add_u32 s", "s_cmp_")) or (ins.startswith("s_cselect") and prev.startswith("s_cselect"))
            assert ok, f"SCC consumer not behind its producer: {prev} / {ins}"
        if ins.startswith("global_load_lds"):
            assert not L[i - 1].startswith("s_add_u32 m0"), "M0 write directly in front of the LDS-DMA that reads it"


def clobbers():
    c = ['"memory"', '"vcc"', '"scc"']
    c += [f'"v{i}"' for i in range(G.vlo, G.vhi)]
    c += [f'"s{i}"' for i in range(S_G["a"], S_LAST + 1)]
    return ", ".join(c)


SCHED = {"read_span": 40, "dma_first": 4, "dma_step": 6, "snake": True}
SCHED_B16 = {"read_span": 24, "dma_first": 2, "dma_step": 3, "snake": True}


def main_b16(args):
    """gemm_b16_loop.inc: A k-contiguous (two 64-k slots), B k-strided (forward) or k-contiguous (dgrad, 32-k ring)"""
    global G
    G = GEOMS["b16"]
    sched = dict(SCHED_B16)
    if args.read_span != SCHED["read_span"]: sched["read_span"] = args.read_span
    if args.dma_first != SCHED["dma_first"]: sched["dma_first"] = args.dma_first
    if args.dma_step != SCHED["dma_step"]: sched["dma_step"] = args.dma_step
    sched["snake"] = not args.no_snake
    out = args.out.replace("gemm_a16_loop.inc", "gemm_b16_loop.inc")          # (an explicit --out is taken as given)
    txt = ["// GENERATED by tools/gen_gemm_a16.py --geom b16 -- do not edit; the generator is the source (design notes in its docstring and in Geom).",
           f"// schedule: {sched}", ""]
    for b_kc in (True, False):
        name = f"NEKO_B16_LOOP_KC_{'KC' if b_kc else 'KS'}"
        L, counts, n0 = stream("kc64", "kc32" if b_kc else "ks", sched)
        txt.append(f"// {name}: vmcnt at the prologue wait {n0}, at the {G.trip} tiles of a trip {counts}")
        txt.append(f"#define {name} \\")
        txt += [f'  "{ins}\\n\\t" \\' for ins in L[:-1]]
        txt.append(f'  "{L[-1]}"')
        txt.append("")
    txt.append(f"#define NEKO_B16_CLOBBERS {clobbers()}")
    txt.append("")
    open(out, "w").write("\n".join(txt))
    print(out, sum(len(t) for t in txt), "bytes")


# ---- p16 (round 6): two waves per SIMD, alternating roles -------------------------------------------------------------------------------
# One workgroup = 8 waves = one 256 x 256 tile; wave w owns rows 128 (w >> 2) .. +127 and columns 64 (w & 3) .. +63: 8 x 4 blocks of
# v_mfma_f32_16x16x32_bf16, 128 accumulators a[0:127] (block (ti, tj) = a[4 (4 ti + tj) : +3]), ONE fragment set of 48 VGPRs (v[64:111]).
# Waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically); the first half (w < 4, "X") and the second half
# ("Y", static s_setprio 1: MI355X_MICROARCH.md "Two waves per SIMD" item 4) run two different instruction streams, and every k-tile j
# is two segments separated by block barriers:
#     S0(j):  X  C(j)   = the 32 MFMAs of tile j, nothing else in the stream      |  Y  L(j)   = fragment reads of tile j + DMA requests
#     -- B0(j): every wave has waited for its own pieces of tile j + 1 -> tile j + 1 is visible; tile j has been read by both halves --
#     S1(j):  X  L(j+1) = fragment reads of tile j + 1 + DMA requests             |  Y  C(j)
#     -- B1(j) --
# so on every SIMD one wave issues matrix instructions back to back while its partner issues the LDS reads and the L2 -> LDS requests
# (one in-order wave per SIMD had to issue all three streams itself: 1385-1550 clocks per k-tile against 1024 of bare MFMAs).
# L(j) requests into what tile j - 1 has freed (both halves have read tile j - 1 before B0(j-1)):
#   32-k ring operand (4 stages): the stage of tile j + 3;
#   k-contiguous operand in 64-k slots of whole 128-B lines (tools/probe/dma_rate_probe.hip: 64-B half-line pieces travel at HALF the
#   rate of whole lines, 28 against 55 B/clk/CU): A in THREE slots (96 KB at 0), B in two (64 KB at 96 KB); the slot freed by an odd tile
#   f is requested piece by piece in L(f + 1 + delay[piece]) -- the delays (P16_DELAY) level the load segments at four requests each.
# The loop body is lcm(6, 4) = 12 k-tiles when A is k-contiguous, 4 otherwise.
P16_DELAY = {}          # (operand, other operand's mode) -> per-piece delays, filled by main_p16


def p16_requests(op, other, j):
    """[(unit, piece)] of operand op that L(j) issues"""
    out = []
    if op.mode == "kc64":
        ns = op.nslot
        for pc, d in enumerate(P16_DELAY[(op.w, other.mode)]):
            assert d <= 2 * ns - 4, "a piece requested in the segment in front of the barrier that needs it"
            f = j - 1 - d
            if f % 2 == 1:
                out.append(((f - 1) // 2 + ns, pc))
    else:
        out += [(j - 1 + G.ring, pc) for pc in range(op.npiece)]
    return [(u, pc) for u, pc in out if u >= 0]


def p16_last_piece(op, other):
    """the piece of a unit that is issued last (the source base advances behind it)"""
    if op.mode != "kc64":
        return op.npiece - 1
    d = P16_DELAY[(op.w, other.mode)]
    return max(range(len(d)), key=lambda pc: (d[pc], pc))


def p16_loader(A, B, j, issue, sched, zero=None, skip=()):
    """L(j): the fragment reads of tile j (none in the prologue, j < 0: accumulator zeroing instead) and the DMA requests; an M0 write
    may not sit directly in front of the LDS-DMA that uses it, so a read (or a zeroing move) goes between them.  Units in `skip` were
    requested by the early block (p16_early): only their bookkeeping (the source base moving on) stays"""
    reads = (B.reads(j, 0) + A.reads(j, 0)) if j >= 0 else []
    reqs = []
    for op, other in ((A, B), (B, A)):
        for unit, pc in p16_requests(op, other, j):
            setm0, req = op.pieces(unit)[pc]
            reqs.append((setm0, req, (op.w, unit), op.advance() if pc == p16_last_piece(op, other) else []))
    lines = []
    ri = min(sched["p16_reads_first"], len(reads))
    lines += reads[:ri]

    def filler():
        nonlocal ri
        if ri < len(reads):
            ri += 1
            return reads[ri - 1]
        if zero:
            return zero.pop(0)
        return "s_nop 0"
    for setm0, req, tag, adv in reqs:
        if tag in skip:
            lines += adv
            continue
        lines += [setm0, filler(), req]
        issue.add(tag)
        lines += adv
        for _ in range(sched["p16_reads_per_piece"] - 1):
            if ri < len(reads) or zero:
                lines.append(filler())
    lines += reads[ri:]
    return lines


def p16_compute(A, B):
    order = [(ti, tj) for ti in range(G.fa) for tj in (range(G.fb) if ti % 2 == 0 else range(G.fb - 1, -1, -1))]
    return [mfma(A, B, ti, tj, 0) for ti, tj in order]


def p16_wait(A, B, issue, j):
    """vmcnt in front of B0(j): this wave's pieces of tile j + 1 have landed"""
    need = {("a", A.unit_of_tile(j + 1)), ("b", B.unit_of_tile(j + 1))}
    for tag in need:
        assert tag in issue.log, f"tile {j + 1}: {tag} not requested before the barrier that publishes it"
    return issue.wait_count(need)


def p16_early(A, B, issue):
    """The early block: the requests of the units that hold k-tile 0, as an asm statement of its own.  gemm_p16.hip issues it for a
    workgroup's NEXT tile in front of the epilogue of the current one (the k-loop has ended behind a block barrier: the ring is free; the
    epilogue's slabs keep clear of the two places these units land in), so that the first k-tile of every tile but a workgroup's first
    is in LDS when its loop starts.  Loads return in order among loads: the loop's own counted waits see these requests as the
    oldest ones whatever stores the epilogue has put between them and the rest of the prologue."""
    E = ["s_nop 4", f"s_mov_b32 s{S_M0}, m0"]
    for op in (A, B):
        sg = S_G[op.w]
        E += [f"s_mov_b32 s{sg}, %[g{op.w}lo]", f"s_mov_b32 s{sg + 1}, %[g{op.w}hi]"]
    tags = []
    for op in (A, B):
        unit = op.unit_of_tile(0)
        for setm0, req in op.pieces(unit):
            E += [setm0, "s_nop 0", req]
            issue.add((op.w, unit))
        tags.append((op.w, unit))
    E += [f"s_mov_b32 m0, s{S_M0}"]
    return E, tuple(tags)


def stream_p16(a_mode, b_mode, sched):
    A, B = Op("a", a_mode), Op("b", b_mode)
    trip = 12 if a_mode == "kc64" and A.nslot == 3 else 4
    L = ["s_nop 4", f"s_mov_b32 s{S_M0}, m0"]
    for op in (A, B):
        sg = S_G[op.w]
        L += [f"s_mov_b32 s{sg}, %[g{op.w}lo]", f"s_mov_b32 s{sg + 1}, %[g{op.w}hi]", f"s_mov_b32 s{S_NEXT[op.w]}, 0",
              f"s_lshr_b32 s{S_LIM[op.w]}, %[nkt], 1" if op.mode == "kc64" else f"s_mov_b32 s{S_LIM[op.w]}, %[nkt]"]
    L += [f"s_mov_b32 s{S_CNT}, %[ntrips]"]
    for op in (A, B):
        if op.mode == "ks":                      # per-block LDS addresses: base + ((t ^ h) << 5), h = hh(k-row) ^ first block of the wave
            vb = G.v_ks[op.w]
            for t in range(op.nfrag):
                L += [f"v_xor_b32 v{vb + t}, {t}, %[h{op.w}]", f"v_lshl_add_u32 v{vb + t}, v{vb + t}, 5, %[r{op.w}0]"]
    zero = [f"v_accvgpr_write_b32 a{i}, 0" for i in range(4 * G.nm)]
    issue = Issue()
    E, skip = p16_early(A, B, issue) if sched.get("p16_early") else ([], ())
    # prologue: what L(-7) .. L(-1) would have requested, in that order (accumulators zeroed underneath)
    need0 = {("a", A.unit_of_tile(0)), ("b", B.unit_of_tile(0))}
    early = False
    for j in range(-7, 0):
        L += p16_loader(A, B, j, issue, dict(sched, p16_reads_per_piece=4), zero, skip)
        if sched["p16_early_wait"] and not early and all(t in issue.log for t in need0):
            # --p16-early-wait: tile 0's units land before the rest of the prologue is requested (every CU of the chip is in its
            # prologue at the same time: ~160 KB per CU requested at once delay the 48 KB the first k-tile needs)
            L += [f"s_waitcnt vmcnt({issue.wait_count(need0)})"]
            early = True
    L += zero
    assert all(t in issue.log for t in need0)
    n0 = issue.wait_count(need0)
    L += [f"s_waitcnt vmcnt({n0})", "s_barrier", "s_cmp_eq_u32 %[half], 0", "s_cbranch_scc0 YPROG_%="]
    # The two halves issue the same requests in the same order (L(0), L(1), ...) and either stream has issued exactly L(0) .. L(j) when it
    # waits in front of B0(j), so one replay of the issue order serves both.
    ld = {0: p16_loader(A, B, 0, issue, sched)}
    counts = {}
    for j in range(3 * trip):
        counts[j] = p16_wait(A, B, issue, j)
        ld[j + 1] = p16_loader(A, B, j + 1, issue, sched)
    comp = p16_compute(A, B)
    # B1 is a scheduling barrier only (LDS visibility and slot reuse hang on B0): --p16-no-b1 drops it, the two halves then overlap
    # [L, C] with [C, L] between two B0s as the hardware arbitrates
    b1 = ["s_barrier"] if sched["p16_b1"] else []

    def tile_x(j):            # C(j) | B0 | L(j + 1) | B1
        return comp + [f"s_waitcnt vmcnt({counts[j]})", "s_barrier"] + ld[j + 1] + ["s_waitcnt lgkmcnt(0)"] + b1

    def tile_y(j):            # L(j) | B0 | C(j) | B1
        return ld[j] + [f"s_waitcnt vmcnt({counts[j]})", "s_waitcnt lgkmcnt(0)", "s_barrier"] + comp + b1

    def body(tile, rep, lo=0, hi=None):
        return [ins for j in range(trip * rep + lo, trip * rep + (trip if hi is None else hi)) for ins in tile(j)]
    # one loop body per half: the first trip (which follows the prologue's issue order) must equal the steady state
    for tile in (tile_x, tile_y):
        assert body(tile, 0) == body(tile, 1) == body(tile, 2), "loop body not periodic from the first trip on"
    assert sum(i == "s_barrier" for i in body(tile_x, 1)) == sum(i == "s_barrier" for i in body(tile_y, 1)) == (2 if sched["p16_b1"] else 1) * trip
    loop_tail = [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0"]

    def tail(tile, tag):
        """contraction ranges that are whole trips plus 4 or 8 k-tiles (K % 128 == 0, e.g. 2048 or 8192): behind the loop the tile index is a
        multiple of `trip` again, so the remainder is the first one or two 4-tile groups of the very same body (%[tail] = (k-tiles % 12) / 4)"""
        if trip == 4:
            return []
        return ["s_cmp_eq_u32 %[tail], 0", f"s_cbranch_scc1 {tag}END_%="] + body(tile, 1, 0, 4) + \
               ["s_cmp_eq_u32 %[tail], 1", f"s_cbranch_scc1 {tag}END_%="] + body(tile, 1, 4, 8) + [f"{tag}END_%=:"]
    X = ld[0] + ["s_waitcnt lgkmcnt(0)", "XLOOP_%=:"] + body(tile_x, 1) + loop_tail + ["s_cbranch_scc1 XLOOP_%="] + tail(tile_x, "X") + ["s_branch PEND_%="]
    Y = (["s_setprio 1"] if sched["p16_setprio"] else []) + ["YLOOP_%=:"] + body(tile_y, 1) + loop_tail + ["s_cbranch_scc1 YLOOP_%="] + tail(tile_y, "Y") + \
        (["s_setprio 0"] if sched["p16_setprio"] else [])
    L += X + ["YPROG_%=:"] + Y + ["PEND_%=:"]
    # surplus DMA landed, surplus fragment reads returned, accumulators readable
    L += ["s_waitcnt vmcnt(0)", "s_waitcnt lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_barrier", f"s_mov_b32 m0, s{S_M0}"]
    check_scc(L)
    return ablate(L), [counts[j] for j in range(trip, 2 * trip)], n0, trip, E


SCHED_P16 = {"p16_reads_first": 0, "p16_reads_per_piece": 2, "p16_setprio": True, "p16_b1": True, "p16_early_wait": False, "p16_early": False}


def main_p16(args):
    """gemm_p16_loop.inc: (A, B) layouts kc/ks (forward), kc/kc (dgrad, LM-head logits), ks/ks (weight gradients)"""
    global G
    G = GEOMS["p16"]
    sched = dict(SCHED_P16)
    sched["p16_reads_first"] = args.p16_reads_first
    sched["p16_reads_per_piece"] = args.p16_reads_per_piece
    sched["p16_setprio"] = not args.p16_no_setprio
    sched["p16_b1"] = not args.p16_no_b1
    sched["p16_early_wait"] = args.p16_early_wait
    sched["p16_early"] = args.p16_early
    G.nslot["a"] = args.p16_nslot_a
    kcb = f"kc{args.p16_kcb}"
    # per-piece delays (load segments after the one that follows the freeing tile): next to a ring operand (2 requests per segment) A's
    # slot goes 2 + 2 -> 4 requests in every load segment; next to a 64-k-slot B (4 requests after every odd tile) A's slot waits one segment
    P16_DELAY.update({("a", "ks"): [0, 0, 1, 1], ("a", "kc32"): [0, 0, 1, 1], ("a", "kc64"): [1, 1, 1, 1], ("b", "kc64"): [0, 0, 0, 0]})
    if args.p16_no_spread or args.p16_nslot_a == 2:
        P16_DELAY.update({("a", "ks"): [0, 0, 0, 0], ("a", "kc32"): [0, 0, 0, 0], ("a", "kc64"): [0, 0, 0, 0]})
    G.npiece64["b"] = 4
    G.slot64["b"] = 32768
    out = args.out.replace("gemm_a16_loop.inc", "gemm_p16_loop.inc")
    txt = ["// GENERATED by tools/gen_gemm_a16.py --geom p16 -- do not edit; the generator is the source (design notes above stream_p16).",
           f"// schedule: {sched}; request delays {dict((k[0] + '|' + k[1], v) for k, v in P16_DELAY.items())}",
           f"#define NEKO_P16_KC_MODE_B {64 if kcb == 'kc64' else 32}",
           f"#define NEKO_P16_TRIP_KC {12 if args.p16_nslot_a == 3 else 4}      // k-tiles per loop trip with a k-contiguous A operand", ""]
    for a_kc, b_kc in ((True, False), (True, True), (False, False)):
        name = f"NEKO_P16_LOOP_{'KC' if a_kc else 'KS'}_{'KC' if b_kc else 'KS'}"
        L, counts, n0, trip, E = stream_p16("kc64" if a_kc else "ks", kcb if b_kc else "ks", sched)
        txt.append(f"// {name}: {trip} k-tiles per loop trip; vmcnt at the prologue wait {n0}, in front of B0 of the tiles of a trip {counts}")
        txt.append(f"#define {name} \\")
        txt += [f'  "{ins}\\n\\t" \\' for ins in L[:-1]]
        txt.append(f'  "{L[-1]}"')
        txt.append("")
        if E:
            txt.append(f"// {name.replace('LOOP', 'EARLY')}: the requests of k-tile 0's units (the loop above starts behind them)")
            txt.append(f"#define {name.replace('LOOP', 'EARLY')} \\")
            txt += [f'  "{ins}\\n\\t" \\' for ins in E[:-1]]
            txt.append(f'  "{E[-1]}"')
            txt.append("")
    txt.append(f"#define NEKO_P16_CLOBBERS {clobbers()}")
    txt.append("")
    open(out, "w").write("\n".join(txt))
    print(out, sum(len(t) for t in txt), "bytes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--geom", default="a16", choices=tuple(GEOMS), help="workgroup geometry (see Geom)")
    ap.add_argument("--ablate", default="", choices=("", "mfma", "dma", "reads", "barrier"),
                    help="timing-only streams (WRONG results) for tools/probe/r05/gemm_loop_ablation.sh: 'mfma' replaces every MFMA by s_nop 0 "
                         "(what the DMA + LDS-read traffic alone costs), 'dma' drops the in-loop DMA requests (matrix pipe + LDS reads alone), "
                         "'reads' drops the fragment reads, 'barrier' the per-k-tile block barriers")
    ap.add_argument("--kc", type=int, default=64, choices=(32, 64), help="layout of a k-contiguous A operand")
    ap.add_argument("--kcb", type=int, default=32, choices=(0, 32, 64),
                    help="layout of a k-contiguous B operand (0: same as --kc).  Default 32: with both operands in two 64-k slots every DMA piece "
                         "of a trip falls into its odd tiles and the queue drains completely every second tile; B on the 4-stage 32-k ring spreads "
                         "them (12 / 4 pieces per tile) -- dgrad fc 65536 x 768 x 3072: 318 -> 301 us, 8192^3 NT -1.3 % (profiles/r04_gemm_a16_ab.txt)")
    ap.add_argument("--read-span", type=int, default=SCHED["read_span"], help="the next tile's fragment reads are spread over the first N MFMAs")
    ap.add_argument("--dma-first", type=int, default=SCHED["dma_first"], help="MFMA index behind which the first DMA piece of a tile is issued")
    ap.add_argument("--dma-step", type=int, default=SCHED["dma_step"], help="MFMAs between two DMA pieces (shrunk when a tile issues many)")
    ap.add_argument("--no-snake", action="store_true", help="row-major MFMA order instead of the serpentine one")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neko_amd", "csrc",
                                                  "gemm_a16_loop.inc"))
    ap.add_argument("--p16-kcb", type=int, default=64, choices=(32, 64), help="p16: a k-contiguous B in two 64-k slots of whole lines (default) or in the 32-k ring")
    ap.add_argument("--p16-nslot-a", type=int, default=3, choices=(2, 3), help="p16: 64-k slots of a k-contiguous A operand (2: four k-tiles per loop trip, any K % 128 == 0)")
    ap.add_argument("--p16-no-b1", action="store_true", help="p16: no barrier between a half's matrix segment and the other half's (one barrier per k-tile)")
    ap.add_argument("--p16-early", action="store_true", help="p16 experiment (tools/probe/p16_tile_walk_early_block.patch, measured level): k-tile 0's requests as a statement of their own")
    ap.add_argument("--p16-early-wait", action="store_true", help="p16: the prologue waits for tile 0's units before it requests the rest")
    ap.add_argument("--p16-no-setprio", action="store_true", help="p16: no static s_setprio 1 on the second half (A/B runs)")
    ap.add_argument("--p16-no-spread", action="store_true", help="p16: a slot's four requests in ONE load segment (A/B runs)")
    ap.add_argument("--p16-reads-first", type=int, default=SCHED_P16["p16_reads_first"], help="p16: fragment reads in front of the first DMA request of a load segment")
    ap.add_argument("--p16-reads-per-piece", type=int, default=SCHED_P16["p16_reads_per_piece"], help="p16: fragment reads issued with every DMA request")
    ap.add_argument("--ring-lead", type=int, default=3, choices=(3, 4),
                    help="k-tiles of lead of the 32-k ring requests (a16 only: 4 = the stage of tile t, whose fragments were read during tile "
                         "t - 1, is refilled with tile t + 4 during tile t -- probe, profiles/r05_gemm_ring_lead.txt)")
    args = ap.parse_args()
    global ABLATE, RING_LEAD
    ABLATE = args.ablate
    RING_LEAD = args.ring_lead
    assert RING_LEAD == 3 or args.geom == "a16"
    if args.geom == "b16":
        return main_b16(args)
    if args.geom == "p16":
        return main_p16(args)
    SCHED.update(read_span=args.read_span, dma_first=args.dma_first, dma_step=args.dma_step, snake=not args.no_snake)
    kc = f"kc{args.kc}"
    kcb = f"kc{args.kcb or args.kc}"
    txt = ["// GENERATED by tools/gen_gemm_a16.py -- do not edit; the generator is the source (design notes in its docstring).",
           f"#define NEKO_A16_KC_MODE_A {args.kc}", f"#define NEKO_A16_KC_MODE_B {args.kcb or args.kc}", ""]
    for a_kc in (True, False):
        for b_kc in (True, False):
            name = f"NEKO_A16_LOOP_{'KC' if a_kc else 'KS'}_{'KC' if b_kc else 'KS'}"
            L, counts, n0 = stream(kc if a_kc else "ks", kcb if b_kc else "ks", SCHED)
            txt.append(f"// {name}: vmcnt at the prologue wait {n0}, at the four tiles of a trip {counts}")
            txt.append(f"#define {name} \\")
            txt += [f'  "{ins}\\n\\t" \\' for ins in L[:-1]]
            txt.append(f'  "{L[-1]}"')
            txt.append("")
    txt.append(f"#define NEKO_A16_CLOBBERS {clobbers()}")
    txt.append("")
    open(args.out, "w").write("\n".join(txt))
    print(args.out, sum(len(t) for t in txt), "bytes")


if __name__ == "__main__":
    sys.exit(main())
