"""Generator of the hand-scheduled main loop of coco-dr_amd/csrc/gemm_a4.hip (writes csrc/gemm_a4_loop.inc).

The kernel: 256 x 256 output tile, FOUR waves (one per SIMD, 2 x 2), 128 x 128 per wave = 8 x 8 blocks of
v_mfma_f32_16x16x32_bf16 in the 256 accumulator registers a[0:255]; operands by LDS-DMA (`buffer_load_dwordx4 ... lds`) in
1-KiB pieces of whole cache lines (requests of 64-byte half rows moved 20-25 % fewer bytes per second through the L2,
profiles/r06_gemm_a4.md) into TWO 64-KiB slots, one slot = the 64-deep images of A and of B (a "K-tile").

Per K-tile T (128 MFMAs per wave: 64 on the fragments of its first 32-deep half, 64 on the second), slot T % 2:
    MFMA   1 ..  32   in their shadow: the fragment reads of (T, second half)
    MFMA  44          s_waitcnt lgkmcnt(0); s_barrier        - everybody has read slot T % 2 for the last time
    MFMA  46 ..  91   the 16 DMA pieces of K-tile T + 2 into slot T % 2 (in place), one per three MFMAs
    MFMA  88          s_waitcnt vmcnt(pieces younger than K-tile T + 1); s_barrier   - K-tile T + 1 has landed for everybody
    MFMA  89 .. 120   the fragment reads of (T + 1, first half)
    MFMA 127          s_waitcnt lgkmcnt(0)
so a K-tile has one whole iteration (~1 us) between its last request and its first use, 64-128 KiB are in flight per CU, and every
non-MFMA instruction sits in the shadow of an MFMA (one wave per SIMD: a 16-clock MFMA leaves ~3 issue slots).  4-byte instructions
are emitted in pairs so the stream of 8-byte instructions stays 8-byte aligned.

Operand images (either operand, by the layout of its matrix in memory):
  * contraction index fastest (A of every form but TN, B of NT): [256 rows][64 k], 128 bytes per row, 16-byte chunk c of row r
    stored at chunk c ^ (r & 6); a piece = 8 rows; one ds_read_b128 per 16-row block and half;
  * contraction index slowest ("tr": B of NN = [K, N] weights, A and B of TN = the weight gradient's [tokens, .] operands):
    [64 k][256 columns], 512 bytes per row, chunk c of row k stored at chunk c ^ ((k & 3) << 2 | ((k >> 3) & 1) << 1); a piece =
    2 rows; two ds_read_b64_tr_b16 per 16-column block and half (k .. k + 3 and k + 4 .. k + 7 of a lane group's 8 k).
Both swizzles make every fragment read conflict-free (checked below) and are XORs of address bits the per-lane offsets do not
otherwise touch, so the variants a wave needs are made inside the statement from ONE per-lane input each.

Register contract (physical registers, see the asm statements in gemm_a4.hip):
    in : s[44:47] = A descriptor, s[48:51] = B descriptor, s52 / s53 = bytes between consecutive pieces of A / B (16 ld, or 4 ld
         for a tr operand), s54 = K / 128 (>= 2), s55 = wave * 8192, s72 / s73 = bytes per K-tile of a tr operand (128 ld),
         v0 / v1 = this lane's byte offsets of K-tile 0 in A / B, v2 / v3 = the same for the NEXT tile of this workgroup
         (walk variants), v4 .. v7 = fragment read addresses in slot 0: A first half (tr: block 0), A second half, B first half
         (tr: block 0), B second half; s70 (walk variants): bit 0 = the previous tile of this workgroup requested K-tiles 0 and 1
         and left the first-half fragments of K-tile 0 in v16 .. v79, bit 1 = there is a next tile
    out: a[0:255] = acc[i][j][r] at a[(8 i + j) * 4 + r]  (i = 16-row block of the wave's 128 rows, j = 16-column block);
         walk variants: v16 .. v79 in/out (see s70)
    scratch: v8 .. v143 (.. v187 with a tr operand), s56 .. s69, s71, m0, scc.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "coco-dr_amd", "csrc", "gemm_a4_loop.inc")

SLOT = 65536          # one K-tile: A image 32 KiB then B image 32 KiB
B_OFF = 32768
RSRC = {"A": "s[44:47]", "B": "s[48:51]"}
S_STEP = {"A": "s52", "B": "s53"}
S_NBODY, S_WAVE = "s54", "s55"
S_OFF = {"A": ["0"] + [f"s{56 + q}" for q in range(7)],   # scalar offsets of pieces 1..7: q * step - (q & 3) * 1024 (the instruction
         "B": ["0"] + [f"s{63 + q}" for q in range(7)]}   # offset, which places the piece in LDS, also moves the source address)
S_FLAGS = "s70"
S_CNT = "s71"
S_KSTEP = {"A": "s72", "B": "s73"}
V_IN_OFF = {"A": "v0", "B": "v1"}
V_IN_NEXT = {"A": "v2", "B": "v3"}
V_IN_AD = {("A", 0): "v4", ("A", 1): "v5", ("B", 0): "v6", ("B", 1): "v7"}
FRAG = {"A": [16, 80], "B": [48, 112]}   # register set of the first / second half: A fragments (8 x 4 registers), then B

OPT = {"dma": True, "reads": True, "barrier": True, "mfma": True, "place": None}   # timing ablations (wrong results without any)


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


class Regs:
    """VGPR plan of one form.  tr[op]: the operand's contraction index is the slow one of its matrix."""

    def __init__(self, ta: bool, tb: bool):
        self.tr = {"A": ta, "B": tb}
        self.run = {"A": ["v12"], "B": ["v13"]}      # running DMA offsets (tr: one per swizzle class of a piece)
        self.nxt = {"A": ["v14"], "B": ["v15"]}      # ... of the next tile
        self.ad = {}                                 # fragment read addresses
        for k, (op, h) in enumerate((("A", 0), ("A", 1), ("B", 0), ("B", 1))):
            self.ad[(op, h, 0)] = V_IN_AD[(op, h)]
            self.ad[(op, h, 1)] = f"v{8 + k}"
        nxt_free = 144
        for op in "AB":
            if self.tr[op]:
                self.run[op] += [f"v{nxt_free + c}" for c in range(3)]
                self.nxt[op] += [f"v{nxt_free + 3 + c}" for c in range(3)]
                nxt_free += 6
                for slot in (0, 1):
                    for j in range(8):
                        if j == 0 and slot == 0:
                            self.ad[(op, "tr", j, slot)] = V_IN_AD[(op, 0)]
                        else:
                            self.ad[(op, "tr", j, slot)] = f"v{nxt_free}"
                            nxt_free += 1
        self.last_v = max(143, nxt_free - 1)


class Emitter:
    def __init__(self):
        self.lines = []
        self.pending4 = 0

    def raw(self, s):
        self.lines.append(s)

    def i8(self, s):
        assert self.pending4 == 0, "8-byte instruction behind an unpaired 4-byte one: " + s
        self.lines.append(s)

    def i4(self, s):
        self.lines.append(s)
        self.pending4 ^= 1

    def pad(self):
        if self.pending4:
            self.i4("s_nop 0")


class Stream:
    """Program-order bookkeeping of one wave's DMA pieces: which K-tile every outstanding piece belongs to."""

    def __init__(self):
        self.issued = []

    def dma(self, tile):
        self.issued.append(tile)

    def vmcnt_for(self, tile):
        """largest vmcnt that guarantees every piece of K-tiles <= `tile` has landed (pieces complete in order)"""
        last = -1
        for k, t in enumerate(self.issued):
            if t <= tile:
                last = k
        return len(self.issued) - 1 - last


def m0_write(slot, operand, hi):
    off = slot * SLOT + (B_OFF if operand == "B" else 0) + (4096 if hi else 0)
    if off == 0:   # (a zero literal would assemble to the 4-byte inline-constant form)
        return ("i4", f"s_mov_b32 m0, {S_WAVE}")
    return ("i8", f"s_add_u32 m0, {S_WAVE}, 0x{off:x}")


def piece_class(q):
    """swizzle class of piece q of a tr operand (rows 2 q, 2 q + 1 of the wave's 16): k bit 1 and k bit 3"""
    return (q & 1) | (((q >> 2) & 1) << 1)


CLASS_XOR = [0x00, 0x80, 0x20, 0xa0]   # byte-offset XOR of class c against class 0: chunk ^ 8 (k bit 1 -> (k & 3) << 2), chunk ^ 2 (k bit 3)


def dma_piece(R: Regs, operand, q, regs):
    v = regs[operand][piece_class(q) if R.tr[operand] else 0]
    return f"buffer_load_dwordx4 {v}, {RSRC[operand]}, {S_OFF[operand][q]} offen offset:{(q & 3) * 1024} lds"


def advance(R: Regs, regs, ktiles=1):
    """instructions that move the running offsets `regs` forward by `ktiles` K-tiles: [(kind, text)]"""
    out = []
    for op in "AB":
        for v in regs[op]:
            if R.tr[op]:
                out += [("i4", f"v_add_u32_e32 {v}, {S_KSTEP[op]}, {v}")] * ktiles
            else:
                out.append(("i8", f"v_add_u32_e32 {v}, 0x{0x80 * ktiles:x}, {v}"))   # (128 is not an inline constant: 8 bytes)
    return out


def frag_reads(R: Regs, slot, half):
    """the fragment reads of one half of a K-tile: B blocks first (all eight feed the first MFMAs), then A"""
    out = []
    for op in "BA":
        for blk in range(8):
            base = FRAG[op][half] + 4 * blk
            if R.tr[op]:
                ad = R.ad[(op, "tr", blk, slot)]
                out.append(f"ds_read_b64_tr_b16 {vr(base, 2)}, {ad} offset:{half * 16384}")
                out.append(f"ds_read_b64_tr_b16 {vr(base + 2, 2)}, {ad} offset:{half * 16384 + 2048}")
            else:
                out.append(f"ds_read_b128 {vr(base)}, {R.ad[(op, half, slot)]} offset:{blk * 2048}")
    return out


def spread(n, first, span):
    """n instructions over MFMA slots first .. first + span - 1"""
    return [first + (k * span) // n for k in range(n)]


def gen_ktile(R: Regs, e: Emitter, st: Stream, T: int, *, first: bool, dma_tile, dma_next: bool, read_next: bool, wait_tile):
    """One K-tile (128 MFMAs).  T: K-tile index (only T % 2 = slot matters); dma_tile: K-tile id the pieces issued here belong to
    (None: none), dma_next: they read the NEXT output tile's offsets; read_next: issue the fragment reads of (T + 1, first half);
    wait_tile: K-tile that must have landed at the second barrier (None: no vmcnt wait)."""
    slot = T & 1
    extra = {m: [] for m in range(128)}
    if OPT["reads"]:
        rd = frag_reads(R, slot, 1)
        for m, r in zip(spread(len(rd), 1, 32), rd):
            extra[m].append(("i8", r))
    if dma_tile is not None and OPT["dma"]:
        regs = R.nxt if dma_next else R.run
        pieces = [("A", q) for q in range(8)] + [("B", q) for q in range(8)]
        for k, (op, q) in enumerate(pieces):
            m = 46 + 3 * k
            if q in (0, 4):
                extra[m - 1].append(m0_write(slot, op, q == 4))
            extra[m].append(("dma", dma_piece(R, op, q, regs)))
        m, used = 94, 0
        for kind, txt in advance(R, regs):   # one 8-byte or two 4-byte instructions per MFMA slot
            size = 8 if kind == "i8" else 4
            if used + size > 8:
                m, used = m + 1, 0
            extra[m].append((kind, txt))
            used += size
    if read_next and OPT["reads"]:
        rd = frag_reads(R, slot ^ 1, 0)
        for m, r in zip(spread(len(rd), 89, 32), rd):
            extra[m].insert(0, ("i8", r))
    for m in range(128):
        h, mm = m >> 6, m & 63
        i, j = mm >> 3, mm & 7
        c = 4 * (8 * i + j)
        acc = f"a[{c}:{c + 3}]"
        src_c = "0" if (first and h == 0) else acc
        if OPT["mfma"] or (first and h == 0):
            e.i8(f"v_mfma_f32_16x16x32_bf16 {acc}, {vr(FRAG['B'][h] + 4 * j)}, {vr(FRAG['A'][h] + 4 * i)}, {src_c}")
        for kind, txt in extra[m]:
            if kind == "dma":
                e.pad()
                st.dma(dma_tile)
                e.i8(txt)
            elif kind == "i8":
                e.pad()
                e.i8(txt)
            else:
                e.i4(txt)
        e.pad()
        if m == 44:   # everybody has read slot T % 2 for the last time (its DMA pieces follow)
            e.i4("s_waitcnt lgkmcnt(0)")
            e.i4("s_barrier" if OPT["barrier"] else "s_nop 0")
        if m == 88 and read_next:   # K-tile T + 1 has landed, for everybody
            e.i4(f"s_waitcnt vmcnt({st.vmcnt_for(wait_tile)})" if (wait_tile is not None and OPT["dma"]) else "s_nop 0")
            e.i4("s_barrier" if OPT["barrier"] else "s_nop 0")
    if read_next:
        e.i4("s_waitcnt lgkmcnt(0)")
        e.i4("s_nop 0")


def place(e):
    """where a loop body starts: 8-byte aligned by default; `place` = n pins it to 8 n bytes behind a 64-byte boundary (placement
    experiments: the offsets run within +-1.5 % of each other, profiles/r06_gemm_a4.md)"""
    if OPT["place"] is None:
        e.raw(".p2align 3")
    else:
        e.raw(".p2align 6")
        for _ in range(2 * OPT["place"]):
            e.raw("s_nop 0")


def generate(walk: bool, ta: bool = False, tb: bool = False):
    R = Regs(ta, tb)
    e = Emitter()
    e.raw("s_nop 4")
    for q in range(1, 8):   # scalar offsets of pieces 1..7
        for op in "AB":
            e.raw(f"s_mul_i32 {S_OFF[op][q]}, {S_STEP[op]}, {q}")
            if q & 3:
                e.raw(f"s_sub_u32 {S_OFF[op][q]}, {S_OFF[op][q]}, 0x{(q & 3) * 1024:x}")
    for op in "AB":   # running offsets (this tile, next tile) and the fragment addresses the statement derives from its inputs
        for regs, src in ((R.run, V_IN_OFF), (R.nxt, V_IN_NEXT)):
            for c, v in enumerate(regs[op]):
                e.raw(f"v_mov_b32_e32 {v}, {src[op]}" if c == 0 else f"v_xor_b32_e32 {v}, 0x{CLASS_XOR[c]:x}, {src[op]}")
        if R.tr[op]:
            for slot in (0, 1):
                for j in range(8):
                    if j == 0 and slot == 0:
                        continue
                    if slot == 0:
                        e.raw(f"v_xor_b32_e32 {R.ad[(op, 'tr', j, 0)]}, 0x{j << 5:x}, {V_IN_AD[(op, 0)]}")
                    else:
                        e.raw(f"v_add_u32_e32 {R.ad[(op, 'tr', j, 1)]}, 0x10000, {R.ad[(op, 'tr', j, 0)]}")
        else:
            for h in (0, 1):
                e.raw(f"v_add_u32_e32 {R.ad[(op, h, 1)]}, 0x10000, {R.ad[(op, h, 0)]}")
    e.raw("s_nop 1")
    st = Stream()
    # ---- prologue: K-tiles 0 and 1 requested into slots 0 and 1, K-tile 0 awaited, its first-half fragments read.  Walk variant, bit 0
    # of s70 set: the previous tile of this workgroup did all of that in its tail (the fragments arrive in v16 .. v79)
    if walk:
        e.raw(f"s_bitcmp1_b32 {S_FLAGS}, 0")
        e.raw("s_cbranch_scc1 L_a4_have_%=")
    for T in range(2):
        for op in "AB":
            for q in range(8):
                if q in (0, 4):
                    e.raw(m0_write(T, op, q == 4)[1])
                    e.raw("s_nop 0")
                e.raw(dma_piece(R, op, q, R.run))
                st.dma(T)
        for _, txt in advance(R, R.run):
            e.raw(txt)
    e.raw(f"s_waitcnt vmcnt({st.vmcnt_for(0)})")
    e.raw("s_barrier")
    for r in frag_reads(R, 0, 0):
        e.raw(r)
    if walk:
        e.raw("s_branch L_a4_go_%=")
        e.raw("L_a4_have_%=:")
        for _, txt in advance(R, R.run, 2):
            e.raw(txt)
        e.raw("L_a4_go_%=:")
    e.raw(f"s_sub_u32 {S_CNT}, {S_NBODY}, 2")
    e.raw("s_waitcnt lgkmcnt(0)")

    def body(first_body: bool, last_body: bool, T0: int):
        for k in range(2):
            T = T0 + k
            if last_body:
                gen_ktile(R, e, st, T, first=False, dma_tile=(T + 2) if walk else None, dma_next=True, read_next=(k == 0), wait_tile=T + 1)
            else:
                gen_ktile(R, e, st, T, first=(first_body and k == 0), dma_tile=T + 2, dma_next=False, read_next=True, wait_tile=T + 1)

    # ---- first body (K-tile 0's first half starts the accumulators: srcC = 0)
    place(e)
    body(True, False, 0)
    e.raw(f"s_cmp_eq_u32 {S_CNT}, 0")
    e.raw("s_cbranch_scc1 L_a4_last_%=")
    # ---- steady body (the bookkeeping is periodic: every body starts with the pieces of one K-tile outstanding)
    place(e)
    e.raw("L_a4_loop_%=:")
    body(False, False, 2)
    e.raw(f"s_sub_u32 {S_CNT}, {S_CNT}, 1")
    e.raw(f"s_cmp_eq_u32 {S_CNT}, 0")
    e.raw("s_cbranch_scc0 L_a4_loop_%=")
    # ---- last body
    place(e)
    e.raw("L_a4_last_%=:")
    body(False, True, 4)
    if walk:
        # ---- tail (bit 1 of s70: this workgroup has a next tile): its K-tile 0 - requested two K-tiles ago - is awaited HERE, in
        # front of the epilogue's stores, and its first-half fragments are read into v16 .. v79, which travel through the epilogue
        # as operands of the statement.  (Loads and stores share vmcnt and complete out of order with respect to each other: a
        # wait for these pieces behind the stores would wait for the stores as well.)
        e.raw(f"s_bitcmp1_b32 {S_FLAGS}, 1")
        e.raw("s_cbranch_scc0 L_a4_end_%=")
        e.raw(f"s_waitcnt vmcnt({st.vmcnt_for(6)})")
        e.raw("s_barrier")
        for r in frag_reads(R, 0, 0):
            e.raw(r)
        e.raw("s_waitcnt lgkmcnt(0)")
        e.raw("L_a4_end_%=:")
    e.raw("s_nop 15")
    e.raw("s_nop 15")
    text = "\n".join(e.lines)
    keep = set(range(16, 80)) if walk else set()   # the walk variants' first-half fragment registers are in/out operands
    clob = [f"v{i}" for i in range(8, R.last_v + 1) if i not in keep] + [f"s{i}" for i in range(56, 70)] + [S_CNT, "scc", "memory"]
    return text, clob


# lane groups one LDS cycle serves (MI355X_MICROARCH.md, LDS table): a group is conflict-free when its accesses cover every bank once
B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def check_bank_conflicts():
    # row-major image, ds_read_b128: 16 lanes x 16 bytes = the 256-byte bank row
    for half in (0, 1):
        for g in B128_GROUPS:
            cols = set()
            for lane in g:
                row, ch = lane & 15, 4 * half + (lane >> 4)
                addr = row * 128 + ((ch ^ (row & 6)) << 4)
                cols.add((addr >> 4) & 15)
            assert len(cols) == 16, (half, g, sorted(cols))
    # tr image, ds_read_b64_tr_b16: 32 lanes x 8 bytes = the 256-byte bank row; and the block-j address is base ^ (j << 5)
    for wsel in (0, 1):
        for j in range(8):
            for hi in (0, 1):
                for half32 in (0, 1):
                    banks = set()
                    for lane in range(32 * half32, 32 * half32 + 32):
                        c, g = lane & 15, lane >> 4
                        row = 8 * g + (c >> 2) + 4 * hi
                        col = wsel * 128 + 16 * j + 4 * (c & 3)
                        sw = ((row & 3) << 2) | (((row >> 3) & 1) << 1)
                        addr = row * 512 + (((col >> 3) ^ sw) << 4) + (col & 7) * 2
                        col0 = wsel * 128 + 4 * (c & 3)
                        base0 = (8 * g + (c >> 2)) * 512 + (((col0 >> 3) ^ ((((c >> 2) & 3) << 2) | ((g & 1) << 1))) << 4) + (col0 & 7) * 2
                        assert addr == (base0 ^ (j << 5)) + 2048 * hi, "the block-j address is not base ^ (j << 5)"
                        banks.add((addr >> 3) & 31)
                    assert len(banks) == 32, (wsel, j, hi, half32, sorted(banks))


def as_c_string(text: str) -> str:
    return "\n".join('  "' + ln + '\\n"' for ln in text.split("\n"))


def main():
    check_bank_conflicts()
    parts = ["// GENERATED by tools/gen_gemm_a4.py - do not edit.  The hand-scheduled main loop of gemm_a4.hip (see that file and the generator).",
             "// clang-format off"]
    variants = [("GEMM_A4_LOOP_ASM", False, False, False, {}), ("GEMM_A4_LOOP_ASM_WALK", True, False, False, {}),
                ("GEMM_A4_LOOP_ASM_WALK_NN", True, False, True, {}), ("GEMM_A4_LOOP_ASM_WALK_TN", True, True, True, {}),
                ("GEMM_A4_LOOP_ASM_ABL1", False, False, False, {"dma": False}), ("GEMM_A4_LOOP_ASM_ABL2", False, False, False, {"reads": False}),
                ("GEMM_A4_LOOP_ASM_ABL3", False, False, False, {"barrier": False}), ("GEMM_A4_LOOP_ASM_ABL4", False, False, False, {"mfma": False}),
                ("GEMM_A4_LOOP_ASM_ABL5", False, False, False, {"dma": False, "reads": False, "barrier": False})]
    n_lines = n_mfma = 0
    for name, walk, ta, tb, opt in variants:
        OPT.update({"dma": True, "reads": True, "barrier": True, "mfma": True, "place": None})
        OPT.update(opt)
        text, clob = generate(walk, ta, tb)
        parts.append(f"#define {name} \\")
        parts.append(" \\\n".join(as_c_string(text).split("\n")))
        parts.append("")
        if name == "GEMM_A4_LOOP_ASM":
            n_lines, n_mfma = len(text.splitlines()), text.count("v_mfma")
    OPT.update({"dma": True, "reads": True, "barrier": True, "mfma": True, "place": None})
    for name, walk, ta, tb in (("GEMM_A4_CLOBBERS", False, False, False), ("GEMM_A4_WALK_CLOBBERS", True, False, False),
                               ("GEMM_A4_WALK_CLOBBERS_NN", True, False, True), ("GEMM_A4_WALK_CLOBBERS_TN", True, True, True)):
        _, clob = generate(walk, ta, tb)
        parts.append(f"#define {name} " + ", ".join(f'"{c}"' for c in clob))
    parts.append("#define GEMM_A4_ACC_CLOBBERS " + ", ".join(f'"a{i}"' for i in range(256)))
    parts.append("// clang-format on")
    with open(OUT, "w") as f:
        f.write("\n".join(parts) + "\n")
    print(f"wrote {OUT}: {n_lines} lines, {n_mfma} MFMAs in the shipped NT variant")


if __name__ == "__main__":
    sys.exit(main())
