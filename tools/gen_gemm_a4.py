"""Generator of the hand-scheduled main loop of coco-dr_amd/csrc/gemm_a4.hip (writes csrc/gemm_a4_loop.inc).

The kernel: 256 x 256 output tile, FOUR waves (one per SIMD, 2 x 2), 128 x 128 per wave = 8 x 8 blocks of
v_mfma_f32_16x16x32_bf16 in the 256 accumulator registers a[0:255]; operands by LDS-DMA (`buffer_load_dwordx4 ... lds`) in
pieces of 8 rows x 128 bytes (full cache lines: requests of 64-byte half rows moved 20-25 % fewer bytes per second through the
L2, profiles/r06_gemm_a4.md) into TWO 64-KiB slots, one slot = the [256 rows][64 k] images of A and of B (a "K-tile").

Per K-tile T (128 MFMAs per wave: 64 on the fragments of its first 32-deep half, 64 on the second), slot T % 2:
    MFMA   0 ..  31   in their shadow: the 16 ds_read_b128 of (T, second half)
    MFMA  44          s_waitcnt lgkmcnt(0); s_barrier        - everybody has read slot T % 2 for the last time
    MFMA  46 ..  91   the 16 DMA pieces of K-tile T + 2 into slot T % 2 (in place), one per three MFMAs
    MFMA  88          s_waitcnt vmcnt(pieces younger than K-tile T + 1); s_barrier   - K-tile T + 1 has landed for everybody
    MFMA  89 .. 119   the 16 ds_read_b128 of (T + 1, first half)
    MFMA 127          s_waitcnt lgkmcnt(0)
so a K-tile has one whole iteration (~1 us) between its last request and its first use, 64-128 KiB are in flight per CU, and every
non-MFMA instruction sits in the shadow of an MFMA (one wave per SIMD: a 16-clock MFMA leaves ~3 issue slots).  4-byte instructions
are emitted in pairs so the stream of 8-byte instructions stays 8-byte aligned.

LDS image of an operand: row-major, 128 bytes per row, 16-byte chunk c of row r stored at chunk c ^ (r & 6): a fragment read
(16 rows x 4 chunks per ds_read_b128) then hits 16 distinct 16-byte columns in each of its four 16-lane groups (checked below),
and the swizzle depends only on the row inside an 8-row piece, so one per-lane source offset serves every piece.

Register contract (physical registers, see the asm statement in gemm_a4.hip):
    in : s[44:47] = A descriptor, s[48:51] = B descriptor, s52 / s53 = 16 * lda / 16 * ldb (bytes between 8-row pieces),
         s54 = K / 128 (>= 2), s55 = wave * 8192, v0 / v1 = this lane's byte offsets of K-tile 0 in A / B,
         v2 / v3 = the same for the NEXT tile of this workgroup (persistent walk; only the WALK variant reads them; with s70 != 0
         the walk variant also skips its prologue: the previous tile requested K-tiles 0 and 1),
         v4 .. v7 = fragment read addresses in slot 0: A first half, A second half, B first half, B second half
    out: a[0:255] = acc[i][j][r] at a[(8 i + j) * 4 + r]  (i = 16-row block of the wave's 128 rows, j = 16-column block)
    scratch: v8 .. v143, s56 .. s69, s71, m0, scc.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "coco-dr_amd", "csrc", "gemm_a4_loop.inc")

SLOT = 65536          # one K-tile: A image 32 KiB then B image 32 KiB
B_OFF = 32768
RA, RB = "s[44:47]", "s[48:51]"
S_STEP_A, S_STEP_B, S_NBODY, S_WAVE = "s52", "s53", "s54", "s55"
S_OFFA = ["0"] + [f"s{56 + q}" for q in range(7)]   # scalar offsets of pieces 1..7: q * step - (q & 3) * 1024 (the instruction
S_OFFB = ["0"] + [f"s{63 + q}" for q in range(7)]   # offset, which places the piece in LDS, also moves the source address)
S_SKIP = "s70"
S_CNT = "s71"
V_A0, V_B0, V_NA, V_NB = "v0", "v1", "v2", "v3"
V_AD = {("A", 0, 0): "v4", ("A", 1, 0): "v5", ("B", 0, 0): "v6", ("B", 1, 0): "v7",     # (operand, half, slot) -> address register
        ("A", 0, 1): "v8", ("A", 1, 1): "v9", ("B", 0, 1): "v10", ("B", 1, 1): "v11"}
V_A, V_B = "v12", "v13"               # running DMA offsets
V_RNA, V_RNB = "v14", "v15"           # running next-tile offsets
FRAG_A = [16, 80]                     # register set of the first / second half: A fragments (8 x 4 registers), then B
FRAG_B = [48, 112]
LAST_V = 143

OPT = {"dma": True, "reads": True, "barrier": True, "mfma": True, "place": None}   # timing ablations (results are wrong without any of them)


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


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


def dma_piece(operand, q, va, vb):
    v, r, so = (va, RA, S_OFFA[q]) if operand == "A" else (vb, RB, S_OFFB[q])
    return f"buffer_load_dwordx4 {v}, {r}, {so} offen offset:{(q & 3) * 1024} lds"


def frag_reads(tile_slot, half):
    """the 16 fragment reads of one half of a K-tile: B blocks first (all eight feed the first MFMAs), then A"""
    out = [f"ds_read_b128 {vr(FRAG_B[half] + 4 * j)}, {V_AD[('B', half, tile_slot)]} offset:{j * 2048}" for j in range(8)]
    out += [f"ds_read_b128 {vr(FRAG_A[half] + 4 * i)}, {V_AD[('A', half, tile_slot)]} offset:{i * 2048}" for i in range(8)]
    return out


def gen_ktile(e: Emitter, st: Stream, T: int, *, first: bool, dma_tile, dma_next: bool, read_next: bool, wait_tile):
    """One K-tile (128 MFMAs).  T: K-tile index (only T % 2 = slot matters); dma_tile: K-tile id the pieces issued here belong to
    (None: none), dma_next: they read the NEXT output tile's offsets; read_next: issue the fragment reads of (T + 1, first half);
    wait_tile: K-tile that must have landed at the second barrier (None: no vmcnt wait)."""
    slot = T & 1
    extra = {m: [] for m in range(128)}
    if OPT["reads"]:
        for k, r in enumerate(frag_reads(slot, 1)):
            extra[1 + 2 * k].append(("i8", r))
    do_dma = dma_tile is not None and OPT["dma"]
    if do_dma:
        va, vb = (V_RNA, V_RNB) if dma_next else (V_A, V_B)
        pieces = [("A", q) for q in range(8)] + [("B", q) for q in range(8)]
        for k, (op, q) in enumerate(pieces):
            m = 46 + 3 * k
            if q in (0, 4):
                extra[m - 1].append(m0_write(slot, op, q == 4))
            extra[m].append(("dma", dma_piece(op, q, va, vb)))
        extra[94].append(("i8", f"v_add_u32_e32 {va}, 0x80, {va}"))   # (128 is not an inline constant: 8 bytes)
        extra[95].append(("i8", f"v_add_u32_e32 {vb}, 0x80, {vb}"))
    if read_next and OPT["reads"]:
        for k, r in enumerate(frag_reads(slot ^ 1, 0)):
            extra[89 + 2 * k].append(("i8", r))
    for m in range(128):
        h, mm = m >> 6, m & 63
        i, j = mm >> 3, mm & 7
        c = 4 * (8 * i + j)
        acc = f"a[{c}:{c + 3}]"
        src_c = "0" if (first and h == 0) else acc
        if OPT["mfma"] or (first and h == 0):
            e.i8(f"v_mfma_f32_16x16x32_bf16 {acc}, {vr(FRAG_B[h] + 4 * j)}, {vr(FRAG_A[h] + 4 * i)}, {src_c}")
        for kind, txt in extra[m]:
            if kind == "dma":
                st.dma(dma_tile)
                e.i8(txt)
            elif kind == "i8":
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
    experiments: the same stream runs 5-7 % apart at different offsets, profiles/r06_gemm_a4.md)"""
    if OPT["place"] is None:
        e.raw(".p2align 3")
    else:
        e.raw(".p2align 6")
        for _ in range(2 * OPT["place"]):
            e.raw("s_nop 0")


def generate(walk: bool):
    e = Emitter()
    e.raw("s_nop 4")
    for q in range(1, 8):   # scalar offsets of pieces 1..7
        for so, step in ((S_OFFA, S_STEP_A), (S_OFFB, S_STEP_B)):
            e.raw(f"s_mul_i32 {so[q]}, {step}, {q}")
            if q & 3:
                e.raw(f"s_sub_u32 {so[q]}, {so[q]}, 0x{(q & 3) * 1024:x}")
    e.raw(f"v_mov_b32_e32 {V_A}, {V_A0}")
    e.raw(f"v_mov_b32_e32 {V_B}, {V_B0}")
    e.raw(f"v_mov_b32_e32 {V_RNA}, {V_NA}")
    e.raw(f"v_mov_b32_e32 {V_RNB}, {V_NB}")
    for op in "AB":
        for h in (0, 1):
            e.raw(f"v_add_u32_e32 {V_AD[(op, h, 1)]}, 0x10000, {V_AD[(op, h, 0)]}")
    e.raw("s_nop 1")
    st = Stream()
    # ---- prologue: K-tiles 0 and 1 requested into slots 0 and 1, K-tile 0 awaited, its first-half fragments read.  Walk variant, bit 0
    # of s70 set: the previous tile of this workgroup did all of that in its tail (the fragments arrive in v16 .. v79)
    if walk:
        e.raw(f"s_bitcmp1_b32 {S_SKIP}, 0")
        e.raw("s_cbranch_scc1 L_a4_have_%=")
    for T in range(2):
        for op in "AB":
            for q in range(8):
                if q in (0, 4):
                    e.raw(m0_write(T, op, q == 4)[1])
                    e.raw("s_nop 0")
                e.raw(dma_piece(op, q, V_A, V_B))
                st.dma(T)
        e.raw(f"v_add_u32_e32 {V_A}, 0x80, {V_A}")
        e.raw(f"v_add_u32_e32 {V_B}, 0x80, {V_B}")
    e.raw(f"s_waitcnt vmcnt({st.vmcnt_for(0)})")
    e.raw("s_barrier")
    for r in frag_reads(0, 0):
        e.raw(r)
    if walk:
        e.raw("s_branch L_a4_go_%=")
        e.raw("L_a4_have_%=:")
        e.raw(f"v_add_u32_e32 {V_A}, 0x100, {V_A}")
        e.raw(f"v_add_u32_e32 {V_B}, 0x100, {V_B}")
        e.raw("L_a4_go_%=:")
    e.raw(f"s_sub_u32 {S_CNT}, {S_NBODY}, 2")
    e.raw("s_waitcnt lgkmcnt(0)")

    def body(first_body: bool, last_body: bool, T0: int):
        for k in range(2):
            T = T0 + k
            if last_body:
                gen_ktile(e, st, T, first=False, dma_tile=(T + 2) if walk else None, dma_next=True, read_next=(k == 0), wait_tile=T + 1)
            else:
                gen_ktile(e, st, T, first=(first_body and k == 0), dma_tile=T + 2, dma_next=False, read_next=True, wait_tile=T + 1)

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
        e.raw(f"s_bitcmp1_b32 {S_SKIP}, 1")
        e.raw("s_cbranch_scc0 L_a4_end_%=")
        e.raw(f"s_waitcnt vmcnt({st.vmcnt_for(6)})")
        e.raw("s_barrier")
        for r in frag_reads(0, 0):
            e.raw(r)
        e.raw("s_waitcnt lgkmcnt(0)")
        e.raw("L_a4_end_%=:")
    e.raw("s_nop 15")
    e.raw("s_nop 15")
    text = "\n".join(e.lines)
    keep = set(range(16, 80)) if walk else set()   # the walk variant's first-half fragment registers are in/out operands
    clob = [f"v{i}" for i in range(8, LAST_V + 1) if i not in keep] + [f"s{i}" for i in range(56, 70)] + [S_CNT, "scc", "memory"]
    return text, clob


def check_bank_conflicts():
    """ds_read_b128 services a wave in four groups of 16 lanes; a group is conflict-free when its 16 addresses fall into 16
    different 16-byte columns of the 256-byte bank row (MI355X_MICROARCH.md, LDS table)."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for half in (0, 1):
        for g in groups:
            cols = set()
            for lane in g:
                row, ch = lane & 15, 4 * half + (lane >> 4)
                addr = row * 128 + ((ch ^ (row & 6)) << 4)
                cols.add((addr >> 4) & 15)
            assert len(cols) == 16, (half, g, sorted(cols))


def as_c_string(text: str) -> str:
    return "\n".join('  "' + ln + '\\n"' for ln in text.split("\n"))


def main():
    check_bank_conflicts()
    parts = ["// GENERATED by tools/gen_gemm_a4.py - do not edit.  The hand-scheduled main loop of gemm_a4.hip (see that file and the generator).",
             "// clang-format off"]
    variants = [("GEMM_A4_LOOP_ASM", False, {}), ("GEMM_A4_LOOP_ASM_WALK", True, {}),
                ("GEMM_A4_LOOP_ASM_ABL1", False, {"dma": False}), ("GEMM_A4_LOOP_ASM_ABL2", False, {"reads": False}),
                ("GEMM_A4_LOOP_ASM_ABL3", False, {"barrier": False}), ("GEMM_A4_LOOP_ASM_ABL4", False, {"mfma": False}),
                ("GEMM_A4_LOOP_ASM_ABL5", False, {"dma": False, "reads": False, "barrier": False})]
    variants += [(f"GEMM_A4_LOOP_ASM_PLACE{n}", False, {"place": n}) for n in range(8)]
    n_lines = n_mfma = 0
    for name, walk, opt in variants:
        OPT.update({"dma": True, "reads": True, "barrier": True, "mfma": True, "place": None})
        OPT.update(opt)
        text, clob = generate(walk)
        parts.append(f"#define {name} \\")
        parts.append(" \\\n".join(as_c_string(text).split("\n")))
        parts.append("")
        if name == "GEMM_A4_LOOP_ASM":
            n_lines, n_mfma = len(text.splitlines()), text.count("v_mfma")
    for name, walk in (("GEMM_A4_CLOBBERS", False), ("GEMM_A4_WALK_CLOBBERS", True)):
        OPT.update({"dma": True, "reads": True, "barrier": True, "mfma": True, "place": None})
        _, clob = generate(walk)
        parts.append(f"#define {name} " + ", ".join(f'"{c}"' for c in clob))
    parts.append("#define GEMM_A4_ACC_CLOBBERS " + ", ".join(f'"a{i}"' for i in range(256)))
    parts.append("// clang-format on")
    with open(OUT, "w") as f:
        f.write("\n".join(parts) + "\n")
    print(f"wrote {OUT}: {n_lines} lines, {n_mfma} MFMAs in the shipped variant")


if __name__ == "__main__":
    sys.exit(main())
