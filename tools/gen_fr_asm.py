#!/usr/bin/env python3
"""Generates mobilequant_amd/csrc/mq_gemm_fr_asm.inc: the WHOLE 256x176 int8 GEMM workgroup program (prologue, main
loop, epilogue) as hand-scheduled gfx950 ISA -- GEMM variant "t256x176_w8x1_fr_asm" (mq_w8a8_linear_tiled, 8-bit output grid).

Why a second generated kernel (the first one is tools/gen_pp_asm.py, a ping-pong main loop between a C++ prologue and epilogue):
stamps of that kernel (profiles/r02/a_stamps_*.log) show 1 866 cycles per K = 128 stage against 1 408 of pure MFMA time, and the
loop's four s_barriers per stage alone cost ~140 of them (MFMA-only ablation: 1 547).  Here the eight waves run FREE: every wave is
software-pipelined on its own (W fragments of the next k-step are read from the LDS into a second register set while the MFMAs of
the current k-step issue), the two waves of a SIMD share the matrix pipe by hardware arbitration instead of by barrier-separated
phases, and ONE s_barrier per stage orders the shared W ring.  Prologue and epilogue are ISA too, so the accumulators never move:

  VGPR  v[0:87]     accumulators acc(i,j) = v[(2j+i)*4 : +3]  (i = A fragment 0/1, j = W fragment 0..10): initialised with the
                    zero-point correction by v_mad_i32_i24, converted in place by the epilogue (cvt, fma, cvt_pk_u8)
        v[98:127]   temporaries;  v[88:97] is left to hipcc for the eight input operands
  AGPR  a[0:43]     W fragments, register set 0 (MFMA srcA);  a[44:87] set 1
        a[88:103]   A fragments of even stages [ks0 i0][ks0 i1][ks1 i0][ks1 i1] (MFMA srcB), loaded straight from the
        a[104:119]  fragment-blocked activations (mq_quantize_tiled): one fully coalesced 1-KiB global_load_dwordx4 each;  odd stages
  LDS   [0, 4*22528)        W ring: four K = 128 stages of 176 rows x 128 B, filled by LDS-DMA three stages ahead (XOR-swizzled
                            source addresses, conflict-free ds_read_b128 -- same image as the other variants)
        [90112, +2816)      per-n epilogue vectors alpha' | bias' | -w_zp | col_term
        [92928, +8*5632)    per-wave staging tiles of the epilogue (2 x 16 rows x 176 B): disjoint from the ring, so a wave
                            starts its epilogue while slower waves still read W

Per stage t (A register set t & 1, ring slot t % 4) a wave issues, in this order:
  wait A(t).ks0 | k-step 0: 22 MFMA (W set 0) + 11 ds_read W(t).ks1 -> set 1 + A(t+1).ks1 (2 loads) + its W(t+3) LDS-DMA pieces
  lgkmcnt(0), wait A(t).ks1 and own W(t+1) pieces, s_barrier
  k-step 1: 22 MFMA (W set 1) + 11 ds_read W(t+1).ks0 -> set 0 + A(t+2).ks0 (2 loads); lgkmcnt(0)
All vmcnt immediates are computed by simulating the wave's VMEM queue (class Queue); the steady-state body is checked to be a fixed
point.  Waves 0-5 own three W pieces per stage, waves 6-7 two: the program is emitted twice (no branch inside the loop).
K % 256 == 0 and K >= 768 (two head stages + pairs of steady stages + four tail stages).

Run:  python tools/gen_fr_asm.py   (writes the .inc next to mq_gemm.hip; the file is committed and checked by the tests)."""
import os

BK = 128
RING = 4

# Variants: the same program parameterised by the N tile (FN W fragments of 16 columns), the number of waves stacked along M (32 rows
# each) and the epilogue.  "fr" is the original 256 x 176 kernel (its .inc must stay byte-identical: tests/test_cabi.py).
VARIANTS = {
    #  name     BN   NW  epilogue  macro prefix   file
    "fr":      (176, 8, "u8",   "MQ_FR",      "mq_gemm_fr_asm.inc"),
    "fr128":   (128, 8, "u8",   "MQ_FR128",   "mq_gemm_fr128_asm.inc"),      # q|k|v (N = 2560): 256 x 128 tiles, per-column output grids
    "fr128r":  (128, 4, "f32r", "MQ_FR128R",  "mq_gemm_fr128r_asm.inc"),     # o_proj / w2 (N = 2048): 128 x 128 tiles, x + Q16(linear) in fp32
    "fr128r8": (128, 8, "f32r", "MQ_FR128R8", "mq_gemm_fr128r8_asm.inc"),    # the same epilogue on 256 x 128 tiles
    # round 4: 256 x 128 tiles with the K loop SPLIT over two workgroups (2 x tiles workgroups, each half of K): twice the MFMAs per
    # operand byte and two waves per SIMD where "fr128r" has one, on all 256 CUs at M = 2048.  The two workgroups of a tile swap half of
    # their int32 partial sums through a scratch buffer (wave to wave, one flag each way) and each finishes one 16-row block per wave
    "fr128rs": (128, 8, "f32rs", "MQ_FR128RS", "mq_gemm_fr128rs_asm.inc"),
    "fr160":   (160, 4, "u8",   "MQ_FR160",   "mq_gemm_fr160_asm.inc"),      # q|k|v at M = 2048: 128 x 160 tiles = 16 x 16 = one per CU
    # w3 of a gated FFN with the rest of the chain in its epilogue: its 8-bit output index and w1's (read back from the first launch's
    # output) go through the 256 x 256 gated table (64 KiB, LDS-resident) -> w2's int8 input image, fragment-blocked, + row sums
    "frg":     (176, 8, "gate", "MQ_FRG",     "mq_gemm_frg_asm.inc"),
    "frg128":  (128, 8, "gate", "MQ_FRG128",  "mq_gemm_frg128_asm.inc"),     # the same on 256 x 128 tiles (Gemma's FFN: N = 16384)
    # round 4: PACKED 4-bit weights (mq_pack_w4: two unsigned nibbles per byte) -- LDS-DMA of the packed rows, ONE ds_read_b128 per 16
    # output columns and K = 128 stage, nibbles split in registers (v_and / v_lshrrev) under the MFMAs (program_w4 below)
    "frw4":     (176, 8, "u8w4", "MQ_FRW4",    "mq_gemm_frw4_asm.inc"),       # 256 x 176 tiles, one output grid (TinyLlama / StableLM w1, w3)
    "frw4_128": (128, 8, "u8w4", "MQ_FRW4_128", "mq_gemm_frw4_128_asm.inc"),  # 256 x 128 tiles, per-column grids (q | k | v; Gemma's FFN)
    # the same packed image, expanded ONCE per workgroup instead of once per wave: every wave loads the 16-row pieces it owns straight into
    # registers (no LDS-DMA), splits the nibbles and writes the int8 rows into the W ring; the loop, the deferred init and the final block
    # are the int8 kernel's (frw4 spends 12 VALU per 4 MFMAs in EVERY wave on the unpack and measured 30 % slower than the int8 image)
    "frw4x":     (176, 8, "u8w4x", "MQ_FRW4X",     "mq_gemm_frw4x_asm.inc"),
    "frw4x_128": (128, 8, "u8w4x", "MQ_FRW4X_128", "mq_gemm_frw4x_128_asm.inc"),
    # ... in front of the gate epilogue (w3 of a gated FFN, packed): 256 x 176 and 256 x 128 tiles
    "frgw4x":     (176, 8, "gatew4x", "MQ_FRGW4X",     "mq_gemm_frgw4x_asm.inc"),
    "frgw4x_128": (128, 8, "gatew4x", "MQ_FRGW4X_128", "mq_gemm_frgw4x_128_asm.inc"),
    # ... and in front of the residual epilogue: o_proj / w2 (N = 2048) from packed nibbles on 128 x 128 tiles, four waves
    "frw4x_128r": (128, 4, "f32rw4x", "MQ_FRW4X_128R", "mq_gemm_frw4x_128r_asm.inc"),
}

# Measured-negative variants (NOTES.md 9.6 / DESIGN.md 7): their .inc files are NOT committed (9 000 lines of dead
# ISA in csrc/); `python -m mobilequant_amd.build --experiments` regenerates them into csrc/ before it compiles with -DMQ_BUILD_EXPERIMENTS
EXPERIMENTAL = ("fr128rs", "frw4", "frw4_128")


def configure(name):
    """binds the module-level tile constants of one variant (the emitters read them at call time)"""
    global BN, NW, EPI, PREFIX, FILE, BM, W_BYTES, PAR, STG, ROWP, STG_WAVE, LDS_BYTES, FN, PIECES, HALF, RING_BASE, SCALAR_GRID
    BN, NW, EPI, PREFIX, FILE = VARIANTS[name]
    SCALAR_GRID = (BN == 176 or EPI not in ("u8", "u8w4", "u8w4x"))      # ONE output grid per launch (scalars); otherwise per-column grids (q | k | v segments)
    # round 4: the u8 epilogue is interleaved with the MFMAs of the LAST TWO stages (final_block) unless MQ_FR_TAIL=0
    global TAIL
    TAIL = EPI == "u8" and os.environ.get("MQ_FR_TAIL", "1") != "0"
    BM = 32 * NW
    FN = BN // 16
    global SPLITK
    SPLITK = EPI == "f32rs"
    if SPLITK:
        EPI = "f32r"
    global W4, W4X
    W4X = EPI in ("u8w4x", "f32rw4x", "gatew4x")
    if EPI == "gatew4x":
        EPI = "gate"
        SCALAR_GRID = True
        TAIL = False
    elif EPI == "f32rw4x":
        EPI = "f32r"
        SCALAR_GRID = True
        TAIL = False
    elif W4X:
        EPI = "u8"
        SCALAR_GRID = BN == 176
        TAIL = True
    # round 5 experiment (MQ_FR_W4X_LDS=1; measured NEGATIVE, not built by default): frw4x / frw4x_128 (u8 epilogue inside the final block:
    # the output staging area is unused) park the PACKED pieces in the LDS by LDS-DMA three stages ahead -- the int8 ring's lead -- and
    # read them back when they are expanded, instead of holding them in registers for one stage.  Bit-identical outputs; TinyLlama w1
    # 22.7 us against 21.3 (registers) / 21.0 (int8 image), Gemma w1 68.4 against 62.5 / 59.5: one more LDS-DMA issue (60-185 cycles
    # beside MFMAs), an LDS write and an LDS read per piece cost more than the deeper lead recovers (profiles/r05/bench_w4_lds.log)
    global W4XL
    W4XL = W4X and EPI == "u8" and os.environ.get("MQ_FR_W4X_LDS", "0") == "1"
    W4 = EPI == "u8w4"
    if W4:
        EPI = "u8"
        W_BYTES = BN * BK // 2        # packed: 64 bytes per row and stage
        RING_BASE = 0
        PAR = RING * W_BYTES
        STG = PAR + 16 * BN
        ROWP, STG_WAVE = BN, 0
        LDS_BYTES = STG
        PIECES = (FN + NW - 1) // NW  # W LDS-DMA pieces are 16 rows x 64 B
        SCALAR_GRID = BN == 176
        TAIL = True
        return
    W_BYTES = BN * BK                 # 22528 / 16384
    RING_BASE = 65536 if EPI == "gate" else 0     # gate: the table sits at LDS offset 0 (ds_read_u8 addresses = ia * 256 + ib)
    PAR = RING_BASE + RING * W_BYTES  # alpha' | bias' | -w_zp | col_term
    STG = PAR + 16 * BN
    if EPI == "gate":
        # staging tiles alias the W ring (one barrier after the loop): table 64 KiB + ring 88 KiB + vectors = 158 464 B of the 160 KiB
        ROWP = BN if BN % 128 else BN + 16
        STG_WAVE = 2 * 16 * ROWP
        STG = RING_BASE
        LDS_BYTES = PAR + 16 * BN
        PIECES = (FN + NW - 1) // NW if W4X else (BN // 8 + NW - 1) // NW
        assert LDS_BYTES <= 160 * 1024
        return
    if EPI == "u8":
        # staging pitch (u8 rows): 176 B as is (writes <= 2-way conflicted); 128 B rows padded to 144 (bank = 36 frow + kq: conflict-free)
        ROWP = BN if BN % 128 else BN + 16          # (160: bank = 40 frow + kq, 2-way)
        STG_WAVE = 2 * 16 * ROWP
    else:
        # fp32 staging of HALF a 16-row block's columns at a time (64 columns = 256-byte row pieces), pitch 272
        HALF = 64
        ROWP = HALF * 4 + 16
        STG_WAVE = 16 * ROWP
    LDS_BYTES = STG + NW * STG_WAVE
    PIECES = (BN // 8 + NW - 1) // NW   # most W LDS-DMA pieces (8 rows x 128 B) a wave owns per stage
    if W4X:
        PIECES = (FN + NW - 1) // NW    # packed pieces are 16 rows x 64 B
    assert LDS_BYTES <= 160 * 1024


# ---- registers -------------------------------------------------------------------------------------------------------------------
V_T = 98                          # first temporary VGPR
V_RD, V_WOFF0, V_WOFF1 = 98, 99, 100        # LDS read address, per-lane W read offsets (k-step 0 / 1)
V_PAR = 101                       # epilogue: LDS address of this lane's alpha' chunk (PAR + kq*16)
V_STW = 102                       # epilogue: staging write address
V_RS0, V_RS1 = 103, 104           # row sums of the lane's two rows (prologue)
V_P0 = 106                        # 106..109: parameter loads pa, pb, pz, pc (prologue); 106..113: alpha' / bias' set 1 (epilogue)
V_LDSO = 106                      # epilogue (after the conversion): 106..108 staging read offsets, 109..111 global store offsets
V_GOFS = 109
V_E = 114                         # 114..121 (register tuples must be even-aligned): -w_zp / col_term chunk (prologue); alpha' / bias' set 0 (epilogue)
V_TMP = 127
S0 = 58                           # first temporary SGPR
S_ABASE, S_WBASE = 58, 60         # pairs: activation pointer of stage t+1; weight pointer of the stage whose DMA is issued next
S_CUR, S_NXT, S_DMA = 62, 63, 64  # ring slot byte offsets: stage t, t+1, t+3
S_WK = (65, 66, 67, 71, 80)       # wave*1024 + i*NW*1024: LDS offset of this wave's piece i inside a slot
S_CNT = 68                        # steady-state pairs left
S_TMP, S_TMP2 = 69, 70
S_EXEC = 72                       # pair
S_TS = 74                         # 74..81: four s_memtime stamps (stamp builds)
S_MR = 82
S_RT = 84                         # 84..87: s_memrealtime (constant 100 MHz) at kernel start / end (stamp builds)
S_FM = 88                         # final block: 88..95 exec masks of the group stores (row block i, last group?)
S_OB1 = 82                        # final block: output pointer of row block 1 (pair; = S_MR of the classic epilogue)
V_LDSG, V_GOG, V_RD3 = 103, 104, 105   # final block: staging read-back address / global offset of a lane's 16-byte piece; 4th W read address
S_SO, S_OO, S_ISO = 96, 97, 98    # the output grid: scale, offset (s_load at the top of the program), 1 / scale (IEEE divide, prologue step 3)

# what-if switches for profiling builds (results are wrong): MQ_FR_NO_A / _NO_W / _NO_READ / _NO_MFMA drop the in-loop activation
# loads / W LDS-DMA / W fragment reads / MFMAs
NO_A, NO_W, NO_READ, NO_MFMA = (bool(os.environ.get("MQ_FR_" + k)) for k in ("NO_A", "NO_W", "NO_READ", "NO_MFMA"))
# cache policy of the output stores: "" (write-back, default) | nt | sc1 | "sc0 sc1" | none.  Same kernel time for all (profiles/r02),
# but nt pushes the partial 32 B sectors of the 176-byte tile rows out before their neighbours merge: WRITE_SIZE 14.5 MB vs the exact
# M * N = 11.5 MB with write-back
STORE_POLICY = os.environ.get("MQ_FR_STORE", "")

out = []


def emit(s):
    out.append(s)


_uid = [0]


def label(prefix):
    _uid[0] += 1
    return f".Lfr_{prefix}_{_uid[0]}%="


def acc(i, j):
    b = (2 * j + i) * 4
    return f"v[{b}:{b + 3}]"


def accr(i, j, e):
    return f"v{(2 * j + i) * 4 + e}"


def wreg(set_, j):
    b = 4 * FN * set_ + 4 * j
    return f"a[{b}:{b + 3}]"


def areg(set_, ks, i):
    b = 8 * FN + 16 * set_ + 8 * ks + 4 * i
    return f"a[{b}:{b + 3}]"


def areg_spare(i):
    """A(KT-1).ks1 of the final block: the eight AGPRs behind the two stage sets (requested while both sets are still in use)"""
    b = 8 * FN + 32 + 4 * i
    return f"a[{b}:{b + 3}]"


class Queue:
    """The wave's VMEM queue (vmcnt retires in issue order): issue(tag) appends, wait_for(tags) emits the s_waitcnt that leaves
    only the operations issued after the youngest of `tags` in flight."""

    def __init__(self):
        self.q = []
        self.log = []

    def issue(self, tag):
        self.q.append(tag)

    def wait_for(self, *tags):
        idx = max((i for i, t in enumerate(self.q) if t in tags), default=-1)
        if idx < 0:
            return                                   # already retired by an earlier wait
        n = len(self.q) - 1 - idx
        assert n <= 63
        emit(f"s_waitcnt vmcnt({n})")
        self.log.append(n)
        self.q = self.q[idx + 1:]


def issue_w(q, t, nw, slot_sgpr):
    """this wave's LDS-DMA pieces of W(t) -> ring slot `slot_sgpr`; source k offset = S_WBASE (advanced by 128 afterwards)"""
    for i in range(nw):
        emit(f"s_add_u32 m0, s{slot_sgpr}, s{S_WK[i]}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 %[sw{i}], s[{S_WBASE}:{S_WBASE + 1}]")
        q.issue(("W", t))
    emit(f"s_add_u32 s{S_WBASE}, s{S_WBASE}, {BK}")
    emit(f"s_addc_u32 s{S_WBASE + 1}, s{S_WBASE + 1}, 0")


def w_piece(q, t, i, slot_sgpr):
    def f():
        emit(f"s_add_u32 m0, s{slot_sgpr}, s{S_WK[i]}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 %[sw{i}], s[{S_WBASE}:{S_WBASE + 1}]")
        q.issue(("W", t))
    return f


def a_load(q, t, ks, i, set_, off):
    """A(t) fragment (ks, i) -> register set; address = S_ABASE + av{i} + off"""
    def f():
        emit(f"global_load_dwordx4 {areg(set_, ks, i)}, %[av{i}], s[{S_ABASE}:{S_ABASE + 1}]" + (f" offset:{off}" if off else ""))
        q.issue(("A", t, ks))
    return f


class LQueue:
    """the wave's LDS queue (lgkmcnt retires LDS operations in issue order)"""

    def __init__(self):
        self.q = []

    def issue(self, tag):
        self.q.append(tag)

    def wait_for(self, *tags):
        idx = max((i for i, t in enumerate(self.q) if t in tags), default=-1)
        if idx < 0:
            return
        n = len(self.q) - 1 - idx
        if n > 15:                                 # (the counter has four bits: wait for the oldest of the younger operations too)
            idx += n - 15
            n = 15
        emit(f"s_waitcnt lgkmcnt({n})")
        self.q = self.q[idx + 1:]


def kstep(cur, aset, ks, rd_slot, rd_woff, nxt, vmem, read=True, c_zero=False, extra=None, lq=None):
    """22 MFMAs on W register set `cur` x A(aset, ks); 11 ds_reads of slot `rd_slot` (+ per-lane offset register rd_woff) into W
    register set `nxt`, one after each of the first MFMAs; `vmem`: callables (VMEM issues) spread behind the later MFMAs.
    c_zero: the accumulators START here (src2 = 0); extra: {position: [callables]} more fillers (they share the LDS queue `lq`)."""
    lq = lq or LQueue()
    if read:
        emit(f"v_add_u32 v{V_RD}, s{rd_slot}, v{rd_woff}")
    places = {}
    for n, (kind, f) in enumerate(vmem):      # A loads early (their registers are free), DMA pieces in the second half
        na = sum(1 for k, _ in vmem[:n] if k == kind)
        nwp = sum(1 for k, _ in vmem if k == "w")
        wstep = 3 if FN == 11 else max(1, (FN - 1) // max(nwp, 1))
        places.setdefault(1 + 2 * na if kind == "a" else FN + 1 + wstep * na, []).append(f)
    m = 0
    for j in range(FN):
        for i in range(2):
            if not NO_MFMA:
                emit(f"v_mfma_i32_16x16x64_i8 {acc(i, j)}, {wreg(cur, j)}, {areg(aset, ks, i)}, " + ("0" if c_zero else acc(i, j)))
            if read and m < FN and not NO_READ:
                emit(f"ds_read_b128 {wreg(nxt, m)}, v{V_RD} offset:{m * 16 * BK}")
                lq.issue(("F", m))
            for f in places.get(m, []):
                f()
            for f in (extra or {}).get(m, []):
                f()
            m += 1
    if lq.q:
        emit("s_waitcnt lgkmcnt(0)")
        lq.q = []


def deferred_init(lq, ks, phase):
    """round 4: the zero-point correction acc += -w_zp[n] * row_sum[m] (phase "wz": stage 1) + col_term[n] (phase "ct": stage KT-4) as
    fillers between the MFMAs -- the accumulators start at 0 in stage 0 (MFMA src2 = 0) instead of being initialised by 88 mads per wave
    on the prologue's critical path.  Two VALU per gap (a wave hides ~5 issue slots per MFMA beside its partner; four per gap measured
    +950 cycles): gap m of k-step ks touches elements 2 ks, 2 ks + 1 of the tile 11 MFMAs away -- its last MFMA is long complete, its
    next far off.  Returns (pre, extra): callables in front of the k-step, {position: [callables]}."""
    # a k-step touches elements 2 ks, 2 ks + 1 of a chunk only: 8-byte reads into two-register sets
    SETS = [106, 108, 124, 126] if W4X else [V_E, V_E + 2, V_E + 4, V_E + 6, V_P0, V_P0 + 2, V_P0 + 4, V_P0 + 6]
    LEAD = int(os.environ.get("MQ_FR_DLEAD", "4"))
    NT = 2 * FN
    order = [(m + FN) % NT for m in range(NT)]
    cols = []
    for m, t in enumerate(order):
        if not cols or cols[-1][0] != t // 2:
            cols.append([t // 2, m, len(cols)])
    base = (2 if phase == "wz" else 3) * 4 * BN
    pre, extra = [], {}

    def rd(c, k):
        def f():
            t = SETS[k % len(SETS)]
            emit(f"ds_read_b64 v[{t}:{t + 1}], v{V_PAR} offset:{base + c * 64 + 8 * ks}")
            lq.issue(("C", ks, k))
        return f
    for c, first, k in cols:
        at = first - LEAD
        (pre if at < 0 else extra.setdefault(at, [])).append(rd(c, k))
    kof = {}
    for c, first, k in cols:
        for m in range(first, NT):
            if order[m] // 2 != c:
                break
            kof[m] = k
    for m, t in enumerate(order):
        i, j, k = t & 1, t // 2, kof[m]
        st = SETS[k % len(SETS)]
        fl = extra.setdefault(m, [])
        fl.append(lambda k=k: lq.wait_for(("C", ks, k)))
        for d, e in enumerate((2 * ks, 2 * ks + 1)):
            if phase == "wz":
                fl.append(lambda i=i, j=j, e=e, d=d, st=st: emit(f"v_mad_i32_i24 {accr(i, j, e)}, v{st + d}, v{V_RS0 + i}, {accr(i, j, e)}"))
            else:
                fl.append(lambda i=i, j=j, e=e, d=d, st=st: emit(f"v_add_u32 {accr(i, j, e)}, {accr(i, j, e)}, v{st + d}"))
    return pre, extra


def rotate():
    emit(f"s_mov_b32 s{S_CUR}, s{S_NXT}")
    for s in (S_NXT, S_DMA):
        emit(f"s_add_u32 s{s}, s{s}, {W_BYTES}")
        emit(f"s_cmp_eq_u32 s{s}, {RING * W_BYTES}")
        emit(f"s_cselect_b32 s{s}, 0, s{s}")
    emit(f"s_add_u32 s{S_ABASE}, s{S_ABASE}, {2 * 1024}")
    emit(f"s_addc_u32 s{S_ABASE + 1}, s{S_ABASE + 1}, 0")


# ---- round 4: packed 4-bit weights expanded once per workgroup (variants frw4x / frw4x_128) ------------------------------------------------
# A wave owns the packed pieces wave + NW i (16 rows x 64 B of a K = 128 stage): lane l loads 16 bytes = chunk c = l & 3 (32 consecutive k:
# element p low, p + 16 high nibble of byte p) of row r = l >> 2 with a plain global_load_dwordx4, and one stage later writes the int8
# row pieces k = 32 c .. + 15 (low nibbles) and + 16 .. + 31 (high) into the W ring at the int8 kernels' XOR-swizzled positions (chunks
# 2 c, 2 c + 1 of the 128-byte row).  Loads of W(t + 2) are issued in k-step 1 of stage t, expanded in k-step 0 of stage t + 1 (before
# the mid-stage barrier that publishes W(t + 2) to the other waves' fragment reads); everything else is the int8 program.
X_P = (110, 114)                   # packed quads of the wave's (up to two) pieces
X_P2 = (106, 124)                  # a second set (pre-final stage only: W(KT-1) next to W(KT-2))
X_L = 118                          # low nibbles of the piece being expanded
X_A0, X_A1, X_T = 122, 123, 105    # per-lane ring offsets of the low / high 16 bytes, address temporary
S_M4 = 99                          # 0x0f0f0f0f (shared with program_w4)


def w4x_load(q, t, i, regs):
    def f():
        emit(f"global_load_dwordx4 v[{regs[i]}:{regs[i] + 3}], %[sw{i}], s[{S_WBASE}:{S_WBASE + 1}]")
        q.issue(("W", t))
    return f


def w4x_advance():
    emit(f"s_add_u32 s{S_WBASE}, s{S_WBASE}, {BK // 2}")
    emit(f"s_addc_u32 s{S_WBASE + 1}, s{S_WBASE + 1}, 0")


# W4XL: the packed image of a stage (BN rows x 64 B) goes through a four-slot ring in the (otherwise unused) output staging area:
# piece wave + NW i of W(t) is written by THIS wave's LDS-DMA (lane-linear: lane l = row l >> 2, chunk l & 3 -- the register path's
# layout) in stage t - 3 and read back by the same wave (ds_read_b128 of its own 16 bytes: no barrier) when it is expanded in stage t - 1.
S_PKR, S_PKD, S_PK2, S_PKT = 76, 77, 78, 79      # packed-ring slot offsets: expansion source of this stage, DMA target, pre-final's second source; scratch


def pk_bytes():
    return BN * BK // 2


def w4xl_dma(q, t, i, slot_sgpr):
    def f():
        emit(f"s_add_u32 m0, s{slot_sgpr}, s{S_WK[i]}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 %[sw{i}], s[{S_WBASE}:{S_WBASE + 1}]")
        q.issue(("W", t))
    return f


def w4x_readback(q, lq, t, nw, regs, pk):
    """callables in FRONT of a k-step: this wave's packed pieces of W(t) (its own LDS-DMA of three stages ago) -> `regs`"""
    def rd():
        q.wait_for(("W", t))
        emit(f"s_add_u32 s{S_PKT}, s{pk}, s{S_WK[0]}")
        emit(f"v_and_b32 v{X_T}, 63, %[tid]")
        emit(f"v_lshl_add_u32 v{X_T}, v{X_T}, 4, s{S_PKT}")
        for i in range(nw):
            emit(f"ds_read_b128 v[{regs[i]}:{regs[i] + 3}], v{X_T} offset:{i * NW * 1024}")
            lq.issue(("PK", t, i))
    return [rd]


def w4xl_rotate():
    for s_ in (S_PKR, S_PKD):
        emit(f"s_add_u32 s{s_}, s{s_}, {pk_bytes()}")
        emit(f"s_cmp_eq_u32 s{s_}, {STG + RING * pk_bytes()}")
        emit(f"s_cselect_b32 s{s_}, {STG}, s{s_}")


def w4x_expand(q, lq, t, nw, slot_sgpr, regs, pk=None):
    """callables (small groups of instructions) that turn the packed pieces of W(t) in `regs` into int8 rows of ring slot `slot_sgpr`.
    pk: SGPR with the packed-ring slot of W(t) (W4XL): the pieces are first read back from the LDS into `regs`."""
    out_ = [lambda: q.wait_for(("W", t))]
    if pk is not None:
        # the read-back goes FIRST in its k-step (w4x_readback, in front of the W fragment reads): LDS operations return in order, so a
        # counted wait for a packed piece issued BEHIND the eleven fragment reads would drain them mid-k-step (measured: Gemma w1 62 -> 74 us)
        out_ = []
    for i in range(nw):
        x = regs[i]
        off = i * NW * 16 * BK

        def lo(x=x, off=off, i=i):
            if pk is not None:
                lq.wait_for(("PK", t, i))
            for e in range(4):
                emit(f"v_and_b32 v{X_L + e}, s{S_M4}, v{x + e}")
            emit(f"v_add_u32 v{X_T}, s{slot_sgpr}, v{X_A0}")
            emit(f"ds_write_b128 v{X_T}, v[{X_L}:{X_L + 3}] offset:{off}")
            lq.issue(("X", t))

        def hi(x=x, off=off):
            for e in range(4):
                emit(f"v_lshrrev_b32 v{x + e}, 4, v{x + e}")
            for e in range(4):
                emit(f"v_and_b32 v{x + e}, s{S_M4}, v{x + e}")
            emit(f"v_add_u32 v{X_T}, s{slot_sgpr}, v{X_A1}")
            emit(f"ds_write_b128 v{X_T}, v[{x}:{x + 3}] offset:{off}")
            lq.issue(("X", t))
        out_ += [lo, hi]
    return out_


def stage(q, t, nw, kt=None, sym=None, first=False, dinit=False):
    """One K = 128 stage.  t: stage number used for the queue tags; kt: total stages when the tail conditions apply (None = steady
    state: everything is issued).  first: the accumulators start in this stage's k-step 0; dinit: "wz" / "ct" = that half of the deferred
    zero-point correction rides in this stage (deferred_init)."""
    set_ = t & 1
    more1 = kt is None or t + 1 < kt
    more2 = kt is None or t + 2 < kt
    more3 = kt is None or t + 3 < kt
    emit(f"; ---- stage {sym or t}: A set {set_}")
    q.wait_for(("A", t, 0))
    v0 = []
    if more1 and not NO_A:      # S_ABASE = activation pointer of stage t+1
        v0 += [("a", a_load(q, t + 1, 1, 0, 1 - set_, 1024)), ("a", a_load(q, t + 1, 1, 1, 1 - set_, 1024))]
    if more3 and not NO_W and not W4X:
        v0 += [("w", w_piece(q, t + 3, i, S_DMA)) for i in range(nw)]
    lq0, lq1 = LQueue(), LQueue()
    pre0, ex0 = deferred_init(lq0, 0, dinit) if dinit else ([], None)
    pre1, ex1 = deferred_init(lq1, 1, dinit) if dinit else ([], None)
    if W4XL and more3 and not NO_W:       # packed pieces of W(t + 3) -> packed ring (LDS-DMA, as the int8 kernel's W pieces)
        v0 += [("w", w4xl_dma(q, t + 3, i, S_PKD)) for i in range(nw)]
    if W4X and more1:            # W(t + 1): packed pieces (loaded a stage ago) -> int8 rows of slot(t + 1), behind the later MFMAs of k-step 0
        ex0 = dict(ex0 or {})
        if W4XL:
            pre0 = list(pre0) + w4x_readback(q, lq0, t + 1, nw, X_P, S_PKR)
        for n, f in enumerate(w4x_expand(q, lq0, t + 1, nw, S_NXT, X_P, pk=S_PKR if W4XL else None)):
            ex0.setdefault(FN + 1 + n, []).append(f)
    for f in pre0:
        f()
    kstep(0, set_, 0, S_CUR, V_WOFF1, 1, v0, c_zero=first, extra=ex0, lq=lq0)
    if more3 and not W4X:
        emit(f"s_add_u32 s{S_WBASE}, s{S_WBASE}, {BK}")
        emit(f"s_addc_u32 s{S_WBASE + 1}, s{S_WBASE + 1}, 0")
    if more3 and W4XL:
        w4x_advance()
    q.wait_for(("A", t, 1), ("W", t + 1))
    emit("s_barrier")
    v1 = []
    if more2 and not NO_A:
        v1 += [("a", a_load(q, t + 2, 0, 0, set_, 2048)), ("a", a_load(q, t + 2, 0, 1, set_, 2048))]
    if W4X and more2 and not W4XL:            # W(t + 2): this wave's packed pieces -> registers (free since k-step 0's expansion)
        v1 += [("w", w4x_load(q, t + 2, i, X_P)) for i in range(nw)]
    for f in pre1:
        f()
    kstep(1, set_, 1, S_NXT, V_WOFF0, 0, v1, read=more1, extra=ex1, lq=lq1)
    if W4X and more2 and not W4XL:
        w4x_advance()
    rotate()
    if W4XL:
        w4xl_rotate()


# ---- round 4: the u8 epilogue inside the last two stages ---------------------------------------------------------------------------------
# Classic tail: ... stage KT-2, stage KT-1 (k outer: every accumulator is final only at the very end), then ~330 VALU / LDS instructions
# of epilogue per wave with the matrix pipe idle, then the stores, then the L2 write-back at the kernel boundary.
# final_block: the 8 * FN MFMAs of stages KT-2 and KT-1 run COLUMN outer -- for j: 4 k-steps x 2 row blocks on acc(., j) -- so column
# chunk j is final after its eighth MFMA and its conversion (cvt, fma, cvt_pk_u8, staging write) issues between the MFMAs of column j + 1;
# a group of four chunks (64 bytes of 16 rows) is read back as 16-byte pieces and stored while later columns still multiply.  Needs both
# stages' operands resident: W(KT-2), W(KT-1) are in the ring anyway, A(KT-2) and A(KT-1).ks0 in the two register sets, A(KT-1).ks1
# in the eight spare AGPRs behind them (requested during stage KT-3).  W fragments come from the LDS per column (4 ds_read_b128 per
# 8 MFMAs: the classic loop's ratio) through a ring of 16-register slots in a[0 : 8 FN), two columns ahead.
RDA = (V_WOFF0, V_WOFF1, V_RD, V_RD3)       # LDS read addresses of (stage KT-2, ks0 / ks1), (stage KT-1, ks0 / ks1)


def fb_slot(j, f):
    b = 16 * (j % ((8 * FN) // 16)) + 4 * f
    return f"a[{b}:{b + 3}]"


def fb_read(lq, j):
    def mk(f):
        def fn():
            emit(f"ds_read_b128 {fb_slot(j, f)}, v{RDA[f]} offset:{j * 16 * BK}")
            lq.issue(("R", j))
        return fn
    return [mk(f) for f in range(4)]


def a_load_spare(q, t, i, off):
    def f():
        emit(f"global_load_dwordx4 {areg_spare(i)}, %[av{i}], s[{S_ABASE}:{S_ABASE + 1}] offset:{off}")
        q.issue(("A", t, 1))
    return f


def stage_pre_final(q, lq, t, nw):
    """stage KT-3: classic k-step 0; k-step 1 without the classic read-ahead but with the final block's first two columns of W fragments"""
    set_ = t & 1
    emit(f"; ---- stage KT-3 (last classic stage): A set {set_}; A(KT-1).ks1 -> spare AGPRs; columns 0, 1 of the final block are read ahead")
    q.wait_for(("A", t, 0))
    v0 = []
    if not NO_A:
        v0 += [("a", a_load(q, t + 1, 1, 0, 1 - set_, 1024)), ("a", a_load(q, t + 1, 1, 1, 1 - set_, 1024)),
               ("a", a_load_spare(q, t + 2, 0, 3072)), ("a", a_load_spare(q, t + 2, 1, 3072))]
    ex0 = None
    if W4X:
        # W(KT-2) (packed pieces loaded in stage KT-4) and W(KT-1) (requested at the head of this k-step into a second register set)
        # both become int8 rows before the barrier: the final block reads both slots
        emit(f"s_add_u32 s{S_TMP2}, s{S_NXT}, {W_BYTES}")
        emit(f"s_cmp_eq_u32 s{S_TMP2}, {RING * W_BYTES}")
        emit(f"s_cselect_b32 s{S_TMP2}, 0, s{S_TMP2}")
        ex0 = {}
        if W4XL:                 # both stages sit in the packed ring already (DMA of stages KT-6 / KT-5): read back and expand, one after the other
            emit(f"s_add_u32 s{S_PK2}, s{S_PKR}, {pk_bytes()}")
            emit(f"s_cmp_eq_u32 s{S_PK2}, {STG + RING * pk_bytes()}")
            emit(f"s_cselect_b32 s{S_PK2}, {STG}, s{S_PK2}")
            for f in w4x_readback(q, lq, t + 1, nw, X_P, S_PKR) + w4x_readback(q, lq, t + 2, nw, X_P2, S_PK2):
                f()
            for n, f in enumerate(w4x_expand(q, lq, t + 1, nw, S_NXT, X_P, pk=S_PKR)):
                ex0.setdefault(4 + n, []).append(f)
            for n, f in enumerate(w4x_expand(q, lq, t + 2, nw, S_TMP2, X_P2, pk=S_PK2)):
                ex0.setdefault(2 * FN - 5 + n, []).append(f)
        else:
            for i in range(nw):
                v0.insert(i, ("a", w4x_load(q, t + 2, i, X_P2)))
            for n, f in enumerate(w4x_expand(q, lq, t + 1, nw, S_NXT, X_P)):
                ex0.setdefault(4 + n, []).append(f)
            for n, f in enumerate(w4x_expand(q, lq, t + 2, nw, S_TMP2, X_P2)):
                ex0.setdefault(2 * FN - 5 + n, []).append(f)
    kstep(0, set_, 0, S_CUR, V_WOFF1, 1, v0, extra=ex0, lq=lq if W4X else None)
    q.wait_for(("A", t, 1), ("W", t + 1), ("W", t + 2))
    emit("s_barrier")                       # W(KT-2) and W(KT-1) of every wave have landed
    rotate()                                # S_CUR / S_NXT = ring slots of stages KT-2 / KT-1, S_ABASE -> stage KT-2
    emit(f"v_add_u32 v{V_RD}, s{S_NXT}, v{V_WOFF0}")
    emit(f"v_add_u32 v{V_RD3}, s{S_NXT}, v{V_WOFF1}")
    emit(f"v_add_u32 v{V_WOFF0}, s{S_CUR}, v{V_WOFF0}")
    emit(f"v_add_u32 v{V_WOFF1}, s{S_CUR}, v{V_WOFF1}")
    fill = []
    if not NO_A:                            # A(KT-1).ks0 -> this stage's own set (its ks0 registers are free now); S_ABASE = stage KT-1 already
        fill += [a_load(q, t + 2, 0, 0, set_, 0), a_load(q, t + 2, 0, 1, set_, 0)]
    fill += fb_read(lq, 0) + fb_read(lq, 1)
    m = 0
    for j in range(FN):
        for i in range(2):
            if not NO_MFMA:
                emit(f"v_mfma_i32_16x16x64_i8 {acc(i, j)}, {wreg(1, j)}, {areg(set_, 1, i)}, {acc(i, j)}")
            if m >= 1 and fill:
                fill.pop(0)()
            m += 1
    assert not fill


PROBE = False      # set per variant in main(): the headline kernel ("fr") always stamps its start / end clocks and stores them when asked
DINIT = os.environ.get("MQ_FR_DINIT", "1") != "0"       # deferred accumulator initialisation (round 4); 0 = in the prologue
PRO_SPLIT = bool(os.environ.get("MQ_FR_PRO_SPLIT"))
PSTAMP = bool(os.environ.get("MQ_FR_PSTAMP"))      # stamp build: five more s_memtime stamps inside the prologue (dbg slots 8..12)
PST = (88, 90, 92, 94, 100)


def pstamp(k, stamp):
    """diagnostic: stamp k -> lanes 2k, 2k+1 of v105 (unused until stage KT-3); after the fifth, lanes 0..9 store to dbg slots 8..12 and
    the wave drains its VMEM queue (the store is invisible to the queue simulation): the timeline behind the prologue is perturbed"""
    if PSTAMP and stamp:
        emit(f"s_memtime s[{PST[0]}:{PST[0] + 1}]")
        emit("s_waitcnt lgkmcnt(0)")
        emit(f"v_writelane_b32 v{V_RD3}, s{PST[0]}, {2 * k}")
        emit(f"v_writelane_b32 v{V_RD3}, s{PST[0] + 1}, {2 * k + 1}")
        if k == 4:
            emit(f"v_and_b32 v{V_TMP}, 63, %[tid]")
            emit(f"v_cmp_gt_u32 vcc, 10, v{V_TMP}")
            emit("s_and_b64 exec, exec, vcc")
            emit(f"v_lshlrev_b32 v{V_TMP}, 2, v{V_TMP}")
            emit(f"global_store_dword v{V_TMP}, v{V_RD3}, %[dbg] offset:64")
            emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
            emit("s_waitcnt vmcnt(0)")
FB_NOCVT, FB_NOP, FB_NOST, FB_NOSWAP = (bool(os.environ.get("MQ_FR_FB_" + k)) for k in ("NOCVT", "NOP", "NOST", "NOSWAP"))   # what-if builds


def final_block(q, lq, stamp):
    NG = (FN + 3) // 4                                  # store groups of (up to) four 16-byte chunks
    CAP = int(os.environ.get("MQ_FR_CAP", "5"))         # fillers behind one MFMA
    EA = [V_E, V_P0]
    emit("; ==== final block: stages KT-2 and KT-1 column by column, the epilogue between the MFMAs")
    if stamp:
        emit(f"s_memtime s[{S_TS + 4}:{S_TS + 5}]")
        emit("s_waitcnt lgkmcnt(0)")            # (scalar memory returns out of order: nothing of it may be in flight under counted waits)
        lq.q = []
    q.wait_for(("A", 6, 0), ("A", 6, 1), ("A", 7, 0), ("A", 7, 1))
    assert q.q == [], q.q
    B = [lambda i: areg(0, 0, i), lambda i: areg(0, 1, i), lambda i: areg(1, 0, i), areg_spare]
    fill = []

    def F(minidx, fn):
        fill.append((minidx, fn))

    def E(minidx, text):
        F(minidx, lambda: emit(text))

    # -- set-up of the group stores (first needed behind column 3): lane (row = lane & 15, q = lane >> 4) stores chunk 4 G + q of its row
    rem = FN - 4 * (NG - 1)
    for text in (
            f"v_and_b32 v{V_TMP}, 63, %[tid]",
            f"v_and_b32 v{V_GOG}, 15, v{V_TMP}",                                        # row
            f"v_lshrrev_b32 v{V_TMP}, 4, v{V_TMP}",                                     # q
            f"v_cmp_gt_i32_e64 s[{S_FM}:{S_FM + 1}], %[mrem], v{V_GOG}",                # row block 0: row < rows left
            f"v_add_u32 v{V_LDSG}, 16, v{V_GOG}",
            f"v_cmp_gt_i32_e64 s[{S_FM + 2}:{S_FM + 3}], %[mrem], v{V_LDSG}",           # row block 1
            f"v_cmp_gt_u32_e64 s[{S_FM + 4}:{S_FM + 5}], {rem}, v{V_TMP}",              # last group: only `rem` chunks exist
            f"s_and_b64 s[{S_FM + 6}:{S_FM + 7}], s[{S_FM + 2}:{S_FM + 3}], s[{S_FM + 4}:{S_FM + 5}]",
            f"s_and_b64 s[{S_FM + 4}:{S_FM + 5}], s[{S_FM}:{S_FM + 1}], s[{S_FM + 4}:{S_FM + 5}]",
            f"v_mul_lo_u32 v{V_GOG}, v{V_GOG}, %[ldn]",
            f"v_lshl_add_u32 v{V_GOG}, v{V_TMP}, 4, v{V_GOG}",                          # row * ldn + q * 16
            f"s_lshl_b32 s{S_TMP2}, %[ldn], 4",
            f"s_mov_b64 s[{S_OB1}:{S_OB1 + 1}], %[outw]",
            f"s_add_u32 s{S_OB1}, s{S_OB1}, s{S_TMP2}",
            f"s_addc_u32 s{S_OB1 + 1}, s{S_OB1 + 1}, 0"):
        E(0, text)

    def params(j):
        st = j & 1
        def fn0():
            emit(f"ds_read_b128 v[{EA[st]}:{EA[st] + 3}], v{V_PAR} offset:{j * 64}")
            lq.issue(("P", j))
        def fn1():
            emit(f"ds_read_b128 v[{EA[st] + 4}:{EA[st] + 7}], v{V_PAR} offset:{4 * BN + j * 64}")
            lq.issue(("P", j))
        return [fn0, fn1]

    def tbase(g, i):
        """the four registers that collect group g's packed dwords of row block i: the accumulator tile of the group's FIRST column
        (dead once that column is converted; a store's data registers must be consecutive)"""
        return (2 * (4 * g) + i) * 4

    def convert(minidx, j):
        """column chunk j: index = cvt_pk_u8(fma(float(acc), alpha', bias')), four bytes per lane -> register j % 4 of the group's tuple"""
        st = j & 1
        g, k = j // 4, j % 4
        if not FB_NOP:
            F(minidx, lambda: lq.wait_for(("P", j)))
        if not FB_NOCVT:
            for i in range(2):
                for e in range(4):
                    E(minidx, f"v_cvt_f32_i32 {accr(i, j, e)}, {accr(i, j, e)}")
            for i in range(2):
                for e in range(4):
                    E(minidx, f"v_fma_f32 {accr(i, j, e)}, {accr(i, j, e)}, v{EA[st] + e}, v{EA[st] + 4 + e}")
            for i in range(2):
                d = tbase(g, i) + k
                for e in range(4):
                    E(minidx, f"v_cvt_pk_u8_f32 v{d}, {accr(i, j, e)}, {e}, " + (f"v{d}" if e else "0"))
        if (k == 3 or j == FN - 1) and not FB_NOST:
            store_group(minidx, g)

    def store_group(minidx, g):
        """lane (row, q) holds dword q of chunks 4g .. 4g+3 of its row; a 4 x 4 transpose between the register index and the lane's
        16-lane row (two v_permlane32_swap + two v_permlane16_swap) gives it the 16 bytes of chunk 4g + q: one 16-byte store per lane,
        64 contiguous bytes per row -- no LDS staging"""
        last = g == NG - 1 and rem != 4
        for i in range(2):
            t = tbase(g, i)
            # (gfx950: two wait states between a VALU write of a register and a v_permlane*_swap that reads it)
            if not FB_NOSWAP:
                E(minidx, "s_nop 1")
                E(minidx, f"v_permlane32_swap_b32 v{t}, v{t + 2}")
                E(minidx, f"v_permlane32_swap_b32 v{t + 1}, v{t + 3}")
                E(minidx, "s_nop 1")
                E(minidx, f"v_permlane16_swap_b32 v{t}, v{t + 1}")
                E(minidx, f"v_permlane16_swap_b32 v{t + 2}, v{t + 3}")
                E(minidx, "s_nop 1")
                for k in range(4):
                    E(minidx, f"v_xor_b32 v{t + k}, %[xorv], v{t + k}")
            m = S_FM + 2 * i + (4 if last else 0)
            base = "%[outw]" if i == 0 else f"s[{S_OB1}:{S_OB1 + 1}]"
            def st(i=i, base=base, t=t, m=m):
                # ONE filler: no MFMA may issue under the store's exec mask (a ragged row block masks lanes off -- all of them when
                # the block lies past M -- and an MFMA issued under EXEC = 0 is dropped)
                emit(f"s_mov_b64 exec, s[{m}:{m + 1}]")
                if STORE_POLICY != "none":
                    pol = STORE_POLICY if g == NG - 1 else os.environ.get("MQ_FR_STORE_EARLY", STORE_POLICY)
                    emit(f"global_store_dwordx4 v{V_GOG}, v[{t}:{t + 3}], {base} offset:{g * 64} {pol}".rstrip())
                    q.issue(("S", g, i))
                emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
            F(minidx, st)

    for j in range(FN):
        base = 8 * j
        if j + 2 < FN:
            for fn in fb_read(lq, j + 2):
                F(base, fn)
        if not FB_NOP:
            for fn in params(j):
                F(base, fn)
        if j >= 1:
            convert(base + 2, j - 1)
    convert(8 * FN + 1000, FN - 1)                      # behind the last MFMA

    fi = 0
    n = 0
    for j in range(FN):
        lq.wait_for(("R", j))
        for f in range(4):
            for i in range(2):
                if not NO_MFMA:
                    emit(f"v_mfma_i32_16x16x64_i8 {acc(i, j)}, {fb_slot(j, f)}, {B[f](i)}, {acc(i, j)}")
                k = 0
                while fi < len(fill) and fill[fi][0] <= n and k < CAP:
                    fill[fi][1]()
                    fi += 1
                    k += 1
                n += 1
    # what is left of column FN - 2 (nothing when CAP is large enough), then the last column: its accumulators are being written by
    # the last MFMAs -- 19 wait states between an XDL write and a VALU read of the same register
    while fi < len(fill) and fill[fi][0] < 8 * FN:
        fill[fi][1]()
        fi += 1
    emit("s_nop 15")
    emit("s_nop 3")
    while fi < len(fill):
        fill[fi][1]()
        fi += 1
    if stamp:
        emit("s_waitcnt vmcnt(0)")
        emit(f"s_memtime s[{S_TS + 6}:{S_TS + 7}]")
        emit(f"s_memrealtime s[{S_RT + 2}:{S_RT + 3}]")
        emit("s_waitcnt lgkmcnt(0)")
        emit(f"v_and_b32 v{V_TMP}, 63, %[tid]")
        emit(f"v_cmp_eq_u32 vcc, 0, v{V_TMP}")
        emit("s_and_b64 exec, exec, vcc")
        emit(f"v_mov_b32 v{V_TMP}, 0")
        for k in range(6):
            src = S_TS + 2 * k if k < 4 else S_RT + 2 * (k - 4)
            emit(f"v_mov_b32 v0, s{src}")
            emit(f"v_mov_b32 v1, s{src + 1}")
            emit(f"global_store_dwordx2 v{V_TMP}, v[0:1], %[dbg] offset:{8 * k}")
        emit("v_mov_b32 v0, %[tentry_lo]")
        emit("v_mov_b32 v1, %[tentry_hi]")
        emit(f"global_store_dwordx2 v{V_TMP}, v[0:1], %[dbg] offset:48")
        if PRO_SPLIT:
            emit("v_mov_b32 v0, s100")
            emit("v_mov_b32 v1, s101")
            emit(f"global_store_dwordx2 v{V_TMP}, v[0:1], %[dbg] offset:56")
        emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    if PROBE and not stamp:
        # the probe's end: both counters, and -- only when a probe buffer is set -- lane 0 of every wave stores [cycles, real-time ticks]
        lskip = label("np")
        emit(f"s_memtime s[{S_TS + 6}:{S_TS + 7}]")
        emit(f"s_memrealtime s[{S_RT + 2}:{S_RT + 3}]")
        emit("s_cmp_eq_u64 %[dbg], 0")
        emit(f"s_cbranch_scc1 {lskip}")
        emit("s_waitcnt lgkmcnt(0)")
        emit(f"s_sub_u32 s{S_TS + 6}, s{S_TS + 6}, s{S_TS}")
        emit(f"s_subb_u32 s{S_TS + 7}, s{S_TS + 7}, s{S_TS + 1}")
        emit(f"s_sub_u32 s{S_RT + 2}, s{S_RT + 2}, s{S_RT}")
        emit(f"s_subb_u32 s{S_RT + 3}, s{S_RT + 3}, s{S_RT + 1}")
        emit(f"v_and_b32 v{V_TMP}, 63, %[tid]")
        emit(f"v_cmp_eq_u32 vcc, 0, v{V_TMP}")
        emit("s_and_b64 exec, exec, vcc")
        emit(f"v_mov_b32 v{V_TMP}, 0")
        emit(f"v_mov_b32 v{V_LDSG}, s{S_TS + 6}")
        emit(f"v_mov_b32 v{V_GOG}, s{S_RT + 2}")
        emit(f"global_store_dword v{V_TMP}, v{V_LDSG}, %[dbg]")
        emit(f"global_store_dword v{V_TMP}, v{V_GOG}, %[dbg] offset:4")
        emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
        emit(f"{lskip}:")
    emit("s_waitcnt vmcnt(0)")



def prologue(q, nw, stamp):
    emit("; ==== prologue")
    if stamp:
        emit(f"s_memtime s[{S_TS}:{S_TS + 1}]")
        emit(f"s_memrealtime s[{S_RT}:{S_RT + 1}]")
        emit("s_waitcnt lgkmcnt(0)")
    elif PROBE:
        # clock probe (mq_gemm_set_clock_probe): shader-clock and 100-MHz real-time counters at the program's start; they return with the
        # output grid's scalar loads (waited for in step 3, before any counted LDS wait)
        emit(f"s_memtime s[{S_TS}:{S_TS + 1}]")
        emit(f"s_memrealtime s[{S_RT}:{S_RT + 1}]")
    # the output grid lives in device memory (no host read-back): both scalar loads leave before anything else and are waited for in
    # step (3), behind the first stages' requests -- never in front of them (round 4: the C++ preamble used to dereference and divide
    # BEFORE the first LDS-DMA could be issued: two dependent cold scalar loads ahead of every launch's first byte)
    if SCALAR_GRID:
        emit(f"s_load_dword s{S_SO}, %[soptr], 0x0")
        emit(f"s_load_dword s{S_OO}, %[ooptr], 0x0")
    # loop state
    emit(f"s_mov_b64 s[{S_ABASE}:{S_ABASE + 1}], %[aptr]")
    emit(f"s_mov_b64 s[{S_WBASE}:{S_WBASE + 1}], %[wptr]")
    emit(f"s_lshl_b32 s{S_WK[0]}, %[wave], 10")
    if RING_BASE:
        emit(f"s_add_u32 s{S_WK[0]}, s{S_WK[0]}, {RING_BASE}")
    for i in range(1, max(PIECES, 3) if BN == 176 else PIECES):
        emit(f"s_add_u32 s{S_WK[i]}, s{S_WK[0]}, {i * NW * 1024}")
    emit(f"s_mov_b32 s{S_CUR}, 0")
    emit(f"s_mov_b32 s{S_NXT}, {W_BYTES}")
    emit(f"s_mov_b32 s{S_DMA}, {3 * W_BYTES}")
    # (1) ordinary loads first: row sums of the lane's two rows, per-n vectors of column tid (tid < 176)
    emit(f"global_load_dword v{V_RS0}, %[rsofs0], %[rsptr]")
    q.issue("P")
    emit(f"global_load_dword v{V_RS1}, %[rsofs1], %[rsptr]")
    q.issue("P")
    emit(f"s_mov_b64 s[{S_EXEC}:{S_EXEC + 1}], exec")
    emit(f"v_cmp_gt_u32 vcc, {BN}, %[tid]")
    emit("s_and_b64 exec, exec, vcc")
    emit(f"v_lshlrev_b32 v{V_TMP}, 2, %[tid]")
    for k, ptr in enumerate(("alpha", "bias", "wzp", "ct")):
        emit(f"global_load_dword v{V_P0 + k}, v{V_TMP}, %[{ptr}]")
        q.issue("P")
    emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    # (2) the first stages, in the order they are needed: W(0), A(0), W(1), A(1).ks0, W(2)
    if W4X:
        emit(f"s_mov_b32 s{S_M4}, 0x0f0f0f0f")
    for t, slot in ((0, 0), (1, W_BYTES), (2, 2 * W_BYTES)):
        emit(f"s_mov_b32 s{S_TMP}, {slot}")
        if W4XL:                # packed pieces of W(0), W(1), W(2) -> packed ring slots 0, 1, 2 (1 KiB per wave instruction)
            emit(f"s_mov_b32 s{S_TMP}, {STG} + {t * pk_bytes()}")
            for i in range(nw):
                w4xl_dma(q, t, i, S_TMP)()
            w4x_advance()
        elif W4X:               # packed pieces of W(0) -> registers; W(1) follows once W(0) is expanded (step 5)
            if t == 0:
                for i in range(nw):
                    w4x_load(q, 0, i, X_P)()
                w4x_advance()
        else:
            issue_w(q, t, nw, S_TMP)
        if t == 0:
            for ks in range(2):
                for i in range(2):
                    a_load(q, 0, ks, i, 0, 1024 * ks)()
            emit(f"s_add_u32 s{S_ABASE}, s{S_ABASE}, {2 * 1024}")       # -> stage 1
            emit(f"s_addc_u32 s{S_ABASE + 1}, s{S_ABASE + 1}, 0")
            if PRO_SPLIT:        # what-if: how soon does stage 0 ALONE arrive?  (stamp 7; the later stages are requested behind it)
                emit("s_waitcnt vmcnt(0)")
                q.q = []
                if stamp:
                    emit("s_memtime s[100:101]")
                    emit("s_waitcnt lgkmcnt(0)")
        if t == 1:
            a_load(q, 1, 0, 0, 1, 0)()
            a_load(q, 1, 0, 1, 1, 0)()
    if EPI == "gate":
        # the 64-KiB gated table -> LDS [0, 65536): 64 pieces of 1 KiB by LDS-DMA, eight per wave, behind the first stages in the queue
        emit(f"v_and_b32 v{V_RD}, 63, %[tid]")
        emit(f"v_lshlrev_b32 v{V_RD}, 4, v{V_RD}")                           # lane * 16
        emit(f"s_lshl_b32 s{S_TMP}, %[wave], 13")                            # wave * 8 KiB
        emit(f"s_mov_b64 s[{S_TS}:{S_TS + 1}], %[table]")
        emit(f"s_add_u32 s{S_TS}, s{S_TS}, s{S_TMP}")
        emit(f"s_addc_u32 s{S_TS + 1}, s{S_TS + 1}, 0")
        for i in range(8):
            emit(f"s_add_u32 m0, s{S_TMP}, {i * 1024}")
            emit("s_nop 0")
            emit(f"global_load_lds_dwordx4 v{V_RD}, s[{S_TS}:{S_TS + 1}]")
            q.issue("T")
            emit(f"s_add_u32 s{S_TS}, s{S_TS}, 1024")
            emit(f"s_addc_u32 s{S_TS + 1}, s{S_TS + 1}, 0")
    # per-lane constants while the loads fly
    emit(f"v_and_b32 v{V_TMP}, 63, %[tid]")                                # lane
    emit(f"v_and_b32 v{V_WOFF0}, 15, v{V_TMP}")                            # frow
    emit(f"v_lshrrev_b32 v{V_PAR}, 4, v{V_TMP}")                           # kq
    emit(f"v_and_b32 v{V_STW}, 7, v{V_TMP}")                               # lane & 7
    emit(f"v_xor_b32 v{V_STW}, v{V_STW}, v{V_PAR}")                        # kq ^ (lane & 7)
    emit(f"v_lshlrev_b32 v{V_STW}, 4, v{V_STW}")
    emit(f"v_lshl_add_u32 v{V_WOFF0}, v{V_WOFF0}, 7, v{V_STW}")            # frow*128 + swizzled chunk
    if RING_BASE:
        emit(f"v_add_u32 v{V_WOFF0}, {RING_BASE}, v{V_WOFF0}")
    emit(f"v_xor_b32 v{V_WOFF1}, 64, v{V_WOFF0}")
    # staging write address: STG + wave*STG_WAVE + frow*ROWP + kq*4 ; alpha' chunk address: PAR + kq*16
    emit(f"v_and_b32 v{V_STW}, 15, v{V_TMP}")
    emit(f"v_mul_u32_u24 v{V_STW}, {ROWP}, v{V_STW}")
    emit(f"v_lshl_add_u32 v{V_STW}, v{V_PAR}, {4 if EPI == 'f32r' else 2}, v{V_STW}")   # + kq*4 bytes (u8) / kq*16 (fp32)
    emit(f"s_mul_i32 s{S_TMP}, %[wave], {STG_WAVE}")
    emit(f"s_add_u32 s{S_TMP}, s{S_TMP}, {STG}")
    emit(f"v_add_u32 v{V_STW}, s{S_TMP}, v{V_STW}")
    emit(f"v_lshlrev_b32 v{V_PAR}, 4, v{V_PAR}")
    emit(f"v_add_u32 v{V_PAR}, {PAR}, v{V_PAR}")
    # (3) parameters: park alpha' = alpha/so, bias' = bias/so + oo, -w_zp, col_term in LDS (same expressions as the C++ prologue
    # of the other variants: one multiply, one multiply + one add, no contraction)
    pstamp(0, stamp)
    q.wait_for("P")
    pstamp(1, stamp)
    if SCALAR_GRID:
        # 1 / so, correctly rounded: the IEEE divide sequence hipcc emits for __fdiv_rn(1.0f, so) (the C++ epilogues of the other
        # variants divide the same way: identical alpha' / bias' bits)
        emit("s_waitcnt lgkmcnt(0)")
        vs, v1, v2, v0, v3, v4 = 122, 123, 124, 125, 126, V_TMP
        emit(f"v_mov_b32 v{vs}, s{S_SO}")
        emit(f"v_div_scale_f32 v{v1}, vcc, v{vs}, v{vs}, 1.0")
        emit(f"v_rcp_f32 v{v2}, v{v1}")
        emit("s_nop 0")
        emit(f"v_fma_f32 v{v0}, -v{v1}, v{v2}, 1.0")
        emit(f"v_fma_f32 v{v2}, v{v0}, v{v2}, v{v2}")
        emit(f"v_div_scale_f32 v{v0}, vcc, 1.0, v{vs}, 1.0")
        emit(f"v_mul_f32 v{v3}, v{v0}, v{v2}")
        emit(f"v_fma_f32 v{v4}, -v{v1}, v{v3}, v{v0}")
        emit(f"v_fma_f32 v{v3}, v{v4}, v{v2}, v{v3}")
        emit(f"v_fma_f32 v{v0}, -v{v1}, v{v3}, v{v0}")
        emit(f"v_div_fmas_f32 v{v0}, v{v0}, v{v2}, v{v3}")
        emit(f"v_div_fixup_f32 v{v0}, v{v0}, v{vs}, 1.0")
        emit("s_nop 0")
        emit(f"v_readfirstlane_b32 s{S_ISO}, v{v0}")
    emit("s_bitcmp1_b32 %[flags], 1")                                        # bit 1: row sums present
    l = label("rs")
    emit(f"s_cbranch_scc1 {l}")
    emit(f"v_mov_b32 v{V_RS0}, 0")
    emit(f"v_mov_b32 v{V_RS1}, 0")
    emit(f"{l}:")
    emit(f"v_cmp_gt_u32 vcc, {BN}, %[tid]")
    emit("s_and_b64 exec, exec, vcc")
    emit("s_bitcmp1_b32 %[flags], 0")                                        # bit 0: bias present
    l = label("nb")
    emit(f"s_cbranch_scc1 {l}")
    emit(f"v_mov_b32 v{V_P0 + 1}, 0")
    emit(f"{l}:")
    # 176: one output grid (scalars);  128 / u8: this column's grid (per-lane operands: q | k | v segments);  f32r: 16-bit grid,
    # the offset stays outside the fma (added after the rounding, as in the C++ epilogue)
    inv, oo = (f"s{S_ISO}", f"s{S_OO}") if SCALAR_GRID else ("%[invc]", "%[ooc]")
    emit(f"v_mul_f32 v{V_P0}, {inv}, v{V_P0}")
    emit(f"v_mul_f32 v{V_P0 + 1}, {inv}, v{V_P0 + 1}")
    if EPI in ("u8", "gate"):
        emit(f"v_add_f32 v{V_P0 + 1}, {oo}, v{V_P0 + 1}")
    if SPLITK:                       # the second K half starts from zero: the zero-point correction is the first half's
        emit("s_bitcmp1_b32 %[flags], 2")
        l = label("nz")
        emit(f"s_cbranch_scc0 {l}")
        emit(f"v_mov_b32 v{V_P0 + 2}, 0")
        emit(f"v_mov_b32 v{V_P0 + 3}, 0")
        emit(f"{l}:")
    emit(f"v_sub_u32 v{V_P0 + 2}, 0, v{V_P0 + 2}")
    emit(f"v_lshlrev_b32 v{V_TMP}, 2, %[tid]")
    emit(f"v_add_u32 v{V_TMP}, {PAR}, v{V_TMP}")
    for k in range(4):
        emit(f"ds_write_b32 v{V_TMP}, v{V_P0 + k} offset:{k * 4 * BN}")
    emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    if not DINIT:
        emit("s_waitcnt lgkmcnt(0)")
        emit("s_barrier")
    pstamp(2, stamp)
    # (4) accumulators = col_term[n] - w_zp[n] * row_sum[m] while the first stage is in flight.  Round 4: stage 0 lands ~1.6 k cycles
    # after the program starts, and the old form of this step (two reads, a full wait, eight mads, eleven times) alone took ~2.2 k cycles
    # behind it.  Now every -w_zp chunk is read straight into row block 0's accumulator registers (11 reads in flight), the col_term
    # chunks rotate through five 4-register sets four deep, and each column costs eight mads behind a counted wait.
    if DINIT:
        pass                        # stage 0 starts the accumulators at 0, stage 1 adds the correction (deferred_init)
    elif os.environ.get("MQ_FR_OLD_INIT"):
        for j in range(FN):
            emit(f"ds_read_b128 v[{V_E}:{V_E + 3}], v{V_PAR} offset:{2 * 4 * BN + j * 64}")
            emit(f"ds_read_b128 v[{V_E + 4}:{V_E + 7}], v{V_PAR} offset:{3 * 4 * BN + j * 64}")
            emit("s_waitcnt lgkmcnt(0)")
            for i in range(2):
                for e in range(4):
                    emit(f"v_mad_i32_i24 {accr(i, j, e)}, v{V_E + e}, v{V_RS0 + i}, v{V_E + 4 + e}")
    else:
        lqi = LQueue()
        CT = [V_E, V_E + 4, V_P0, V_P0 + 4, 122]
        def rd_ct(j):
            t = CT[j % 5]
            emit(f"ds_read_b128 v[{t}:{t + 3}], v{V_PAR} offset:{3 * 4 * BN + j * 64}")
            lqi.issue(("ct", j))
        for j in range(FN):
            emit(f"ds_read_b128 {acc(0, j)}, v{V_PAR} offset:{2 * 4 * BN + j * 64}")
            lqi.issue(("wz", j))
        for j in range(min(4, FN)):
            rd_ct(j)
        for j in range(FN):
            lqi.wait_for(("ct", j))
            t = CT[j % 5]
            for e in range(4):       # row block 1 first: it reads -w_zp from row block 0's register, which the second mad overwrites
                emit(f"v_mad_i32_i24 {accr(1, j, e)}, {accr(0, j, e)}, v{V_RS0 + 1}, v{t + e}")
                emit(f"v_mad_i32_i24 {accr(0, j, e)}, {accr(0, j, e)}, v{V_RS0}, v{t + e}")
            if j + 4 < FN:
                rd_ct(j + 4)
    # (5) W(0) of every wave landed -> first fragments
    pstamp(3, stamp)
    if W4X:
        # ring offsets of this lane's two 16-byte pieces: row = 16 wave + (lane >> 2), chunks 2 c and 2 c + 1 (c = lane & 3) at the
        # int8 image's swizzled positions (chunk ^ (row & 7)); then W(0) -> slot 0, and W(1)'s packed pieces are requested
        emit(f"v_and_b32 v{V_TMP}, 63, %[tid]")
        emit(f"v_lshrrev_b32 v{X_T}, 2, v{V_TMP}")                             # r
        emit(f"v_and_b32 v{V_TMP}, 3, v{V_TMP}")                               # c
        emit(f"v_lshlrev_b32 v{V_TMP}, 1, v{V_TMP}")                           # 2 c
        emit(f"v_and_b32 v{X_A1}, 7, v{X_T}")                                  # row & 7
        emit(f"v_xor_b32 v{X_A0}, v{V_TMP}, v{X_A1}")                          # p0
        emit(f"v_xor_b32 v{X_A1}, 1, v{X_A0}")                                 # p1
        emit(f"s_lshl_b32 s{S_TMP}, %[wave], 4")
        emit(f"v_add_u32 v{X_T}, s{S_TMP}, v{X_T}")                            # tile row
        emit(f"v_lshlrev_b32 v{X_T}, 7, v{X_T}")                               # * 128
        emit(f"v_lshl_add_u32 v{X_A0}, v{X_A0}, 4, v{X_T}")
        emit(f"v_lshl_add_u32 v{X_A1}, v{X_A1}, 4, v{X_T}")
        if RING_BASE:            # gate variants: the table sits in front of the ring
            emit(f"v_add_u32 v{X_A0}, {RING_BASE}, v{X_A0}")
            emit(f"v_add_u32 v{X_A1}, {RING_BASE}, v{X_A1}")
        lqx = LQueue()
        if W4XL:
            emit(f"s_mov_b32 s{S_PKR}, {STG}")                                  # W(0)'s packed slot; stage 0 expands slot 1, its DMA fills slot 3
            for f in w4x_readback(q, lqx, 0, nw, X_P, S_PKR) + w4x_expand(q, lqx, 0, nw, S_CUR, X_P, pk=S_PKR):
                f()
            emit(f"s_mov_b32 s{S_PKR}, {STG} + {pk_bytes()}")
            emit(f"s_mov_b32 s{S_PKD}, {STG} + {3 * pk_bytes()}")
        else:
            for f in w4x_expand(q, lqx, 0, nw, S_CUR, X_P):
                f()
            for i in range(nw):
                w4x_load(q, 1, i, X_P)()
            w4x_advance()
    q.wait_for(("W", 0))
    if DINIT:
        emit("s_waitcnt lgkmcnt(0)")            # this wave's parameter writes are in the LDS: the barrier publishes them with W(0)
    emit("s_barrier")
    pstamp(4, stamp)
    emit(f"v_add_u32 v{V_RD}, s{S_CUR}, v{V_WOFF0}")
    for j in range(FN):
        emit(f"ds_read_b128 {wreg(0, j)}, v{V_RD} offset:{j * 16 * BK}")
    emit("s_waitcnt lgkmcnt(0)")
    if stamp:
        emit(f"s_memtime s[{S_TS + 2}:{S_TS + 3}]")
        emit("s_waitcnt lgkmcnt(0)")


# ---- gate epilogue tail: (w1 index, w3 index) -> table -> w2's int8 input image (fragment-blocked) + row sums -------------------------
# At entry the wave's staging tile holds this tile's w3 indices (2 x 16 rows x 176 B), V_LDSO+r / V_GOFS+r / V_E+r are the LDS offset,
# the row-major global offset and the row of chunk c = lane + 64 r (as in the u8 epilogue), s[S_TMP] the tile's LDS base.
S_GM = 84                           # 84..95: exec masks of (i, r)


def epilogue_gate_tail(R):
    assert (R, FN) in ((3, 11), (2, 8))
    NR6 = (FN + 1) // 2             # rounds of the image store: two chunk columns per round
    A0, T0, AD0 = 24, 48, 64        # w1 index chunks v[24:47]; looked-up bytes v[48:63]; addresses v[64:79]
    GO1 = 117                       # 117..119: global offsets of the second row block (V_E + 3 .. 5 are free)
    emit(f"s_lshl_b32 s{S_TMP2}, %[ldn], 4")                                 # 16 rows further down
    for r in range(R):
        emit(f"v_add_u32 v{GO1 + r}, s{S_TMP2}, v{V_GOFS + r}")
    # masks, then the w1 indices of the six chunks (row-major u8 [M, N], written by the first launch)
    for i in range(2):
        for r in range(R):
            m = S_GM + 2 * (i * R + r)
            emit(f"v_add_u32 v{V_TMP}, {16 * i}, v{V_E + r}")
            emit(f"v_cmp_gt_i32_e64 s[{m}:{m + 1}], %[mrem], v{V_TMP}")
            if 64 * (r + 1) > 16 * FN:
                emit(f"v_cmp_gt_u32_e64 s[{S_MR}:{S_MR + 1}], 16, v{V_E + r}")
                emit(f"s_and_b64 s[{m}:{m + 1}], s[{m}:{m + 1}], s[{S_MR}:{S_MR + 1}]")
    for i in range(2):
        for r in range(R):
            m = S_GM + 2 * (i * R + r)
            a = A0 + (i * R + r) * 4
            emit(f"s_mov_b64 exec, s[{m}:{m + 1}]")
            emit(f"global_load_dwordx4 v[{a}:{a + 3}], v{(V_GOFS if i == 0 else GO1) + r}, %[aidx]")
    emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    emit("s_waitcnt lgkmcnt(0)")
    for i in range(2):
        for r in range(R):
            b = (i * R + r) * 4
            emit(f"ds_read_b128 v[{b}:{b + 3}], v{V_LDSO + r} offset:{i * 16 * ROWP}")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    emit("s_mov_b32 s78, 0x05010400")                                        # bytes [a1 b1 a0 b0] / [a3 b3 a2 b2] of (w1 dword, w3 dword)
    emit("s_mov_b32 s79, 0x07030602")
    for i in range(2):
        for r in range(R):
            a, b = A0 + (i * R + r) * 4, (i * R + r) * 4
            for d in range(4):
                for h in range(2):                               # v_perm_b32 interleaves two (w1, w3) byte pairs into two 16-bit table addresses
                    j = 4 * d + 2 * h
                    emit(f"v_perm_b32 v{AD0 + j}, v{a + d}, v{b + d}, s{78 + h}")
                    emit(f"v_lshrrev_b32 v{AD0 + j + 1}, 16, v{AD0 + j}")
                    emit(f"v_and_b32 v{AD0 + j}, 0xffff, v{AD0 + j}")
                    emit(f"ds_read_u8 v{T0 + j}, v{AD0 + j}")
                    emit(f"ds_read_u8 v{T0 + j + 1}, v{AD0 + j + 1}")
            emit("s_waitcnt lgkmcnt(0)")
            for d in range(4):
                emit(f"v_lshl_or_b32 v{b + d}, v{T0 + 4 * d + 1}, 8, v{T0 + 4 * d}")
                emit(f"v_lshl_or_b32 v{b + d}, v{T0 + 4 * d + 2}, 16, v{b + d}")
                emit(f"v_lshl_or_b32 v{b + d}, v{T0 + 4 * d + 3}, 24, v{b + d}")
            if 64 * (r + 1) > 16 * FN:                       # lanes past the last chunk hold no chunk: their "row" lies outside the block
                emit(f"s_mov_b64 exec, s[{S_MR}:{S_MR + 1}]")
            emit(f"ds_write_b128 v{V_LDSO + r}, v[{b}:{b + 3}] offset:{i * 16 * ROWP}")
            if 64 * (r + 1) > 16 * FN:
                emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    emit("s_waitcnt lgkmcnt(0)")
    # ---- row sums: lane < 32 owns row32 = lane of the wave's 32 rows: 11 x 16 bytes, sum of (byte ^ 0x80) - 128 * 176
    V_L, V_RA, V_ACC, V_T2 = 124, 125, 126, 127
    emit(f"v_and_b32 v{V_L}, 63, %[tid]")
    emit(f"v_and_b32 v{V_RA}, 31, v{V_L}")                                   # row32
    emit(f"v_lshrrev_b32 v{V_T2}, 4, v{V_RA}")                               # row block 0 / 1
    emit(f"v_mul_u32_u24 v{V_T2}, {16 * ROWP}, v{V_T2}")
    emit(f"v_and_b32 v{V_ACC}, 15, v{V_RA}")
    emit(f"v_mul_u32_u24 v{V_ACC}, {ROWP}, v{V_ACC}")
    emit(f"v_add3_u32 v{V_T2}, v{V_T2}, v{V_ACC}, s{S_TMP}")                 # LDS address of the row
    emit(f"v_cmp_gt_i32 vcc, %[mrem], v{V_RA}")
    emit(f"s_mov_b64 s[{S_GM}:{S_GM + 1}], vcc")                             # row32 < rows left (all lanes)
    for c in range(FN):
        emit(f"ds_read_b128 v[{4 * c}:{4 * c + 3}], v{V_T2} offset:{16 * c}")
    emit(f"v_mov_b32 v{V_ACC}, 0")
    emit("s_waitcnt lgkmcnt(0)")
    for k in range(4 * FN):
        emit(f"v_xor_b32 v{T0}, 0x80808080, v{k}")
        emit(f"v_sad_u8 v{V_ACC}, v{T0}, 0, v{V_ACC}")
    emit(f"v_add_u32 v{V_ACC}, {(-128 * BN) & 0xffffffff}, v{V_ACC}")
    emit(f"v_lshlrev_b32 v{T0 + 1}, 2, v{V_RA}")
    emit(f"v_cmp_gt_u32 vcc, 32, v{V_L}")
    emit(f"s_and_b64 vcc, vcc, s[{S_GM}:{S_GM + 1}]")
    emit("s_and_b64 exec, exec, vcc")
    emit(f"global_atomic_add v{T0 + 1}, v{V_ACC}, %[rsout]")
    emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    # ---- image: unit u = lane + 64 r6 -> chunk ch = 2 r6 + (lane >> 5), row32 = lane & 31: 16 lanes = 16 rows of one chunk = 256 bytes
    V_HI, V_BLK, V_R16 = 122, 123, 121
    emit(f"v_lshrrev_b32 v{V_HI}, 5, v{V_L}")                                # lane >> 5
    emit(f"v_lshl_add_u32 v{V_T2}, v{V_HI}, 4, v{V_T2}")                     # LDS: row address + 16 * (lane >> 5)
    emit(f"s_lshr_b32 s{S_TMP2}, %[ldn], 6")                                 # 64-column blocks per row of blocks
    emit(f"v_lshrrev_b32 v{V_BLK}, 4, v{V_RA}")
    emit(f"v_add_u32 v{V_BLK}, %[mb0], v{V_BLK}")
    emit(f"v_mul_lo_u32 v{V_BLK}, v{V_BLK}, s{S_TMP2}")                      # (mb0 + (row32 >> 4)) * blocks per row
    emit(f"v_and_b32 v{V_R16}, 15, v{V_RA}")
    emit(f"v_lshlrev_b32 v{V_R16}, 4, v{V_R16}")                             # (row32 & 15) * 16
    for r6 in range(NR6):
        emit(f"ds_read_b128 v[{4 * r6}:{4 * r6 + 3}], v{V_T2} offset:{32 * r6}")
    emit("s_waitcnt lgkmcnt(0)")
    for r6 in range(NR6):
        emit(f"s_add_u32 s{S_TMP}, %[cg0], {2 * r6}")
        emit(f"v_add_u32 v{T0}, s{S_TMP}, v{V_HI}")                          # chunk column of the whole matrix
        emit(f"v_lshrrev_b32 v{T0 + 1}, 2, v{T0}")
        emit(f"v_add_u32 v{T0 + 1}, v{T0 + 1}, v{V_BLK}")                    # block index
        emit(f"v_and_b32 v{T0}, 3, v{T0}")
        emit(f"v_lshlrev_b32 v{T0}, 8, v{T0}")
        emit(f"v_lshl_add_u32 v{T0 + 1}, v{T0 + 1}, 10, v{T0}")
        emit(f"v_add_u32 v{T0 + 1}, v{T0 + 1}, v{V_R16}")
        if 2 * r6 + 1 >= FN:
            emit(f"v_cmp_eq_u32 vcc, 0, v{V_HI}")                            # chunk FN does not exist
            emit(f"s_and_b64 vcc, vcc, s[{S_GM}:{S_GM + 1}]")
            emit("s_and_b64 exec, exec, vcc")
        else:
            emit(f"s_and_b64 exec, exec, s[{S_GM}:{S_GM + 1}]")
        emit(f"global_store_dwordx4 v{T0 + 1}, v[{4 * r6}:{4 * r6 + 3}], %[qout]")
        emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    emit("s_waitcnt vmcnt(0)")


def epilogue(stamp):
    emit("; ==== epilogue: u8 = cvt_pk_u8(fma(float(acc), alpha', bias')) -> staging tile -> whole-row 16-byte stores")
    if stamp:
        emit(f"s_memtime s[{S_TS + 4}:{S_TS + 5}]")
        emit("s_waitcnt lgkmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 3")
    if EPI == "gate":
        emit("s_barrier")         # the staging tiles alias the W ring: every wave has left the main loop
    EA = [V_E, V_P0]              # two sets of (alpha'[4], bias'[4]): v[114:121] and v[106:113] (free after the prologue)
    VP = [122, 123]               # packed dwords for i = 0, 1

    def load_params(j, s):
        emit(f"ds_read_b128 v[{EA[s]}:{EA[s] + 3}], v{V_PAR} offset:{j * 64}")
        emit(f"ds_read_b128 v[{EA[s] + 4}:{EA[s] + 7}], v{V_PAR} offset:{4 * BN + j * 64}")

    load_params(0, 0)
    for j in range(FN):
        s = j & 1
        if j + 1 < FN:
            load_params(j + 1, 1 - s)
            emit("s_waitcnt lgkmcnt(2)")
        else:
            emit("s_waitcnt lgkmcnt(0)")
        for i in range(2):
            for e in range(4):
                emit(f"v_cvt_f32_i32 {accr(i, j, e)}, {accr(i, j, e)}")
        for i in range(2):
            for e in range(4):
                emit(f"v_fma_f32 {accr(i, j, e)}, {accr(i, j, e)}, v{EA[s] + e}, v{EA[s] + 4 + e}")
        for i in range(2):
            for e in range(4):
                emit(f"v_cvt_pk_u8_f32 v{VP[i]}, {accr(i, j, e)}, {e}, " + (f"v{VP[i]}" if e else "0"))
        for i in range(2):
            if EPI != "gate":
                emit(f"v_xor_b32 v{VP[i]}, %[xorv], v{VP[i]}")
            emit(f"ds_write_b32 v{V_STW}, v{VP[i]} offset:{i * 16 * ROWP + j * 16}")
    # chunk c = lane + 64 r of the 16 x FN chunks of a row block: row = c / FN, ch = c % FN
    R = (16 * FN + 63) // 64
    emit(f"v_and_b32 v{V_TMP}, 63, %[tid]")
    emit(f"s_mul_i32 s{S_TMP}, %[wave], {STG_WAVE}")
    emit(f"s_add_u32 s{S_TMP}, s{S_TMP}, {STG}")
    for r in range(R):
        c, row, ch = 124, 125, 126
        emit(f"v_add_u32 v{c}, {64 * r}, v{V_TMP}")
        if FN == 11:
            emit(f"v_mul_u32_u24 v{row}, 5958, v{c}")                        # floor(c / 11) for c < 192 (5958 = ceil(2^16 / 11))
            emit(f"v_lshrrev_b32 v{row}, 16, v{row}")
            emit(f"v_mul_u32_u24 v{ch}, 11, v{row}")
            emit(f"v_sub_u32 v{ch}, v{c}, v{ch}")
        elif FN == 10:
            emit(f"v_mul_u32_u24 v{row}, 6554, v{c}")                        # floor(c / 10) for c < 192 (6554 = ceil(2^16 / 10))
            emit(f"v_lshrrev_b32 v{row}, 16, v{row}")
            emit(f"v_mul_u32_u24 v{ch}, 10, v{row}")
            emit(f"v_sub_u32 v{ch}, v{c}, v{ch}")
        else:
            assert FN == 8
            emit(f"v_lshrrev_b32 v{row}, 3, v{c}")
            emit(f"v_and_b32 v{ch}, 7, v{c}")
        emit(f"v_lshlrev_b32 v{ch}, 4, v{ch}")                               # ch * 16 bytes
        emit(f"v_mul_u32_u24 v{V_LDSO + r}, {ROWP}, v{row}")
        emit(f"v_add_u32 v{V_LDSO + r}, v{V_LDSO + r}, v{ch}")
        emit(f"v_add_u32 v{V_LDSO + r}, s{S_TMP}, v{V_LDSO + r}")
        emit(f"v_mul_lo_u32 v{V_GOFS + r}, v{row}, %[ldn]")
        emit(f"v_add_u32 v{V_GOFS + r}, v{V_GOFS + r}, v{ch}")
        emit(f"v_mov_b32 v{V_E + r}, v{row}")                                # row index kept for the M bound
    if EPI == "gate":
        epilogue_gate_tail(R)
        return
    emit("s_waitcnt lgkmcnt(0)")
    for i in range(2):
        for r in range(R):
            b = (i * R + r) * 4
            emit(f"ds_read_b128 v[{b}:{b + 3}], v{V_LDSO + r} offset:{i * 16 * ROWP}")
    emit(f"s_lshl_b32 s{S_TMP2}, %[ldn], 4")                                 # 16 rows further down
    emit("s_waitcnt lgkmcnt(0)")
    for i in range(2):
        for r in range(R):
            b = (i * R + r) * 4
            # rows valid for this wave: %[mrem] (may be <= 0 or > 32); lane active if i*16 + row < mrem (and c < 16 FN in a partial round)
            emit(f"v_add_u32 v{V_TMP}, {16 * i}, v{V_E + r}")
            emit(f"v_cmp_gt_i32 vcc, %[mrem], v{V_TMP}")
            if 64 * (r + 1) > 16 * FN:
                emit(f"v_cmp_gt_u32_e64 s[{S_MR}:{S_MR + 1}], 16, v{V_E + r}")
                emit(f"s_and_b64 vcc, vcc, s[{S_MR}:{S_MR + 1}]")
            emit("s_and_b64 exec, exec, vcc")
            if i == 1:
                emit(f"v_add_u32 v{V_GOFS + r}, s{S_TMP2}, v{V_GOFS + r}")
            if STORE_POLICY != "none":
                emit(f"global_store_dwordx4 v{V_GOFS + r}, v[{b}:{b + 3}], %[outw] {STORE_POLICY}".rstrip())
            emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    if stamp:
        emit("s_waitcnt vmcnt(0)")                                           # the wave's stores have left
        emit(f"s_memtime s[{S_TS + 6}:{S_TS + 7}]")
        emit(f"s_memrealtime s[{S_RT + 2}:{S_RT + 3}]")
        emit("s_waitcnt lgkmcnt(0)")
        # lane 0 of every wave: four stamps -> dbg[(block*8 + wave)*16 + 0..3]
        emit(f"v_and_b32 v{V_TMP}, 63, %[tid]")
        emit(f"v_cmp_eq_u32 vcc, 0, v{V_TMP}")
        emit("s_and_b64 exec, exec, vcc")
        emit(f"v_mov_b32 v{V_TMP}, 0")
        for k in range(6):
            src = S_TS + 2 * k if k < 4 else S_RT + 2 * (k - 4)
            emit(f"v_mov_b32 v0, s{src}")
            emit(f"v_mov_b32 v1, s{src + 1}")
            emit(f"global_store_dwordx2 v{V_TMP}, v[0:1], %[dbg] offset:{8 * k}")
        emit("v_mov_b32 v0, %[tentry_lo]")                                   # s_memrealtime at the kernel's first instruction (C++)
        emit("v_mov_b32 v1, %[tentry_hi]")
        emit(f"global_store_dwordx2 v{V_TMP}, v[0:1], %[dbg] offset:48")
        emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    emit("s_waitcnt vmcnt(0)")


# ---- fp32 epilogue with the residual add: out = resid + (clamp(rint(fma(float(acc), alpha', bias')) + oo, qmin, qmax) - oo) * so ------
# (o_proj / w2: a 16-bit output grid in front of the residual stream; the same operations in the same order as the C++ epilogue of
# gemm_i8_kernel<.., MQ_F32, OUTQ>, so both kernels produce the same bits.)  A wave converts one UNIT = (row block i, column half h)
# at a time: 16 rows x 64 columns through a wave-private fp32 staging tile, read back as 256-byte row pieces -- lane (row0 = lane >> 4,
# ch = lane & 15) handles chunk ch of rows row0 + 4 r, r = 0..3.  The residual rows of unit u + 1 are requested before unit u is
# converted (two register sets: v[78:93] and the accumulators unit 0 has released).
V_RDB, V_G = 98, (99, 94, 95, 96)   # staging read-back address; global byte offsets of the lane's chunk in rows row0 + 4 r
V_ROW, V_QMIN, V_QMAX = 97, 122, 123
S_OB, S_RB, S_STEP = 74, 76, 78     # output / residual pointers of the current unit (pairs); 4 rows in bytes
S_MASK = 80                         # 80..95: exec masks of (i, r)


def splitk_exchange(keep):
    """fr128rs: this wave sends its int32 partial sums of row block 1 - keep to the same wave of the partner workgroup and adds the
    partner's sums of row block `keep` to its own.  %[xch] -> the tile's scratch [2 (destination role)][8 waves][8 quads][64 lanes x 16 B],
    %[xfl] -> its flags [2][8] (zero when the launch starts, zero again when it ends).  The partner is resident: the first-half
    workgroups have the lower block ids (dispatched first), and 2 x tiles <= the number of CUs.  System-scope accesses: the two
    workgroups need not share an L2."""
    send = 1 - keep
    if os.environ.get("MQ_FR_XNOX"):       # what-if (wrong sums): no exchange at all -- what the halved loop + half an epilogue cost alone
        return
    VX, VF, VT = 98, 99, 100
    XB, FL, SC = 74, 76, 78
    # cache policies: the partner runs on the SAME XCD (block ids t and t + tiles, tiles % 8 == 0: round-robin dispatch), so the partial
    # sums travel through that XCD's L2 -- plain stores (the vector L1 writes through), loads that miss the L1 (sc1); system-scope
    # accesses (sc0 sc1 both ways) cost ~11 us per launch at 16.8 MB each way
    POL_ST = os.environ.get("MQ_FR_XPOL_ST", "")
    POL_LD = os.environ.get("MQ_FR_XPOL_LD", "sc1")
    emit(f"; ---- split-K exchange: keep row block {keep}, send row block {send}")
    emit(f"v_and_b32 v{VX}, 63, %[tid]")
    emit(f"v_lshlrev_b32 v{VX}, 4, v{VX}")                                   # lane * 16
    emit(f"s_lshl_b32 s{S_TMP}, %[wave], 13")                                # wave * 8 KiB
    emit(f"s_mov_b64 s[{XB}:{XB + 1}], %[xch]")
    emit(f"s_add_u32 s{XB}, s{XB}, s{S_TMP}")
    emit(f"s_addc_u32 s{XB + 1}, s{XB + 1}, 0")
    emit(f"s_mov_b64 s[{FL}:{FL + 1}], s[{XB}:{XB + 1}]")                    # (FL doubles as the read base until the flags are needed)
    if send:
        emit(f"s_add_u32 s{XB}, s{XB}, 65536")
        emit(f"s_addc_u32 s{XB + 1}, s{XB + 1}, 0")
    else:
        emit(f"s_add_u32 s{FL}, s{FL}, 65536")
        emit(f"s_addc_u32 s{FL + 1}, s{FL + 1}, 0")
    for j in range(FN):
        if j == 4:
            emit(f"s_add_u32 s{XB}, s{XB}, 4096")
            emit(f"s_addc_u32 s{XB + 1}, s{XB + 1}, 0")
        emit(f"global_store_dwordx4 v{VX}, {acc(send, j)}, s[{XB}:{XB + 1}] offset:{(j % 4) * 1024} {POL_ST}".rstrip())
    emit("s_waitcnt vmcnt(0)")
    # flags: mine to raise = [send][wave], mine to wait for = [keep][wave]
    emit(f"s_lshl_b32 s{S_TMP}, %[wave], 2")
    emit(f"s_add_u32 s{S_TMP2}, s{S_TMP}, {32 * keep}")
    emit(f"s_add_u32 s{S_TMP}, s{S_TMP}, {32 * send}")
    emit(f"v_mov_b32 v{VF}, s{S_TMP}")
    emit(f"v_mov_b32 v{VT}, 1")
    emit(f"global_store_dword v{VF}, v{VT}, %[xfl] {POL_ST}".rstrip())
    emit(f"v_mov_b32 v{VF}, s{S_TMP2}")
    emit(f"s_mov_b32 s{SC}, 0")
    lp, ld = label("xw"), label("xd")
    emit(f"{lp}:")
    emit(f"global_load_dword v{VT}, v{VF}, %[xfl] {POL_LD}".rstrip())
    emit("s_waitcnt vmcnt(0)")
    emit(f"v_readfirstlane_b32 s{S_TMP}, v{VT}")
    emit(f"s_cmp_eq_u32 s{S_TMP}, 1")
    emit(f"s_cbranch_scc1 {ld}")
    emit("s_sleep 1")
    emit(f"s_add_u32 s{SC}, s{SC}, 1")
    emit(f"s_cmp_lt_u32 s{SC}, {1 << 21}")                                   # a lost partner must not hang the device: wrong sums, caught by the tests
    emit(f"s_cbranch_scc1 {lp}")
    emit(f"{ld}:")
    emit(f"v_mov_b32 v{VT}, 0")
    emit(f"global_store_dword v{VF}, v{VT}, %[xfl] {POL_ST}".rstrip())                   # consumed: the flag is zero again for the next launch
    for j in range(FN):
        if j == 4:
            emit(f"s_add_u32 s{FL}, s{FL}, 4096")
            emit(f"s_addc_u32 s{FL + 1}, s{FL + 1}, 0")
        emit(f"global_load_dwordx4 {acc(send, j)}, v{VX}, s[{FL}:{FL + 1}] offset:{(j % 4) * 1024} {POL_LD}".rstrip())
    emit("s_waitcnt vmcnt(0)")
    for j in range(FN):
        for e in range(4):
            emit(f"v_add_u32 {accr(keep, j, e)}, {accr(keep, j, e)}, {accr(send, j, e)}")


def epilogue_f32r():
    if not SPLITK:
        return epilogue_f32r_units((0, 1, 2, 3))
    emit("s_nop 15")
    emit("s_nop 3")
    l1, lend = label("role"), label("xend")
    emit("s_bitcmp1_b32 %[flags], 2")
    emit(f"s_cbranch_scc1 {l1}")
    splitk_exchange(0)
    epilogue_f32r_units((0, 1))
    emit(f"s_branch {lend}")
    emit(f"{l1}:")
    splitk_exchange(1)
    epilogue_f32r_units((2, 3))
    emit(f"{lend}:")


def epilogue_f32r_units(units):
    emit("; ==== epilogue: fp32 x + Qout16(linear), unit by unit")
    emit("s_nop 15")
    emit("s_nop 3")
    D = [106 + 4 * r for r in range(4)]
    RS = [[78 + 4 * r for r in range(4)], [(2 * jj) * 4 for jj in range(4)]]     # residual register sets (set 1 = acc(0, 0..3))
    EA = [V_E, V_P0]
    q = Queue()

    emit(f"v_mov_b32 v{V_QMIN}, %[qmin]")
    emit(f"v_mov_b32 v{V_QMAX}, %[qmax]")
    emit(f"v_and_b32 v{V_TMP}, 63, %[tid]")
    emit(f"v_lshrrev_b32 v{V_ROW}, 4, v{V_TMP}")                             # row0
    emit(f"v_and_b32 v{V_TMP}, 15, v{V_TMP}")
    emit(f"v_lshlrev_b32 v{V_TMP}, 4, v{V_TMP}")                             # ch * 16 bytes
    emit(f"s_mul_i32 s{S_TMP}, %[wave], {STG_WAVE}")
    emit(f"s_add_u32 s{S_TMP}, s{S_TMP}, {STG}")
    emit(f"v_mul_u32_u24 v{V_RDB}, {ROWP}, v{V_ROW}")
    emit(f"v_add_u32 v{V_RDB}, v{V_RDB}, v{V_TMP}")
    emit(f"v_add_u32 v{V_RDB}, s{S_TMP}, v{V_RDB}")
    emit(f"s_lshl_b32 s{S_TMP}, %[ldn], 2")                                  # bytes per output row
    emit(f"s_lshl_b32 s{S_STEP}, %[ldn], 4")                                 # 4 rows
    emit(f"v_mul_lo_u32 v{V_G[0]}, v{V_ROW}, s{S_TMP}")
    emit(f"v_add_u32 v{V_G[0]}, v{V_G[0]}, v{V_TMP}")
    for r in range(1, 4):
        emit(f"v_add_u32 v{V_G[r]}, s{S_STEP}, v{V_G[r - 1]}")
    emit(f"s_mov_b64 s[{S_OB}:{S_OB + 1}], %[outw]")
    emit(f"s_mov_b64 s[{S_RB}:{S_RB + 1}], %[resid]")
    emit(f"s_lshl_b32 s{S_TMP2}, %[ldn], 6")                                 # 16 rows ...
    if units[0] == 2:                                                        # (split-K, second role: row block 1 only)
        for sp in (S_OB, S_RB):
            emit(f"s_add_u32 s{sp}, s{sp}, s{S_TMP2}")
            emit(f"s_addc_u32 s{sp + 1}, s{sp + 1}, 0")
    emit(f"s_sub_u32 s{S_TMP2}, s{S_TMP2}, {HALF * 4}")                      # ... minus the half row already advanced
    for i in range(2):
        for r in range(4):
            m = S_MASK + 2 * (4 * i + r)
            emit(f"v_add_u32 v{V_TMP}, {16 * i + 4 * r}, v{V_ROW}")
            emit(f"v_cmp_gt_i32_e64 s[{m}:{m + 1}], %[mrem], v{V_TMP}")

    def advance(sp, u):
        if u == 3:
            return
        if u == 1:
            emit(f"s_add_u32 s{sp}, s{sp}, s{S_TMP2}")
        else:
            emit(f"s_add_u32 s{sp}, s{sp}, {HALF * 4}")
        emit(f"s_addc_u32 s{sp + 1}, s{sp + 1}, 0")

    def issue_resid(u):
        i = u >> 1
        for r in range(4):
            m = S_MASK + 2 * (4 * i + r)
            b = RS[u & 1][r]
            emit(f"s_mov_b64 exec, s[{m}:{m + 1}]")
            emit(f"global_load_dwordx4 v[{b}:{b + 3}], v{V_G[r]}, s[{S_RB}:{S_RB + 1}]")
            q.issue(("R", u))
        emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
        advance(S_RB, u)

    def load_params(j, st):
        emit(f"ds_read_b128 v[{EA[st]}:{EA[st] + 3}], v{V_PAR} offset:{j * 64}")
        emit(f"ds_read_b128 v[{EA[st] + 4}:{EA[st] + 7}], v{V_PAR} offset:{4 * BN + j * 64}")

    def convert(u):
        i, h = u >> 1, u & 1
        js = list(range(4 * h, 4 * h + 4))
        load_params(js[0], 0)
        for n, j in enumerate(js):
            st = n & 1
            if n + 1 < len(js):
                load_params(js[n + 1], 1 - st)
                emit("s_waitcnt lgkmcnt(2)")
            else:
                emit("s_waitcnt lgkmcnt(0)")
            regs = [accr(i, j, e) for e in range(4)]
            for e, x in enumerate(regs):
                emit(f"v_cvt_f32_i32 {x}, {x}")
            for e, x in enumerate(regs):
                emit(f"v_fma_f32 {x}, {x}, v{EA[st] + e}, v{EA[st] + 4 + e}")
            for x in regs:
                emit(f"v_rndne_f32 {x}, {x}")
            for x in regs:
                emit(f"v_add_f32 {x}, s{S_OO}, {x}")
            for x in regs:
                emit(f"v_med3_f32 {x}, {x}, v{V_QMIN}, v{V_QMAX}")
            for x in regs:
                emit(f"v_subrev_f32 {x}, s{S_OO}, {x}")
            for x in regs:
                emit(f"v_mul_f32 {x}, s{S_SO}, {x}")
            emit(f"ds_write_b128 v{V_STW}, {acc(i, j)} offset:{n * 64}")

    issue_resid(units[0])
    for u in units:
        i = u >> 1
        if u != units[0]:
            emit("s_nop 2")                                                  # the stores of unit u - 1 have taken their data (D = the parameter registers)
        convert(u)
        if u == units[0]:
            issue_resid(u + 1)                                               # set 1 = accumulators unit 0 has just released (split-K: sent away)
        for r in range(4):
            emit(f"ds_read_b128 v[{D[r]}:{D[r] + 3}], v{V_RDB} offset:{r * 4 * ROWP}")
        emit("s_waitcnt lgkmcnt(0)")
        q.wait_for(("R", u))
        for r in range(4):
            b = RS[u & 1][r]
            for e in range(4):
                emit(f"v_add_f32 v{D[r] + e}, v{b + e}, v{D[r] + e}")
        for r in range(4):
            m = S_MASK + 2 * (4 * i + r)
            emit(f"s_mov_b64 exec, s[{m}:{m + 1}]")
            emit(f"global_store_dwordx4 v{V_G[r]}, v[{D[r]}:{D[r] + 3}], s[{S_OB}:{S_OB + 1}] nt")
            q.issue(("S", u))
        emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
        advance(S_OB, u)
        if u + 2 <= units[-1]:
            issue_resid(u + 2)                                               # its register set was consumed by the adds above
    emit("s_waitcnt vmcnt(0)")


def program(nw, stamp):
    q = Queue()
    prologue(q, nw, stamp)
    if os.environ.get("MQ_FR_PRIO"):       # what-if: static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md)
        lp = label("prio")
        emit(f"s_cmp_lt_u32 %[wave], {NW // 2}")
        emit(f"s_cbranch_scc1 {lp}")
        emit("s_setprio 1")
        emit(f"{lp}:")
    stage(q, 0, nw, first=DINIT)
    stage(q, 1, nw, dinit="wz" if DINIT else None)
    # steady state: pairs (t, t+1), t = 2, 4, ..., kt - 6;  pairs = (kt - 6) / 2  (>= 0)
    emit(f"s_sub_u32 s{S_CNT}, %[kt], 6")
    emit(f"s_lshr_b32 s{S_CNT}, s{S_CNT}, 1")
    lend, lloop = label("tail"), label("loop")
    emit(f"s_cmp_eq_u32 s{S_CNT}, 0")
    emit(f"s_cbranch_scc1 {lend}")
    emit(f"{lloop}:")
    before = list(q.q)
    start = len(out)
    nlog = len(q.log)
    stage(q, 2, nw, sym="T")
    stage(q, 3, nw, sym="T+1")
    body_waits = q.log[nlog:]
    shifted = [tuple(x[:1]) + (x[1] - 2,) + tuple(x[2:]) if isinstance(x, tuple) else x for x in q.q]
    assert shifted == before, (before, q.q)        # the loop body is a fixed point of the VMEM queue
    # a second simulated iteration must produce the same immediates
    q2 = Queue(); q2.q = [tuple(x[:1]) + (x[1] + 2,) + tuple(x[2:]) if isinstance(x, tuple) else x for x in before]
    saved = len(out)
    stage(q2, 4, nw); stage(q2, 5, nw)
    assert q2.log == body_waits, (q2.log, body_waits)
    del out[saved:]
    emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    emit(f"s_cmp_lg_u32 s{S_CNT}, 0")
    emit(f"s_cbranch_scc1 {lloop}")
    emit(f"{lend}:")
    # tail: the last four stages; renumber the queue as if kt = 8 (tags are relative)
    KT = 8
    q.q = [tuple(x[:1]) + (x[1] + 2,) + tuple(x[2:]) if isinstance(x, tuple) else x for x in before]
    if TAIL:
        stage(q, 4, nw, kt=KT, sym="KT-4", dinit="ct" if DINIT else None)
        lq = LQueue()
        stage_pre_final(q, lq, 5, nw)
        final_block(q, lq, stamp)
        return q
    for t in range(4, 8):
        stage(q, t, nw, kt=KT, sym=f"KT-{KT - t}", dinit="ct" if DINIT and t == 4 else None)
    assert q.q == [], q.q
    if EPI in ("u8", "gate"):
        epilogue(stamp)
    else:
        assert not stamp
        epilogue_f32r()
    return q


# ---- round 4: packed 4-bit weights ---------------------------------------------------------------------------------------------------
# mq_pack_w4's image: row n = K / 2 bytes; 16 bytes = 32 consecutive k, element p in the low and p + 16 in the high nibble of byte p.
# A K = 128 stage of a row is four such chunks.  Lane (frow = lane & 15, kq = lane >> 4) reads chunk kq of row 16 j + frow with ONE
# ds_read_b128 and owns 32 k of the stage: its low nibbles are its 16 bytes of the stage's FIRST MFMA (k = 32 kq + 0..15), its high
# nibbles those of the SECOND (k = 32 kq + 16..31).  The activation fragments follow that split: the fragment-blocked image holds x[row,
# 64 kb + 16 q + p] at block kb, lane 16 q + row, so "type A" (low) fragments are gathered at per-lane offset (kq >> 1) KiB + (2 (kq & 1)
# 16 + row) 16 and "type B" (high) 256 bytes further (C++ forms av0 / av1 accordingly) -- the same 2 KiB per stage as the int8 kernel,
# as 256-byte runs.  LDS image of a stage: BN rows x 64 B, DMA pieces of 16 rows (lane -> row = lane >> 2, stored chunk lane & 3 holds
# logical chunk (lane & 3) ^ g(row), g = {0, 3, 2, 1}[(row >> 2) & 3]: conflict-free ds_read_b128 at the 64-byte pitch).
# Every stage runs COLUMN outer (for j: unpack, 4 MFMAs), so only two raw quads and one low-nibble quad live in VGPRs:
#   v98 read address | v99 per-lane read offset | v100 parameter address | v101 / v102 row sums (until stage 1) | v103 temporary
#   v[104:111] deferred-init chunk sets (prologue: parameter loads, divide temporaries 108..115) | v[116:119] / v[120:123] raw quads
#   (high nibbles in place) | v[124:127] low nibbles.  Final block: parameter sets v[100:107] / v[108:115]; the second read address, the
#   parameter address and the store offset move into the (dead) LDS-DMA source operands sw0 / sw1 / sw2 ("+v" operands).
#   AGPR a[0:31] activation sets (set, type, row block), a[32:47] the last stage's activations (final block).
W_RD, W_WOFF, W_PAR, W_RS0, W_RS1, W_TMP = 98, 99, 100, 101, 102, 103
W_D = (104, 108)
W_R = (116, 120)
W_L = 124
S_M4 = 99                          # 0x0f0f0f0f


def wa(set_, typ, i):
    b = (32 if set_ == 2 else 16 * set_) + 8 * typ + 4 * i
    return f"a[{b}:{b + 3}]"


def w4_a_load(q, t, typ, i, set_, off, base=None):
    def f():
        o = off + 256 * typ
        assert o < 4096
        b = base if base is not None else S_ABASE
        emit(f"global_load_dwordx4 {wa(set_, typ, i)}, %[av{i}], s[{b}:{b + 1}]" + (f" offset:{o}" if o else ""))
        q.issue(("A", t))
    return f


def w4_piece(q, t, i, slot_sgpr):
    def f():
        emit(f"s_add_u32 m0, s{slot_sgpr}, s{S_WK[i]}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 %[sw{i}], s[{S_WBASE}:{S_WBASE + 1}]")
        q.issue(("W", t))
    return f


def w4_unpack_mfma(j, r, aset, fill=()):
    """column j of one stage from raw quad r: low nibbles -> W_L, 2 MFMAs; high nibbles in place, 2 MFMAs.  fill: callables spread
    behind the four MFMAs (one list per MFMA)."""
    fill = list(fill) + [[]] * 4
    for e in range(4):
        emit(f"v_and_b32 v{W_L + e}, s{S_M4}, v{r + e}")
    emit("s_nop 1")
    for i in range(2):
        emit(f"v_mfma_i32_16x16x64_i8 {acc(i, j)}, v[{W_L}:{W_L + 3}], {wa(aset, 0, i)}, {acc(i, j)}" if not w4_unpack_mfma.zero else
             f"v_mfma_i32_16x16x64_i8 {acc(i, j)}, v[{W_L}:{W_L + 3}], {wa(aset, 0, i)}, 0")
        if i == 0:
            for e in range(4):
                emit(f"v_lshrrev_b32 v{r + e}, 4, v{r + e}")
        else:
            for e in range(4):
                emit(f"v_and_b32 v{r + e}, s{S_M4}, v{r + e}")
        for f in fill[i]:
            f()
    emit("s_nop 1")
    for i in range(2):
        emit(f"v_mfma_i32_16x16x64_i8 {acc(i, j)}, v[{r}:{r + 3}], {wa(aset, 1, i)}, {acc(i, j)}")
        for f in fill[2 + i]:
            f()


w4_unpack_mfma.zero = False


def w4_rotate():
    emit(f"s_mov_b32 s{S_CUR}, s{S_NXT}")
    for sg in (S_NXT, S_DMA):
        emit(f"s_add_u32 s{sg}, s{sg}, {W_BYTES}")
        emit(f"s_cmp_eq_u32 s{sg}, {RING * W_BYTES}")
        emit(f"s_cselect_b32 s{sg}, 0, s{sg}")
    emit(f"s_add_u32 s{S_ABASE}, s{S_ABASE}, 2048")
    emit(f"s_addc_u32 s{S_ABASE + 1}, s{S_ABASE + 1}, 0")


def w4_stage(q, lq, t, nw, kt=None, sym=None, first=False, dinit=None, pre_final=False):
    """one K = 128 stage, column outer.  At entry raw(t, column 0) is in flight into W_R[0] (address register W_RD = slot(t) + W_WOFF);
    at exit raw(t + 1, 0) is (unless pre_final: the final block prefetches its own)."""
    set_ = t & 1
    more1 = kt is None or t + 1 < kt
    more3 = kt is None or t + 3 < kt
    emit(f"; ---- W4 stage {sym or t}: A set {set_}")
    q.wait_for(("A", t))
    vm = {}
    if more1 and not NO_A:                       # A(t + 1): both types of both row blocks -> the other set
        for n, (typ, i) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            vm.setdefault(n, []).append(w4_a_load(q, t + 1, typ, i, 1 - set_, 2048))
        if pre_final:                            # and A(KT - 1) -> the third set (both register sets are in use until the final block)
            def base2():                         # (two stages ahead is beyond the 12-bit offset field: a second base in s[74:75])
                emit(f"s_add_u32 s{S_TS}, s{S_ABASE}, 4096")
                emit(f"s_addc_u32 s{S_TS + 1}, s{S_ABASE + 1}, 0")
            vm.setdefault(4, []).append(base2)
            for n, (typ, i) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
                vm.setdefault(4 + n, []).append(w4_a_load(q, t + 2, typ, i, 2, 0, base=S_TS))
    if more3 and not NO_W:
        for i in range(nw):
            vm.setdefault(FN // 2 + 1 + i, []).append(w4_piece(q, t + 3, i, S_DMA))
    # deferred zero-point correction: at column j the tiles of column (j + FN // 2) % FN, chunk read one column ahead
    base = (2 if dinit == "wz" else 3) * 4 * BN
    def chunk_read(c, k):
        def f():
            emit(f"ds_read_b128 v[{W_D[k]}:{W_D[k] + 3}], v{W_PAR} offset:{base + c * 64}")
            lq.issue(("C", k, c))
        return f
    if dinit:
        chunk_read((FN // 2) % FN, 0)()
    w4_unpack_mfma.zero = first
    par = (t * FN) % 2                           # FN odd: the raw-quad parity of column 0 alternates from stage to stage
    for j in range(FN):
        r, rn = W_R[(j + par) % 2], W_R[(j + 1 + par) % 2]
        last = j == FN - 1
        if not last:
            emit(f"ds_read_b128 v[{rn}:{rn + 3}], v{W_RD} offset:{(j + 1) * 1024}")
            lq.issue(("R", t, j + 1))
        elif more1:
            q.wait_for(("W", t + 1), *((("W", t + 2),) if pre_final else ()))
            emit("s_barrier")                    # W(t + 1) of every wave has landed (and nobody still reads the slot W(t + 4) will take)
            if not pre_final:
                emit(f"v_add_u32 v{W_RD}, s{S_NXT}, v{W_WOFF}")
                emit(f"ds_read_b128 v[{rn}:{rn + 3}], v{W_RD} offset:0")
                lq.issue(("R", t + 1, 0))
        fill = [[], [], [], []]
        if dinit:
            c = (j + FN // 2) % FN
            k = j % 2
            if not last:
                fill[0].append(chunk_read((j + 1 + FN // 2) % FN, 1 - k))
            fill[0].append(lambda k=k, c=c: lq.wait_for(("C", k, c)))
            ops_ = []
            for i in range(2):
                for e in range(4):
                    if dinit == "wz":
                        ops_.append(f"v_mad_i32_i24 {accr(i, c, e)}, v{W_D[k] + e}, v{W_RS0 + i}, {accr(i, c, e)}")
                    else:
                        ops_.append(f"v_add_u32 {accr(i, c, e)}, {accr(i, c, e)}, v{W_D[k] + e}")
            for n, text in enumerate(ops_):
                fill[n // 2].append(lambda text=text: emit(text))
        for n, fs in vm.items():
            if n == j:
                fill[3] = fill[3] + fs
        lq.wait_for(("R", t, j))
        w4_unpack_mfma(j, r, set_, fill)
        if first:
            pass
    w4_unpack_mfma.zero = False
    if more3:
        emit(f"s_add_u32 s{S_WBASE}, s{S_WBASE}, {BK // 2}")
        emit(f"s_addc_u32 s{S_WBASE + 1}, s{S_WBASE + 1}, 0")
    w4_rotate()


def w4_prologue(q, lq, nw):
    emit("; ==== W4 prologue")
    if SCALAR_GRID:
        emit(f"s_load_dword s{S_SO}, %[soptr], 0x0")
        emit(f"s_load_dword s{S_OO}, %[ooptr], 0x0")
    emit(f"s_mov_b32 s{S_M4}, 0x0f0f0f0f")
    emit(f"s_mov_b64 s[{S_ABASE}:{S_ABASE + 1}], %[aptr]")
    emit(f"s_mov_b64 s[{S_WBASE}:{S_WBASE + 1}], %[wptr]")
    emit(f"s_lshl_b32 s{S_WK[0]}, %[wave], 10")
    for i in range(1, PIECES):
        emit(f"s_add_u32 s{S_WK[i]}, s{S_WK[0]}, {i * NW * 1024}")
    emit(f"s_mov_b32 s{S_CUR}, 0")
    emit(f"s_mov_b32 s{S_NXT}, {W_BYTES}")
    emit(f"s_mov_b32 s{S_DMA}, {3 * W_BYTES}")
    emit(f"global_load_dword v{W_RS0}, %[rsofs0], %[rsptr]")
    q.issue("P")
    emit(f"global_load_dword v{W_RS1}, %[rsofs1], %[rsptr]")
    q.issue("P")
    emit(f"s_mov_b64 s[{S_EXEC}:{S_EXEC + 1}], exec")
    emit(f"v_cmp_gt_u32 vcc, {BN}, %[tid]")
    emit("s_and_b64 exec, exec, vcc")
    emit(f"v_lshlrev_b32 v{W_TMP}, 2, %[tid]")
    P0 = W_D[0]
    for k, ptr in enumerate(("alpha", "bias", "wzp", "ct")):
        emit(f"global_load_dword v{P0 + k}, v{W_TMP}, %[{ptr}]")
        q.issue("P")
    emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    for t, slot in ((0, 0), (1, W_BYTES), (2, 2 * W_BYTES)):
        emit(f"s_mov_b32 s{S_TMP}, {slot}")
        for i in range(nw):
            w4_piece(q, t, i, S_TMP)()
        emit(f"s_add_u32 s{S_WBASE}, s{S_WBASE}, {BK // 2}")
        emit(f"s_addc_u32 s{S_WBASE + 1}, s{S_WBASE + 1}, 0")
        if t == 0:
            for typ in range(2):
                for i in range(2):
                    w4_a_load(q, 0, typ, i, 0, 0)()
    # per-lane constants: read offset = frow * 64 + ((kq ^ g(frow)) << 4), g = (4 - (frow >> 2)) & 3;  parameter address = PAR + kq * 16
    emit(f"v_and_b32 v{W_TMP}, 63, %[tid]")
    emit(f"v_and_b32 v{W_WOFF}, 15, v{W_TMP}")                     # frow
    emit(f"v_lshrrev_b32 v{W_PAR}, 4, v{W_TMP}")                   # kq
    emit(f"v_lshrrev_b32 v{W_RD}, 2, v{W_WOFF}")                   # frow >> 2
    emit(f"v_sub_u32 v{W_RD}, 4, v{W_RD}")
    emit(f"v_and_b32 v{W_RD}, 3, v{W_RD}")                         # g
    emit(f"v_xor_b32 v{W_RD}, v{W_RD}, v{W_PAR}")                  # kq ^ g
    emit(f"v_lshlrev_b32 v{W_RD}, 4, v{W_RD}")
    emit(f"v_lshl_add_u32 v{W_WOFF}, v{W_WOFF}, 6, v{W_RD}")
    emit(f"v_lshlrev_b32 v{W_PAR}, 4, v{W_PAR}")
    emit(f"v_add_u32 v{W_PAR}, {PAR}, v{W_PAR}")
    q.wait_for("P")
    if SCALAR_GRID:
        emit("s_waitcnt lgkmcnt(0)")
        vs, v1, v2, v0, v3, v4 = 108, 109, 110, 111, 112, 113
        emit(f"v_mov_b32 v{vs}, s{S_SO}")
        emit(f"v_div_scale_f32 v{v1}, vcc, v{vs}, v{vs}, 1.0")
        emit(f"v_rcp_f32 v{v2}, v{v1}")
        emit("s_nop 0")
        emit(f"v_fma_f32 v{v0}, -v{v1}, v{v2}, 1.0")
        emit(f"v_fma_f32 v{v2}, v{v0}, v{v2}, v{v2}")
        emit(f"v_div_scale_f32 v{v0}, vcc, 1.0, v{vs}, 1.0")
        emit(f"v_mul_f32 v{v3}, v{v0}, v{v2}")
        emit(f"v_fma_f32 v{v4}, -v{v1}, v{v3}, v{v0}")
        emit(f"v_fma_f32 v{v3}, v{v4}, v{v2}, v{v3}")
        emit(f"v_fma_f32 v{v0}, -v{v1}, v{v3}, v{v0}")
        emit(f"v_div_fmas_f32 v{v0}, v{v0}, v{v2}, v{v3}")
        emit(f"v_div_fixup_f32 v{v0}, v{v0}, v{vs}, 1.0")
        emit("s_nop 0")
        emit(f"v_readfirstlane_b32 s{S_ISO}, v{v0}")
    emit("s_bitcmp1_b32 %[flags], 1")
    l = label("rs")
    emit(f"s_cbranch_scc1 {l}")
    emit(f"v_mov_b32 v{W_RS0}, 0")
    emit(f"v_mov_b32 v{W_RS1}, 0")
    emit(f"{l}:")
    emit(f"v_cmp_gt_u32 vcc, {BN}, %[tid]")
    emit("s_and_b64 exec, exec, vcc")
    emit("s_bitcmp1_b32 %[flags], 0")
    l = label("nb")
    emit(f"s_cbranch_scc1 {l}")
    emit(f"v_mov_b32 v{P0 + 1}, 0")
    emit(f"{l}:")
    inv, oo = (f"s{S_ISO}", f"s{S_OO}") if SCALAR_GRID else ("%[invc]", "%[ooc]")
    emit(f"v_mul_f32 v{P0}, {inv}, v{P0}")
    emit(f"v_mul_f32 v{P0 + 1}, {inv}, v{P0 + 1}")
    emit(f"v_add_f32 v{P0 + 1}, {oo}, v{P0 + 1}")
    emit(f"v_sub_u32 v{P0 + 2}, 0, v{P0 + 2}")
    emit(f"v_lshlrev_b32 v{W_TMP}, 2, %[tid]")
    emit(f"v_add_u32 v{W_TMP}, {PAR}, v{W_TMP}")
    for k in range(4):
        emit(f"ds_write_b32 v{W_TMP}, v{P0 + k} offset:{k * 4 * BN}")
    emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    q.wait_for(("W", 0))
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    emit(f"v_add_u32 v{W_RD}, s{S_CUR}, v{W_WOFF}")
    emit(f"ds_read_b128 v[{W_R[0]}:{W_R[0] + 3}], v{W_RD} offset:0")
    lq.issue(("R", 0, 0))


def w4_final_block(q, lq):
    """stages KT-2 (A set 0, slot S_CUR) and KT-1 (A in the third set, slot S_NXT) column by column: per column two raw quads -> 8 MFMAs;
    the u8 conversion of column j - 1 and the group stores ride between them (as final_block does for the int8 kernels)."""
    NG = (FN + 3) // 4
    EA = [100, 108]
    RDB, PARV, GOG = "%[sw0]", "%[sw1]", "%[sw2]"
    emit("; ==== W4 final block")
    q.wait_for(("A", 6), ("A", 7))
    assert q.q == [], q.q
    # the second read address, the parameter address (v100 is about to hold parameters): into the dead LDS-DMA source operands
    emit(f"v_mov_b32 {PARV}, v{W_PAR}")
    emit(f"v_add_u32 {RDB}, s{S_NXT}, v{W_WOFF}")
    emit(f"v_add_u32 v{W_RD}, s{S_CUR}, v{W_WOFF}")
    emit(f"ds_read_b128 v[{W_R[0]}:{W_R[0] + 3}], v{W_RD} offset:0")
    lq.issue(("R", 6, 0))
    emit(f"ds_read_b128 v[{W_R[1]}:{W_R[1] + 3}], {RDB} offset:0")
    lq.issue(("R", 7, 0))
    rem = FN - 4 * (NG - 1)
    fill = []

    def F(minidx, fn):
        fill.append((minidx, fn))

    def E(minidx, text):
        F(minidx, lambda: emit(text))
    T = W_WOFF                    # v99: free now
    for text in (
            f"v_and_b32 v{T}, 63, %[tid]",
            f"v_and_b32 {GOG}, 15, v{T}",
            f"v_lshrrev_b32 v{T}, 4, v{T}",
            f"v_cmp_gt_i32_e64 s[{S_FM}:{S_FM + 1}], %[mrem], {GOG}",
            f"v_add_u32 v{W_TMP}, 16, {GOG}",
            f"v_cmp_gt_i32_e64 s[{S_FM + 2}:{S_FM + 3}], %[mrem], v{W_TMP}",
            f"v_cmp_gt_u32_e64 s[{S_FM + 4}:{S_FM + 5}], {rem}, v{T}",
            f"s_and_b64 s[{S_FM + 6}:{S_FM + 7}], s[{S_FM + 2}:{S_FM + 3}], s[{S_FM + 4}:{S_FM + 5}]",
            f"s_and_b64 s[{S_FM + 4}:{S_FM + 5}], s[{S_FM}:{S_FM + 1}], s[{S_FM + 4}:{S_FM + 5}]",
            f"v_mul_lo_u32 {GOG}, {GOG}, %[ldn]",
            f"v_lshl_add_u32 {GOG}, v{T}, 4, {GOG}",
            f"s_lshl_b32 s{S_TMP2}, %[ldn], 4",
            f"s_mov_b64 s[{S_OB1}:{S_OB1 + 1}], %[outw]",
            f"s_add_u32 s{S_OB1}, s{S_OB1}, s{S_TMP2}",
            f"s_addc_u32 s{S_OB1 + 1}, s{S_OB1 + 1}, 0"):
        E(0, text)

    def params(j):
        st = j & 1
        def fn0():
            emit(f"ds_read_b128 v[{EA[st]}:{EA[st] + 3}], {PARV} offset:{j * 64}")
            lq.issue(("P", j))
        def fn1():
            emit(f"ds_read_b128 v[{EA[st] + 4}:{EA[st] + 7}], {PARV} offset:{4 * BN + j * 64}")
            lq.issue(("P", j))
        return [fn0, fn1]

    def tbase(g, i):
        return (2 * (4 * g) + i) * 4

    def store_group(minidx, g):
        last = g == NG - 1 and rem != 4
        for i in range(2):
            t = tbase(g, i)
            E(minidx, "s_nop 1")
            E(minidx, f"v_permlane32_swap_b32 v{t}, v{t + 2}")
            E(minidx, f"v_permlane32_swap_b32 v{t + 1}, v{t + 3}")
            E(minidx, "s_nop 1")
            E(minidx, f"v_permlane16_swap_b32 v{t}, v{t + 1}")
            E(minidx, f"v_permlane16_swap_b32 v{t + 2}, v{t + 3}")
            E(minidx, "s_nop 1")
            for k in range(4):
                E(minidx, f"v_xor_b32 v{t + k}, %[xorv], v{t + k}")
            m = S_FM + 2 * i + (4 if last else 0)
            base = "%[outw]" if i == 0 else f"s[{S_OB1}:{S_OB1 + 1}]"
            def st(base=base, t=t, g=g, m=m):
                emit(f"s_mov_b64 exec, s[{m}:{m + 1}]")       # (one filler: no MFMA under the store's exec mask)
                emit(f"global_store_dwordx4 {GOG}, v[{t}:{t + 3}], {base} offset:{g * 64}")
                q.issue(("S", g))
                emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
            F(minidx, st)

    def convert(minidx, j):
        st = j & 1
        g, k = j // 4, j % 4
        F(minidx, lambda: lq.wait_for(("P", j)))
        for i in range(2):
            for e in range(4):
                E(minidx, f"v_cvt_f32_i32 {accr(i, j, e)}, {accr(i, j, e)}")
        for i in range(2):
            for e in range(4):
                E(minidx, f"v_fma_f32 {accr(i, j, e)}, {accr(i, j, e)}, v{EA[st] + e}, v{EA[st] + 4 + e}")
        for i in range(2):
            d = tbase(g, i) + k
            for e in range(4):
                E(minidx, f"v_cvt_pk_u8_f32 v{d}, {accr(i, j, e)}, {e}, " + (f"v{d}" if e else "0"))
        if k == 3 or j == FN - 1:
            store_group(minidx, g)

    for j in range(FN):
        base = 8 * j
        for fn in params(j):
            F(base, fn)
        if j >= 1:
            convert(base + 2, j - 1)
    convert(8 * FN + 1000, FN - 1)

    CAP = 4
    fi = [0]
    n = [0]

    def drain(cap):
        k = 0
        out_ = []
        while fi[0] < len(fill) and fill[fi[0]][0] <= n[0] and k < cap:
            out_.append(fill[fi[0]][1])
            fi[0] += 1
            k += 1
        return out_

    for j in range(FN):
        for half, (tt, aset, r, rd) in enumerate(((6, 0, W_R[0], f"v{W_RD}"), (7, 2, W_R[1], RDB))):
            lq.wait_for(("R", tt, j))
            fl = []
            for m_ in range(4):
                fl.append(drain(CAP))
                n[0] += 1
            if j + 1 < FN:                        # the quad is free once its high-nibble MFMAs have issued: re-read it for column j + 1
                def rr(tt=tt, r=r, rd=rd, j=j):
                    emit(f"ds_read_b128 v[{r}:{r + 3}], {rd} offset:{(j + 1) * 1024}")
                    lq.issue(("R", tt, j + 1))
                fl[3] = [rr] + fl[3]
            w4_unpack_mfma(j, r, aset, fl)
    while fi[0] < len(fill) and fill[fi[0]][0] < 8 * FN:
        fill[fi[0]][1]()
        fi[0] += 1
    emit("s_nop 15")
    emit("s_nop 3")
    while fi[0] < len(fill):
        fill[fi[0]][1]()
        fi[0] += 1
    emit("s_waitcnt vmcnt(0)")


def program_w4(nw):
    q, lq = Queue(), LQueue()
    w4_prologue(q, lq, nw)
    w4_stage(q, lq, 0, nw, first=DINIT)
    w4_stage(q, lq, 1, nw, dinit="wz")
    emit(f"s_sub_u32 s{S_CNT}, %[kt], 6")
    emit(f"s_lshr_b32 s{S_CNT}, s{S_CNT}, 1")
    lend, lloop = label("tail"), label("loop")
    emit(f"s_cmp_eq_u32 s{S_CNT}, 0")
    emit(f"s_cbranch_scc1 {lend}")
    emit(f"{lloop}:")
    before, lbefore = list(q.q), list(lq.q)
    w4_stage(q, lq, 2, nw, sym="T")
    w4_stage(q, lq, 3, nw, sym="T+1")
    shift = lambda x, d: (tuple(x[:1]) + (x[1] + d,) + tuple(x[2:]) if isinstance(x, tuple) and isinstance(x[1], int) else x)   # noqa: E731
    assert [shift(x, -2) for x in q.q] == before, (before, q.q)
    assert [shift(x, -2) for x in lq.q] == lbefore, (lbefore, lq.q)
    emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    emit(f"s_cmp_lg_u32 s{S_CNT}, 0")
    emit(f"s_cbranch_scc1 {lloop}")
    emit(f"{lend}:")
    KT = 8
    q.q = [shift(x, 2) for x in before]
    lq.q = [shift(x, 2) for x in lbefore]
    w4_stage(q, lq, 4, nw, kt=KT, sym="KT-4", dinit="ct")
    w4_stage(q, lq, 5, nw, kt=KT, sym="KT-3", pre_final=True)
    w4_final_block(q, lq)


def generate_w4():
    emit("; generated by tools/gen_fr_asm.py -- do not edit")
    assert DINIT
    if FN % NW == 0:
        program_w4(FN // NW)
        return
    full = FN - (PIECES - 1) * NW             # waves that own PIECES pieces
    l2, lend = label("w1"), label("done")
    emit(f"s_cmp_lt_u32 %[wave], {full}")
    emit(f"s_cbranch_scc0 {l2}")
    program_w4(PIECES)
    emit(f"s_branch {lend}")
    emit(f"{l2}:")
    program_w4(PIECES - 1)
    emit(f"{lend}:")


def generate(stamp=False):
    emit("; generated by tools/gen_fr_asm.py -- do not edit")
    if W4X:
        if FN % NW == 0:
            program(FN // NW, stamp)
            return
        full = FN - (PIECES - 1) * NW
        l2, lend = label("w1"), label("done")
        emit(f"s_cmp_lt_u32 %[wave], {full}")
        emit(f"s_cbranch_scc0 {l2}")
        program(PIECES, stamp)
        emit(f"s_branch {lend}")
        emit(f"{l2}:")
        program(PIECES - 1, stamp)
        emit(f"{lend}:")
        return
    if (BN // 8) % NW == 0:          # every wave owns the same number of W pieces: one program
        program(BN // 8 // NW, stamp)
        return
    assert BN == 176 and NW == 8
    l6, lend = label("w2"), label("done")
    emit("s_cmp_lt_u32 %[wave], 6")
    emit(f"s_cbranch_scc0 {l6}")
    program(3, stamp)
    emit(f"s_branch {lend}")
    emit(f"{l6}:")
    program(2, stamp)
    emit(f"{lend}:")


def main(path=None, variant="fr"):
    configure(variant)
    del out[:]
    _uid[0] = 0
    stamp = bool(os.environ.get("MQ_FR_STAMP")) and variant == "fr"
    global PROBE
    PROBE = variant == "fr" and TAIL
    if W4:
        generate_w4()
    else:
        generate(stamp)
    here = os.path.dirname(os.path.abspath(__file__))
    path = path or os.path.join(here, "..", "mobilequant_amd", "csrc", FILE)
    # VGPRs between the accumulators and the temporaries are left to hipcc for the asm statement's vector operands
    vregs = [f'"v{r}"' for r in list(range(0, 88 if EPI == "gate" else 8 * FN)) + list(range(V_T if (8 * FN > 78 or EPI == "gate") else 78, 128))]
    aregs = [f'"a{r}"' for r in range(0, 8 * FN + 32 + (8 if TAIL else 0))]
    if W4:
        vregs = [f'"v{r}"' for r in list(range(0, 8 * FN)) + list(range(98, 128))]
        aregs = [f'"a{r}"' for r in range(0, 48)]
    sregs = [f'"s{r}"' for r in range(S0, S_MASK + 16)] + [f'"s{r}"' for r in (S_SO, S_OO, S_ISO)] + (['"s100"', '"s101"'] if (PRO_SPLIT or PSTAMP) else []) + (['"s99"'] if (W4 or W4X) else []) + \
        (['"m0"'] if EPI == "gate" else [])
    with open(path, "w") as f:
        f.write("// Generated by tools/gen_fr_asm.py -- do not edit (see that file for the register map, the LDS map and the schedule).\n")
        f.write(f"#define {PREFIX}_ASM_STAMP {1 if stamp else 0}\n")
        f.write(f"#define {PREFIX}_ASM_PROBE {1 if PROBE and not stamp else 0}\n")
        f.write(f"#define {PREFIX}_LDS_BYTES {LDS_BYTES}\n")
        f.write(f"#define {PREFIX}_ASM_BODY \\\n")
        for line in out:
            f.write('  "%s\\n\\t" \\\n' % line.replace('"', '\\"'))
        f.write('  ""\n')
        f.write(f"#define {PREFIX}_ASM_CLOBBERS " + ", ".join(vregs + aregs + sregs + ['"vcc"', '"scc"', '"memory"']) + "\n")
    print("wrote", os.path.normpath(path), len(out), "instructions/labels")


if __name__ == "__main__":
    import sys
    for v in (sys.argv[1:] or [k for k in VARIANTS if k not in EXPERIMENTAL]):
        main(variant=v)
