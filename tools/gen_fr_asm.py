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

BK, BM, BN = 128, 256, 176
W_BYTES = BN * BK                 # 22528
RING = 4
PAR = RING * W_BYTES              # 90112: alpha' | bias' | -w_zp | col_term
STG = PAR + 16 * BN               # 92928
ROWP = BN                         # staging pitch (u8): 176 B rows, writes <= 2-way conflicted
STG_WAVE = 2 * 16 * ROWP          # 5632
LDS_BYTES = STG + 8 * STG_WAVE    # 137984
FN = BN // 16

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
S_WK0, S_WK1, S_WK2 = 65, 66, 67  # wave*1024 + i*8192: LDS offset of this wave's piece i inside a slot
S_CNT = 68                        # steady-state pairs left
S_TMP, S_TMP2 = 69, 70
S_EXEC = 72                       # pair
S_TS = 74                         # 74..81: four s_memtime stamps (stamp builds)
S_MR = 82
S_RT = 84                         # 84..87: s_memrealtime (constant 100 MHz) at kernel start / end (stamp builds)

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
    b = 44 * set_ + 4 * j
    return f"a[{b}:{b + 3}]"


def areg(set_, ks, i):
    b = 88 + 16 * set_ + 8 * ks + 4 * i
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
        emit(f"s_add_u32 m0, s{slot_sgpr}, s{S_WK0 + i}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 %[sw{i}], s[{S_WBASE}:{S_WBASE + 1}]")
        q.issue(("W", t))
    emit(f"s_add_u32 s{S_WBASE}, s{S_WBASE}, {BK}")
    emit(f"s_addc_u32 s{S_WBASE + 1}, s{S_WBASE + 1}, 0")


def w_piece(q, t, i, slot_sgpr):
    def f():
        emit(f"s_add_u32 m0, s{slot_sgpr}, s{S_WK0 + i}")
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


def kstep(cur, aset, ks, rd_slot, rd_woff, nxt, vmem, read=True, mfma_first=0):
    """22 MFMAs on W register set `cur` x A(aset, ks); 11 ds_reads of slot `rd_slot` (+ per-lane offset register rd_woff) into W
    register set `nxt`, one after each of the first MFMAs; `vmem`: callables (VMEM issues) spread behind the later MFMAs."""
    if read:
        emit(f"v_add_u32 v{V_RD}, s{rd_slot}, v{rd_woff}")
    places = {}
    for n, (kind, f) in enumerate(vmem):      # A loads early (their registers are free), DMA pieces in the second half
        na = sum(1 for k, _ in vmem[:n] if k == kind)
        places.setdefault(1 + 2 * na if kind == "a" else 12 + 3 * na, []).append(f)
    m = 0
    for j in range(FN):
        for i in range(2):
            if not NO_MFMA:
                emit(f"v_mfma_i32_16x16x64_i8 {acc(i, j)}, {wreg(cur, j)}, {areg(aset, ks, i)}, {acc(i, j)}")
            if read and m < FN and not NO_READ:
                emit(f"ds_read_b128 {wreg(nxt, m)}, v{V_RD} offset:{m * 16 * BK}")
            for f in places.get(m, []):
                f()
            m += 1
    if read:
        emit("s_waitcnt lgkmcnt(0)")


def rotate():
    emit(f"s_mov_b32 s{S_CUR}, s{S_NXT}")
    for s in (S_NXT, S_DMA):
        emit(f"s_add_u32 s{s}, s{s}, {W_BYTES}")
        emit(f"s_cmp_eq_u32 s{s}, {RING * W_BYTES}")
        emit(f"s_cselect_b32 s{s}, 0, s{s}")
    emit(f"s_add_u32 s{S_ABASE}, s{S_ABASE}, {2 * 1024}")
    emit(f"s_addc_u32 s{S_ABASE + 1}, s{S_ABASE + 1}, 0")


def stage(q, t, nw, kt=None, sym=None):
    """One K = 128 stage.  t: stage number used for the queue tags; kt: total stages when the tail conditions apply (None = steady
    state: everything is issued)."""
    set_ = t & 1
    more1 = kt is None or t + 1 < kt
    more2 = kt is None or t + 2 < kt
    more3 = kt is None or t + 3 < kt
    emit(f"; ---- stage {sym or t}: A set {set_}")
    q.wait_for(("A", t, 0))
    v0 = []
    if more1 and not NO_A:      # S_ABASE = activation pointer of stage t+1
        v0 += [("a", a_load(q, t + 1, 1, 0, 1 - set_, 1024)), ("a", a_load(q, t + 1, 1, 1, 1 - set_, 1024))]
    if more3 and not NO_W:
        v0 += [("w", w_piece(q, t + 3, i, S_DMA)) for i in range(nw)]
    kstep(0, set_, 0, S_CUR, V_WOFF1, 1, v0)
    if more3:
        emit(f"s_add_u32 s{S_WBASE}, s{S_WBASE}, {BK}")
        emit(f"s_addc_u32 s{S_WBASE + 1}, s{S_WBASE + 1}, 0")
    q.wait_for(("A", t, 1), ("W", t + 1))
    emit("s_barrier")
    v1 = []
    if more2 and not NO_A:
        v1 += [("a", a_load(q, t + 2, 0, 0, set_, 2048)), ("a", a_load(q, t + 2, 0, 1, set_, 2048))]
    kstep(1, set_, 1, S_NXT, V_WOFF0, 0, v1, read=more1)
    rotate()


def prologue(q, nw, stamp):
    emit("; ==== prologue")
    if stamp:
        emit(f"s_memtime s[{S_TS}:{S_TS + 1}]")
        emit(f"s_memrealtime s[{S_RT}:{S_RT + 1}]")
        emit("s_waitcnt lgkmcnt(0)")
    # loop state
    emit(f"s_mov_b64 s[{S_ABASE}:{S_ABASE + 1}], %[aptr]")
    emit(f"s_mov_b64 s[{S_WBASE}:{S_WBASE + 1}], %[wptr]")
    emit(f"s_lshl_b32 s{S_WK0}, %[wave], 10")
    emit(f"s_add_u32 s{S_WK1}, s{S_WK0}, 8192")
    emit(f"s_add_u32 s{S_WK2}, s{S_WK0}, 16384")
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
    for t, slot in ((0, 0), (1, W_BYTES), (2, 2 * W_BYTES)):
        emit(f"s_mov_b32 s{S_TMP}, {slot}")
        issue_w(q, t, nw, S_TMP)
        if t == 0:
            for ks in range(2):
                for i in range(2):
                    a_load(q, 0, ks, i, 0, 1024 * ks)()
            emit(f"s_add_u32 s{S_ABASE}, s{S_ABASE}, {2 * 1024}")       # -> stage 1
            emit(f"s_addc_u32 s{S_ABASE + 1}, s{S_ABASE + 1}, 0")
        if t == 1:
            a_load(q, 1, 0, 0, 1, 0)()
            a_load(q, 1, 0, 1, 1, 0)()
    # per-lane constants while the loads fly
    emit(f"v_and_b32 v{V_TMP}, 63, %[tid]")                                # lane
    emit(f"v_and_b32 v{V_WOFF0}, 15, v{V_TMP}")                            # frow
    emit(f"v_lshrrev_b32 v{V_PAR}, 4, v{V_TMP}")                           # kq
    emit(f"v_and_b32 v{V_STW}, 7, v{V_TMP}")                               # lane & 7
    emit(f"v_xor_b32 v{V_STW}, v{V_STW}, v{V_PAR}")                        # kq ^ (lane & 7)
    emit(f"v_lshlrev_b32 v{V_STW}, 4, v{V_STW}")
    emit(f"v_lshl_add_u32 v{V_WOFF0}, v{V_WOFF0}, 7, v{V_STW}")            # frow*128 + swizzled chunk
    emit(f"v_xor_b32 v{V_WOFF1}, 64, v{V_WOFF0}")
    # staging write address: STG + wave*STG_WAVE + frow*ROWP + kq*4 ; alpha' chunk address: PAR + kq*16
    emit(f"v_and_b32 v{V_STW}, 15, v{V_TMP}")
    emit(f"v_mul_u32_u24 v{V_STW}, {ROWP}, v{V_STW}")
    emit(f"v_lshl_add_u32 v{V_STW}, v{V_PAR}, 2, v{V_STW}")
    emit(f"s_mul_i32 s{S_TMP}, %[wave], {STG_WAVE}")
    emit(f"s_add_u32 s{S_TMP}, s{S_TMP}, {STG}")
    emit(f"v_add_u32 v{V_STW}, s{S_TMP}, v{V_STW}")
    emit(f"v_lshlrev_b32 v{V_PAR}, 4, v{V_PAR}")
    emit(f"v_add_u32 v{V_PAR}, {PAR}, v{V_PAR}")
    # (3) parameters: park alpha' = alpha/so, bias' = bias/so + oo, -w_zp, col_term in LDS (same expressions as the C++ prologue
    # of the other variants: one multiply, one multiply + one add, no contraction)
    q.wait_for("P")
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
    emit(f"v_mul_f32 v{V_P0}, %[inv_so], v{V_P0}")
    emit(f"v_mul_f32 v{V_P0 + 1}, %[inv_so], v{V_P0 + 1}")
    emit(f"v_add_f32 v{V_P0 + 1}, %[oo], v{V_P0 + 1}")
    emit(f"v_sub_u32 v{V_P0 + 2}, 0, v{V_P0 + 2}")
    emit(f"v_lshlrev_b32 v{V_TMP}, 2, %[tid]")
    emit(f"v_add_u32 v{V_TMP}, {PAR}, v{V_TMP}")
    for k in range(4):
        emit(f"ds_write_b32 v{V_TMP}, v{V_P0 + k} offset:{k * 4 * BN}")
    emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    # (4) accumulators = col_term[n] - w_zp[n] * row_sum[m] while the first stage is in flight
    for j in range(FN):
        emit(f"ds_read_b128 v[{V_E}:{V_E + 3}], v{V_PAR} offset:{2 * 4 * BN + j * 64}")
        emit(f"ds_read_b128 v[{V_E + 4}:{V_E + 7}], v{V_PAR} offset:{3 * 4 * BN + j * 64}")
        emit("s_waitcnt lgkmcnt(0)")
        for i in range(2):
            for e in range(4):
                emit(f"v_mad_i32_i24 {accr(i, j, e)}, v{V_E + e}, v{V_RS0 + i}, v{V_E + 4 + e}")
    # (5) W(0) of every wave landed -> first fragments
    q.wait_for(("W", 0))
    emit("s_barrier")
    emit(f"v_add_u32 v{V_RD}, s{S_CUR}, v{V_WOFF0}")
    for j in range(FN):
        emit(f"ds_read_b128 {wreg(0, j)}, v{V_RD} offset:{j * 16 * BK}")
    emit("s_waitcnt lgkmcnt(0)")
    if stamp:
        emit(f"s_memtime s[{S_TS + 2}:{S_TS + 3}]")
        emit("s_waitcnt lgkmcnt(0)")


def epilogue(stamp):
    emit("; ==== epilogue: u8 = cvt_pk_u8(fma(float(acc), alpha', bias')) -> staging tile -> whole-row 16-byte stores")
    if stamp:
        emit(f"s_memtime s[{S_TS + 4}:{S_TS + 5}]")
        emit("s_waitcnt lgkmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 3")
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
            emit(f"v_xor_b32 v{VP[i]}, %[xorv], v{VP[i]}")
            emit(f"ds_write_b32 v{V_STW}, v{VP[i]} offset:{i * 16 * ROWP + j * 16}")
    # chunk c = lane + 64 r (r = 0..2) of the 16 x 11 chunks of a row block: row = c / 11, ch = c % 11
    emit(f"v_and_b32 v{V_TMP}, 63, %[tid]")
    emit(f"s_mul_i32 s{S_TMP}, %[wave], {STG_WAVE}")
    emit(f"s_add_u32 s{S_TMP}, s{S_TMP}, {STG}")
    for r in range(3):
        c, row, ch = 124, 125, 126
        emit(f"v_add_u32 v{c}, {64 * r}, v{V_TMP}")
        emit(f"v_mul_u32_u24 v{row}, 5958, v{c}")                            # floor(c / 11) for c < 192 (5958 = ceil(2^16 / 11))
        emit(f"v_lshrrev_b32 v{row}, 16, v{row}")
        emit(f"v_mul_u32_u24 v{ch}, 11, v{row}")
        emit(f"v_sub_u32 v{ch}, v{c}, v{ch}")
        emit(f"v_lshlrev_b32 v{ch}, 4, v{ch}")                               # ch * 16 bytes
        emit(f"v_mul_u32_u24 v{V_LDSO + r}, {ROWP}, v{row}")
        emit(f"v_add_u32 v{V_LDSO + r}, v{V_LDSO + r}, v{ch}")
        emit(f"v_add_u32 v{V_LDSO + r}, s{S_TMP}, v{V_LDSO + r}")
        emit(f"v_mul_lo_u32 v{V_GOFS + r}, v{row}, %[ldn]")
        emit(f"v_add_u32 v{V_GOFS + r}, v{V_GOFS + r}, v{ch}")
        emit(f"v_mov_b32 v{V_E + r}, v{row}")                                # row index kept for the M bound
    emit("s_waitcnt lgkmcnt(0)")
    for i in range(2):
        for r in range(3):
            b = (i * 3 + r) * 4
            emit(f"ds_read_b128 v[{b}:{b + 3}], v{V_LDSO + r} offset:{i * 16 * ROWP}")
    emit(f"s_lshl_b32 s{S_TMP2}, %[ldn], 4")                                 # 16 rows further down
    emit("s_waitcnt lgkmcnt(0)")
    for i in range(2):
        for r in range(3):
            b = (i * 3 + r) * 4
            # rows valid for this wave: %[mrem] (may be <= 0 or > 32); lane active if i*16 + row < mrem (and c < 176 for r = 2)
            emit(f"v_add_u32 v{V_TMP}, {16 * i}, v{V_E + r}")
            emit(f"v_cmp_gt_i32 vcc, %[mrem], v{V_TMP}")
            if r == 2:
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
        emit(f"s_mov_b64 exec, s[{S_EXEC}:{S_EXEC + 1}]")
    emit("s_waitcnt vmcnt(0)")


def program(nw, stamp):
    q = Queue()
    prologue(q, nw, stamp)
    stage(q, 0, nw)
    stage(q, 1, nw)
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
    for t in range(4, 8):
        stage(q, t, nw, kt=KT, sym=f"KT-{KT - t}")
    assert q.q == [], q.q
    epilogue(stamp)
    return q


def generate(stamp=False):
    emit("; generated by tools/gen_fr_asm.py -- do not edit")
    l6, lend = label("w2"), label("done")
    emit("s_cmp_lt_u32 %[wave], 6")
    emit(f"s_cbranch_scc0 {l6}")
    program(3, stamp)
    emit(f"s_branch {lend}")
    emit(f"{l6}:")
    program(2, stamp)
    emit(f"{lend}:")


def main(path=None):
    stamp = bool(os.environ.get("MQ_FR_STAMP"))
    generate(stamp)
    here = os.path.dirname(os.path.abspath(__file__))
    path = path or os.path.join(here, "..", "mobilequant_amd", "csrc", "mq_gemm_fr_asm.inc")
    vregs = [f'"v{r}"' for r in list(range(0, 88)) + list(range(V_T, 128))]
    aregs = [f'"a{r}"' for r in range(0, 120)]
    sregs = [f'"s{r}"' for r in range(S0, S_RT + 4)]
    with open(path, "w") as f:
        f.write("// Generated by tools/gen_fr_asm.py -- do not edit (see that file for the register map, the LDS map and the schedule).\n")
        f.write(f"#define MQ_FR_ASM_STAMP {1 if stamp else 0}\n")
        f.write(f"#define MQ_FR_LDS_BYTES {LDS_BYTES}\n")
        f.write("#define MQ_FR_ASM_BODY \\\n")
        for line in out:
            f.write('  "%s\\n\\t" \\\n' % line.replace('"', '\\"'))
        f.write('  ""\n')
        f.write("#define MQ_FR_ASM_CLOBBERS " + ", ".join(vregs + aregs + sregs + ['"vcc"', '"scc"', '"memory"']) + "\n")
    print("wrote", os.path.normpath(path), len(out), "instructions/labels")


if __name__ == "__main__":
    main()
