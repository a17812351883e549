#!/usr/bin/env python3
"""Generates mobilequant_amd/csrc/mq_gemm_pp_asm.inc: the hand-scheduled gfx950 main loop of the 256x176 ping-pong
int8 GEMM on fragment-blocked activations (GEMM variant "t256x176_w8x1_pp_asm", entry point mq_w8a8_linear_tiled).

Why generated ISA: the A rows of the 8x1 wave layout are private to a wave, so staging them through the LDS only costs
LDS-DMA writes and fragment reads (measured what-if: -10 % kernel time).  With the activations
stored fragment-blocked by mq_quantize_tiled (1-KiB blocks of 16 rows x 64 k = one MFMA operand), a fragment is ONE
fully coalesced global_load_dwordx4 straight into registers.  Keeping two register sets of A fragments next to 88
accumulator and 44 W-fragment registers is beyond what hipcc allocates without spilling (attempts recorded in
DESIGN.md), so the registers are fixed by hand:

  AGPR  a[0:87]     accumulators, acc[i][j] = a[(2j+i)*4 : +3]        (i = A fragment 0/1, j = W fragment 0..10)
        a[88:103]   A fragments of even stages  [ks0 i0][ks0 i1][ks1 i0][ks1 i1]
        a[104:119]  A fragments of odd stages
  VGPR  v[76:119]   W fragments of the current k-step (ds_read_b128 x 11)
        v120        LDS read address;  v[121:123] W LDS-DMA source offsets;  v[124:125] A load offsets
  SGPR  s[84:94]    loop state (below)

Schedule = the one of the C++ ping-pong loop (mq_gemm.hip): phases E/O, one s_barrier each, group 0 (waves 0-3) and
group 1 (waves 4-7) offset by one phase, W in a ring of three LDS buffers filled by LDS-DMA two stages ahead, counted
vmcnt waits; A(t+2) is requested where the C++ loop issued the A pieces (same count, so the waits carry over).
K % 256 == 0 (the loop body covers two K = 128 stages so the A register sets are static).
(Row-major activations were tried first: a fragment is then 16 rows x 64 B = 16 half cache lines per load, 4 us slower.)

Run:  python tools/gen_pp_asm.py   (writes the .inc next to mq_gemm.hip; the file is committed)."""
import os

BK = 128
BM, BN = 256, 176
A_BYTES = BM * BK
W_BYTES = BN * BK
W_BASE = 2 * A_BYTES                      # the A LDS buffers stay allocated (shared layout with the C++ variants)
FN = BN // 16                             # 11 W fragments
WF0 = 76
V_RD, V_SW, V_A = 120, 121, 124
# SGPRs
S_T, S_LAST, S_MORE, S_TAIL, S_WAVEK, S_K2 = 84, 85, 86, 87, 88, 89
S_WCUR, S_WNXT, S_WPRV, S_TMP, S_TMP2 = 90, 91, 92, 93, 94

out = []
uid = [0]


def emit(s):
    out.append(s)


def label(prefix):
    uid[0] += 1
    return f".Lmq_{prefix}_{uid[0]}%="       # %= : unique per asm instance


def acc(i, j):
    b = (2 * j + i) * 4
    return f"a[{b}:{b + 3}]"


def xa(set_, ks, i):
    b = 88 + 16 * set_ + 8 * ks + 4 * i
    return f"a[{b}:{b + 3}]"


def wf(j):
    return f"v[{WF0 + 4 * j}:{WF0 + 4 * j + 3}]"


def mfma_phase(set_, ks):
    emit("s_setprio 1")
    for j in range(FN):
        for i in range(2):
            emit(f"v_mfma_i32_16x16x64_i8 {acc(i, j)}, {wf(j)}, {xa(set_, ks, i)}, {acc(i, j)}")
    emit("s_setprio 0")


def read_w(ring_sgpr, ks):
    emit(f"v_add_u32 v{V_RD}, s{ring_sgpr}, %[woff{ks}]")
    for j in range(FN):
        emit(f"ds_read_b128 {wf(j)}, v{V_RD} offset:{j * 16 * BK}")


def issue_w(ring_sgpr):
    """W(t+2) -> ring buffer `ring_sgpr`: this wave's pieces wave, wave+8 and (waves 0-5) wave+16; source k = s_k2."""
    skip_tail = label("notail")
    for i in range(3):
        if i == 2:
            emit(f"s_cmp_eq_u32 s{S_TAIL}, 0")
            emit(f"s_cbranch_scc1 {skip_tail}")
        emit(f"v_add_u32 v{V_SW + i}, s{S_K2}, %[sw{i}]")
        emit(f"s_add_u32 s{S_TMP}, s{ring_sgpr}, s{S_WAVEK}")
        emit(f"s_add_u32 m0, s{S_TMP}, {W_BASE + i * 8 * 1024}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 v{V_SW + i}, %[wptr]")
    emit(f"{skip_tail}:")


TILED = True                                       # fragment-blocked A: one fragment = 1 KiB contiguous (mq_quantize_tiled)
NO_A = bool(os.environ.get("MQ_ASM_NO_A"))        # what-if: no A loads after the prologue (results wrong)
A_ORDER = os.environ.get("MQ_ASM_A_ORDER", "ks")     # "ks": (i0,i1) of k-step 0 then k-step 1;  "row": both halves of a row back to back
_prologue = [True]


def load_a(set_):
    """A(kt) with k offset s_k2 -> register set."""
    if NO_A and not _prologue[0]:
        return
    if TILED:      # k offset of a stage = 2 blocks of 1 KiB = 16 x the row-major k offset
        emit(f"s_lshl_b32 s{S_TMP}, s{S_K2}, 4")
        emit(f"v_add_u32 v{V_A}, s{S_TMP}, %[av0]")
        emit(f"v_add_u32 v{V_A + 1}, s{S_TMP}, %[av1]")
    else:
        emit(f"v_add_u32 v{V_A}, s{S_K2}, %[av0]")
        emit(f"v_add_u32 v{V_A + 1}, s{S_K2}, %[av1]")
    order = [(0, 0), (0, 1), (1, 0), (1, 1)] if A_ORDER == "ks" else [(0, 0), (1, 0), (0, 1), (1, 1)]
    for ks, i in order:
        if TILED:
            emit(f"global_load_dwordx4 {xa(set_, ks, i)}, v{V_A + i}, %[aptr]" + (" offset:1024" if ks else ""))
        else:
            emit(f"global_load_dwordx4 {xa(set_, ks, i)}, v{V_A + i}, %[aptr]" + (" offset:64" if ks else ""))


def wait_vm(base):
    """s_waitcnt vmcnt(base + n_w) where n_w = 3 for tail owners (waves 0-5), else 2."""
    lt, ld = label("wt"), label("wd")
    emit(f"s_cmp_eq_u32 s{S_TAIL}, 1")
    emit(f"s_cbranch_scc1 {lt}")
    emit(f"s_waitcnt vmcnt({base + 2})")
    emit(f"s_branch {ld}")
    emit(f"{lt}:")
    emit(f"s_waitcnt vmcnt({base + 3})")
    emit(f"{ld}:")


def if_more(body, else_body=None):
    """Runs body when s_more == 1 (stage t + 2 exists)."""
    ls, le = label("nomore"), label("endmore")
    emit(f"s_cmp_eq_u32 s{S_MORE}, 0")
    emit(f"s_cbranch_scc1 {ls}")
    body()
    if else_body is not None:
        emit(f"s_branch {le}")
    emit(f"{ls}:")
    if else_body is not None:
        else_body()
        emit(f"{le}:")


def rotate():
    emit(f"s_mov_b32 s{S_TMP2}, s{S_WCUR}")
    emit(f"s_mov_b32 s{S_WCUR}, s{S_WNXT}")
    emit(f"s_mov_b32 s{S_WNXT}, s{S_WPRV}")
    emit(f"s_mov_b32 s{S_WPRV}, s{S_TMP2}")
    emit(f"s_add_u32 s{S_K2}, s{S_K2}, {BK}")
    emit(f"s_add_u32 s{S_T}, s{S_T}, 1")


def barrier():
    emit("s_barrier")


def lgkm0():
    emit("s_waitcnt lgkmcnt(0)")


def group0_stage(set_, odd):
    emit(f"; ---- group 0, {'odd' if odd else 'even'} stage: E_(t,0)")
    mfma_phase(set_, 0)
    barrier()
    emit("; O_(t,0): W fragments of k-step 1, then W(t+2) -> ring slot (t+2)%3")
    read_w(S_WCUR, 1)
    if_more(lambda: issue_w(S_WPRV))
    lgkm0()
    barrier()
    emit("; E_(t,1); W(t+1) and A(t+1) of this wave must have landed before the barrier")
    mfma_phase(set_, 1)
    if_more(lambda: wait_vm(0), lambda: emit("s_waitcnt vmcnt(0)"))
    barrier()
    emit("; O_(t,1): W fragments of stage t+1, k-step 0; A(t+2) -> the register set stage t just left")
    if odd:
        if_more(lambda: (read_w(S_WNXT, 0), load_a(set_)))
    else:
        read_w(S_WNXT, 0)
        if_more(lambda: load_a(set_))
    lgkm0()
    barrier()
    rotate()


def group1_stage(set_, odd):
    other = 1 - set_
    emit(f"; ---- group 1, {'odd' if odd else 'even'} stage: E_(t,0): own A(t) landed; read W(t) k-step 0; A(t+1) -> other set")
    if odd:
        if_more(lambda: wait_vm(0), lambda: emit("s_waitcnt vmcnt(0)"))
        read_w(S_WCUR, 0)
        # A(t+1) with k = (t+1)*128 = s_k2 - 128
        def la():
            emit(f"s_sub_u32 s{S_K2}, s{S_K2}, {BK}")
            load_a(other)
            emit(f"s_add_u32 s{S_K2}, s{S_K2}, {BK}")
        if_more(la)
    else:
        l0 = label("t0")
        emit(f"s_cmp_eq_u32 s{S_T}, 0")
        emit(f"s_cbranch_scc1 {l0}")
        wait_vm(0)
        emit(f"{l0}:")
        read_w(S_WCUR, 0)
        l1 = label("t0b")
        emit(f"s_cmp_eq_u32 s{S_T}, 0")
        emit(f"s_cbranch_scc1 {l1}")
        emit(f"s_sub_u32 s{S_K2}, s{S_K2}, {BK}")
        load_a(other)
        emit(f"s_add_u32 s{S_K2}, s{S_K2}, {BK}")
        emit(f"{l1}:")
    lgkm0()
    barrier()
    emit("; O_(t,0)")
    mfma_phase(set_, 0)
    barrier()
    emit("; E_(t,1): W fragments of k-step 1; W(t+2) -> ring; this wave's W(t+1) pieces must have landed")
    read_w(S_WCUR, 1)
    if_more(lambda: issue_w(S_WPRV))
    lgkm0()
    if odd:
        if_more(lambda: wait_vm(4), lambda: emit("s_waitcnt vmcnt(0)"))
    else:
        # allowed in flight: A(t+1) (4 loads, t > 0) + W(t+2) share (more)
        l0, l1 = label("e1t0"), label("e1d")
        emit(f"s_cmp_eq_u32 s{S_T}, 0")
        emit(f"s_cbranch_scc1 {l0}")
        if_more(lambda: wait_vm(4), lambda: emit("s_waitcnt vmcnt(4)"))
        emit(f"s_branch {l1}")
        emit(f"{l0}:")
        if_more(lambda: wait_vm(0), lambda: emit("s_waitcnt vmcnt(0)"))
        emit(f"{l1}:")
    barrier()
    emit("; O_(t,1)")
    mfma_phase(set_, 1)
    barrier()
    rotate()


# ---------------------------------------------------------------------------------------------------------------------
# Stage-granular ping-pong (MQ_ASM_STAGE=1): one phase = one whole K = 128 stage (44 MFMAs | 22 ds_reads + issues), two
# s_barriers per stage instead of four.  Possible because the accumulators sit in AGPRs: a wave keeps the W fragments of
# BOTH k-steps (88 VGPRs).  Phases P_2t: G0 MFMA(t) | G1 READ(t);  P_2t+1: G0 READ(t+1) | G1 MFMA(t).
#   G0 in P_2t+1 issues A(t+2) -> set t&1 and its W(t+3) pieces -> ring slot t%3 (read by everyone by the end of P_2t);
#   G1 in P_2t   issues A(t+1) -> set (t+1)&1 (t > 0) and its W(t+2) pieces -> slot (t+2)%3 (last read in P_2t-2).
#   The C++ prologue issues W(0), W(1) for every wave and W(2) for the waves of group 0.
# Waits (vmcnt is in order; A before W inside a phase so that A can be awaited without its phase's W):
#   G0 end of P_2t   (after MFMA(t)):  own W(t+1) landed   -> allowed in flight: A(t+1)?? no: see wait tables below.
STAGE = bool(os.environ.get("MQ_ASM_STAGE"))
WA0, WB0 = 36, 80              # W fragments of k-step 0 / 1 in stage mode: v[36:79], v[80:123]
V_T0, V_T1, V_T2, V_T3 = 124, 125, 126, 127
S_MORE2 = 95


def wfs(ks, j):
    b = (WA0 if ks == 0 else WB0) + 4 * j
    return f"v[{b}:{b + 3}]"


def s_mfma(set_):
    emit("s_setprio 1")
    for ks in range(2):
        for j in range(FN):
            for i in range(2):
                emit(f"v_mfma_i32_16x16x64_i8 {acc(i, j)}, {wfs(ks, j)}, {xa(set_, ks, i)}, {acc(i, j)}")
    emit("s_setprio 0")


def s_read(ring_sgpr):
    for ks in range(2):
        emit(f"v_add_u32 v{V_T0}, s{ring_sgpr}, %[woff{ks}]")
        for j in range(FN):
            emit(f"ds_read_b128 {wfs(ks, j)}, v{V_T0} offset:{j * 16 * BK}")


def s_issue_w(ring_sgpr, kdelta):
    """this wave's W pieces of the stage whose k offset is s_k2 + kdelta -> ring slot ring_sgpr"""
    skip_tail = label("notail")
    emit(f"s_add_u32 s{S_TMP2}, s{S_K2}, {kdelta}") if kdelta else emit(f"s_mov_b32 s{S_TMP2}, s{S_K2}")
    for i in range(3):
        if i == 2:
            emit(f"s_cmp_eq_u32 s{S_TAIL}, 0")
            emit(f"s_cbranch_scc1 {skip_tail}")
        emit(f"v_add_u32 v{V_T1}, s{S_TMP2}, %[sw{i}]")
        emit(f"s_add_u32 s{S_TMP}, s{ring_sgpr}, s{S_WAVEK}")
        emit(f"s_add_u32 m0, s{S_TMP}, {W_BASE + i * 8 * 1024}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 v{V_T1}, %[wptr]")
    emit(f"{skip_tail}:")


def s_load_a(set_, kdelta):
    """A of the stage whose row-major k offset is s_k2 + kdelta (fragment-blocked: x16) -> register set"""
    emit(f"s_add_u32 s{S_TMP}, s{S_K2}, {kdelta}") if kdelta else emit(f"s_mov_b32 s{S_TMP}, s{S_K2}")
    emit(f"s_lshl_b32 s{S_TMP}, s{S_TMP}, 4")
    emit(f"v_add_u32 v{V_T2}, s{S_TMP}, %[av0]")
    emit(f"v_add_u32 v{V_T3}, s{S_TMP}, %[av1]")
    for ks, i in [(0, 0), (0, 1), (1, 0), (1, 1)]:
        emit(f"global_load_dwordx4 {xa(set_, ks, i)}, v{V_T2 + i}, %[aptr]" + (" offset:1024" if ks else ""))


def s_wait(fixed, nw):
    """s_waitcnt vmcnt(fixed + nw * n_w), n_w = 3 for tail owners (waves 0-5) else 2"""
    if nw == 0:
        emit(f"s_waitcnt vmcnt({fixed})")
        return
    lt, ld = label("wt"), label("wd")
    emit(f"s_cmp_eq_u32 s{S_TAIL}, 1")
    emit(f"s_cbranch_scc1 {lt}")
    emit(f"s_waitcnt vmcnt({fixed + 2 * nw})")
    emit(f"s_branch {ld}")
    emit(f"{lt}:")
    emit(f"s_waitcnt vmcnt({fixed + 3 * nw})")
    emit(f"{ld}:")


def if_flag(sreg, body, else_body=None):
    ls, le = label("f0"), label("f1")
    emit(f"s_cmp_eq_u32 s{sreg}, 0")
    emit(f"s_cbranch_scc1 {ls}")
    body()
    if else_body is not None:
        emit(f"s_branch {le}")
    emit(f"{ls}:")
    if else_body is not None:
        else_body()
        emit(f"{le}:")


def s_rotate():
    rotate()


def stage_g0(set_, odd):
    # P_2t: MFMA(t);  then own W(t+1) pieces must have landed (deadline: G1 reads W(t+1) in P_2t+2 ... G0 in P_2t+1):
    # queue tail at this point: A(t+1), W(t+2) [issued in P_2t-1]  ->  allowed = 4*has1 + n_w*has2
    emit(f"; ==== G0 {'odd' if odd else 'even'} stage: P_2t  MFMA(t)")
    s_mfma(set_)
    if odd:
        if_flag(S_MORE, lambda: s_wait(4, 1), lambda: emit("s_waitcnt vmcnt(0)"))
    else:
        if_flag(S_MORE, lambda: s_wait(4, 1), lambda: emit("s_waitcnt vmcnt(4)"))
    barrier()
    emit("; P_2t+1: READ(t+1); A(t+2) -> set t&1; W(t+3) -> slot t%3; then A(t+1) must have landed")
    if odd:
        def body():
            s_read(S_WNXT)
            s_load_a(set_, 0)                          # A(t+2): k = s_k2
            if_flag(S_MORE2, lambda: s_issue_w(S_WCUR, BK))   # W(t+3)
            lgkm0()
            # allowed: W(t+2) + A(t+2) + W(t+3)
            if_flag(S_MORE2, lambda: s_wait(4, 2), lambda: s_wait(4, 1))
        if_flag(S_MORE, body, lambda: emit("s_waitcnt vmcnt(0)"))
    else:
        s_read(S_WNXT)
        def body():
            s_load_a(set_, 0)
            s_issue_w(S_WCUR, BK)
        if_flag(S_MORE, body)
        lgkm0()
        # t == 0: W(2) came from the prologue, ahead of A(0) / A(1) in the queue -> only A(2), W(3) may stay in flight
        l0, l1 = label("p1t0"), label("p1d")
        emit(f"s_cmp_eq_u32 s{S_T}, 0")
        emit(f"s_cbranch_scc1 {l0}")
        if_flag(S_MORE, lambda: s_wait(4, 2), lambda: emit("s_waitcnt vmcnt(0)"))
        emit(f"s_branch {l1}")
        emit(f"{l0}:")
        if_flag(S_MORE, lambda: s_wait(4, 1), lambda: emit("s_waitcnt vmcnt(0)"))
        emit(f"{l1}:")
    barrier()
    s_rotate()


def stage_g1(set_, odd):
    other = 1 - set_
    emit(f"; ==== G1 {'odd' if odd else 'even'} stage: P_2t  READ(t); A(t+1) -> other set; W(t+2) -> slot (t+2)%3")
    s_read(S_WCUR)
    if odd:
        def body():
            s_load_a(other, -BK)                       # A(t+1): k = s_k2 - 128
            s_issue_w(S_WPRV, 0)                       # W(t+2)
        if_flag(S_MORE, body)
        lgkm0()
        if_flag(S_MORE, lambda: s_wait(4, 1), lambda: emit("s_waitcnt vmcnt(0)"))
    else:
        l0 = label("t0")
        emit(f"s_cmp_eq_u32 s{S_T}, 0")
        emit(f"s_cbranch_scc1 {l0}")
        s_load_a(other, -BK)
        emit(f"{l0}:")
        if_flag(S_MORE, lambda: s_issue_w(S_WPRV, 0))
        lgkm0()
        l1, l2 = label("w0"), label("w1")
        emit(f"s_cmp_eq_u32 s{S_T}, 0")
        emit(f"s_cbranch_scc1 {l1}")
        if_flag(S_MORE, lambda: s_wait(4, 1), lambda: emit("s_waitcnt vmcnt(4)"))
        emit(f"s_branch {l2}")
        emit(f"{l1}:")
        if_flag(S_MORE, lambda: s_wait(0, 1), lambda: emit("s_waitcnt vmcnt(0)"))
        emit(f"{l2}:")
    barrier()
    emit("; P_2t+1: MFMA(t)")
    s_mfma(set_)
    barrier()
    s_rotate()


def generate_stage():
    emit("; generated by tools/gen_pp_asm.py (stage-granular ping-pong) -- do not edit")
    emit("s_nop 4")
    emit(f"s_mov_b32 s{S_T}, 0")
    emit(f"s_sub_u32 s{S_LAST}, %[kt], 2")
    emit(f"s_lshl_b32 s{S_WAVEK}, %[wave], 10")
    emit(f"s_cmp_lt_u32 %[wave], 6")
    emit(f"s_cselect_b32 s{S_TAIL}, 1, 0")
    emit(f"s_mov_b32 s{S_WCUR}, 0")
    emit(f"s_mov_b32 s{S_WNXT}, {W_BYTES}")
    emit(f"s_mov_b32 s{S_WPRV}, {2 * W_BYTES}")
    emit(f"s_mov_b32 s{S_K2}, 0")
    s_load_a(0, 0)
    s_load_a(1, BK)
    emit(f"s_mov_b32 s{S_K2}, {2 * BK}")            # k offset (row-major bytes) of stage t + 2
    emit("s_waitcnt vmcnt(4)")                        # W(0), W(1), (W(2)), A(0) landed; A(1) may fly
    barrier()
    g1, end = label("g1"), label("end")
    emit("s_cmp_lt_u32 %[wave], 4")
    emit(f"s_cbranch_scc0 {g1}")
    # group 0
    s_read(S_WCUR)
    lgkm0()
    barrier()                                          # P_-1
    loop0 = label("loop0")
    emit(f"{loop0}:")
    emit(f"s_cmp_lt_u32 s{S_T}, s{S_LAST}")
    emit(f"s_cselect_b32 s{S_MORE}, 1, 0")
    emit(f"s_add_u32 s{S_TMP}, s{S_T}, 4")
    emit(f"s_cmp_lt_u32 s{S_TMP}, %[kt]")
    emit(f"s_cselect_b32 s{S_MORE2}, 1, 0")
    stage_g0(0, False)
    stage_g0(1, True)
    emit(f"s_cmp_lt_u32 s{S_T}, %[kt]")
    emit(f"s_cbranch_scc1 {loop0}")
    emit(f"s_branch {end}")
    # group 1
    emit(f"{g1}:")
    barrier()                                          # P_-1
    loop1 = label("loop1")
    emit(f"{loop1}:")
    emit(f"s_cmp_lt_u32 s{S_T}, s{S_LAST}")
    emit(f"s_cselect_b32 s{S_MORE}, 1, 0")
    stage_g1(0, False)
    stage_g1(1, True)
    emit(f"s_cmp_lt_u32 s{S_T}, %[kt]")
    emit(f"s_cbranch_scc1 {loop1}")
    emit(f"{end}:")
    emit("s_nop 15")
    emit("s_nop 15")


def generate():
    emit("; generated by tools/gen_pp_asm.py -- do not edit")
    emit("s_nop 4")
    # loop state
    emit(f"s_mov_b32 s{S_T}, 0")
    emit(f"s_sub_u32 s{S_LAST}, %[kt], 2")
    emit(f"s_lshl_b32 s{S_WAVEK}, %[wave], 10")
    emit(f"s_mov_b32 s{S_TAIL}, 0")
    emit(f"s_cmp_lt_u32 %[wave], 6")
    emit(f"s_cselect_b32 s{S_TAIL}, 1, 0")
    emit(f"s_mov_b32 s{S_WCUR}, 0")
    emit(f"s_mov_b32 s{S_WNXT}, {W_BYTES}")
    emit(f"s_mov_b32 s{S_WPRV}, {2 * W_BYTES}")
    # A(0) -> set 0, A(1) -> set 1 (issued after the W(0), W(1) LDS-DMA of the C++ prologue)
    emit(f"s_mov_b32 s{S_K2}, 0")
    load_a(0)
    emit(f"s_mov_b32 s{S_K2}, {BK}")
    load_a(1)
    emit(f"s_mov_b32 s{S_K2}, {2 * BK}")            # k offset of stage t + 2
    _prologue[0] = False
    emit("s_waitcnt vmcnt(4)")                        # W(0), W(1), A(0) landed; A(1) may fly
    barrier()
    g1, end = label("g1"), label("end")
    emit("s_cmp_lt_u32 %[wave], 4")
    emit(f"s_cbranch_scc0 {g1}")
    # ---------------- group 0
    read_w(S_WCUR, 0)
    lgkm0()
    barrier()
    loop0 = label("loop0")
    emit(f"{loop0}:")
    emit(f"s_cmp_lt_u32 s{S_T}, s{S_LAST}")
    emit(f"s_cselect_b32 s{S_MORE}, 1, 0")
    group0_stage(0, False)
    group0_stage(1, True)
    emit(f"s_cmp_lt_u32 s{S_T}, %[kt]")
    emit(f"s_cbranch_scc1 {loop0}")
    emit(f"s_branch {end}")
    # ---------------- group 1
    emit(f"{g1}:")
    barrier()
    loop1 = label("loop1")
    emit(f"{loop1}:")
    emit(f"s_cmp_lt_u32 s{S_T}, s{S_LAST}")
    emit(f"s_cselect_b32 s{S_MORE}, 1, 0")
    group1_stage(0, False)
    group1_stage(1, True)
    emit(f"s_cmp_lt_u32 s{S_T}, %[kt]")
    emit(f"s_cbranch_scc1 {loop1}")
    emit(f"{end}:")
    emit("s_nop 15")          # MFMA results -> v_accvgpr_read of the copy-out statements (up to 18 wait states)
    emit("s_nop 15")


def main(path=None):
    generate_stage() if STAGE else generate()
    here = os.path.dirname(os.path.abspath(__file__))
    path = path or os.path.join(here, "..", "mobilequant_amd", "csrc", "mq_gemm_pp_asm.inc")
    vregs = [f'"v{r}"' for r in (range(WA0, 128) if STAGE else range(WF0, V_A + 2))]
    aregs = [f'"a{r}"' for r in range(0, 120)]
    sregs = [f'"s{r}"' for r in range(S_T, (S_MORE2 if STAGE else S_TMP2) + 1)]
    with open(path, "w") as f:
        f.write("// Generated by tools/gen_pp_asm.py -- do not edit (see that file for the register map and the schedule).\n")
        f.write("// Operands: kt, wave (SGPR); aptr, wptr (SGPR pairs); woff0, woff1, av0, av1, sw0, sw1, sw2 (VGPR).\n")
        f.write("#define MQ_PP_ASM_STAGE_MODE %d   // 1: the C++ prologue also issues W(2) for the waves of group 0\n" % (1 if STAGE else 0))
        f.write("#define MQ_PP_ASM_BODY \\\n")
        for line in out:
            f.write('  "%s\\n\\t" \\\n' % line.replace('"', '\\"'))
        f.write('  ""\n')
        for c in range(4):        # accumulator transfer, 22 registers per asm statement (operand limit 30)
            f.write(f"#define MQ_PP_ASM_COPYIN{c} " + " ".join('"v_accvgpr_write_b32 a%d, %%%d\\n\\t"' % (22 * c + k, k) for k in range(22)) + "\n")
            f.write(f"#define MQ_PP_ASM_COPYOUT{c} " + " ".join('"v_accvgpr_read_b32 %%%d, a%d\\n\\t"' % (k, 22 * c + k) for k in range(22)) + "\n")
        f.write("#define MQ_PP_ASM_ACLOBBERS " + ", ".join(aregs) + "\n")
        f.write("#define MQ_PP_ASM_CLOBBERS " + ", ".join(vregs + aregs + sregs + ['"scc"', '"memory"']) + "\n")
    print("wrote", os.path.normpath(path), len(out), "instructions/labels")


if __name__ == "__main__":
    main()
