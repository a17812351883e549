#!/usr/bin/env python3
"""Audit of the compiled kernels' s_barrier instructions (round 6: a loop whose header block starts with `s_barrier` and whose latch ends in
ds_write -- hipcc put the release fence's `s_waitcnt lgkmcnt(0)` BEHIND the barrier, and another wave read the LDS tile before the last
write had landed: tests/fuzz_qmatmul.py case 85).
usage: barrier_audit.py file.s ...   (hipcc -S --cuda-device-only output).  For every barrier the instructions in front of it are walked
backwards through the control-flow graph (fall-through and branch predecessors, depth 6): a path that meets a ds_* instruction before an
`s_waitcnt ... lgkmcnt(0)` is reported as PENDING (an LDS operation of this wave may be in flight when it signals the barrier)."""
import re, sys


def audit(path):
    lines = open(path).read().split("\n")
    # functions
    starts = [n for n, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    bad = total = 0
    for fi, s in enumerate(starts):
        e = starts[fi + 1] if fi + 1 < len(starts) else len(lines)
        name = lines[s].split(":")[0]
        label_at = {}
        for n in range(s, e):
            m = re.match(r"^(\.LBB\d+_\d+):", lines[n])
            if m:
                label_at[m.group(1)] = n
        branches = {}
        for n in range(s, e):
            t = lines[n].strip()
            m = re.match(r"^s_(?:c)?branch\w*\s+(\.LBB\d+_\d+)", t)
            if m:
                branches.setdefault(m.group(1), []).append(n)

        def walk(n, depth, seen):
            """backwards from line n (exclusive); returns True if some path meets a ds_ op before an lgkmcnt(0) wait"""
            k = n - 1
            while k >= s:
                t = lines[k].strip()
                if t.startswith("s_waitcnt") and ("lgkmcnt(0)" in t or t == "s_waitcnt 0"):
                    return False
                if t.startswith("ds_"):
                    return True
                if t.startswith("s_barrier") or t.startswith("s_endpgm"):
                    return False
                m = re.match(r"^(\.LBB\d+_\d+):", lines[k])
                if m:
                    if depth == 0 or (m.group(1), ) in seen:
                        return False
                    seen = seen | {(m.group(1), )}
                    res = False
                    # fall-through predecessor: the instruction above unless it is an unconditional branch
                    j = k - 1
                    while j >= s and (not lines[j].strip() or lines[j].strip().startswith(";")):
                        j -= 1
                    if j >= s and not lines[j].strip().startswith(("s_branch", "s_endpgm", "s_setpc")):
                        res |= walk(k, depth - 1, seen) if False else walk_from(j + 1, depth - 1, seen)
                    for b in branches.get(m.group(1), []):
                        res |= walk_from(b, depth - 1, seen)
                    return res
                k -= 1
            return False

        def walk_from(n, depth, seen):
            return walk(n, depth, seen)

        for n in range(s, e):
            if lines[n].strip().startswith("s_barrier"):
                total += 1
                if walk(n, 6, frozenset()):
                    bad += 1
                    print(f"{path.split('/')[-1]}:{n + 1} PENDING {name[:70]}")
    print(path.split("/")[-1], "barriers", total, "with an LDS operation possibly in flight", bad)


for p in sys.argv[1:]:
    audit(p)
