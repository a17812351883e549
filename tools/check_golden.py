#!/usr/bin/env python3
"""Does oracle/gen_golden.py still reproduce the committed fixtures?  (VERDICT r03 "What's weak" 9: three fixtures had gone stale
against their generator without any test noticing.)

    python tools/check_golden.py                      # the quick generators (seconds to a minute each)
    python tools/check_golden.py --all                # every generator (the BASELINE-sized ones take minutes)
    python tools/check_golden.py gen_decode_case_w4   # selected generators

Needs the reference checkout (/root/reference or $MQ_REFERENCE): the generators import it.  Each generator runs in a subprocess with
MQ_GOLDEN_OUT pointing at a temp dir; every file it writes is compared with tests/golden/<file>: .npz key by key (arrays bit for bit,
strings verbatim), .json / .pth by content.  Exit status 1 on any difference or generator failure.  A generator that reads another
one's output (DEPS) runs behind its producers in one temp dir.  Every gen_* function of oracle/gen_golden.py must be in QUICK or SLOW
(tests/test_host_api.py checks that)."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
QUICK = ["gen_scale_offset_grid", "gen_quantizer_cases", "gen_nonfinite", "gen_quantizer_grads", "gen_qrmsnorm_cases", "gen_qact_cases",
         "gen_lwc_cases", "gen_qmatmul_cases", "gen_decode_case", "gen_decode_case_w4", "gen_decode_case_w8pc_mha", "gen_decode_case_gelu",
         "gen_decode_case_stablelm", "gen_decode_case_gemma", "gen_generate_case", "gen_api_surface", "gen_toy_lm_nll",
         "gen_qlinear_dynamic_cases", "gen_qlinear_grouped_cases"]
SLOW = ["gen_qlinear_cases", "gen_calib_stream", "gen_checksums", "gen_smooth_cases", "gen_artifacts", "gen_train_step", "gen_layer_case",
        "gen_full_depth_case", "gen_full_depth_stable_case"]
# generators that READ another generator's output from MQ_GOLDEN_OUT: the producers run first, in the same temp dir and process
DEPS = {"gen_generate_case": ["gen_decode_case"], "gen_artifacts": ["gen_api_surface", "gen_smooth_cases"]}

def compare(fresh, gold):
    if fresh.endswith(".npz"):
        a, b = np.load(fresh, allow_pickle=False), np.load(gold, allow_pickle=False)
        if sorted(a.files) != sorted(b.files):
            return f"keys differ: only fresh {sorted(set(a.files) - set(b.files))[:4]}, only committed {sorted(set(b.files) - set(a.files))[:4]}"
        for k in a.files:
            x, y = a[k], b[k]
            if x.dtype != y.dtype or x.shape != y.shape:
                return f"{k}: {x.dtype}{x.shape} vs {y.dtype}{y.shape}"
            same = (x == y).all() if x.dtype.kind in "US" else x.tobytes() == y.tobytes()          # floats bit for bit (NaN payloads, -0.0)
            if not same:
                return f"{k}: values differ"
        return None
    if fresh.endswith(".json"):
        return None if json.load(open(fresh)) == json.load(open(gold)) else "json differs"
    if fresh.endswith(".pth"):
        import torch
        a, b = torch.load(fresh), torch.load(gold)
        ok = a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
        return None if ok else "tensors differ"
    return None if open(fresh, "rb").read() == open(gold, "rb").read() else "bytes differ"


def run(names, ref=None, jobs=None):
    ref = ref or os.environ.get("MQ_REFERENCE", "/root/reference")
    if not os.path.isdir(ref):
        print(f"check_golden: no reference checkout at {ref}: nothing to compare")
        return 0
    from concurrent.futures import ThreadPoolExecutor

    def one(name):
        """-> (name, lines to print, number of problems, {file: verdict})"""
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(os.environ, MQ_GOLDEN_OUT=tmp, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=ref, MQ_REFERENCE=ref)
            chain = DEPS.get(name, []) + [name]
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), *chain], cwd=tmp, env=env, capture_output=True, text=True)
            if r.returncode != 0:
                return name, [f"{name}: generator FAILED\n{r.stderr[-2000:]}"], 1, {}
            files = sorted(f for f in os.listdir(tmp) if os.path.isfile(os.path.join(tmp, f)))
            if not files:
                return name, [f"{name}: wrote nothing"], 1, {}
            verdicts = {}
            for f in files:
                gold = os.path.join(GOLD, f)
                verdicts[f] = "not committed" if not os.path.exists(gold) else compare(os.path.join(tmp, f), gold)
            return name, [], 0, verdicts
    bad = 0
    seen = {}                                     # file -> verdict already printed (a producer run again as somebody's dependency)
    with ThreadPoolExecutor(max_workers=jobs or max(1, min(6, (os.cpu_count() or 2) - 1))) as pool:       # generators are single-threaded
        for name, lines, problems, verdicts in pool.map(one, names):
            for line in lines:
                print(line, flush=True)
            bad += problems
            for f, why in verdicts.items():
                if f in seen:
                    continue
                seen[f] = why
                print(f"{name}: {f}: " + ("identical" if why is None else "DIFFERS -- " + why), flush=True)
                bad += why is not None
    return 1 if bad else 0


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    sys.exit(run(args or (QUICK + SLOW if "--all" in sys.argv else QUICK)))
