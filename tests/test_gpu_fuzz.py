"""A bounded, fixed-seed slice of every random-shape fuzzer inside `pytest -m gpu` (the fuzzers used to be run by
hand only, so nothing random-shaped ran under the driver).  Each fuzzer is a script (tests/fuzz_*.py: seeded numpy generators, exits 1
on any mismatch); here each runs as a child process with a small case count, so a failure prints the offending shape and the parent
process keeps its own library state.  The full-length runs stay available by hand: `python tests/fuzz_round3.py 200`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (script, case count): sized for <= ~10 s of GPU work each (plus the child's torch import)
SLICES = [("fuzz_linear.py", 10), ("fuzz_attention.py", 16), ("fuzz_qmatmul.py", 100), ("fuzz_round3.py", 6), ("fuzz_decode.py", 10), ("stress_qmatmul_race.py", 20)]


@pytest.mark.parametrize("script,cases", SLICES, ids=[s for s, _ in SLICES])
def test_fuzz_slice(script, cases):
    path = os.path.join(ROOT, "tests", script)
    r = subprocess.run([sys.executable, path, str(cases)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    tail = "\n".join(r.stdout.splitlines()[-25:])
    print(tail)
    assert r.returncode == 0, f"{script} {cases}: exit {r.returncode}\n{tail}"
