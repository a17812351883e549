"""Repeat-run stress of mq_qmatmul shapes around the row-panel kernel (40 launches each, every output against oracle.qmatmul_exact): a
race shows up as a run that differs (round 6: the barrier in front of which hipcc dropped the LDS wait).  Test infrastructure; python tests/stress_qmatmul_race.py"""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import mq_oracle as O
from mobilequant_amd import ops
import test_gpu_round5 as T
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
total_bad = 0
for (M, N, K, kt) in [(79, 120, 67, False), (79, 120, 68, False), (79, 120, 67, True), (79, 120, 68, True), (79, 120, 64, False), (79, 120, 130, False), (200, 300, 67, False), (64, 120, 67, False)]:
    a = (rng.standard_normal((M, K), dtype=np.float32)).astype(np.float32)
    b = (rng.standard_normal((K, N), dtype=np.float32)).astype(np.float32)
    g1 = T._grid(8, True, float(a.min()), float(a.max())); g2 = T._grid(6, True, float(b.min()), float(b.max()))
    fp = a @ b
    go = T._grid(16, True, float(fp.min()) * 0.9, float(fp.max()) * 0.9)
    want = O.qmatmul_exact(a, b, g1, g2, go)
    ta = torch.from_numpy(a).to(dev)
    tb = torch.from_numpy(np.ascontiguousarray(b.T)).to(dev).t() if kt else torch.from_numpy(b).to(dev)
    nbad = 0; where = set()
    for rep in range(REPS):
        got = ops.qmatmul(ta, tb, T._dev_grid(g1, dev), T._dev_grid(g2, dev), T._dev_grid(go, dev)).cpu().numpy()
        neq = got.view(np.uint32) != want.view(np.uint32)
        if neq.any():
            nbad += 1; w = np.argwhere(neq); where.add((int(w[0][0]), int(w[0][1]), int(neq.sum())))
    total_bad += nbad
    print((M, N, K, kt), "bad runs", nbad, "of", REPS, sorted(where)[:6], flush=True)
sys.exit(1 if total_bad else 0)
