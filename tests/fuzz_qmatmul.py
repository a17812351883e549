"""Random-shape fuzz of mq_qmatmul against oracle.qmatmul_exact, every output bit for bit (test infrastructure; run by hand on an MI355X:
python tests/fuzz_qmatmul.py [cases])."""
import sys, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import mq_oracle as O
from mobilequant_amd import ops
import test_gpu_round5 as T
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = 0
for it in range(n_cases):
    lead = tuple(int(x) for x in rng.integers(1, 4, size=int(rng.integers(0, 3))))
    M, N, K = int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.integers(1, 300))
    if rng.random() < 0.2:
        K = int(rng.integers(300, 1500))
    kt = bool(rng.integers(0, 2))
    b1 = int(rng.choice([4, 8, 8, 8, 12, 16, 16])); s1 = bool(rng.integers(0, 2))
    b2 = int(rng.choice([4, 6, 8, 8, 8])); s2 = bool(rng.integers(0, 2))
    bo = [None, (8, False), (8, True), (16, False), (16, True)][int(rng.integers(0, 5))]
    a = (rng.standard_normal(lead + (M, K), dtype=np.float32) * float(rng.choice([0.1, 1.0, 7.0])) + float(rng.choice([0.0, 0.5, -2.0]))).astype(np.float32)
    b = (rng.standard_normal(lead + (K, N), dtype=np.float32) * float(rng.choice([0.05, 1.0, 3.0])) + float(rng.choice([0.0, 0.3]))).astype(np.float32)
    clip = float(rng.choice([0.8, 1.0]))
    g1 = T._grid(b1, s1, float(a.min()) * clip, float(a.max()) * clip)
    g2 = T._grid(b2, s2, float(b.min()) * clip, float(b.max()) * clip)
    fp = np.matmul(a, b)
    go = None if bo is None else T._grid(bo[0], bo[1], float(fp.min()) * 0.9, float(fp.max()) * 0.9)
    if os.environ.get("ONLY") and it != int(os.environ["ONLY"]):
        continue
    want = O.qmatmul_exact(a, b, g1, g2, go)
    ta = torch.from_numpy(a).to(dev)
    tb = torch.from_numpy(np.ascontiguousarray(np.swapaxes(b, -1, -2))).to(dev).transpose(-1, -2) if kt else torch.from_numpy(b).to(dev)
    got = ops.qmatmul(ta, tb, T._dev_grid(g1, dev), T._dev_grid(g2, dev), None if go is None else T._dev_grid(go, dev)).cpu().numpy()
    neq = got.view(np.uint32) != want.view(np.uint32)
    if neq.any():
        bad += 1
        if os.environ.get("ONLY"):
            print("where", np.argwhere(neq)[:8].tolist(), np.argwhere(neq)[-3:].tolist(), got[neq][:4], want[neq][:4])
        print("BAD", it, lead, M, N, K, kt, (b1, s1), (b2, s2), bo, int(neq.sum()), "of", neq.size, np.abs(got - want).max(), flush=True)
print("cases", n_cases, "bad", bad)
sys.exit(1 if bad else 0)
