"""Host-side behaviour of bench.py (launcher contract) and of the torch-CPU baseline restatement -- no GPU needed."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import mq_oracle as O
from oracle import mq_oracle_torch as T


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_flag_fails_loudly_when_the_box_has_fewer_gpus(monkeypatch):
    b = _bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        b.maybe_spawn(types.SimpleNamespace(gpus=2))
    assert "only 1 GPU" in str(e.value)
    b.maybe_spawn(types.SimpleNamespace(gpus=1))            # N = 1: nothing to launch


def test_gpus_flag_must_agree_with_an_outer_launcher(monkeypatch):
    b = _bench()
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        b.maybe_spawn(types.SimpleNamespace(gpus=8))
    assert "WORLD_SIZE=4" in str(e.value)
    b.maybe_spawn(types.SimpleNamespace(gpus=4))            # consistent: the outer torchrun's ranks proceed


def test_gpus_flag_relaunches_under_torch_distributed_run(monkeypatch):
    b = _bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    seen = {}
    import subprocess

    def fake_call(cmd, *a, **k):
        seen["cmd"] = cmd
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20"])
    with pytest.raises(SystemExit) as e:
        b.maybe_spawn(types.SimpleNamespace(gpus=4))
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "20"]


# ---- the torch-CPU baseline computes the oracle's arithmetic --------------------------------------------------------
@pytest.mark.parametrize("bits,sym", [(8, False), (8, True), (4, False), (16, False)])
def test_torch_baseline_quantizer_is_bit_exact_with_the_oracle(bits, sym):
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((64, 96)) * 2.5).astype(np.float32)
    x[0, :5] = [0.5, 1.5, 2.5, -0.5, -1.5]
    q = T.Quantizer(bits, sym).set_range(-2.25, 3.5)
    s, o, qmin, qmax = O.scale_offset_from_min_max(-2.25, 3.5, bits, sym)
    assert np.array_equal(np.float32(q.scale), s) and np.array_equal(np.float32(q.offset), o)
    got = q(torch.from_numpy(x)).numpy()
    assert np.array_equal(got.view(np.uint32), O.fake_quant(x, s, o, qmin, qmax).view(np.uint32))
    # first-forward per-row ranges (weights)
    qw = T.Quantizer(bits, sym, per_channel=True)
    got = qw(torch.from_numpy(x)).numpy()
    mn, mx = O.min_max_from_tensor(x, True)
    s, o, qmin, qmax = O.scale_offset_from_min_max(mn, mx, bits, sym)
    assert np.array_equal(got.view(np.uint32), O.fake_quant(x, s, o, qmin, qmax).view(np.uint32))


def test_torch_baseline_qlinear_matches_the_oracle_simulation():
    rng = np.random.default_rng(6)
    x = rng.standard_normal((48, 128)).astype(np.float32)
    w = (rng.standard_normal((40, 128)) * 0.05).astype(np.float32)
    y = x @ w.T
    wq, iq, oq = O.QuantizerOracle(8), O.QuantizerOracle(8), O.QuantizerOracle(8)
    iq.set_from_minmax(float(x.min()), float(x.max()))
    oq.set_from_minmax(float(y.min()), float(y.max()))
    want = O.qlinear_sim(x, w, None, wq, iq, oq)
    twq, tiq, toq = T.Quantizer(8), T.Quantizer(8).set_range(float(x.min()), float(x.max())), T.Quantizer(8).set_range(float(y.min()), float(y.max()))
    got = T.qlinear(torch.from_numpy(x), torch.from_numpy(w), None, twq, tiq, toq).numpy()
    d = np.abs(got - want)
    assert d.max() <= float(oq.scale) * 1.001 and (d == 0).mean() > 0.99       # BLAS summation order only


def test_torch_baseline_layer_runs_and_reports_physical_cores():
    layer = T.SimLayer(hidden=128, heads=4, kv_heads=2, head_dim=32, ffn=256)
    out = layer.forward(torch.randn(16, 128))
    assert out.shape == (16, 128) and torch.isfinite(out).all()
    assert 1 <= T.physical_cores() <= (os.cpu_count() or 1)
