"""The N>1 calibration path on CPU: two gloo ranks shard a sample stream round-robin, keep running
[min, max] per tensor, and merge with ONE packed all-reduce(MAX) -- identical to the unsharded oracle.
(The per-tensor reductions themselves are HIP kernels and are covered by the gpu tests; here each rank's
running statistics are produced by the oracle so the collective, packing and layout logic run without a GPU.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_npz
from oracle import mq_oracle as O
from toy_models import CalibToy


def _stream(z, prefix):
    items = {}
    for key in z.files:
        if key.startswith(prefix + "|"):
            _, name, field, idx = key.split("|")
            items.setdefault(int(idx), []).append((name, field, z[key]))
    return [items[i] for i in sorted(items)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, per_channel, empty_rank, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mobilequant_amd.calibration import ActRangeCollector
        z = load_npz("calib_stream.npz")
        samples = _stream(z, "stream_pc" if per_channel else "stream_pt")
        if empty_rank:
            samples = samples[:1]              # fewer samples than ranks: rank 1 sees nothing
        col = ActRangeCollector(CalibToy(), per_channel=per_channel, device="cpu")
        calls = {"n": 0}
        real = dist.all_reduce

        def counting(*a, **k):
            calls["n"] += 1
            return real(*a, **k)
        dist.all_reduce = counting
        real_gather = dist.all_gather_object

        def counting_gather(*a, **k):          # any second (object) collective would show up in the count
            calls["n"] += 1
            return real_gather(*a, **k)
        dist.all_gather_object = counting_gather
        mine = O.ActRangeOracle(per_channel)
        # get_act_range's sharding: round-robin, and a rank that owns no sample re-runs a duplicate (min / max are idempotent)
        for s in (samples[rank::world] or [samples[rank % len(samples)]]):
            for name, field, t in s:
                mine.update(name, field, t)
        for (name, field), i in col.slots.items():          # inject this rank's running statistics
            v = mine.act_dict.get(name, {}).get(field)
            if v is None:
                continue
            if per_channel:
                col._pc[i] = (torch.from_numpy(v[0].copy()), torch.from_numpy(v[1].copy()))
            else:
                col._mn[i], col._mx[i] = float(v[0]), float(v[1])
        col.all_reduce()
        dist.all_reduce, dist.all_gather_object = real, real_gather
        got = col.act_dict()
        full = O.ActRangeOracle(per_channel)
        for s in samples:
            for name, field, t in s:
                full.update(name, field, t)
        ok = calls["n"] == 1 and got.keys() == full.act_dict.keys()
        for name, fields in full.act_dict.items():
            for f, v in fields.items():
                if per_channel:
                    ok &= bool(np.array_equal(got[name][f].numpy(), v))
                else:
                    ok &= got[name][f] == [np.float32(v[0]), np.float32(v[1])]
        q.put((rank, bool(ok), calls["n"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("per_channel,empty_rank", [(False, False), (True, False), (False, True), (True, True)])
def test_two_rank_calibration_merge(per_channel, empty_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, per_channel, empty_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, 1), (1, True, 1)], res


def test_collector_slot_layout_is_data_independent():
    from mobilequant_amd.calibration import ActRangeCollector
    a = ActRangeCollector(CalibToy(), device="cpu")
    assert list(a.slots) == [("fc1", "input"), ("fc1", "output"), ("act", "input"), ("act", "output"), ("fc2", "input"),
                             ("fc2", "output"), ("bmm", "input"), ("bmm", "output"), ("bmm", "input2"),
                             ("ln", "input"), ("ln", "output")]
    assert a.act_dict() == {}          # nothing observed yet -> nothing reported


def _divergent_worker(rank, world, port, mode, q):
    """4 ranks, per-channel statistics.  mode 'verify': rank 3 never ran the module behind slot ('bmm', *) -> with verify_layout
    the ranks agree on the union layout and the merge equals the unsharded oracle.  mode 'swap': ranks observed different slots
    of EQUAL width (same buffer length) -> the in-buffer layout checksum raises instead of mis-merging."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mobilequant_amd.calibration import ActRangeCollector
        z = load_npz("calib_stream.npz")
        samples = _stream(z, "stream_pc")
        col = ActRangeCollector(CalibToy(), per_channel=True, device="cpu")
        mine = O.ActRangeOracle(True)
        full = O.ActRangeOracle(True)
        skip = (lambda name, field: name == "bmm") if (mode == "verify" and rank == 3) else (lambda name, field: False)
        for k, s in enumerate(samples):
            for name, field, t in s:
                if not (mode == "verify" and k % world == 3 and name == "bmm"):
                    full.update(name, field, t)              # what the job as a whole observed
                if k % world == rank and not skip(name, field):
                    mine.update(name, field, t)
        for (name, field), i in col.slots.items():
            v = mine.act_dict.get(name, {}).get(field)
            if v is not None:
                col._pc[i] = (torch.from_numpy(v[0].copy()), torch.from_numpy(v[1].copy()))
        if mode == "swap":
            i_in, i_out = col.slots[("fc1", "input")], col.slots[("ln", "input")]
            w = col._pc[i_in][0].numel()
            col._pc[i_out] = (torch.zeros(w), torch.ones(w))          # same width as fc1.input
            col._pc[i_in if rank % 2 else i_out] = None                 # odd ranks drop one, even ranks the other: equal lengths
            try:
                col.all_reduce()
                q.put((rank, "no error"))
            except RuntimeError as e:
                q.put((rank, "raised" if "different slot layouts" in str(e) else str(e)))
            return
        col.all_reduce(verify_layout=True)
        got = col.act_dict()
        ok = got.keys() == full.act_dict.keys()
        for name, fields in full.act_dict.items():
            for f, v in fields.items():
                ok &= bool(np.array_equal(got[name][f].numpy(), v))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["verify", "swap"])
def test_four_rank_merge_with_rank_divergent_observations(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_divergent_worker, args=(r, 4, port, mode, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True if mode == "verify" else "raised") for r in range(4)], res


def _cpu_reductions():
    """Test stand-ins for the three HIP reductions get_act_range launches (mq_minmax_init / _tensor / _cols: covered on the GPU box):
    the same running update with torch's amin / amax, so that the WHOLE data-parallel path -- hooks on a real model, round-robin shards,
    the duplicate of a sample-less rank, packing, the layout checksum, the one collective, unpacking -- runs on CPU ranks."""
    from mobilequant_amd import ops

    def new(n, device):
        return torch.full((n,), float("inf")), torch.full((n,), float("-inf"))

    def tensor_(x, mn, mx):
        mn.copy_(torch.minimum(mn, x.amin().reshape(1).float()))
        mx.copy_(torch.maximum(mx, x.amax().reshape(1).float()))

    def cols_(x2d, mn, mx):
        mn.copy_(torch.minimum(mn, x2d.amin(0).float()))
        mx.copy_(torch.maximum(mx, x2d.amax(0).float()))
    ops.minmax_new, ops.minmax_tensor_, ops.minmax_cols_ = new, tensor_, cols_


def _toy_llama():
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    m = LlamaForCausalLM(LlamaShape(hidden=64, layers=2, heads=4, kv_heads=2, head_dim=16, ffn=128, vocab=50, max_pos=32))
    m.reset_parameters(seed=11, std=0.2)
    return m.eval().requires_grad_(False)


def _toy_samples(n):
    g = torch.Generator().manual_seed(5)
    return [torch.randint(0, 50, (1, 24), generator=g) for _ in range(n)]


def _full_path_worker(rank, world, port, per_channel, n_samples, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _cpu_reductions()
        from mobilequant_amd.calibration import get_act_range
        calls = {"all_reduce": 0, "other": 0}
        real, real_gather = dist.all_reduce, dist.all_gather_object

        def counting(*a, **k):
            calls["all_reduce"] += 1
            return real(*a, **k)

        def counting_gather(*a, **k):
            calls["other"] += 1
            return real_gather(*a, **k)
        dist.all_reduce, dist.all_gather_object = counting, counting_gather
        act = get_act_range(_toy_llama(), _toy_samples(n_samples), per_channel=per_channel)
        dist.all_reduce, dist.all_gather_object = real, real_gather
        flat = {f"{n}|{f}": (v.numpy().tolist() if torch.is_tensor(v) else [float(v[0]), float(v[1])]) for n, d in act.items() for f, v in d.items()}
        q.put((rank, flat, calls["all_reduce"], calls["other"], dist.get_world_size()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("per_channel,n_samples", [(False, 12), (True, 12), (False, 5)])
def test_eight_rank_get_act_range_on_a_real_model_equals_the_single_process_run(per_channel, n_samples):
    """the first real 8-GPU run must not fail on plumbing.  EIGHT gloo ranks run the full get_act_range path
    (ptq/generate_act_range.py:49-122's data-parallel counterpart) on the toy llama graph -- forward hooks on every calibrated leaf
    incl. both matmuls, samples round-robin (12 samples: ranks own 2 or 1; 5 samples: three ranks own none and re-run a duplicate),
    ONE all-reduce of the packed statistics and no other collective -- and every rank ends with exactly the act_dict a single
    process computes over all samples (min / max are exact and order independent)."""
    _cpu_reductions()
    from mobilequant_amd.calibration import get_act_range
    want = get_act_range(_toy_llama(), _toy_samples(n_samples), per_channel=per_channel)
    want = {f"{n}|{f}": (v.numpy().tolist() if torch.is_tensor(v) else [float(v[0]), float(v[1])]) for n, d in want.items() for f, v in d.items()}
    assert len(want) >= 40                      # 2 layers x (7 linears + 2 norms + 2 matmuls + act) x (input, output[, input2]) + final norm
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_full_path_worker, args=(r, 8, port, per_channel, n_samples, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, flat, n_ar, n_other, size in res:
        assert size == 8 and n_ar == 1 and n_other == 0, (rank, n_ar, n_other, size)
        assert flat == want, (rank, [k for k in want if flat.get(k) != want[k]][:5])


# ---- the same path on the real thing: one rank per GPU over RCCL -------------------------------------------------
def _rccl_worker(rank, world, port, per_channel, n_samples, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from mobilequant_amd.calibration import get_act_range
        calls = {"all_reduce": 0}
        real = dist.all_reduce

        def counting(*a, **k):
            calls["all_reduce"] += 1
            return real(*a, **k)
        dist.all_reduce = counting
        model = _toy_llama().to(torch.device("cuda", rank))
        act = get_act_range(model, _toy_samples(n_samples), per_channel=per_channel)
        dist.all_reduce = real
        flat = {f"{n}|{f}": (v.cpu().numpy().tolist() if torch.is_tensor(v) else [float(v[0]), float(v[1])]) for n, d in act.items() for f, v in d.items()}
        q.put((rank, flat, calls["all_reduce"], dist.get_world_size(), dist.get_backend()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("per_channel", [False, True])
def test_get_act_range_over_rccl_on_every_gpu_of_the_box(per_channel):
    """One process per visible GPU, backend "nccl" (= RCCL over xGMI), the full get_act_range path with its HIP reductions: every rank
    must end with the act_dict of rank 0 running ALL samples alone (min / max are exact and order independent), after exactly ONE
    all-reduce.  `gpurun` leases one GPU, so this skips there; the first box with more than one GPU runs it without a code change."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"{n} GPU visible: the multi-rank RCCL path needs at least two (covered on CPU by the gloo tests above)")
    from mobilequant_amd.calibration import get_act_range
    n_samples = 2 * n + 1
    want = get_act_range(_toy_llama().to("cuda:0"), _toy_samples(n_samples), per_channel=per_channel)
    want = {f"{k}|{f}": (v.cpu().numpy().tolist() if torch.is_tensor(v) else [float(v[0]), float(v[1])]) for k, d in want.items() for f, v in d.items()}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, n, port, per_channel, n_samples, q)) for r in range(n)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, flat, n_ar, size, backend in res:
        assert size == n and backend == "nccl" and n_ar == 1, (rank, n_ar, size, backend)
        assert flat == want, (rank, [k for k in want if flat.get(k) != want[k]][:5])
