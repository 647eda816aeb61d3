"""The N>1 path on CPU: world_size-2 gloo run of the gradient-arena reduction and the view sharding."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import ROOT, PKG_NAME


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import importlib
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    d = importlib.import_module(f"{PKG_NAME}.dist")
    r, w = d.init_from_env(backend="gloo")
    n = 1000
    arena = torch.full((d.ARENA_FLOATS * n,), float(rank + 1))
    arena[:3] = torch.tensor([1.0, 2.0, 3.0]) * (rank + 1)
    d.reduce_gradients(arena, w, average=True)
    views = d.views_for_rank(8, r, w)
    parts = d.arena_views(arena, n)
    # factored exchange: the 11-float arena is averaged, every rank ends with every view's payload in rank order
    small = torch.full((d.SMALL_ARENA_FLOATS * n,), float(rank + 1))
    payload = torch.arange(3 * n + 4, dtype=torch.float32) + 10000.0 * rank
    gathered = d.exchange_factored(small, payload, average=True)
    fact = (tuple(gathered.shape), float(small[0]), float(small[-1]), gathered[:, 0].tolist(), gathered[:, -1].tolist(),
            {k: tuple(v.shape) for k, v in d.small_arena_views(small, n).items()})
    # the overlapped form: async gather started first, async reduce inside finish(); the HIP rebuild is replaced by a probe
    small2 = torch.full((d.SMALL_ARENA_FLOATS * n,), float(rank + 1))
    ex = d.FactoredExchange()
    ex.start_gather(payload)
    probe = {}
    d.sh_gradients_from_views = lambda means, gathered, degree, average=True, out=None: probe.setdefault("g", gathered.clone())
    res = ex.finish({"_arena": small2}, None, 3, average=True)
    over = (tuple(probe["g"].shape), probe["g"][:, 0].tolist(), float(small2[0]), sorted(res), tuple(res["dL_drot"].shape))
    q.put((r, w, arena[:4].tolist(), float(arena[-1]), views, {k: tuple(v.shape) for k, v in parts.items()}, fact, over))
    torch.distributed.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, w, head, tail, views, shapes, fact, over in res:
        assert over == ((2, 3004), [0.0, 10000.0], 1.5, ["dL_dmean3D", "dL_dopacity", "dL_drot", "dL_dscale", "dL_dshs"], (1000, 4))
        assert fact[0] == (2, 3004) and fact[1] == 1.5 and fact[2] == 1.5
        assert fact[3] == [0.0, 10000.0] and fact[4] == [3003.0, 13003.0]
        assert fact[5] == {"dL_dmean3D": (1000, 3), "dL_dscale": (1000, 3), "dL_drot": (1000, 4), "dL_dopacity": (1000,)}
        assert w == 2
        np.testing.assert_allclose(head, [1.5, 3.0, 4.5, 1.5])    # mean of rank+1 scalings
        assert tail == 1.5
        assert shapes == {"dL_dmean3D": (1000, 3), "dL_dscale": (1000, 3), "dL_drot": (1000, 4), "dL_dopacity": (1000,),
                          "dL_dshs": (16000, 3)}
    assert res[0][4] == [0, 2, 4, 6] and res[1][4] == [1, 3, 5, 7]


def _launch_mod():
    import importlib
    return importlib.import_module(f"{PKG_NAME}.launch")


def test_self_launcher_two_ranks_gloo(tmp_path, capfd):
    """The launcher behind `bench.py --gpus N` / `examples/train.py --gpus N`: two child ranks, gloo, CPU tensors."""
    rc = _launch_mod().launch_ranks(os.path.join(ROOT, "tests", "dist_worker.py"), ["cpu", str(tmp_path)], 2, timeout=240)
    assert rc == 0
    out = capfd.readouterr().out
    assert out.count('"rank0": "done"') == 1          # rank 0's stdout is relayed, rank 1's is dropped
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in ("arena", "small", "gathered"):
        np.testing.assert_array_equal(r0[k], r1[k])
    assert float(r0["arena"][0]) == 1.5 and float(r0["small"][-1]) == 1.5
    assert r0["gathered"].shape == (2, 1504) and r0["gathered"][1, 0] == 1000.0


def test_self_launcher_reports_failed_rank(tmp_path):
    """A rank that dies ends the job with its exit code; the surviving rank is terminated, nothing is retried."""
    import time
    t0 = time.monotonic()
    rc = _launch_mod().launch_ranks(os.path.join(ROOT, "tests", "dist_worker.py"), ["fail", str(tmp_path)], 2, timeout=240)
    assert rc == 3
    assert time.monotonic() - t0 < 50                 # rank 0 (sleeping 60 s) was stopped, not waited for


def test_bench_launcher_branch_makes_no_gpu_call():
    """bench.py decides to self-launch before torch is imported (the parent must not touch the GPU)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("launch_ranks(") < src.index("import torch\n")
    assert "import torch" not in open(os.path.join(ROOT, PKG_NAME, "launch.py")).read()
