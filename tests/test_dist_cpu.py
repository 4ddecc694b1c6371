"""CPU, 2 processes, gloo: the path's exchange step (packed all-gather, AllGather / AllGather2
autograd semantics, the 5-scalar loss reduce).  RCCL itself only runs on the GPU box."""
import os
import socket
from types import SimpleNamespace

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neighborretr_amd.dist import packed_allgather, reduce_losses
        from neighborretr_amd.until_module import AllGather, AllGather2
        args = SimpleNamespace(world_size=world, local_rank=rank)
        b, Nt, Nv, d = 3, 5, 4, 16
        g = torch.Generator().manual_seed(100 + rank)
        tf = torch.randn(b, Nt, d, generator=g, requires_grad=True)
        vf = torch.randn(b, Nv, d, generator=g, requires_grad=True)
        idx = torch.arange(b) + 10 * rank
        tm = (torch.rand(b, Nt, generator=g) > 0.3).long()
        vm = (torch.rand(b, Nv, generator=g) > 0.3).long()
        G_tf, G_vf, G_idx, G_tm, G_vm = packed_allgather(tf, vf, idx, tm, vm, args)
        assert G_tf.shape == (world * b, Nt, d) and G_vm.dtype == torch.float32 and G_idx.dtype == torch.int64
        # every rank must see rank r's shard at rows [r*b, (r+1)*b)
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            e_tf = torch.randn(b, Nt, d, generator=gr)
            e_vf = torch.randn(b, Nv, d, generator=gr)
            assert torch.equal(G_tf[r * b:(r + 1) * b], e_tf) and torch.equal(G_vf[r * b:(r + 1) * b], e_vf)
            assert G_idx[r * b:(r + 1) * b].tolist() == (torch.arange(b) + 10 * r).tolist()
        assert torch.equal(G_tm[rank * b:(rank + 1) * b], tm.float()) and torch.equal(G_vm[rank * b:(rank + 1) * b], vm.float())
        # backward = this rank's slice of the upstream gradient, no reduction (until_module.py:383-388)
        w = torch.arange(G_tf.numel(), dtype=torch.float32).view_as(G_tf)
        ((G_tf * w).sum() + 2.0 * G_vf.sum()).backward()
        assert torch.equal(tf.grad, w[rank * b:(rank + 1) * b])
        assert torch.equal(vf.grad, torch.full_like(vf, 2.0))

        # reference-named single-tensor functions
        x = torch.full((2, 3), float(rank + 1), requires_grad=True)
        y = AllGather.apply(x, args)
        assert y.shape == (2 * world, 3) and float(y[2 * (world - 1)].mean()) == float(world)
        (y * (rank + 1)).sum().backward()
        assert torch.equal(x.grad, torch.full_like(x, float(rank + 1)))
        x2 = torch.full((2, 3), 1.0, requires_grad=True)
        y2 = AllGather2.apply(x2, args)
        (y2 * (rank + 1)).sum().backward()                      # summed over ranks: 1 + 2 = 3
        assert torch.equal(x2.grad, torch.full_like(x2, float(sum(range(1, world + 1)))))

        red = reduce_losses([torch.tensor(float(rank + k)) for k in range(5)], args)
        if rank == 0:
            assert torch.allclose(red, torch.tensor([0.5, 1.5, 2.5, 3.5, 4.5]))
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_exchange_step_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_single_rank_is_identity():
    from neighborretr_amd.dist import packed_allgather
    from neighborretr_amd.until_module import AllGather
    args = SimpleNamespace(world_size=1, local_rank=0)
    tf = torch.randn(2, 3, 4, requires_grad=True)
    vf = torch.randn(2, 2, 4, requires_grad=True)
    out = packed_allgather(tf, vf, torch.arange(2), torch.ones(2, 3), torch.ones(2, 2), args)
    assert torch.equal(out[0], tf) and torch.equal(out[1], vf)
    (out[0].sum() + out[1].sum()).backward()
    assert torch.equal(tf.grad, torch.ones_like(tf))
    x = torch.randn(3, 2, requires_grad=True)
    assert AllGather.apply(x, args) is x or torch.equal(AllGather.apply(x, args), x)


def _eval_pad_worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neighborretr_amd.evaluator import gather_eval_features, rank_sample_indices
        args = SimpleNamespace(world_size=world, local_rank=rank)
        Nt, Nv, d = 4, 3, 8
        g = torch.Generator().manual_seed(3)
        T, V = torch.randn(n, Nt, d, generator=g), torch.randn(n, Nv, d, generator=g)
        TM, VM = (torch.rand(n, Nt, generator=g) > 0.3).long(), (torch.rand(n, Nv, generator=g) > 0.3).long()
        mine = rank_sample_indices(n, world, rank)
        assert len(mine) == -(-n // world)                     # the same row count on every rank
        t, v, tm, vm = gather_eval_features(T[mine], V[mine], mine, TM[mine], VM[mine], args)
        assert t.shape[0] == n and torch.equal(t, T) and torch.equal(v, V)
        assert torch.equal(tm, TM.float()) and torch.equal(vm, VM.float())
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_eval_gather_with_a_test_set_not_divisible_by_the_rank_count():
    """N % W != 0 (e.g. the 1000-sample default on 3 ranks) must not issue a collective with unequal byte counts:
    every rank pads its index list to ceil(N / W) by wrapping around (DistributedSampler's rule); dataset_order drops the
    duplicates."""
    from neighborretr_amd.evaluator import rank_sample_indices
    assert rank_sample_indices(10, 3, 0).tolist() == [0, 3, 6, 9] and rank_sample_indices(10, 3, 1).tolist() == [1, 4, 7, 0]
    assert rank_sample_indices(10, 3, 2).tolist() == [2, 5, 8, 1] and rank_sample_indices(8, 2, 1).tolist() == [1, 3, 5, 7]
    assert rank_sample_indices(2, 3, 2).tolist() == [0]        # more ranks than samples
    for world, n in ((3, 10), (2, 5)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_eval_pad_worker, args=(r, world, port, n, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=180) for _ in procs]
        for p in procs:
            p.join(timeout=60)
        assert all(r[1] == "ok" for r in res), res


def _toy_step(x, rank, world):
    """A step with the collective pattern of the sharded loss: gather -> max exchange -> compute -> sum of row terms -> the
    differentiable gather's reduce-scatter form."""
    from neighborretr_amd import comm
    comm.begin_step()
    n = x.shape[0]
    full = torch.empty((world * n,) + tuple(x.shape[1:]))
    comm.all_gather_into_tensor(full, x)
    m = full[rank * n:(rank + 1) * n].abs().max().reshape(1)
    comm.all_reduce(m, op="max")
    rows = torch.zeros(world * n)
    rows[rank * n:(rank + 1) * n] = (full[rank * n:(rank + 1) * n] @ full.sum(0)) / m
    comm.all_reduce(rows)
    part = full * (rank + 1.0)
    mine = torch.empty_like(x)
    if comm.backend() == "gloo":
        part = part.clone()
        comm.all_reduce(part)
        mine = part[rank * n:(rank + 1) * n].clone()
    else:
        comm.reduce_scatter_tensor(mine, part)
    return rows, mine


def _toy_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = torch.randn(3, 4, generator=torch.Generator().manual_seed(7 + rank))
        rows, mine = _toy_step(x, rank, world)
        q.put((rank, rows.numpy(), mine.numpy()))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_emulated_world_matches_a_real_two_rank_job():
    """comm.EmulatedWorld (how the per-rank step at W = 2 / 4 / 8 is measured on one GPU): the ranks of a job run one at a time,
    the peers' parts of every collective come from settled buffers -- same results as the same step on two real gloo ranks;
    a step whose ranks disagree about the kind of a collective is refused."""
    import numpy as np
    import pytest
    from neighborretr_amd import comm
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_toy_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    real = {r: (a, b) for r, a, b in (q.get(timeout=180) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
    assert all(not isinstance(v[0], str) for v in real.values()), real
    ew = comm.EmulatedWorld(world, real_collectives=False)
    got = {}

    def run(r):
        x = torch.randn(3, 4, generator=torch.Generator().manual_seed(7 + r))
        with comm.use(ew.comm(r)):
            assert comm.get_rank() == r and comm.get_world_size() == world and comm.backend() == "emulated"
            got[r] = _toy_step(x, r, world)
    sweeps = ew.settle(run)
    assert 2 <= sweeps <= 5 and not ew.recording
    run(0), run(1)                              # frozen form: the peers' parts are read, nothing is recorded
    for r in range(world):
        np.testing.assert_allclose(got[r][0].numpy(), real[r][0], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(got[r][1].numpy(), real[r][1], rtol=1e-5, atol=1e-6)
    assert comm.get_world_size() == 1 and comm.backend() == "none"        # outside the block: torch.distributed (not initialised here)

    bad = comm.EmulatedWorld(2, real_collectives=False)

    def diverging(r):
        with comm.use(bad.comm(r)):
            comm.begin_step()
            t = torch.ones(2)
            if r == 0:
                comm.all_reduce(t)
            else:
                comm.all_gather_into_tensor(torch.empty(4), t)
    with pytest.raises(RuntimeError, match="collective sequences differ"):
        bad.settle(diverging)


def _capture_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neighborretr_amd import comm
        log = []
        cc = comm.CollectiveCapture(world, rank, timeout_s=60, log=log.append)
        state = {"v": torch.zeros(3), "frozen": []}

        def eager():
            t = torch.full((3,), float(rank + 1))
            comm.all_reduce(t)                               # a collective inside the step: every rank must keep taking part
            state["v"] = t

        def form(fail_on=None, wrong_on=None):
            def make():
                if fail_on == rank:
                    raise RuntimeError("capture refused here")

                def replay():
                    t = torch.full((3,), float(rank + 1))
                    comm.all_reduce(t)
                    state["v"] = t + (1.0 if wrong_on == rank else 0.0)
                return replay, "keep"
            return make
        same = lambda a, b: bool(torch.equal(a, b))          # noqa: E731
        freeze = lambda on: state["frozen"].append(on)        # noqa: E731
        got = [cc.attempt("whole-step", eager, form(fail_on=0), lambda: state["v"], same, freeze),      # rank 0 cannot capture
               cc.attempt("whole-step", eager, form(wrong_on=1), lambda: state["v"], same, freeze),     # rank 1 replays wrongly
               cc.attempt("segmented", eager, form(), lambda: state["v"], same, freeze)]               # fine everywhere
        ok = (got[0] is None and got[1] is None and got[2] is not None and got[2][1] == "keep"
              and state["frozen"] == [True, False] * 3 and cc.agree(True) and not cc.agree(rank == 0))
        q.put((rank, "ok" if ok else f"unexpected: {[g is not None for g in got]} {state['frozen']}", log))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc(), []))
    finally:
        dist.destroy_process_group()


def test_capture_decisions_are_collective():
    """ADVICE r3 (bench.py:397): a rank whose graph capture fails, or whose replay fails its validation, must not leave the
    common sequence of collectives on its own.  comm.CollectiveCapture: rank 0 "cannot capture" the first form, rank 1 "replays
    wrongly" the second, the third works -- both ranks end up with (None, None, form), having matched every collective on the
    way (a mismatch would hang this test until the side group's timeout)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_capture_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (msg, log) for r, msg, log in (q.get(timeout=180) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
    assert all(v[0] == "ok" for v in res.values()), res
    assert any("whole-step capture unavailable (RuntimeError: capture refused here)" in l for l in res[0][1])
    assert any("replayed whole-step step differs" in l for l in res[1][1]) and not res[0][1][1:]
