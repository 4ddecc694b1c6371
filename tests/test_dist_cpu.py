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
