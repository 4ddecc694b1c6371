"""GPU, two processes (gloo) sharing one card: the exchange step's DEVICE branch -- nr_pack_shard -> one collective ->
nr_unpack_gathered (neighborretr_amd/dist.py) -- end to end, against the byte-exact expectation and against the
pure-torch CPU branch the gloo CPU test covers; with and without caller-supplied static destinations
(`args._gather_out`, what the captured graph reads), and the backward slice.  RCCL itself ("nccl" backend) needs one
GPU per rank and only runs in the driver's multi-GPU bench; everything around the collective call is what runs here."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shard(rank, b, Nt, Nv, d):
    g = torch.Generator().manual_seed(100 + rank)
    tf = torch.randn(b, Nt, d, generator=g)
    vf = torch.randn(b, Nv, d, generator=g)
    idx = torch.arange(b) + 1000 * rank
    tm = (torch.rand(b, Nt, generator=g) > 0.3).long()
    vm = (torch.rand(b, Nv, generator=g) > 0.3).long()
    return tf, vf, idx, tm, vm


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    from types import SimpleNamespace
    import torch.distributed as dist
    from neighborretr_amd.dist import packed_allgather
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    b, Nt, Nv, d = 5, 24, 12, 512                      # 5 samples: record size not a multiple of 16 before padding
    tf, vf, idx, tm, vm = _shard(rank, b, Nt, Nv, d)
    args = SimpleNamespace(world_size=world, local_rank=rank)
    res = {}
    # device branch, fresh outputs
    tfd = tf.to(dev).requires_grad_(True)
    vfd = vf.to(dev).requires_grad_(True)
    out = packed_allgather(tfd, vfd, idx.to(dev), tm.to(dev), vm.to(dev), args)
    res["dev"] = [o.detach().cpu() for o in out]
    w = torch.arange(out[0].numel(), dtype=torch.float32, device=dev).view_as(out[0])
    ((out[0] * w).sum() + 3.0 * out[1].sum()).backward()
    res["grad_t"], res["grad_v"], res["w"] = tfd.grad.cpu(), vfd.grad.cpu(), w.cpu()
    # device branch writing into static destinations
    static = tuple(torch.full_like(o, -7) for o in out)
    args2 = SimpleNamespace(world_size=world, local_rank=rank, _gather_out=static)
    with torch.no_grad():
        out2 = packed_allgather(tf.to(dev), vf.to(dev), idx.to(dev), tm.to(dev), vm.to(dev), args2)
    res["same_storage"] = all(o.data_ptr() == s.data_ptr() for o, s in zip(out2, static))
    res["static"] = [s.cpu() for s in static]
    # the pure-torch CPU branch of the same function, same collective backend
    with torch.no_grad():
        res["cpu"] = list(packed_allgather(tf, vf, idx, tm, vm, args))
    torch.save(res, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_packed_allgather_device_branch_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, 29617
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    b, Nt, Nv, d = 5, 24, 12, 512
    shards = [_shard(r, b, Nt, Nv, d) for r in range(world)]
    want = [torch.cat([s[k] for s in shards]) for k in range(5)]
    want[3], want[4] = want[3].float(), want[4].float()
    for rank in range(world):
        res = torch.load(f"{out}.{rank}")
        for name in ("dev", "static", "cpu"):
            for got, ref in zip(res[name], want):
                assert got.dtype == ref.dtype and got.shape == ref.shape, (name, got.dtype, ref.dtype)
                assert torch.equal(got, ref), name                   # byte-exact
        assert res["same_storage"]
        sl = slice(rank * b, (rank + 1) * b)
        assert torch.equal(res["grad_t"], res["w"][sl])              # backward = this rank's slice, no reduction
        assert torch.equal(res["grad_v"], torch.full((b, Nv, d), 3.0))


def test_nr_allgather_packed_with_a_real_rccl_communicator():
    """The C-ABI exchange step (nr_allgather_packed) on RCCL itself: a ONE-rank communicator is all a one-GPU box
    can build (ncclCommInitRank through ctypes on the librccl the process already holds), which still runs the real
    pack kernel -> ncclAllGather -> unpack kernel on the stream."""
    import ctypes
    sys.path.insert(0, ROOT)
    from neighborretr_amd import hip
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        dev = "cuda"
        b, Nt, Nv, d = 5, 24, 12, 512
        tf, vf, idx, tm, vm = _shard(0, b, Nt, Nv, d)
        pieces = [tf.to(dev), vf.to(dev), idx.to(dev), tm.to(torch.uint8).to(dev), vm.to(torch.uint8).to(dev)]
        sizes = [p.numel() * p.element_size() for p in pieces]
        offs = [sum(sizes[:k]) for k in range(5)]
        total = (sum(sizes) + 15) // 16 * 16
        packed = torch.empty(total, dtype=torch.uint8, device=dev)
        gathered = torch.empty(total, dtype=torch.uint8, device=dev)
        outs = [torch.empty_like(pieces[0]), torch.empty_like(pieces[1]), torch.empty_like(pieces[2]),
                torch.empty(tm.shape, dtype=torch.float32, device=dev), torch.empty(vm.shape, dtype=torch.float32, device=dev)]
        P5, Z5, I5 = ctypes.c_void_p * 5, ctypes.c_size_t * 5, ctypes.c_int * 5
        hip.call("nr_allgather_packed", comm, 1, 5, P5(*[p.data_ptr() for p in pieces]), Z5(*sizes), Z5(*offs), total,
                 hip.ptr(packed), hip.ptr(gathered), P5(*[o.data_ptr() for o in outs]), I5(0, 0, 0, 1, 1), hip.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(outs[0].cpu(), tf) and torch.equal(outs[1].cpu(), vf) and torch.equal(outs[2].cpu(), idx)
        assert torch.equal(outs[3].cpu(), tm.float()) and torch.equal(outs[4].cpu(), vm.float())
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)
