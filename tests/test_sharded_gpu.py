"""GPU, two processes (gloo) on one card: the loss sharded over ranks (row slabs of S, 1/W of the bank
products, all-gather of the centrality slices, all-reduce of the row terms) equals the replicated loss."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from neighborretr_amd import modeling, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, Nt, Nv, M, K = 32, 24, 12, 64, 8
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
    m = m.to(dev).train()
    m.config.world_size = world
    p = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(2024, B, Nt, Nv, M).items()}
    nz = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_noise(2024, B, Nt, Nv).items()}
    c = m.config
    args = (p["text_feat"], p["video_feat"], p["text_mask"].float(), p["video_mask"].float(), p["mb_feat_t"], p["mb_feat_v"],
            p["mb_mask_t"].float(), p["mb_mask_v"].float(), c.centrality_scale, c.beta, K, c.temperature,
            torch.tensor(100.0, device=dev))
    with torch.no_grad():
        m.shard_loss = False
        ref = torch.stack(m._compute_losses(*args, noise=nz)).cpu()
        m.shard_loss = True
        got = torch.stack(m._compute_losses(*args, noise=nz)).cpu()
    torch.save({"ref": ref, "got": got}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_loss_equals_replicated_loss(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, 29611
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    for r in res:
        assert torch.isfinite(r["ref"]).all()
        assert torch.allclose(r["got"], r["ref"], rtol=2e-6, atol=2e-6), (r["got"], r["ref"])
    assert torch.equal(res[0]["got"], res[1]["got"])              # every rank holds the same losses
