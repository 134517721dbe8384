"""CPU, world_size 2 over gloo: the collective weight average reproduces the reference's sequential
file-based blend (licos/federation_utils.py:47-53) and tile sharding covers the batch exactly."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import licos_amd
from licos_amd import federation
from oracle import model as om


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, losses, best, out, save_path=None, schedule="ring"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
        sd = om.perturb_state(om.make_factorized_state(3, 1, seed=rank), seed=rank)
        net.load_state_dict(sd)
        before = {k: v.clone() for k, v in net.state_dict().items()}
        cfg = None if save_path is None else {"save_path": save_path, "save_checkpoints_over_time": True}
        fs = federation.update_central_model(rank, "cpu", 5, net, losses[rank], best[rank], 12.5, cfg, schedule=schedule)
        assert federation.clock_sync(1) == world
        # parameters are views of the bucket: the module sees the averaged values without a load
        after = {k: v.clone() for k, v in net.state_dict().items()}
        torch.save({"before": before, "after": after, "numel": fs.numel}, out.format(rank=rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,schedule", [(2, "ring"), (2, "direct"), (3, "direct")])
def test_weighted_allreduce_matches_sequential_blend(tmp_path, world, schedule):
    """Both schedules of the blend (one all-reduce; the two-step point-to-point exchange of SURVEY 5.8), world 2 and - for
    the direct one, whose chunking depends on it - a world that does not divide the bucket."""
    losses, best = [0.9, 0.6, 1.2][:world], [0.7, 0.6, 0.8][:world]
    out = str(tmp_path / "r{rank}.pt")
    save_path = str(tmp_path / "central_model")
    mp.spawn(_worker, args=(world, _free_port(), losses, best, out, save_path, schedule), nprocs=world, join=True)
    res = [torch.load(out.format(rank=r)) for r in range(world)]
    float_keys = [k for k, v in res[0]["before"].items() if v.dtype == torch.float32]
    ref = om.sequential_federation([{k: r["before"][k] for k in float_keys} for r in res], losses, best)
    assert res[0]["numel"] == sum(res[0]["before"][k].numel() for k in float_keys)
    for r in res:
        for k in float_keys:
            assert torch.allclose(r["after"][k], ref[k], rtol=1e-5, atol=1e-7), k
            assert torch.equal(r["after"][k], res[0]["after"][k]), k  # every rank holds the same bits
        for k, v in r["before"].items():
            if v.dtype != torch.float32:
                assert torch.equal(r["after"][k], v)  # integer tables untouched


    # the averaged model is left on disk as the reference's "central model" (federation_utils.py:58-83), with its dict
    # layout, plus per-rank over-time checkpoints (licos/utils.py:82-111)
    central = torch.load(save_path + ".pth.tar", weights_only=False)
    assert set(central) == {"batch_idx", "state_dict", "loss", "local_time"} and central["batch_idx"] == 5
    for k in float_keys:
        assert torch.allclose(central["state_dict"][k], ref[k], rtol=1e-5, atol=1e-7), k
    for r in range(world):
        assert os.path.exists(os.path.join(save_path, "central_model_time_checkpoints",
                                           "central_model_rank_%d_sim_time=12.5.pth.tar" % r))


def test_reference_coefficients_closed_form():
    losses, best = [0.9, 0.7, 1.1, 0.6], [0.8, 0.7, 0.9, 0.5]
    coef = federation.reference_coefficients(losses, best)
    assert abs(sum(coef) - 1.0) < 1e-12
    g = torch.Generator().manual_seed(0)
    states = [{"w": torch.randn(64, generator=g)} for _ in range(4)]
    ref = om.sequential_federation(states, losses, best)
    got = sum(c * s["w"].double() for c, s in zip(coef, states))
    assert torch.allclose(got, ref["w"].double(), atol=1e-6)


@pytest.mark.parametrize("total,world", [(1024, 8), (1000, 8), (7, 8), (0, 2), (13, 4)])
def test_shard_range_partitions_exactly(total, world):
    covered = []
    for r in range(world):
        a, b = federation.shard_range(total, r, world)
        assert 0 <= a <= b <= total
        covered += list(range(a, b))
    assert covered == list(range(total))
    sizes = [federation.shard_range(total, r, world)[1] - federation.shard_range(total, r, world)[0] for r in range(world)]
    assert max(sizes) - min(sizes) <= 1


def _tiny_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = torch.nn.Linear(9, 7, bias=False)  # 63 elements + the coefficient = 64: a bucket that is a multiple of 64
        with torch.no_grad():
            net.weight.fill_(float(rank + 1))
        fs = federation.FlatState(net)
        assert fs.flat.numel() % 64 == 0 and fs.flat.numel() % world != 0
        federation.weighted_average_(fs, 1.0 / world, schedule="direct")
        torch.save(net.weight.detach().clone(), out.format(rank=rank))
    finally:
        dist.destroy_process_group()


def test_direct_schedule_on_a_bucket_of_64_with_three_ranks(tmp_path):
    """The direct schedule ships ceil(n / world) elements per peer; for a world that does not divide 64 a bucket padded
    to a multiple of 64 only is too short (n = 64, world = 3 needs 66).  FlatState pads by a further 64."""
    out = str(tmp_path / "t{rank}.pt")
    mp.spawn(_tiny_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    for r in range(3):
        assert torch.allclose(torch.load(out.format(rank=r)), torch.full((7, 9), 2.0), atol=1e-6)
