"""GPU: packed-operand caches must follow raw-pointer writes to the weights.

The fused Adam kernel, the scale kernels and the collective on a FlatState bucket write parameter memory behind
torch's version counters.  The eval-mode caches (fp16 MFMA fragments in engine.py, reparametrised GDN beta/gamma in
layers.py, the packed EB MLP in entropy_models.py) key on the package's weights epoch, which those writers bump.
Shape of the checks: /root/reference/licos/train.py:262-303 (train -> validate in eval mode -> train -> validate) and
federation_utils.py:85 (the model adopts the blend, then trains / validates on).  Every comparison is against the
oracle evaluated on the state_dict the module reports AFTER the write.
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import licos_amd
from licos_amd import federation, ops
from oracle import model as om

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _cpu_state(net):
    return {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}


def _check_against_oracle(net, x, fp16=True):
    """Stage-wise fp32 parity (1e-5, identical bytes on identical latents) + fp16 rate/quality against the oracle run
    on the module's current state_dict.  Returns the fp32 x_hat."""
    net.eval().set_precision("fp32")
    net.update(force=True)
    sd = _cpu_state(net)
    om.eb_update(sd)
    eb = net.entropy_bottleneck
    assert torch.equal(eb._quantized_cdf.cpu(), sd["entropy_bottleneck._quantized_cdf"])
    ref = om.forward(x, sd)
    ref_c = om.compress(x, sd)
    with torch.no_grad():
        y = net.g_a(x.to(DEV))
        assert rel_err(y, ref["y"]) < 1e-5
        y_ref = ref["y"].to(DEV)
        y_hat, lik = eb(y_ref)
        assert torch.equal(y_hat.cpu(), ref["y_hat"])
        rl = ref["likelihoods"]["y"]
        assert bool(((lik.cpu() - rl).abs() <= 1e-5 * rl + 3e-7).all())
        x_hat = net.g_s(y_hat)
        assert rel_err(x_hat, ref["x_hat"]) < 1e-5
        assert eb.compress(y_ref) == ref_c["strings"][0]
        if fp16:
            net.set_precision("fp16")
            out16 = net(x.to(DEV))
            comp16 = net.compress(x.to(DEV))
            dec16 = net.decompress(comp16["strings"], comp16["shape"])
            net.set_precision("fp32")
            bpp16, bpp = licos_amd.metrics.compute_bpp(out16), om.compute_bpp(ref)
            assert abs(bpp16 - bpp) < 5e-3 * bpp, (bpp16, bpp)
            p16 = licos_amd.metrics.compute_psnr(out16["x_hat"].clamp(0, 1), x.to(DEV))
            p = om.compute_psnr(ref["x_hat"].clamp(0, 1), x)
            assert abs(p16 - p) < 0.05, (p16, p)
            assert torch.equal(dec16["x_hat"], out16["x_hat"].clamp(0, 1))
    return x_hat


def _train_step(net, opt, crit, x, seed):
    net.train()
    g = torch.Generator().manual_seed(seed)
    noise = (torch.rand(x.shape[0], 192, x.shape[2] // 16, x.shape[3] // 16, generator=g) - 0.5).to(DEV)
    opt["net"].zero_grad()
    opt["aux"].zero_grad()
    out = net(x.to(DEV), noise=noise)
    res = crit(out, x.to(DEV))
    res["loss"].backward()
    licos_amd.optimizers.clip_grad_norm_(list(net.parameters()), 1.0, opt["net"])
    opt["net"].step()
    aux = net.aux_loss()
    aux.backward()
    opt["aux"].step()


def test_eval_after_fused_adam_steps_sees_the_new_weights():
    sd = om.perturb_state(om.make_factorized_state(3, quality=1, seed=42), seed=13, y_gain=20.0)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    net.load_state_dict(sd)
    net = net.to(DEV)
    x = om.synthetic_tiles(2, 3, 64, seed=3)
    crit = licos_amd.RateDistortionLoss(lmbda=1e-2)
    opt = licos_amd.net_aux_optimizer(net, {"net": {"type": "Adam", "lr": 1e-4}, "aux": {"type": "Adam", "lr": 1e-3}})  # train.py's rates
    assert isinstance(opt["net"], licos_amd.optimizers.FusedAdam)
    x0 = _check_against_oracle(net, x)          # validation 1 fills every cache (fp32 and fp16)
    for rnd in range(2):                        # train -> validate -> train -> validate
        for s in range(2):
            _train_step(net, opt, crit, x, seed=10 * rnd + s)
        x1 = _check_against_oracle(net, x)
        assert bool(torch.isfinite(x1).all())
        assert rel_err(x1, x0) > 1e-4           # the steps did move the output: a stale cache would not show it
        x0 = x1


def test_scaled_flat_state_is_seen_by_the_next_forward():
    sd = om.perturb_state(om.make_factorized_state(3, quality=1, seed=42), seed=5)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    net.load_state_dict(sd)
    net = net.to(DEV)
    x = om.synthetic_tiles(2, 3, 64, seed=1)
    x0 = _check_against_oracle(net, x)
    fs = federation.FlatState(net)
    x0b = _check_against_oracle(net, x)         # re-homing changes storage, not values
    assert rel_err(x0b, x0) < 1e-6
    gain = 1.25  # (2.0 drives this random model's g_s beyond fp16 range, where the split-operand fp32 path does not go)
    ops.scale_f32(fs.flat, gain)                   # every float of the state scaled in place, behind torch's back
    for k, off, n in fs.keys:                   # constants of the model definition (bounds, pedestals, EB target) are
        if k.endswith(("bound", "pedestal", "target")):  # not weights: put them back through the same raw-write kernel
            ops.scale_f32(fs.flat[off:off + n], 1.0 / gain)
    doubled = _cpu_state(net)
    assert torch.allclose(doubled["g_a.2.weight"], gain * sd["g_a.2.weight"])
    assert torch.equal(doubled["g_a.1.beta_reparam.pedestal"], sd["g_a.1.beta_reparam.pedestal"])
    x1 = _check_against_oracle(net, x)
    assert bool(torch.isfinite(x1).all()) and rel_err(x1, x0) > 1e-2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fed_worker(rank, world, port, losses, best, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # two ranks share the one GPU of the test box
    try:
        net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
        sd = om.perturb_state(om.make_factorized_state(3, 1, seed=rank), seed=rank)
        net.load_state_dict(sd)
        net = net.to(DEV).eval()
        net.update(force=True)
        x = om.synthetic_tiles(2, 3, 64, seed=9)
        with torch.no_grad():
            for prec in ("fp32", "fp16"):  # fill every cache with the pre-average operands
                net.set_precision(prec)
                net(x.to(DEV))
                net.compress(x.to(DEV))
        before = _cpu_state(net)
        federation.update_central_model(rank, DEV, 5, net, losses[rank], best[rank], 12.5)
        assert federation.clock_sync(1) == world
        after = _cpu_state(net)
        # the module adopted the blend (federation_utils.py:85) and keeps validating with it (train.py:262-303)
        x_hat = _check_against_oracle(net, x)
        with torch.no_grad():
            net.set_precision("fp16")
            strings16 = list(net.compress(x.to(DEV))["strings"][0])
        torch.save({"before": before, "after": after, "x_hat": x_hat.cpu(), "strings16": strings16}, out.format(rank=rank))
    finally:
        dist.destroy_process_group()


def test_forward_after_two_rank_weighted_average(tmp_path):
    world = 2
    losses, best = [0.9, 0.6], [0.7, 0.6]
    out = str(tmp_path / "r{rank}.pt")
    mp.spawn(_fed_worker, args=(world, _free_port(), losses, best, out), nprocs=world, join=True)
    res = [torch.load(out.format(rank=r), weights_only=False) for r in range(world)]
    float_keys = [k for k, v in res[0]["before"].items() if v.dtype == torch.float32]
    blend = om.sequential_federation([{k: r["before"][k] for k in float_keys} for r in res], losses, best)
    x = om.synthetic_tiles(2, 3, 64, seed=9)
    for r in res:
        for k in float_keys:
            assert torch.allclose(r["after"][k], blend[k].float(), rtol=1e-5, atol=1e-7), k
    # both ranks hold the same central model and (this is the point) evaluate the same thing with it
    assert torch.equal(res[0]["x_hat"], res[1]["x_hat"])
    assert res[0]["strings16"] == res[1]["strings16"]
    assert rel_err(res[0]["after"]["g_a.2.weight"], res[0]["before"]["g_a.2.weight"]) > 1e-2


def _native_worker(rank, world, port, out, schedule="ring"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # only carries the communicator id
    try:
        net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
        sd = om.perturb_state(om.make_factorized_state(3, 1, seed=3), seed=3)
        net.load_state_dict(sd)
        net = net.to(DEV).eval()
        net.update(force=True)
        x = om.synthetic_tiles(1, 3, 64, seed=2)
        with torch.no_grad():
            net.set_precision("fp16")
            y0 = net.g_a(x.to(DEV))
        before = _cpu_state(net)
        with federation.NativeComm() as comm:
            fs = federation.update_central_model(rank, DEV, 1, net, 0.8, 0.5, 0.0, native=comm, schedule=schedule)
            with torch.no_grad():
                y1 = net.g_a(x.to(DEV))
        assert not comm._comm  # destroyed on exit
        torch.save({"before": before, "after": _cpu_state(net), "coef": float(fs.flat[-1]), "dy": rel_err(y1, y0)}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("schedule", ["ring", "direct"])
def test_native_rccl_export_single_rank(tmp_path, schedule):
    """licos_comm_* / licos_allreduce_weighted(_direct) through a real RCCL communicator, both schedules.
    One GPU here, so one rank: the exchange is the identity and the blend must return the model unchanged (to fp32
    rounding of x * c / c), with the coefficient in the bucket's last element and the caches invalidated.  More ranks:
    test_rccl_two_ranks_both_schedules below (needs two GPUs) and the driver's SCALE run (bench.py `fedavg_allreduce`)."""
    out = str(tmp_path / "n.pt")
    mp.spawn(_native_worker, args=(1, _free_port(), out, schedule), nprocs=1, join=True)
    r = torch.load(out, weights_only=False)
    assert abs(r["coef"] - 1.0) < 1e-6 and r["dy"] < 1e-5
    for k, v in r["before"].items():
        if v.dtype == torch.float32:
            assert torch.allclose(r["after"][k], v, rtol=1e-6, atol=1e-12), k


def _rccl_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        res = {}
        with federation.NativeComm() as comm:
            for name, kw in (("torch_ring", {}), ("torch_direct", {"schedule": "direct"}),
                             ("native_ring", {"native": comm}), ("native_direct", {"native": comm, "schedule": "direct"})):
                net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
                net.load_state_dict(om.perturb_state(om.make_factorized_state(3, 1, seed=rank), seed=rank))
                net = net.to(dev).eval()
                federation.update_central_model(rank, dev, 1, net, [0.9, 0.6][rank], [0.7, 0.6][rank], 0.0, **kw)
                res[name] = {k: v.detach().cpu() for k, v in net.state_dict().items() if v.dtype == torch.float32}
        torch.save(res, out.format(rank=rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on one node (RCCL refuses two ranks on one device)")
def test_rccl_two_ranks_both_schedules(tmp_path):
    """The federated blend over RCCL with two ranks on two GPUs: torch.distributed and the library's own communicator,
    ring and direct schedule - all four against the oracle's sequential blend, and identical on both ranks."""
    out = str(tmp_path / "r{rank}.pt")
    mp.spawn(_rccl_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = [torch.load(out.format(rank=r), weights_only=False) for r in range(2)]
    states = [om.perturb_state(om.make_factorized_state(3, 1, seed=r), seed=r) for r in range(2)]
    keys = list(res[0]["torch_ring"])
    ref = om.sequential_federation([{k: s[k] for k in keys} for s in states], [0.9, 0.6], [0.7, 0.6])
    for name in ("torch_ring", "torch_direct", "native_ring", "native_direct"):
        for k in keys:
            assert torch.allclose(res[0][name][k], ref[k].float(), rtol=1e-5, atol=1e-7), (name, k)
            assert torch.equal(res[0][name][k], res[1][name][k]), (name, k)


# ---- the library's own communicator with MORE than one rank on a one-GPU box (test transport) ---------------------------
_FAKE_SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "fake_rccl.cpp")
_FAKE_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "_build", "libfakerccl.so")


def _build_fake_rccl():
    """tests/fake_rccl/fake_rccl.cpp -> _build/libfakerccl.so (g++ against the HIP runtime; test infrastructure)."""
    import subprocess
    if os.path.exists(_FAKE_SO) and os.path.getmtime(_FAKE_SO) >= os.path.getmtime(_FAKE_SRC):
        return _FAKE_SO
    os.makedirs(os.path.dirname(_FAKE_SO), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", _FAKE_SRC,
                           "-o", _FAKE_SO, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-pthread"])
    return _FAKE_SO


_TT_LOSS = [0.9, 0.6, 0.75]
_TT_BEST = [0.7, 0.6, 0.5]


def _test_transport_worker(rank, world, port, out, fake_so):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LICOS_RCCL_TEST_LIB"] = fake_so  # read when the library first needs RCCL (collective.hip)
    os.environ.setdefault("LICOS_FAKE_RCCL_TIMEOUT_S", "45")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # carries the 128-byte id only
    try:
        coef = federation.reference_coefficients(_TT_LOSS[:world], _TT_BEST[:world])[rank]
        res = {}
        with federation.NativeComm() as comm:
            for name, schedule in (("native_ring", "ring"), ("native_direct", "direct"), ("native_direct_again", "direct")):
                net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
                net.load_state_dict(om.perturb_state(om.make_factorized_state(3, 1, seed=rank), seed=rank))
                net = net.to(dev).eval()
                fs = federation.FlatState(net)
                federation.weighted_average_(fs, coef, native=comm, schedule=schedule)
                torch.cuda.synchronize()
                res[name] = {k: v.detach().cpu() for k, v in net.state_dict().items() if v.dtype == torch.float32}
        torch.save(res, out.format(rank=rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_native_communicator_with_several_ranks_over_the_test_transport(tmp_path, world):
    """licos_allreduce_weighted / licos_allreduce_weighted_direct (the replacement of /root/reference/licos/federation_utils.py:
    27-85) with nranks = 2 and 3 on ONE GPU: RCCL refuses two ranks per device, so the library is pointed at a stand-in with
    librccl's C API (tests/fake_rccl: shared-memory mailboxes + hipMemcpy; LICOS_RCCL_TEST_LIB, never set in production).
    What runs for real: the direct schedule's grouped send / recv loop, its chunk offsets (a bucket whose size is not a
    multiple of the world: 3 ranks), the scratch layout, the fixed-order peer reduction, stream ordering against the
    scale / normalise kernels - against the oracle's sequential blend, identical bits on every rank, twice in a row."""
    fake = _build_fake_rccl()
    out = str(tmp_path / "r{rank}.pt")
    mp.spawn(_test_transport_worker, args=(world, _free_port(), out, fake), nprocs=world, join=True)
    res = [torch.load(out.format(rank=r), weights_only=False) for r in range(world)]
    states = [om.perturb_state(om.make_factorized_state(3, 1, seed=r), seed=r) for r in range(world)]
    keys = list(res[0]["native_ring"])
    ref = om.sequential_federation([{k: s[k] for k in keys} for s in states], _TT_LOSS[:world], _TT_BEST[:world])
    for name in ("native_ring", "native_direct", "native_direct_again"):
        for k in keys:
            assert torch.allclose(res[0][name][k], ref[k].float(), rtol=1e-5, atol=1e-7), (name, k)
            for r in range(1, world):
                assert torch.equal(res[0][name][k], res[r][name][k]), (name, k, r)
