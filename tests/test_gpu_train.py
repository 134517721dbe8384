"""GPU: one training step in the shape of licos/train.py:186-200 (forward with noise, RD loss, backward,
clip, Adam; aux loss on the quantiles) - HIP forward AND backward kernels strung together by licos_amd/autograd.py -
against the oracle differentiated by torch on the CPU."""
import pytest
import torch

import licos_amd
from oracle import model as om

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _LowerBoundRef(torch.autograd.Function):
    """[CAI] ops/bound_ops.py: max(x, bound) whose gradient passes where x >= bound or grad < 0 (test reference)."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        return ((x >= bound) | (g < 0)) * g, None


def lower_bound_ref(x, bound):
    return _LowerBoundRef.apply(x, bound)


def test_train_step_matches_cpu_autograd():
    sd = om.perturb_state(om.make_factorized_state(3, quality=1, seed=42), seed=13, y_gain=20.0)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    x = om.synthetic_tiles(2, 3, 64, seed=3)
    g = torch.Generator().manual_seed(0)
    noise = torch.rand(2, 192, 4, 4, generator=g) - 0.5
    crit = licos_amd.RateDistortionLoss(lmbda=1e-2)
    conf = {"net": {"type": "Adam", "lr": 1e-4}, "aux": {"type": "Adam", "lr": 1e-3}}
    opt = licos_amd.net_aux_optimizer(net, conf)
    opt["net"].zero_grad()
    opt["aux"].zero_grad()
    out = net(x.to(DEV), noise=noise.to(DEV))
    res = crit(out, x.to(DEV))
    res["loss"].backward()
    # CPU reference: the oracle's forward under torch autograd
    ref_sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and v.dim() > 0 and "bound" not in k
                  and "pedestal" not in k and "target" not in k else v) for k, v in sd.items()}
    ref_out = om.forward(x, ref_sd, training=True, noise=noise)
    ref_res = om.rate_distortion_loss(ref_out, x, 1e-2)
    ref_res["loss"].backward()
    assert abs(float(res["loss"].detach()) - float(ref_res["loss"].detach())) < 1e-4 * abs(float(ref_res["loss"].detach()))
    checked = 0
    for name, p in net.named_parameters():
        rg = ref_sd[name].grad
        if rg is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        err = float((p.grad.cpu() - rg).abs().max() / rg.abs().max().clamp_min(1e-20))
        assert err < 2e-3, (name, err)
        checked += 1
    assert checked >= 40
    torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
    before = net.g_a[0].weight.detach().clone()
    opt["net"].step()
    assert not torch.equal(before, net.g_a[0].weight.detach())
    aux = net.aux_loss()
    aux.backward()
    assert net.entropy_bottleneck.quantiles.grad is not None
    opt["aux"].step()
    # the fp16 path refuses autograd loudly
    net.set_precision("fp16")
    with pytest.raises(NotImplementedError):
        net(x.to(DEV))
    # ... but only in train() mode: in eval() mode, forward / compress / decompress run without a graph even when the
    # caller forgot torch.no_grad() (CompressAI's compress() works that way too)
    net.eval()
    net.update(force=True)
    out = net(x.to(DEV))
    assert out["x_hat"].grad_fn is None
    comp = net.compress(x.to(DEV))
    dec = net.decompress(comp["strings"], comp["shape"])["x_hat"]
    assert torch.equal(dec, out["x_hat"].clamp(0, 1))


@pytest.mark.parametrize("cin,cout,h,w,k,s", [(3, 16, 20, 28, 5, 2), (16, 24, 17, 33, 5, 2), (8, 8, 9, 9, 3, 1),
                                               (32, 64, 16, 16, 5, 2), (130, 70, 8, 8, 5, 2)])
def test_conv_backward_kernels(cin, cout, h, w, k, s):
    """dgrad (transposed-conv kernel), wgrad and bias-grad kernels against torch CPU autograd."""
    from licos_amd import autograd
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(3, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(cout, generator=g, requires_grad=True)
    ref = torch.nn.functional.conv2d(x, wt, b, stride=s, padding=k // 2)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, wt, b))
    out = autograd.ConvHip.apply(xd, wd, bd, s, k // 2)
    out.backward(go.to(DEV))
    for got, want in ((xd.grad, x.grad), (wd.grad, wt.grad), (bd.grad, b.grad)):
        assert float((got.cpu() - want).abs().max() / want.abs().max()) < 2e-5


@pytest.mark.parametrize("cin,cout,h,w", [(16, 3, 10, 14), (24, 16, 9, 17), (192, 40, 4, 4)])
def test_deconv_backward_kernels(cin, cout, h, w):
    from licos_amd import autograd
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(2, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cin, cout, 5, 5, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(cout, generator=g, requires_grad=True)
    ref = torch.nn.functional.conv_transpose2d(x, wt, b, stride=2, padding=2, output_padding=1)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, wt, b))
    out = autograd.DeconvHip.apply(xd, wd, bd, 2, 2, 1)
    out.backward(go.to(DEV))
    for got, want in ((xd.grad, x.grad), (wd.grad, wt.grad), (bd.grad, b.grad)):
        assert float((got.cpu() - want).abs().max() / want.abs().max()) < 2e-5


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("c,scale", [(128, 2.0), (192, 2.0), (128, 300.0), (64, 2.0)])
def test_gdn_backward_kernel(inverse, c, scale):
    """128 / 192 channels take the matrix-core route (norm and gamma^T.t as split-operand 1x1 products, the norm operand
    pre-scaled as (x/16)^2 so that activations in the hundreds stay inside fp16); other widths the direct kernel."""
    g = torch.Generator().manual_seed(5)
    sd = {}
    om._gdn_init(sd, "g.", c)
    sd["g.gamma"] = (sd["g.gamma"] + 0.05 * torch.rand(c, c, generator=g))
    sd["g.beta"] = sd["g.beta"] * (0.5 + torch.rand(c, generator=g))
    x = (scale * torch.randn(2, c, 9, 11, generator=g)).requires_grad_(True)
    ref_sd = dict(sd)
    ref_sd["g.gamma"] = sd["g.gamma"].clone().requires_grad_(True)
    ref_sd["g.beta"] = sd["g.beta"].clone().requires_grad_(True)
    ref = om.gdn(x, ref_sd, "g.", inverse=inverse)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    m = licos_amd.GDN(c, inverse=inverse)
    m.load_state_dict({k[2:]: v for k, v in sd.items()})
    m = m.to(DEV)
    xd = x.detach().to(DEV).requires_grad_(True)
    out = m(xd)
    out.backward(go.to(DEV))
    for got, want in ((xd.grad, x.grad), (m.gamma.grad, ref_sd["g.gamma"].grad), (m.beta.grad, ref_sd["g.beta"].grad)):
        assert float((got.cpu() - want).abs().max() / want.abs().max()) < 5e-5


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("scale,gmag", [(2.0, 1.0), (300.0, 1.0), (2.0, 1e-9), (0.05, 1e-4)])
def test_gdn_backward_one_pass_kernel(inverse, scale, gmag, monkeypatch):
    """128 channels over whole 32-pixel tiles: forward keeps the norm, backward is ONE kernel (mfma_gdn_bwd_f32.hip:
    t, gamma^T . t on the matrix cores with a per-pixel power-of-two scale, dx).  Against float64 autograd, for
    activations of 0.05 .. 300 and output gradients down to 1e-9; and against the layer-by-layer backward."""
    from licos_amd import autograd as ag
    c = 128
    g = torch.Generator().manual_seed(7)
    sd = {}
    om._gdn_init(sd, "g.", c)
    sd["g.gamma"] = (sd["g.gamma"] + 0.05 * torch.rand(c, c, generator=g))
    sd["g.beta"] = sd["g.beta"] * (0.5 + torch.rand(c, generator=g))
    x = scale * torch.randn(2, c, 8, 16, generator=g)
    go = gmag * torch.randn(2, c, 8, 16, generator=g)
    ref_sd = {k: v.double() for k, v in sd.items()}
    ref_sd["g.gamma"].requires_grad_(True)
    ref_sd["g.beta"].requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    om.gdn(x64, ref_sd, "g.", inverse=inverse).backward(go.double())
    grads = {}
    for fused in (True, False):
        monkeypatch.setattr(ag, "GDN_BWD_FUSED", fused)
        m = licos_amd.GDN(c, inverse=inverse)
        m.load_state_dict({k[2:]: v for k, v in sd.items()})
        m = m.to(DEV)
        xd = x.to(DEV).requires_grad_(True)
        out = m(xd)
        out.backward(go.to(DEV))
        grads[fused] = (xd.grad.cpu(), m.gamma.grad.cpu(), m.beta.grad.cpu())
    for got, want in zip(grads[True], (x64.grad, ref_sd["g.gamma"].grad, ref_sd["g.beta"].grad)):
        assert float((got.double() - want).abs().max() / want.abs().max()) < 2e-5
    if gmag >= 1e-4:  # (the layer-by-layer form feeds t to fp16 unscaled: below ~1e-6 its gamma^T . t underflows)
        for a, b in zip(grads[True], grads[False]):
            assert float((a - b).abs().max() / b.abs().max()) < 5e-5


def test_fused_adam_and_clip_match_torch():
    from licos_amd import optimizers
    g = torch.Generator().manual_seed(1)
    ps = [torch.randn(1000, generator=g), torch.randn(7, 13, generator=g)]
    grads = [[torch.randn(p.shape, generator=g) * (3.0 if it == 0 else 0.01) for p in ps] for it in range(3)]
    ref_p = [p.clone().requires_grad_(True) for p in ps]
    ref_opt = torch.optim.Adam(ref_p, lr=1e-2)
    dev_p = [p.clone().to(DEV).requires_grad_(True) for p in ps]
    opt = optimizers.FusedAdam(dev_p, lr=1e-2)
    for it in range(3):
        for p, q, gr in zip(ref_p, dev_p, grads[it]):
            p.grad = gr.clone()
            q.grad = gr.clone().to(DEV)
        n_ref = torch.nn.utils.clip_grad_norm_(ref_p, 1.0)
        n_dev = optimizers.clip_grad_norm_(dev_p, 1.0, optimizer=opt)
        assert abs(float(n_ref) - float(n_dev)) < 1e-4 * float(n_ref)
        ref_opt.step()
        opt.step()
        for p, q in zip(ref_p, dev_p):
            assert torch.allclose(p.detach(), q.detach().cpu(), rtol=1e-5, atol=1e-7)
    # without an optimizer the gradients themselves are rescaled, like torch's
    for q, gr in zip(dev_p, grads[0]):
        q.grad = gr.clone().to(DEV)
    optimizers.clip_grad_norm_(dev_p, 1.0)
    tot = sum(float((q.grad.double() ** 2).sum()) for q in dev_p) ** 0.5
    assert abs(tot - 1.0) < 1e-3


def test_gaussian_conditional_gradients_match_cpu_autograd():
    """Scale-hyperprior training: the y likelihood must carry gradients to the latents AND to the predicted scales
    (h_s), with CompressAI's LowerBound gradient rule on the scale and likelihood bounds."""
    g = torch.Generator().manual_seed(4)
    y = (3.0 * torch.randn(2, 8, 6, 5, generator=g)).requires_grad_(True)
    scales = (torch.rand(2, 8, 6, 5, generator=g) * 2.0 + 0.02).requires_grad_(True)  # some below the 0.11 bound
    noise = torch.rand(2, 8, 6, 5, generator=g) - 0.5
    gc = licos_amd.GaussianConditional(None).to(DEV).train()
    yd, sd_ = y.detach().to(DEV).requires_grad_(True), scales.detach().to(DEV).requires_grad_(True)
    out, lik = gc(yd, sd_, noise=noise.to(DEV))
    (torch.log2(lik).sum() + 0.1 * out.sum()).backward()

    s = lower_bound_ref(scales, torch.tensor([0.11]))
    v = torch.abs(y + noise)
    c = -(2 ** -0.5)
    ref = 0.5 * torch.erfc(c * ((0.5 - v) / s)) - 0.5 * torch.erfc(c * ((-0.5 - v) / s))
    ref = lower_bound_ref(ref, torch.tensor([1e-9]))
    (torch.log2(ref).sum() + 0.1 * (y + noise).sum()).backward()
    assert torch.allclose(lik.detach().cpu(), ref.detach(), rtol=2e-5, atol=1e-9)
    for a, b in ((yd.grad.cpu(), y.grad), (sd_.grad.cpu(), scales.grad)):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max())


def test_hyperprior_training_reaches_every_parameter():
    net = licos_amd.get_model("bmshj2018-hyperprior", False, 3, 1).to(DEV).train()
    x = torch.rand(2, 3, 128, 128, device=DEV)
    out = net(x)
    res = licos_amd.RateDistortionLoss(lmbda=1e-2)(out, x)
    res["loss"].backward()
    missing = [n for n, p in net.named_parameters() if p.requires_grad and p.grad is None and not n.endswith("quantiles")]
    assert missing == [], missing
    assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.grad is not None)
    for name in ("g_a.0.weight", "h_a.0.weight", "h_s.0.weight", "g_s.0.weight"):
        assert float(dict(net.named_parameters())[name].grad.abs().max()) > 0.0, name


@pytest.mark.parametrize("cin,cout,h,w,batch", [(128, 128, 32, 32, 16), (13, 128, 40, 24, 5), (128, 3, 16, 16, 20), (192, 320, 8, 8, 16)])
def test_mfma_weight_gradient_accuracy_and_reproducibility(cin, cout, h, w, batch):
    """5x5 stride-2 weight gradient on the matrix cores (three split-operand passes, per-strip partial sums added in a
    fixed order): against float64 as good as the fp32 VALU kernel, and bit-identical from run to run (no atomics)."""
    from licos_amd import ops
    g = torch.Generator().manual_seed(cin + cout + batch)
    x = torch.randn(batch, cin, h, w, generator=g)
    dy = torch.randn(batch, cout, h // 2, w // 2, generator=g)
    wt = torch.zeros(cout, cin, 5, 5, dtype=torch.float64, requires_grad=True)
    F = torch.nn.functional
    (F.conv2d(x.double(), wt, None, stride=2, padding=2) * dy.double()).sum().backward()
    ref = wt.grad
    saved = ops.WGRAD_MFMA
    try:
        ops.WGRAD_MFMA = True
        a = ops.conv2d_wgrad_f32(x.to(DEV), dy.to(DEV), cin, cout, 5, 2, 2)
        b = ops.conv2d_wgrad_f32(x.to(DEV), dy.to(DEV), cin, cout, 5, 2, 2)
        ops.WGRAD_MFMA = False
        v = ops.conv2d_wgrad_f32(x.to(DEV), dy.to(DEV), cin, cout, 5, 2, 2)
    finally:
        ops.WGRAD_MFMA = saved
    assert torch.equal(a, b)
    den = float(ref.abs().max())
    ea, ev = (float((t.cpu().double() - ref).abs().max()) / den for t in (a, v))
    assert ea < 3e-6 and ea <= 2 * ev + 5e-7, (ea, ev)


def test_training_converges_with_the_repos_own_step():
    """licos/train.py:186-200's recipe (lambda 1e-2, Adam 1e-4 / aux Adam 1e-3, clip 1.0, batches of 16) on seeded synthetic
    tiles, every kernel of the step the repo's own: 300 steps must lower the RD loss by far more than 30 % and the aux loss
    must fall.  (The full run behind licos_amd/weights/ is tools/train_weights.py; its curve is profiles/r02_train_q3_c3.jsonl.)"""
    from licos_amd import synthetic
    torch.manual_seed(42)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(DEV).train()
    crit = licos_amd.RateDistortionLoss(lmbda=1e-2)
    opt = licos_amd.net_aux_optimizer(net, {"net": {"type": "Adam", "lr": 1e-4}, "aux": {"type": "Adam", "lr": 1e-3}})
    first = last = aux_first = aux_last = None
    for step in range(300):
        x = synthetic.tiles(16, 3, 128, seed=5000 + step, device=DEV)
        opt["net"].zero_grad()
        opt["aux"].zero_grad()
        res = crit(net(x), x)
        res["loss"].backward()
        licos_amd.optimizers.clip_grad_norm_(list(net.parameters()), 1.0, opt["net"])
        opt["net"].step()
        aux = net.aux_loss()
        aux.backward()
        opt["aux"].step()
        if step < 5:
            first = float(res["loss"].detach()) if first is None else max(first, float(res["loss"].detach()))
            aux_first = float(aux.detach()) if aux_first is None else aux_first
        if step >= 295:
            last = float(res["loss"].detach()) if last is None else max(last, float(res["loss"].detach()))
            aux_last = float(aux.detach())
    assert all(bool(torch.isfinite(p).all()) for p in net.parameters())
    assert last < 0.7 * first, (first, last)
    assert last < 0.2 * first, (first, last)  # measured: 138 -> ~8 in 300 steps
    assert aux_last < aux_first, (aux_first, aux_last)


def test_shipped_operating_point_codes_images():
    """licos_amd/weights/factorized_q3_c3.pth.tar (trained by tools/train_weights.py with the step above): a real
    rate / distortion point on held-out tiles, through the fp16 codec and against the oracle on the same weights."""
    import os
    from licos_amd import checkpoint, synthetic
    path = os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q3_c3.pth.tar")
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(DEV).eval()
    meta = checkpoint.load_checkpoint(path, net)
    assert "recipe" in meta and meta["batch_idx"] >= 10000
    x = synthetic.tiles(8, 3, 256, seed=100, device=DEV)
    net.set_precision("fp16")
    with torch.no_grad():
        out = net(x)
        c = net.compress(x)
        d = net.decompress(c["strings"], c["shape"])
    psnr = licos_amd.metrics.compute_psnr(d["x_hat"], x)
    bpp = 8.0 * sum(len(s) for s in c["strings"][0]) / (8 * 256 * 256)
    assert psnr > 30.0 and 0.05 < bpp < 0.6, (psnr, bpp)
    sd = {k: v.detach().cpu().float() if v.dtype.is_floating_point else v.detach().cpu() for k, v in net.state_dict().items()}
    ref = om.forward(x.cpu(), sd)
    assert abs(licos_amd.metrics.compute_bpp(out) - om.compute_bpp(ref)) < 2e-3 * om.compute_bpp(ref)
    assert abs(licos_amd.metrics.compute_psnr(out["x_hat"].clamp(0, 1), x) - om.compute_psnr(ref["x_hat"].clamp(0, 1), x.cpu())) < 0.02
    # escapes are rare on the trained distribution
    y = net.g_a(x)
    eb = net.entropy_bottleneck
    sym = torch.round(y - eb.medians_vec().reshape(1, -1, 1, 1)).int() - eb._offset.reshape(1, -1, 1, 1)
    esc = ((sym < 0) | (sym >= (eb._cdf_length - 2).reshape(1, -1, 1, 1))).float().mean()
    assert float(esc) < 1e-3, float(esc)


@pytest.mark.parametrize("cin,kind,min_psnr", [(1, "s2", 36.0), (13, "s2-merged", 33.0)])
def test_shipped_q5_points_of_the_sentinel_configs(cin, kind, min_psnr):
    """licos_amd/weights/factorized_q5_c{1,13}.pth.tar (BASELINE configs[2]: q = 5 on single bands / the 13 merged bands,
    trained by tools/train_weights.py at lambda 0.025): held-out tiles through the fp16 codec - whose first stage for 13
    bands reads the NCHW fp32 image in place - against the oracle on the same weights."""
    import os
    from licos_amd import checkpoint, synthetic
    path = os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q5_c%d.pth.tar" % cin)
    net = licos_amd.get_model("bmshj2018-factorized", False, cin, 5).to(DEV).eval()
    meta = checkpoint.load_checkpoint(path, net)
    assert "recipe" in meta and meta["batch_idx"] >= 4000
    x = synthetic.tiles(4, cin, 256, seed=100, kind=kind, device=DEV)
    net.set_precision("fp16")
    with torch.no_grad():
        out = net(x)
        c = net.compress(x)
        d = net.decompress(c["strings"], c["shape"])
    psnr = licos_amd.metrics.compute_psnr(d["x_hat"], x)
    bpp = 8.0 * sum(len(s) for s in c["strings"][0]) / (4 * 256 * 256)
    assert psnr > min_psnr and 0.02 < bpp < 1.5, (psnr, bpp)
    sd = {k: v.detach().cpu().float() if v.dtype.is_floating_point else v.detach().cpu() for k, v in net.state_dict().items()}
    ref = om.forward(x.cpu(), sd)
    assert abs(licos_amd.metrics.compute_bpp(out) - om.compute_bpp(ref)) < 2e-3 * om.compute_bpp(ref)
    assert abs(licos_amd.metrics.compute_psnr(out["x_hat"].clamp(0, 1), x) - om.compute_psnr(ref["x_hat"].clamp(0, 1), x.cpu())) < 0.02


def _real_crops():
    import os
    import numpy as np
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_crops.npz")
    g = np.load(path)
    return torch.from_numpy(g["x_u8"].astype(np.float32) / 255.0), [str(n) for n in g["names"]]


def test_shipped_operating_point_codes_real_photos():
    """The same weights (fine-tuned on crops of the reference's own test photos, tools/train_weights.py --data mix) on
    tests/golden/real_crops.npz - 256 x 256 crops of licos/tests/test_data/train/*.jpg, the data train.py:67-72 reads:
    rate / distortion of the fp16 codec, agreement with the oracle on the same weights, escape rate on real texture."""
    import os
    from licos_amd import checkpoint
    path = os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q3_c3.pth.tar")
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(DEV).eval()
    meta = checkpoint.load_checkpoint(path, net)
    assert "photos" in meta["recipe"]
    x_all, names = _real_crops()
    x = x_all.to(DEV)
    net.set_precision("fp16")
    with torch.no_grad():
        out = net(x)
        c = net.compress(x)
        d = net.decompress(c["strings"], c["shape"])
    n = x.shape[0]
    psnr = licos_amd.metrics.compute_psnr(d["x_hat"], x)
    bpp = 8.0 * sum(len(s) for s in c["strings"][0]) / (n * 256 * 256)
    print(f"real crops ({n}): {bpp:.4f} bpp, {psnr:.2f} dB")
    assert psnr > 29.0 and 0.05 < bpp < 1.5, (psnr, bpp)
    assert float((d["x_hat"] - out["x_hat"].clamp(0, 1)).abs().max()) < 1e-5
    sd = {k: v.detach().cpu().float() if v.dtype.is_floating_point else v.detach().cpu() for k, v in net.state_dict().items()}
    ref = om.forward(x_all[:8], sd)
    out8 = {"x_hat": out["x_hat"][:8], "likelihoods": {"y": out["likelihoods"]["y"][:8]}}
    assert abs(licos_amd.metrics.compute_bpp(out8) - om.compute_bpp(ref)) < 3e-3 * om.compute_bpp(ref)
    assert abs(licos_amd.metrics.compute_psnr(out8["x_hat"].clamp(0, 1), x[:8]) - om.compute_psnr(ref["x_hat"].clamp(0, 1), x_all[:8])) < 0.03
    y = net.g_a(x)
    eb = net.entropy_bottleneck
    sym = torch.round(y - eb.medians_vec().reshape(1, -1, 1, 1)).int() - eb._offset.reshape(1, -1, 1, 1)
    esc = ((sym < 0) | (sym >= (eb._cdf_length - 2).reshape(1, -1, 1, 1))).float().mean()
    print(f"escape rate on real crops: {float(esc):.2e}")
    assert float(esc) < 5e-3, float(esc)


@pytest.mark.parametrize("cin,size,weights", [(13, 512, "hyperprior_q5_c13.pth.tar"), (3, 256, "hyperprior_q5_c3.pth.tar")])
def test_shipped_hyperprior_points_match_the_oracle(cin, size, weights, monkeypatch):
    """BASELINE configs[4]'s model with weights trained by the repo's own step (tools/train_weights.py --model
    bmshj2018-hyperprior): the fp16 codec's rate / quality against the oracle on the same weights, 13-band 512 x 512
    synthetic tiles and real-photo crops (3-channel point), through the large-batch pipeline (device coder)."""
    import os
    from licos_amd import checkpoint, ops, synthetic
    monkeypatch.setattr(ops, "HOST_CODER", "0")
    path = os.path.join(os.path.dirname(licos_amd.__file__), "weights", weights)
    net = licos_amd.get_model("bmshj2018-hyperprior", False, cin, 5).to(DEV).eval()
    meta = checkpoint.load_checkpoint(path, net)
    assert "recipe" in meta and meta["batch_idx"] >= 5000
    if cin == 3:
        x = _real_crops()[0][:6].to(DEV)
    else:
        x = synthetic.tiles(3, cin, size, seed=100, kind="s2-merged", device=DEV)
    net.set_precision("fp16")
    net.chunk = 2  # two pipeline chunks
    with torch.no_grad():
        out = net(x)
        c = net.compress(x)
        d = net.decompress(c["strings"], c["shape"])
    n = x.shape[0]
    psnr = licos_amd.metrics.compute_psnr(d["x_hat"], x)
    bpp = 8.0 * sum(len(s) for lst in c["strings"] for s in lst) / (n * size * size)
    bpp_lik = licos_amd.metrics.compute_bpp(out)
    print(f"hyperprior {cin}ch: {bpp:.4f} bpp coded ({bpp_lik:.4f} from likelihoods), {psnr:.2f} dB")
    # (the 3-channel point saw 4 minutes of training; its crops are the hardest photo textures of the fixture)
    assert psnr > (30.0 if cin == 13 else 26.0) and 0.05 < bpp < 2.0 and abs(bpp - bpp_lik) < 0.05 * bpp_lik + 0.01
    assert float((d["x_hat"] - out["x_hat"].clamp(0, 1)).abs().max()) < 1e-5
    sd = {k: v.detach().cpu().float() if v.dtype.is_floating_point else v.detach().cpu() for k, v in net.state_dict().items()}
    ref = om.hyper_forward(x[:2].cpu(), sd)
    out2 = {"x_hat": out["x_hat"][:2], "likelihoods": {k: v[:2] for k, v in out["likelihoods"].items()}}
    assert abs(licos_amd.metrics.compute_bpp(out2) - om.compute_bpp(ref)) < 5e-3 * om.compute_bpp(ref)
    assert abs(licos_amd.metrics.compute_psnr(out2["x_hat"].clamp(0, 1), x[:2]) - om.compute_psnr(ref["x_hat"].clamp(0, 1), x[:2].cpu())) < 0.05


@pytest.mark.parametrize("filters,form", [((3, 3, 3, 3), "plain"), ((1, 1, 3, 3), "plain"), ((13, 13, 3, 3), "plain"),
                                          ((3, 3, 3, 3), "signflip"), ((5, 2, 4), "plain")])
def test_entropy_bottleneck_backward_kernel(filters, form):
    """licos_eb_likelihood_bwd (per-channel MLP at v -+ 1/2, analytic backward, in-kernel reductions over the batch,
    LowerBound gradient rule) against torch autograd of CompressAI's definition on the CPU - for the three filter tuples
    LICOS instantiates (model_utils.py:25-29), the sign-flip likelihood form and a generic shape."""
    import torch.nn.functional as F
    c, b, h, w = 7, 5, 9, 11  # 495 elements per channel: one slice, ragged last wave
    torch.manual_seed(len(filters) + filters[0])
    eb = licos_amd.EntropyBottleneck(c, filters=filters, likelihood_form=form)
    with torch.no_grad():
        for p in list(eb.matrices) + list(eb.biases) + list(eb.factors):
            p.add_(0.3 * torch.randn_like(p))
    eb = eb.to(DEV).train()
    g = torch.Generator().manual_seed(1)
    x = 4.0 * torch.randn(b, c, h, w, generator=g)
    x[0, 0, 0, :4] = torch.tensor([60.0, -60.0, 45.0, -45.0])  # far tails: likelihood at the 1e-9 bound
    noise = torch.rand(b, c, h, w, generator=g) - 0.5
    wgt = torch.randn(b, c, h, w, generator=g)                 # random upstream gradient, both signs
    xd = x.to(DEV).requires_grad_(True)
    out, lik = eb(xd, noise=noise.to(DEV))
    ((wgt.to(DEV) * lik).sum() + 0.1 * out.sum()).backward()
    # CPU reference
    params = [p.detach().cpu().clone().requires_grad_(True) for p in list(eb.matrices) + list(eb.biases) + list(eb.factors)]
    nl = len(filters) + 1
    xr = x.clone().requires_grad_(True)
    outs = xr + noise
    v = outs.transpose(0, 1).reshape(c, 1, -1)

    def logits(t):
        for i in range(nl):
            t = torch.matmul(F.softplus(params[i]), t) + params[nl + i]
            if i < nl - 1:
                t = t + torch.tanh(params[2 * nl + i]) * torch.tanh(t)
        return t

    lo, up = logits(v - 0.5), logits(v + 0.5)
    if form == "plain":
        ref = torch.sigmoid(up) - torch.sigmoid(lo)
    else:
        sign = -torch.sign(lo + up).detach()
        ref = torch.abs(torch.sigmoid(sign * up) - torch.sigmoid(sign * lo))
    ref = lower_bound_ref(ref, torch.tensor([1e-9]))
    ref = ref.reshape(c, b, h, w).transpose(0, 1)
    ((wgt * ref).sum() + 0.1 * outs.sum()).backward()
    # (far positive tail, plain form: sigmoid(up) - sigmoid(lo) cancels to fp32 noise ~6e-8 on either side)
    assert bool(((lik.detach().cpu() - ref.detach()).abs() <= 2e-5 * ref.detach() + 3e-7).all())
    assert float((xd.grad.cpu() - xr.grad).abs().max()) <= 2e-4 * float(xr.grad.abs().max())
    for p, r in zip(list(eb.matrices) + list(eb.biases) + list(eb.factors), params):
        assert p.grad is not None and p.grad.shape == r.grad.shape
        assert float((p.grad.cpu() - r.grad).abs().max()) <= 3e-4 * float(r.grad.abs().max()) + 1e-7


def test_entropy_bottleneck_backward_slices_are_deterministic():
    """Many elements per channel -> several slices per channel; two runs give bit-identical gradients."""
    eb = licos_amd.EntropyBottleneck(192, filters=(3, 3, 3, 3)).to(DEV).train()
    g = torch.Generator().manual_seed(0)
    x = (3.0 * torch.randn(16, 192, 16, 16, generator=g)).to(DEV)
    noise = (torch.rand(16, 192, 16, 16, generator=g) - 0.5).to(DEV)
    grads = []
    for _ in range(2):
        eb.zero_grad()
        xd = x.clone().requires_grad_(True)
        _, lik = eb(xd, noise=noise)
        torch.log2(lik).sum().backward()
        grads.append([xd.grad.clone()] + [p.grad.clone() for p in eb.parameters() if p.grad is not None])
    assert len(grads[0]) > 10 and all(torch.equal(a, b) for a, b in zip(*grads))


@pytest.mark.parametrize("transposed", [False, True])
def test_relu_and_abs_masks_in_the_conv_backward(transposed):
    """The hyper transforms' fused ReLU (h_a, h_s) and |y| input (h_a[0]): HIP forward and backward against torch autograd."""
    import torch.nn.functional as F
    from licos_amd import autograd
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 24, 10, 14, generator=g)
    w = 0.1 * torch.randn(24, 16, 5, 5, generator=g) if transposed else 0.1 * torch.randn(16, 24, 5, 5, generator=g)
    b = torch.randn(16, generator=g)
    xs = [t.clone().requires_grad_(True) for t in (x, w, b)]
    xd = [t.clone().to(DEV).requires_grad_(True) for t in (x, w, b)]
    if transposed:
        ref = F.relu(F.conv_transpose2d(xs[0], xs[1], xs[2], stride=2, padding=2, output_padding=1))
        out = autograd.DeconvHip.apply(xd[0], xd[1], xd[2], 2, 2, 1, True)
    else:
        ref = F.relu(F.conv2d(torch.abs(xs[0]), xs[1], xs[2], stride=2, padding=2))
        out = autograd.ConvHip.apply(xd[0], xd[1], xd[2], 2, 2, True, True)
    up = torch.randn(ref.shape, generator=g)
    (ref * up).sum().backward()
    (out * up.to(DEV)).sum().backward()
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 2e-5 * float(ref.abs().max())
    for a, r in zip(xd, xs):
        assert float((a.grad.cpu() - r.grad).abs().max()) <= 5e-5 * float(r.grad.abs().max()), a.shape


def test_conv_backward_with_odd_height_and_even_width():
    """dgrad's output padding is per axis (an odd H with an even W used to lose the last column's gradient)."""
    import torch.nn.functional as F
    from licos_amd import autograd
    g = torch.Generator().manual_seed(5)
    for h, w in ((9, 12), (12, 9), (9, 9), (12, 12)):
        x = torch.randn(2, 8, h, w, generator=g)
        wt = 0.1 * torch.randn(6, 8, 5, 5, generator=g)
        xr, xd = x.clone().requires_grad_(True), x.clone().to(DEV).requires_grad_(True)
        F.conv2d(xr, wt, None, stride=2, padding=2).square().sum().backward()
        autograd.ConvHip.apply(xd, wt.to(DEV), None, 2, 2).square().sum().backward()
        assert float((xd.grad.cpu() - xr.grad).abs().max()) <= 2e-5 * float(xr.grad.abs().max()), (h, w)


def test_config4_train_step_gradients_at_full_size():
    """cfg/raw_merged.toml's step itself - 16 patches of 13 x 256 x 256 (channel 12 zero) - every parameter gradient
    against the oracle under CPU autograd (BASELINE configs[3]'s per-rank workload)."""
    import os
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    sd = om.perturb_state(om.make_factorized_state(13, quality=1, seed=42), seed=17, y_gain=20.0)
    net = licos_amd.get_model("bmshj2018-factorized", False, 13, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    x = om.synthetic_tiles(16, 13, 256, seed=9, kind="s2-merged")
    g = torch.Generator().manual_seed(2)
    noise = torch.rand(16, 192, 16, 16, generator=g) - 0.5
    res = licos_amd.RateDistortionLoss(lmbda=1e-2)(net(x.to(DEV), noise=noise.to(DEV)), x.to(DEV))
    res["loss"].backward()
    ref_sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and v.dim() > 0 and "bound" not in k
                  and "pedestal" not in k and "target" not in k else v) for k, v in sd.items()}
    ref = om.rate_distortion_loss(om.forward(x, ref_sd, training=True, noise=noise), x, 1e-2)
    ref["loss"].backward()
    assert abs(float(res["loss"].detach()) - float(ref["loss"].detach())) < 1e-4 * abs(float(ref["loss"].detach()))
    checked = 0
    for name, p in net.named_parameters():
        rg = ref_sd[name].grad
        if rg is None:
            continue
        err = float((p.grad.cpu() - rg).abs().max() / rg.abs().max().clamp_min(1e-20))
        assert err < 2e-3, (name, err)
        checked += 1
    assert checked >= 40


@pytest.mark.parametrize("mag", [1.0, 1e-8, 1e-12])
def test_gdn_gamma_gradient_keeps_tiny_gradients(mag):
    """licos_gdn_gamma_grad_f32 (the 128 x 128 Gram product over all pixels, three-pass split operands on the matrix
    cores) against float64 for dL/dnorm at the magnitudes a mean-reduced RD loss produces (1e-8) and far below (1e-12):
    the operand is pre-scaled by a power of two taken from max|t|, so fp16's range does not eat the bits."""
    from licos_amd import ops
    g = torch.Generator().manual_seed(3)
    b, h, w = 3, 24, 20
    x = torch.randn(b, 128, h, w, generator=g) * 3.0
    t = torch.randn(b, 128, h, w, generator=g) * mag
    got = ops.conv2d_wgrad_f32(x.to(DEV), t.to(DEV), 128, 128, 1, 1, 0, square_input=True).cpu().double().reshape(128, 128)
    ref = torch.einsum("bip,bjp->ij", t.double().reshape(b, 128, -1), (x.double() ** 2).reshape(b, 128, -1))
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, (mag, err)


def test_likelihood_backward_accepts_expanded_gradients():
    """loss = likelihoods.sum(): autograd hands the backward an expanded (stride-0) gradient; the HIP backward kernels
    take it (made contiguous at the wrapper) - entropy bottleneck and Gaussian conditional."""
    eb = licos_amd.EntropyBottleneck(8).to(DEV).train()
    x = torch.randn(2, 8, 6, 6, device=DEV, requires_grad=True)
    _, lik = eb(x, noise=torch.zeros_like(x))
    lik.sum().backward()
    assert x.grad is not None and bool(torch.isfinite(x.grad).all()) and float(x.grad.abs().max()) > 0
    gc = licos_amd.GaussianConditional(None).to(DEV).train()
    y = torch.randn(2, 8, 6, 6, device=DEV, requires_grad=True)
    s = (torch.rand(2, 8, 6, 6, device=DEV) + 0.2).requires_grad_(True)
    _, lik = gc(y, s, noise=torch.zeros_like(y))
    lik.sum().backward()
    assert float(y.grad.abs().max()) > 0 and float(s.grad.abs().max()) > 0
