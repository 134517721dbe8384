"""torch-CPU restatement of the CompressAI modules the LICOS hot path runs.

TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle/__init__.py).

Everything works on a plain ``state_dict`` (CompressAI key names) so that the
same weights can be pushed through this oracle and through the HIP product.

Restated from (upstream CompressAI file -> reference call site):
  models/google.py  FactorizedPrior / ScaleHyperprior  -> licos/model_utils.py:19, licos/train.py:190
  layers/gdn.py, ops/parametrizers.py, ops/bound_ops.py -> (inside g_a / g_s)
  entropy_models/entropy_models.py EntropyBottleneck    -> licos/model_utils.py:25-29, eval_script.py:72
  losses/rate_distortion.py                             -> licos/train.py:123,192
  licos/model_utils.py:6-49 (channel surgery), eval_utils.py:145-186 (metrics),
  licos/federation_utils.py:47-53 (weight blend)
The convolutions are torch.nn.functional.conv2d / conv_transpose2d on CPU, which
is literally the reference's CPU arithmetic (CompressAI's g_a/g_s are stock
torch.nn modules).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import rans

PEDESTAL = (2.0 ** -18) ** 2
BETA_MIN = 1e-6
GAMMA_INIT = 0.1
LIKELIHOOD_BOUND = 1e-9
TAIL_MASS = 1e-9
INIT_SCALE = 10.0


def quality_to_nm(quality):
    """compressai/zoo/image.py cfgs for bmshj2018-factorized / -hyperprior."""
    if not 1 <= quality <= 8:
        raise ValueError(f'Invalid quality "{quality}", should be between (1, 8)')
    return (128, 192) if quality <= 5 else (192, 320)


# --------------------------------------------------------------------------- init
def _conv_init(cout, cin, k, gen, transpose=False):
    """torch.nn.Conv2d / ConvTranspose2d default reset_parameters()."""
    shape = (cin, cout, k, k) if transpose else (cout, cin, k, k)
    w = torch.empty(shape)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=gen)
    fan_in = shape[1] * k * k
    bound = 1 / math.sqrt(fan_in)
    b = torch.empty(cout).uniform_(-bound, bound, generator=gen)
    return w, b


def _gdn_init(sd, prefix, c):
    sd[prefix + "beta"] = torch.sqrt(torch.clamp(torch.ones(c) + PEDESTAL, min=PEDESTAL))
    sd[prefix + "gamma"] = torch.sqrt(torch.clamp(GAMMA_INIT * torch.eye(c) + PEDESTAL, min=PEDESTAL))
    sd[prefix + "beta_reparam.pedestal"] = torch.tensor([PEDESTAL])
    sd[prefix + "beta_reparam.lower_bound.bound"] = torch.tensor([(BETA_MIN + PEDESTAL) ** 0.5])
    sd[prefix + "gamma_reparam.pedestal"] = torch.tensor([PEDESTAL])
    sd[prefix + "gamma_reparam.lower_bound.bound"] = torch.tensor([PEDESTAL ** 0.5])


def eb_init(sd, prefix, channels, filters=(3, 3, 3, 3), gen=None):
    f = (1,) + tuple(filters) + (1,)
    scale = INIT_SCALE ** (1 / (len(filters) + 1))
    for i in range(len(filters) + 1):
        init = np.log(np.expm1(1 / scale / f[i + 1]))
        sd[f"{prefix}matrices.{i}"] = torch.full((channels, f[i + 1], f[i]), float(init))
        sd[f"{prefix}biases.{i}"] = torch.empty(channels, f[i + 1], 1).uniform_(-0.5, 0.5, generator=gen)
        if i < len(filters):
            sd[f"{prefix}factors.{i}"] = torch.zeros(channels, f[i + 1], 1)
    sd[prefix + "quantiles"] = torch.tensor([-INIT_SCALE, 0.0, INIT_SCALE]).repeat(channels, 1, 1)
    t = np.log(2 / TAIL_MASS - 1)
    sd[prefix + "target"] = torch.tensor([-t, 0.0, t], dtype=torch.float32)
    sd[prefix + "likelihood_lower_bound.bound"] = torch.tensor([LIKELIHOOD_BOUND])
    sd[prefix + "_offset"] = torch.zeros(0, dtype=torch.int32)
    sd[prefix + "_quantized_cdf"] = torch.zeros(0, dtype=torch.int32)
    sd[prefix + "_cdf_length"] = torch.zeros(0, dtype=torch.int32)


def make_factorized_state(in_channels=3, quality=1, seed=42, eb_filters=None):
    """A seeded random-init bmshj2018-factorized state_dict after the LICOS channel
    surgery of licos/model_utils.py:25-45 (EB filters = (in,in,3,3) there)."""
    n, m = quality_to_nm(quality)
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    chans = [(in_channels, n), (n, n), (n, n), (n, m)]
    for li, (ci, co) in enumerate(chans):
        sd[f"g_a.{2 * li}.weight"], sd[f"g_a.{2 * li}.bias"] = _conv_init(co, ci, 5, gen)
        if li < 3:
            _gdn_init(sd, f"g_a.{2 * li + 1}.", n)
    chans = [(m, n), (n, n), (n, n), (n, in_channels)]
    for li, (ci, co) in enumerate(chans):
        sd[f"g_s.{2 * li}.weight"], sd[f"g_s.{2 * li}.bias"] = _conv_init(co, ci, 5, gen, transpose=True)
        if li < 3:
            _gdn_init(sd, f"g_s.{2 * li + 1}.", n)
    if eb_filters is None:
        eb_filters = (in_channels, in_channels, 3, 3)
    eb_init(sd, "entropy_bottleneck.", m, eb_filters, gen)
    return sd


def count_parameters(sd):
    """Learnable parameters only (buffers excluded) - CompressAI's demo prints
    2 998 147 for bmshj2018_factorized(quality=1) with stock (3,3,3,3) filters."""
    n = 0
    for k, v in sd.items():
        leaf = k.split(".")[-1]
        if leaf in ("pedestal", "bound", "target", "_offset", "_quantized_cdf", "_cdf_length", "scale_table", "scale_bound"):
            continue
        n += v.numel()
    return n


# --------------------------------------------------------------------------- GDN
def gdn(x, sd, prefix, inverse=False):
    """layers/gdn.py GDN.forward with ops/parametrizers.py NonNegativeParametrizer."""
    ped = sd[prefix + "beta_reparam.pedestal"]
    beta = torch.max(sd[prefix + "beta"], sd[prefix + "beta_reparam.lower_bound.bound"]) ** 2 - ped
    ped = sd[prefix + "gamma_reparam.pedestal"]
    gamma = torch.max(sd[prefix + "gamma"], sd[prefix + "gamma_reparam.lower_bound.bound"]) ** 2 - ped
    c = x.shape[1]
    norm = F.conv2d(x ** 2, gamma.reshape(c, c, 1, 1), beta)
    norm = torch.sqrt(norm) if inverse else torch.rsqrt(norm)
    return x * norm


def g_a(x, sd, prefix="g_a."):
    for li in range(4):
        x = F.conv2d(x, sd[f"{prefix}{2 * li}.weight"], sd[f"{prefix}{2 * li}.bias"], stride=2, padding=2)
        if li < 3:
            x = gdn(x, sd, f"{prefix}{2 * li + 1}.")
    return x


def g_s(y, sd, prefix="g_s."):
    for li in range(4):
        y = F.conv_transpose2d(y, sd[f"{prefix}{2 * li}.weight"], sd[f"{prefix}{2 * li}.bias"],
                               stride=2, padding=2, output_padding=1)
        if li < 3:
            y = gdn(y, sd, f"{prefix}{2 * li + 1}.", inverse=True)
    return y


# --------------------------------------------------------------------------- EB
def _eb_key(sd, prefix, kind, i):
    new = {"matrix": "matrices", "bias": "biases", "factor": "factors"}[kind]
    k = f"{prefix}{new}.{i}"
    return sd[k] if k in sd else sd[f"{prefix}_{kind}{i}"]


def eb_num_layers(sd, prefix):
    n = 0
    while f"{prefix}matrices.{n}" in sd or f"{prefix}_matrix{n}" in sd:
        n += 1
    return n


def logits_cumulative(v, sd, prefix="entropy_bottleneck."):
    """EntropyBottleneck._logits_cumulative; v is (C, 1, n)."""
    nl = eb_num_layers(sd, prefix)
    logits = v
    for i in range(nl):
        logits = torch.matmul(F.softplus(_eb_key(sd, prefix, "matrix", i)), logits)
        logits = logits + _eb_key(sd, prefix, "bias", i)
        if i < nl - 1:
            logits = logits + torch.tanh(_eb_key(sd, prefix, "factor", i)) * torch.tanh(logits)
    return logits


def likelihood(v, sd, prefix="entropy_bottleneck.", form="plain"):
    """EntropyBottleneck._likelihood.  form="plain": current releases
    (sigmoid(upper) - sigmoid(lower)); form="signflip": 1.1/1.2-era releases.
    Returns (likelihood, lower, upper)."""
    lower = logits_cumulative(v - 0.5, sd, prefix)
    upper = logits_cumulative(v + 0.5, sd, prefix)
    if form == "plain":
        lik = torch.sigmoid(upper) - torch.sigmoid(lower)
    else:
        sign = -torch.sign(lower + upper)
        lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
    return lik, lower, upper


def eb_medians(sd, prefix="entropy_bottleneck."):
    return sd[prefix + "quantiles"][:, :, 1:2]


def eb_forward(y, sd, prefix="entropy_bottleneck.", training=False, noise=None, form="plain"):
    """EntropyBottleneck.forward: returns (y_hat, likelihoods) shaped like y.
    training=True uses the supplied noise tensor (same shape as y) in place of
    uniform_(-.5,.5) so tests stay deterministic."""
    b, c = y.shape[:2]
    v = y.transpose(0, 1).contiguous()
    shape = v.shape
    v = v.reshape(c, 1, -1)
    med = eb_medians(sd, prefix)
    if training:
        nz = noise.transpose(0, 1).contiguous().reshape(c, 1, -1)
        out = v + nz
    else:
        out = torch.round(v - med) + med
    lik, _, _ = likelihood(out, sd, prefix, form)
    lik = torch.clamp(lik, min=float(sd[prefix + "likelihood_lower_bound.bound"]))
    out = out.reshape(shape).transpose(0, 1).contiguous()
    lik = lik.reshape(shape).transpose(0, 1).contiguous()
    return out, lik


def eb_aux_loss(sd, prefix="entropy_bottleneck."):
    logits = logits_cumulative(sd[prefix + "quantiles"], sd, prefix)
    return torch.abs(logits - sd[prefix + "target"]).sum()


def eb_update(sd, prefix="entropy_bottleneck.", form="plain"):
    """EntropyBottleneck.update(force=True): fills _offset/_quantized_cdf/_cdf_length."""
    q = sd[prefix + "quantiles"]
    medians = q[:, 0, 1]
    minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
    maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
    offset = -minima
    pmf_start = medians - minima
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max().item())
    samples = torch.arange(max_length)
    samples = samples[None, :] + pmf_start[:, None, None]
    pmf, lower, upper = likelihood(samples, sd, prefix, form)
    pmf = pmf[:, 0, :]
    tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
    cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
    for i, p in enumerate(pmf):
        prob = torch.cat((p[: pmf_length[i]], tail_mass[i]), dim=0)
        _cdf = rans.pmf_to_quantized_cdf(prob.numpy(), 16)
        cdf[i, : _cdf.size] = torch.from_numpy(_cdf)
    sd[prefix + "_offset"] = offset
    sd[prefix + "_quantized_cdf"] = cdf
    sd[prefix + "_cdf_length"] = (pmf_length + 2).int()
    return True


def eb_symbols(y, sd, prefix="entropy_bottleneck."):
    med = eb_medians(sd, prefix).reshape(1, -1, 1, 1)
    return torch.round(y - med).int()


def eb_compress(y, sd, prefix="entropy_bottleneck."):
    if sd[prefix + "_offset"].numel() == 0:
        raise ValueError("Uninitialized CDFs. Run update() first")
    sym = eb_symbols(y, sd, prefix)
    b, c, h, w = sym.shape
    idx = torch.arange(c, dtype=torch.int32).view(c, 1, 1).expand(c, h, w).reshape(-1).numpy()
    return [rans.encode_with_indexes(sym[i].reshape(-1).numpy(), idx, sd[prefix + "_quantized_cdf"].numpy(),
                                     sd[prefix + "_cdf_length"].numpy(), sd[prefix + "_offset"].numpy())
            for i in range(b)]


def eb_decompress(strings, size, sd, prefix="entropy_bottleneck."):
    c = sd[prefix + "_quantized_cdf"].shape[0]
    h, w = size
    idx = torch.arange(c, dtype=torch.int32).view(c, 1, 1).expand(c, h, w).reshape(-1).numpy()
    out = torch.empty(len(strings), c, h, w)
    for i, s in enumerate(strings):
        v = rans.decode_with_indexes(s, idx, sd[prefix + "_quantized_cdf"].numpy(),
                                     sd[prefix + "_cdf_length"].numpy(), sd[prefix + "_offset"].numpy())
        out[i] = torch.from_numpy(v).reshape(c, h, w).float()
    return out + eb_medians(sd, prefix).reshape(1, -1, 1, 1)


# --------------------------------------------------------------------------- model
def forward(x, sd, training=False, noise=None, form="plain"):
    y = g_a(x, sd)
    y_hat, lik = eb_forward(y, sd, training=training, noise=noise, form=form)
    x_hat = g_s(y_hat, sd)
    return {"x_hat": x_hat, "likelihoods": {"y": lik}, "y": y, "y_hat": y_hat}


def compress(x, sd):
    y = g_a(x, sd)
    return {"strings": [eb_compress(y, sd)], "shape": tuple(y.shape[-2:])}


def decompress(strings, shape, sd):
    assert isinstance(strings, list) and len(strings) == 1
    y_hat = eb_decompress(strings[0], shape, sd)
    return {"x_hat": g_s(y_hat, sd).clamp_(0, 1)}


# --------------------------------------------------------------------------- loss / metrics
def rate_distortion_loss(out, target, lmbda=1e-2):
    """losses/rate_distortion.py RateDistortionLoss(metric="mse") (licos/train.py:123,192)."""
    n, _, h, w = target.shape
    num_pixels = n * h * w
    bpp = sum(torch.log(l).sum() / (-math.log(2) * num_pixels) for l in out["likelihoods"].values())
    mse = F.mse_loss(out["x_hat"], target)
    return {"bpp_loss": bpp, "mse_loss": mse, "loss": lmbda * 255 ** 2 * mse + bpp}


def compute_bpp(out):
    """eval_utils.py:172-186."""
    size = out["x_hat"].size()
    num_pixels = size[0] * size[2] * size[3]
    return sum(torch.log(l).sum() / (-math.log(2) * num_pixels) for l in out["likelihoods"].values()).item()


def compute_psnr(a, b):
    """eval_utils.py:145-156."""
    mse = torch.mean((a - b) ** 2).item()
    return -10 * math.log10(mse)


# --------------------------------------------------------------------------- federation
def blend(local_sd, central_sd, loss, best_loss):
    """licos/federation_utils.py:47-53: pair-wise convex blend of every key."""
    wl = best_loss / (best_loss + loss)
    wc = loss / (best_loss + loss)
    out = {}
    for k in local_sd:
        t = wl * local_sd[k]
        t = t + wc * central_sd[k]
        out[k] = t
    return out


def sequential_federation(states, losses, best_losses):
    """Ranks 0..N-1 visit the ground station in order (federation_utils.py:38-85):
    the first visitor seeds the central model, every later one blends and adopts."""
    central = {k: v.clone() for k, v in states[0].items()}
    for r in range(1, len(states)):
        central = blend(states[r], central, losses[r], best_losses[r])
    return central


# --------------------------------------------------------------------------- synthetic inputs
def synthetic_tiles(batch, channels=3, size=256, seed=0, kind="aid"):
    """Deterministic synthetic tiles on the 8-bit grid.
    kind="aid": RGB-like; kind="s2": 12-bit DN grid first (raw_utils.py:128,
    raw_image_folder.py:192-196); kind="s2-merged": 13 ch with channel 12 zero
    (raw_image_folder.py:172)."""
    rng = np.random.default_rng(seed)
    out = np.empty((batch, channels, size, size), dtype=np.float32)
    for b in range(batch):
        low = rng.standard_normal((channels, size // 8, size // 8)).astype(np.float32)
        up = F.interpolate(torch.from_numpy(low)[None], scale_factor=8, mode="bilinear", align_corners=False)[0].numpy()
        f = up + 0.05 * rng.standard_normal((channels, size, size)).astype(np.float32)
        f = (f - f.min()) / (f.max() - f.min())
        if kind != "aid":
            f = np.rint(f * 4095.0) / 4095.0
        f = np.rint(f * 255.0) / 255.0
        if kind == "s2-merged" and channels == 13:
            f[12] = 0.0
        out[b] = f.astype(np.float32)
    return torch.from_numpy(out)


# --------------------------------------------------------------------------- fixture weights
def perturb_state(sd, seed=7, y_gain=60.0, eb_init_scale=None):
    """Random-init nets give max|y| ~ 0.3 (all symbols 0).  For fixtures: scale g_a[6] so the
    latents span tens of quantisation bins, de-trivialise GDN and the EB MLP, move the medians
    off zero and give channels different table ranges (last row)."""
    gen = torch.Generator().manual_seed(seed)
    sd = {k: v.clone() for k, v in sd.items()}
    sd["g_a.6.weight"] = sd["g_a.6.weight"] * y_gain
    sd["g_a.6.bias"] = sd["g_a.6.bias"] * y_gain
    for k in list(sd.keys()):
        if k.endswith(".gamma") and sd[k].dim() == 2:
            c = sd[k].shape[0]
            sd[k] = sd[k] + 0.02 * torch.rand(c, c, generator=gen)
        elif k.endswith(".beta") and sd[k].dim() == 1:
            sd[k] = sd[k] * (0.5 + torch.rand(sd[k].shape, generator=gen))
        elif ".factors." in k or "_factor" in k:
            sd[k] = 0.5 * torch.randn(sd[k].shape, generator=gen)
        elif ".matrices." in k or "_matrix" in k:
            sd[k] = sd[k] + 0.3 * torch.randn(sd[k].shape, generator=gen)
    q = sd["entropy_bottleneck.quantiles"]
    c = q.shape[0]
    med = 3.0 * torch.randn(c, generator=gen)
    lo = 2.0 + 12.0 * torch.rand(c, generator=gen)
    hi = 2.0 + 12.0 * torch.rand(c, generator=gen)
    sd["entropy_bottleneck.quantiles"] = torch.stack((med - lo, med, med + hi), dim=1).reshape(c, 1, 3)
    return sd


# =========================================================================== ScaleHyperprior (config 5)
# CompressAI models/google.py ScaleHyperprior, entropy_models.py GaussianConditional, zoo/image.py cfgs;
# allowed by licos/model_utils.py:20-24 (EntropyBottleneck gets channels=N there).
SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(mn=SCALES_MIN, mx=SCALES_MAX, levels=SCALES_LEVELS):
    return torch.exp(torch.linspace(math.log(mn), math.log(mx), levels))


def make_hyperprior_state(in_channels=3, quality=1, seed=42, eb_filters=None):
    n, m = quality_to_nm(quality)
    sd = make_factorized_state(in_channels, quality, seed, eb_filters=(3, 3, 3, 3))
    for k in [k for k in sd if k.startswith("entropy_bottleneck.")]:
        del sd[k]
    gen = torch.Generator().manual_seed(seed + 1)
    sd["h_a.0.weight"], sd["h_a.0.bias"] = _conv_init(n, m, 3, gen)
    sd["h_a.2.weight"], sd["h_a.2.bias"] = _conv_init(n, n, 5, gen)
    sd["h_a.4.weight"], sd["h_a.4.bias"] = _conv_init(n, n, 5, gen)
    sd["h_s.0.weight"], sd["h_s.0.bias"] = _conv_init(n, n, 5, gen, transpose=True)
    sd["h_s.2.weight"], sd["h_s.2.bias"] = _conv_init(n, n, 5, gen, transpose=True)
    sd["h_s.4.weight"], sd["h_s.4.bias"] = _conv_init(m, n, 3, gen)
    if eb_filters is None:
        eb_filters = (in_channels, in_channels, 3, 3)
    eb_init(sd, "entropy_bottleneck.", n, eb_filters, gen)
    p = "gaussian_conditional."
    sd[p + "scale_table"] = torch.zeros(0)
    sd[p + "scale_bound"] = torch.tensor([SCALES_MIN])
    sd[p + "lower_bound_scale.bound"] = torch.tensor([SCALES_MIN])
    sd[p + "likelihood_lower_bound.bound"] = torch.tensor([LIKELIHOOD_BOUND])
    sd[p + "_offset"] = torch.zeros(0, dtype=torch.int32)
    sd[p + "_quantized_cdf"] = torch.zeros(0, dtype=torch.int32)
    sd[p + "_cdf_length"] = torch.zeros(0, dtype=torch.int32)
    return sd


def h_a(y, sd):
    z = F.relu(F.conv2d(torch.abs(y), sd["h_a.0.weight"], sd["h_a.0.bias"], stride=1, padding=1))
    z = F.relu(F.conv2d(z, sd["h_a.2.weight"], sd["h_a.2.bias"], stride=2, padding=2))
    return F.conv2d(z, sd["h_a.4.weight"], sd["h_a.4.bias"], stride=2, padding=2)


def h_s(z, sd):
    s = F.relu(F.conv_transpose2d(z, sd["h_s.0.weight"], sd["h_s.0.bias"], stride=2, padding=2, output_padding=1))
    s = F.relu(F.conv_transpose2d(s, sd["h_s.2.weight"], sd["h_s.2.bias"], stride=2, padding=2, output_padding=1))
    return F.relu(F.conv2d(s, sd["h_s.4.weight"], sd["h_s.4.bias"], stride=1, padding=1))


def _std_cumulative(x):
    return 0.5 * torch.erfc(-(2 ** -0.5) * x)


def gc_update(sd, prefix="gaussian_conditional.", scale_table=None):
    """GaussianConditional.update_scale_table + update()."""
    from scipy.stats import norm
    table = get_scale_table() if scale_table is None else scale_table
    sd[prefix + "scale_table"] = table
    multiplier = -norm.ppf(TAIL_MASS / 2)
    pmf_center = torch.ceil(table * multiplier).int()
    pmf_length = 2 * pmf_center + 1
    max_length = int(torch.max(pmf_length).item())
    samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
    scale = table.unsqueeze(1).float()
    upper = _std_cumulative((0.5 - samples) / scale)
    lower = _std_cumulative((-0.5 - samples) / scale)
    pmf = upper - lower
    tail_mass = 2 * lower[:, :1]
    cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
    for i, p in enumerate(pmf):
        prob = torch.cat((p[: pmf_length[i]], tail_mass[i]), dim=0)
        c = rans.pmf_to_quantized_cdf(prob.numpy(), 16)
        cdf[i, : c.size] = torch.from_numpy(c)
    sd[prefix + "_quantized_cdf"] = cdf
    sd[prefix + "_offset"] = -pmf_center
    sd[prefix + "_cdf_length"] = (pmf_length + 2).int()


def gc_likelihood(y_hat, scales, sd, prefix="gaussian_conditional."):
    s = torch.max(scales, sd[prefix + "lower_bound_scale.bound"])
    v = torch.abs(y_hat)
    lik = _std_cumulative((0.5 - v) / s) - _std_cumulative((-0.5 - v) / s)
    return torch.clamp(lik, min=float(sd[prefix + "likelihood_lower_bound.bound"]))


def gc_build_indexes(scales, sd, prefix="gaussian_conditional."):
    s = torch.max(scales, sd[prefix + "lower_bound_scale.bound"])
    table = sd[prefix + "scale_table"]
    idx = torch.full(s.shape, len(table) - 1, dtype=torch.int32)
    for t in table[:-1]:
        idx -= (s <= t).int()
    return idx


def hyper_update(sd, form="plain"):
    gc_update(sd)
    eb_update(sd, form=form)


def hyper_forward(x, sd, form="plain"):
    y = g_a(x, sd)
    z = h_a(y, sd)
    z_hat, z_lik = eb_forward(z, sd, form=form)
    scales = h_s(z_hat, sd)
    y_hat = torch.round(y)
    y_lik = gc_likelihood(y_hat, scales, sd)
    x_hat = g_s(y_hat, sd)
    return {"x_hat": x_hat, "likelihoods": {"y": y_lik, "z": z_lik}, "y": y, "z": z, "scales": scales, "y_hat": y_hat,
            "z_hat": z_hat}


def hyper_compress(x, sd):
    y = g_a(x, sd)
    z = h_a(y, sd)
    z_strings = eb_compress(z, sd)
    z_hat = eb_decompress(z_strings, z.shape[-2:], sd)
    scales = h_s(z_hat, sd)
    idx = gc_build_indexes(scales, sd)
    p = "gaussian_conditional."
    sym = torch.round(y).int()
    y_strings = [rans.encode_with_indexes(sym[i].reshape(-1).numpy(), idx[i].reshape(-1).numpy(),
                                          sd[p + "_quantized_cdf"].numpy(), sd[p + "_cdf_length"].numpy(),
                                          sd[p + "_offset"].numpy()) for i in range(y.shape[0])]
    return {"strings": [y_strings, z_strings], "shape": tuple(z.shape[-2:])}


def hyper_decompress(strings, shape, sd):
    assert isinstance(strings, list) and len(strings) == 2
    z_hat = eb_decompress(strings[1], shape, sd)
    scales = h_s(z_hat, sd)
    idx = gc_build_indexes(scales, sd)
    p = "gaussian_conditional."
    y_hat = torch.empty(scales.shape)
    for i, s in enumerate(strings[0]):
        v = rans.decode_with_indexes(s, idx[i].reshape(-1).numpy(), sd[p + "_quantized_cdf"].numpy(),
                                     sd[p + "_cdf_length"].numpy(), sd[p + "_offset"].numpy())
        y_hat[i] = torch.from_numpy(v).reshape(scales.shape[1:]).float()
    return {"x_hat": g_s(y_hat, sd).clamp_(0, 1)}
