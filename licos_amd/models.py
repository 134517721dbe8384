"""FactorizedPrior with CompressAI's nn.Module surface (``models/google.py``,
``models/base.py``), the object /root/reference/licos/model_utils.py:19 obtains from
``image_models[...]`` and that train.py:190 / eval_utils.py:200-201 drive through
``forward`` / ``compress`` / ``decompress`` / ``update`` / ``aux_loss``.

``g_a`` / ``g_s`` are indexable, item-assignable ``nn.Sequential`` containers of stock
``torch.nn.Conv2d`` / ``ConvTranspose2d`` modules (parameter holders - LICOS swaps
``g_a[0]`` and ``g_s[6]`` for fresh stock modules, model_utils.py:31-45) and ``GDN``
modules.  Their ``forward`` never calls torch's convolutions: it launches the HIP
kernels with whatever weights currently sit in the slots.

precision:
  "fp32"  direct fp32 kernels; latents/likelihoods within 1e-5 of the CPU reference
          arithmetic and identical rANS bytes (parity path);
  "fp16"  fused MFMA pipeline (throughput path, BASELINE.json configs[1]).
"""
import os

import torch
import torch.nn as nn

from . import autograd, ops
import math

from .entropy_models import EntropyBottleneck, GaussianConditional
from .layers import GDN, conv, conv_geometry, deconv


FUSE_FP32 = os.environ.get("LICOS_FUSE_FP32", "1") != "0"  # A/B switch: the fp32 chain layer by layer (NCHW fp32 between all of them)


class TransformSequential(nn.Sequential):
    """nn.Sequential whose forward runs the HIP transform pipeline."""

    precision = "fp32"
    abs_input = False   # ScaleHyperprior.h_a consumes |y|
    fp32_only = False   # pins a chain to the generic fp32 kernels whatever the model precision

    def forward(self, x):
        if self.precision == "fp16" and not self.fp32_only:
            # the fp16 MFMA path is inference-only.  In train() mode a needed gradient is an error (silently returning
            # a detached tensor would train nothing); in eval() mode - evaluation loops, compress()/decompress()
            # called without torch.no_grad(), as CompressAI allows - it simply runs without a graph
            if self.training and autograd.needs_grad(x, *self.parameters()):
                raise NotImplementedError("licos_amd: training (autograd) runs on precision='fp32'; the fp16 MFMA "
                                          "path is inference-only (use net.eval() / torch.no_grad() to evaluate)")
            from .engine import run_chain_fp16
            with torch.no_grad():
                return run_chain_fp16(self, x)
        return run_chain_fp32(self, x)


def _x3_layer(m, relu):
    """Is this conv module served by the one-launch split-operand form (ops.x3_route)?"""
    if not isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        return False
    geo = conv_geometry(m)
    if isinstance(m, nn.ConvTranspose2d):
        return ops.x3_route(m.in_channels, m.out_channels, geo[0], geo[1], geo[2], geo[3], relu)
    return ops.x3_route(m.in_channels, m.out_channels, geo[0], geo[1], geo[2], None, relu)


def _is_relu(mods, i):
    return i < len(mods) and isinstance(mods[i], nn.ReLU)


def _takes_split3(gdn, x, nxt, nxt_relu):
    """A stand-alone GDN whose consumer is a convolution on the one-launch split-operand route, with no gradient wanted."""
    if not isinstance(nxt, (nn.Conv2d, nn.ConvTranspose2d)) or x.dim() != 4:
        return False
    if autograd.needs_grad(x, gdn.beta, gdn.gamma, nxt.weight, nxt.bias):
        return False
    if not ops.gdn_f32_split3_applies(x.shape[1], x.shape[2] * x.shape[3]):
        return False
    return _x3_layer(nxt, nxt_relu)


def run_chain_fp32(seq, x):
    """The transform on the fp32 kernels.  Without gradients (the parity path of compress / decompress / eval forward) the
    chain is fused where the kernels allow it: a convolution on the split-operand route applies the (I)GDN that follows
    it in its epilogue (ops.EPI_NORM32) and hands the next such convolution its operand already split (ops.Split3) - no
    NCHW fp32 round trip between the two, the values of the unfused chain."""
    if x.dtype != torch.float32:
        raise ValueError("licos_amd: inputs must be float32")
    x = x.contiguous()
    mods = list(seq)
    i = 0
    first = True
    while i < len(mods):
        m = mods[i]
        relu = _is_relu(mods, i + 1)
        abs_in = first and getattr(seq, "abs_input", False)
        first = False
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            transposed = isinstance(m, nn.ConvTranspose2d)
            geo = conv_geometry(m)
            bias = None if m.bias is None else m.bias.detach()
            if not isinstance(x, ops.Split3) and autograd.needs_grad(x, m.weight, m.bias):
                if transposed:
                    x = autograd.DeconvHip.apply(x, m.weight, m.bias, geo[1], geo[2], geo[3], relu)  # HIP forward and backward
                else:
                    x = autograd.ConvHip.apply(x, m.weight, m.bias, geo[1], geo[2], relu, abs_in)
                i += 2 if relu else 1
                continue
            # inference: what can ride in this layer's epilogue
            gdn, step = None, (2 if relu else 1)
            if FUSE_FP32 and _x3_layer(m, relu) and m.out_channels > 32:
                g = mods[i + 1] if i + 1 < len(mods) else None
                # (the norm rides with forward convolutions only: the transposed kernels that write fp32 work one output
                # phase of an 8 x 32 tile per workgroup, and 64 KB of gamma fragments per such workgroup cost more than
                # the IGDN kernel they would save - measured 233 against 135 ms per 4096 tiles)
                if (isinstance(g, GDN) and not relu and not transposed and not autograd.needs_grad(g.beta, g.gamma)
                        and g.in_channels == m.out_channels and g.packed_f32split() is not None):
                    gdn, step = (g.packed_f32split(), g.inverse), 2
                nxt = mods[i + step] if i + step < len(mods) else None
                split3 = (m.out_channels % 16 == 0 and _x3_layer(nxt, _is_relu(mods, i + step + 1))
                          and not autograd.needs_grad(nxt.weight, nxt.bias))
            else:
                split3 = False
            if transposed:
                x = ops.deconv2d_f32(x, m.weight.detach(), bias, geo[1], geo[2], geo[3], relu, gdn=gdn, split3_out=split3)
            else:
                x = ops.conv2d_f32(x, m.weight.detach(), bias, geo[1], geo[2], relu, abs_input=abs_in, gdn=gdn, split3_out=split3)
            i += step
        elif isinstance(m, GDN):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if FUSE_FP32 and _takes_split3(m, x, nxt, _is_relu(mods, i + 2)):
                # a GDN no convolution could take along: its own kernel, the result already split for the next layer
                beta, gamma = m.effective()
                x = ops.gdn_f32_split3(x.contiguous(), gamma, beta, m.inverse)
            else:
                x = m(x)
            i += 1
        else:
            raise TypeError(f"licos_amd: unsupported module in transform: {type(m).__name__}")
    return x


class CompressionModel(nn.Module):
    """CompressAI ``models/base.py`` CompressionModel surface."""

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def update(self, scale_table=None, force=False):
        updated = False
        for m in self.children():
            if isinstance(m, GaussianConditional):
                updated |= m.update_scale_table(get_scale_table() if scale_table is None else scale_table, force=force)
            if isinstance(m, EntropyBottleneck):
                updated |= m.update(force=force)
        return updated

    def set_precision(self, precision):
        if precision not in ("fp32", "fp16"):
            raise ValueError("precision must be 'fp32' or 'fp16'")
        self.precision = precision
        for m in self.children():
            if isinstance(m, TransformSequential):
                m.precision = precision
        return self


def get_scale_table(min=0.11, max=256, levels=64):
    """CompressAI models/google.py get_scale_table."""
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


class FactorizedPrior(CompressionModel):
    def __init__(self, N, M, precision="fp32", **kwargs):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(M)
        self.g_a = TransformSequential(conv(3, N), GDN(N), conv(N, N), GDN(N), conv(N, N), GDN(N), conv(N, M))
        self.g_s = TransformSequential(deconv(M, N), GDN(N, inverse=True), deconv(N, N), GDN(N, inverse=True),
                                       deconv(N, N), GDN(N, inverse=True), deconv(N, 3))
        self.N = N
        self.M = M
        self.precision = precision
        self.chunk = 1024  # tiles per pipeline chunk of the fp16 codec (licos_amd/codec.py)

    @property
    def downsampling_factor(self):
        return 2 ** 4

    def _sync_precision(self):
        self.g_a.precision = self.precision
        self.g_s.precision = self.precision

    def forward(self, x, noise=None):
        self._sync_precision()
        y = self.g_a(x)
        y_hat, y_likelihoods = self.entropy_bottleneck(y, noise=noise)
        x_hat = self.g_s(y_hat)
        return {"x_hat": x_hat, "likelihoods": {"y": y_likelihoods}}

    def compress(self, x):
        self._sync_precision()
        if x.shape[0] == 0:  # empty batch: nothing to launch
            return {"strings": [[]], "shape": torch.Size((x.shape[2] // 16, x.shape[3] // 16))}
        # one pipeline for both precisions (licos_amd/codec.py): chunks of `self.chunk` tiles bound the activations - the
        # fp32 parity path's first stage alone is 8.4 MB per 256 x 256 tile - and hide the serial coder under the transforms
        from .codec import compress_chunked
        return compress_chunked(self, x, chunk=self.chunk if self.precision == "fp16" else min(self.chunk, 1024))

    def decompress(self, strings, shape):
        assert isinstance(strings, list) and len(strings) == 1
        self._sync_precision()
        if len(strings[0]) == 0:
            dev = self.entropy_bottleneck.quantiles.device
            cout = self.g_s[len(self.g_s) - 1].out_channels
            return {"x_hat": torch.zeros((0, cout, int(shape[0]) * 16, int(shape[1]) * 16), device=dev)}
        from .codec import decompress_chunked
        return decompress_chunked(self, strings, shape, chunk=self.chunk if self.precision == "fp16" else min(self.chunk, 1024))

    @classmethod
    def from_state_dict(cls, state_dict):
        N = state_dict["g_a.0.weight"].size(0)
        M = state_dict["g_a.6.weight"].size(0)
        net = cls(N, M)
        net.load_state_dict(state_dict)
        return net


class FactorizedPriorReLU(FactorizedPrior):
    """bmshj2018-factorized-relu: every GDN replaced by ReLU (CompressAI models/google.py)."""

    def __init__(self, N, M, **kwargs):
        super().__init__(N=N, M=M, **kwargs)
        self.g_a = TransformSequential(conv(3, N), nn.ReLU(inplace=True), conv(N, N), nn.ReLU(inplace=True),
                                       conv(N, N), nn.ReLU(inplace=True), conv(N, M))
        self.g_s = TransformSequential(deconv(M, N), nn.ReLU(inplace=True), deconv(N, N), nn.ReLU(inplace=True),
                                       deconv(N, N), nn.ReLU(inplace=True), deconv(N, 3))


class ScaleHyperprior(CompressionModel):
    """CompressAI ``ScaleHyperprior`` (bmshj2018-hyperprior, BASELINE config 5; allowed by
    licos/model_utils.py:20-24).  All four transforms follow ``precision``: on "fp16" the hyper transforms run on
    the same MFMA kernels (3x3 stride-1 stages as one phase of the transposed-conv kernel, ReLU in the epilogue,
    |y| folded into the layout conversion)."""

    def __init__(self, N, M, precision="fp32", **kwargs):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.g_a = TransformSequential(conv(3, N), GDN(N), conv(N, N), GDN(N), conv(N, N), GDN(N), conv(N, M))
        self.g_s = TransformSequential(deconv(M, N), GDN(N, inverse=True), deconv(N, N), GDN(N, inverse=True),
                                       deconv(N, N), GDN(N, inverse=True), deconv(N, 3))
        self.h_a = TransformSequential(conv(M, N, stride=1, kernel_size=3), nn.ReLU(inplace=True), conv(N, N),
                                       nn.ReLU(inplace=True), conv(N, N))
        self.h_a.abs_input = True
        self.h_s = TransformSequential(deconv(N, N), nn.ReLU(inplace=True), deconv(N, N), nn.ReLU(inplace=True),
                                       conv(N, M, stride=1, kernel_size=3), nn.ReLU(inplace=True))
        self.gaussian_conditional = GaussianConditional(None)
        self.N = int(N)
        self.M = int(M)
        self.precision = precision
        # tiles of 512 x 512 per pipeline chunk of the large-batch codec (licos_amd/codec.py); larger tiles get
        # proportionally fewer (_chunk_for).  2048: 28 GB of 13-channel input and 34 GB of first-stage activations per
        # chunk.  Measured at 4096 tiles of 13 x 512^2: 512 -> 10.7k, 1024 -> 11.7k, 2048 -> 12.1k, 4096 -> 12.0k tiles/s -
        # the serial coder of a chunk does not hide under the next chunk's transforms for free (they share the CUs), so
        # fewer, larger chunks win until the last chunk's coder tail is all that is left.
        self.chunk = 2048

    def _chunk_for(self, h, w):
        return max(1, min(self.chunk, int(self.chunk * (512 * 512) / max(1, h * w))))

    @property
    def downsampling_factor(self):
        return 2 ** (4 + 2)

    def _sync_precision(self):
        for t in (self.g_a, self.g_s, self.h_a, self.h_s):
            t.precision = self.precision

    def forward(self, x, noise=None):
        self._sync_precision()
        y = self.g_a(x)
        z = self.h_a(y)
        z_hat, z_likelihoods = self.entropy_bottleneck(z, noise=None if noise is None else noise.get("z"))
        scales_hat = self.h_s(z_hat)
        y_hat, y_likelihoods = self.gaussian_conditional(y, scales_hat, noise=None if noise is None else noise.get("y"))
        x_hat = self.g_s(y_hat)
        return {"x_hat": x_hat, "likelihoods": {"y": y_likelihoods, "z": z_likelihoods}}

    def compress(self, x):
        self._sync_precision()
        from . import codec
        if x.shape[0] and codec.hyper_fast_path(self, x.shape[0]):
            return codec.compress_hyper(self, x, chunk=self._chunk_for(x.shape[2], x.shape[3]))
        y = self.g_a(x)
        z = self.h_a(y)
        z_strings = self.entropy_bottleneck.compress(z)
        z_hat = self.entropy_bottleneck.decompress(z_strings, z.size()[-2:])
        scales_hat = self.h_s(z_hat)
        indexes = self.gaussian_conditional.build_indexes_interleaved(scales_hat)
        y_strings = self.gaussian_conditional.compress(y, indexes)
        return {"strings": [y_strings, z_strings], "shape": z.size()[-2:]}

    def decompress(self, strings, shape):
        assert isinstance(strings, list) and len(strings) == 2
        self._sync_precision()
        from . import codec
        if len(strings[0]) and codec.hyper_fast_path(self, len(strings[0])):
            return codec.decompress_hyper(self, strings, shape, chunk=self._chunk_for(shape[0] * 64, shape[1] * 64))
        z_hat = self.entropy_bottleneck.decompress(strings[1], shape)
        scales_hat = self.h_s(z_hat)
        indexes = self.gaussian_conditional.build_indexes_interleaved(scales_hat)
        y_hat = self.gaussian_conditional.decompress(strings[0], indexes, tuple(scales_hat.shape[1:]))
        x_hat = self.g_s(y_hat).clamp_(0, 1)
        return {"x_hat": x_hat}
