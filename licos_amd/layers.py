"""GDN and the conv factories, mirroring CompressAI ``layers/gdn.py``,
``ops/parametrizers.py``, ``ops/bound_ops.py`` and ``models/utils.py`` (conv/deconv)
as instantiated through /root/reference/licos/model_utils.py:19.

The modules are parameter containers with CompressAI's attribute and state_dict
names; the arithmetic runs in the HIP kernels behind licos_amd.ops.
"""
import torch
import torch.nn as nn

from . import ops


class LowerBound(nn.Module):
    """``max(x, bound)`` with CompressAI's buffer name (``bound``)."""

    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))
        self.bound_value = float(torch.tensor(float(bound), dtype=torch.float32))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self.bound_value = float(self.bound.detach().cpu()[0])


class NonNegativeParametrizer(nn.Module):
    """out = max(x, bound)^2 - pedestal; buffers ``pedestal`` and ``lower_bound.bound``."""

    def __init__(self, minimum=0.0, reparam_offset=2 ** -18):
        super().__init__()
        self.minimum = float(minimum)
        self.reparam_offset = float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.pedestal_value = float(torch.tensor(pedestal, dtype=torch.float32))
        bound = (self.minimum + self.reparam_offset ** 2) ** 0.5
        self.lower_bound = LowerBound(bound)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self.pedestal_value = float(self.pedestal.detach().cpu()[0])


class GDN(nn.Module):
    """Generalised divisive normalisation, y = x * rsqrt(beta + gamma . x^2) (sqrt when inverse)."""

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.in_channels = int(in_channels)
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=beta_min)
        beta = self.beta_reparam.init(torch.ones(in_channels))
        self.beta = nn.Parameter(beta)
        self.gamma_reparam = NonNegativeParametrizer()
        gamma = self.gamma_reparam.init(gamma_init * torch.eye(in_channels))
        self.gamma = nn.Parameter(gamma)
        self._eff_key = None
        self._eff = None
        self._p32_key = None
        self._p32 = None

    def reparam_args(self):
        """(beta_bound, gamma_bound, pedestal) as the kernels take them."""
        return (self.beta_reparam.lower_bound.bound_value, self.gamma_reparam.lower_bound.bound_value,
                self.beta_reparam.pedestal_value)

    def effective(self):
        """Reparametrised (beta, gamma) on the device, cached per parameter version and weights epoch."""
        key = (self.beta.data_ptr(), self.beta._version, self.gamma.data_ptr(), self.gamma._version, ops.weights_epoch())
        if self._eff_key != key:
            bb, gb, ped = self.reparam_args()
            self._eff = ops.gdn_reparam_f32(self.beta.detach(), self.gamma.detach(), bb, gb, ped)
            self._eff_key = key
        return self._eff

    def packed_f32split(self):
        """The operand of a convolution epilogue that applies this layer at fp32 accuracy (ops.EPI_NORM32), cached like
        effective(); None when the channel count is not served."""
        key = (self.beta.data_ptr(), self.beta._version, self.gamma.data_ptr(), self.gamma._version, ops.weights_epoch())
        if self._p32_key != key:
            bb, gb, ped = self.reparam_args()
            self._p32 = ops.pack_gdn_f32split(self.beta.detach(), self.gamma.detach(), bb, gb, ped)
            self._p32_key = key
        return self._p32

    def forward(self, x):
        from . import autograd
        if autograd.needs_grad(x, self.beta, self.gamma):
            bb, gb, ped = self.reparam_args()
            return autograd.GdnHip.apply(x, self.beta, self.gamma, self.inverse, bb, gb, ped)  # HIP forward and backward
        beta, gamma = self.effective()
        return ops.gdn_f32(x.contiguous(), gamma, beta, self.inverse)


def conv(in_channels, out_channels, kernel_size=5, stride=2):  # noqa: E302 (CompressAI models/utils.py conv)
    return nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


def deconv(in_channels, out_channels, kernel_size=5, stride=2):
    return nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              output_padding=stride - 1, padding=kernel_size // 2)


def _square(v, what):
    if isinstance(v, (tuple, list)):
        if len(v) != 2 or v[0] != v[1]:
            raise ValueError(f"licos_amd: only square {what} is supported, got {v}")
        return int(v[0])
    return int(v)


def conv_geometry(m):
    """(kernel, stride, padding[, output_padding]) of a stock conv module, validated."""
    if m.groups != 1 or _square(m.dilation, "dilation") != 1 or getattr(m, "padding_mode", "zeros") != "zeros":
        raise ValueError("licos_amd: groups/dilation/padding_mode other than the defaults are not supported")
    k = _square(m.kernel_size, "kernel")
    s = _square(m.stride, "stride")
    p = _square(m.padding, "padding")
    if isinstance(m, nn.ConvTranspose2d):
        return k, s, p, _square(m.output_padding, "output_padding")
    return k, s, p
