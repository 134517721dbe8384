"""Training glue (SURVEY.md 8(a) row H(i), 8(f1) first step): HIP forward, stock-PyTorch backward.

``licos/train.py:186-200`` runs forward -> RateDistortionLoss -> backward -> clip -> Adam on the module.
The forward arithmetic of every op here is the HIP kernel; for the backward pass the op is re-evaluated
with stock PyTorch-ROCm operators under autograd and differentiated there (the survey's sanctioned
first step; dedicated dgrad / wgrad / GDN-backward kernels are the next row).  Nothing here runs unless
a gradient is actually required.
"""
import torch
import torch.nn.functional as F


class _LowerBoundRef(torch.autograd.Function):
    """CompressAI ops/bound_ops.py: max(x, bound) whose gradient passes where x >= bound or grad < 0."""

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


class HipForward(torch.autograd.Function):
    """out = hip_fn(*tensors) in forward; gradients from ref_fn(*tensors) (same maths in torch ops)."""

    @staticmethod
    def forward(ctx, hip_fn, ref_fn, *tensors):
        with torch.no_grad():
            out = hip_fn(*[t.detach() for t in tensors])
        ctx.ref_fn = ref_fn
        ctx.save_for_backward(*tensors)
        ctx.multi = isinstance(out, tuple)
        return out

    @staticmethod
    def backward(ctx, *grads):
        tensors = ctx.saved_tensors
        needs = ctx.needs_input_grad[2:]
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(bool(n) and t.is_floating_point()) for t, n in zip(tensors, needs)]
            out = ctx.ref_fn(*ins)
            outs = out if isinstance(out, tuple) else (out,)
            pairs = [(o, g) for o, g in zip(outs, grads) if g is not None and o.requires_grad]
            wanted = [t for t in ins if t.requires_grad]
            got = torch.autograd.grad([o for o, _ in pairs], wanted, [g for _, g in pairs], allow_unused=True) if wanted and pairs else ()
        it = iter(got)
        res = [next(it) if t.requires_grad else None for t in ins]
        return (None, None, *res)


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def conv_ref(stride, pad, relu, abs_input):
    def fn(x, w, b=None):
        y = F.conv2d(torch.abs(x) if abs_input else x, w, b, stride=stride, padding=pad)
        return F.relu(y) if relu else y
    return fn


def deconv_ref(stride, pad, out_pad, relu):
    def fn(x, w, b=None):
        y = F.conv_transpose2d(x, w, b, stride=stride, padding=pad, output_padding=out_pad)
        return F.relu(y) if relu else y
    return fn


def gdn_ref(inverse, beta_bound, gamma_bound, pedestal):
    def fn(x, beta_raw, gamma_raw):
        bb = torch.tensor([beta_bound], device=x.device, dtype=x.dtype)
        gb = torch.tensor([gamma_bound], device=x.device, dtype=x.dtype)
        beta = lower_bound_ref(beta_raw, bb) ** 2 - pedestal
        gamma = lower_bound_ref(gamma_raw, gb) ** 2 - pedestal
        c = x.shape[1]
        norm = F.conv2d(x ** 2, gamma.reshape(c, c, 1, 1), beta)
        norm = torch.sqrt(norm) if inverse else torch.rsqrt(norm)
        return x * norm
    return fn
