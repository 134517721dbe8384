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


class ConvHip(torch.autograd.Function):
    """Conv2d with HIP forward AND backward (dgrad = transposed-conv kernel, wgrad / bias-grad kernels)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad):
        from . import ops
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, b is not None)
        return ops.conv2d_f32(x, w.detach(), None if b is None else b.detach(), stride, pad)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, w = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        dy = dy.contiguous()
        k = w.shape[2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            out_pad = x.shape[2] - ((dy.shape[2] - 1) * stride - 2 * pad + k)
            dx = ops.deconv2d_f32(dy, w.detach(), None, stride, pad, out_pad)
            if dx.shape != x.shape:  # width and height may need different output paddings
                dx = torch.nn.functional.pad(dx, (0, x.shape[3] - dx.shape[3], 0, 0))
        if ctx.needs_input_grad[1]:
            dw = ops.conv2d_wgrad_f32(x, dy, w.shape[1], w.shape[0], k, stride, pad)
        if has_bias and ctx.needs_input_grad[2]:
            db = ops.bias_grad_f32(dy)
        return dx, dw, db, None, None


class DeconvHip(torch.autograd.Function):
    """ConvTranspose2d with HIP forward and backward (dgrad = conv kernel on the same weights)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, out_pad):
        from . import ops
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, b is not None)
        return ops.deconv2d_f32(x, w.detach(), None if b is None else b.detach(), stride, pad, out_pad)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, w = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        dy = dy.contiguous()
        k = w.shape[2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_f32(dy, w.detach(), None, stride, pad)
        if ctx.needs_input_grad[1]:
            dw = ops.conv2d_wgrad_f32(dy, x, w.shape[1], w.shape[0], k, stride, pad)
        if has_bias and ctx.needs_input_grad[2]:
            db = ops.bias_grad_f32(dy)
        return dx, dw, db, None, None, None


class GdnHip(torch.autograd.Function):
    """GDN / IGDN with HIP forward and backward, incl. the reparametrisation chain."""

    @staticmethod
    def forward(ctx, x, beta_raw, gamma_raw, inverse, beta_bound, gamma_bound, pedestal):
        from . import ops
        x = x.contiguous()
        beta, gamma = ops.gdn_reparam_f32(beta_raw.detach(), gamma_raw.detach(), beta_bound, gamma_bound, pedestal)
        ctx.save_for_backward(x, beta_raw, gamma_raw, beta, gamma)
        ctx.cfg = (inverse, beta_bound, gamma_bound)
        return ops.gdn_f32(x, gamma, beta, inverse)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, beta_raw, gamma_raw, beta, gamma = ctx.saved_tensors
        inverse, beta_bound, gamma_bound = ctx.cfg
        c = x.shape[1]
        dx, t = ops.gdn_bwd_f32(x, dy.contiguous(), gamma, beta, inverse)
        dbeta = dgamma = None
        if ctx.needs_input_grad[1]:
            dbeta = ops.reparam_bwd_f32(beta_raw.detach(), ops.bias_grad_f32(t), beta_bound)
        if ctx.needs_input_grad[2]:
            dg_eff = ops.conv2d_wgrad_f32(x, t, c, c, 1, 1, 0, square_input=True).reshape(c, c)
            dgamma = ops.reparam_bwd_f32(gamma_raw.detach(), dg_eff, gamma_bound)
        return (dx if ctx.needs_input_grad[0] else None), dbeta, dgamma, None, None, None, None


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
