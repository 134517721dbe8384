"""Training glue: autograd Functions whose forward AND backward are HIP kernels.

``licos/train.py:186-200`` runs forward -> RateDistortionLoss -> backward -> clip -> Adam on the module.  Every op of
that step that touches a tensor of the model's size is a kernel of liblicos_hip.so: the (transposed) convolutions
(dgrad = the same kernels with the roles swapped, wgrad, bias grad; ReLU / |x| masks fused as point-wise kernels), GDN
/ IGDN incl. the reparametrisation chain, the entropy bottleneck's factorised likelihood (per-channel MLP, analytic
backward with in-kernel reductions over the batch) and the Gaussian conditional's.  PyTorch's autograd only strings
them together (and differentiates the scalar loss arithmetic on top).  Nothing here runs unless a gradient is required.
"""
import os

import torch
import torch.nn.functional as F


class ConvHip(torch.autograd.Function):
    """Conv2d with HIP forward AND backward (dgrad = transposed-conv kernel, wgrad / bias-grad kernels).  `relu` fuses a
    ReLU behind it, `abs_input` feeds |x| (ScaleHyperprior.h_a): both are masks on the gradient."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, relu=False, abs_input=False):
        from . import ops
        x = x.contiguous()
        y = ops.conv2d_f32(x, w.detach(), None if b is None else b.detach(), stride, pad, relu, abs_input=abs_input)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.cfg = (stride, pad, b is not None, relu, abs_input)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, w, y = ctx.saved_tensors
        stride, pad, has_bias, relu, abs_input = ctx.cfg
        dy = dy.contiguous()
        if relu:
            dy = ops.mask_mul_f32(dy, y, "relu")
        k = w.shape[2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # output_padding per axis: an odd height with an even width (or the reverse) needs different ones
            oph = x.shape[2] - ((dy.shape[2] - 1) * stride - 2 * pad + k)
            opw = x.shape[3] - ((dy.shape[3] - 1) * stride - 2 * pad + k)
            dx = ops.deconv2d_f32(dy, w.detach(), None, stride, pad, max(oph, opw))
            if dx.shape != x.shape:
                dx = dx[:, :, : x.shape[2], : x.shape[3]].contiguous()
            if abs_input:
                dx = ops.mask_mul_f32(dx, x, "abs")
        if ctx.needs_input_grad[1]:
            xin = ops.mask_mul_f32(x, x, "abs") if abs_input else x  # x * sign(x) = |x|
            dw = ops.conv2d_wgrad_f32(xin, dy, w.shape[1], w.shape[0], k, stride, pad)
        if has_bias and ctx.needs_input_grad[2]:
            db = ops.bias_grad_f32(dy)
        return dx, dw, db, None, None, None, None


class DeconvHip(torch.autograd.Function):
    """ConvTranspose2d with HIP forward and backward (dgrad = conv kernel on the same weights); `relu` as above."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, out_pad, relu=False):
        from . import ops
        x = x.contiguous()
        y = ops.deconv2d_f32(x, w.detach(), None if b is None else b.detach(), stride, pad, out_pad, relu)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.cfg = (stride, pad, b is not None, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, w, y = ctx.saved_tensors
        stride, pad, has_bias, relu = ctx.cfg
        dy = dy.contiguous()
        if relu:
            dy = ops.mask_mul_f32(dy, y, "relu")
        k = w.shape[2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_f32(dy, w.detach(), None, stride, pad)
        if ctx.needs_input_grad[1]:
            dw = ops.conv2d_wgrad_f32(dy, x, w.shape[1], w.shape[0], k, stride, pad)
        if has_bias and ctx.needs_input_grad[2]:
            db = ops.bias_grad_f32(dy)
        return dx, dw, db, None, None, None, None


class EbLikelihoodHip(torch.autograd.Function):
    """EntropyBottleneck in training mode: outputs = x + noise, likelihoods of the factorised density; backward =
    licos_eb_likelihood_bwd (analytic, incl. CompressAI's LowerBound gradient rule).  Parameters arrive as
    (*matrices, *biases, *factors)."""

    @staticmethod
    def forward(ctx, x, noise, medians, filters, channels, bound, form, sum_log2, *params):
        from . import ops
        nl = len(filters) + 1
        outs = ops.eb_quantize(x.contiguous(), medians, "noise", noise=noise)
        packed = ops.eb_pack([p.detach() for p in params[:nl]], [p.detach() for p in params[nl:2 * nl]],
                             [p.detach() for p in params[2 * nl:]], filters, channels)
        lik = ops.eb_likelihood(outs, packed, filters, bound, form, sum_log2)
        ctx.save_for_backward(outs, packed)
        ctx.cfg = (tuple(filters), bound, form, [tuple(p.shape) for p in params])
        return outs, lik

    @staticmethod
    def backward(ctx, g_out, g_lik):
        from . import ops
        outs, packed = ctx.saved_tensors
        filters, bound, form, shapes = ctx.cfg
        if g_lik is None:
            g_lik = torch.zeros_like(outs)
        dv, dp = ops.eb_likelihood_bwd(outs, g_lik, packed, filters, bound, form)
        dx = dv if g_out is None else dv + g_out  # outputs = x + noise
        # split the per-channel record (matrix, bias, factor per layer) back into the parameters' shapes
        f = (1,) + filters + (1,)
        nl = len(filters) + 1
        mats, bias, facs = [], [], []
        off = 0
        for i in range(nl):
            rows, cols = f[i + 1], f[i]
            mats.append(dp[:, off: off + rows * cols].reshape(-1, rows, cols))
            off += rows * cols
            bias.append(dp[:, off: off + rows].reshape(-1, rows, 1))
            off += rows
            if i < nl - 1:
                facs.append(dp[:, off: off + rows].reshape(-1, rows, 1))
                off += rows
        grads = [g.contiguous() if n else None for g, n in zip(mats + bias + facs, ctx.needs_input_grad[8:])]
        return (dx if ctx.needs_input_grad[0] else None, None, None, None, None, None, None, None, *grads)


class GcLikelihoodHip(torch.autograd.Function):
    """GaussianConditional: outputs = x + noise (training) or round(x), likelihoods given the predicted scales; backward
    = licos_gc_likelihood_bwd (gradients w.r.t. the latents and the scales, both LowerBounds' gradient rules)."""

    @staticmethod
    def forward(ctx, x, scales, noise, training, scale_bound, lik_bound, sum_log2):
        from . import ops
        zeros = torch.zeros(x.shape[1], device=x.device, dtype=torch.float32)
        if training:
            outs = ops.eb_quantize(x.contiguous(), zeros, "noise", noise=noise)
        else:
            outs = ops.eb_quantize(x.contiguous(), zeros, "dequantize")
        scales = scales.contiguous()
        lik = ops.gc_likelihood(outs, scales, scale_bound, lik_bound, sum_log2)
        ctx.save_for_backward(outs, scales)
        ctx.cfg = (training, scale_bound, lik_bound)
        return outs, lik

    @staticmethod
    def backward(ctx, g_out, g_lik):
        from . import ops
        outs, scales = ctx.saved_tensors
        training, scale_bound, lik_bound = ctx.cfg
        if g_lik is None:
            g_lik = torch.zeros_like(outs)
        dv, ds = ops.gc_likelihood_bwd(outs, scales, g_lik, scale_bound, lik_bound)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = (dv if g_out is None else dv + g_out) if training else torch.zeros_like(outs)  # d round(x)/dx = 0
        return dx, (ds if ctx.needs_input_grad[1] else None), None, None, None, None, None


GDN_BWD_FUSED = os.environ.get("LICOS_GDN_BWD_FUSED", "1") != "0"  # A/B switch: the layer-by-layer backward (three-pass 1x1 products)


class GdnHip(torch.autograd.Function):
    """GDN / IGDN with HIP forward and backward, incl. the reparametrisation chain."""

    @staticmethod
    def forward(ctx, x, beta_raw, gamma_raw, inverse, beta_bound, gamma_bound, pedestal):
        from . import ops
        x = x.contiguous()
        beta, gamma = ops.gdn_reparam_f32(beta_raw.detach(), gamma_raw.detach(), beta_bound, gamma_bound, pedestal)
        ctx.cfg = (inverse, beta_bound, gamma_bound)
        if GDN_BWD_FUSED and x.dim() == 4 and ops.gdn_f32_split3_applies(x.shape[1], x.shape[2] * x.shape[3]):
            # one-pass kernels: the forward keeps norm = beta + gamma . x^2 for the backward (mfma_gdn_bwd_f32.hip)
            y, norm = ops.gdn_f32_fwd_norm(x, gamma, beta, inverse)
            ctx.save_for_backward(x, beta_raw, gamma_raw, beta, gamma, norm)
            return y
        ctx.save_for_backward(x, beta_raw, gamma_raw, beta, gamma)
        return ops.gdn_f32(x, gamma, beta, inverse)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, beta_raw, gamma_raw, beta, gamma = ctx.saved_tensors[:5]
        inverse, beta_bound, gamma_bound = ctx.cfg
        c = x.shape[1]
        dg_eff = None
        if len(ctx.saved_tensors) == 6:
            dx, t, dg_eff = ops.gdn_bwd_fused_f32(x, dy.contiguous(), ctx.saved_tensors[5], gamma, inverse,
                                                  want_dgamma=ctx.needs_input_grad[2] and ops.GDN_MFMA)
        else:
            dx, t = ops.gdn_bwd_f32(x, dy.contiguous(), gamma, beta, inverse)
        dbeta = dgamma = None
        if ctx.needs_input_grad[1]:
            dbeta = ops.reparam_bwd_f32(beta_raw.detach(), ops.bias_grad_f32(t), beta_bound)
        if ctx.needs_input_grad[2]:
            if dg_eff is None:
                dg_eff = ops.conv2d_wgrad_f32(x, t, c, c, 1, 1, 0, square_input=True).reshape(c, c)
            dgamma = ops.reparam_bwd_f32(gamma_raw.detach(), dg_eff, gamma_bound)
        return (dx if ctx.needs_input_grad[0] else None), dbeta, dgamma, None, None, None, None


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
