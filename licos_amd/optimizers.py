"""net_aux_optimizer with CompressAI's interface (``optimizers/net_aux.py``), called from
/root/reference/licos/utils.py:65-73."""
import torch
import torch.optim as optim

from . import ops


class FusedAdam(optim.Optimizer):
    """torch.optim.Adam semantics (amsgrad=False, weight_decay=0) with the update done by licos_adam_f32."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = 1.0  # set by clip_grad_norm_ below for the next step only

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                ops.adam_f32(p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2,
                             group["eps"], st["step"], self.grad_scale)
        self.grad_scale = 1.0


def clip_grad_norm_(parameters, max_norm, optimizer=None):
    """torch.nn.utils.clip_grad_norm_ (licos/train.py:194-195) with the norm reduced by licos_sumsq_f32.  With a
    FusedAdam `optimizer` the clip coefficient is folded into its next step instead of rescaling every gradient."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.0)
    acc = torch.zeros(1, device=grads[0].device, dtype=torch.float64)
    for g in grads:
        ops.sumsq_f32(g.contiguous(), acc)
    total = float(acc.sqrt().item())
    coef = min(1.0, max_norm / (total + 1e-6))
    if isinstance(optimizer, FusedAdam):
        optimizer.grad_scale = coef
    elif coef < 1.0:
        for g in grads:
            ops.scale_f32(g, coef)
    return torch.tensor(total)


def net_aux_optimizer(net, conf):
    parameters = {
        "net": {n for n, p in net.named_parameters() if p.requires_grad and not n.endswith(".quantiles")},
        "aux": {n for n, p in net.named_parameters() if p.requires_grad and n.endswith(".quantiles")},
    }
    params_dict = dict(net.named_parameters())
    inter = parameters["net"] & parameters["aux"]
    union = parameters["net"] | parameters["aux"]
    assert len(inter) == 0 and len(union) - len(params_dict.keys()) == 0

    def make(key):
        kwargs = dict(conf[key])
        kind = kwargs.pop("type")
        ps = [params_dict[n] for n in sorted(parameters[key])]
        if kind == "Adam" and ps and all(p.is_cuda for p in ps) and set(kwargs) <= {"lr", "betas", "eps"}:
            return FusedAdam(ps, **kwargs)  # same update rule, one HIP kernel per parameter
        return getattr(optim, kind)(ps, **kwargs)

    return {"net": make("net"), "aux": make("aux")}
