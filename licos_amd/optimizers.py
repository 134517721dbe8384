"""net_aux_optimizer with CompressAI's interface (``optimizers/net_aux.py``), called from
/root/reference/licos/utils.py:65-73."""
import torch.optim as optim


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
        return getattr(optim, kind)((params_dict[n] for n in sorted(parameters[key])), **kwargs)

    return {"net": make("net"), "aux": make("aux")}
