"""Dev tool: time of one training step in the shape of licos/train.py:186-200 (cfg/raw_merged.toml: batch 16 of
13 x 256 x 256 patches, lambda 1e-2, clip 1.0, Adam 1e-4 / aux Adam 1e-3) on the GPU (HIP forward and backward,
fp32), next to the same step of the oracle under torch autograd on the CPU.

  python tools/train_bench.py [batch=16] [channels=13] [steps=10] [cpu_steps=1] [model=bmshj2018-factorized]
(the CPU leg is the factorized oracle's: pass cpu_steps = 0 with another model)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd
from licos_amd import synthetic
from oracle import model as om

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
C = int(sys.argv[2]) if len(sys.argv) > 2 else 13
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cpu_steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
MODEL = sys.argv[5] if len(sys.argv) > 5 else "bmshj2018-factorized"
net = licos_amd.get_model(MODEL, False, C, 1).to(dev).train()
crit = licos_amd.RateDistortionLoss(lmbda=1e-2)
opt = licos_amd.net_aux_optimizer(net, {"net": {"type": "Adam", "lr": 1e-4}, "aux": {"type": "Adam", "lr": 1e-3}})
kind = "aid" if C == 3 else ("s2-merged" if C == 13 else "s2")
x = synthetic.tiles(B, C, 256, seed=1, kind=kind, device=dev)


def step():
    opt["net"].zero_grad()
    opt["aux"].zero_grad()
    out = net(x)
    res = crit(out, x)
    res["loss"].backward()
    licos_amd.optimizers.clip_grad_norm_(net.parameters(), 1.0)
    opt["net"].step()
    aux = net.aux_loss()
    aux.backward()
    opt["aux"].step()
    return res


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    res = step()
torch.cuda.synchronize()
gpu_ms = 1e3 * (time.perf_counter() - t0) / steps
print("GPU  train step: %.1f ms for %d x %d x 256 x 256 (%.1f patches/s), loss %.4f" % (gpu_ms, B, C, 1e3 * B / gpu_ms, float(res["loss"].detach())))

if cpu_steps > 0:
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and v.dim() > 0 and "bound" not in k
                and "pedestal" not in k and "target" not in k else v) for k, v in sd.items()}
    xc = x.cpu()
    params = [v for v in leaf.values() if isinstance(v, torch.Tensor) and v.requires_grad]
    optc = torch.optim.Adam(params, lr=1e-4)
    t0 = time.perf_counter()
    for _ in range(cpu_steps):
        optc.zero_grad()
        noise = torch.rand(B, 192, 16, 16) - 0.5
        out = om.forward(xc, leaf, training=True, noise=noise)
        r = om.rate_distortion_loss(out, xc, 1e-2)
        r["loss"].backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        optc.step()
    cpu_ms = 1e3 * (time.perf_counter() - t0) / cpu_steps
    print("CPU  oracle step: %.0f ms on %d threads (%.2f patches/s) -> GPU/CPU %.0fx" % (cpu_ms, threads, 1e3 * B / cpu_ms, cpu_ms / gpu_ms))
