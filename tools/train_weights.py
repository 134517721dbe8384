"""Trains an operating point with the repo's own training step and writes it as a LICOS checkpoint.

Recipe = /root/reference/licos/train.py:186-200 with cfg/raw_merged.toml's values: forward with U(-1/2, 1/2) noise,
lambda * 255^2 * MSE + bpp (lambda = 1e-2), backward, clip_grad_norm 1.0, Adam 1e-4 on the network, Adam 1e-3 on the
entropy bottleneck's quantiles (aux loss), batches of 16 patches of 256 x 256.  The data are the seeded synthetic
AID-style tiles of SURVEY.md section 8(d) (licos_amd/synthetic.py; there is no dataset in this image), a fresh batch
per step.  Everything runs through the HIP forward / backward kernels (fp32 path) and the fused Adam.

  python tools/train_weights.py [--steps 12000] [--channels 3] [--quality 3] [--out licos_amd/weights/...]

The checkpoint is the reference's dict ({"batch_idx", "state_dict", "loss", "local_time"} + a "recipe" string), floats
stored as fp16 to keep the file at 6 MB; bench.py loads it when present (--weights trained)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import licos_amd  # noqa: E402
from licos_amd import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12000)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--quality", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lmbda", type=float, default=1e-2)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--out", default=None)
    ap.add_argument("--log", default=None, help="JSON lines: step, loss, mse, bpp, aux")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(42)  # train.py:28-30
    net = licos_amd.get_model("bmshj2018-factorized", False, args.channels, args.quality).to(dev).train()
    crit = licos_amd.RateDistortionLoss(lmbda=args.lmbda)
    opt = licos_amd.net_aux_optimizer(net, {"net": {"type": "Adam", "lr": args.lr}, "aux": {"type": "Adam", "lr": 1e-3}})
    kind = "aid" if args.channels == 3 else ("s2-merged" if args.channels == 13 else "s2")
    log = open(args.log, "w") if args.log else None
    t0 = time.perf_counter()
    hist = []
    for step in range(args.steps):
        x = synthetic.tiles(args.batch, args.channels, 256, seed=1_000_000 + step, kind=kind, device=dev)
        opt["net"].zero_grad()
        opt["aux"].zero_grad()
        res = crit(net(x), x)
        res["loss"].backward()
        licos_amd.optimizers.clip_grad_norm_(list(net.parameters()), 1.0, opt["net"])
        opt["net"].step()
        aux = net.aux_loss()
        aux.backward()
        opt["aux"].step()
        if step % 100 == 0 or step == args.steps - 1:
            rec = {"step": step, "loss": float(res["loss"].detach()), "mse": float(res["mse_loss"].detach()),
                   "bpp": float(res["bpp_loss"].detach()), "aux": float(aux.detach()), "s": round(time.perf_counter() - t0, 1)}
            hist.append(rec)
            if log:
                log.write(json.dumps(rec) + "\n")
                log.flush()
            if step % 1000 == 0 or step == args.steps - 1:
                print(rec, flush=True)
    net.eval()
    net.update(force=True)
    # held-out check through the codec (fp16 path: what bench.py runs)
    net.set_precision("fp16")
    xv = synthetic.tiles(64, args.channels, 256, seed=100, kind=kind, device=dev)
    with torch.no_grad():
        c = net.compress(xv)
        d = net.decompress(c["strings"], c["shape"])
    bpp = 8.0 * sum(len(s) for s in c["strings"][0]) / (64 * 256 * 256)
    psnr = licos_amd.metrics.compute_psnr(d["x_hat"], xv)
    print("held-out (bench seed 100, 64 tiles, fp16 codec): %.4f bpp, %.2f dB" % (bpp, psnr), flush=True)
    out = args.out or os.path.join(ROOT, "licos_amd", "weights", "factorized_q%d_c%d.pth.tar" % (args.quality, args.channels))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    params = {n for n, _ in net.named_parameters()}  # parameters as fp16; buffers (pedestals, bounds ~1e-9..1e-11) stay fp32
    sd = {k: (v.detach().cpu().half() if k in params else v.detach().cpu()) for k, v in net.state_dict().items()}
    state = {"batch_idx": args.steps, "state_dict": sd, "loss": hist[-1]["loss"], "local_time": 0.0,
             "recipe": "%d steps of train.py:186-200 (lambda %g, Adam %g / aux 1e-3, clip 1.0, batch %d) on seeded synthetic "
                       "%s tiles; held-out %.3f bpp, %.2f dB" % (args.steps, args.lmbda, args.lr, args.batch, kind, bpp, psnr)}
    torch.save(state, out)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
