"""Trains an operating point with the repo's own training step and writes it as a LICOS checkpoint.

Recipe = /root/reference/licos/train.py:186-200 with cfg/raw_merged.toml's values: forward with U(-1/2, 1/2) noise,
lambda * 255^2 * MSE + bpp, backward, clip_grad_norm 1.0, Adam 1e-4 on the network, Adam 1e-3 on the entropy
bottleneck's quantiles (aux loss), batches of 16 patches of 256 x 256 (train.py:33-39 RandomCrop(256)).  Data:

  synthetic   the seeded synthetic tiles of SURVEY.md section 8(d) (licos_amd/synthetic.py), a fresh batch per step;
  real        256 x 256 crops of the reference's own test photos (tests/golden/make_real_crops.py --pool writes the pool;
              3 channels only), random flips / transposes per draw;
  mix         every other batch of each.

Everything runs through the HIP forward / backward kernels (fp32 path) and the fused Adam.

  python tools/train_weights.py [--model bmshj2018-factorized|bmshj2018-hyperprior] [--steps 12000] [--channels 3]
                                [--quality 3] [--lmbda 1e-2] [--data synthetic] [--init CKPT] [--out ...]

The checkpoint is the reference's dict ({"batch_idx", "state_dict", "loss", "local_time"} + a "recipe" string), floats
stored as fp16 to keep the file small; bench.py loads it when present (--weights trained)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import licos_amd  # noqa: E402
from licos_amd import synthetic  # noqa: E402

SHORT = {"bmshj2018-factorized": "factorized", "bmshj2018-factorized-relu": "factorized_relu", "bmshj2018-hyperprior": "hyperprior"}


def weights_path(model, quality, channels):
    return os.path.join(ROOT, "licos_amd", "weights", "%s_q%d_c%d.pth.tar" % (SHORT[model], quality, channels))


class RealCrops:
    """Batches of 256 x 256 crops from an (N, 3, S, S) uint8 pool, S >= 256, with the 8 flips / transposes."""

    def __init__(self, path, device, seed=0):
        self.pool = torch.from_numpy(np.load(path)["x_u8"]).to(device)
        self.gen = torch.Generator(device="cpu").manual_seed(seed)

    def batch(self, n):
        idx = torch.randint(0, self.pool.shape[0], (n,), generator=self.gen)
        s = self.pool.shape[-1]
        out = []
        for i in idx.tolist():
            y0, x0 = (int(torch.randint(0, s - 255, (1,), generator=self.gen)) for _ in range(2))
            t = self.pool[i, :, y0:y0 + 256, x0:x0 + 256]
            k = int(torch.randint(0, 8, (1,), generator=self.gen))
            if k & 1:
                t = t.flip(-1)
            if k & 2:
                t = t.flip(-2)
            if k & 4:
                t = t.transpose(-1, -2)
            out.append(t)
        return (torch.stack(out).float() / 255.0).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="bmshj2018-factorized", choices=sorted(SHORT))
    ap.add_argument("--steps", type=int, default=12000)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--quality", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lmbda", type=float, default=1e-2)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "real", "mix"])
    ap.add_argument("--pool", default=os.path.join(ROOT, "build", "real_pool.npz"))
    ap.add_argument("--init", default=None, help="checkpoint to start from (fine-tuning)")
    ap.add_argument("--eval-size", type=int, default=256, help="tile edge of the held-out codec check")
    ap.add_argument("--out", default=None)
    ap.add_argument("--log", default=None, help="JSON lines: step, loss, mse, bpp, aux")
    ap.add_argument("--minutes", type=float, default=0.0, help="stop early once this much wall time is spent (0 = never)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(42)  # train.py:28-30
    net = licos_amd.get_model(args.model, False, args.channels, args.quality).to(dev)
    if args.init:
        from licos_amd import checkpoint
        checkpoint.load_checkpoint(args.init, net, update=False)
        for p in net.parameters():  # shipped checkpoints store fp16 parameters
            p.data = p.data.float()
    net.train()
    crit = licos_amd.RateDistortionLoss(lmbda=args.lmbda)
    opt = licos_amd.net_aux_optimizer(net, {"net": {"type": "Adam", "lr": args.lr}, "aux": {"type": "Adam", "lr": 1e-3}})
    kind = "aid" if args.channels == 3 else ("s2-merged" if args.channels == 13 else "s2")
    real = RealCrops(args.pool, dev) if args.data != "synthetic" else None
    if real is not None and args.channels != 3:
        raise SystemExit("the real-photo pool is RGB: --channels 3")
    log = open(args.log, "w") if args.log else None
    t0 = time.perf_counter()
    hist = []
    for step in range(args.steps):
        if args.minutes and time.perf_counter() - t0 > 60 * args.minutes:
            args.steps = step
            print("time budget reached: stopping after", step, "steps", flush=True)
            break
        if real is not None and (args.data == "real" or step % 2):
            x = real.batch(args.batch)
        else:
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
    nv = 64 if args.eval_size <= 256 else 16
    xv = synthetic.tiles(nv, args.channels, args.eval_size, seed=100, kind=kind, device=dev)
    with torch.no_grad():
        c = net.compress(xv)
        d = net.decompress(c["strings"], c["shape"])
    bpp = 8.0 * sum(len(s) for lst in c["strings"] for s in lst) / (nv * args.eval_size * args.eval_size)
    psnr = licos_amd.metrics.compute_psnr(d["x_hat"], xv)
    note = "held-out synthetic (seed 100, %d tiles of %d^2, fp16 codec): %.4f bpp, %.2f dB" % (nv, args.eval_size, bpp, psnr)
    print(note, flush=True)
    if hasattr(net, "gaussian_conditional"):
        with torch.no_grad():
            net.set_precision("fp32")
            y = net.g_a(xv[:4])
            sc = net.h_s(net.entropy_bottleneck(net.h_a(y))[0])
            idx = net.gaussian_conditional.build_indexes(sc)
        hist_idx = torch.bincount(idx.flatten().long(), minlength=64).float()
        hist_idx /= hist_idx.sum()
        print("scale-index histogram (fraction per table row):", [round(float(v), 4) for v in hist_idx], flush=True)
    out = args.out or weights_path(args.model, args.quality, args.channels)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    params = {n for n, _ in net.named_parameters()}  # parameters as fp16; buffers (pedestals, bounds ~1e-9..1e-11) stay fp32
    sd = {k: (v.detach().cpu().half() if k in params else v.detach().cpu()) for k, v in net.state_dict().items()}
    state = {"batch_idx": args.steps, "state_dict": sd, "loss": hist[-1]["loss"], "local_time": 0.0,
             "recipe": "%s: %d steps of train.py:186-200 (lambda %g, Adam %g / aux 1e-3, clip 1.0, batch %d) on %s%s; %s"
                       % (args.model, args.steps, args.lmbda, args.lr, args.batch,
                          {"synthetic": "seeded synthetic %s tiles" % kind, "real": "crops of the reference's test photos",
                           "mix": "seeded synthetic %s tiles and crops of the reference's test photos, alternating" % kind}[args.data],
                          (", from " + os.path.basename(args.init)) if args.init else "", note)}
    torch.save(state, out)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
