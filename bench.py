"""Headline benchmark: 256x256 tiles/s, encode + decode (BASELINE.json metric), on N MI355X.

One step = compress(x) -> byte strings -> decompress(strings) for one batch of B synthetic
3x256x256 tiles per GPU that are already resident in HBM (BASELINE.json configs[1]:
bmshj2018_factorized q=3, 3 channels, fp16 MFMA path).  Tiles shard across ranks with no
data-path collective (weak scaling: B tiles per GPU).  Rank 0 prints ONE JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_TILE_A3 = 2 * 1677721600  # g_a[2]: 128->128 5x5 s2 at 128^2 -> 64^2
PEAK_F16_TFLOPS = 2500.0           # MI355X dense fp16 MFMA (MI355X_MICROARCH.md chip table)


def stage_flops(kind, cin, cout, h, w):
    """Algorithmic FLOPs of one stage per tile (conv + fused GDN MACs x 2), input size h x w."""
    if kind == "conv3":
        return 2.0 * h * w * 9 * cin * cout
    pix_out = (h // 2) * (w // 2) if kind == "conv" else (2 * h) * (2 * w)
    taps = 25 if kind == "conv" else 25 / 4.0
    return 2.0 * pix_out * taps * cin * cout


def quality_match(net, sd, x8):
    """bpp / PSNR of the GPU fp16 path vs the CPU oracle on the same 8 tiles (the "matched" of the metric)."""
    import torch
    import licos_amd
    from oracle import model as om
    with torch.no_grad():
        out = net(x8)
    ref = om.forward(x8.cpu(), sd)
    return {"tiles": int(x8.shape[0]),
            "bpp_gpu": round(licos_amd.metrics.compute_bpp(out), 5), "bpp_oracle": round(om.compute_bpp(ref), 5),
            "psnr_gpu": round(licos_amd.metrics.compute_psnr(out["x_hat"].clamp(0, 1), x8), 4),
            "psnr_oracle": round(om.compute_psnr(ref["x_hat"].clamp(0, 1), x8.cpu()), 4)}


def cpu_baseline(sd, cin, batch, reps):
    """The oracle (torch-CPU conv = the reference's CPU arithmetic; restated EB; C rANS) timed on
    this host's cores on a bounded sample of the same workload."""
    import torch
    from oracle import model as om
    threads = min(os.cpu_count() or 1, 16)  # the GPU box gives one GPU a 16-core share; more threads only thrash
    torch.set_num_threads(threads)
    x = om.synthetic_tiles(batch, cin, 256, seed=0)
    om.compress(x[:2], sd)  # warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        c = om.compress(x, sd)
        om.decompress(c["strings"], c["shape"], sd)
    dt = time.perf_counter() - t0
    return batch * reps / dt, dt, threads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16384, help="tiles per GPU per step")
    ap.add_argument("--chunk", type=int, default=4096, help="tiles per pipeline chunk inside a step")
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--quality", type=int, default=3)
    ap.add_argument("--model", default="bmshj2018-factorized", help="zoo name (secondary configs: bmshj2018-hyperprior)")
    ap.add_argument("--size", type=int, default=256, help="tile edge (BASELINE configs[4] uses 512)")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import licos_amd
    from licos_amd import _lib, engine, synthetic

    if not os.path.exists(_lib.SO_PATH):  # a source-only checkout: compile the HIP library first (hipcc, ~1 min)
        import __graft_entry__
        __graft_entry__.build()
    _lib.load()  # the HIP library is the product: fail loudly if it is missing
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else local_rank % max(ndev, 1)  # > ndev only in a gloo rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    torch.manual_seed(42)
    net = licos_amd.get_model(args.model, False, args.channels, args.quality)
    net = net.to(dev).eval().set_precision(args.precision)
    net.chunk = args.chunk
    with torch.no_grad():
        synthetic.make_trained_like(net, seed=0)
    B = args.batch
    kind = "aid" if args.channels == 3 else ("s2-merged" if args.channels == 13 else "s2")
    x = synthetic.tiles(B, args.channels, args.size, seed=100 + rank, kind=kind, device=dev)

    def step():
        with torch.no_grad():
            comp = net.compress(x)
            dec = net.decompress(comp["strings"], comp["shape"])
        return comp, dec

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        comp, dec = step()
    fence()
    engine.stage_events = {} if args.precision == "fp16" else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        comp, dec = step()
    fence()
    elapsed = time.perf_counter() - t0
    events, engine.stage_events = engine.stage_events, None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # quality of what was just coded (bpp from the actual bytes; PSNR of the decoded tiles)
    nbytes = sum(len(s) for lst in comp["strings"] for s in lst)
    bpp = nbytes * 8.0 / (B * args.size * args.size)
    psnr = licos_amd.metrics.compute_psnr(dec["x_hat"], x)

    stages = {}
    roof = roof_d2 = None
    if events:
        for key, evs in events.items():
            ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)
            fl = stage_flops(*key[:5]) * key[5]
            stages["%s_%d_%d_%dx%d_b%d" % key] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 2), "launches": len(evs)}
        # the two MFMA kernels the time goes to: g_a[2] (SURVEY.md section 8(d)'s target kernel, `roofline`) and
        # g_s[2], the largest single kernel of the step (`roofline_g_s2`); same algorithmic FLOPs per tile
        def roofline_of(kind, kernel_name, traffic_file):
            keys = [k for k in events if k[:5] == (kind, 128, 128, 128 if kind == "conv" else 64, 128 if kind == "conv" else 64)]
            if not keys:
                return None
            key = max(keys, key=lambda k: k[5])
            LB = key[5]
            ms = sum(e0.elapsed_time(e1) for e0, e1 in events[key]) / len(events[key])
            ach = FLOP_PER_TILE_A3 * LB / (ms * 1e-3) / 1e12
            traffic = None
            tfile = os.path.join(ROOT, "profiles", traffic_file)
            if os.path.exists(tfile):  # PMC passes cannot run inside the timed process; see profiles/README.md
                tj = json.load(open(tfile))
                traffic = tj["hbm_bytes_per_launch"] * LB / tj["tiles_per_launch"]
            return {"kernel": kernel_name, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, scaled "
                                      "to this launch size)" % traffic_file,
                    "avg_launch_ms": round(ms, 4), "launches": len(events[key]), "tiles_per_launch": LB,
                    "algorithmic_flop_per_launch": FLOP_PER_TILE_A3 * LB,
                    "algorithmic_bytes_per_launch": 5242880 * LB}
        roof = roofline_of("conv", "conv5x5s2_mfma8_kernel<4,GDN> (g_a[2], 128->128 @128^2->64^2)", "r01_pmc_traffic_conv_a3.json")
        roof_d2 = roofline_of("deconv", "deconv5x5s2_mfma_kernel<4,2,8,32,IGDN> (g_s[2], 128->128 @64^2->128^2)",
                              "r01_pmc_traffic_deconv_s2.json")

    # federated weight averaging step: one RCCL all-reduce of the flat fp32 state
    fed = None
    if world > 1:
        from licos_amd import federation
        fs = federation.FlatState(net)
        for _ in range(3):
            federation.weighted_average_(fs, 1.0 / world)
        fence()
        t1 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            federation.weighted_average_(fs, 1.0 / world)
        fence()
        ft = torch.tensor([(time.perf_counter() - t1) / reps], device=dev, dtype=torch.float64)
        dist.all_reduce(ft, op=dist.ReduceOp.MAX)
        nbytes_bucket = fs.flat.numel() * 4
        fed = {"ms": round(1e3 * float(ft.item()), 4), "bucket_bytes": nbytes_bucket,
               "busbw_GBps": round(2 * (world - 1) / world * nbytes_bucket / float(ft.item()) / 1e9, 2),
               "what": "scale + RCCL all-reduce(SUM) + normalise of the whole floating state, per averaging step"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "bmshj2018-factorized" and args.size == 256:
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        tps, dt, threads = cpu_baseline(sd, args.channels, 32, 8)
        match = quality_match(net, sd, x[:8])
        cpu = {"value": round(tps, 2), "unit": "tiles/s", "cores": threads, "kind": "port", "quality_match": match,
               "sample": "oracle compress+decompress (torch-CPU conv, C rANS), 8 reps x 32 tiles of the same 3x256x256 "
                         "workload, %.1f s, %d torch threads on a %d-core host" % (dt, threads, os.cpu_count() or 0)}

    if rank == 0:
        value = world * B * args.steps / elapsed
        line = {
            "metric": "256x256 tiles/s encode+decode (bpp+PSNR matched)" if args.size == 256 else
            "%dx%d tiles/s encode+decode" % (args.size, args.size), "value": round(value, 1), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if args.precision == "fp16" else "f32", "data": "synthetic",
            "config": {"workload": "%s q=%d, %d-ch %dx%d tiles, compress()+decompress() through the "
                                   "module API, %d tiles per GPU per step" % (args.model.replace("-", "_"), args.quality,
                                                                              args.channels, args.size, args.size, B),
                       "tiles_per_gpu_per_step": B, "precision": args.precision, "weights": "synthetic trained-like (seeded)"},
            "bpp_actual": round(bpp, 4), "psnr_db": round(psnr, 3),
            "roofline": roof, "roofline_g_s2": roof_d2, "cpu_baseline": cpu, "fedavg_allreduce": fed, "stages": stages,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
