"""Headline benchmark: 256x256 tiles/s, encode + decode (BASELINE.json metric), on N MI355X.

One step = compress(x) -> byte strings -> decompress(strings) for one batch of B synthetic 3x256x256 tiles per GPU
that are already resident in HBM (BASELINE.json configs[1]: bmshj2018_factorized q=3, 3 channels, fp16 MFMA path).
Tiles shard across ranks with no data-path collective (weak scaling: B tiles per GPU).  Rank 0 prints ONE JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Invoked bare with --gpus N > 1 (no WORLD_SIZE in the environment) the process never touches the GPU: it starts N fresh
worker processes of itself (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relays rank 0's line and
exits with the workers' status.

Besides the headline the N = 1 line carries (SURVEY.md section 8(d) grid; all timed AFTER the headline region):
  batches            encode+decode ms and tiles/s at B = 1, 16, 64, 1024 (reference batch sizes: eval B = 1,
                     cfg/raw_merged.toml batch_size 16, test_batch_size 64) and at twice the headline's step
  whole_granule      one 1 x 2304 x 2592 band image (eval_script.py's call): as ONE stream, and tiled to 256x256
  decode_from_plain_bytes   the headline decode fed a plain list[bytes] (re-join + staging) instead of PackedStrings
  configs            1-channel (raw split) and 13-channel (raw merged) models, fp32 parity path
  train_step         cfg/raw_merged.toml's step: 16 patches of 13 x 256 x 256, forward + RD loss + backward + clip + Adam x2
  cpu_baseline       the oracle on this host: 1 thread and all share threads, B = 1 and 16, encode / decode split
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0           # MI355X dense fp16 MFMA (MI355X_MICROARCH.md chip table)
XGMI_PEAK_GBPS = 7 * 153.0         # per GPU: 7 links x ~153 GB/s
NATIVE_WATCHDOG_S = float(os.environ.get("LICOS_NATIVE_WATCHDOG_S", "120"))  # the native-RCCL leg's time limit


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=5)  # (the host's share of a call settles on the box's host rate in the first calls)
    ap.add_argument("--batch", type=int, default=16384, help="tiles per GPU per step")
    ap.add_argument("--chunk", type=int, default=4096, help="tiles per pipeline chunk inside a step")
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--quality", type=int, default=3)
    ap.add_argument("--model", default="bmshj2018-factorized", help="zoo name (secondary configs: bmshj2018-hyperprior)")
    ap.add_argument("--size", type=int, default=256, help="tile edge (BASELINE configs[4] uses 512)")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--weights", default="trained", choices=["trained", "synthetic"],
                    help="trained: licos_amd/weights (the repo's own training recipe, tools/train_weights.py) when present")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (skip the measurement grid)")
    ap.add_argument("--native-rccl", action="store_true",
                    help="also time the federated average through liblicos_hip.so's own RCCL communicator "
                         "(licos_allreduce_weighted); off by default: it has only ever run with one rank")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    return ap.parse_args()


def spawn_workers(args, limit_s=3000.0, cmd=None):
    """--gpus N without a launcher: N fresh worker processes, started BEFORE anything here touches the GPU.
    (`cmd`: the worker command line, this script by default - the CPU tests pass a stand-in.)
    The parent watches ALL of them: the first rank to exit non-zero (RCCL initialisation, out of memory, a GPU fault)
    ends the others - left alone they would sit in their next collective until its timeout - and becomes the exit
    status; so does the wall-clock limit.  The workers form their own process group, which is killed as a whole when the
    parent is told to stop (SIGTERM / SIGINT) or exits for any reason, so no rank survives holding a GPU."""
    import atexit
    import signal
    import socket
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, logs = [], []
    out0 = tempfile.TemporaryFile()
    pgid = None

    def kill_group(*_):
        # (pgid is cleared once every worker has been reaped: a process-group id may be reused by strangers afterwards)
        if pgid is not None:
            try:
                os.killpg(pgid, signal.SIGKILL)
            except OSError:  # ProcessLookupError: already gone; PermissionError: no longer ours
                pass

    def on_signal(signum, _frame):
        kill_group()
        raise SystemExit(128 + signum)

    atexit.register(kill_group)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, on_signal)
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        log = tempfile.TemporaryFile()  # every rank's stderr is kept: the failing rank's tail is what explains an exit
        logs.append(log)
        # rank 0 leads a new process group (same session), the others join it
        p = subprocess.Popen(cmd or ([sys.executable, os.path.abspath(__file__)] + sys.argv[1:]), env=env,
                             stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=log,
                             preexec_fn=(lambda g=(pgid or 0): os.setpgid(0, g)))
        if pgid is None:
            pgid = p.pid
        procs.append(p)
    t0 = time.monotonic()
    status = 0
    while True:
        rcs = [p.poll() for p in procs]
        failed = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if failed:
            status = abs(failed[0][1]) or 1
            sys.stderr.write("bench.py: rank %d exited with status %d; stopping the other ranks\n" % failed[0])
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() - t0 > limit_s:
            status = 124
            sys.stderr.write("bench.py: workers still running after %.0f s; stopping them\n" % limit_s)
            break
        time.sleep(0.2)
    if status:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 10
        while time.monotonic() < deadline and any(p.poll() is None for p in procs):
            time.sleep(0.1)
        kill_group()
        for p in procs:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                pass
        for r, log in enumerate(logs):
            log.seek(0)
            tail = log.read().decode("utf-8", "replace")[-1500:]
            if tail.strip():
                sys.stderr.write("---- rank %d stderr (tail) ----\n%s\n" % (r, tail))
    if all(p.poll() is not None for p in procs):
        pgid = None  # every worker reaped: nothing of ours is left under that id
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return status


def stage_flops(kind, cin, cout, h, w, norm=False):
    """Algorithmic FLOPs of one stage per tile, input size h x w: the convolution's MACs x 2 and, when the stage carries
    a fused GDN / IGDN (norm), that layer's cout x cout MACs per output pixel x 2 - SURVEY.md 8(d) counts a tile's work
    as "conv + GDN MACs x 2" (its 11.056 GFLOP per tile is 4823.4 M conv MACs + 704.6 M GDN MACs, both directions)."""
    if kind == "conv3":
        pix_out, taps = h * w, 9
    else:
        pix_out = (h // 2) * (w // 2) if kind == "conv" else (2 * h) * (2 * w)
        taps = 25 if kind == "conv" else 25 / 4.0
    return 2.0 * pix_out * (taps * cin * cout + (cout * cout if norm else 0))


PEAK_HBM_TBPS = 8.0  # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s measured with a float4 copy)


def stage_entry(key, ms, launches):
    """One `stages` entry.  The two end stages move bytes, not FLOPs (the first: 4 MiB of fp16 out per 256^2 tile for 0.5
    GFLOP; the last: 4 MiB in for 0.3 GFLOP): they are priced against the HBM roof - algorithmic bytes = input read once +
    output written once, in the dtypes the kernels use (fp32 image side, fp16 blk16 activations) - the others against the
    MFMA roof."""
    kind, cin, cout, h, w, b, norm = key[:7]
    fl = stage_flops(kind, cin, cout, h, w, norm=norm) * b
    ent = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 2), "launches": launches, "bound": "mfma"}
    few_in, few_out = kind == "conv" and cin <= 16, kind == "deconv" and cout <= 16
    if few_in or few_out:
        if few_in:
            nbytes = (cin * h * w * 4 + cout * (h // 2) * (w // 2) * 2) * b
        else:
            nbytes = (cin * h * w * 2 + cout * (2 * h) * (2 * w) * 4) * b
        ent.update({"bound": "hbm", "GBps": round(nbytes / ms / 1e6, 1), "frac_hbm": round(nbytes / ms / 1e9 / PEAK_HBM_TBPS, 4),
                    "algorithmic_bytes_per_launch": nbytes})
        # HBM-side traffic from the PMC passes of the 3-band 256^2 launches (profiles/README.md), scaled to this launch size
        tfile = None
        if few_in and (cin, h, w) == (3, 256, 256):
            tfile = _traffic_file("pmc_traffic_first.json")
        elif few_out and (cout, h, w) == (3, 128, 128):
            tfile = _traffic_file("pmc_traffic_rows.json")
        elif few_in and (cin, h, w) == (13, 512, 512):
            tfile = _traffic_file("pmc_traffic_first16.json")
        elif few_out and (cout, h, w) == (13, 256, 256):
            tfile = _traffic_file("pmc_traffic_last16.json")
        if tfile and os.path.exists(os.path.join(ROOT, "profiles", tfile)):
            tj = json.load(open(os.path.join(ROOT, "profiles", tfile)))
            ent["traffic"] = tj["hbm_bytes_per_launch"] * b / tj["tiles_per_launch"]
            ent["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % tfile
    return ent


def _traffic_file(name):
    """The newest round's PMC summary of that kernel under profiles/ (rNN_<name>), or None."""
    for rnd in ("r05", "r04", "r03"):
        if os.path.exists(os.path.join(ROOT, "profiles", "%s_%s" % (rnd, name))):
            return "%s_%s" % (rnd, name)
    return None


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


def cpu_baseline(sd, cin, budget_s=22.0):
    """The oracle (torch-CPU conv = the reference's CPU arithmetic; restated EB; C rANS) timed on this host's cores on
    a bounded sample of the same workload: threads in {16 (a one-GPU box's share), 1} x B in {16, 1}, plus every core
    of the affinity mask at B = 16; encode and decode separately, up to 5 repetitions per cell (median).  Returns
    (best cell's tiles/s, seconds spent, its thread count, the grid)."""
    import statistics
    import torch
    from oracle import model as om
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    share = max(1, share)
    x = om.synthetic_tiles(16, cin, 256, seed=0)
    torch.set_num_threads(min(share, 16))
    om.decompress(**{k: v for k, v in om.compress(x[:1], sd).items() if k in ("strings", "shape")}, sd=sd)  # warm-up
    grid, t_start = {}, time.perf_counter()
    # 16 threads = the share of the host a one-GPU box gets; every core of the affinity mask last and bounded (on a shared
    # 256-core host torch's pool at 256 threads is ~10x SLOWER than one thread: oversubscription, reported as measured)
    cells = [(t, b) for t in dict.fromkeys((min(share, 16), 1)) for b in (16, 1)]
    if share > 16:
        cells.append((share, 16))
    for threads, b in cells:
        torch.set_num_threads(threads)
        enc, dec = [], []
        limit = budget_s if threads <= 16 else budget_s + 12.0
        while len(enc) < 5 and ((time.perf_counter() - t_start) < limit or len(enc) < 1):
            t0 = time.perf_counter()
            c = om.compress(x[:b], sd)
            t1 = time.perf_counter()
            om.decompress(c["strings"], c["shape"], sd)
            t2 = time.perf_counter()
            enc.append(t1 - t0)
            dec.append(t2 - t1)
            if threads > 16 and t2 - t0 > 4.0:
                break  # one repetition tells the story
        e, d = statistics.median(enc), statistics.median(dec)
        grid["threads%d_B%d" % (threads, b)] = {
            "encode_tiles_s": round(b / e, 2), "decode_tiles_s": round(b / d, 2), "tiles_s": round(b / (e + d), 2),
            "reps": len(enc), "threads": threads}
    torch.set_num_threads(min(share, 16))
    best = max(grid.values(), key=lambda g: g["tiles_s"])
    return best["tiles_s"], time.perf_counter() - t_start, best["threads"], grid


def timed_codec(net, x, reps, plain=False, split=False, after_warmup=None, warmup=1):
    """Median wall time (ms) of compress(x) + decompress(...) over `reps` runs, device-synchronised.  The first `warmup`
    runs are untimed (they pay this size's allocations); `after_warmup()` runs right behind them - where per-stage event
    collection is armed, so that no cold hipMalloc lands between a stage's two events."""
    import torch
    enc_t, dec_t = [], []
    with torch.no_grad():
        for i in range(1 - warmup, reps + 1):  # runs i <= 0 untimed
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c = net.compress(x)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            strings = [[bytes(s) for s in lst] for lst in c["strings"]] if plain else c["strings"]
            t1b = time.perf_counter()
            d = net.decompress(strings, c["shape"])
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if i > 0:
                enc_t.append(1e3 * (t1 - t0))
                dec_t.append(1e3 * (t2 - t1b))
            elif i == 0 and after_warmup is not None:
                after_warmup()
    enc_t.sort()
    dec_t.sort()
    e, dd = enc_t[len(enc_t) // 2], dec_t[len(dec_t) // 2]
    res = {"ms": round(e + dd, 3), "tiles_s": round(1e3 * x.shape[0] / (e + dd), 1)}
    if split:
        res.update(encode_ms=round(e, 3), decode_ms=round(dd, 3))
    return res, c, d


def extras(args, net, x, dev):
    """The SURVEY 8(d) grid at N = 1 (everything here runs after the timed headline region)."""
    import torch
    import licos_amd
    from licos_amd import synthetic, tiling
    out = {}
    # batch sizes of the reference: eval B = 1, training 16, test 64; plus 1024
    out["batches"] = {}
    for b in (1, 16, 64, 1024):
        if b <= x.shape[0]:
            out["batches"]["B%d" % b] = timed_codec(net, x[:b].contiguous(), 5 if b >= 1024 else 9, split=True)[0]
    # the same cells with the DEVICE coder only (LICOS_HOST_CODER=0: no tile of any call is coded by the host cores): what
    # the MI355X does by itself; the cells above are what the MI355X and 16 threads of its host do together
    from licos_amd import ops
    keep_hc, ops.HOST_CODER = ops.HOST_CODER, "0"
    try:
        out["batches_device_only"] = {}
        for b in (1, 16, 64, 1024):
            if b <= x.shape[0]:
                out["batches_device_only"]["B%d" % b] = timed_codec(net, x[:b].contiguous(), 3 if b >= 1024 else 5, split=True)[0]
        hb_ = min(x.shape[0], args.batch)
        out["batches_device_only"]["B%d" % hb_] = timed_codec(net, x[:hb_], 3, split=True)[0]
    finally:
        ops.HOST_CODER = keep_hc
    # twice the headline's step: the un-hidden coder latency (one decode head + one encode tail per compress /
    # decompress call, ~19 ms whatever the batch) amortised over twice the tiles
    torch.cuda.empty_cache()  # (the step's 13 GB blocks cannot back 26 GB tensors: without this every run re-allocates)
    if x.shape[0] >= 16384 and torch.cuda.mem_get_info(dev)[0] > 160 * 2 ** 30:
        x2 = torch.cat((x, synthetic.tiles(x.shape[0], args.channels, args.size, seed=977, device=dev)))
        out["batches"]["B%d" % x2.shape[0]] = timed_codec(net, x2, 3, split=True)[0]
        del x2
        torch.cuda.empty_cache()
    # the headline decode fed a plain list[bytes] (no PackedStrings shortcut)
    hb = min(x.shape[0], args.batch)
    packed, _, _ = timed_codec(net, x[:hb], 3, split=True)
    plain, _, _ = timed_codec(net, x[:hb], 3, plain=True, split=True)  # (median of 3: a host-side path, noisy on a shared box)
    out["decode_from_plain_bytes"] = {"tiles": hb, "packed": packed, "plain_list": plain}
    # fp32 parity path (split-operand MFMA convolutions, one launch per layer; fp32 GDN / EB)
    if args.precision == "fp16":
        net.set_precision("fp32")
        chunk16 = net.chunk
        net.chunk = 1024  # fp32 NCHW activations: 8.4 MB per tile after the first stage
        out["fp32_path"] = dict(timed_codec(net, x[:256].contiguous(), 3, split=True)[0], tiles=256)
        nb = min(x.shape[0], 16384)
        out["fp32_path_B%d" % nb] = dict(timed_codec(net, x[:nb], 2, split=True)[0], tiles=nb, chunk=1024,
                                         what="the strict-parity path (one-launch split-operand MFMA transforms with the fp32 GDN in the "
                                              "convolution's epilogue, fp32 EB, device coder) through the same chunk pipeline")
        net.chunk = chunk16
        net.set_precision("fp16")
        torch.cuda.empty_cache()
    # BASELINE configs[2]: bmshj2018_factorized q = 5 on single Sentinel-2 bands (raw split), and the same model family
    # on the 13 merged bands; plus one whole granule as the reference feeds it.  Weights: the repo's own training
    # recipe when a checkpoint is shipped (licos_amd/weights/factorized_q5_c{1,13}.pth.tar), else seeded trained-like ones.
    from licos_amd import checkpoint
    configs = {}
    for cin, kind, b in ((1, "s2", 4096), (13, "s2-merged", 2048)):
        torch.manual_seed(42)
        n2 = licos_amd.get_model("bmshj2018-factorized", False, cin, 5).to(dev).eval().set_precision("fp16")
        n2.chunk = args.chunk
        wf = os.path.join(ROOT, "licos_amd", "weights", "factorized_q5_c%d.pth.tar" % cin)
        if args.weights == "trained" and os.path.exists(wf):
            wnote = "trained: " + str(checkpoint.load_checkpoint(wf, n2).get("recipe", os.path.basename(wf)))[:160]
        else:
            with torch.no_grad():
                synthetic.make_trained_like(n2, seed=0)
            wnote = "synthetic trained-like (seeded)"
        xb = synthetic.tiles(b, cin, 256, seed=7, kind=kind, device=dev)
        res, c2, d2 = timed_codec(n2, xb, 3, split=True)
        res.update(tiles=b, quality=5, weights=wnote,
                   bpp_actual=round(8.0 * sum(len(s_) for s_ in c2["strings"][0]) / (b * 256 * 256), 4),
                   psnr_db=round(licos_amd.metrics.compute_psnr(d2["x_hat"], xb), 3))
        del c2, d2
        if not args.no_cpu_baseline:  # "matched": the same 4 tiles through the CPU oracle on the same weights
            sd2 = {k: v.detach().cpu().float() for k, v in n2.state_dict().items()}
            res["quality_match"] = quality_match(n2, sd2, xb[:4].contiguous())
        configs["%dch" % cin] = res
        if cin == 1:
            g = synthetic.tiles(1, 1, 2592, seed=11, kind="s2", device=dev)[:, :, :2304, :].contiguous()  # raw_utils.py:131
            one, c, d = timed_codec(n2, g, 3, split=True)  # eval_script.py:138-165: ONE image, one rANS stream of 4.5 M symbols
            one["bytes"] = sum(len(s) for s in c["strings"][0])
            with torch.no_grad():
                ts = []
                for i in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    coded = tiling.compress_image(n2, g)
                    dec = tiling.decompress_image(n2, coded)
                    torch.cuda.synchronize()
                    ts.append(1e3 * (time.perf_counter() - t0))
            ts = sorted(ts[1:])
            out["whole_granule_1x2304x2592"] = {
                "one_stream": one, "tiled_256": {"ms": round(ts[len(ts) // 2], 3), "tiles": len(coded["strings"][0]),
                                                 "bytes": sum(len(s) for s in coded["strings"][0])}}
        del n2, xb
    torch.cuda.empty_cache()
    try:
        configs["hyperprior_13x512"] = hyperprior_grid(args, dev)
    except (torch.OutOfMemoryError, ValueError, RuntimeError) as e:  # reported, never fatal for the headline
        configs["hyperprior_13x512"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    out["configs"] = configs
    torch.cuda.empty_cache()
    out["train_step"] = train_step_ms(dev, steps=10)[0]
    out["train_step_hyperprior"] = train_step_ms(dev, steps=10, model="bmshj2018-hyperprior")[0]
    return out


def hyperprior_grid(args, dev):
    """BASELINE configs[4]: bmshj2018_hyperprior q=5 on 13-band 512 x 512 tiles (licos/model_utils.py:20-24 admits the
    model), compress() + decompress() through the module API at 2048 / 4096 / 256 tiles per call: tiles/s with the
    encode / decode split, per-stage device times, the dominant MFMA kernel against the fp16 roof, and the serial
    coders' latency per symbol (a y stream is 196 608 symbols, one GPU lane each)."""
    import torch
    import licos_amd
    from licos_amd import checkpoint, codec, engine, synthetic
    torch.manual_seed(42)
    net = licos_amd.get_model("bmshj2018-hyperprior", False, 13, 5).to(dev).eval().set_precision("fp16")
    wfile = os.path.join(ROOT, "licos_amd", "weights", "hyperprior_q5_c13.pth.tar")
    if os.path.exists(wfile) and args.weights == "trained":
        meta = checkpoint.load_checkpoint(wfile, net)
        weights = "trained with the repo's own step: %s" % meta.get("recipe", os.path.basename(wfile))
    else:
        with torch.no_grad():
            synthetic.make_trained_like(net, seed=0)
        weights = "synthetic trained-like (seeded)"
    out = {"weights": weights, "gflop_per_tile": 54.76}
    free = torch.cuda.mem_get_info(dev)[0]
    sizes = [b for b in (2048, 4096, 256) if b * 2 * 13 * 512 * 512 * 4 * 1.6 < free]
    if not sizes:
        return dict(out, error="not enough free device memory for 256 tiles of 13x512x512 (%.1f GiB free)" % (free / 2 ** 30))
    x = synthetic.tiles(max(sizes), 13, 512, seed=300, kind="s2-merged", device=dev)

    def arm():
        engine.stage_events, codec.trace.coder_events = {}, {}

    for b in sizes:
        xb = x[:b]
        detailed = b == 2048 or (2048 not in sizes and b == sizes[0])
        # (two untimed calls: a size's pipeline buffers settle on the second - the allocator still grows there - and
        # the third call of a size has shown a one-off ~60 ms stall of the device on some boxes)
        res, c, d = timed_codec(net, xb, 5, split=True, after_warmup=arm if detailed else None, warmup=3 if b >= 2048 else 1)
        res["timed_calls"], res["untimed_calls"] = 5, 3 if b >= 2048 else 1
        res["gflops_frac_of_peak"] = round(54.76e9 * res["tiles_s"] / 1e12 / PEAK_F16_TFLOPS, 4)
        if engine.stage_events is not None:
            ev, cev = engine.stage_events, codec.trace.coder_events
            engine.stage_events = codec.trace.coder_events = None
            nbytes = sum(len(s) for lst in c["strings"] for s in lst)
            res["bpp_actual"] = round(nbytes * 8.0 / (b * 512 * 512), 4)
            res["psnr_db"] = round(licos_amd.metrics.compute_psnr(d["x_hat"], xb), 3)
            stages, dom = {}, None
            tj = None
            tname = _traffic_file("pmc_traffic_hyper_deconv.json")
            if tname:
                tj = json.load(open(os.path.join(ROOT, "profiles", tname)))
            for key, evs in ev.items():
                ms_all = [e0.elapsed_time(e1) for e0, e1 in evs]
                ms = sum(ms_all) / len(ms_all)
                fl = stage_flops(*key[:5], norm=key[6]) * key[5]
                stages["%s_%d_%d_%dx%d_b%d" % key[:6]] = stage_entry(key, ms, len(evs))
                if key[1] >= 128 and key[2] >= 128 and (dom is None or sum(ms_all) > dom[1]):
                    dom = (key, sum(ms_all), ms, fl)
            res["stages"] = stages
            if dom:
                key, _, ms, fl = dom
                res["roofline"] = {"kernel": "%s_%d_%d_%dx%d" % key[:5], "bound": "mfma", "achieved": round(fl / ms / 1e9, 2),
                                   "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / PEAK_F16_TFLOPS, 4),
                                   "frac_conv_only": round(stage_flops(*key[:5]) * key[5] / ms / 1e9 / PEAK_F16_TFLOPS, 4),
                                   "avg_launch_ms": round(ms, 4), "tiles_per_launch": key[5], "launches": len(ev[key]),
                                   "traffic": (tj["hbm_bytes_per_launch"] * key[5] / tj["tiles_per_launch"]) if tj else None,
                                   "traffic_source": ("profiles/" + tname) if tj else None}
            nsym = {"y": 192 * 32 * 32, "z": 128 * 8 * 8}
            coders = {}
            for key, evs in cev.items():
                ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)[len(evs) // 2]
                coders[key] = {"ms_per_launch": round(ms, 3), "ns_per_symbol": round(1e6 * ms / nsym[key[0]], 1), "launches": len(evs)}
            res["coders"] = coders
        out["B%d" % b] = res
        del c, d
    # the same calls with the device coders only (no host share): the MI355X by itself
    from licos_amd import ops
    keep_hc, ops.HOST_CODER = ops.HOST_CODER, "0"
    try:
        out["device_only"] = {}
        for b in [b_ for b_ in (4096, 256) if b_ in sizes]:
            r_, c, d = timed_codec(net, x[:b], 3, split=True, warmup=2 if b >= 2048 else 1)
            out["device_only"]["B%d" % b] = r_
            del c, d
    finally:
        ops.HOST_CODER = keep_hc
    # "matched": the trained point against the CPU oracle on the same two tiles
    if not args.no_cpu_baseline:
        from oracle import model as om
        sd = {k: v.detach().cpu().float() for k, v in net.state_dict().items()}
        x2 = x[:2].contiguous()
        with torch.no_grad():
            o = net(x2)
        ref = om.hyper_forward(x2.cpu(), sd)
        out["quality_match"] = {"tiles": 2, "bpp_gpu": round(licos_amd.metrics.compute_bpp(o), 5), "bpp_oracle": round(om.compute_bpp(ref), 5),
                                "psnr_gpu": round(licos_amd.metrics.compute_psnr(o["x_hat"].clamp(0, 1), x2), 4),
                                "psnr_oracle": round(om.compute_psnr(ref["x_hat"].clamp(0, 1), x2.cpu()), 4)}
    return out


def train_step_ms(dev, steps=10, world=1, model="bmshj2018-factorized"):
    """cfg/raw_merged.toml's training step on the device (HIP forward and backward, fp32): licos/train.py:186-200.
    `model`: licos/model_utils.py:20-24 admits bmshj2018-hyperprior as well."""
    import gc
    import torch
    import licos_amd
    from licos_amd import synthetic
    gc.collect()
    gc.freeze()  # the step is ~690 launches of host work: keep the collector off everything this process built so far
    torch.manual_seed(42)
    net = licos_amd.get_model(model, False, 13, 1).to(dev).train()
    crit = licos_amd.RateDistortionLoss(lmbda=1e-2)
    opt = licos_amd.net_aux_optimizer(net, {"net": {"type": "Adam", "lr": 1e-4}, "aux": {"type": "Adam", "lr": 1e-3}})
    x = synthetic.tiles(16, 13, 256, seed=1, kind="s2-merged", device=dev)

    def step():
        opt["net"].zero_grad()
        opt["aux"].zero_grad()
        res = crit(net(x), x)
        res["loss"].backward()
        licos_amd.optimizers.clip_grad_norm_(list(net.parameters()), 1.0, opt["net"])
        opt["net"].step()
        net.aux_loss().backward()
        opt["aux"].step()
        return res

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    return {"ms": round(ms, 3), "patches_s": round(16e3 / ms, 1), "loss": round(float(res["loss"].detach()), 4),
            "what": "16 x 13x256x256 patches: forward (noise) + RD loss + backward + clip 1.0 + Adam 1e-4 / aux Adam 1e-3"}, net


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_workers(args))
    # stdout carries ONE JSON line: whatever libraries print there (gloo's rendezvous banner, a build log) goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

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
    weights = "synthetic trained-like (seeded)"
    short = {"bmshj2018-factorized": "factorized", "bmshj2018-factorized-relu": "factorized_relu", "bmshj2018-hyperprior": "hyperprior"}
    wfile = os.path.join(ROOT, "licos_amd", "weights", "%s_q%d_c%d.pth.tar" % (short.get(args.model, args.model), args.quality, args.channels))
    if args.weights == "trained" and os.path.exists(wfile):
        from licos_amd import checkpoint
        meta = checkpoint.load_checkpoint(wfile, net)
        weights = "trained with the repo's own step (tools/train_weights.py): %s" % meta.get("recipe", os.path.basename(wfile))
    else:
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
    if "LICOS_STAGE_EVENTS_MIN" not in os.environ:
        engine.stage_events_min_batch = 512  # the chunk-sized launches: `roofline` and `stages` are about them
    t0 = time.perf_counter()
    step_ends = []
    for _ in range(args.steps):
        comp, dec = step()
        step_ends.append(time.perf_counter())  # (decompress ends with a status read: the step is complete here)
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
    roof = roof_a3 = None
    if events:
        per_stage_ms = {}
        for key, evs in events.items():
            ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)
            per_stage_ms[key] = ms
            stages["%s_%d_%d_%dx%d_b%d" % key[:6]] = stage_entry(key, ms, len(evs))

        def roofline_of(key, kernel_name, traffic_file):
            LB = key[5]
            ms = per_stage_ms[key]
            fl = stage_flops(*key[:5], norm=key[6]) * LB
            fl_conv = stage_flops(*key[:5]) * LB
            ach = fl / (ms * 1e-3) / 1e12
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
                    "algorithmic_flop_per_launch": fl,
                    "algorithmic_flop_note": "convolution MACs x 2 + the fused GDN / IGDN's 128 x 128 MACs per output pixel x 2 "
                                             "(SURVEY.md 8(d): a tile's work is conv + GDN MACs x 2)",
                    "frac_conv_only": round(fl_conv / (ms * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4),
                    "algorithmic_bytes_per_launch": 2 * (key[1] * key[3] * key[4] + key[2] * (key[3] * key[4] * (4 if key[0] == "deconv" else 0.25))) * LB,
                    "power_note": "operand-data dependent (DVFS): the g_s[4] / g_a[2] launches measured 13.78 / 11.96 ms on random operands "
                                  "(socket power 1320 - 1330 W mean, 1400 W peak, sclk 1.65 - 1.78 GHz: rocm-smi) and 10.02 / 8.32 ms on "
                                  "all-zero ones (1100 - 1120 W, 2.39 GHz), same instruction stream and traffic - "
                                  "profiles/r04_power_probe.log (tools/power_probe.py)"}

        # `roofline` = the DOMINANT kernel of the step (largest total time among the MFMA stages, full-size launches);
        # the north-star target kernel g_a[2] rides along as `roofline_g_a2`
        names = {("conv", 128, 128, 128, 128): ("conv5x5s2_mfma8_kernel<4,GDN> (g_a[2], 128->128 @128^2->64^2)",
                                               _traffic_file("pmc_traffic_conv_a3.json") or "none"),
                 ("deconv", 128, 128, 64, 64): ("deconv5x5s2_mfma8_kernel<4,IGDN> (g_s[4], 128->128 @64^2->128^2, 4 phases per workgroup)",
                                               _traffic_file("pmc_traffic_deconv_s4.json") or "none")}
        full = {k: v for k, v in per_stage_ms.items() if k[5] == max(kk[5] for kk in per_stage_ms)}
        dom = max(full, key=lambda k: full[k] * len(events[k]))
        nm = names.get(dom[:5], ("%s_%d_%d_%dx%d" % dom[:5], "none"))
        roof = roofline_of(dom, *nm)
        a3 = [k for k in full if k[:5] == ("conv", 128, 128, 128, 128)]
        if a3:
            roof_a3 = roofline_of(a3[0], *names[("conv", 128, 128, 128, 128)])

    # federated weight averaging step: the blend of the whole flat fp32 state on both schedules -
    # "ring" = one RCCL all-reduce, "direct" = the two-step point-to-point exchange (xGMI is a mesh of direct links) -
    # through torch.distributed and, on RCCL, through the library's own communicator (one C-ABI call per blend)
    fed = None
    if world > 1:
        from licos_amd import federation
        fs = federation.FlatState(net)
        nbytes_bucket = fs.flat.numel() * 4
        reps = 20

        def time_blend(**kw):
            for _ in range(3):
                federation.weighted_average_(fs, 1.0 / world, **kw)
            fence()
            t1 = time.perf_counter()
            for _ in range(reps):
                federation.weighted_average_(fs, 1.0 / world, **kw)
            fence()
            ft = torch.tensor([(time.perf_counter() - t1) / reps], device=dev, dtype=torch.float64)
            dist.all_reduce(ft, op=dist.ReduceOp.MAX)
            busbw = 2 * (world - 1) / world * nbytes_bucket / float(ft.item()) / 1e9
            return {"ms": round(1e3 * float(ft.item()), 4), "busbw_GBps": round(busbw, 2), "frac_of_7x153": round(busbw / XGMI_PEAK_GBPS, 4)}

        fed = {"bucket_bytes": nbytes_bucket, "backend": args.backend,
               "nccl_env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_"))},
               "what": "scale + exchange + normalise of the whole floating state, per averaging step; busbw = 2 (N-1)/N bytes / time"}
        for sched in ("ring", "direct"):
            fed[sched] = time_blend(schedule=sched)
        best = min(("ring", "direct"), key=lambda k: fed[k]["ms"])
        fed.update(ms=fed[best]["ms"], busbw_GBps=fed[best]["busbw_GBps"], frac_of_7x153=fed[best]["frac_of_7x153"], schedule=best)
        # config 4: the training step of cfg/raw_merged.toml on every rank, then the average of its state
        tr, tnet = train_step_ms(dev, steps=5, world=world)
        tfs = federation.FlatState(tnet)
        fence()
        t2 = time.perf_counter()
        for _ in range(5):
            federation.weighted_average_(tfs, 1.0 / world)
        fence()
        tt = torch.tensor([tr["ms"], 1e3 * (time.perf_counter() - t2) / 5], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        fed["config4"] = {"train_step_ms": round(float(tt[0]), 3), "average_ms": round(float(tt[1]), 3),
                          "patches_s_all_ranks": round(16e3 * world / float(tt[0] + tt[1]), 1),
                          "what": "16 x 13x256x256 patches per rank per step (licos/train.py:186-200) + one weight average"}

    cpu = grid = None
    if rank == 0 and world == 1 and args.model == "bmshj2018-factorized" and args.size == 256:
        if not args.no_extras:
            grid = extras(args, net, x, dev)
        if not args.no_cpu_baseline:
            sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
            tps, dt, threads, cgrid = cpu_baseline(sd, args.channels)
            match = quality_match(net, sd, x[:8])
            crops = os.path.join(ROOT, "tests", "golden", "real_crops.npz")
            if args.channels == 3 and os.path.exists(crops):  # the reference's own test photos (tests/golden/make_real_crops.py)
                import numpy as np
                xr = torch.from_numpy(np.load(crops)["x_u8"][:8].astype(np.float32) / 255.0).to(dev)
                match_real = quality_match(net, sd, xr)
                with torch.no_grad():
                    cr = net.compress(xr)
                match_real["bpp_coded"] = round(8.0 * sum(len(s) for s in cr["strings"][0]) / (8 * 256 * 256), 5)
                match["real_photo_crops"] = match_real
            cpu = {"value": tps, "unit": "tiles/s", "cores": threads, "kind": "port", "quality_match": match, "grid": cgrid,
                   "sample": "oracle compress + decompress (torch-CPU conv, C rANS) of the same 3x256x256 workload: B = 16 and "
                             "B = 1 at 16 threads and 1 thread, B = 16 at every core this process may run on; encode / decode timed "
                             "separately, median of up to 5 repetitions per cell, %.1f s in all, on a %d-core host; `value` = the BEST "
                             "cell (%d threads)"
                             % (dt, os.cpu_count() or 0, threads)}

    from licos_amd import codec as _codec
    host_rec = _codec.placement.describe()  # (after every timed region: the factor is what this box's host delivered)
    import threading
    emit_lock = threading.Lock()
    emitted = [False]

    def emit():
        """Prints the line ONCE (the watchdog thread and the main thread may both get here)."""
        with emit_lock:
            if emitted[0] or rank != 0:
                emitted[0] = True
                return
            emitted[0] = True
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
                           "tiles_per_gpu_per_step": B, "precision": args.precision, "weights": weights},
                "bpp_actual": round(bpp, 4), "psnr_db": round(psnr, 3),
                "steps_ms": [round(1e3 * (b_ - a_), 2) for a_, b_ in zip([t0] + step_ends[:-1], step_ends)],
                "roofline": roof, "roofline_g_a2": roof_a3, "cpu_baseline": cpu, "fedavg_allreduce": fed, "grid": grid,
                "stages": stages,
            }
            # the driver keeps the TAIL of the line: the small tables ride at the very end.  `host` says what the host cores
            # contributed (threads, the tiles of a large call they code, the measured slow-down factor of this box's host);
            # `*_device_only` are the same cells with LICOS_HOST_CODER=0 - the MI355X by itself.
            line["host"] = host_rec
            if grid:
                tail = {}
                if "fp32_path_B%d" % min(B, 16384) in grid:
                    tail["fp32_path_B%d" % min(B, 16384)] = grid["fp32_path_B%d" % min(B, 16384)]
                h5 = grid.get("configs", {}).get("hyperprior_13x512", {})
                tail["config5"] = {k: {kk: h5[k][kk] for kk in ("ms", "tiles_s", "encode_ms", "decode_ms") if kk in h5[k]}
                                   for k in ("B4096", "B2048", "B256") if k in h5}
                if "device_only" in h5:
                    tail["config5_device_only"] = h5["device_only"]
                line["tail"] = tail
                if "batches_device_only" in grid:
                    line["batches_device_only"] = grid["batches_device_only"]
                if "batches" in grid:
                    line["batches"] = grid["batches"]
            print(json.dumps(line), file=json_out)
            json_out.flush()

    # The library's own RCCL communicator has never run with more than one rank before the driver's multi-GPU run (one
    # GPU per builder box).  It is timed LAST, under a watchdog: if it has not finished in 120 s - a rank stuck in
    # ncclCommInitRank, say - that is a GPU hang: every rank says on stderr which call it was in, rank 0 prints the line
    # it has (without `native_rccl` figures) and the process exits NON-ZERO (3), so the spawner / the driver stops the
    # peers and records the run as failed.
    if world > 1 and args.backend == "nccl" and fed is not None:
        native_stage = ["federation.NativeComm() (licos_comm_unique_id / licos_comm_init -> ncclCommInitRank)"]
        native_done = [False]

        def bail():
            with emit_lock:
                if native_done[0]:
                    return  # the main thread finished while the timer fired: nothing hung
                fed["native_rccl"] = {"error": "timed out after %.0f s (watchdog) in: %s; the torch.distributed figures "
                                               "above stand; exit status 3" % (NATIVE_WATCHDOG_S, native_stage[0])}
            sys.stderr.write("bench.py: rank %d: native RCCL leg hung in %s; exiting with status 3\n" % (rank, native_stage[0]))
            sys.stderr.flush()
            emit()
            os._exit(3)

        dog = threading.Timer(NATIVE_WATCHDOG_S, bail)
        dog.daemon = True
        dog.start()
        from licos_amd import federation
        res = None
        try:
            with federation.NativeComm() as comm:
                res = {}
                for sched in ("ring", "direct"):
                    native_stage[0] = "licos_allreduce_weighted%s (timing loop)" % ("_direct" if sched == "direct" else "")
                    res[sched] = time_blend(native=comm, schedule=sched)
                res["what"] = "licos_allreduce_weighted / licos_allreduce_weighted_direct: the blend as one C-ABI call"
                # cross-check: the native path leaves the same state as torch.distributed's
                native_stage[0] = "cross-check against torch.distributed"
                probe = fs.flat[:1024].clone()
                federation.weighted_average_(fs, 1.0 / world)
                a = fs.flat[:1024].clone()
                fs.flat[:1024] = probe
                federation.weighted_average_(fs, 1.0 / world, native=comm, schedule="direct")
                res["max_abs_diff_vs_torch"] = float((fs.flat[:1024] - a).abs().max())
                native_stage[0] = "NativeComm.__exit__ (licos_comm_destroy)"
        except Exception as e:  # noqa: BLE001 - reported, never fatal for the headline
            res = {"error": str(e)[:300]}
        dog.cancel()
        with emit_lock:
            native_done[0] = True
            if not emitted[0]:
                fed["native_rccl"] = res
    emit()

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
