"""Dev tool: the two first-stage kernels (mfma_first16.hip: 13 bands; mfma_first.hip in-place form: 3 bands) alone, timed with HIP
events, optionally reading the per-phase s_memtime stamps of a -DLICOS_STAMPS build (tools/ab_build.sh).

  python tools/first_stamps_probe.py first16 TILES [H]        # 13 x H^2 tiles (default 512)
  python tools/first_stamps_probe.py first TILES              # 3 x 256^2 tiles
Variant library through LICOS_HIP_SO, switches through the kernels' own environment variables (LICOS_FIRST16_YWALK, ...).
One process per variant: `python tools/first_stamps_probe.py sweep` runs the list below as children, one after the other."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def run_first16(B, H):
    import torch
    import licos_amd
    from licos_amd import ops, engine, _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand(B, 13, H, H, device=dev, generator=g)
    w = torch.randn(128, 13, 5, 5, device=dev, generator=g) * 0.1
    bp = ops.pad_bias(torch.zeros(128, device=dev), 128, dev)
    gp = engine._packed_gdn(licos_amd.GDN(128).to(dev))
    wp = ops.pack_conv_w_f16(w)
    lib = _lib.load()
    stamps = lib.licos_debug_first16_stamps if hasattr(lib, "licos_debug_first16_stamps") else None
    ts = []
    for it in range(12):
        if it == 4 and stamps is not None:
            stamps(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = ops.conv5x5s2_first16_nchw_f16(x, wp, bp, gp, ops.EPI_GDN, 128)
        e1.record()
        torch.cuda.synchronize()
        if it >= 4:
            ts.append(e0.elapsed_time(e1))
        del y
    ts.sort()
    gb = B * (13 * H * H * 4 + 128 * (H // 2) ** 2 * 2) / 1e9
    print("first16 B=%d H=%d so=%s duo=%s run=%s: median %.3f ms (min %.3f) -> %.2f TB/s algorithmic (%.1f GB)"
          % (B, H, os.path.basename(os.environ.get("LICOS_HIP_SO", "product")), os.environ.get("LICOS_FIRST16_DUO", "1"),
             os.environ.get("LICOS_FIRST16_RUN", "auto"), ts[len(ts) // 2], ts[0], gb / ts[len(ts) // 2], gb))
    if stamps is not None:
        buf = (ctypes.c_ulonglong * 64)()
        stamps(buf, 0)
        if os.environ.get("LICOS_FIRST16_DUO", "1") != "0":
            names = ["K s%d top+mfma" % si for si in range(5)] + ["K raw_store (3 rounds)", "K waitcnt (5)", "K barrier (5)", "E compute", "E barriers (5)",
                                                                   "E acc_init", "K s0: weight request issue", "K s0: round-0 load issue", "K s0: loop head", "K s0: piece 0", "K s0: piece 1", "K s0: piece 2", "K s0: piece 3", "K s0: piece 4"]
            for g in (0, 1):
                n = max(1, buf[32 * g + 24])
                tot = sum(buf[32 * g + i] for i in range(19))
                print("  duo stamps, group %d, %d tile rows (wave 0 of the group), cycles per tile (8 x 32): total %.0f" % (g, n, tot / n))
                for i in range(19):
                    print("   %-28s %8.0f" % (names[i], buf[32 * g + i] / n))
        else:
            n = max(1, buf[24])
            names = []
            for si in range(5):
                names += ["s%d top+mfma" % si, "s%d raw_store" % si, "s%d waitcnt" % si, "s%d barrier" % si]
            names += ["pix/epi-setup", "epilogue", "acc_init+loop", "-"]
            tot = sum(buf[i] for i in range(24))
            print("  stamps over %d tiles (wave 0 of every workgroup), cycles per tile: total %.0f" % (n, tot / n))
            for i in range(23):
                print("   %-16s %8.0f" % (names[i], buf[i] / n))


def run_first(B):
    import torch
    import licos_amd
    from licos_amd import ops, engine, _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand(B, 3, 256, 256, device=dev, generator=g)
    w = torch.randn(128, 3, 5, 5, device=dev, generator=g) * 0.2
    bp = ops.pad_bias(torch.zeros(128, device=dev), 128, dev)
    gp = engine._packed_gdn(licos_amd.GDN(128).to(dev))
    wp = ops.pack_conv_w_first_f16(w)
    lib = _lib.load()
    stamps = lib.licos_debug_first_stamps if hasattr(lib, "licos_debug_first_stamps") else None
    ts = []
    for it in range(25):
        if it == 5 and stamps is not None:
            stamps(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = ops.conv5x5s2_first_nchw_f16(x, wp, bp, gp, ops.EPI_GDN, 128)
        e1.record()
        torch.cuda.synchronize()
        if it >= 5:
            ts.append(e0.elapsed_time(e1))
        del y
    ts.sort()
    gb = B * (3 * 256 * 256 * 4 + 128 * 128 * 128 * 2) / 1e9
    print("first(in place) B=%d so=%s run=%s: median %.3f ms (min %.3f) -> %.2f TB/s algorithmic (%.1f GB)"
          % (B, os.path.basename(os.environ.get("LICOS_HIP_SO", "product")), os.environ.get("LICOS_FIRST_RUN", "auto"),
             ts[len(ts) // 2], ts[0], gb / ts[len(ts) // 2], gb))
    if stamps is not None:
        buf = (ctypes.c_ulonglong * 32)()
        stamps(buf, 0)
        n = max(1, buf[8])
        names = ["K loop", "wait vmcnt", "barrier 1", "repack+lgkm", "barrier 2", "dma issue + pix", "epilogue", "acc_init+loop"]
        print("  stamps over %d tiles, cycles per tile: total %.0f" % (n, sum(buf[i] for i in range(8)) / n))
        for i in range(8):
            print("   %-16s %8.0f" % (names[i], buf[i] / n))


SWEEP = [
    ({"LICOS_FIRST16_DUO": "1"}, ["first16", "1024"]),
    ({"LICOS_FIRST16_DUO": "1", "LICOS_HIP_SO": "build/ab/liblicos_f16d_young1.so"}, ["first16", "1024"]),
    ({"LICOS_FIRST16_DUO": "1", "LICOS_HIP_SO": "build/ab/liblicos_f16d_young2.so"}, ["first16", "1024"]),
    ({"LICOS_FIRST16_DUO": "1", "LICOS_HIP_SO": "build/ab/liblicos_f16d_young1_stamps.so"}, ["first16", "1024"]),
    ({"LICOS_FIRST16_DUO": "1"}, ["first16", "1024"]),
]


def main():
    mode = sys.argv[1]
    if mode == "sweep":
        for env, args in SWEEP:
            e = dict(os.environ)
            for k, v in env.items():
                e[k] = os.path.join(ROOT, v) if k == "LICOS_HIP_SO" else v
            print("==", env, args, flush=True)
            rc = subprocess.call(["timeout", "-k", "10", "240", sys.executable, os.path.abspath(__file__)] + args, env=e)
            if rc != 0:
                print("child failed with", rc, flush=True)
                sys.exit(rc)
        return
    if mode == "first16":
        run_first16(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 512)
    else:
        run_first(int(sys.argv[2]))


if __name__ == "__main__":
    main()
