"""Dev tool: is the big transposed-conv stage (128 -> 128 + IGDN at 64^2 -> 128^2) bounded by its instruction stream or by
the chip's power limit?  The same kernel, the same launch, the same instruction stream on (a) random operands, (b) an
all-zero input with random weights, (c) all-zero input and weights: matrix-core power follows operand toggling, so on a
power-limited launch (b) / (c) run at a higher clock and finish sooner; a stream-bound kernel does not care.
  python tools/power_probe.py [tiles=4096] [reps=12]"""
import sys, os, torch, json, subprocess, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd
from licos_amd import ops, engine


class Smi(threading.Thread):
    """Round 4: the limit observed, not inferred - socket power and shader clock from rocm-smi (a reading every ~0.2 s)
    while the launches run.  Prints nothing when the tool is not there or not readable as this user."""

    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
                card = next(iter(json.loads(out).values()))
                pw = next((float(v) for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), None)
                sclk = next((v for k, v in card.items() if k.lower().startswith("sclk")), None)
                self.rows.append((pw, sclk))
            except Exception:  # noqa: BLE001
                self.rows.append((None, None))
            time.sleep(0.15)

    def summary(self):
        pw = [p for p, _ in self.rows if p is not None]
        ck = [c for _, c in self.rows if c]
        return "socket power %s W (max %s) over %d readings, sclk readings %s" % (
            round(sum(pw) / len(pw), 1) if pw else "n/a", round(max(pw), 1) if pw else "n/a", len(self.rows), sorted(set(ck))[-3:] if ck else "n/a")


B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
gp = engine._packed_gdn(licos_amd.GDN(128, inverse=True).to(dev))
bp = ops.pad_bias(torch.zeros(128, device=dev), 128, dev)
fl = 2.0 * 128 * 128 * (25 / 4.0 * 128 * 128 + 128 * 128) * B
gpf = engine._packed_gdn(licos_amd.GDN(128).to(dev))
fl_a = 2.0 * 64 * 64 * (25.0 * 128 * 128 + 128 * 128) * B
for stage in ("g_s[4] deconv + IGDN 64^2 -> 128^2", "g_a[2] conv + GDN 128^2 -> 64^2"):
    dec = stage.startswith("g_s")
    print(stage)
    for name, xs, ws in (("random x, random w", 1.0, 0.03), ("zero x, random w", 0.0, 0.03), ("zero x, zero w", 0.0, 0.0)):
        hw = 64 if dec else 128
        x = (torch.randn(B, 8, hw, hw, 16, device=dev, generator=g) * xs).half()
        w = torch.randn(128, 128, 5, 5, device=dev, generator=g) * ws
        wp = ops.pack_conv_w_f16(w, transposed=dec)
        ts = []
        smi = Smi()
        smi.start()
        for it in range(reps + 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if dec:
                y = ops.deconv5x5s2_f16(x, wp, bp, gp, ops.EPI_IGDN | ops.EPI_IN_XSPLIT | ops.EPI_OUT_XSPLIT, 128, 128)
            else:
                y = ops.conv5x5s2_f16(x, wp, bp, gpf, ops.EPI_GDN, 128, 128)
            e1.record()
            torch.cuda.synchronize()
            if it >= 3:
                ts.append(e0.elapsed_time(e1))
            del y
        # (keep the chip under this load long enough for a few readings)
        t_end = time.time() + 1.5
        while time.time() < t_end:
            if dec:
                y = ops.deconv5x5s2_f16(x, wp, bp, gp, ops.EPI_IGDN | ops.EPI_IN_XSPLIT | ops.EPI_OUT_XSPLIT, 128, 128)
            else:
                y = ops.conv5x5s2_f16(x, wp, bp, gpf, ops.EPI_GDN, 128, 128)
            torch.cuda.synchronize()
            del y
        smi.stop = True
        smi.join(timeout=6)
        ts.sort()
        med = ts[len(ts) // 2]
        f = fl if dec else fl_a
        print("  %-20s median %.3f ms  min %.3f  max %.3f  -> %.0f TFLOP/s conv + norm MACs (%.3f of 2.5 PF); %s" % (name, med, ts[0], ts[-1], f / med / 1e9, f / med / 1e9 / 2500, smi.summary()), flush=True)
        del x, w, wp
