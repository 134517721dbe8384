"""Dev tool: time one 128->128 stage with and without its (I)GDN epilogue (what does the epilogue cost?)."""
import sys, os, torch, torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from licos_amd import engine
from licos_amd.layers import GDN, conv, deconv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
for kind, hw in (("conv", 128), ("deconv", 64), ("deconv", 32)):
    for epi in (False, True):
        mods = [conv(128, 128) if kind == "conv" else deconv(128, 128)]
        if epi:
            mods.append(GDN(128, inverse=(kind == "deconv")))
        mods.append(conv(128, 128) if kind == "conv" else deconv(128, 128))  # keeps the probed stage off the NCHW-out path
        seq = nn.Sequential(*mods).to(dev).eval()
        x = torch.randn(B, 8, hw, hw, 16, device=dev).half()
        with torch.no_grad():
            for it in range(3):
                engine.stage_events = {} if it == 2 else None
                engine.run_chain_fp16(seq, x_blk=x)
            torch.cuda.synchronize()
        k, v = next(iter(engine.stage_events.items()))
        print("%-7s %3dx%-3d epilogue=%-5s %7.3f ms" % (kind, hw, hw, epi, sum(a.elapsed_time(b) for a, b in v) / len(v)))
        engine.stage_events = None
