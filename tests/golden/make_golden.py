"""Generates the golden vectors under tests/golden/ from the CPU oracle.

The reference itself cannot be run here (its hot path lives in CompressAI, which is not
installed and cannot be fetched), so these vectors pin the ORACLE
(regression anchors + inputs for the GPU parity tests), not the reference: parity with a
real CompressAI build stays "unpinned".  Run from the repo root:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import model as om  # noqa: E402
from oracle import rans  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)  # fixed reduction order inside torch's CPU kernels


def state_checksum(sd):
    return float(sum(v.double().abs().sum() for k, v in sorted(sd.items()) if v.dtype.is_floating_point))


def make_model_case(name, in_channels, size, seed, kind, form="plain"):
    sd = om.perturb_state(om.make_factorized_state(in_channels, quality=1, seed=42), seed=7)
    om.eb_update(sd, form=form)
    x = om.synthetic_tiles(2, in_channels, size, seed=seed, kind=kind)
    out = om.forward(x, sd, form=form)
    comp = om.compress(x, sd)
    dec = om.decompress(comp["strings"], comp["shape"], sd)
    sym = om.eb_symbols(out["y"], sd)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        in_channels=in_channels, size=size, seed=seed, kind=kind, form=form,
        state_checksum=state_checksum(sd),
        x_u8=np.rint(x.numpy() * 255).astype(np.uint8),
        y=out["y"].numpy(), symbols=sym.numpy().astype(np.int32), lik=out["likelihoods"]["y"].numpy(),
        x_hat=out["x_hat"].numpy(), x_dec=dec["x_hat"].numpy(),
        bpp=om.compute_bpp(out), psnr=om.compute_psnr(out["x_hat"].clamp(0, 1), x),
        cdf=sd["entropy_bottleneck._quantized_cdf"].numpy(), cdf_len=sd["entropy_bottleneck._cdf_length"].numpy(),
        offset=sd["entropy_bottleneck._offset"].numpy(),
        string0=np.frombuffer(comp["strings"][0][0], dtype=np.uint8),
        string1=np.frombuffer(comp["strings"][0][1], dtype=np.uint8),
        aux_loss=float(om.eb_aux_loss(sd)),
    )
    print(name, "bpp", om.compute_bpp(out), "bytes", [len(s) for s in comp["strings"][0]],
          "sym range", int(sym.min()), int(sym.max()))


def make_coder_kat():
    """Coder-only known-answer test on hand-made CDFs that hits: negative offsets, values at and
    beyond max (bypass), bypass counts >= 15 nibbles are impossible for int32 (max 8), so the
    15-chain is exercised through the decoder with a hand-built stream in tests instead."""
    cdfs = np.zeros((3, 8), dtype=np.int32)
    cdfs[0, :4] = [0, 30000, 65000, 65536]             # 2 symbols + escape
    cdfs[1, :6] = [0, 1, 2, 40000, 65535, 65536]       # freq-1 symbols
    cdfs[2, :8] = [0, 100, 5000, 20000, 45000, 60000, 65500, 65536]
    cdf_len = np.array([4, 6, 8], dtype=np.int32)
    offset = np.array([0, -2, -3], dtype=np.int32)
    rng = np.random.default_rng(5)
    n = 4000
    idx = rng.integers(0, 3, size=n).astype(np.int32)
    sym = rng.integers(-6, 7, size=n).astype(np.int32)
    sym[:12] = [0, 1, 2, 3, -1, -100, 100, 70000, -70000, 2 ** 24, -(2 ** 24), 2 ** 30]
    data = rans.encode_with_indexes(sym, idx, cdfs, cdf_len, offset)
    assert np.array_equal(rans.decode_with_indexes(data, idx, cdfs, cdf_len, offset), sym)
    assert data == rans.py_encode_with_indexes(sym, idx, cdfs, cdf_len, offset)
    np.savez_compressed(os.path.join(HERE, "coder_kat.npz"), cdfs=cdfs, cdf_len=cdf_len, offset=offset, idx=idx,
                        sym=sym, data=np.frombuffer(data, dtype=np.uint8))
    print("coder_kat bytes", len(data))


def make_pmf_kat():
    rng = np.random.default_rng(11)
    cases = []
    for n in (2, 3, 7, 23, 64, 200):
        p = rng.random(n).astype(np.float32) ** 6
        p /= p.sum()
        cases.append(p)
    p = np.zeros(40, dtype=np.float32)
    p[3] = 1.0  # 39 empty bins: exercises the steal loop heavily
    cases.append(p)
    p = np.full(300, 1e-7, dtype=np.float32)
    p[150] = 0.9
    cases.append(p)
    out = {}
    for i, p in enumerate(cases):
        out[f"pmf{i}"] = p
        out[f"cdf{i}"] = rans.pmf_to_quantized_cdf(p, 16)
        assert np.array_equal(out[f"cdf{i}"], rans.py_pmf_to_quantized_cdf(p, 16))
    np.savez_compressed(os.path.join(HERE, "pmf_kat.npz"), n=len(cases), **out)
    print("pmf_kat cases", len(cases))


if __name__ == "__main__":
    make_pmf_kat()
    make_coder_kat()
    make_model_case("factorized_c3_64", 3, 64, 0, "aid")
    make_model_case("factorized_c1_64", 1, 64, 1, "s2")
    make_model_case("factorized_c13_64", 13, 64, 2, "s2-merged")
    make_model_case("factorized_c3_64_signflip", 3, 64, 0, "aid", form="signflip")
