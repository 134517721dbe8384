"""The scale-conditioned rANS fast path (licos_gc_encode_prepare / licos_rans_encode_records / licos_gc_decode_prepare /
licos_rans_decode_image) against the oracle's coder ([CAI] rans_interface.cpp encode_with_indexes / decode_with_indexes
restated in oracle/rans_oracle.c): identical bytes, identical symbols - on every table row incl. the widest, on values
beyond a row's range (escapes), for ragged stream lengths and batch sizes on both sides of a wave / workgroup."""
import numpy as np
import pytest
import torch

from licos_amd import ops
from licos_amd.entropy_models import EntropyBottleneck, GaussianConditional
from licos_amd.models import get_scale_table
from oracle import model as om
from oracle import rans

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gc():
    gc = GaussianConditional(None).to(DEV)
    gc.update_scale_table(get_scale_table())
    return gc


def _oracle_tables():
    sd = {"gaussian_conditional.lower_bound_scale.bound": torch.tensor([0.11])}
    om.gc_update(sd)
    p = "gaussian_conditional."
    return sd, sd[p + "_quantized_cdf"].numpy(), sd[p + "_cdf_length"].numpy(), sd[p + "_offset"].numpy()


def _case(batch, n, seed, row_lo, row_hi, outliers):
    """Latents y ~ N(0, s) with s log-uniform over table rows [row_lo, row_hi]; a few values far outside the row."""
    g = torch.Generator().manual_seed(seed)
    table = get_scale_table()
    rows = torch.randint(row_lo, row_hi + 1, (batch, n), generator=g)
    scales = table[rows] * (0.9 + 0.2 * torch.rand(batch, n, generator=g))
    y = torch.randn(batch, n, generator=g) * scales
    if outliers:
        pos = torch.randint(0, n, (batch, outliers), generator=g)
        for b in range(batch):
            y[b, pos[b]] = (torch.randn(outliers, generator=g) * 40.0 * (1 + scales[b, pos[b]])).round() + 0.25
    return y.reshape(batch, 1, 1, n).contiguous(), scales.reshape(batch, 1, 1, n).contiguous()


def _encode(gc, y, scales):
    cdf, cdf_len, offset, table = gc.coder_tables()
    n = y[0].numel()
    rec, aux = ops.gc_encode_prepare(y, scales, gc.scale_table, gc.lower_bound_scale.bound_value, table, cdf_len, offset, cdf.shape[1])
    cap = n // 2 + 64
    for _ in range(2):
        words, nwords, status = ops.rans_encode_records(rec, aux, cap)
        host = torch.cat((nwords, status)).cpu().numpy()
        if host[-1] == 0:
            break
        cap = 2 * n + 8
    assert host[-1] == 0
    b = y.shape[0]
    off = np.zeros(b + 1, dtype=np.int64)
    np.cumsum(host[:b].astype(np.int64) * 4, out=off[1:])
    packed = ops.rans_compact(words, nwords, torch.from_numpy(off).to(y.device), int(off[-1])).cpu().numpy()
    return [packed[off[i]:off[i + 1]].tobytes() for i in range(b)]


def _decode(gc, strings, scales):
    image_dev, image_host = gc.coder_image()
    b = len(strings)
    n = scales[0].numel()
    idx16 = ops.gc_decode_prepare(scales, gc.scale_table, gc.lower_bound_scale.bound_value)
    data, off = EntropyBottleneck.pack_strings(strings, torch.device(DEV))
    sym = torch.full((n, b), -12345, device=DEV, dtype=torch.int32)
    status = ops.rans_decode_image(data, off, idx16, n, image_dev, image_host, sym, 1, b, b)
    assert int(status.item()) == 0
    return sym.t().contiguous().cpu()


@pytest.mark.parametrize("batch,n,rows,outliers", [
    (3, 1000, (0, 63), 0),        # every row incl. the ~3100-symbol ones; ragged length (1000 = 62 blocks + 8)
    (64, 512, (0, 30), 4),        # a full wave, escapes
    (65, 256, (10, 50), 2),       # a second wave with one live lane
    (130, 333, (0, 63), 3),       # two workgroups, ragged everything
    (2, 16384, (0, 24), 0),       # long streams on narrow rows: the ring refill path
    (1, 7, (40, 63), 1),          # shorter than one block
])
def test_gc_fast_path_matches_oracle_coder(batch, n, rows, outliers):
    gc = _gc()
    sd, cdf, cdf_len, offset = _oracle_tables()
    assert np.array_equal(gc._quantized_cdf.cpu().numpy(), cdf)
    y, scales = _case(batch, n, 11 * batch + n, rows[0], rows[1], outliers)
    ref_idx = om.gc_build_indexes(scales, sd).reshape(batch, n).numpy()
    ref_sym = torch.round(y).int().reshape(batch, n).numpy()
    strings = _encode(gc, y.to(DEV), scales.to(DEV))
    for i in range(batch):
        want = rans.encode_with_indexes(ref_sym[i], ref_idx[i], cdf, cdf_len, offset)
        assert strings[i] == want, f"stream {i}: {len(strings[i])} vs {len(want)} bytes"
    dec = _decode(gc, strings, scales.to(DEV))
    assert np.array_equal(dec.numpy(), ref_sym)
    if outliers:
        mx = cdf_len[ref_idx] - 2
        v = ref_sym - offset[ref_idx]
        assert int(((v < 0) | (v >= mx)).sum()) >= batch  # the escape path really ran


def test_gc_decoder_reports_truncated_streams():
    gc = _gc()
    y, scales = _case(4, 2048, 3, 10, 40, 0)
    strings = _encode(gc, y.to(DEV), scales.to(DEV))
    strings[2] = strings[2][: max(8, (len(strings[2]) // 8) * 4)]
    image_dev, image_host = gc.coder_image()
    idx16 = ops.gc_decode_prepare(scales.to(DEV), gc.scale_table, gc.lower_bound_scale.bound_value)
    data, off = EntropyBottleneck.pack_strings(strings, torch.device(DEV))
    sym = torch.zeros((2048, 4), device=DEV, dtype=torch.int32)
    status = ops.rans_decode_image(data, off, idx16, 2048, image_dev, image_host, sym, 1, 4, 4)
    assert int(status.item()) != 0
    ref_sym = torch.round(y).int().reshape(4, 2048)
    assert torch.equal(sym.t().cpu()[[0, 1, 3]], ref_sym[[0, 1, 3]])  # the intact streams are unaffected
