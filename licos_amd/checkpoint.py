"""On-disk formats around the path.

Checkpoints: the dict LICOS writes - {"batch_idx", "state_dict", "loss", "local_time"} through torch.save
(/root/reference/licos/federation_utils.py:58-78, licos/utils.py:76-111) - and the way eval reads it back
(/root/reference/eval_script.py:68-72: torch.load -> load_state_dict(ckpt["state_dict"]) -> update()).  State-dict keys
are CompressAI's, so files move between the two implementations.

Bit streams, two containers:
* ``write_image`` / ``read_image``: ONE image per record, **codec.py-style**: the byte layout of the EARLY releases of
  CompressAI's ``examples/codec.py`` (``_encode_image`` / ``_decode_image``: two header bytes - model id, (metric << 4) |
  (quality - 1) -, the original size as two big-endian uint32, then ``write_body``: latent shape (2 x uint32), the
  number of string lists (uint32) and per list the length (uint32) and bytes of its string).  NOT claimed interchangeable
  with the tool: CompressAI is absent from this image, the layout is restated from its published source and has never
  met a file the tool wrote; later releases of codec.py (the ones whose zoo carries ``bmshj2018-factorized-relu``, i.e.
  the ``ids="current"`` table below) are believed to add a codec-type byte to the header and a bit-depth byte after the
  size, which this container does not write.  The two model-id tables exist because the zoo order - hence the id of
  every model after index 0 - differs between those releases; pick ``ids="legacy"`` for the layout written here.
* ``write_strings`` / ``read_strings``: a whole ``compress()`` result (a batch of tiles) in one stream: the same body
  fields with a leading magic and batch count - a licos_amd format, because codec.py has no notion of a batch."""
import os
import shutil
import struct

import torch


def save_checkpoint(state, is_best, filename="checkpoint.pth.tar"):
    """licos/utils.py:76-79."""
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, "checkpoint_best_loss.pth.tar")


def save_model_checkpoint_over_time(cfg, local_time, rank, state):
    """licos/utils.py:82-111: <save_path>/<model>_time_checkpoints/<model>_rank_<r>_sim_time=<t>.pth.tar."""
    save_path = cfg["save_path"] if isinstance(cfg, dict) else cfg.save_path
    model_name = save_path.split("/")[-1].split(".")[0]
    checkpoint_time_dir = os.path.join(save_path, model_name + "_time_checkpoints")
    model_name = model_name + "_rank_" + str(rank)
    os.makedirs(checkpoint_time_dir, exist_ok=True)
    save_checkpoint(state=state, is_best=False,
                    filename=checkpoint_time_dir + "/" + model_name + "_sim_time=" + str(local_time) + ".pth.tar")


def make_state(net, batch_idx, loss, local_time):
    """The dict federation_utils.py:69-78 saves; tensors moved to the host so the file loads anywhere."""
    return {"batch_idx": batch_idx, "state_dict": {k: v.detach().cpu() for k, v in net.state_dict().items()},
            "loss": loss, "local_time": local_time}


def load_checkpoint(path, net, device=None, update=True):
    """eval_script.py:68-72.  Returns the checkpoint dict (batch_idx / loss / local_time stay available)."""
    checkpoint = torch.load(path, map_location=device if device is not None else "cpu", weights_only=False)
    net.load_state_dict(checkpoint["state_dict"])
    if update:
        net.update(force=True)
    return checkpoint


# ------------------------------------------------------------------------------------------- bit-stream container
MAGIC = b"LICS"


def _w(fd, fmt, *v):
    fd.write(struct.pack(fmt, *v))


def write_strings(fd, out):
    """out = {"strings": [[bytes]*B, ...], "shape": (h, w)} as compress() returns it."""
    strings, shape = out["strings"], out["shape"]
    nb = len(strings[0]) if strings else 0
    fd.write(MAGIC)
    _w(fd, ">4I", int(shape[0]), int(shape[1]), len(strings), nb)
    for lst in strings:
        if len(lst) != nb:
            raise ValueError("every string list must hold one entry per image")
        for s in lst:
            _w(fd, ">I", len(s))
            fd.write(bytes(s))


def _read_exact(fd, n):
    data = fd.read(n)
    if len(data) != n:
        raise ValueError("truncated stream container")
    return data


def read_strings(fd):
    if fd.read(4) != MAGIC:
        raise ValueError("not a licos_amd stream container")
    h, w, nl, nb = struct.unpack(">4I", _read_exact(fd, 16))
    strings = []
    for _ in range(nl):
        lst = []
        for _ in range(nb):
            (n,) = struct.unpack(">I", _read_exact(fd, 4))
            lst.append(_read_exact(fd, n))
        strings.append(lst)
    return {"strings": strings, "shape": (h, w)}


# ---- one image per record, CompressAI examples/codec.py layout ---------------------------------------------------
MODEL_IDS = {  # compressai.zoo.image_models order; "legacy" = releases without the -relu variant (<= 1.2.0)
    "legacy": {"bmshj2018-factorized": 0, "bmshj2018-hyperprior": 1, "mbt2018-mean": 2, "mbt2018": 3,
               "cheng2020-anchor": 4, "cheng2020-attn": 5},
    "current": {"bmshj2018-factorized": 0, "bmshj2018-factorized-relu": 1, "bmshj2018-hyperprior": 2, "mbt2018-mean": 3,
                "mbt2018": 4, "cheng2020-anchor": 5, "cheng2020-attn": 6},
}
METRIC_IDS = {"mse": 0, "ms-ssim": 1}


def write_image(fd, out, index, model, quality, original_size, metric="mse", ids="current"):
    """Writes image `index` of a compress() result as one codec.py record; returns the number of bytes written."""
    table = MODEL_IDS[ids]
    if model not in table:
        raise ValueError(f"{model!r} has no id in CompressAI's {ids} model table")
    if not 1 <= int(quality) <= 16:
        raise ValueError("quality must be 1..16")
    strings, shape = out["strings"], out["shape"]
    start = fd.tell()
    fd.write(struct.pack(">2B", table[model], (METRIC_IDS[metric] << 4) | ((int(quality) - 1) & 0x0F)))
    _w(fd, ">2I", int(original_size[0]), int(original_size[1]))
    _w(fd, ">3I", int(shape[0]), int(shape[1]), len(strings))
    for lst in strings:
        s = bytes(lst[index])
        _w(fd, ">I", len(s))
        fd.write(s)
    return fd.tell() - start


def read_image(fd, ids="current"):
    """Reads one codec.py record: (model, metric, quality, original_size, {"strings": [[s], ...], "shape"})."""
    model_id, code = struct.unpack(">2B", _read_exact(fd, 2))
    names = {v: k for k, v in MODEL_IDS[ids].items()}
    if model_id not in names:
        raise ValueError(f"unknown model id {model_id}")
    metric = {v: k for k, v in METRIC_IDS.items()}.get(code >> 4)
    if metric is None:
        raise ValueError(f"unknown metric id {code >> 4}")
    h, w = struct.unpack(">2I", _read_exact(fd, 8))
    s0, s1, nl = struct.unpack(">3I", _read_exact(fd, 12))
    strings = []
    for _ in range(nl):
        (n,) = struct.unpack(">I", _read_exact(fd, 4))
        strings.append([_read_exact(fd, n)])
    return names[model_id], metric, (code & 0x0F) + 1, (h, w), {"strings": strings, "shape": (s0, s1)}
