"""On-disk formats around the path.

Checkpoints: the dict LICOS writes - {"batch_idx", "state_dict", "loss", "local_time"} through torch.save
(/root/reference/licos/federation_utils.py:58-78, licos/utils.py:76-111) - and the way eval reads it back
(/root/reference/eval_script.py:68-72: torch.load -> load_state_dict(ckpt["state_dict"]) -> update()).  State-dict keys
are CompressAI's, so files move between the two implementations.

Bit streams: a length-prefixed container for compress() output, following the layout of CompressAI's
examples/codec.py (big-endian uint32 fields: shape, number of string lists, then per string its length and bytes),
extended with a leading batch count because LICOS codes batches of tiles.  CompressAI is absent here, so the
container is self-consistent rather than verified against that tool."""
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
