"""Federated weight averaging over RCCL (xGMI), replacing the reference's file-based blend.

Reference semantics (/root/reference/licos/federation_utils.py:9-85): when a rank
"visits" the ground station it loads the central checkpoint, blends EVERY state_dict key as
``w_l * local + w_c * central`` with ``w_l = best/(best+loss)``, ``w_c = loss/(best+loss)``
(:47-53), overwrites the central file (:69-78) and adopts the blend (:85).  Visits are sequential
and guarded by a lock file; no collective carries weights.

Here the 8 satellites are the 8 GPUs of a node and reach the averaging point together, so the
sequence of pair-wise blends for visit order 0..N-1 collapses to ONE convex combination
``sum_r a_r * theta_r`` with ``a_0 = prod_{k>0} w_c,k`` and ``a_r = w_l,r * prod_{k>r} w_c,k``
(`reference_coefficients`), computed by a single all-reduce of one flat fp32 bucket that holds the
whole floating state (about 3.0 M elements, 12 MB) plus one element carrying the coefficient sum.
Uniform coefficients give plain FedAvg.  Every rank ends with the central model (the reference
leaves rank r with the central model *as of its visit*; documented difference).

Integer buffers (the entropy coder's tables) are not averaged: blending them is the identity
in exact arithmetic and a truncation hazard in floating point; ``update_central_model`` rebuilds
them from the averaged parameters (``net.update(force=True)``) when the model carries tables.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib, ops


class FlatState:
    """Re-homes every floating tensor of ``net.state_dict()`` into one contiguous fp32 bucket
    (parameters and buffers become views of it), so averaging is a collective on one tensor and
    the "load" afterwards is free.  One extra trailing element carries the coefficient."""

    def __init__(self, net):
        tensors = [(k, v) for k, v in net.state_dict(keep_vars=True).items() if v.dtype == torch.float32]
        if not tensors:
            raise ValueError("no floating state to average")
        dev = tensors[0][1].device
        total = sum(v.numel() for _, v in tensors)
        # the allocation is padded so that it splits into equal chunks for ANY world up to 64 (the direct schedule ships
        # one chunk of ceil(n / world) elements per peer: at most n + world - 1 in all - a multiple of 64 alone is enough
        # for power-of-two worlds only); `flat` is the live part: state + the coefficient element
        self.alloc = torch.zeros(-(-(total + 1) // 64) * 64 + 64, device=dev, dtype=torch.float32)
        self.flat = self.alloc[: total + 1]
        self.keys = []
        off = 0
        with torch.no_grad():
            for k, v in tensors:
                n = v.numel()
                view = self.flat[off:off + n].view(v.shape)
                view.copy_(v.detach())
                v.data = view  # parameter / buffer now lives inside the bucket
                self.keys.append((k, off, n))
                off += n
            self.flat[-1] = 0.0
        self.numel = total

    def state(self):
        return self.flat[: self.numel]


def reference_coefficients(losses, best_losses):
    """Closed form of the reference's sequential visits r = 0..N-1 (federation_utils.py:47-53)."""
    n = len(losses)
    wl = [b / (b + l) for b, l in zip(best_losses, losses)]
    wc = [l / (b + l) for b, l in zip(best_losses, losses)]
    coef = []
    for r in range(n):
        a = 1.0 if r == 0 else wl[r]
        for k in range(r + 1, n):
            a *= wc[k]
        coef.append(a)
    return coef


class NativeComm:
    """An RCCL communicator owned by liblicos_hip.so (licos_comm_init): the collective of the federated average then
    runs as ONE C-ABI call on the caller's stream (licos_allreduce_weighted: scale, all-reduce over xGMI, normalise;
    licos_allreduce_weighted_direct: the two-step point-to-point schedule) instead of kernels around torch.distributed's
    collectives.  The 128-byte id travels through the existing process group (any backend).  Every rank of the group
    must construct it, with its device current.  A context manager: the communicator is destroyed on exit, also when the
    body raises (a rank that dies inside leaves its peers in the collective - the launcher's job to end them)."""

    def __init__(self, group=None):
        lib = _lib.load()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(lib.licos_comm_unique_id(ctypes.cast(ident, ctypes.c_void_p)), "comm_unique_id")
        box = [ident.raw if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = ctypes.create_string_buffer(box[0], 128)
        self._comm = ctypes.c_void_p()
        _lib.check(lib.licos_comm_init(ctypes.byref(self._comm), world, rank, ctypes.cast(ident, ctypes.c_void_p)), "comm_init")
        self.world, self.rank = world, rank
        self._scratch = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def allreduce_weighted_direct_(self, flat_state, coef):
        """The direct schedule on the FlatState's padded allocation."""
        alloc, n = flat_state.alloc, flat_state.flat.numel()
        ops._dev(alloc)
        chunk = -(-n // self.world)
        if chunk * self.world > alloc.numel():
            raise ValueError(f"licos_amd: the direct schedule needs the bucket padded to {chunk * self.world} elements "
                             f"({self.world} ranks); FlatState pads for power-of-two worlds up to 64")
        if self._scratch is None or self._scratch.numel() < chunk * self.world or self._scratch.device != alloc.device:
            self._scratch = torch.empty(chunk * self.world, device=alloc.device, dtype=torch.float32)
        rc = _lib.load().licos_allreduce_weighted_direct(ops._p(alloc), n, alloc.numel(), float(coef), self._comm, self.world,
                                                         self.rank, ops._p(self._scratch), ops._stream())
        _lib.check(rc, "allreduce_weighted_direct")
        ops.touch_weights()
        return flat_state

    def allreduce_weighted_(self, flat, coef):
        ops._dev(flat)
        rc = _lib.load().licos_allreduce_weighted(ops._p(ops._f32(flat)), flat.numel(), float(coef), self._comm, ops._stream())
        _lib.check(rc, "allreduce_weighted")
        ops.touch_weights()
        return flat

    def close(self):
        if self._comm:
            _lib.check(_lib.load().licos_comm_destroy(self._comm), "comm_destroy")
            self._comm = ctypes.c_void_p()


def _direct_allreduce_(flat_state, group=None):
    """SUM over the group on the direct schedule, through torch.distributed point-to-point operations (every backend
    has them: RCCL for device tensors, gloo for the CPU rehearsal): rank r collects chunk r from every peer, adds them
    in a fixed order, and hands the reduced chunk back to every peer - two exchange steps whatever the world size."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = flat_state.flat.numel()
    chunk = -(-n // world)
    alloc = flat_state.alloc
    if chunk * world > alloc.numel():
        raise ValueError(f"licos_amd: the direct schedule needs the bucket padded to {chunk * world} elements ({world} ranks)")
    peers = [p for p in range(world) if p != rank]
    gr = (lambda p: dist.get_global_rank(group, p)) if group is not None else (lambda p: p)
    scratch = torch.empty((world, chunk), device=alloc.device, dtype=torch.float32)
    view = alloc[: chunk * world].view(world, chunk)
    ops1 = []
    for p in peers:
        ops1.append(dist.P2POp(dist.isend, view[p], gr(p), group))
        ops1.append(dist.P2POp(dist.irecv, scratch[p], gr(p), group))
    for w in dist.batch_isend_irecv(ops1):
        w.wait()
    for p in peers:  # ascending: the same order on every rank's own chunk
        view[rank] += scratch[p]
    ops2 = []
    for p in peers:
        ops2.append(dist.P2POp(dist.isend, view[rank], gr(p), group))
        ops2.append(dist.P2POp(dist.irecv, view[p], gr(p), group))
    for w in dist.batch_isend_irecv(ops2):
        w.wait()
    return flat_state


SCHEDULES = ("ring", "direct")


def weighted_average_(flat_state, coef, group=None, native=None, schedule="ring"):
    """In place: bucket <- sum_r coef_r * bucket_r / sum_r coef_r over the process group.  `schedule`: "ring" = one
    all-reduce (RCCL picks its algorithm), "direct" = the two-step point-to-point schedule of SURVEY 5.8 (xGMI is a
    mesh of direct links).  `native`: a NativeComm - the whole step as one C-ABI call of liblicos_hip.so."""
    if schedule not in SCHEDULES:
        raise ValueError(f"schedule must be one of {SCHEDULES}")
    flat = flat_state.flat
    if native is not None:
        with torch.no_grad():
            if schedule == "direct":
                native.allreduce_weighted_direct_(flat_state, coef)
            else:
                native.allreduce_weighted_(flat, coef)
        return flat_state
    with torch.no_grad():
        flat[-1] = 1.0
        if flat.is_cuda:
            ops.scale_f32(flat, coef)          # HIP kernel; last element becomes coef
        else:
            flat.mul_(coef)                    # gloo / CPU: only the collective logic is exercised here
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if schedule == "direct":
                _direct_allreduce_(flat_state, group)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)  # "nccl" == RCCL over xGMI on ROCm
        if flat.is_cuda:
            ops.scale_f32(flat[: flat_state.numel], 1.0, inv_alpha_dev=flat[flat_state.numel:])
        else:
            flat[: flat_state.numel].div_(flat[-1])
    # parameters are views of the bucket and neither the kernels nor the collective move their version counters:
    # tell every packed-operand cache (engine.py, layers.py, entropy_models.py) that the weights changed
    ops.touch_weights()
    return flat_state


def _gather_scalars(values, world, rank, device, group):
    """[world][len(values)] float64 on the host.  One all-reduce of a one-hot-by-rank matrix: works on every backend
    (gloo has no all_gather for device tensors, RCCL none for host tensors)."""
    t = torch.zeros((world, len(values)), dtype=torch.float64, device=device)
    t[rank] = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu()


def update_central_model(rank, device, batch_idx, net, loss, best_loss, local_time, cfg=None, flat_state=None,
                         group=None, uniform=False, native=None, schedule="ring"):
    """Collective counterpart of ``federation_utils.update_central_model`` (same leading arguments).
    Every rank of the group must call it.  Returns the FlatState (reuse it on the next call)."""
    if flat_state is None:
        flat_state = FlatState(net)
    loss = float(loss)
    best_loss = float(best_loss)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if uniform or world == 1:
        coef = 1.0
    else:
        r = dist.get_rank(group)
        vals = _gather_scalars([loss, best_loss], world, r, flat_state.flat.device, group)
        coef = reference_coefficients(vals[:, 0].tolist(), vals[:, 1].tolist())[r]
    weighted_average_(flat_state, coef, group=group, native=native, schedule=schedule)
    # the integer coder tables are functions of the (now averaged) parameters: rebuild them when the model carries any,
    # so that neither the module nor the checkpoint written below pairs new weights with old tables
    if any(getattr(m, "_offset", None) is not None and m._offset.numel() > 0 for m in net.modules()):
        net.update(force=True)
    # the reference leaves the averaged model on disk as the "central model" (federation_utils.py:58-83); with a
    # collective every rank holds it already, so one rank writes the same file for eval_script.py to pick up
    save_path = _cfg_get(cfg, "save_path")
    if save_path:
        from . import checkpoint
        r = dist.get_rank(group) if dist.is_initialized() else 0
        state = None
        if _cfg_get(cfg, "save_checkpoints_over_time"):
            state = checkpoint.make_state(net, batch_idx, loss, local_time)
            checkpoint.save_model_checkpoint_over_time(cfg, local_time, rank, state)
        if r == 0:
            checkpoint.save_checkpoint(state or checkpoint.make_state(net, batch_idx, loss, local_time), False,
                                       filename=save_path + ".pth.tar")
    return flat_state


def _cfg_get(cfg, key):
    if cfg is None:
        return None
    if isinstance(cfg, dict):
        return cfg.get(key)
    return getattr(cfg, key, None)


def clock_sync(running=1, group=None, device=None):
    """The reference's ``comm.allreduce(1, op=MPI.SUM)`` + ``Barrier()`` liveness sync
    (licos/main.py:96-107, :259-263): number of ranks still running."""
    if not (dist.is_available() and dist.is_initialized()):
        return running
    if device is None and dist.get_backend(group) == "nccl":  # RCCL reduces device memory only
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([running], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def shard_range(total, rank, world):
    """Contiguous partition of `total` independent tiles over `world` ranks (no collective)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
