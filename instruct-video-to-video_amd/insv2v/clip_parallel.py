"""Clip-parallel sharding.  The reference inference path is single-process; its
work units -- (video, cfg, size) x prompt, insv2v_run_loveu_tgve.py:83,101 -- are independent, so
unit i runs on rank i % world with a full model replica and the only exchange is ONE all_gather of
the edited frames (RCCL over xGMI on GPUs; gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_units(n_units, rank, world):
    """Indices of the units owned by ``rank`` (round robin)."""
    return list(range(rank, n_units, world))


def units_per_rank(n_units, world):
    return (n_units + world - 1) // world


def gather_frames(local, n_units, rank=None, world=None, item_shape=None, dtype=torch.float16):
    """local: [k, ...] edited frames of this rank's units (k = len(shard_units)); returns
    [n_units, ...] in unit order on every rank.  Ranks with fewer units are padded so the single
    all_gather has equal shapes.  A rank that owns NO unit (n_units < world) cannot know the trailing shape from its
    own data: it passes ``local=None`` (or any [0, ...] tensor) together with ``item_shape`` = shape of one unit
    (and ``dtype`` when ``local`` is None)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    k = units_per_rank(n_units, world)
    if item_shape is not None and (local is None or tuple(local.shape[1:]) != tuple(item_shape)):
        if local is not None and local.shape[0] != 0:
            raise ValueError(f"gather_frames: units of shape {tuple(local.shape[1:])} but item_shape={tuple(item_shape)}")
        raise_dtype = local.dtype if local is not None else dtype
        dev = local.device if local is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
        local = torch.zeros((0, *item_shape), dtype=raise_dtype, device=dev)
    if local.shape[0] < k:
        pad = torch.zeros((k - local.shape[0], *local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    out = torch.empty((world * k, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    # rank r, slot j holds unit j*world + r
    out = out.reshape(world, k, *local.shape[1:]).transpose(0, 1).reshape(world * k, *local.shape[1:])
    return out[:n_units]
