"""Multi-GPU plumbing for the decode path: REPLICAS ONLY.

Independent prompts are sharded across ranks (one process per GPU, torch.distributed; backend
"nccl" is RCCL on ROCm).  The only collective on the data path is ONE broadcast of the packed weight
arena from rank 0 over xGMI at start-up; steady state has no collectives (token ids are gathered on
the host at the end).  The same functions run under the gloo backend on CPU tensors in the tests.
"""
import os

import numpy as np


def shard_units(n_units, rank, world):
    """Prompt indices served by `rank`: g, g + world, g + 2*world, ... (SURVEY 8e 'prompts g, g+8, ...')."""
    return list(range(rank, n_units, world))


def init_process_group(backend):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    # RCCL shares device buffers between the ranks of one node through IPC handles; this host driver only supports the
    # dmabuf form (without the switch: `hipIpcGetMemHandle: invalid argument` at the first collective).  The HSA runtime reads
    # it when it initialises, i.e. at the process's first HIP call -- bench.py therefore also sets it before importing torch.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", "0")),
                            world_size=int(os.environ.get("WORLD_SIZE", "1")))
    return dist


def broadcast_arena(arena_tensor, src=0, chunk_bytes=256 << 20):
    """Broadcast the packed weight arena (a flat uint8 tensor) from `src` to every rank.
    Chunked so that one message stays well inside RCCL's comfortable sizes; with a tree/flat
    broadcast each peer receives over its own xGMI link."""
    import torch.distributed as dist
    n = arena_tensor.numel()
    for off in range(0, n, chunk_bytes):
        dist.broadcast(arena_tensor[off:min(n, off + chunk_bytes)], src=src)
    return arena_tensor


def broadcast_hparams(hp_list, src=0):
    """Broadcast the 8 header ints so that non-root ranks can lay out their arena without the file."""
    import torch
    import torch.distributed as dist
    backend = dist.get_backend()
    dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor(hp_list if hp_list is not None else [0] * 8, dtype=torch.int32, device=dev)
    dist.broadcast(t, src=src)
    return [int(v) for v in t.cpu().tolist()]


def max_over_ranks(seconds):
    """The bench contract's time = max over ranks of the timed region."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value):
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(value):
    """One float per rank, on every rank (per-rank rates in the bench line)."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(o.item()) for o in out]


def gather_strings(text):
    """One short string per rank, on every rank (device names in the bench line)."""
    import torch.distributed as dist
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, str(text))
    return [str(o) for o in out]


def replica_report(elapsed_s, tokens_this_rank, device_name, broadcast_s, arena_bytes):
    """What makes the first N > 1 bench record self-explanatory: how many ranks the communicator saw, which device each bound, each rank's own rate,
    the broadcast's time and size.  Every rank calls it (two all-gathers); the dict is the same on all of them."""
    import torch.distributed as dist
    rates = gather_floats(tokens_this_rank / max(elapsed_s, 1e-12))
    return {"ranks_seen": int(dist.get_world_size()), "backend": str(dist.get_backend()), "devices": gather_strings(device_name),
            "per_rank_tokens_per_s": [round(r, 2) for r in rates],
            "broadcast_ms": None if broadcast_s is None else round(broadcast_s * 1e3, 3), "arena_bytes": None if arena_bytes is None else int(arena_bytes)}


def gather_ids(ids, max_len):
    """All ranks' generated ids on every rank: returns a [world, max_len] int32 array (padded with -1)."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    buf = np.full(max_len, -1, dtype=np.int32)
    buf[:len(ids)] = ids
    mine = torch.from_numpy(buf).to(dev)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return np.stack([o.cpu().numpy() for o in out])
