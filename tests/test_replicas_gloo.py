"""world_size-2 CPU test (gloo) of the multi-GPU plumbing: replicas only, one arena broadcast,
no steady-state collective."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import _pkg
    m = _pkg.load()
    from biogpt_cpp_amd import replicas
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist = replicas.init_process_group("gloo")
    hp = replicas.broadcast_hparams([42384, 24, 16, 1024, 4096, 1024, 2, 40000] if rank == 0 else None)
    n = 3 * 1000 * 1000 + 17
    arena = torch.zeros(n, dtype=torch.uint8)
    if rank == 0:
        arena = torch.from_numpy((np.arange(n) * 2654435761 % 251).astype(np.uint8))
    replicas.broadcast_arena(arena, src=0, chunk_bytes=1 << 20)
    units = replicas.shard_units(8, rank, world)
    ids = np.array([rank * 100 + u for u in units], dtype=np.int32)
    allids = replicas.gather_ids(ids, 8)
    tmax = replicas.max_over_ranks(1.0 + rank)
    tot = replicas.sum_over_ranks(len(units) * 200)
    rep = replicas.replica_report(0.5 * (rank + 1), 400, "fake-device-%d" % rank, 0.0125 if rank == 0 else None, 1000)
    q.put((rank, hp, int(arena.to(torch.int64).sum()), units, allids.tolist(), tmax, tot, rep))
    dist.destroy_process_group()


def test_two_rank_replica_plumbing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, hp0, s0, u0, ids0, t0, tot0, rep0), (r1, hp1, s1, u1, ids1, t1, tot1, rep1) = res
    assert hp0 == hp1 == [42384, 24, 16, 1024, 4096, 1024, 2, 40000]
    assert s0 == s1 and s0 > 0                       # identical arena bytes on both ranks
    assert u0 == [0, 2, 4, 6] and u1 == [1, 3, 5, 7]   # disjoint, complete cover of the 8 prompts
    assert ids0 == ids1 and ids0[0][:4] == [0, 2, 4, 6] and ids0[1][:4] == [101, 103, 105, 107]
    assert t0 == t1 == 2.0 and tot0 == tot1 == 1600.0
    # the fields that make the first N > 1 bench record self-explanatory (bench.py puts rank 0's dict into its JSON line as "replicas")
    assert rep0["ranks_seen"] == 2 and rep0["backend"] == "gloo" and rep0["devices"] == ["fake-device-0", "fake-device-1"] == rep1["devices"]
    assert rep0["per_rank_tokens_per_s"] == [800.0, 400.0] == rep1["per_rank_tokens_per_s"]
    assert rep0["broadcast_ms"] == 12.5 and rep0["arena_bytes"] == 1000 and rep1["broadcast_ms"] is None


def test_bench_self_launch_command_line(tmp_path):
    """`python bench.py --gpus N` without a launcher must start its own N ranks (the driver's command is exactly that): the command it turns
    itself into, without executing it (no GPU here)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BIOGPT_BENCH_SELF_LAUNCH"] = "print"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["self_launch"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    # under a launcher (WORLD_SIZE set) nothing is re-executed; a mismatch between --gpus and the launcher's world size is an error, not a silent 1-GPU run
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    env2.pop("BIOGPT_BENCH_SELF_LAUNCH")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "nproc-per-node must equal --gpus" in (r2.stderr + r2.stdout)
