"""N>1 path on CPU: two and eight `gloo` ranks shard a batch, each fills its shard, rank 0 gathers the scores with the same
dpx_gpu_genomics_project_amd.shard helpers bench.py uses under RCCL.  The per-shard compute here is the CPU oracle
(this is a test of the sharding + collective plumbing; the GPU fill itself is covered by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_pairs, ragged, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import oracle_py as O
    from dpx_gpu_genomics_project_amd.shard import gather_scores, shard_range
    from dpx_gpu_genomics_project_amd.synth import make_batch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sb = make_batch(num_pairs, 40, 56, seed=77)  # every rank sees the same pair table, owns one shard of it
    lo, hi = shard_range(num_pairs, rank, world)
    local = torch.tensor([O.lsw(sb.ref(p), sb.qry(p), want_dir=False).score for p in range(lo, hi)], dtype=torch.int32)
    sizes = [shard_range(num_pairs, r, world)[1] - shard_range(num_pairs, r, world)[0] for r in range(world)]
    allscores = gather_scores(local, rank, world, sizes if ragged else None)
    if rank == 0:
        np.save(out_path, allscores.numpy())
    else:
        assert allscores is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,num_pairs,ragged", [(2, 24, False), (2, 25, True), (8, 64, False), (8, 61, True)])
def test_shard_and_gather_over_gloo(tmp_path, world, num_pairs, ragged):
    """World sizes 2 and 8 (the node the scaling run uses): even shards through dist.gather, ragged ceil(N/G) shards (the last rank short,
    as configs[4] would be on 7 GPUs) through the padded gather."""
    sys.path.insert(0, HERE)
    import oracle_py as O
    from dpx_gpu_genomics_project_amd.synth import make_batch

    out = str(tmp_path / "scores.npy")
    mp.spawn(_worker, args=(world, _free_port(), num_pairs, ragged, out), nprocs=world, join=True)
    got = np.load(out)
    sb = make_batch(num_pairs, 40, 56, seed=77)
    want = np.array([O.lsw(sb.ref(p), sb.qry(p), want_dir=False).score for p in range(num_pairs)], np.int32)
    assert np.array_equal(got, want)
