"""The N>1 path of bench.py on a one-GPU box, in fresh child processes launched exactly as the driver launches the
multi-GPU bench (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
bench.py --gpus N ...`), except that both ranks share cuda:0 and the collective runs over gloo (RCCL refuses two ranks on
one device): per-rank batches, zero-copy view of the engine's device scores, gather to rank 0, MAX all-reduce of the
times, one JSON line.  Also BASELINE.json's configs[4] shard (12 500 of 100 000 pairs) at its full size on one GPU."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_py as O
from dpx_gpu_genomics_project_amd.shard import shard_range
from dpx_gpu_genomics_project_amd.synth import make_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = (3, -1, -2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun_bench(tmp_path, world, extra):
    prefix = str(tmp_path / "scores")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--share-gpu",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--dump-scores", prefix] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-1000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE JSON line, the other ranks nothing
    return json.loads(lines[0]), prefix


def test_bench_two_ranks_weak_scaling_rehearsal(tmp_path):
    out, prefix = _torchrun_bench(tmp_path, 2, ["--pairs", "512"])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["metric"] == "GCUPS"
    assert out["config"]["pairs_per_gpu"] == 512 and out["value"] > 0 and out["roofline"]["frac"] > 0
    # round 3: the line says which backend / world ran and carries every rank's fill time, pool record and score checksum;
    # rank 0 has checked the gathered vector against EVERY rank's checksum
    assert out["config"]["backend"] == "gloo" and out["config"]["world_size"] == 2
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and all(r["gathered_ok"] and r["kernel_ms"] > 0 and r["pairs"] == 512 for r in out["ranks"])
    assert out["ranks"][0]["scores_checksum"] != out["ranks"][1]["scores_checksum"]
    assert out["roofline"]["kernel_ms"] == max(r["kernel_ms"] for r in out["ranks"])
    local = [np.load(f"{prefix}.rank{r}.npy") for r in range(2)]
    gathered = np.load(f"{prefix}.gathered.npy")
    assert np.array_equal(gathered, np.concatenate(local))       # rank order, every rank's own scores
    assert not np.array_equal(local[0], local[1])                 # the ranks really aligned different sub-batches
    for r in range(2):                                            # bench.py's per-rank batch: seed 1 + 1000 r, indices from r * 512
        sb = make_batch(512, 1024, 1024, seed=1 + 1000 * r, first_index=r * 512)
        for p in range(0, 512, 37):
            assert local[r][p] == O.lsw(sb.ref(p), sb.qry(p), *W, want_dir=False).score, (r, p)


def test_bench_three_ranks_strong_scaling_shards(tmp_path):
    """configs[4]'s code path (negative pair count = total over all ranks, shard_range per rank), small total."""
    out, prefix = _torchrun_bench(tmp_path, 3, ["--workload", "lsw_100k_1024_sharded", "--total-pairs", "600"])
    assert out["n_gpus"] == 3 and out["scaling"] == "strong" and out["config"]["pairs_per_gpu"] == 200
    assert out["config"]["world_size"] == 3 and len(out["ranks"]) == 3 and all(r["gathered_ok"] for r in out["ranks"])
    gathered = np.load(f"{prefix}.gathered.npy")
    assert len(gathered) == 600
    for r in range(3):
        lo, hi = shard_range(600, r, 3)
        assert np.array_equal(gathered[lo:hi], np.load(f"{prefix}.rank{r}.npy"))
        sb = make_batch(hi - lo, 1024, 1024, seed=5 + 1000 * r, first_index=lo)
        for p in (0, 96, 100, 199):
            assert gathered[lo + p] == O.lsw(sb.ref(p), sb.qry(p), *W, want_dir=False).score, (r, p)


@pytest.mark.parametrize("rank", [0, 5, 7])
def test_config4_one_rank_shard_of_lsw_100k_1024_at_full_size(gpu, rank):
    """Ranks 0, 5 and 7 of 8 of BASELINE.json's configs[4] (round 2: rank 5 only): shard_range(100000, rank, 8) = 12 500 pairs of
    1024 x 1024, 27.8 GB of H."""
    world = 8
    lo, hi = shard_range(100000, rank, world)
    assert (lo, hi) == (12500 * rank, 12500 * (rank + 1))
    sb = make_batch(hi - lo, 1024, 1024, seed=5 + 1000 * rank, first_index=lo)   # what bench.py builds on that rank
    with gpu.Batch(gpu.ALGO_LSW, sb.sequences, sb.pairs, *W) as b:
        d, info = b.describe(), b.info()
        assert d["kernel"] == "k_linear_fill_pk" and d["couples"] == 6250 and d["singles"] == 0
        assert info["cells"] == 12500 * 1024 * 1024 and info["matrix_bytes"] > 27.5e9
        b.fill()
        sc, er, ec = b.results()
        first = (sc.copy(), er.copy(), ec.copy())
        b.fill()                                                             # refill: same results
        sc, er, ec = b.results()
        assert all(np.array_equal(x, y) for x, y in zip(first, (sc, er, ec)))
        ident = [p for p in range(hi - lo) if (lo + p) % 101 == 100 and (lo + p) % 97 != 96]
        assert len(ident) > 100
        for p in ident:                                                      # query == reference
            assert (sc[p], er[p], ec[p]) == (3 * 1024, 1024, 1024)
        assert sc.max() == 3 * 1024 and sc.min() > 0
        for p in range(0, hi - lo, 25):                                      # 500 pairs: score + start cell vs the oracle
            o = O.lsw(sb.ref(p), sb.qry(p), *W, want_dir=False)
            assert (sc[p], er[p], ec[p]) == (o.score, o.end_row, o.end_col), p
        for p in (0, 6249, 12499):                                           # first / middle / last pair: every cell + printed lines
            o = O.lsw(sb.ref(p), sb.qry(p), *W)
            assert np.array_equal(b.matrix(p).astype(np.int32), o.H)
            assert b.traceback(p) == (("", "", "") if o.score == 0 else O.lsw_traceback(sb.ref(p), sb.qry(p), o))
    # a pair's result does not depend on the shard around it: a sample alone, on another kernel (one wave per stripe)
    sample = np.arange(3, hi - lo, 997)
    with gpu.Batch(gpu.ALGO_LSW, sb.sequences, sb.pairs[sample], *W) as small:
        assert small.describe()["kernel"].startswith("k_linear_split")
        small.fill()
        s1, r1, c1 = small.results()
    assert np.array_equal(s1, sc[sample]) and np.array_equal(r1, er[sample]) and np.array_equal(c1, ec[sample])
