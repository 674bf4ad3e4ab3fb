"""N > 1 path on CPU: two processes over gloo run the same sharding + all-gather code the GPU
ranks run over RCCL.  The per-rank forward is the CPU oracle here (tests only)."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from f8net_amd import dist as f8dist
    from f8net_amd import synth, topology
    from oracle import oracle
    r, w, _ = f8dist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    spec = topology.get('resnet18', num_classes=10)
    params = synth.make_params(spec, seed=2)
    x, fl = synth.make_input(spec, params, n_total, 32, seed=5)
    lo, hi = f8dist.shard_bounds(n_total, world, rank)
    fwd = f8dist.ShardedForward(lambda t: torch.from_numpy(oracle.net_forward(spec, params, t.numpy(), fl)), 10)
    out = fwd(torch.from_numpy(x[lo:hi]), n_total=n_total)
    if n_total % world == 0:
        # pipelined variant (what bench.py runs): three batches through two buffer pairs, collectives asynchronous
        def local(t, o):
            o.copy_(torch.from_numpy(oracle.net_forward(spec, params, t.numpy(), fl)))
        pf = f8dist.PipelinedShardedForward(local, 10, hi - lo, torch.device('cpu'))
        outs = []
        for rep in range(3):
            xr = np.roll(x, rep, axis=0)
            outs.append((pf(torch.from_numpy(xr[lo:hi].copy())), xr))
        pf.finish()
        for o, xr in outs[1:]:          # the first pair has been reused by the third batch
            assert np.array_equal(o.numpy(), oracle.net_forward(spec, params, xr, fl))
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_total', [4, 5])
def test_sharded_forward_world2(n_total):
    from f8net_amd import synth, topology
    from oracle import oracle
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + n_total
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    spec = topology.get('resnet18', num_classes=10)
    params = synth.make_params(spec, seed=2)
    x, fl = synth.make_input(spec, params, n_total, 32, seed=5)
    np.testing.assert_array_equal(got, oracle.net_forward(spec, params, x, fl))


def test_shard_bounds():
    from f8net_amd.dist import shard_bounds
    for n in (0, 1, 7, 128, 2048):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, w, r) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1
