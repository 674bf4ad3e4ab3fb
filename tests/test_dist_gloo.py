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


def _discipline_worker(rank, world, port, lagged, depth, q):
    """PipelinedShardedForward under ASYNCHRONOUS collectives whose completion is deliberately late: every all-gather is a real
    gloo async op wrapped in a handle that (a) marks its buffer pair busy from launch until wait() returns and (b) sleeps before it
    completes.  The per-rank forward checks, at SUBMISSION time, what f8_net_set_pipelined's contract demands of the caller:
    the pair it writes is free, and — lagged schedules — so is the pair the NEXT run will write (one call of slack)."""
    sys.path.insert(0, ROOT)
    import time
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from f8net_amd import dist as f8dist
    f8dist.init_from_env(backend='gloo')
    classes, n_local, steps = 7, 3, 9
    busy, violations, launched = {}, [], []

    class DelayedWork:
        def __init__(self, work, key, delay):
            self.work, self.key, self.delay = work, key, delay
            busy[key] = True

        def wait(self):
            time.sleep(self.delay)                      # the collective "is still running" while the caller moves on
            self.work.wait()
            busy[self.key] = False

    real = dist.all_gather_into_tensor

    def slow_all_gather(out, inp, group=None, async_op=False):
        assert async_op
        launched.append(inp.data_ptr())
        return DelayedWork(real(out, inp, group=group, async_op=True), inp.data_ptr(), 0.02 * (1 + (len(launched) + rank) % 3))

    f8dist.dist.all_gather_into_tensor = slow_all_gather
    pf = None

    def local(x, out):
        k = [t.data_ptr() for t in pf.local].index(out.data_ptr())
        if busy.get(out.data_ptr()):
            violations.append(('written while its collective is in flight', pf.i, k))
        if lagged and busy.get(pf.local[(k + 1) % pf.depth].data_ptr()):
            violations.append(('next run\'s pair not free one call ahead', pf.i, k))
        out.copy_(x[:, :classes] * (rank + 1))

    pf = f8dist.PipelinedShardedForward(local, classes, n_local, torch.device('cpu'), lagged=lagged, depth=depth)
    outs = []
    for i in range(steps):
        x = torch.full((n_local, classes), float(i + 1))
        outs.append((pf(x), i))
    pf.finish()
    assert not any(busy.values()) and len(launched) == steps
    for o, i in outs[-depth:]:                           # earlier pairs have been reused
        want = torch.cat([torch.full((n_local, classes), float((i + 1) * (r + 1))) for r in range(world)])
        assert torch.equal(o, want), (i, o)
    if rank == 0:
        q.put(violations)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('lagged,depth', [(True, 2), (True, 3), (False, 2)])
def test_pipelined_buffer_discipline_with_late_async_collectives(lagged, depth):
    """bench.py's loop (PipelinedShardedForward, lagged under f8_net_set_pipelined, depth = pipeline_depth) against collectives
    that complete late: no buffer pair is rewritten, or promised to the next run, while its all-gather is still in flight
    (proven on gloo before RCCL first meets it)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29650 + depth + (10 if lagged else 0)
    procs = [ctx.Process(target=_discipline_worker, args=(r, 2, port, lagged, depth, q)) for r in range(2)]
    for p in procs:
        p.start()
    violations = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert violations == []


def test_bench_dry_run_dist_world2():
    """`bench.py --dry-run-dist`: the launch the driver issues for N GPUs (torch.distributed.run, one rank per GPU) over gloo on CPU with a stub
    forward — rendezvous on 127.0.0.1, --gpus == WORLD_SIZE check, per-rank inputs, the pipelined all-gather, barrier + fence on both sides of
    the timed region, max over ranks, ONE JSON line from rank 0 — so that the first 8-GPU launch cannot fail on plumbing."""
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', '29671',
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--bs', '2', '--hw', '32', '--arch', 'resnet18', '--dry-run-dist']
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['higher_is_better'] is True
    assert d['config']['global_batch'] == 4 and d['config']['parallelism'].startswith('dp2') and 'dry_run' in d and d['value'] > 0
    # the launch verifies itself: ranks the backend connected, each rank's own rate, every rank's gathered blocks vs the owners' checksums
    mg = d['multi_gpu']
    assert mg['rccl_ranks'] == 2 and mg['rank_ids'] == [0, 1] and mg['backend'] == 'gloo' and mg['logits_gathered_ok'] is True
    assert len(mg['per_rank_img_s']) == 2 and all(v > 0 for v in mg['per_rank_img_s'])
    # the plain form, no launcher: `python bench.py --gpus 2` starts its own ranks under torch.distributed.run on a free port
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run-dist', '--bs', '2', '--hw', '32',
                         '--arch', 'resnet18'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    lines = [l for l in r1.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r1.stdout
    d1 = json.loads(lines[0])
    assert d1['n_gpus'] == 2 and d1['multi_gpu']['rccl_ranks'] == 2 and d1['multi_gpu']['logits_gathered_ok'] is True and d1['steps'] == 3
    # a failing rank's exit code comes back through the self-launch (here: an unknown architecture)
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run-dist', '--bs', '2', '--hw', '32', '--arch', 'no_such_net'],
                        env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r2.returncode != 0 and not [l for l in r2.stdout.splitlines() if l.startswith('{')]
    # under a launcher --gpus must match its world size
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run-dist', '--bs', '2', '--hw', '32', '--arch', 'resnet18'],
                        env=dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'), cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r3.returncode != 0 and 'WORLD_SIZE' in (r3.stderr + r3.stdout)


def _verify_worker(rank, world, port, corrupt, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch
    import torch.distributed as dist
    from f8net_amd import dist as f8dist
    f8dist.init_from_env(backend='gloo')
    n, classes = 3, 5
    local = (torch.arange(n * classes, dtype=torch.float32).reshape(n, classes) + 1000 * rank)
    full = torch.empty((n * world, classes))
    dist.all_gather_into_tensor(full, local)
    if corrupt and rank == 1:
        full[0, 2] += 1                       # rank 1's copy of rank 0's block differs from what rank 0 computed
    q.put((rank, f8dist.verify_gather(local, full, n, 10, 0.5)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('corrupt', [False, True])
def test_verify_gather_flags_a_block_that_differs_from_its_owners_checksum(corrupt):
    """dist.verify_gather: every rank checks every gathered block against the checksum of the rank that produced it; ONE wrong value on ONE
    rank turns `logits_gathered_ok` false on EVERY rank (the verdicts are AND-reduced)."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29691 + int(corrupt)
    ps = [ctx.Process(target=_verify_worker, args=(r, 2, port, corrupt, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert got[r]['rccl_ranks'] == 2 and got[r]['logits_gathered_ok'] is (not corrupt), got
        assert got[r]['per_rank_img_s'] == [60.0, 60.0]


def _refusing_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch
    import torch.distributed as dist
    from f8net_amd import dist as f8dist
    f8dist.init_from_env(backend='gloo')
    n, classes = 2, 4
    calls = [0]

    def local(t, o):
        calls[0] += 1
        if rank == 1 and calls[0] >= 2:       # f8_net_run refusing once the host mirror shows an earlier run's chain time-out
            raise RuntimeError('f8_net_run: a stage-chain launch of an earlier run gave up waiting')
        o.copy_(t.reshape(n, -1)[:, :classes].to(torch.float32) + 100 * rank)

    pf = f8dist.PipelinedShardedForward(local, classes, n, torch.device('cpu'))
    outs = [pf(torch.full((n, 8), float(step))) for step in range(4)]   # every rank issues 4 collectives
    err = None
    try:
        pf.finish()
    except RuntimeError as e:
        err = str(e)
    last = outs[-1]
    sc = f8dist.verify_gather(pf.local[(pf.i - 1) % pf.depth], last, n, 4, 1.0)
    q.put((rank, err, bool(torch.isnan(last[n:]).all()), bool(torch.isfinite(last[:n]).all()), sc))
    dist.barrier()
    dist.destroy_process_group()


def test_a_rank_whose_forward_refuses_still_enters_the_collective():
    """f8_net_run returns an error on ONE rank (host mirror of a chain time-out) while its peers are already in the all-gather.
    PipelinedShardedForward keeps that rank in every collective with a NaN block and raises from finish(); verify_gather rejects the poisoned
    block although it equals its owner's checksum."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_refusing_worker, args=(r, 2, 29697, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {r: rest for r, *rest in (q.get(timeout=120) for _ in ps)}
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] is None and 'gave up waiting' in got[1][0]          # only the refusing rank raises, and only from finish()
    for r in (0, 1):
        err, peer_nan, own_ok, sc = got[r]
        assert peer_nan and own_ok                                        # rank 1's block arrived on both ranks, poisoned; rank 0's block is intact
        assert sc['logits_gathered_ok'] is False and sc['nonfinite_per_rank'] == [0, 8] and sc['rccl_ranks'] == 2
