"""Batch sharding across the GPUs of one node: one process per GPU, RCCL over xGMI.

The path shards naturally — images are independent, the ≤ 26 MB of packed weights are replicated
per GPU — so ranks run their contiguous slice of the batch with no data-path collective.  The one
exchange step is an all-gather of the per-rank logits (fp32 [n/world, classes], ≈ 0.5 MB per rank
at 128 images), which replaces the reference's `DataParallel` gather
(/root/reference/fix_train.py:269) and metric all-reduce (fix_train.py:705-707,
myutils/distributed.py:79-87).  At that size the collective is latency-bound; ring vs direct over
the 7 xGMI links does not matter.

`torch.distributed` backend "nccl" is RCCL on ROCm; the CPU tests drive the same code over gloo.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group the launcher (torchrun / torch.distributed.run) described in the env.
    Returns (rank, world_size, local_rank).  Single-process runs need no group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_bounds(n_total: int, world: int, rank: int):
    """Contiguous slice [lo, hi) of a batch of n_total images owned by `rank`; the first
    n_total % world ranks take one extra image (ragged batches are legal)."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedForward:
    """Data-parallel forward: `forward_local` maps this rank's images to fp32 logits (on the GPU:
    `F8Net.run`); the logits of all ranks are all-gathered in rank order."""

    def __init__(self, forward_local, num_classes: int, group=None, force_collective=False):
        self.forward_local = forward_local
        self.num_classes = num_classes
        self.group = group
        # tests on a 1-GPU box: a process group of ONE rank still issues the collective (RCCL's stream, its ordering against the net's streams)
        self.force_collective = bool(force_collective)

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def gather(self, local_logits, n_total=None):
        """All-gather per-rank logits [n_r, classes] -> [sum n_r, classes] on every rank."""
        world = self.world
        if world == 1 and not self.force_collective:
            return local_logits
        n_local = local_logits.shape[0]
        if n_total is None or n_total == n_local * world:
            out = local_logits.new_empty((n_local * world, self.num_classes))
            dist.all_gather_into_tensor(out, local_logits.contiguous(), group=self.group)
            return out
        # ragged shards: pad to the largest shard, gather, drop the padding
        sizes = [shard_bounds(n_total, world, r) for r in range(world)]
        cap = max(hi - lo for lo, hi in sizes)
        padded = local_logits.new_zeros((cap, self.num_classes))
        padded[:n_local] = local_logits
        out = local_logits.new_empty((cap * world, self.num_classes))
        dist.all_gather_into_tensor(out, padded, group=self.group)
        return torch.cat([out[r * cap: r * cap + (hi - lo)] for r, (lo, hi) in enumerate(sizes)])

    def __call__(self, x_local, n_total=None):
        return self.gather(self.forward_local(x_local), n_total)


class PipelinedShardedForward(ShardedForward):
    """Same exchange, off the critical path: the all-gather of batch i runs on RCCL's stream (`async_op=True`) while
    batch i+1 is already computing.  `forward_local(x, out)` must write its logits into `out`: two (local, gathered)
    buffer pairs alternate, and a pair is only reused after its collective has been waited for.  The tensor returned by
    `__call__` is complete after `finish()` (or after the pair comes round again).  Equal shards only."""

    def __init__(self, forward_local, num_classes: int, n_local: int, device, group=None, lagged=False, depth=2, force_collective=False):
        super().__init__(forward_local, num_classes, group, force_collective)
        self.n_local = n_local
        self.depth = depth = max(2, int(depth))     # buffer pairs = runs the net keeps in flight (option pipeline_depth)
        # lagged: the net runs with f8_net_set_pipelined(1), whose contract wants a buffer free ONE CALL before the run that
        # writes it: the collective of the previous batch is then waited for at the top of this call (the wait lands on the
        # caller's stream, which the net's sub-batch streams only follow with one call of lag: it stalls no compute)
        self.lagged = lagged
        self.local = [torch.empty((n_local, num_classes), dtype=torch.float32, device=device) for _ in range(depth)]
        world = self.world
        self.full = [torch.empty((n_local * world, num_classes), dtype=torch.float32, device=device) for _ in range(depth)] \
            if (world > 1 or self.force_collective) else self.local
        self.work = [None] * depth
        self.i = 0
        self.error = None           # first exception a forward raised on this rank while collectives were outstanding (re-raised by finish())

    def __call__(self, x_local, n_total=None):
        k = self.i % self.depth
        self.i += 1
        # lagged: the buffer of the run `depth - 1` calls ahead (pair k - 1) has to be free by the time THIS run is submitted
        for j in ((k, (k - 1) % self.depth) if self.lagged else (k,)):
            if self.work[j] is not None:
                self.work[j].wait()        # a pair's previous collective is done before its buffers are rewritten
                self.work[j] = None
        try:
            self.forward_local(x_local, self.local[k])
        except Exception as e:                                      # noqa: BLE001
            # a rank whose forward refuses (f8_net_run returns an error once the host mirror shows a chain time-out) must still take part in this
            # step's collective: its peers are already in it and would block until the backend's time-out.  It contributes a POISONED block
            # (NaN: what the classifier writes while the error word is set) and the error is raised by finish(), after the outstanding collectives.
            if self.world == 1 and not self.force_collective:
                raise
            self.local[k].fill_(float('nan'))
            if self.error is None:
                self.error = e
        if self.world > 1 or self.force_collective:
            self.work[k] = dist.all_gather_into_tensor(self.full[k], self.local[k], group=self.group, async_op=True)
        return self.full[k]

    def finish(self):
        for k in range(self.depth):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None
        if self.error is not None:
            e, self.error = self.error, None
            raise e


def verify_gather(local_logits, gathered_logits, n_local: int, steps: int, dt_own: float, group=None):
    """Self-check of a multi-rank run (bench.py --gpus N), collective on every rank:

    * `rccl_ranks`: the number of distinct rank ids an actual all-gather returned (= the ranks the backend really connected);
    * `per_rank_img_s`: each rank's OWN rate over its own clock (the headline takes the slowest rank's time);
    * `logits_gathered_ok`: every rank compares each block of ITS gathered logits with a checksum the owning rank computed from its local
      logits (an int64 sum and an order-sensitive weighted int64 sum, both exact), then the verdicts are AND-reduced.  A block that holds a
      non-finite value — the poison a timed-out chain launch leaves in its logits — fails the check even when it matches its owner's;
    * `nonfinite_per_rank`: non-finite values in each rank's local block.

    Replaces what the reference's DataParallel gather / metric all-reduce would silently assume (fix_train.py:269, myutils/distributed.py:79-87)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = local_logits.device

    def checksum(t):
        # the logits are int32 values converted to float32 (integer-valued): summed as int64 the checksum is exact and independent of the reduction
        # order (int64 wrap-around included); a poisoned buffer (NaN: f8_fc.hip) is marked, not converted
        t = t.detach().reshape(-1)
        bad = (~torch.isfinite(t)).sum().to(torch.float64)
        v = torch.nan_to_num(t, nan=0.0, posinf=0.0, neginf=0.0).to(torch.int64)
        w = torch.arange(1, v.numel() + 1, dtype=torch.int64, device=v.device) % 65521
        return torch.stack([v.sum().to(torch.float64), (v * w).sum().remainder(1 << 40).to(torch.float64), bad])

    mine = torch.cat([torch.tensor([float(rank), n_local * steps / max(dt_own, 1e-12)], dtype=torch.float64, device=dev), checksum(local_logits[:n_local])])
    if world > 1:
        table = torch.empty((world, mine.numel()), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(table, mine[None].contiguous(), group=group)
    else:
        table = mine[None]
    ok = 1
    for r in range(world):
        blk = gathered_logits[r * n_local:(r + 1) * n_local] if world > 1 else local_logits[:n_local]
        cs = checksum(blk)
        if blk.shape[0] != n_local or not torch.equal(cs, table[r, 2:]) or float(cs[2]) > 0 or float(table[r, 4]) > 0:
            ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    ids = sorted({int(v) for v in table[:, 0].tolist()})
    return {'rccl_ranks': len(ids), 'rank_ids': ids, 'backend': dist.get_backend(group) if dist.is_initialized() else None,
            'per_rank_img_s': [round(float(v), 1) for v in table[:, 1].tolist()], 'nonfinite_per_rank': [int(v) for v in table[:, 4].tolist()],
            'logits_gathered_ok': bool(int(flag.item()) == 1)}
