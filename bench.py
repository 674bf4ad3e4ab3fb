#!/usr/bin/env python3
"""Headline benchmark: ResNet-50 fixed-point-8 integer forward, bs = 128 images per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: re-executes itself under torch.distributed.run,
                                                                  one rank per GPU, free port on 127.0.0.1, exit code propagated)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path over one synthetic batch already resident in HBM:
int32 NCHW images (the reference's input format, fix_train.py:683-692) -> fused HIP launches
(libf8net.so) -> fp32 logits; with N > 1 every rank runs its own 128 images (weak scaling) and the
logits are all-gathered over RCCL (the path's one exchange step).  Rank 0 prints ONE JSON line.

Parameters: the reference's real learned fraction lengths (ResNet-50: NVIDIA-pretrained run,
fraclen_visual/res50_fix_quant_nvidia_pretrained.out:492-1138, `normalize: True`, signed head input;
MobileNet-V2: fraclen_visual/mbv2_fix_quant.out:1267-1901) with seeded synthetic int8 weights —
Model-Zoo checkpoints are unreachable (no network).

Other BASELINE.json configurations:  --arch resnet18 --bs 128 (C2), --arch mobilenet_v2 --bs 128 (C3),
--arch resnet50 --bs 256 (C4), --gpus 8 --bs 256 (C5: 2048 images over 8 GPUs).

value            : the timed region is EXACTLY --steps steps with F8_PIPELINE_DEPTH (default 3) batches in flight (f8_net_set_pipelined(2),
                   options arena_copies = pipeline_depth = 3), on the library's DEFAULT plan: integer-only requantisation in every epilogue
                   (option requant_float = 0: shift / round-half-even / clamp, no float instruction — BASELINE north_star).  The timed loop
                   rotates NX = 4 DISTINCT device-resident input batches (308 MB at bs 128: no step re-reads the batch of the step before).
value_float_requant: the same steps on a handle planned with requant_float = 1 (float-converter requantisation where the planner bounds the value).
value_extended / value_float_requant_extended: when the K-step region lasted under 0.2 s (the driver's 20 steps = 17 ms: +-5 % jitter), both plans again over
                   a region of >= 0.25 s — the only pair of numbers of a short run whose RATIO means anything.
value_unpipelined: the same steps, one batch in flight (runs back to back; each run = two concurrent sub-batches).
latency          : per-batch latency (submission on the host -> logits complete on the device), closed loop with 1 and with `depth` batches
                   outstanding: p50 / p99 over the batches, beside the closed-loop rate.
roofline         : dominant kernel symbol by time; achieved = sum(algorithmic bytes of its launches) / sum(their
                   durations), durations from HIP events on the launch stream (f8_net_run_profiled).  `traffic` (HBM bytes per
                   launch, separate rocprofv3 --pmc passes) and `mfma` (INT8-MFMA busy fraction, rocprofv3 --pmc) are taken from
                   profiles/pmc_*.json ONLY when those files were produced from the same kernel sources (sha256 stamp).
cpu_baseline     : the CPU oracle (oracle/, a port of the reference's int32 CPU forward) timed on the host cores for a bounded
                   sample of the same workload, rank 0 at N = 1 only; `c1` = BASELINE config 1 (ResNet-18, bs 1) timed the same way.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_I8_PEAK_TOPS = 5033.0     # 256 CU x 4 SIMD x 2048 op/clk x 2.4 GHz
# per image: integer ops (2 * MACs) and the structural byte model of SURVEY.md §8d
OPS_PER_IMG = {'resnet50': 8.178368512e9, 'resnet18': 3.628146688e9, 'mobilenet_v2': 0.601548544e9, 'mobilenet_v1': 1.137480704e9}
STRUCT_BYTES_PER_IMG = {'resnet50': 93444000.0, 'resnet18': 20830112.0, 'mobilenet_v2': 20602656.0, 'mobilenet_v1': 28911520.0}
PRETTY = {'resnet50': 'ResNet-50', 'resnet18': 'ResNet-18', 'mobilenet_v2': 'MobileNet-V2', 'mobilenet_v1': 'MobileNet-V1'}
MIN_TIMED_S = 0.2              # below this the two-deep pipeline's fill / drain is a visible share of the timed region


def csrc_sha256():
    """Stamp of the kernel sources the in-tree library is built from (profiles/pmc_*.json carry the stamp of the build they measured)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'f8net_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h', '.cpp')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()


def stamped_json(name, stamp):
    """profiles/<name> if it was measured on this build, else (None, reason)."""
    path = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(path):
        return None, f'profiles/{name} missing'
    try:
        d = json.load(open(path))
    except Exception as e:                                          # noqa: BLE001
        return None, f'profiles/{name} unreadable ({e})'
    if d.get('csrc_sha256') != stamp:
        return None, f'profiles/{name} was measured on other kernel sources (stamp mismatch): dropped'
    return d, None


def cpu_baseline(spec, params, x_np, x_fl, ref_logits, budget_s=12.0):
    """Time the oracle on a bounded sample (~10-15 s of CPU work) and check it against the GPU."""
    import numpy as np
    from oracle import oracle
    oracle.build()
    y1 = oracle.net_forward(spec, params, x_np[:1], x_fl)            # warm-up (thread pool, page faults), also checked below
    t0 = time.time()
    oracle.net_forward(spec, params, x_np[:4], x_fl)
    t4 = (time.time() - t0) / 4                                     # seconds per image at a small batch
    n = int(max(1, min(x_np.shape[0], budget_s / max(t4, 1e-3))))
    t0 = time.time()
    y = oracle.net_forward(spec, params, x_np[:n], x_fl)
    dt = time.time() - t0
    ok = bool(np.array_equal(y, ref_logits[:n]) and np.array_equal(y1, ref_logits[:1]))
    return {'value': round(n / dt, 3), 'unit': 'img/s', 'cores': oracle.num_threads(), 'kind': 'port',
            'sample': f'{n} images of the same batch ({PRETTY.get(spec.arch, spec.arch)}, {x_np.shape[2]}x{x_np.shape[3]}), one forward, {dt:.1f} s; '
                      f'host has {os.cpu_count()} logical cpus',
            'matches_gpu_bit_exact': ok}


def cpu_config1():
    """BASELINE.json configs[0]: ResNet-18 fix_quant int_op_only, bs = 1, CPU forward (the reference's own CPU-runnable case,
    BASELINE.md §3: 0.287 s per forward on 8 Xeon vCPUs with the reference's ATen path) — the oracle, timed per forward."""
    from f8net_amd import synth, topology
    from oracle import oracle
    spec = topology.get('resnet18')
    params = synth.make_params(spec, seed=1234)
    x, fl = synth.make_input(spec, params, 1, 224, seed=1)
    oracle.net_forward(spec, params, x, fl)
    reps, t0 = 0, time.time()
    while reps < 5 or time.time() - t0 < 2.0:
        oracle.net_forward(spec, params, x, fl)
        reps += 1
    dt = (time.time() - t0) / reps
    return {'config': 'ResNet-18 fix_quant int_op_only, bs=1, 224x224, CPU forward', 'sec_per_forward': round(dt, 5), 'value': round(1.0 / dt, 2), 'unit': 'img/s',
            'cores': oracle.num_threads(), 'kind': 'port', 'sample': f'{reps} forwards of one image'}


def launch_ranks(n):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: run this very command line under `python -m torch.distributed.run`, one rank per
    GPU of this node, rendezvous on 127.0.0.1 at a free port (what the reference's distributed_run.sh:1-12 / fix_train.py:269 leave to the user's
    launcher).  The ranks inherit stdout / stderr — rank 0 still prints the ONE JSON line — and the launcher's exit code is this process's."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: without it RCCL's hipIpcGetMemHandle fails on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    print(f'bench.py: --gpus {n} without a launcher: {" ".join(cmd[1:9])} ...', file=sys.stderr)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--arch', default='resnet50', help='resnet50 (headline) | resnet18 | mobilenet_v2 | mobilenet_v1')
    ap.add_argument('--bs', type=int, default=128, help='images per GPU (the headline metric is quoted at 128)')
    ap.add_argument('--per-layer', action='store_true', help='also print the per-launch table to stderr')
    ap.add_argument('--dry-run-dist', action='store_true', help='no GPU: the rendezvous / sharding / fence / max-over-ranks / JSON path of an N-GPU launch over gloo '
                    'with a STUB forward (a deterministic function of the images, not a measurement): what a first 8-GPU launch must not fail on')
    ap.add_argument('--hw', type=int, default=224, help=argparse.SUPPRESS)
    ap.add_argument('--autotune', action='store_true', help='measured tile selection (f8_net_autotune) instead of the planner heuristics; '
                    'measured: re-tiles ~12 launches, gain within run-to-run noise, so off by default')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    from f8net_amd import dist as f8dist
    from f8net_amd import synth, topology
    from f8net_amd.net import build_net

    BS = args.bs
    dry = args.dry_run_dist
    rank, world, local_rank = f8dist.init_from_env(backend='gloo' if dry else None)
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    if world > 1 and not dry:
        assert dist.get_backend() == 'nccl', f'N > 1 runs over RCCL (backend nccl), not {dist.get_backend()}'
    if dry:
        dev = torch.device('cpu')
    else:
        assert torch.cuda.is_available(), 'bench.py needs a GPU (there is no CPU product path)'
        dev = torch.device('cuda', local_rank)
        torch.cuda.set_device(dev)

    def dev_sync():
        if dev.type == 'cuda':
            torch.cuda.synchronize(dev)

    normalize = args.arch == 'resnet50'
    spec = topology.get(args.arch, normalize=normalize)
    params = synth.reference_params(spec, seed=1234)
    fr_name = {'resnet50': 'NVIDIA-pretrained fraclens (normalize: True)', 'mobilenet_v2': "the reference log's learned fraclens (mbv2_fix_quant.out)"}.get(
        args.arch, 'seeded fraclens (weight_format [8,7]-style)')
    x_np, x_fl = synth.make_input(spec, params, BS, args.hw, seed=1 + rank)
    NX = 4                          # distinct input batches the timed loops rotate (one tensor re-read every step stays warm in the 256 MB memory-side cache)
    pipe_mode = int(os.environ.get('F8_BENCH_PIPELINED', '2'))
    # planning hint: under pipelining mode 2 every launch covers the whole batch (matters for the 14x14 fusion rule at bs 64..127)
    # ... and `depth` whole batches are in flight, one arena copy each (a run with ONE batch in flight still cuts it `split` = 2 ways)
    depth = max(2, min(4, int(os.environ.get('F8_PIPELINE_DEPTH', '3')))) if pipe_mode == 2 else 2
    if dry:
        class _Stub:            # stands in for F8Net on a box without a GPU: logits[i, c] = (sum of image i + c) mod 1000 — plumbing only, never a result
            def run(self, t, out=None):
                out.copy_(((t.reshape(t.shape[0], -1).sum(1, keepdim=True) + torch.arange(spec.num_classes)) % 1000).to(torch.float32))
                return out
            def set_pipelined(self, mode): pass
            def check(self): return self
            def get_option(self, key): return 1
        net = _Stub()
        retiled = 0
    else:
        net = build_net(spec, params, max_batch=BS, hw=args.hw,
                        options={'whole_batch_launches': 1, 'arena_copies': depth, 'pipeline_depth': depth} if pipe_mode == 2 else None)
        net.upload()
        retiled = net.autotune(BS, dev) if args.autotune else 0      # one-time, outside the timed region
    x = torch.from_numpy(x_np).to(dev)
    xs = [x] + [torch.from_numpy(synth.make_input(spec, params, BS, args.hw, seed=1 + rank + 7919 * j)[0]).to(dev) for j in range(1, NX)]
    # the all-gather of step i overlaps the compute of step i+1 (double-buffered logits); fence() completes every
    # outstanding collective before the clock stops
    # consecutive steps overlap inside the library as well (f8_net_set_pipelined: static input, double-buffered outputs)
    # F8_BENCH_PIPELINED: 0 = runs back to back, 1 = lagged sub-batches, 2 = whole batches alternating between two streams

    def timed(mode, steps, warmup, net=net):
        net.set_pipelined(mode)
        sharded = f8dist.PipelinedShardedForward(lambda t, out: net.run(t, out=out), spec.num_classes, BS, dev, lagged=mode != 0, depth=depth)

        def fence():
            sharded.finish()
            if world > 1:
                dist.barrier()
            dev_sync()

        out = None
        for j in range(warmup):
            sharded(xs[j % NX])
        fence()
        t0 = time.perf_counter()
        for j in range(steps):
            out = sharded(xs[(warmup + j) % NX])
        fence()
        dt_own = dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        assert out.shape == (BS * world, spec.num_classes)
        net.check()                         # device error words (a chain launch's halo wait timed out; int32 input outside the head's format): raises
        # the LAST step: index of its input batch, its local logits, the gathered logits, this rank's own clock
        last = {'x': (warmup + steps - 1) % NX, 'local': sharded.local[(sharded.i - 1) % sharded.depth], 'full': out, 'dt_own': dt_own}
        return dt, last

    def closed_loop(mode, outstanding, steps, warmup):
        """Per-batch latency: a batch is submitted only while fewer than `outstanding` are in flight (the host waits for the oldest one's logits);
        latency = host submission -> completion event of that batch's logits (device timeline anchored to the host clock at a synchronised start)."""
        net.set_pipelined(mode)
        nb = max(2, depth) + 1
        outs = [torch.empty((BS, spec.num_classes), dtype=torch.float32, device=dev) for _ in range(nb)]
        for j in range(warmup):
            net.run(xs[j % NX], out=outs[j % nb])
        dev_sync()
        base = torch.cuda.Event(enable_timing=True)
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        base.record()
        base.synchronize()
        t0 = time.perf_counter()
        sub_t = []
        for j in range(steps):
            if j >= outstanding:
                ends[j - outstanding].synchronize()
            sub_t.append(time.perf_counter() - t0)
            net.run(xs[j % NX], out=outs[j % nb])
            ends[j].record()
        dev_sync()
        dtc = time.perf_counter() - t0
        lat = sorted(base.elapsed_time(ends[j]) - 1e3 * sub_t[j] for j in range(steps))
        net.check()
        return {'outstanding': outstanding, 'p50_ms': round(lat[len(lat) // 2], 4), 'p99_ms': round(lat[min(len(lat) - 1, int(0.99 * len(lat)))], 4),
                'max_ms': round(lat[-1], 4), 'img_per_s': round(BS * steps / dtc, 1), 'batches': steps}

    if dry:
        dt, last = timed(pipe_mode, args.steps, args.warmup)
        selfcheck = f8dist.verify_gather(last['local'], last['full'], BS, args.steps, last['dt_own'])
        # every rank's gathered logits must hold every rank's shard, in rank order (the stub forward is a function of the images alone)
        full = f8dist.ShardedForward(lambda t: net.run(t, out=torch.empty((BS, spec.num_classes))), spec.num_classes)(x)
        for r in range(world):
            xr = torch.from_numpy(synth.make_input(spec, params, BS, args.hw, seed=1 + r)[0])
            assert torch.equal(full[r * BS:(r + 1) * BS], net.run(xr, out=torch.empty((BS, spec.num_classes)))), f'rank {rank}: shard {r} of the gathered logits'
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({'metric': f'images/sec at bs={BS} ({PRETTY.get(args.arch, args.arch)} INT8)', 'value': round(BS * world * args.steps / dt, 1), 'unit': 'img/s',
                              'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 4), 'higher_is_better': True,
                              'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int8 x int8 -> int32 (exact integer)', 'data': 'synthetic',
                              'multi_gpu': selfcheck,
                              'dry_run': 'gloo on CPU with a stub forward: NOT a measurement (rendezvous, sharding, fence, max over ranks and this line only)',
                              'config': {'workload': f'{spec.arch} dry run', 'global_batch': BS * world, 'parallelism': f'dp{world} (batch shards + all-gather of logits)'}}))
        return
    lean = os.environ.get('F8_BENCH_LEAN', '0') == '1'      # profiling runs (tools/profile.sh): the headline loop only, fewer kernel records
    # Order: the secondary measurements first (one batch in flight; the per-launch roofline pass), the headline region last.
    # Each region has its own warm-up and is bracketed by barrier + synchronize; running the secondary ones first also means the
    # headline region does not start on a cold device (measured at --steps 20: 5 warm-up steps from idle 81.7 k, from a busy
    # device 83.2 k img/s — power state / caches, not arithmetic).
    # (1) the same K steps with ONE batch in flight (every rank takes part: the loop holds collectives)
    dt0, _ = timed(0, args.steps, min(args.warmup, 5)) if (pipe_mode != 0 and not lean) else (None, None)
    # (2) per-launch durations for the roofline (rank 0; HIP events on the launch stream; the pipelining mode is set so that the
    #     profiled pass issues the launches the headline region issues: whole batch vs sub-batches)
    samples = []
    if rank == 0:
        net.set_pipelined(pipe_mode)
        scratch = torch.empty((BS, spec.num_classes), dtype=torch.float32, device=dev)
        for _ in range(7):
            _, ms_ = net.run_profiled(xs[len(samples) % NX], out=scratch)
            samples.append(ms_)
    if world > 1:
        dist.barrier()
    # (3) the headline: W warm-up steps, then EXACTLY K timed steps
    dt, last = timed(pipe_mode, args.steps, args.warmup)
    logits, x_last = last['local'], last['x']
    # N > 1: the first multi-GPU launch verifies itself: the ranks RCCL really connected, each rank's own rate, and every rank's
    # gathered block against a checksum its owner computed
    selfcheck = f8dist.verify_gather(last['local'], last['full'], BS, args.steps, last['dt_own']) if world > 1 else None
    if dt0 is None:
        dt0 = dt
    # a timed region under MIN_TIMED_S is dominated by pipeline fill / drain and launch jitter: report a longer one beside it
    ext = None
    if dt < MIN_TIMED_S and not lean:
        k2 = int(max(args.steps * 2, min(20000, args.steps * (1.25 * MIN_TIMED_S / max(dt, 1e-6)))))
        dte, _ = timed(pipe_mode, k2, 2)
        ext = (k2, dte)
        if rank == 0:
            print(f'bench.py: the timed region of {args.steps} steps lasted {dt * 1e3:.1f} ms (< {MIN_TIMED_S} s); also timed {k2} steps '
                  f'({dte * 1e3:.1f} ms) -> value_extended', file=sys.stderr)
    net.set_pipelined(pipe_mode)
    # (4b) the same K steps with the FLOAT-CONVERTER requantisation (option requant_float = 1: ReLU -> unsigned 8-bit right shifts of values the planner
    #      bounds run v_cvt_f32_i32, v_mul_f32 by 2^-n, v_cvt_pk_u8_f32 — exact, compared over all 2^32 values on the device:
    #      tests/test_gpu_requant_probe.py).  The headline plan above is the library default: INTEGER shift / round-half-even / clamp in every
    #      epilogue, no float instruction.  Same logits, two arithmetic paths.
    rq_float = bool(net.get_option('requant_float'))
    assert not rq_float or os.environ.get('F8_REQUANT_FLOAT') == '1', 'the headline runs the integer-only plan'
    extra_rq = {}
    if world == 1 and not lean and not rq_float:
        opts_f = {'requant_float': 1}
        if pipe_mode == 2:
            opts_f.update({'whole_batch_launches': 1, 'arena_copies': depth, 'pipeline_depth': depth})
        net_f = build_net(spec, params, max_batch=BS, hw=args.hw, options=opts_f)
        net_f.upload()
        dtf, last_f = timed(pipe_mode, args.steps, args.warmup, net=net_f)
        extra_rq = {'value_float_requant': round(BS * args.steps / dtf, 1), 'float_requant_matches': bool(last_f['x'] == x_last and torch.equal(last_f['local'][:BS], logits[:BS]))}
        if ext is not None:     # a region under MIN_TIMED_S cannot resolve the few-percent integer / float difference: the ratio to quote is value_extended / value_float_requant_extended
            dtfe, _ = timed(pipe_mode, ext[0], 2, net=net_f)
            extra_rq['value_float_requant_extended'] = round(BS * ext[0] / dtfe, 1)
        del net_f
        net.set_pipelined(pipe_mode)
    # (4c) per-batch latency, closed loop: one batch outstanding (mode 0) and `depth` outstanding (the headline's schedule)
    latency = None
    if world == 1 and not lean:
        nlat = max(20, min(args.steps, 200))
        latency = {'definition': 'host submission -> logits complete on the device, closed loop (a batch is submitted when fewer than `outstanding` are in flight)',
                   'depth1': closed_loop(0, 1, nlat, 3), f'depth{depth}': closed_loop(pipe_mode, depth, nlat, depth + 2) if pipe_mode == 2 else None}
        net.set_pipelined(pipe_mode)

    # (5) the drop-in module path and the host-fed path (single GPU, full runs only): the same K steps
    #     (a) through IntModel.forward — the nn.Module the reference's fix_resnet.py / fix_mobilenet_v*.py callers hold;
    #     (b) from HOST-resident uint8 batches (page-locked, NHWC as a decoder writes them): H2D on a copy stream, f8_net_run_u8 (ToTensor /
    #         Normalize / input quantisation inside the input kernel), top-k on the device — f8net_amd/stream_eval.py, the caller side of the
    #         reference's test epoch (fix_train.py:959-969).  `value` itself stays the device-resident rate.
    extra = {}
    if world == 1 and not lean:
        from f8net_amd import int_model, stream_eval
        m = int_model.from_params(spec, params).to(dev)
        m.set_pipelined(pipe_mode if pipe_mode else 0, depth=depth)
        xi = xs[x_last].clone(); setattr(xi, 'output_fraclen', x_fl)
        NO = depth + 1
        outs = [torch.empty((BS, spec.num_classes), dtype=torch.float32, device=dev) for _ in range(NO)]
        for i in range(min(args.warmup, 10) + 2):
            m.forward(xi, out=outs[i % NO])
        dev_sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            m.forward(xi, out=outs[i % NO])
        dev_sync()
        dtm = time.perf_counter() - t0
        extra['value_intmodel'] = round(BS * args.steps / dtm, 1)
        extra['intmodel_matches'] = bool(torch.equal(outs[(args.steps - 1) % NO], logits[:BS]))
        del m
        if spec.head.cin == 3:
            hnet = build_net(spec, params, max_batch=BS, hw=args.hw, options={'whole_batch_launches': 1, 'arena_copies': depth, 'pipeline_depth': depth})
            ev = stream_eval.StreamEvaluator(hnet, normalize=normalize, mean=stream_eval.IMAGENET_MEAN, std=stream_eval.IMAGENET_STD, device=dev, depth=depth + 1)
            g = torch.Generator().manual_seed(11)
            pool = [torch.randint(0, 256, (BS, args.hw, args.hw, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(4)]
            labels = [torch.randint(0, spec.num_classes, (BS,), dtype=torch.int64, generator=g) for _ in range(4)]
            ev.run(((pool[i % 4], labels[i % 4]) for i in range(min(args.warmup, 10) + 2)))
            r = ev.run(((pool[i % 4], labels[i % 4]) for i in range(args.steps)))
            extra['value_host_fed'] = round(r['img_per_s'], 1)
            extra['host_fed'] = {'input': f'uint8 NHWC [{BS},{args.hw},{args.hw},3] per batch in page-locked host memory ({BS * args.hw * args.hw * 3 / 1e6:.1f} MB), H2D on a copy stream, '
                                          f'f8_net_run_u8 + f8_topk_correct_f32, {depth} batches in flight', 'top1_on_random_labels': r['top1']}
            del ev, hnet

    result = None
    if rank == 0:
        imgs = BS * world * args.steps
        value = imgs / dt
        # ---- roofline of the dominant kernel, measured live (HIP events on the launch stream); the pipelining mode stays
        #      set so that the profiled pass issues the launches the timed region issued (whole batch vs sub-batches)
        n_l = net.num_launches
        reps = len(samples)
        # per-launch median over the repetitions (one stray multi-millisecond event pair must not pick the "dominant" kernel)
        ms = [sorted(s[i] for s in samples)[reps // 2] for i in range(n_l)]
        by_kernel = {}
        rows = []
        for i in range(n_l):
            name, nbytes, nops = net.launch_info(i, BS)
            k = net.launch_kernel(i)
            e = by_kernel.setdefault(k, {'ms': 0.0, 'bytes': 0.0, 'ops': 0.0, 'valu': 0.0, 'launches': 0})
            e['ms'] += ms[i]; e['bytes'] += nbytes; e['ops'] += nops; e['valu'] += net.launch_valu(i, BS); e['launches'] += net.step_launches(i, BS)
            rows.append((i, name, ms[i], nbytes, nops))
        parts = net.num_parts(BS)       # each planned launch is issued once per sub-batch
        dom = max(by_kernel, key=lambda k: by_kernel[k]['ms'])
        d = by_kernel[dom]
        d_launches = d['launches']       # kernel launches per step (sub-batches / chunks included)
        achieved = d['bytes'] / (d['ms'] * 1e-3) / 1e9
        ach_tops = d['ops'] / (d['ms'] * 1e-3) / 1e12
        mfma_frac = ach_tops / MFMA_I8_PEAK_TOPS
        stamp = csrc_sha256()
        notes = []
        traffic = mfma = None
        tj, why = stamped_json(f'pmc_traffic_{args.arch}_bs{BS}.json', stamp)    # HBM bytes per launch from separate rocprofv3 --pmc passes
        if tj is not None and tj.get('workload') == f'{args.arch}/bs{BS}':
            traffic = tj.get('kernels', {}).get(dom, {}).get('hbm_bytes_per_launch')
        elif why:
            notes.append(why)
        mj, why = stamped_json(f'pmc_mfma_{args.arch}_bs{BS}.json', stamp)       # INT8-MFMA busy cycles per kernel (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES)
        if mj is not None and mj.get('workload') == f'{args.arch}/bs{BS}':
            mk = mj.get('kernels', {}).get(dom)
            mfma = {'counter': 'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CU x 4 SIMD), rocprofv3 --pmc',
                    'dominant_kernel_busy_frac': mk.get('mfma_busy_frac') if mk else None,
                    'whole_net_busy_frac': mj.get('whole_net_mfma_busy_frac'),
                    'whole_net_mfma_i8_insts_per_img': mj.get('whole_net_mfma_insts_per_img')}
        elif why:
            notes.append(why)
        limiter = None
        lj, why = stamped_json(f'pmc_limiter_{args.arch}_bs{BS}.json', stamp)    # what the waves do: two SQ counter passes of their own (tools/profile.sh)
        if lj is not None and lj.get('workload') == f'{args.arch}/bs{BS}':
            lk = lj.get('kernels', {}).get(dom)
            if lk:
                limiter = {'limiter': lk.get('limiter'),
                           'evidence': 'rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT '
                                       '(own pass; profiles/rocprof_*_valu.md): a pipe busy >= half of the launch names the limiter; "a+b in turn" when matrix pipe and vector issue together cover half of it in alternating phases; else latency (waves parked at s_waitcnt / s_barrier) or issue-stall',
                           **{k: lk.get(k) for k in ('wave_parked_frac', 'wave_issue_stall_frac', 'wave_issuing_frac', 'wave_issuing_valu_frac', 'valu_pipe_busy_frac', 'valu_issue_busy_frac', 'lds_busy_frac',
                                                     'mfma_busy_frac', 'hbm_frac_measured_bytes', 'lds_bank_conflict_per_lds_active')}}
        elif why:
            notes.append(why)
        total_ms = sum(ms)
        if args.per_layer:
            for i, name, m, b, o in rows:
                print(f'{i:3d} {name:58s} {m*1e3:8.1f} us {b/1e6:8.1f} MB {b/m/1e6 if m else 0:7.0f} GB/s '
                      f'{o/m/1e9 if m else 0:7.0f} TOP/s', file=sys.stderr)
            for k, e in sorted(by_kernel.items(), key=lambda kv: -kv[1]['ms']):
                if e['ms'] <= 0:         # a step that launches nothing (the input step when the stem launch reads the caller's buffer)
                    continue
                print(f'  {e["ms"]*1e3:8.1f} us {100*e["ms"]/total_ms:5.1f}% x{e["launches"]:2d}  '
                      f'{e["bytes"]/e["ms"]/1e6:7.0f} GB/s  {k}', file=sys.stderr)
        headline = args.arch == 'resnet50' and BS == 128
        result = {
            'metric': f'images/sec at bs={BS} ({PRETTY.get(args.arch, args.arch)} INT8)', 'value': round(value, 1), 'unit': 'img/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'int8 x int8 -> int32 (exact integer)', 'data': 'synthetic',
            'value_unpipelined': None if lean else round(imgs / dt0, 1),
            'config': {'workload': f'{spec.arch} fix_quant INT8 int_op_only forward, bs={BS} per GPU, 224x224, {fr_name}, '
                                   f'int32 NCHW input resident in HBM' + ('' if headline else ' [not the headline configuration]'),
                       'global_batch': BS * world, 'parallelism': f'dp{world} (batch shards + RCCL all-gather of logits)',
                       'launches_per_step': sum(net.step_launches(i, BS) for i in range(n_l)), 'sub_batches': parts, 'autotuned_launches': retiled,
                       'requant': 'integer: shift / round-half-even / clamp in every epilogue (v_bfe_u32, v_add3_u32, v_ashr_pk_u8_i32 / v_ashrrev_i32 + v_med3_i32), no float instruction '
                                  '(option requant_float = 0, the library default); value_float_requant = the same steps with requant_float = 1' if not rq_float else
                                  'float-converter where the planner bounds the value (F8_REQUANT_FLOAT=1 in the environment): NOT the headline arithmetic',
                       'inputs': f'{NX} distinct device-resident batches rotated step by step',
                       'schedule': {0: 'runs back to back (two concurrent sub-batches per run)',
                                    1: 'pipelined: sub-batches of consecutive runs overlap (f8_net_set_pipelined(1))',
                                    2: f'pipelined: {depth} consecutive batches in flight (one arena copy each), each launch covers a whole batch '
                                       '(f8_net_set_pipelined(2)); every timed step completes inside the timed region; '
                                       'value_unpipelined = the same steps with one batch in flight'}[pipe_mode]},
            # the roof that binds the dominant kernel = the one it is closer to: the stage-chain launches keep the int32 stream on the
            # chip and move ~1/8 of the bytes of the per-block launches, so for them it is the INT8-MFMA roof, not HBM
            'roofline': {'bound': 'mfma' if mfma_frac > achieved / HBM_PEAK_GBS else 'hbm', 'kernel': dom, 'launches_per_step': d_launches,
                         'achieved': round(ach_tops, 2) if mfma_frac > achieved / HBM_PEAK_GBS else round(achieved, 1),
                         'peak': MFMA_I8_PEAK_TOPS if mfma_frac > achieved / HBM_PEAK_GBS else HBM_PEAK_GBS,
                         'unit': 'TOP/s (int8, 2 ops per MAC; the TFLOP/s slot of the contract)' if mfma_frac > achieved / HBM_PEAK_GBS else 'GB/s',
                         'frac': round(max(mfma_frac, achieved / HBM_PEAK_GBS), 4), 'traffic': traffic,
                         'hbm_frac': round(achieved / HBM_PEAK_GBS, 4), 'mfma_frac': round(mfma_frac, 4),
                         'alg_ops_per_launch': round(d['ops'] / d_launches, 0),
                         'avg_launch_us': round(1e3 * d['ms'] / d_launches, 2),
                         'alg_bytes_per_launch': round(d['bytes'] / d_launches, 0),
                         'kernel_share_of_step': round(d['ms'] / total_ms, 3),
                         'mfma': mfma, 'limiter': (limiter or {}).get('limiter'), 'limiter_detail': limiter},
            'whole_net': {'sum_kernel_ms': round(total_ms, 4),
                          'mfma_int8_frac_of_peak': round(value / world * OPS_PER_IMG.get(args.arch, 0.0) / 1e12 / MFMA_I8_PEAK_TOPS, 4),
                          'hbm_frac_structural_bytes': round(value / world * STRUCT_BYTES_PER_IMG.get(args.arch, 0.0) / 1e9 / HBM_PEAK_GBS, 4),
                          'hbm_frac_algorithmic_bytes': round(value / world * sum(r[3] for r in rows) / BS / 1e9 / HBM_PEAK_GBS, 4),
                          'alg_bytes_per_img': round(sum(r[3] for r in rows) / BS, 0)},
            # per kernel symbol: launches per step, live duration (HIP events), algorithmic ops / bytes / essential vector lane-operations per step — tools/summarize_prof.py
            # divides the MFMA and vector instructions the counters saw by them (issued / algorithmic: halo recompute, tile padding; issued / essential: addressing, swaps ...)
            'per_kernel': {k: {'launches': e['launches'], 'us': round(1e3 * e['ms'], 2), 'alg_ops': round(e['ops'], 0), 'alg_bytes': round(e['bytes'], 0), 'alg_valu': round(e['valu'], 0)}
                           for k, e in by_kernel.items() if e['ms'] > 0},
            'build': {'csrc_sha256': stamp[:16]},
        }
        result.update(extra)
        result.update(extra_rq)
        if latency is not None:
            result['latency'] = latency
        if selfcheck is not None:
            result['multi_gpu'] = selfcheck
        if ext is not None:
            result['value_extended'] = round(BS * world * ext[0] / ext[1], 1)
            result['steps_extended'] = ext[0]
        if notes:
            result['notes'] = notes
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(spec, params, xs[x_last].cpu().numpy(), x_fl, logits[:BS].cpu().numpy())
            result['cpu_baseline']['c1'] = cpu_config1()
        elif world == 1:
            result['cpu_baseline'] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == '__main__':
    main()
