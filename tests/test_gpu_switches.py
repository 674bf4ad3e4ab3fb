"""Every tuning switch of DESIGN.md §9 changes the plan or the schedule, never the integers.  The switches are read once
per process, so each setting runs in its own interpreter: ResNet-50 (real fraclen table) and MobileNet-V2 at 64x64, bs 3,
against the golden-pinned CPU oracle.  (Two opt-in paths that had bit-rotted were found this way and removed.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
import numpy as np
import torch
from f8net_amd import synth, topology
from f8net_amd.net import build_net
from oracle import oracle
oracle.build()
for arch in ('resnet50', 'mobilenet_v2'):
    spec = topology.get(arch, normalize=(arch == 'resnet50'))
    fr = topology.R50_NVIDIA_FRACLENS if arch == 'resnet50' else None
    params = synth.make_params(spec, seed=321, fraclens=fr)
    x, x_fl = synth.make_input(spec, params, 3, 64, seed=5)
    want = oracle.net_forward(spec, params, x, x_fl)
    net = build_net(spec, params, max_batch=3, hw=64)
    xd = torch.from_numpy(x).cuda()
    got = net.run(xd).cpu().numpy()
    assert np.array_equal(got, want), arch
    for mode in (1, 2):                      # pipelined schedules: several runs in flight, two output buffers
        net.set_pipelined(mode)
        outs = [torch.empty_like(torch.from_numpy(want)).cuda() for _ in range(2)]
        keep = []
        for rep in range(4):
            net.run(xd, out=outs[rep & 1])
            keep.append(outs[rep & 1].clone())
        torch.cuda.synchronize()
        net.set_pipelined(False)
        assert all(np.array_equal(k.cpu().numpy(), want) for k in keep), (arch, mode)
print('OK')
'''

SWITCHES = ['F8_FUSE_BLOCKS=0', 'F8_FUSE_STAGES=0', 'F8_FUSE_STAGES=7', 'F8_FUSE_DS=0', 'F8_FUSE_DUAL=0', 'F8_FUSE_STEM=0',
            'F8_PATCH3X3=0', 'F8_SPLIT=1', 'F8_SPLIT=3', 'F8_SPLIT_STREAMS=0', 'F8_GRAPH=1', 'F8_BK128=1', 'F8_DEEP_NK=1',
            'F8_DUAL_WIDE=0', 'F8_CHUNK=0', 'F8_CHUNK=1', 'F8_CHUNK28=0', 'F8_CHUNK28=2', 'F8_BN=32', 'F8_BM=64', 'F8_RES_BN=128', 'F8_DW_DOT4=0', 'F8_STAGGER=0', 'F8_STEM_WPC=1', 'F8_FUSE_CHAIN=0', 'F8_FUSE_BCHAIN=0', 'F8_FUSE_BCHAIN=1', 'F8_STEM_ROWS=0', 'F8_DW_MMA=0', 'F8_FUSE_HEAD2=0', 'F8_REQUANT_FLOAT=1', 'F8_FUSE_TAIL=0', 'F8_FUSE_POOL=0']


@pytest.mark.parametrize('switch', SWITCHES)
def test_switch_keeps_results_bit_exact(switch):
    k, v = switch.split('=')
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    env[k] = v
    r = subprocess.run([sys.executable, '-c', CHILD], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('OK'), f'{switch}: {r.stdout[-500:]}\n{r.stderr[-1500:]}'


CHUNK_CHILD = r'''
import numpy as np
import torch
from f8net_amd import synth, topology
from f8net_amd.net import build_net
from oracle import oracle
oracle.build()
spec = topology.get('resnet50', normalize=True)
params = synth.make_params(spec, seed=77, fraclens=topology.R50_NVIDIA_FRACLENS)
x, x_fl = synth.make_input(spec, params, 9, 224, seed=3)
want = oracle.net_forward(spec, params, x, x_fl)
net = build_net(spec, params, max_batch=9, hw=224)
assert 'fused_bottleneck' in net.describe()
xd = torch.from_numpy(x).cuda()
launches = sum(net.step_launches(i, 9) for i in range(net.num_launches))
for mode in (0, 2):                          # one stream per run: chunks of 2 (8 at 14x14) images through the fused blocks
    net.set_pipelined(mode)
    outs = [torch.empty((9, 1000), dtype=torch.float32, device='cuda') for _ in range(2)]
    for rep in range(3):
        net.run(xd, out=outs[rep & 1])
    torch.cuda.synchronize()
    assert np.array_equal(outs[0].cpu().numpy(), want) and np.array_equal(outs[1].cpu().numpy(), want), mode
    y, ms = net.run_profiled(xd)
    assert np.array_equal(y.cpu().numpy(), want)
net.set_pipelined(2)
print('OK', launches, sum(net.step_launches(i, 9) for i in range(net.num_launches)), net.num_launches)
'''


def test_chunked_execution_is_bit_exact_with_ragged_chunks():
    """F8_CHUNK / F8_CHUNK28 = 2, F8_CHUNK14 = 8 on a 9-image batch at 224x224: the 56x56 / 28x28 fused blocks run as 2+2+2+2+1
    images, the 14x14 ones as 8+1 (pipelining mode 2 and the profiled pass)."""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''), F8_CHUNK='2', F8_CHUNK28='2',
               F8_CHUNK14='8', F8_FUSE_STAGES='7', F8_FUSE_CHAIN='0')
    r = subprocess.run([sys.executable, '-c', CHUNK_CHILD], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().startswith('OK'), f'{r.stdout[-500:]}\n{r.stderr[-1500:]}'
    _, split_launches, alt_launches, planned = r.stdout.split()
    assert int(alt_launches) > int(planned)          # chunks add launches where one stream runs the whole batch


# Round-2 / round-3 planning and scheduling keys at a size where their kernels ARE planned (224x224, 8 images): per-handle options,
# one process.  Each entry: options of one handle; results must equal the oracle bit for bit and the plan must change as stated.
OPTION_SETS = [
    ({}, 'stage_chain_x6_tail'), ({'fuse_tail': 0}, 'stage_chain_x5'),      # the joins of the stride-2 opening blocks open their stages' chain launches / the round-3 plan
    ({}, 'stage_chain_x3_tail+avgpool'), ({'fuse_chain7': 0}, 'fused_p12:'), ({'fuse_chain7': 0, 'requant_float': 1}, 'conv1x1_res+avgpool'),      # round 6: the 7x7 stage on the cluster kernel (f8_cchain.hip) / as rounds 3 - 5 ran it
    ({'fuse_tail': 0, 'fuse_pool': 0}, 'stage_chain_x2:'),      # ... its identity-first form (the stream comes in as an int32 tensor) with the pool as a launch of its own
    ({'fuse_chain': 0}, 'fused_bottleneck_R'),
    ({'fuse_chain': 0, 'fuse_stages': 7}, 'fused_bottleneck_R7'),
    ({'fuse_chain': 0, 'fuse_ds': 0}, '_dual:'),
    ({'wstat': 0}, None), ({'wstat_min_tiles': 0}, 'wstat'), ({'s2wreg': 0}, None), ({'wreg': 0}, None), ({'fuse_p12': 0}, None),
    ({'fuse_opener': 0}, None), ({'opener_stg': 0}, 'fused_opener_s2'), ({'fuse_fc': 0}, 'output'), ({'fuse_input': 0}, None),
    ({'wstat_fast': 0}, None), ({'split': 3, 'pipeline_depth': 3}, None),
    ({'arena_copies': 3, 'pipeline_depth': 3}, None), ({'arena_copies': 4, 'pipeline_depth': 4}, None),      # bench.py's schedule: whole batches in flight, split stays 2
    ({'shared_streams': 0}, None),
    ({'fuse_pool': 0}, 'avgpool_sum'),      # the last join and the average pool as two launches (default: one, conv1x1_res+avgpool)
    ({'requant_float': 1}, None), ({'requant_float': 1, 'fuse_chain': 0}, None),      # float-converter requantisation where the planner bounds the value (chains; per-block launches); default: integer everywhere
    ({'stem_rows': 0}, None), ({'stem_rows': 0, 'fuse_input': 0}, None),      # the tile kernel of the head (f8_stem.hip), raw input / haloed form
]


def test_options_keep_results_bit_exact_at_full_resolution():
    import numpy as np
    import torch
    from f8net_amd import synth, topology
    from f8net_amd.net import build_net
    from oracle import oracle
    oracle.build()
    spec = topology.get('resnet50', normalize=True)
    params = synth.make_params(spec, seed=99, fraclens=topology.R50_NVIDIA_FRACLENS)
    x, x_fl = synth.make_input(spec, params, 8, 224, seed=6)
    want = oracle.net_forward(spec, params, x, x_fl)
    xd = torch.from_numpy(x).cuda()
    plans = set()
    for opts, marker in OPTION_SETS:
        net = build_net(spec, params, max_batch=8, hw=224, options=dict(opts, whole_batch_launches=1))
        plan = net.describe()
        plans.add(plan)
        if marker:
            assert marker in plan, (opts, plan)
        got = net.run(xd).cpu().numpy()
        net.check()
        assert np.array_equal(got, want), opts
        net.set_pipelined(2)
        nb = max(3, opts.get('pipeline_depth', 2) + 1)
        outs = [torch.empty((8, 1000), dtype=torch.float32, device='cuda') for _ in range(nb)]
        for rep in range(2 * nb + 1):
            net.run(xd, out=outs[rep % nb])
        torch.cuda.synchronize()
        net.check()
        assert all(np.array_equal(o.cpu().numpy(), want) for o in outs), opts
    assert len(plans) >= 9          # the keys really change the plan at this size
    mb = topology.get('mobilenet_v2')
    pm = synth.reference_params(mb, seed=98)          # the reference log's learned fraclens
    xm, xm_fl = synth.make_input(mb, pm, 8, 224, seed=6)
    wantm = oracle.net_forward(mb, pm, xm, xm_fl)
    for fi in (0, 1, 2):
        net = build_net(mb, pm, max_batch=8, hw=224, options={'fuse_ir': fi})
        assert np.array_equal(net.run(torch.from_numpy(xm).cuda()).cpu().numpy(), wantm), fi
