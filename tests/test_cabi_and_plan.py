"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol the header
declares; graph validation mirrors the reference's asserts; the planner fuses what DESIGN.md says."""
import ctypes
import os
import re

import numpy as np
import pytest

from f8net_amd import _lib, synth, topology
from f8net_amd.net import F8Net, build_net

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'f8net.h')).read()
    declared = set(re.findall(r'\b(f8_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'f8_status'}
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f'libf8net.so does not export {name}'
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, f'binding and header disagree: {declared ^ bound}'
    assert _lib.lib().f8_version() >= 100
    assert _lib.lib().f8_status_string(-1) == b'invalid argument'


def test_no_torch_types_in_abi():
    hdr = open(os.path.join(ROOT, 'include', 'f8net.h')).read()
    code = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)          # strip comments
    assert 'torch' not in code.lower() and 'at::' not in code and '#include <hip' not in code


def test_builder_rejects_what_the_reference_asserts():
    net = F8Net()
    t = net.input(64, 8, 8, 10)
    w = np.zeros((64, 64, 1, 1), np.int32)
    with pytest.raises(_lib.F8Error):      # signed fl must be <= 7 (fix_quant_ops.py:93-94)
        net.conv(t, w, None, stride=1, pad=0, groups=1, weight_fl=7, input_fl=8, input_signed=True)
    with pytest.raises(_lib.F8Error):      # unsigned fl must be <= 8
        net.conv(t, w, None, stride=1, pad=0, groups=1, weight_fl=7, input_fl=9, input_signed=False)
    w_bad = w.copy()
    w_bad[0, 0, 0, 0] = 300                # not an 8-bit weight
    with pytest.raises(_lib.F8Error):
        net.conv(t, w_bad, None, stride=1, pad=0, groups=1, weight_fl=7, input_fl=4, input_signed=False)
    with pytest.raises(_lib.F8Error):      # groups other than 1 / depthwise
        net.conv(t, np.zeros((64, 32, 3, 3), np.int32), None, stride=1, pad=1, groups=2, weight_fl=7,
                 input_fl=4, input_signed=False)
    with pytest.raises(_lib.F8Error):      # cin mismatch
        net.conv(t, np.zeros((64, 32, 1, 1), np.int32), None, stride=1, pad=0, groups=1, weight_fl=7,
                 input_fl=4, input_signed=False)
    a = net.conv(t, w, None, stride=1, pad=0, groups=1, weight_fl=7, input_fl=4, input_signed=False)
    with pytest.raises(_lib.F8Error):      # avgpool fraclen > 32 (fix_quant_ops.py:129)
        net.avgpool_sum(a, 30)
    with pytest.raises(_lib.F8Error):
        net.finalize(4)                    # no output marked
    net.output(a, as_float=False)
    net.finalize(4)
    with pytest.raises(_lib.F8Error):
        net.finalize(4)                    # twice


def test_run_without_gpu_fails_loudly():
    """No CPU fallback: with no device the run call must raise, not compute on the host."""
    torch = pytest.importorskip('torch')
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    spec = topology.get('resnet18')
    net = build_net(spec, synth.make_params(spec, 1), max_batch=1, hw=32)
    x = torch.zeros((1, 3, 32, 32), dtype=torch.int32)
    with pytest.raises(ValueError):
        net.run(x)
    from f8net_amd import ops
    with pytest.raises(ValueError):
        ops.int_op_only_fix_quant(x, 8, 4, 6, True)
    assert _lib.lib().f8_device_count() == 0
    rc = _lib.lib().f8_net_upload(net._h)
    assert rc < 0 and b'hip' in _lib.lib().f8_last_error().lower()


@pytest.mark.parametrize('arch,launches,fused,dual', [('resnet18', 23, 0, 0), ('resnet50', 37, 5, 3), ('mobilenet_v1', 30, 0, 0),
                                                       ('mobilenet_v2', 28, 0, 0)])
def test_plan_fuses_requant_relu_residual(arch, launches, fused, dual):
    spec = topology.get(arch)
    # one launch per block (the stage-chain launches of f8_chain.hip / f8_bchain.hip have their own plan tests below)
    net = build_net(spec, synth.make_params(spec, 1), max_batch=8, hw=224, options={'fuse_chain': 0, 'fuse_bchain': 0})
    plan = net.describe()
    assert net.num_launches == launches, plan
    # MobileNet-V2: head conv + depthwise + 1x1 of stage 0 are one row-walking launch that reads the caller's buffer itself (f8_stem.hip, H2)
    assert ('head3x3s2+dw3x3+1x1' in plan) == (arch == 'mobilenet_v2')
    # ResNets: the head (stem conv + max-pool) is one launch, whatever forms (int32 / int8) the pool output needs
    assert ('stem7x7s2+maxpool3x3s2' in plan) == arch.startswith('resnet')
    # no stand-alone add / requant launches: every residual join rides in a conv epilogue
    assert 'add:' not in plan and 'requant:' not in plan
    # bottleneck identity blocks of stages 0-1 run as ONE launch each (1x1 -> 3x3 -> 1x1 + residual); stage 2 joins them
    # only when a launch fills the chip (max_batch 8 here: it does not — see the bs-128 plan below)
    assert plan.count('fused_bottleneck') == fused
    # bottleneck downsample blocks: body.4 and the shortcut conv are ONE dual-GEMM launch (no int32 tensor between them)
    assert plan.count('_dual:') == dual
    n_res_blocks = sum(1 for b in spec.blocks if b.residual)
    # the stage-1 opening block (1x1 -> 3x3 / 2 -> [1x1 + strided shortcut]) is ONE launch when body.0 and the shortcut read the
    # same int8 form of the block input (f8_opener.hip)
    opener = plan.count('fused_opener_s2')
    assert opener == (1 if arch == 'resnet50' else 0)
    # MobileNet-V2: every inverted-residual block (expand -> depthwise -> project [+ residual]) is ONE launch (f8_ir.hip)
    ir = [l for l in plan.splitlines() if 'fused_ir_' in l]
    assert len(ir) == (12 if arch == 'mobilenet_v2' else 0)     # the blocks where the fused launch wins: all but the 7x7 ones (option fuse_ir = 2: all 16)
    # (the network's last 1x1 conv runs with the average pool behind it in one launch, f8_pool.hip: `conv1x1_res+avgpool` / `conv1x1+avgpool`)
    assert ('+avgpool:' in plan) == (arch in ('resnet50', 'mobilenet_v2')) and ('avgpool_sum:' in plan) == (arch in ('resnet18', 'mobilenet_v1'))
    assert plan.count('_res:') + plan.count('_res+avgpool:') + plan.count('_dual:') + fused + opener + sum('res=1' in l for l in ir) == n_res_blocks
    # the 7x7 identity blocks of ResNet-50: body.0 + body.2 are one launch (f8_p12.hip), the residual-carrying 1x1 stays
    assert plan.count('fused_p12:') == (2 if arch == 'resnet50' else 0)
    # the classifier writes the caller's logits buffer itself (f8_fc.hip): no output launch
    assert 'linear_dense:classifier.0' in plan and 'output:' not in plan
    assert net.weight_bytes > 0 and net.arena_bytes > 0


def test_plan_runs_each_stage_as_one_chain_launch():
    """Default ResNet-50 plan: the bottleneck blocks of stages 0-2 that follow one another at one resolution are ONE launch per
    stage (f8_chain.hip): the int32 residual stream exists in HBM only where a chain starts with an identity block (its input),
    never between blocks, and a chain writes int8 forms only when nothing downstream needs 32 bits."""
    spec = topology.get('resnet50', normalize=True)
    for mb in (128, 4):
        net = build_net(spec, synth.make_params(spec, 1234, topology.R50_NVIDIA_FRACLENS), max_batch=mb, hw=224)
        lines = net.describe().splitlines()
        chains = [l for l in lines if 'stage_chain_x' in l]
        # stage 0: opened by its (same-resolution) opening block; stages 1 and 2: by the JOIN of their stride-2 opening block (round 4: TAIL)
        # ... and stage 3 (round 6, f8_cchain.hip: clusters of eight workgroups) with the average pool behind its last block
        assert [l.split()[1].split(':')[0] for l in chains] == ['stage_chain_x3_ds', 'stage_chain_x4_tail', 'stage_chain_x6_tail', 'stage_chain_x3_tail+avgpool'], net.describe()
        assert all('i32=0' in l and 'i8=1' in l for l in chains)          # the next stage's opening block (the classifier) reads int8 only
        assert not any('fused_bottleneck' in l for l in lines)
        assert 'stage_0_layer_0.body.0..stage_0_layer_2.body.4' in chains[0] and 'stage_2_layer_0.body.4..stage_2_layer_5.body.4' in chains[2]
        assert 'stage_3_layer_0.body.4..stage_3_layer_2.body.4' in chains[3] and not any('avgpool_sum' in l or '_dual:' in l or 'fused_p12:' in l for l in lines)
        # body.0 + body.2 of the stage-1 opener are one launch that writes mid2 as int8; no int32 tensor exists between a stage's blocks or in front of its chain
        opener = [l for l in lines if 'fused_opener_s2' in l]
        assert len(opener) == 1 and 'fused_opener_s2_p12' in opener[0] and 'i32=0 i8=1' in opener[0]
        assert sum('i32=1' in l for l in lines) == 0                      # no int32 tensor is left in HBM anywhere in the network
        # option fuse_chain7 = 0: the 7x7 stage as rounds 3 - 5 ran it — dual-GEMM join, fused_p12 + residual-carrying 1x1 per identity block, last join + pool
        old = build_net(spec, synth.make_params(spec, 1234, topology.R50_NVIDIA_FRACLENS), max_batch=mb, hw=224, options={'fuse_chain7': 0}).describe().splitlines()
        assert [l.split()[1].split(':')[0] for l in old if 'stage_chain_x' in l] == ['stage_chain_x3_ds', 'stage_chain_x4_tail', 'stage_chain_x6_tail']
        assert sum('fused_p12:' in l for l in old) == 2 and sum('_dual:' in l for l in old) == 1 and sum('_res+avgpool:' in l for l in old) == 1
        assert sum('i32=1' in l for l in old) == 2                        # stage 3: its opener's join and the first 7x7 identity join (the last one is pooled in its launch)
    assert net.num_launches <= 12
    off = build_net(spec, synth.make_params(spec, 1234, topology.R50_NVIDIA_FRACLENS), max_batch=128, hw=224, options={'fuse_tail': 0}).describe()
    assert 'stage_chain_x3:' in off and 'stage_chain_x5:' in off and '_tail' not in off and '_p12_R' not in off      # the round-3 plan
    # ResNet-18 / MobileNets have no bottleneck blocks: nothing changes for them
    for arch in ('resnet18', 'mobilenet_v2'):
        sp = topology.get(arch)
        assert 'stage_chain' not in build_net(sp, synth.make_params(sp, 1), max_batch=8, hw=224).describe()


def test_plan_runs_basic_block_stages_as_chain_launches():
    """Default ResNet-18 plan: the BasicBlocks of a stage are ONE launch (f8_bchain.hip: every 3x3 of every block, the int32 stream in
    registers): stage 0 (56x56 x 64) its two identity blocks, stages 1-2 (28x28 x 128, 14x14 x 256) the stage-opening block (3x3 / 2,
    3x3, 1x1 / 2 shortcut) and the identity block behind it; the 7x7 stage keeps its per-conv launches.  fuse_bchain = 1 leaves the
    opening blocks out, fuse_bchain = 0 gives the per-conv plan back."""
    spec = topology.get('resnet18')
    params = synth.make_params(spec, 1)
    net = build_net(spec, params, max_batch=8, hw=224)
    lines = net.describe().splitlines()
    chains = [l.split()[1] for l in lines if 'basic_chain_x' in l]
    assert chains == ['basic_chain_x2:stage_0_layer_0.body.0..stage_0_layer_1.body.2',
                      'basic_chain_x2_ds:stage_1_layer_0.body.0..stage_1_layer_1.body.2',
                      'basic_chain_x2_ds:stage_2_layer_0.body.0..stage_2_layer_1.body.2'], net.describe()
    assert net.num_launches == 12
    # the stem hands the first chain the int32 stream only: the chain makes its own int8 copy
    stem = [l for l in lines if 'stem7x7s2+maxpool3x3s2' in l]
    assert len(stem) == 1 and 'i32=1 i8=0' in stem[0]
    # a chain hands the next stage's opening block int8 forms only (its 3x3 / 2 and its shortcut may ask for different formats)
    assert all('i32=0' in l for l in lines if 'basic_chain_x2:' in l or 'stage_1_layer_0' in l)
    # 7x7 stage: not chained (49 pixels per image do not fill a tile)
    assert any('stage_3_layer_1.body.2' in l and '_res:' in l for l in lines)
    assert any('stage_3_layer_0.shortcut.0' in l and '_res:' in l for l in lines)
    mid = build_net(spec, params, max_batch=8, hw=224, options={'fuse_bchain': 1})
    assert [l.split()[1].split(':')[0] for l in mid.describe().splitlines() if 'basic_chain_x' in l] == ['basic_chain_x2', 'basic_chain_x1', 'basic_chain_x1']
    assert mid.num_launches == 18
    off = build_net(spec, params, max_batch=8, hw=224, options={'fuse_bchain': 0})
    assert 'basic_chain' not in off.describe() and off.num_launches == 23
    # a resolution the kernel has no instance for: per-conv plan
    assert 'basic_chain' not in build_net(spec, params, max_batch=8, hw=96).describe()


def test_plan_keeps_int32_only_where_semantics_need_it():
    spec = topology.get('resnet50', normalize=True)
    net = build_net(spec, synth.make_params(spec, 1234, topology.R50_NVIDIA_FRACLENS), max_batch=128, hw=224, options={'fuse_chain': 0, 'fuse_pool': 0})
    lines = net.describe().splitlines()
    body0 = [l for l in lines if '.body.0 ' in l or '.body.2 ' in l]
    assert body0 and all('i32=0' in l for l in body0)          # inside a block: int8 only
    res = [l for l in lines if '_res:' in l or '_dual:' in l or 'fused_bottleneck' in l or 'fused_opener' in l]
    assert len(res) == 16
    # the int32 residual stream exists only where an identity block follows (a downsample block's convs read int8)
    assert sum('i32=0' in l for l in res) == 3
    # downsample blocks: no int32 tensor between body.4 and the shortcut conv (one dual-GEMM launch)
    import re
    assert not any(re.search(r'conv1x1s1_t\d+x\d+x\d+:stage_\d_layer_0\.body\.4 ', l) for l in lines)
    # the stage-0 opening block (body.0 and shortcut.0 share one int8 form of the block input in the real fraclen table)
    # is ONE launch: 1x1 -> 3x3 -> [1x1 + shortcut 1x1] + join
    assert sum('fused_bottleneck_ds' in l for l in lines) == 1 and net.num_launches == 26
    # ... and so is the stage-1 opening block with its stride-2 3x3 (the 56x56x128 intermediate never exists in HBM)
    assert sum('fused_opener_s2' in l for l in lines) == 1
    # the five 14x14 identity blocks are fused too at this batch (64 images per launch = 128 workgroups), not at bs 32
    assert sum('fused_bottleneck' in l and 'stage_2' in l for l in lines) == 5
    small = build_net(spec, synth.make_params(spec, 1234, topology.R50_NVIDIA_FRACLENS), max_batch=32, hw=224, options={'fuse_chain': 0, 'fuse_pool': 0})
    assert small.num_launches == 36 and not any('fused_bottleneck' in l and 'stage_2' in l for l in small.describe().splitlines())
    # the 1x1 convs around the stage-2 / stage-3 opening blocks and the closing 1x1 of the 7x7 blocks run weight-stationary (f8_wstat.hip)
    # when a launch gives every workgroup at least two pixel tiles; smaller launches keep the tile-per-workgroup kernels
    assert [l.split()[1].split(':')[0] for l in lines if 'wstat' in l] == ['conv1x1_wstat', 'conv1x1_wstat_dual', 'conv1x1_wstat', 'conv1x1_wstat_dual',
                                                                          'conv1x1_wstat_res', 'conv1x1_wstat_res']
    assert sum('wstat' in l for l in small.describe().splitlines()) < 6
    # the head is ONE launch: stem conv + ReLU + requant + max-pool (requant commutes with max; the 112x112 map stays in LDS)
    assert any('stem7x7s2+maxpool3x3s2' in l for l in lines) and not any('maxpool_i' in l for l in lines)
    # algorithmic bytes are reported per launch and sum to less than the structural model
    total = sum(net.launch_info(i, 128)[1] for i in range(net.num_launches)) / 128
    assert 40e6 < total < 93.444e6


def test_two_consumer_formats_and_fallback():
    """A tensor read by convs with different input_fraclen gets one int8 form per format; a third
    format falls back to an int32 form + stand-alone requant."""
    net = F8Net()
    t = net.input(32, 8, 8, 8)
    w = np.ones((32, 32, 1, 1), np.int32)
    a = net.conv(t, w, None, stride=1, pad=0, groups=1, weight_fl=5, input_fl=8, input_signed=False, quant_input=False)
    outs = [net.conv(a, w, None, stride=1, pad=0, groups=1, weight_fl=5, input_fl=fl, input_signed=False) for fl in (3, 4, 5)]
    s = net.add(outs[0], outs[1])
    s = net.add(s, outs[2])
    net.output(s, as_float=False)
    net.finalize(2)
    plan = net.describe()
    assert plan.count('requant:') == 1, plan


def test_smoke_plans_hold_the_kernels_smoke_asserts():
    """__graft_entry__.smoke() asserts that its small nets run the round's kernels; the same plans, checked without a GPU (a plan
    change that silently drops one of them would otherwise only fail on the GPU box)."""
    r50 = topology.get('resnet50', normalize=True)
    plan = build_net(r50, synth.make_params(r50, seed=3, fraclens=topology.R50_NVIDIA_FRACLENS), max_batch=2, hw=224).describe()
    assert all(k in plan for k in ('stem7x7s2+maxpool3x3s2', 'stage_chain_x3_ds', 'stage_chain_x6_tail', 'fused_opener_s2_p12', 'stage_chain_x3_tail+avgpool')), plan
    r18 = topology.get('resnet18')
    plan = build_net(r18, synth.make_params(r18, seed=3), max_batch=2, hw=224).describe()
    assert 'basic_chain_x2_ds' in plan and 'patch' in plan, plan


def test_intmodel_parameter_fingerprint_sees_rebound_storage():
    """`param.data = new_tensor` (how the reference installs integer weights, fix_quant_ops.py:705-706) bumps no `_version` counter: the
    fingerprint IntModel.forward compares before it reuses a plan must change anyway, and so must an in-place edit."""
    import torch
    from f8net_amd import int_model, synth, topology
    spec = topology.get('resnet18', num_classes=8)
    m = int_model.from_params(spec, synth.make_params(spec, seed=2))
    v0 = m._param_version()
    assert m._param_version() == v0
    w = m.head[0].weight
    w.data = w.data.clone()                        # same values, new storage, same version counter
    v1 = m._param_version()
    assert v1 != v0
    with torch.no_grad():
        w[0, 0, 0, 0] += 1                         # in-place edit: version counter
    assert m._param_version() != v1
    # two parameters of one shape swapping their storages: no version moves, the SET of pointers is the same (an XOR fold missed it)
    a, b = m.stage_0_layer_0.body[0].weight, m.stage_0_layer_0.body[2].weight
    assert a.shape == b.shape
    v2 = m._param_version()
    a.data, b.data = b.data, a.data
    assert m._param_version() != v2


def test_tail_chain_is_planned_only_for_an_input_map_of_exactly_twice_the_output():
    """chain_kernel<TAIL> addresses the stride-2 shortcut's operand as [N][2H][2W][CIN0]: a 55x55 stage-0 map (hw 217..220) also
    gives a 28x28 stage 1, and a 27x27 one (hw 209..216) a 14x14 stage 2 — those joins must stay on the generic dual GEMM."""
    r50 = topology.get('resnet50', normalize=True)
    p = synth.make_params(r50, seed=3, fraclens=topology.R50_NVIDIA_FRACLENS)
    assert build_net(r50, p, max_batch=2, hw=224).describe().count('_tail:') == 2
    plan = build_net(r50, p, max_batch=2, hw=220).describe()      # 55 -> 28 -> 14: stage 1's join stays generic, stage 2's (28 = 2 x 14) is a TAIL
    assert plan.count('_tail:') == 1 and '_tail:stage_2_layer_0' in plan and '_dual:stage_1_layer_0' in plan, plan
    plan = build_net(r50, p, max_batch=2, hw=212).describe()      # 53 -> 27 -> 14
    assert '_tail:' not in plan, plan


def test_every_planned_kernel_name_is_a_symbol_of_the_library():
    """`launch_kernel(i)` is the device symbol as rocprofv3 prints it: bench.py joins its live timings with the stamped counter files of profiles/ on that
    string.  A template parameter added to a kernel without its name string following (round 4: bchain_kernel's wave count) silently detaches the
    counters from the bench line; checked here, without a GPU, against the library's own symbol table for the four nets' default plans."""
    import shutil
    import subprocess
    nm = shutil.which('nm')
    if nm is None:
        pytest.skip('needs binutils nm')
    syms = subprocess.run([nm, '-C', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    have = set(re.findall(r'(?:void )?(f8::[A-Za-z0-9_]+(?:<[^()]*>)?)\(', syms))
    assert len(have) > 50, 'no kernel symbols found'
    for arch, kw in (('resnet50', dict(normalize=True)), ('resnet18', {}), ('mobilenet_v2', {}), ('mobilenet_v1', {})):
        spec = topology.get(arch, **kw)
        for opts in ({}, {'requant_float': 1}):
            net = build_net(spec, synth.reference_params(spec), max_batch=128, hw=224, options=opts)
            for i in range(net.num_launches):
                k = net.launch_kernel(i)
                if not k:
                    continue
                # a name without template arguments stands for a family whose instance is picked at run time (input / output kernels)
                ok = k in have if '<' in k else any(h == k or h.startswith(k + '<') for h in have)
                assert ok, f'{arch} {opts}: launch {i} ({net.launch_info(i, 128)[0]}) names {k!r}, which the library does not define'


def test_planner_bounds_the_residual_stream_before_it_plans_the_float_requantisation():
    """f8_net.cpp tensor_amax: the int32 stream of a chain launch is the shifted sum of bounded conv accumulators; the float-converter instance
    (template argument 1; option requant_float = 1) is planned only while that bound (and every accumulator's) stays below 2^31 - 2^16 — a bias that
    lifts ONE stream channel next to 2^31 plans the integer instance (2) for that launch alone; the default (requant_float = 0) plans it for every launch."""
    def chains(net):     # (the 56x56 / 28x28 / 14x14 launches; the 7x7 cluster chain, f8::cchain_kernel<instance>, below)
        return [net.launch_kernel(i) for i in range(net.num_launches) if '::chain_kernel<' in net.launch_kernel(i) or 'bchain_kernel<' in net.launch_kernel(i)]
    def cchain(net):
        return [net.launch_kernel(i) for i in range(net.num_launches) if 'cchain_kernel<' in net.launch_kernel(i)]
    r50 = topology.get('resnet50', normalize=True)
    p = synth.reference_params(r50)
    fl = {'requant_float': 1}
    ks = chains(build_net(r50, p, max_batch=8, hw=224, options=fl))
    assert len(ks) == 3 and all(re.search(r', 1, false, (false|true), [48]>$', k) for k in ks), ks
    ks = chains(build_net(r50, p, max_batch=8, hw=224))
    assert len(ks) == 3 and all(re.search(r', 2, false, (false|true), [48]>$', k) for k in ks), ks
    # (the 7x7 cluster chain has no float-converter instance: the integer one is exact wherever the float form is)
    assert cchain(build_net(r50, p, max_batch=8, hw=224, options=fl)) == ['f8::cchain_kernel<2>'] and cchain(build_net(r50, p, max_batch=8, hw=224)) == ['f8::cchain_kernel<2>']
    q = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in p.items()}
    q['stage_1_layer_2.body.4.bias'][5] = 2 ** 31 - 2 ** 18          # body.4 feeds only the stream: no accumulator that is requantised grows
    ks = chains(build_net(r50, q, max_batch=8, hw=224, options=fl))
    assert [re.search(r', (\d), false, (false|true), [48]>$', k).group(1) for k in ks] == ['1', '2', '1'], ks
    r18 = topology.get('resnet18')
    ks = chains(build_net(r18, synth.reference_params(r18), max_batch=8, hw=224, options=fl))
    assert len(ks) == 3 and all(re.search(r', 1, (false|true), 8>$', k) for k in ks), ks       # stage 0 reads the max-pooled head output: bounded through the pool node


def test_essential_vector_work_per_launch_follows_the_reference_semantics():
    """f8_net_launch_valu (VERDICT r5 #1b): 3 lane-operations per int8 value a launch produces (in HBM or only in LDS), 2 per joined int32 value, 1 per
    max-pooled conv value — checked by hand for the stage-0 chain and the head of ResNet-50 and as a whole-net total; scales with N."""
    from f8net_amd import synth, topology
    from f8net_amd.net import build_net
    spec = topology.get('resnet50', normalize=True)
    net = build_net(spec, synth.reference_params(spec, seed=1234), max_batch=128, hw=224, options={'whole_batch_launches': 1})
    by = {net.launch_info(i, 1)[0].split(':')[0]: net.launch_valu(i, 1) for i in range(net.num_launches)}
    px = 56 * 56
    # opening block + 2 identity blocks at 56x56: per block body.0 / body.2 int8 outputs (64 + 64), the join (256 x 2), the next block's int8 input (256 x 3; last block: one int8 output form)
    assert by['stage_chain_x3_ds'] == px * (3 * 2 * 64 * 3 + 3 * 256 * 2 + 2 * 256 * 3 + 3 * 256)
    # head: one max per conv value (112 x 112 x 64), the requantisation on the pooled values
    assert by['stem7x7s2+maxpool3x3s2'] == 112 * 112 * 64 + 3 * px * 64
    total = sum(net.launch_valu(i, 1) for i in range(net.num_launches))
    assert 30e6 < total < 45e6                                   # ~ 37 M per image (5 1/4 per stream value + 3 per inner value)
    assert net.launch_valu(3, 128) == 128 * net.launch_valu(3, 1)
    assert net.get_option('err_mirror') == 0                     # read-only state key: no mirror before the upload
    with pytest.raises(Exception):
        net.set_option('err_mirror', 1)
