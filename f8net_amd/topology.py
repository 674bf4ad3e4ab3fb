"""Topology tables of the exported integer networks (data, not code, of the reference).

The reference builds its integer nets by walking QAT modules
(`/root/reference/models/fix_resnet.py:413-544`, `fix_mobilenet_v2.py:276-423`,
`fix_mobilenet_v1.py:171-281`).  The int path only needs the *shape* of that
graph: which conv sits under which state_dict key, its geometry, whether its
8-bit input is signed, and how blocks join.  This module states that shape as
plain tables so that the oracle, the HIP executor and the tests agree on it.

Key naming follows the reference's exported `IntModel.state_dict()`:
`head.0`, `stage_{i}_layer_{j}.body.{0,2,4}` (ReLUs occupy odd indices),
`stage_{i}_layer_{j}.shortcut.0`, `tail.0` (MobileNet-V2), `classifier.0`.
"""
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class ConvSpec:
    key: str            # state_dict prefix
    cin: int
    cout: int
    k: int
    stride: int
    pad: int
    groups: int = 1
    signed_in: bool = False   # `input_symmetric` of the reference layer
    relu: bool = False        # an nn.ReLU directly follows this conv in the int graph


@dataclass
class BlockSpec:
    name: str
    body: List[ConvSpec]
    shortcut: Optional[ConvSpec] = None   # ResNet downsample conv
    residual: bool = False                # int32 align-add with the block input
    post_relu: bool = False               # ResNet: ReLU after the add


@dataclass
class NetSpec:
    arch: str
    head: ConvSpec
    head_maxpool: bool
    blocks: List[BlockSpec]
    tail: Optional[ConvSpec]
    fc_key: str
    fc_in: int
    num_classes: int
    fc_signed_in: bool = False
    normalize: bool = False   # head input signed (fix_train.py:683-687) vs u8 (fix_train.py:689-692)

    def convs(self) -> List[ConvSpec]:
        out = [self.head]
        for b in self.blocks:
            out.extend(b.body)
            if b.shortcut is not None:
                out.append(b.shortcut)
        if self.tail is not None:
            out.append(self.tail)
        return out

    def layer_keys(self) -> List[str]:
        return [c.key for c in self.convs()] + [self.fc_key]


_RESNET_BLOCKS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3],
                  101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


def resnet(depth: int = 50, num_classes: int = 1000, normalize: bool = False) -> NetSpec:
    """ResNet-v1.5 ("b": stride on the 3x3), fix_resnet.py:413-486 / int_block :207-221,:308-319."""
    bottleneck = depth >= 50
    exp = 4 if bottleneck else 1
    head = ConvSpec('head.0', 3, 64, 7, 2, 3, signed_in=normalize, relu=True)
    blocks = []
    ch = 64
    for si, n in enumerate(_RESNET_BLOCKS[depth]):
        outp = [64, 128, 256, 512][si] * exp
        for li in range(n):
            stride = 2 if (li == 0 and si != 0) else 1
            name = f'stage_{si}_layer_{li}'
            if bottleneck:
                mid = outp // 4
                body = [ConvSpec(f'{name}.body.0', ch, mid, 1, 1, 0, relu=True),
                        ConvSpec(f'{name}.body.2', mid, mid, 3, stride, 1, relu=True),
                        ConvSpec(f'{name}.body.4', mid, outp, 1, 1, 0)]
            else:
                body = [ConvSpec(f'{name}.body.0', ch, outp, 3, stride, 1, relu=True),
                        ConvSpec(f'{name}.body.2', outp, outp, 3, 1, 1)]
            identity = (stride == 1 and ch == outp)
            sc = None if identity else ConvSpec(f'{name}.shortcut.0', ch, outp, 1, stride, 0)
            blocks.append(BlockSpec(name, body, sc, residual=True, post_relu=True))
            ch = outp
    return NetSpec(f'resnet{depth}', head, True, blocks, None, 'classifier.0', ch,
                   num_classes, normalize=normalize)


def mobilenet_v2(num_classes: int = 1000, normalize: bool = False) -> NetSpec:
    """fix_mobilenet_v2.py:276-372 (block table :282-291), int_block :168-176."""
    setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2],
               [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]
    head = ConvSpec('head.0', 3, 32, 3, 2, 1, signed_in=normalize, relu=True)
    blocks = []
    ch = 32
    for si, (t, c, n, s) in enumerate(setting):
        for li in range(n):
            stride = s if li == 0 else 1
            # first conv of every block takes a signed input except in stage 0 (:311-331)
            ds = not (si == 0 and li == 0)
            name = f'stage_{si}_layer_{li}'
            e = ch * t
            if t != 1:
                body = [ConvSpec(f'{name}.body.0', ch, e, 1, 1, 0, signed_in=ds, relu=True),
                        ConvSpec(f'{name}.body.2', e, e, 3, stride, 1, groups=e, relu=True),
                        ConvSpec(f'{name}.body.4', e, c, 1, 1, 0)]
            else:
                body = [ConvSpec(f'{name}.body.0', e, e, 3, stride, 1, groups=e,
                                 signed_in=ds, relu=True),
                        ConvSpec(f'{name}.body.2', e, c, 1, 1, 0)]
            blocks.append(BlockSpec(name, body, None,
                                    residual=(stride == 1 and ch == c), post_relu=False))
            ch = c
    tail = ConvSpec('tail.0', ch, 1280, 1, 1, 0, signed_in=True, relu=True)
    return NetSpec('mobilenet_v2', head, False, blocks, tail, 'classifier.0', 1280,
                   num_classes, normalize=normalize)


def mobilenet_v1(num_classes: int = 1000, normalize: bool = False) -> NetSpec:
    """fix_mobilenet_v1.py:171-236, int_block :82-92."""
    setting = [[64, 1, 1], [128, 2, 2], [256, 2, 2], [512, 6, 2], [1024, 2, 2]]
    head = ConvSpec('head.0', 3, 32, 3, 2, 1, signed_in=normalize, relu=True)
    blocks = []
    ch = 32
    for si, (c, n, s) in enumerate(setting):
        for li in range(n):
            stride = s if li == 0 else 1
            name = f'stage_{si}_layer_{li}'
            body = [ConvSpec(f'{name}.body.0', ch, ch, 3, stride, 1, groups=ch, relu=True),
                    ConvSpec(f'{name}.body.2', ch, c, 1, 1, 0, relu=True)]
            blocks.append(BlockSpec(name, body, None, residual=False, post_relu=False))
            ch = c
    return NetSpec('mobilenet_v1', head, False, blocks, None, 'classifier.0', ch,
                   num_classes, normalize=normalize)


def get(arch: str, num_classes: int = 1000, normalize: bool = False) -> NetSpec:
    if arch.startswith('resnet'):
        return resnet(int(arch[len('resnet'):]), num_classes, normalize)
    if arch in ('mobilenet_v2', 'mbv2'):
        return mobilenet_v2(num_classes, normalize)
    if arch in ('mobilenet_v1', 'mbv1'):
        return mobilenet_v1(num_classes, normalize)
    raise ValueError(f'unknown arch {arch!r}')


# Learned fraction lengths of the reference's NVIDIA-pretrained ResNet-50 run, as printed in its
# committed log (`/root/reference/fraclen_visual/res50_fix_quant_nvidia_pretrained.out:492-1138`;
# SURVEY.md App. E).  (input_fl, weight_fl) per exported layer key.  Used as the bench config.
R50_NVIDIA_FRACLENS = {}
_r50_raw = """head.0:5/5 s0l0b0:3/7 s0l0b1:4/7 s0l0b2:4/7 s0l0sc0:3/7 s0l1b0:5/5 s0l1b1:4/7 s0l1b2:4/7
s0l2b0:5/6 s0l2b1:4/7 s0l2b2:4/7 s1l0b0:5/6 s1l0b1:4/7 s1l0b2:4/6 s1l0sc0:5/7 s1l1b0:5/6 s1l1b1:4/7
s1l1b2:3/5 s1l2b0:4/7 s1l2b1:4/7 s1l2b2:2/7 s1l3b0:4/7 s1l3b1:3/7 s1l3b2:2/7 s2l0b0:4/6 s2l0b1:3/7
s2l0b2:4/7 s2l0sc0:4/7 s2l1b0:2/7 s2l1b1:3/7 s2l1b2:3/7 s2l2b0:1/7 s2l2b1:2/7 s2l2b2:3/7 s2l3b0:2/7
s2l3b1:3/7 s2l3b2:3/7 s2l4b0:2/7 s2l4b1:3/7 s2l4b2:2/7 s2l5b0:2/7 s2l5b1:3/7 s2l5b2:2/7 s3l0b0:2/7
s3l0b1:2/7 s3l0b2:3/7 s3l0sc0:2/7 s3l1b0:2/7 s3l1b1:4/7 s3l1b2:4/6 s3l2b0:2/7 s3l2b1:4/7 s3l2b2:4/6
fc.0:4/7"""


def _expand_short_key(k: str) -> str:
    if k == 'head.0':
        return 'head.0'
    if k == 'fc.0':
        return 'classifier.0'
    s, rest = k[1:].split('l', 1)
    if 'sc' in rest:
        l, _ = rest.split('sc')
        return f'stage_{s}_layer_{l}.shortcut.0'
    l, b = rest.split('b')
    return f'stage_{s}_layer_{l}.body.{2 * int(b)}'


for _tok in _r50_raw.split():
    _k, _v = _tok.split(':')
    _a, _b = _v.split('/')
    R50_NVIDIA_FRACLENS[_expand_short_key(_k)] = (int(_a), int(_b))


# Learned fraction lengths of the reference's MobileNet-V2 run, as printed in its committed log
# (`/root/reference/fraclen_visual/mbv2_fix_quant.out:1267-1901`; input_fl rounded as `int_conv` does,
# fix_quant_ops.py:695-699).  Depthwise layers sit at 8/0, 8/1, 8/6, project convs read unsigned fraclen-8 inputs: requant
# shifts up to 11 (avg-pool -> classifier) and unsigned `input_fl = 8` on non-head layers.  BASELINE config C3.
MBV2_LOG_FRACLENS = {}
_mbv2_raw = """head.0:6/7 s0l0b0:8/0 s0l0b1:1/7 s1l0b0:1/7 s1l0b1:4/7 s1l0b2:8/7 s1l1b0:7/5 s1l1b1:6/7 s1l1b2:8/7 s2l0b0:7/7
s2l0b1:8/0 s2l0b2:7/7 s2l1b0:6/7 s2l1b1:6/7 s2l1b2:8/7 s2l2b0:6/7 s2l2b1:6/7 s2l2b2:8/7 s3l0b0:6/7 s3l0b1:6/7 s3l0b2:8/7
s3l1b0:6/7 s3l1b1:6/7 s3l1b2:8/7 s3l2b0:6/7 s3l2b1:8/0 s3l2b2:6/7 s3l3b0:6/7 s3l3b1:6/7 s3l3b2:8/7 s4l0b0:6/7 s4l0b1:6/7
s4l0b2:8/7 s4l1b0:7/7 s4l1b1:8/1 s4l1b2:8/7 s4l2b0:7/7 s4l2b1:8/1 s4l2b2:8/7 s5l0b0:7/7 s5l0b1:8/7 s5l0b2:8/7 s5l1b0:7/7
s5l1b1:8/6 s5l1b2:8/7 s5l2b0:7/7 s5l2b1:8/0 s5l2b2:7/7 s6l0b0:7/7 s6l0b1:8/6 s6l0b2:8/7 tail.0:6/7 fc.0:8/7"""
for _tok in _mbv2_raw.split():
    _k, _v = _tok.split(':')
    _a, _b = _v.split('/')
    MBV2_LOG_FRACLENS['tail.0' if _k == 'tail.0' else _expand_short_key(_k)] = (int(_a), int(_b))


def real_fraclens(arch: str):
    """The reference's own learned (input_fl, weight_fl) table for `arch`, where its logs hold one (else None)."""
    return {'resnet50': R50_NVIDIA_FRACLENS, 'mobilenet_v2': MBV2_LOG_FRACLENS}.get(arch)
