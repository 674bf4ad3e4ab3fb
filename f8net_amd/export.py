"""Float -> int exporter (ResNets, MobileNet-V1 / V2): what `Model.int_model()` of the reference does before the int_op_only path runs
(SURVEY.md §8f-1).  Input: the `state_dict()` of a trained F8Net float model (`best_model.pt`, keys `head.0.conv.weight`,
`stage_i_layer_j.body.k.bn.running_var`, `….alpha`, `….input_fraclen`, `classifier.0.weight`, …) plus the handful of
yml flags that shape the export.  Output: the int32 parameter set of the exported IntModel under ITS keys
(`head.0.weight`, `stage_i_layer_j.body.{0,2,4}.bias`, `….weight_fraclen`, `….input_fraclen`), i.e. exactly what
`f8net_amd.int_model.IntModel.load_state_dict` / `f8net_amd.net.build_net` take.

Restated from (not copied; same float32 operation ORDER, because rounding at .5 and the fraclen search decide integers):
  /root/reference/models/fix_quant_ops.py
      fix_quant :64-87, metric2fraclen :30-37, fraclen_gridsearch :17-27,
      ReLUClipFXQConvBN.fix_scaling :503-520, float_weight :535-573, float_bias :575-584, int_weight :586-600,
      int_bias :602-618, get_weight_fraclen :660-678, int_conv :680-714,
      ReLUClipFXQLinear.fix_scaling :987-1004, float_bias :1022-1061, int_weight :1063-1076, int_bias :1078-1096,
      get_weight_fraclen :1147-1163, int_fc :1165-1195,  FXQAvgPool2d.scale :118-124
  /root/reference/models/fix_resnet.py
      master / following wiring: BasicBlock :125-147,196-207  Bottleneck :227-254,302-311  Model :437-486,
      int_block :209-224,313-326, int_model :526-544

  /root/reference/models/fix_mobilenet_v1.py :53-93,186-232,262-279   fix_mobilenet_v2.py :76-170,275-352,405-423

This is one-time host-side float32 work (torch on the CPU): nothing here is on the timed path.
"""
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import topology


@dataclass
class ExportConfig:
    """The yml flags that reach the exporter (apps/imagenet/*/…int_op_only*.yml; broadcast to layers at fix_train.py:270-295)."""
    weight_format: Tuple[int, int] = (8, 7)
    input_format: Tuple[int, int] = (8, 6)
    format_from_metric: bool = True
    metric: str = 'std'
    format_grid_search: bool = False
    no_clipping: bool = False
    input_fraclen_sharing: bool = False
    rescale_forward: bool = True            # ReLUClipFXQLinear only (fix_train.py:292-293)
    rescale_forward_conv: bool = False      # ReLUClipFXQConvBN (fix_train.py:290-291)
    rescale_type: str = 'constant'
    normalize: bool = False                 # head conv: weight_only = not normalize, double_side = normalize
    quant_avgpool: bool = True
    pool_fusing: bool = True
    bn_eps: float = 1e-5
    # The FLOAT model's own `int_infer` evaluation (fix_quant_ops.py:418-431) instead of `int_model()`'s export: ResNets and
    # MobileNet-V1 store the pool's 64 / 49 on the last BLOCK module (fix_resnet.py:472-477, fix_mobilenet_v1.py:218-223), where no conv reads it —
    # a conv's own `avgpool_scale` stays 1.0 until `int_conv(avgpool_scale=...)` runs — so in that mode the last conv is quantised WITHOUT the
    # scale (other integers, other weight fraclen) and the pool is sum / 64.  MobileNet-V2 stores it on `tail[0]` itself: no difference there.
    int_infer_eval: bool = False


def float_key(int_key: str) -> str:
    """IntModel key -> float-model key: the exported body interleaves ReLUs (`body.0, body.2, body.4`), the float body
    does not (`body.0, body.1, body.2`)."""
    parts = int_key.split('.')
    if len(parts) == 3 and parts[1] == 'body':
        parts[2] = str(int(parts[2]) // 2)
    return '.'.join(parts)


def fix_quant(x, wl, fl, align_dim, signed):
    """fix_quant_ops.py:64-87 (non-floating): round(x * 2^fl), clamp, / 2^fl."""
    expand = x.dim() - align_dim - 1
    fl = fl[(...,) + (None,) * expand]
    res = x * (2 ** fl)
    res.round_()
    bound = 2 ** (wl - 1) - 1 if signed else 2 ** wl - 1
    res.clamp_(max=bound, min=-bound if signed else 0)
    res.div_(2 ** fl)
    return res


def metric2fraclen(metric_tensor, metric='std', N=1, signed=True):
    coeff = ({'std': 40, 'mae': 30, 'rms': 40} if signed else {'std': 70, 'mae': 30, 'rms': 50})[metric]
    fl = torch.floor(torch.log2(coeff * N / metric_tensor))
    fl.clamp_(max=8 - int(signed), min=0)
    return fl


def fraclen_gridsearch(x, wl, align_dim, signed):
    errs = []
    for fl in range(wl + 1 - int(signed)):
        res = fix_quant(x, wl, torch.ones(x.shape[align_dim]) * fl * 1.0, align_dim, signed)
        errs.append(torch.mean((x - res) ** 2) ** 0.5)
    return torch.argmin(torch.tensor(errs)) * 1.0


def _weight_metric(w, metric):
    dims = tuple(range(w.dim()))
    if metric == 'std':
        return torch.std(w, axis=dims)
    if metric == 'mae':
        return torch.mean(torch.abs(w), axis=dims)
    if metric == 'rms':
        return torch.mean(w ** 2, axis=dims) ** 0.5
    raise NotImplementedError(metric)


class _Layer:
    """One ReLUClipFXQConvBN / ReLUClipFXQLinear as the exporter sees it."""

    def __init__(self, key, sd, cfg, linear=False, weight_only=False, double_side=False, bita_min=None, groups=1, kernel=1, cout=0):
        self.key, self.cfg, self.linear = key, cfg, linear
        self.weight_only, self.double_side = weight_only, double_side
        self.groups, self.kernel, self.cout = groups, kernel, cout
        self.master: Optional['_Layer'] = None
        self.following: Optional['_Layer'] = None
        fk = float_key(key)

        def t(name):
            full = f'{fk}.{name}'
            if full not in sd:
                raise KeyError(f'export: float state_dict has no `{full}`')
            return torch.as_tensor(sd[full]).detach().to(torch.float32).cpu().clone()
        if linear:
            self.weight, self.bias = t('weight'), t('bias')
        else:
            self.weight = t('conv.weight')
            self.bn_w, self.bn_b = t('bn.weight'), t('bn.bias')
            self.bn_mean, self.bn_var = t('bn.running_mean'), t('bn.running_var')
        self.alpha = t('alpha')
        self.input_fraclen = t('input_fraclen')
        x_wl, x_fl = cfg.input_format
        if bita_min is not None:
            x_wl = max(x_wl, bita_min)
        self.input_format = (x_wl, x_fl)
        self.weight_format = cfg.weight_format
        self.avgpool_scale = 1.0

    # -- sharing through the master chain
    def get_alpha(self):
        if self.master is not None:
            return self.master.get_alpha()
        if self.weight_only:
            return torch.ones_like(self.alpha)
        return self.alpha

    def get_input_format(self):
        return self.master.get_input_format() if self.master is not None else self.input_format

    def get_input_fraclen(self):
        if self.weight_only:
            return torch.ones_like(self.input_fraclen) * 8
        if self.master is not None and self.cfg.input_fraclen_sharing:
            return self.master.get_input_fraclen()
        return self.input_fraclen

    def rounded_input_fraclen(self):
        x_wl, _ = self.get_input_format()
        fl = torch.round(self.get_input_fraclen())
        return torch.clamp(fl, max=x_wl - int(self.double_side), min=0)

    @property
    def fix_scaling(self):
        alpha = torch.abs(self.get_alpha())
        if self.cfg.no_clipping:
            return torch.ones_like(alpha)
        if self.weight_only:
            return alpha
        x_wl, _ = self.get_input_format()
        return 2 ** self.rounded_input_fraclen() * alpha / (2 ** (x_wl - int(self.double_side)) - 1)

    # -- float parameters with BN and the activation scales folded in
    @property
    def float_weight(self):
        if self.linear:
            return self.weight
        weight = self.weight
        if self.cfg.rescale_forward_conv:
            if self.cfg.rescale_type == 'stddev':
                weight_scale = torch.std(self.weight)
            elif self.cfg.rescale_type == 'constant':
                weight_scale = 1.0 / (self.cout * self.kernel * self.kernel) ** 0.5
            else:
                raise NotImplementedError(self.cfg.rescale_type)
            weight_scale /= torch.std(weight)
        else:
            weight_scale = 1.0
        weight = weight * weight_scale
        bn_std = torch.sqrt(self.bn_var + self.cfg.bn_eps)
        if self.groups == 1:
            return (self.bn_w / bn_std)[:, None, None, None] * weight * self.fix_scaling[(...,) + (None, None)] / \
                self.following.fix_scaling[(...,) + (None, None, None)]
        return (self.bn_w / bn_std)[:, None, None, None] * weight * self.fix_scaling[(...,) + (None, None, None)] / \
            self.following.fix_scaling[(...,) + (None, None, None)]

    def get_weight_fraclen(self):
        weight = self.float_weight * self.avgpool_scale if not self.linear else self.float_weight
        w_wl, _ = self.weight_format
        if self.cfg.format_grid_search:
            return fraclen_gridsearch(weight, w_wl, 0, True)
        if self.cfg.format_from_metric:
            assert w_wl == 8
            fl = metric2fraclen(_weight_metric(weight, self.cfg.metric), self.cfg.metric, 1, True)
            return torch.clamp(fl, max=w_wl - 1, min=0)
        raise NotImplementedError('export needs format_grid_search or format_from_metric')

    @property
    def float_bias(self):
        if not self.linear:
            bn_std = torch.sqrt(self.bn_var + self.cfg.bn_eps)
            return (self.bn_b - self.bn_w / bn_std * self.bn_mean) / self.following.fix_scaling
        weight = self.weight * 1.0
        w_wl, _ = self.weight_format
        weight = fix_quant(weight, w_wl, self.get_weight_fraclen(), 0, True)
        if self.cfg.rescale_forward:
            if self.cfg.rescale_type == 'stddev':
                weight_scale = torch.std(self.weight)
            elif self.cfg.rescale_type == 'constant':
                weight_scale = 1.0 / (self.weight.shape[0]) ** 0.5
            else:
                raise NotImplementedError(self.cfg.rescale_type)
            weight_scale /= torch.std(weight)
        else:
            weight_scale = 1.0
        return self.bias / self.fix_scaling / weight_scale

    def export(self) -> Dict[str, torch.Tensor]:
        w_wl, _ = self.weight_format
        wfl = self.get_weight_fraclen()
        src = self.float_weight if self.linear else self.float_weight * self.avgpool_scale
        int_weight = (fix_quant(src, w_wl, wfl, 0, True) * (2 ** wfl)).int()
        in_fl = self.rounded_input_fraclen()
        fb = self.float_bias if self.linear else self.float_bias * self.avgpool_scale
        int_bias = (fix_quant(fb, 32, in_fl + wfl, 0, True) * (2 ** (in_fl + wfl))).int()
        return {f'{self.key}.weight': int_weight, f'{self.key}.bias': int_bias,
                f'{self.key}.weight_fraclen': wfl.int(), f'{self.key}.input_fraclen': in_fl.int()}


def export_int_state(spec: topology.NetSpec, float_state: dict, cfg: ExportConfig) -> Dict[str, torch.Tensor]:
    """`model.int_model().state_dict()` of the reference (ResNets, MobileNet-V1 / V2) from the float model's state_dict.

    Wiring (who shares whose activation scale = `master`, whose scale divides my weights = `following`):
      ResNet   fix_resnet.py:125-147,196-207,227-254,302-311,437-486     identity blocks chain their first conv to the stage's
               master, a downsample block's first conv AND shortcut take the previous master and reset it
      MBV2     fix_mobilenet_v2.py:76-134,162-170,311-352                same chaining through `residual_connection` blocks;
               the tail conv takes the last master; signed (`double_side`) inputs as in the topology table
      MBV1     fix_mobilenet_v1.py:53-80,204-230                         no masters
    """
    resnet, mbv2, mbv1 = spec.arch.startswith('resnet'), spec.arch == 'mobilenet_v2', spec.arch == 'mobilenet_v1'
    if not (resnet or mbv2 or mbv1):
        raise NotImplementedError(f'export: {spec.arch}')
    layers: Dict[str, _Layer] = {}

    def conv(c: topology.ConvSpec, **kw):
        kw.setdefault('double_side', c.signed_in)
        L = _Layer(c.key, float_state, cfg, groups=c.groups, kernel=c.k, cout=c.cout, **kw)
        layers[c.key] = L
        return L

    head = conv(spec.head, weight_only=not cfg.normalize, double_side=cfg.normalize, bita_min=8)
    prev_tail = [head]                      # layers whose `following` is the next block's first conv
    master = None
    for b in spec.blocks:
        body = [conv(c) for c in b.body]
        sc = None
        if resnet:
            body[0].master = master
            if b.shortcut is not None:
                sc = conv(b.shortcut)
                sc.master = master
                master = None
            else:
                master = body[0]
        elif mbv2:
            body[0].master = master
            master = body[0] if b.residual else None
        for L in prev_tail:
            L.following = body[0]
        for a, nxt in zip(body, body[1:]):
            a.following = nxt
        prev_tail = [body[-1]] + ([sc] if sc is not None else [])
    last_conv = layers[spec.blocks[-1].body[-1].key]
    if spec.tail is not None:
        tail = conv(spec.tail)
        tail.master = master
        for L in prev_tail:
            L.following = tail
        prev_tail = [tail]
        last_conv = tail
    fc = _Layer(spec.fc_key, float_state, cfg, linear=True)
    layers[spec.fc_key] = fc
    for L in prev_tail:
        L.following = fc
    if cfg.quant_avgpool and not (cfg.int_infer_eval and (resnet or mbv1)):
        # FXQAvgPool2d(7).scale = 2^round(log2(49)) / 49 goes to the last conv before the pool
        # (fix_resnet.py:536-539, fix_mobilenet_v1.py:271-275, fix_mobilenet_v2.py:419-420)
        shiftnum = torch.round(torch.log2(torch.tensor(7 ** 2))).int().item()
        last_conv.avgpool_scale = 2 ** shiftnum / (7 ** 2)
    out: Dict[str, torch.Tensor] = {}
    with torch.no_grad():
        for key in spec.layer_keys():
            out.update(layers[key].export())
    return out


def load_float_checkpoint(path_or_dict) -> dict:
    """A reference checkpoint (`best_model.pt`: `{'model': DataParallel(model).state_dict()}`, fix_train.py:1113-1115; or a
    bare state_dict) -> float state with the `module.` prefix of the wrapper removed and bookkeeping buffers dropped."""
    ck = torch.load(path_or_dict, map_location='cpu') if isinstance(path_or_dict, (str, bytes)) or hasattr(path_or_dict, 'read') else path_or_dict
    if isinstance(ck, dict) and 'model' in ck and isinstance(ck['model'], dict):
        ck = ck['model']                                   # fix_train.py:881-882
    out = {}
    for k, v in ck.items():
        k = k[len('module.'):] if k.startswith('module.') else k
        if k.endswith('num_batches_tracked'):
            continue
        out[k] = v
    return out


def int_infer_model_from_float(arch: str, float_state, cfg: ExportConfig, num_classes: int = 1000):
    """The reference's `int_infer` evaluation mode of the FLOAT model (`int_infer: True` in every shipped test yml; fix_quant_ops.py:418-431,
    916-929; fix_resnet.py:158-187, 489-505) as a GPU module: every conv / linear of that mode is `conv(int weights, int(input * 2^fl)) / 2^fl` with
    float ReLU / residual adds / pool in between, i.e. the integer network evaluated on values float32 carries exactly — so it runs as the planned
    integer network on the integers THAT mode uses (`ExportConfig.int_infer_eval`: see there for where they differ from `int_model()`'s), and
    `IntModel.forward_int_infer(x)` quantises the float batch as the float head does and returns real-valued logits."""
    import dataclasses
    return int_model_from_float(arch, float_state, dataclasses.replace(cfg, int_infer_eval=True), num_classes)


def int_model_from_float(arch: str, float_state, cfg: ExportConfig, num_classes: int = 1000):
    """float checkpoint / state_dict -> `f8net_amd.int_model.IntModel` holding the exported integers (the GPU counterpart of
    `model.int_model()`, fix_train.py:930-935)."""
    from . import int_model
    spec = topology.get(arch, num_classes=num_classes, normalize=cfg.normalize)
    sd = export_int_state(spec, load_float_checkpoint(float_state), cfg)
    m = int_model.IntModel(spec)
    ref = m.state_dict()
    m.load_state_dict({k: v.reshape(ref[k].shape) for k, v in sd.items()}, strict=True)
    return m
