"""IntModel / IntBlock-shaped module tree with the reference's state_dict keys.

Mirrors the exported integer models of the reference (`Model.int_model()`,
/root/reference/models/fix_resnet.py:526-544, fix_mobilenet_v2.py:405-423,
fix_mobilenet_v1.py:258-281): `head.0`, `stage_{i}_layer_{j}.body.{0,2,4}`,
`stage_{i}_layer_{j}.shortcut.0`, `tail.0`, `classifier.0`, each with `weight`, `bias` (int32),
`weight_fraclen`, `input_fraclen`.  `load_state_dict` therefore accepts the reference export's
state_dict unchanged.

Two forwards over the same parameters:
  * `forward(x)`            — whole network as ONE planned chain of fused HIP launches (the
                              performance path; plan cached per input size / batch capacity);
  * `forward_op_level(x)`   — the reference's own control flow (IntBlock.forward,
                              fix_resnet.py:26-77) calling our op-level kernels one by one, NCHW
                              int32 at every seam (the parity granularity).
Both take what `forward_loss` (fix_train.py:683-692) hands the reference: an int32 NCHW tensor
carrying the attribute `output_fraclen`, here resident in HBM, and return float32 logits.
"""
import torch
import torch.nn as nn

from . import topology
from .net import build_net
from .ops import (F8Conv2d, F8Linear, F8MaxPool2d, FXQAvgPool2d, IntReLU, add_align_,
                  int_op_only_fix_quant)


def _conv_module(c: topology.ConvSpec) -> F8Conv2d:
    return F8Conv2d(c.cin, c.cout, c.k, stride=c.stride, padding=c.pad, groups=c.groups,
                    input_symmetric=c.signed_in)


class IntBlock(nn.Module):
    """fix_resnet.py:12-77 / fix_mobilenet_v2.py:11-48 / fix_mobilenet_v1.py:18-38."""

    def __init__(self, bspec: topology.BlockSpec):
        super().__init__()
        layers = []
        for c in bspec.body:
            # ReLUs occupy the odd indices, so conv keys are body.0/.2/.4 as in the export
            layers.append(_conv_module(c))
            layers.append(IntReLU() if c.relu else None)
        while layers and layers[-1] is None:
            layers.pop()
        self.body = nn.Sequential(*[l if l is not None else nn.Identity() for l in layers])
        self.has_shortcut = bspec.shortcut is not None
        if self.has_shortcut:
            self.shortcut = nn.Sequential(_conv_module(bspec.shortcut))
        self.residual = bspec.residual
        self.post_relu = IntReLU() if bspec.post_relu else None
        self.int_op_only = True

    def forward(self, x):
        res = x
        for layer_ in self.body:
            if isinstance(layer_, nn.Conv2d):
                res = int_op_only_fix_quant(res, 8, layer_.input_fraclen.item(), res.output_fraclen,
                                            layer_.input_symmetric)
                res = layer_(res)
                setattr(res, 'output_fraclen', (layer_.weight_fraclen + layer_.input_fraclen).item())
            elif not isinstance(layer_, nn.Identity):
                fl = res.output_fraclen
                res = layer_(res)
                setattr(res, 'output_fraclen', fl)
        if self.has_shortcut:
            sc = self.shortcut[0]
            sx = int_op_only_fix_quant(x, 8, sc.input_fraclen.item(), x.output_fraclen, sc.input_symmetric)
            sx = sc(sx)
            res, _ = add_align_(res, sx, res.output_fraclen, (sc.weight_fraclen + sc.input_fraclen).item())
        elif self.residual:
            res, _ = add_align_(res, x, res.output_fraclen, x.output_fraclen)
        if self.post_relu is not None:
            fl = res.output_fraclen
            res = self.post_relu(res)
            setattr(res, 'output_fraclen', fl)
        return res


class IntModel(nn.Module):
    """fix_resnet.py:322-383 / fix_mobilenet_v2.py:179-241 / fix_mobilenet_v1.py:95-147."""

    def __init__(self, spec: topology.NetSpec):
        super().__init__()
        self.spec = spec
        head = [_conv_module(spec.head), IntReLU()]
        if spec.head_maxpool:
            head.append(F8MaxPool2d(3, 2, 1))
        self.head = nn.Sequential(*head)
        for b in spec.blocks:
            setattr(self, b.name, IntBlock(b))
        if spec.tail is not None:
            self.tail = nn.Sequential(_conv_module(spec.tail), IntReLU())
        self.avgpool = FXQAvgPool2d(7)
        self.classifier = nn.Sequential(F8Linear(spec.fc_in, spec.num_classes, input_symmetric=spec.fc_signed_in))
        self.int_op_only = True
        self._plans = {}
        self._ptensors = None
        self._head_fl = None
        self._pipelined = 0
        self._depth = 2

    # -- performance path ------------------------------------------------------------------
    def _param_version(self):
        # Sum of the in-place version counters over a CACHED list of the parameter / buffer tensors: versions only grow, so any
        # in-place edit changes the sum.  (Walking state_dict() and hashing 216 (version, pointer) pairs per forward cost 0.32 ms —
        # a fifth of a ResNet-50 step; this is ~15 us.)  The list is rebuilt after load_state_dict / .to() / replan().
        ts = self._ptensors
        if ts is None:
            ts = self._ptensors = tuple(self.state_dict(keep_vars=True).values())
        # ... and an ORDER-SENSITIVE fold of the storage pointers: `param.data = new_tensor` — how the reference itself installs integer
        # weights (fix_quant_ops.py:705-706) — bumps no version counter, it only rebinds the storage; a plain XOR would miss
        # two parameters swapping their storages, the multiply-add chain does not
        v = p = 0
        for t in ts:
            v += t._version
            p = (p * 1000003 + t.data_ptr()) & 0xFFFFFFFFFFFFFFFF
        return (v, p)

    def _head_fraclen(self):
        # `head.input_fraclen.item()` on a device buffer is a device -> host synchronisation on EVERY forward (it serialised the
        # pipelined schedule: 94 k vs 109 k img/s); the value is cached per parameter version
        ver = self._param_version()
        if self._head_fl is None or self._head_fl[0] != ver:
            self._head_fl = (ver, int(self.head[0].input_fraclen.item()))
        return self._head_fl[1]

    def _apply(self, fn, *args, **kwargs):
        self._plans = {}
        self._ptensors = None
        self._head_fl = None
        return super()._apply(fn, *args, **kwargs)

    def plan(self, hw: int, max_batch: int, device=None):
        """The planned net for hw x hw inputs on `device` (a handle is bound to one device).  Re-planned when the batch
        capacity grows or any parameter was edited in place since the plan was built."""
        dev = None if device is None else torch.device(device).index
        key = (dev, hw)
        ver = self._param_version()
        ent = self._plans.get(key)
        if ent is None or ent[0] != ver or ent[1].max_batch < max_batch:
            opts = {'whole_batch_launches': 1, 'arena_copies': self._depth, 'pipeline_depth': self._depth} if self._pipelined == 2 else None
            net = build_net(self.spec, self.state_dict(), max_batch, hw, options=opts)
            if self._pipelined:
                net.set_pipelined(self._pipelined)
            ent = (ver, net)
            self._plans[key] = ent
        return ent[1]

    def replan(self):
        """Drop every cached plan.  Plans are re-built automatically after `load_state_dict` and after in-place edits that bump
        `Tensor._version` (`weight.copy_(...)`, `weight[...] = v`); edits through `.data` have a private version counter and
        are invisible to that check (fingerprinting 25 M weights on every forward is not an option on this path): call this
        (also after replacing a parameter OBJECT, e.g. `conv.weight = nn.Parameter(...)`)."""
        self._plans = {}
        self._ptensors = None

    def set_pipelined(self, mode, depth=2):
        """Let consecutive forwards overlap inside the library (f8_net_set_pipelined; 2 = `depth` whole batches in flight, 2..4:
        one arena copy each).
        CONTRACT (include/f8net.h): a run then no longer waits for work queued on the stream after the PREVIOUS run's entry.
        `forward` / `forward_f32` therefore refuse to allocate in this mode: pass `out=` (buffers that rotate with at least the
        pipeline depth) and hand over inputs produced on another stream with `input_ready=` (an event recorded behind the
        producer); inputs produced on the current stream must have been complete one call earlier."""
        new = 2 if mode in (2, 'alternate') else int(bool(mode))
        depth = max(2, min(4, int(depth)))
        if (new == 2) != (self._pipelined == 2) or depth != self._depth:
            self._plans = {}                                  # whole-batch planning hint / arena copies are decided when a plan is built
        self._pipelined, self._depth = new, depth
        for _, net in self._plans.values():
            net.set_pipelined(self._pipelined)

    def _check_pipelined(self, out):
        if self._pipelined and out is None:
            raise ValueError('IntModel: pipelined mode needs a caller-owned, rotating `out=` buffer (a fresh allocation could recycle a block '
                             'that an in-flight run still writes); see IntModel.set_pipelined')

    def forward(self, x, out=None, input_ready=None):
        if not hasattr(x, 'output_fraclen'):
            raise ValueError('IntModel.forward: input must carry `output_fraclen` (fix_train.py:687,692)')
        head_fl = self._head_fraclen()
        if x.output_fraclen != head_fl:
            raise ValueError(f'input output_fraclen {x.output_fraclen} != head.input_fraclen {head_fl}')
        assert x.shape[2] == x.shape[3], 'square inputs'
        self._check_pipelined(out)
        return self.plan(int(x.shape[2]), int(x.shape[0]), x.device).run(x.contiguous(), out=out, input_ready=input_ready)

    def forward_f32(self, images, normalize=False, out=None, input_ready=None):
        """forward_loss's input quantisation (fix_train.py:683-692) fused into the input kernel: `images` is the
        float32 batch the data loader yields; no int32 copy of it is ever written."""
        assert images.shape[2] == images.shape[3], 'square inputs'
        self._check_pipelined(out)
        return self.plan(int(images.shape[2]), int(images.shape[0]), images.device).run_f32(images.contiguous(), normalize, out=out,
                                                                                              input_ready=input_ready)

    def forward_integize(self, x):
        """The reference's float-carried evaluation of the IntModel — the branches taken when `int_op_only` is unset
        (fix_resnet.py:384-409 with IntBlock :78-118; fix_mobilenet_v2.py / fix_mobilenet_v1.py likewise): `x` is the REAL-
        valued float batch; it enters as `(x * 2^head.input_fraclen).int()` (`:391`; with FLAGS.normalize through
        fix_quant: round + clamp to +-127, `:386-389`), every layer works on real values (`conv(int) / 2^fl`), and the
        classifier returns its raw integer accumulators (`:408`).  Here the integers are never left: the input is
        quantised the same way and the planned integer network runs; the result equals the reference's integize output
        wherever float32 carries the reference's accumulators exactly (all golden nets: tests/test_oracle_golden.py)."""
        head = self.head[0]
        fl = int(head.input_fraclen.item())
        if self.spec.normalize:
            xi = torch.clamp(torch.round(x * float(2 ** fl)), -127, 127).to(torch.int32)
        else:
            xi = (x * float(2 ** fl)).to(torch.int32)          # `.int()`: truncation toward zero, as the reference
        setattr(xi, 'output_fraclen', fl)
        return self.forward(xi)

    def forward_int_infer(self, x):
        """The FLOAT model's `int_infer` evaluation (fix_quant_ops.py:418-431; build the module with `export.int_infer_model_from_float`): `x` is the
        real-valued float batch.  The head of that mode sees `int(relu(x) * 2^fl)` (a `weight_only` head: fix_quant_ops.py:218-226, truncation) or,
        with FLAGS.normalize, `int(fix_quant(x, 8, fl, signed) * 2^fl)` (round, clamp to +-127); every later layer re-quantises real values that
        float32 holds exactly, so the integers never have to be left; the classifier's accumulators are divided by 2^(weight + input fraclen)
        (`:929`): REAL-valued logits, unlike `forward_integize`."""
        head, fc = self.head[0], self.classifier[0]
        fl = int(head.input_fraclen.item())
        if self.spec.normalize:
            xi = torch.clamp(torch.round(x * float(2 ** fl)), -127, 127).to(torch.int32)
        else:
            xi = (torch.relu(x) * float(2 ** fl)).to(torch.int32)
        setattr(xi, 'output_fraclen', fl)
        return self.forward(xi) / float(2 ** (int(fc.weight_fraclen.item()) + int(fc.input_fraclen.item())))

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._plans = {}
        self._ptensors = None

    # -- parity path: the reference's control flow over op-level kernels ----------------------
    def forward_op_level(self, x):
        fl_in = x.output_fraclen
        h = self.head[0](x.contiguous())
        h = self.head[1](h)
        if self.spec.head_maxpool:
            h = self.head[2](h)
        setattr(h, 'output_fraclen', (self.head[0].weight_fraclen + self.head[0].input_fraclen).item())
        assert fl_in == self.head[0].input_fraclen.item()
        x = h
        for b in self.spec.blocks:
            x = getattr(self, b.name)(x)
        if self.spec.tail is not None:
            t0 = self.tail[0]
            x = int_op_only_fix_quant(x, 8, t0.input_fraclen.item(), x.output_fraclen, t0.input_symmetric)
            x = self.tail[1](t0(x))
            setattr(x, 'output_fraclen', (t0.weight_fraclen + t0.input_fraclen).item())
        x = self.avgpool(x)
        fl = x.output_fraclen
        x = x.view(x.size(0), -1)
        fc = self.classifier[0]
        x = int_op_only_fix_quant(x, 8, fc.input_fraclen.item(), fl, fc.input_symmetric)
        return fc(x).float()


def from_params(spec: topology.NetSpec, params: dict) -> IntModel:
    """Build an IntModel and load an exported-IntModel parameter dict (numpy or torch)."""
    m = IntModel(spec)
    sd = {}
    for k, v in params.items():
        t = v if isinstance(v, torch.Tensor) else torch.from_numpy(__import__('numpy').ascontiguousarray(v))
        sd[k] = t.to(torch.int32)
    m.load_state_dict(sd, strict=True)
    return m
