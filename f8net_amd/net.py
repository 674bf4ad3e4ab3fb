"""Host-side mirror of the reference's IntModel on top of libf8net.so.

`F8Net` walks a topology table + an exported-IntModel parameter dict and records the same integer
graph the reference's `IntModel.forward` executes (/root/reference/models/fix_resnet.py:352-383,
fix_mobilenet_v2.py:207-241, fix_mobilenet_v1.py:120-147) through the C ABI; the library plans it
into fused HIP launches.  torch is used for device memory and streams only.
"""
import ctypes

import numpy as np

from . import _lib, topology
from ._lib import ConvDesc, LinearDesc, check

AVGPOOL_SHIFT = 6   # FXQAvgPool2d(7).shiftnum = round(log2(49)), fix_quant_ops.py:121-122


def _np_i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


class F8Net:
    """One planned integer network (or single op) living in libf8net.so."""

    def __init__(self):
        self._L = _lib.lib()
        self._h = self._L.f8_net_create()
        if not self._h:
            raise MemoryError('f8_net_create failed')
        self.max_batch = 0
        self.out_elems = 0
        self.out_float = True
        self.in_shape = None

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self._L.f8_net_destroy(h)

    # ---- builder (thin wrappers over the C ABI) ------------------------------------------
    def input(self, C, H, W, fraclen):
        self.in_shape = (C, H, W)
        return check(self._L.f8_net_input(self._h, C, H, W, fraclen))

    def conv(self, src, weight, bias, *, stride, pad, groups, weight_fl, input_fl, input_signed,
             quant_input=True, relu=False, label=None):
        w = _np_i32(weight)
        cout, cin_g, k, k2 = w.shape
        assert k == k2
        d = ConvDesc(cin=cin_g * groups, cout=cout, kernel=k, stride=stride, pad=pad, groups=groups,
                     weight_fl=int(weight_fl), input_fl=int(input_fl), input_signed=int(bool(input_signed)),
                     quant_input=int(bool(quant_input)), relu=int(bool(relu)))
        b = None if bias is None else _np_i32(bias)
        t = check(self._L.f8_net_conv(self._h, src, ctypes.byref(d), w.ctypes.data,
                                      None if b is None else b.ctypes.data))
        if label:
            self._L.f8_net_set_label(self._h, t, label.encode())
        return t

    def linear(self, src, weight, bias, *, weight_fl, input_fl, input_signed, quant_input=True, label=None):
        w = _np_i32(weight)
        d = LinearDesc(in_features=w.shape[1], out_features=w.shape[0], weight_fl=int(weight_fl),
                       input_fl=int(input_fl), input_signed=int(bool(input_signed)),
                       quant_input=int(bool(quant_input)))
        b = None if bias is None else _np_i32(bias)
        t = check(self._L.f8_net_linear(self._h, src, ctypes.byref(d), w.ctypes.data,
                                        None if b is None else b.ctypes.data))
        if label:
            self._L.f8_net_set_label(self._h, t, label.encode())
        return t

    def add(self, a, b, relu=False, label=None):
        t = check(self._L.f8_net_add(self._h, a, b, int(bool(relu))))
        if label:
            self._L.f8_net_set_label(self._h, t, label.encode())
        return t

    def maxpool(self, src, k=3, stride=2, pad=1, label=None):
        t = check(self._L.f8_net_maxpool(self._h, src, k, stride, pad))
        if label:
            self._L.f8_net_set_label(self._h, t, label.encode())
        return t

    def avgpool_sum(self, src, shift=AVGPOOL_SHIFT, label=None):
        t = check(self._L.f8_net_avgpool_sum(self._h, src, shift))
        if label:
            self._L.f8_net_set_label(self._h, t, label.encode())
        return t

    def output(self, src, as_float=True):
        self.out_float = bool(as_float)
        check(self._L.f8_net_output(self._h, src, int(self.out_float)))

    def finalize(self, max_batch):
        check(self._L.f8_net_finalize(self._h, int(max_batch)))
        self.max_batch = int(max_batch)
        self.out_elems = int(self._L.f8_net_output_elems(self._h))
        return self

    # ---- introspection ----------------------------------------------------------------------
    def describe(self) -> str:
        n = self._L.f8_net_describe(self._h, None, 0)
        buf = ctypes.create_string_buffer(n)
        self._L.f8_net_describe(self._h, buf, n)
        return buf.value.decode()

    @property
    def num_launches(self):
        return check(self._L.f8_net_num_launches(self._h))

    @property
    def arena_bytes(self):
        return int(self._L.f8_net_arena_bytes(self._h))

    @property
    def weight_bytes(self):
        return int(self._L.f8_net_weight_bytes(self._h))

    @property
    def output_fraclen(self):
        return check(self._L.f8_net_output_fraclen(self._h))

    def launch_info(self, i, N):
        name = ctypes.create_string_buffer(256)
        b, o = ctypes.c_double(), ctypes.c_double()
        check(self._L.f8_net_launch_info(self._h, i, N, name, 256, ctypes.byref(b), ctypes.byref(o)))
        return name.value.decode(), b.value, o.value

    def num_parts(self, N):
        return check(self._L.f8_net_num_parts(self._h, int(N)))

    def step_launches(self, i, N):
        """Kernel launches planned launch i issues per run of N images (sub-batches / chunks), f8_net_step_launches."""
        return check(self._L.f8_net_step_launches(self._h, i, int(N)))

    def launch_valu(self, i, N):
        """Essential vector lane-operations of planned launch i for N images (f8_net_launch_valu)."""
        v = ctypes.c_double(0.0)
        check(self._L.f8_net_launch_valu(self._h, i, int(N), ctypes.byref(v)))
        return v.value

    def launch_kernel(self, i):
        buf = ctypes.create_string_buffer(256)
        check(self._L.f8_net_launch_kernel(self._h, i, buf, 256))
        return buf.value.decode()

    # ---- execution (torch = device memory + stream plumbing) ---------------------------------
    def upload(self):
        check(self._L.f8_net_upload(self._h))

    def _check_input(self, x):
        import torch
        if not x.is_cuda:
            raise ValueError('F8Net.run: input must be a CUDA/HIP tensor (there is no CPU path)')
        if x.dtype != torch.int32 or not x.is_contiguous():
            raise ValueError('F8Net.run: input must be contiguous int32 NCHW')
        if tuple(x.shape[1:]) != tuple(self.in_shape):
            raise ValueError(f'F8Net.run: input shape {tuple(x.shape)} != [N,{self.in_shape}]')
        if not (1 <= x.shape[0] <= self.max_batch):
            raise ValueError(f'F8Net.run: batch {x.shape[0]} outside [1,{self.max_batch}]')

    def set_option(self, key, value):
        """Per-handle tuning option (f8_net_set_option; keys in include/f8net.h).  Planning keys before finalize()."""
        check(self._L.f8_net_set_option(self._h, key.encode(), int(value)))
        return self

    def get_option(self, key):
        v = ctypes.c_int(0)
        check(self._L.f8_net_get_option(self._h, key.encode(), ctypes.byref(v)))
        return v.value

    def _input_ready(self, ev):
        """One-shot: the next run also waits for `ev` (a torch.cuda.Event recorded behind the producer of its input)."""
        if ev is not None:
            check(self._L.f8_net_set_input_ready(self._h, ctypes.c_void_p(ev.cuda_event)))

    def run(self, x, out=None, input_ready=None):
        """x: int32 CUDA tensor [N,C,H,W] (reference input format).  Returns [N, out_elems].
        input_ready: torch.cuda.Event recorded behind the producer of `x` when that is another stream (pipelined callers)."""
        import torch
        self._check_input(x)
        N = x.shape[0]
        if out is None:
            out = torch.empty((N, self.out_elems), dtype=torch.float32 if self.out_float else torch.int32,
                              device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            self._input_ready(input_ready)
            check(self._L.f8_net_run(self._h, x.data_ptr(), out.data_ptr(), N, ctypes.c_void_p(stream)))
        return out

    def check(self):
        """Synchronise the device and raise if a kernel of an earlier run reported a failure (f8_net_check)."""
        check(self._L.f8_net_check(self._h))
        return self

    def autotune(self, N, device=None):
        """Measured tile selection for runs of N images (f8_net_autotune).  Returns the number of launches re-tiled."""
        import torch
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            return check(self._L.f8_net_autotune(self._h, int(N), ctypes.c_void_p(stream)))

    def set_pipelined(self, on=True):
        """Let consecutive runs overlap (f8_net_set_pipelined): the caller keeps inputs / outputs of consecutive runs in
        buffers that were ready one call earlier (static input, double-buffered outputs).  on=2 / 'alternate': whole
        batches of consecutive runs alternate between two internal streams instead of splitting every run in two."""
        mode = 2 if on in (2, 'alternate') else int(bool(on))
        check(self._L.f8_net_set_pipelined(self._h, mode))

    def run_f32(self, images, normalize, out=None, input_ready=None):
        """images: float32 CUDA tensor [N,C,H,W] as forward_loss receives them (fix_train.py:676-692); the input
        quantisation runs inside the input kernel.  normalize: FLAGS.normalize of the reference."""
        import torch
        if not images.is_cuda:
            raise ValueError('F8Net.run_f32: input must be a CUDA/HIP tensor (there is no CPU path)')
        if images.dtype != torch.float32 or not images.is_contiguous():
            raise ValueError('F8Net.run_f32: input must be contiguous float32 NCHW')
        if tuple(images.shape[1:]) != tuple(self.in_shape):
            raise ValueError(f'F8Net.run_f32: input shape {tuple(images.shape)} != [N,{self.in_shape}]')
        N = images.shape[0]
        if not (1 <= N <= self.max_batch):
            raise ValueError(f'F8Net.run_f32: batch {N} outside [1,{self.max_batch}]')
        if out is None:
            out = torch.empty((N, self.out_elems), dtype=torch.float32 if self.out_float else torch.int32,
                              device=images.device)
        stream = torch.cuda.current_stream(images.device).cuda_stream
        with torch.cuda.device(images.device):
            self._input_ready(input_ready)
            check(self._L.f8_net_run_f32(self._h, images.data_ptr(), int(bool(normalize)), out.data_ptr(), N,
                                         ctypes.c_void_p(stream)))
        return out

    def run_u8(self, images, normalize=False, mean=None, std=None, nhwc=False, out=None, input_ready=None):
        """images: uint8 CUDA tensor, NCHW [N,3,H,W] (or NHWC [N,H,W,3] with nhwc=True) as a decoder yields them; ToTensor /
        Normalize(mean, std) / the input quantisation of forward_loss are a table lookup inside the input kernel (f8_net_run_u8)."""
        import torch
        if not images.is_cuda:
            raise ValueError('F8Net.run_u8: input must be a CUDA/HIP tensor (there is no CPU path)')
        if images.dtype != torch.uint8 or not images.is_contiguous():
            raise ValueError('F8Net.run_u8: input must be contiguous uint8')
        C, H, W = self.in_shape
        want = (H, W, C) if nhwc else (C, H, W)
        if tuple(images.shape[1:]) != want:
            raise ValueError(f'F8Net.run_u8: input shape {tuple(images.shape)} != [N,{want}]')
        N = images.shape[0]
        if not (1 <= N <= self.max_batch):
            raise ValueError(f'F8Net.run_u8: batch {N} outside [1,{self.max_batch}]')
        if out is None:
            out = torch.empty((N, self.out_elems), dtype=torch.float32 if self.out_float else torch.int32, device=images.device)
        m = (ctypes.c_float * 3)(*[float(v) for v in mean]) if mean is not None else None
        s = (ctypes.c_float * 3)(*[float(v) for v in std]) if std is not None else None
        stream = torch.cuda.current_stream(images.device).cuda_stream
        with torch.cuda.device(images.device):
            self._input_ready(input_ready)
            check(self._L.f8_net_run_u8(self._h, images.data_ptr(), int(bool(nhwc)), int(bool(normalize)), m, s, out.data_ptr(), N,
                                        ctypes.c_void_p(stream)))
        return out

    def run_profiled(self, x, out=None):
        """Like run(); also returns per-launch milliseconds (HIP events on the launch stream)."""
        import torch
        self._check_input(x)
        N = x.shape[0]
        if out is None:
            out = torch.empty((N, self.out_elems), dtype=torch.float32 if self.out_float else torch.int32,
                              device=x.device)
        n = self.num_launches
        ms = (ctypes.c_float * n)()
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            check(self._L.f8_net_run_profiled(self._h, x.data_ptr(), out.data_ptr(), N,
                                              ctypes.c_void_p(stream), ms, n))
        return out, [float(v) for v in ms]


def _fl(params, key, name):
    return int(np.asarray(params[f'{key}.{name}']).reshape(-1)[0])


def build_net(spec: topology.NetSpec, params: dict, max_batch: int, hw: int = 224,
              input_fraclen=None, options=None) -> F8Net:
    """Record IntModel.forward for `spec` with exported parameters `params` (numpy or torch-cpu
    tensors keyed like the reference state_dict) and plan it for batches up to max_batch.
    options: {key: value} for f8_net_set_option, applied before planning."""
    net = record_net(spec, params, hw, input_fraclen)
    for k, v in (options or {}).items():
        net.set_option(k, v)
    return net.finalize(max_batch)


def record_net(spec: topology.NetSpec, params: dict, hw: int = 224, input_fraclen=None) -> F8Net:
    """The recording half of build_net: the graph is in the handle, not yet planned (set planning options, then finalize)."""
    params = {k: (v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)) for k, v in params.items()}
    net = F8Net()
    head_in_fl = _fl(params, spec.head.key, 'input_fraclen')
    if input_fraclen is None:
        input_fraclen = head_in_fl if spec.normalize else 8     # fix_train.py:683-692
    if input_fraclen != head_in_fl:
        raise ValueError(f'network input fraclen {input_fraclen} != head.input_fraclen {head_in_fl}: the reference '
                         f'feeds the head conv head-format integers without requantising (fix_resnet.py:356-358)')

    def conv(src, c: topology.ConvSpec, quant_input=True):
        return net.conv(src, params[c.key + '.weight'], params[c.key + '.bias'], stride=c.stride, pad=c.pad,
                        groups=c.groups, weight_fl=_fl(params, c.key, 'weight_fraclen'),
                        input_fl=_fl(params, c.key, 'input_fraclen'), input_signed=c.signed_in,
                        quant_input=quant_input, relu=c.relu, label=c.key)

    t = net.input(3, hw, hw, input_fraclen)
    t = conv(t, spec.head, quant_input=False)                    # fix_resnet.py:356-358
    if spec.head_maxpool:
        t = net.maxpool(t, 3, 2, 1, label='head.maxpool')        # fix_resnet.py:359
    for b in spec.blocks:                                        # IntBlock.forward, fix_resnet.py:26-77
        x = t
        r = x
        for c in b.body:
            r = conv(r, c)
        if b.shortcut is not None:
            sx = conv(x, b.shortcut)
            r = net.add(r, sx, relu=b.post_relu, label=b.name)
        elif b.residual:
            r = net.add(r, x, relu=b.post_relu, label=b.name)
        t = r
    if spec.tail is not None:                                    # fix_mobilenet_v2.py:218-224
        t = conv(t, spec.tail)
    t = net.avgpool_sum(t, AVGPOOL_SHIFT, label='avgpool')       # fix_quant_ops.py:126-134
    k = spec.fc_key
    t = net.linear(t, params[k + '.weight'], params[k + '.bias'], weight_fl=_fl(params, k, 'weight_fraclen'),
                   input_fl=_fl(params, k, 'input_fraclen'), input_signed=spec.fc_signed_in, label=k)
    net.output(t, as_float=True)                                 # `.float()`, fix_resnet.py:383
    return net
