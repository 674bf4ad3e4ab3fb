"""Op-level seam: drop-in replacements for the pieces `IntBlock.forward` / `IntModel.forward` call.

Same names, argument meaning and error behaviour as the reference
(/root/reference/models/fix_quant_ops.py:90-134 and the int `nn.Conv2d` / `nn.Linear` built by
`int_conv` :680-714 / `int_fc` :1165-1195), operating on int32 tensors that live in HBM.  Every
function here ends in a HIP kernel of libf8net.so; there is no CPU path.

This is the parity granularity (one launch chain per reference op, NCHW int32 at every seam).  The
performance path is the fused whole-net plan in `f8net_amd.net` / `f8net_amd.int_model`.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from ._lib import check
from .net import F8Net


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _require_dev_i32(t, who):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f'{who}: expects a CUDA/HIP tensor (no CPU path)')
    if t.dtype != torch.int32:
        raise TypeError(f'{who}: expects int32, got {t.dtype}')


def int_op_only_fix_quant(input, wl=8, fl=0, input_fl=0, signed=True):
    """fix_quant_ops.py:90-114.  Returns a new int32 tensor tagged with `.output_fraclen = fl`."""
    assert wl >= 0
    assert fl >= 0
    if signed:
        assert fl <= wl - 1
    else:
        assert fl <= wl
    assert type(wl) == int
    assert type(fl) == int
    if wl != 8:
        raise NotImplementedError('only the 8-bit word length the reference nets use is built')
    _require_dev_i32(input, 'int_op_only_fix_quant')
    x = input.contiguous()
    res = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(_lib.lib().f8_requant_i32(x.data_ptr(), res.data_ptr(), x.numel(), int(input_fl), int(fl),
                                        int(bool(signed)), _stream(x)))
    setattr(res, 'output_fraclen', fl)
    return res


def relu_(x):
    """nn.ReLU(inplace=True) on an int32 tensor (fix_resnet.py:39,77)."""
    _require_dev_i32(x, 'relu_')
    assert x.is_contiguous()
    with torch.cuda.device(x.device):
        check(_lib.lib().f8_relu_i32(x.data_ptr(), x.numel(), _stream(x)))
    return x


def add_align_(res, x, res_fraclen, x_fraclen):
    """Residual join of fix_resnet.py:40-54: res (+)= x after aligning fraclens, clamp; in place on
    `res`.  Returns (res, output_fraclen)."""
    _require_dev_i32(res, 'add_align_')
    _require_dev_i32(x, 'add_align_')
    assert res.shape == x.shape and res.is_contiguous() and x.is_contiguous()
    out_fl = ctypes.c_int(0)
    with torch.cuda.device(res.device):
        check(_lib.lib().f8_add_align_i32(res.data_ptr(), x.data_ptr(), res.numel(), int(res_fraclen),
                                          int(x_fraclen), ctypes.byref(out_fl), _stream(res)))
    setattr(res, 'output_fraclen', out_fl.value)
    return res, out_fl.value


class _MiniNetCache:
    """One single-op F8Net per (input shape, batch capacity)."""

    def __init__(self, build):
        self._build = build
        self._nets = {}

    def get(self, x):
        C, H, W = x.shape[1], x.shape[2], x.shape[3]
        N = x.shape[0]
        key = (C, H, W)
        net = self._nets.get(key)
        if net is None or net.max_batch < N:
            net = self._build(C, H, W, max(N, 1))
            self._nets[key] = net
        return net


def _scalar(t):
    return int(t.reshape(-1)[0].item())


class F8Conv2d(nn.Conv2d):
    """The integer `nn.Conv2d` of `int_conv` (fix_quant_ops.py:680-714) with a HIP forward.

    Same parameters / buffers / attributes as the reference export: int32 `weight`, `bias`, buffers
    `weight_fraclen` (0-dim) and `input_fraclen` ([1]), attribute `input_symmetric`, `int_op_only`.
    forward(x): x int32 NCHW holding input_fraclen-format 8-bit integers -> int32 NCHW."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, groups=1,
                 input_symmetric=False):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                         groups=groups, bias=True)
        self.weight.requires_grad_(False)
        self.bias.requires_grad_(False)
        self.weight.data = torch.zeros(self.weight.shape, dtype=torch.int32)
        self.bias.data = torch.zeros(self.bias.shape, dtype=torch.int32)
        self.register_buffer('weight_fraclen', torch.zeros((), dtype=torch.int32))
        self.register_buffer('input_fraclen', torch.zeros(1, dtype=torch.int32))
        self.input_symmetric = bool(input_symmetric)
        self.int_op_only = True
        self._cache = _MiniNetCache(self._build)

    def _build(self, C, H, W, N):
        net = F8Net()
        in_fl = _scalar(self.input_fraclen)
        t = net.input(C, H, W, in_fl)
        t = net.conv(t, self.weight.detach().cpu().numpy(), self.bias.detach().cpu().numpy(),
                     stride=self.stride[0], pad=self.padding[0], groups=self.groups,
                     weight_fl=_scalar(self.weight_fraclen), input_fl=in_fl,
                     input_signed=self.input_symmetric, quant_input=False, relu=False)
        net.output(t, as_float=False)
        return net.finalize(N)

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._cache = _MiniNetCache(self._build)     # parameters changed: re-plan lazily

    def forward(self, x):
        _require_dev_i32(x, 'F8Conv2d')
        x = x.contiguous()
        net = self._cache.get(x)
        out = net.run(x)
        H, W = x.shape[2], x.shape[3]
        P = (H + 2 * self.padding[0] - self.kernel_size[0]) // self.stride[0] + 1
        Q = (W + 2 * self.padding[0] - self.kernel_size[0]) // self.stride[0] + 1
        return out.view(x.shape[0], self.out_channels, P, Q)


class F8Linear(nn.Linear):
    """The integer `nn.Linear` of `int_fc` (fix_quant_ops.py:1165-1195) with a HIP forward."""

    def __init__(self, in_features, out_features, input_symmetric=False):
        super().__init__(in_features, out_features, bias=True)
        self.weight.requires_grad_(False)
        self.bias.requires_grad_(False)
        self.weight.data = torch.zeros(self.weight.shape, dtype=torch.int32)
        self.bias.data = torch.zeros(self.bias.shape, dtype=torch.int32)
        self.register_buffer('weight_fraclen', torch.zeros((), dtype=torch.int32))
        self.register_buffer('input_fraclen', torch.zeros(1, dtype=torch.int32))
        self.input_symmetric = bool(input_symmetric)
        self.int_op_only = True
        self._cache = _MiniNetCache(self._build)

    def _build(self, C, H, W, N):
        net = F8Net()
        in_fl = _scalar(self.input_fraclen)
        t = net.input(C, 1, 1, in_fl)
        t = net.linear(t, self.weight.detach().cpu().numpy(), self.bias.detach().cpu().numpy(),
                       weight_fl=_scalar(self.weight_fraclen), input_fl=in_fl,
                       input_signed=self.input_symmetric, quant_input=False)
        net.output(t, as_float=False)
        return net.finalize(N)

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._cache = _MiniNetCache(self._build)

    def forward(self, x):
        _require_dev_i32(x, 'F8Linear')
        x4 = x.contiguous().view(x.shape[0], self.in_features, 1, 1)
        return self._cache.get(x4).run(x4)


class FXQAvgPool2d(nn.Module):
    """fix_quant_ops.py:117-134, int branch: int64 sum over H,W, truncate, fraclen += shiftnum."""

    def __init__(self, kernel_size):
        super().__init__()
        self.kernel_size = kernel_size
        self.shiftnum = int(torch.round(torch.log2(torch.tensor(float(kernel_size ** 2)))).item())
        self.int_op_only = True
        self._cache = _MiniNetCache(self._build)

    def _build(self, C, H, W, N):
        net = F8Net()
        t = net.input(C, H, W, 0)
        t = net.avgpool_sum(t, 0)
        net.output(t, as_float=False)
        return net.finalize(N)

    def forward(self, x):
        _require_dev_i32(x, 'FXQAvgPool2d')
        output_fraclen = x.output_fraclen + self.shiftnum
        assert output_fraclen <= 32
        x = x.contiguous()
        res = self._cache.get(x).run(x)
        setattr(res, 'output_fraclen', output_fraclen)
        return res


class F8MaxPool2d(nn.Module):
    """Head max-pool (fix_resnet.py:358-359 float detour / FXQMaxPool2d): exact integer max."""

    def __init__(self, kernel_size=3, stride=2, padding=1):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self._cache = _MiniNetCache(self._build)

    def _build(self, C, H, W, N):
        net = F8Net()
        t = net.input(C, H, W, 0)
        t = net.maxpool(t, self.kernel_size, self.stride, self.padding)
        net.output(t, as_float=False)
        return net.finalize(N)

    def forward(self, x):
        _require_dev_i32(x, 'F8MaxPool2d')
        x = x.contiguous()
        H, W = x.shape[2], x.shape[3]
        P = (H + 2 * self.padding - self.kernel_size) // self.stride + 1
        Q = (W + 2 * self.padding - self.kernel_size) // self.stride + 1
        return self._cache.get(x).run(x).view(x.shape[0], x.shape[1], P, Q)


class IntReLU(nn.Module):
    def forward(self, x):
        return relu_(x)
