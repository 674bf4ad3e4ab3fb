"""Op-level seam: drop-in replacements for the pieces `IntBlock.forward` / `IntModel.forward` call.

Same names, argument meaning and error behaviour as the reference
(/root/reference/models/fix_quant_ops.py:90-134 and the int `nn.Conv2d` / `nn.Linear` built by
`int_conv` :680-714 / `int_fc` :1165-1195), operating on int32 tensors that live in HBM.  Every
function here ends in a HIP kernel of libf8net.so — through `torch.ops.f8net.*` (torch_ops.py), i.e. the PyTorch dispatcher
over the C ABI; there is no CPU path.

This is the parity granularity (one launch chain per reference op, NCHW int32 at every seam).  The
performance path is the fused whole-net plan in `f8net_amd.net` / `f8net_amd.int_model`.
"""
import torch
import torch.nn as nn

from . import torch_ops  # noqa: F401  (registers torch.ops.f8net.*; every function below goes through the dispatcher)


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
    res = torch.ops.f8net.requant(input, int(fl), int(input_fl), bool(signed))
    setattr(res, 'output_fraclen', fl)
    return res


def relu_(x):
    """nn.ReLU(inplace=True) on an int32 tensor (fix_resnet.py:39,77)."""
    _require_dev_i32(x, 'relu_')
    return torch.ops.f8net.relu_(x)


def add_align_(res, x, res_fraclen, x_fraclen):
    """Residual join of fix_resnet.py:40-54: res (+)= x after aligning fraclens, clamp; in place on
    `res`.  Returns (res, output_fraclen)."""
    _require_dev_i32(res, 'add_align_')
    _require_dev_i32(x, 'add_align_')
    torch.ops.f8net.add_align_(res, x, int(res_fraclen), int(x_fraclen))
    out_fl = max(int(res_fraclen), int(x_fraclen))
    setattr(res, 'output_fraclen', out_fl)
    return res, out_fl


class _Scalars:
    """(weight_fraclen, input_fraclen) of a module as Python ints, re-read only when the buffers were edited (an `.item()`
    on a device-resident buffer is a synchronisation: not once per forward)."""

    def __init__(self):
        self.ver, self.val = None, None

    def get(self, m):
        ver = (m.weight_fraclen._version, m.weight_fraclen.data_ptr(), m.input_fraclen._version, m.input_fraclen.data_ptr())
        if ver != self.ver:
            self.val = (int(m.weight_fraclen.reshape(-1)[0].item()), int(m.input_fraclen.reshape(-1)[0].item()))
            self.ver = ver
        return self.val


class F8Conv2d(nn.Conv2d):
    """The integer `nn.Conv2d` of `int_conv` (fix_quant_ops.py:680-714) with a HIP forward (`torch.ops.f8net.conv2d`).

    Same parameters / buffers / attributes as the reference export: int32 `weight`, `bias`, buffers
    `weight_fraclen` (0-dim) and `input_fraclen` ([1]), attribute `input_symmetric`, `int_op_only`.
    forward(x): x int32 NCHW holding input_fraclen-format 8-bit integers -> int32 NCHW.  The op plans per (weight tensor,
    its in-place version, input shape, device): editing `weight` / `bias` in place re-plans on the next forward."""

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
        self._fl = _Scalars()

    def forward(self, x):
        _require_dev_i32(x, 'F8Conv2d')
        w_fl, in_fl = self._fl.get(self)
        return torch.ops.f8net.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.groups, w_fl, in_fl,
                                      self.input_symmetric)


class F8Linear(nn.Linear):
    """The integer `nn.Linear` of `int_fc` (fix_quant_ops.py:1165-1195) with a HIP forward (`torch.ops.f8net.linear`)."""

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
        self._fl = _Scalars()

    def forward(self, x):
        _require_dev_i32(x, 'F8Linear')
        w_fl, in_fl = self._fl.get(self)
        return torch.ops.f8net.linear(x, self.weight, self.bias, w_fl, in_fl, self.input_symmetric)


class FXQAvgPool2d(nn.Module):
    """fix_quant_ops.py:117-134, int branch: int64 sum over H,W, truncate, fraclen += shiftnum."""

    def __init__(self, kernel_size):
        super().__init__()
        self.kernel_size = kernel_size
        self.shiftnum = int(torch.round(torch.log2(torch.tensor(float(kernel_size ** 2)))).item())
        self.int_op_only = True

    def forward(self, x):
        _require_dev_i32(x, 'FXQAvgPool2d')
        output_fraclen = x.output_fraclen + self.shiftnum
        assert output_fraclen <= 32
        res = torch.ops.f8net.avgpool_sum(x)
        setattr(res, 'output_fraclen', output_fraclen)
        return res


class F8MaxPool2d(nn.Module):
    """Head max-pool (fix_resnet.py:358-359 float detour / FXQMaxPool2d): exact integer max."""

    def __init__(self, kernel_size=3, stride=2, padding=1):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding

    def forward(self, x):
        _require_dev_i32(x, 'F8MaxPool2d')
        return torch.ops.f8net.maxpool(x, self.kernel_size, self.stride, self.padding)


class IntReLU(nn.Module):
    def forward(self, x):
        return relu_(x)
