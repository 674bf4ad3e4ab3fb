"""Caller side of the int_op_only path: the slices of the reference's `forward_loss`
(/root/reference/fix_train.py:676-718) that sit either side of `model(input)`, on the GPU.

    quantize_input(images, model, normalize)   fix_train.py:683-692  (op-level: returns the int32 tensor + fraclen)
    topk_correct(output, target, topk)         fix_train.py:697-704  (the rows cat'ed behind the loss)
    forward_loss(model, images, target, ...)   the whole evaluation step; input quantisation fused into the net's
                                               input kernel, per-sample correctness flags averaged over ranks exactly as
                                               dist_all_reduce_tensor does (myutils/distributed.py:79-87) over RCCL.

Everything computes in libf8net.so; there is no CPU path.
"""
import ctypes

import torch

from . import _lib
from ._lib import check


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def quantize_input(images, head_input_fraclen=None, input_symmetric=False, normalize=False, check_input=False):
    """fix_train.py:683-692 as an op: float32 CUDA images -> int32 tensor tagged `.output_fraclen`.

    normalize False: (255 * x).round_().int(), fraclen 8 (the reference asserts x >= 0; so do we).
    normalize True : (fix_quant(x, 8, fl, 1, symmetric)[0] * 2**fl).int() with fl = head.input_fraclen."""
    if not isinstance(images, torch.Tensor) or not images.is_cuda:
        raise ValueError('quantize_input: expects a CUDA/HIP tensor (no CPU path)')
    if images.dtype != torch.float32:
        raise TypeError(f'quantize_input: expects float32, got {images.dtype}')
    x = images.contiguous()
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    if normalize:
        if head_input_fraclen is None:
            raise ValueError('quantize_input: normalize=True needs the head conv input_fraclen')
        fl = int(head_input_fraclen)
    else:
        if check_input:               # fix_train.py:689 `assert torch.all(input >= 0)`: a device -> host sync per batch, so opt-in
            assert torch.all(x >= 0)
        fl = 8
    with torch.cuda.device(x.device):
        check(_lib.lib().f8_quantize_input_f32(x.data_ptr(), out.data_ptr(), x.numel(), int(bool(normalize)), fl,
                                               int(bool(input_symmetric)), _stream(x)))
    setattr(out, 'output_fraclen', fl)
    return out


def topk_correct(output, target, topk=(1, 5)):
    """fix_train.py:697-704: float32 [len(topk), N]; row i = 1.0 where the target is among the topk[i] largest logits
    (equal logits rank by lower class index)."""
    if not output.is_cuda or output.dtype != torch.float32:
        raise ValueError('topk_correct: expects float32 CUDA logits (no CPU path)')
    N, C = output.shape
    out = output.contiguous()
    tgt = target.to(device=out.device, dtype=torch.int64).contiguous()
    ks = (ctypes.c_int * len(topk))(*[int(k) for k in topk])
    correct = torch.empty((len(topk), N), dtype=torch.float32, device=out.device)
    with torch.cuda.device(out.device):
        check(_lib.lib().f8_topk_correct_f32(out.data_ptr(), tgt.data_ptr(), N, C, ks, len(topk), correct.data_ptr(),
                                             _stream(out)))
    return correct


def forward_loss(model, images, target, topk=(1, 5), normalize=False, distributed_all_reduce=False, group=None, check_input=False):
    """One evaluation step of the reference (fix_train.py:676-718) around an f8net_amd IntModel.

    Returns (output, top-k error lists): `errors[k]` = list of per-sample 1 - correct_k, after the reference's
    all-reduce-and-divide over ranks when `distributed_all_reduce` (each rank then holds the rank-mean of the flags at
    each batch position, exactly what `meter[...].cache_list` receives in the reference)."""
    if check_input and not normalize:
        # fix_train.py:689 asserts this on every batch; here it would be a device -> host synchronisation in front of every
        # forward (it serialises the pipelined schedule), so it is opt-in.  Out-of-contract (negative) pixels clamp to 0.
        assert torch.all(images >= 0)
    output = model.forward_f32(images, normalize=normalize)
    correct = topk_correct(output, target, topk)
    res = correct.reshape(-1)
    if distributed_all_reduce and torch.distributed.is_available() and torch.distributed.is_initialized():
        world = torch.distributed.get_world_size(group)
        if world > 1:
            torch.distributed.all_reduce(res, group=group)
            res.div_(world)
    res = res.cpu().numpy()
    bs = res.size // len(topk)
    errors = {k: list(1.0 - res[i * bs:(i + 1) * bs]) for i, k in enumerate(topk)}
    return output, errors
