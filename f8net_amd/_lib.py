"""ctypes binding of libf8net.so (the C ABI in include/f8net.h).

The library is the product: if it is missing this module raises at import of the symbols —
there is no Python / CPU fallback for any compute path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('F8NET_LIB') or os.path.join(_HERE, 'libf8net.so')   # override: tuning experiments only

F8_OK = 0


class F8Error(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f'libf8net: {msg} (status {status})')
        self.status = status


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ('cin', 'cout', 'kernel', 'stride', 'pad', 'groups', 'weight_fl', 'input_fl',
                 'input_signed', 'quant_input', 'relu')]


class LinearDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ('in_features', 'out_features', 'weight_fl', 'input_fl', 'input_signed', 'quant_input')]


_lib = None

# every symbol include/f8net.h declares: (name, restype, argtypes)
_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
SYMBOLS = [
    ('f8_status_string', ctypes.c_char_p, [_i]),
    ('f8_last_error', ctypes.c_char_p, []),
    ('f8_version', _i, []),
    ('f8_device_count', _i, []),
    ('f8_requant_i32', _i, [_vp, _vp, _sz, _i, _i, _i, _vp]),
    ('f8_quantize_input_f32', _i, [_vp, _vp, _sz, _i, _i, _i, _vp]),
    ('f8_topk_correct_f32', _i, [_vp, _vp, _i, _i, ctypes.POINTER(_i), _i, _vp, _vp]),
    ('f8_relu_i32', _i, [_vp, _sz, _vp]),
    ('f8_add_align_i32', _i, [_vp, _vp, _sz, _i, _i, ctypes.POINTER(_i), _vp]),
    ('f8_net_create', _vp, []),
    ('f8_net_destroy', None, [_vp]),
    ('f8_net_input', _i, [_vp, _i, _i, _i, _i]),
    ('f8_net_conv', _i, [_vp, _i, ctypes.POINTER(ConvDesc), _vp, _vp]),
    ('f8_net_add', _i, [_vp, _i, _i, _i]),
    ('f8_net_maxpool', _i, [_vp, _i, _i, _i, _i]),
    ('f8_net_avgpool_sum', _i, [_vp, _i, _i]),
    ('f8_net_linear', _i, [_vp, _i, ctypes.POINTER(LinearDesc), _vp, _vp]),
    ('f8_net_output', _i, [_vp, _i, _i]),
    ('f8_net_finalize', _i, [_vp, _i]),
    ('f8_net_describe', _sz, [_vp, ctypes.c_char_p, _sz]),
    ('f8_net_num_launches', _i, [_vp]),
    ('f8_net_arena_bytes', _sz, [_vp]),
    ('f8_net_weight_bytes', _sz, [_vp]),
    ('f8_net_output_fraclen', _i, [_vp]),
    ('f8_net_output_elems', _sz, [_vp]),
    ('f8_net_upload', _i, [_vp]),
    ('f8_net_num_parts', _i, [_vp, _i]),
    ('f8_net_step_launches', _i, [_vp, _i, _i]),
    ('f8_net_run', _i, [_vp, _vp, _vp, _i, _vp]),
    ('f8_net_run_f32', _i, [_vp, _vp, _i, _vp, _i, _vp]),
    ('f8_net_run_u8', _i, [_vp, _vp, _i, _i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _vp, _i, _vp]),
    ('f8_net_run_profiled', _i, [_vp, _vp, _vp, _i, _vp, ctypes.POINTER(ctypes.c_float), _i]),
    ('f8_net_launch_info', _i, [_vp, _i, _i, ctypes.c_char_p, _sz, ctypes.POINTER(ctypes.c_double),
                                ctypes.POINTER(ctypes.c_double)]),
    ('f8_net_launch_kernel', _i, [_vp, _i, ctypes.c_char_p, _sz]),
    ('f8_net_launch_valu', _i, [_vp, _i, _i, ctypes.POINTER(ctypes.c_double)]),
    ('f8_net_set_label', _i, [_vp, _i, ctypes.c_char_p]),
    ('f8_net_set_pipelined', _i, [_vp, _i]),
    ('f8_net_autotune', _i, [_vp, _i, _vp]),
    ('f8_net_set_option', _i, [_vp, ctypes.c_char_p, _i]),
    ('f8_net_get_option', _i, [_vp, ctypes.c_char_p, ctypes.POINTER(_i)]),
    ('f8_net_set_input_ready', _i, [_vp, _vp]),
    ('f8_net_check', _i, [_vp]),
]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'or f8net_amd/csrc/build.sh. There is no fallback path.')
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)        # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status < 0:
        L = lib()
        raise F8Error(status, (L.f8_last_error() or b'').decode() or L.f8_status_string(status).decode())
    return status
