"""ctypes binding of libmargipose_hip.so (include/margipose_hip.h).

There is NO fallback: if the library is missing or a tensor is not a ROCm device tensor the call
raises.  (`python -m margipose_amd.build` or `__graft_entry__.build()` compiles the library.)
"""
import ctypes
import os

import torch

from . import build as _build

_LIB = None

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_int64 = ctypes.c_int64

ABI_VERSION = 16         # must equal mpose_abi_version() of the library (csrc/tail.hip)
MAX_GROUP = 3
MAX_TAPS = 12
MAX_CLASSES = 8


class MposeError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        path = _build.LIB_PATH
        if not os.path.exists(path):
            raise MposeError('libmargipose_hip.so is missing (%s). Build it with `python -m margipose_amd.build`; '
                             'there is no CPU/PyTorch fallback for the MargiPose hot path.' % path)
        _LIB = ctypes.CDLL(path)
        _LIB.mpose_abi_version.restype = c_int
        if _LIB.mpose_abi_version() != ABI_VERSION:
            raise MposeError('libmargipose_hip.so ABI version mismatch')
        _LIB.mpose_bn_bwd_reduce_ws_bytes.restype = c_int64
        _LIB.mpose_h2_bytes.restype = c_int64
        _LIB.mpose_h2_bytes.argtypes = [c_int64, c_int]
    return _LIB


def check(rc, what):
    if rc != 0:
        raise MposeError('%s failed with code %d' % (what, rc))


def stream_ptr():
    # (torch.cuda.current_stream() costs ~9 us of Python per call -- 8 ms of an 890-launch training step; the raw query is 0.3 us)
    return c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def raw_stream(stream):
    """The hipStream_t of a torch.cuda.Stream as a void pointer."""
    return c_void_p(stream.cuda_stream)


def stream_wait(waiter, signaler):
    """`waiter` (a torch.cuda.Stream) waits for everything enqueued on `signaler` so far -- Stream.wait_stream through the library
    (mpose_stream_wait: event record + stream wait), so that a launch plan being recorded sees the dependency (csrc/plan.hip)."""
    check(lib().mpose_stream_wait(raw_stream(waiter), raw_stream(signaler)), 'mpose_stream_wait')


def fill_zero(t):
    """t.zero_() as a launch of this library (recordable by a launch plan; ATen's fill is not)."""
    n = t.numel() * t.element_size()
    if n:
        check(lib().mpose_fill_u32(c_void_p(t.data_ptr()), 0, c_int64(n), stream_ptr()), 'mpose_fill_u32')
    return t


def copy_into(dst, src):
    """dst.copy_(src) for contiguous device tensors of equal byte size, as a launch of this library."""
    n = src.numel() * src.element_size()
    if n != dst.numel() * dst.element_size() or not (src.is_contiguous() and dst.is_contiguous()):
        raise MposeError('copy_into needs contiguous tensors of equal size')
    if n:
        check(lib().mpose_copy_bytes(c_void_p(src.data_ptr()), c_void_p(dst.data_ptr()), c_int64(n), stream_ptr()), 'mpose_copy_bytes')
    return dst


def plan_recording():
    return bool(lib().mpose_plan_recording())


PLAN_HOST_OPS = None        # while train_helpers.PlannedTrainStep records: the host actions of the iteration, in order


def plan_host(fn):
    """Run fn() -- a host action inside an iteration that a launch plan cannot record (issuing a collective, waiting for one).
    While a plan is being recorded the point is marked (mpose_plan_break) and fn is kept: a replay stops there, calls fn() and
    continues.  fn must only touch buffers that live as long as the plan."""
    if PLAN_HOST_OPS is not None and plan_recording():
        check(lib().mpose_plan_break(), 'mpose_plan_break')
        PLAN_HOST_OPS.append(fn)
    return fn()


def dev_f32(t, name='tensor'):
    """Validate a device fp32 contiguous tensor and return it."""
    if not isinstance(t, torch.Tensor):
        raise MposeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise MposeError('%s must live on a ROCm device (got %s): margipose_amd has no CPU path' % (name, t.device))
    if t.dtype != torch.float32:
        raise MposeError('%s must be float32 (got %s)' % (name, t.dtype))
    if not t.is_contiguous():
        raise MposeError('%s must be contiguous' % name)
    return t


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(None)


def ptr_array(tensors, n=MAX_GROUP):
    arr = (c_void_p * n)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


# ---- structs mirroring include/margipose_hip.h ------------------------------------------------
class Tap(ctypes.Structure):
    _fields_ = [('dy', ctypes.c_int8), ('dx', ctypes.c_int8), ('widx', ctypes.c_int8), ('acc', ctypes.c_int8)]


class TapClass(ctypes.Structure):
    _fields_ = [('n_taps', c_int), ('oy', c_int), ('ox', c_int), ('taps', Tap * MAX_TAPS)]


class ConvGeom(ctypes.Structure):
    _fields_ = [('B', c_int), ('IH', c_int), ('IW', c_int), ('Cin', c_int),
                ('OH', c_int), ('OW', c_int), ('Cout0', c_int), ('Cout1', c_int),
                ('GH', c_int), ('GW', c_int), ('in_mul', c_int), ('out_mul', c_int),
                ('n_classes', c_int), ('Npad0', c_int), ('Npad1', c_int),
                ('in_ld', c_int), ('out_ld0', c_int), ('out_ld1', c_int), ('in_mul_x', c_int), ('out_mul_x', c_int),
                ('cls', TapClass * MAX_CLASSES)]


class ConvOperands(ctypes.Structure):
    _fields_ = [('in_', c_void_p), ('in_scale', c_void_p), ('in_shift', c_void_p),
                ('w0', c_void_p), ('w1', c_void_p), ('out0', c_void_p), ('out1', c_void_p),
                ('stats0', c_void_p), ('stats1', c_void_p),
                ('mask_src', c_void_p), ('mask_scale', c_void_p), ('mask_shift', c_void_p), ('in1', c_void_p),
                ('epi_scale0', c_void_p), ('epi_shift0', c_void_p), ('add_src', c_void_p), ('add_scale', c_void_p),
                ('add_shift', c_void_p), ('out0_planes', c_void_p),
                ('in_amax', c_void_p), ('in1_amax', c_void_p), ('w0_amax', c_void_p), ('w1_amax', c_void_p), ('out0_amax', c_void_p),
                ('red_a', c_void_p), ('red_b', c_void_p), ('red_scale', c_void_p), ('red_shift', c_void_p), ('red_sums', c_void_p),
                ('mm0', c_void_p), ('fin0', c_void_p), ('fin1', c_void_p), ('fin_count', c_void_p), ('fin_eps', ctypes.c_float),
                ('fin_momentum', ctypes.c_float)]


class WgradOperands(ctypes.Structure):
    _fields_ = [('in_', c_void_p), ('in_scale', c_void_p), ('in_shift', c_void_p),
                ('gout0', c_void_p), ('gout1', c_void_p), ('dw0', c_void_p), ('dw1', c_void_p),
                ('in_amax', c_void_p), ('gout0_amax', c_void_p), ('gout1_amax', c_void_p), ('single_product', c_int), ('planes_in', c_int)]


class AbsmaxOperands(ctypes.Structure):
    _fields_ = [('src', c_void_p), ('scale', c_void_p), ('shift', c_void_p), ('dst', c_void_p)]


class BnAddOperands(ctypes.Structure):
    _fields_ = [('a', c_void_p), ('a_scale', c_void_p), ('a_shift', c_void_p),
                ('b', c_void_p), ('b_scale', c_void_p), ('b_shift', c_void_p), ('out', c_void_p), ('out_amax', c_void_p)]


class SplitH2Operands(ctypes.Structure):
    _fields_ = [('src', c_void_p), ('scale', c_void_p), ('shift', c_void_p), ('planes', c_void_p), ('amax', c_void_p)]


class BnBwdReduceOperands(ctypes.Structure):
    _fields_ = [('g', c_void_p), ('a', c_void_p), ('b', c_void_p), ('a_scale', c_void_p), ('a_shift', c_void_p),
                ('sums', c_void_p)]


class BnBwdApplyOperands(ctypes.Structure):
    _fields_ = [('g', c_void_p), ('a', c_void_p), ('b', c_void_p), ('coef_a', c_void_p), ('coef_b', c_void_p),
                ('a_scale', c_void_p), ('a_shift', c_void_p), ('da', c_void_p), ('db', c_void_p),
                ('da_amax', c_void_p), ('db_amax', c_void_p)]


# Host-side mirrors of the device-resident job tables (filled into int64 tensors, see engine.py).
PACK_JOB_WORDS = 8        # src, dst, (N,K), (T,Npad), (Kpad,pad), sn, sk, st      -> 8 x int64
UNPACK_JOB_WORDS = 9
BN_JOB_WORDS = 10
BN_COEF_JOB_WORDS = 9
