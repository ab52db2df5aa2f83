"""Tensor plumbing shared by the op wrappers: validation + raw pointers.

PyTorch is used for device memory and streams only; the compute is in
libpn2ops.so. The checks mirror the OP_REQUIRES validation that the reference
performs in OpKernel::Compute (SURVEY.md section 8b "Error convention"), with
the reference's message text.
"""
import torch


def require(cond, msg):
    if not cond:
        raise ValueError(msg)


def _checked(t, name, dtype, dtype_name):
    # fast path first: the messages below cost more to format than the checks do (every operator call pays them)
    if isinstance(t, torch.Tensor) and t.dtype is dtype and t.is_cuda:
        return t if t.is_contiguous() else t.contiguous()
    require(isinstance(t, torch.Tensor), "%s must be a torch.Tensor" % name)
    require(t.dtype == dtype, "%s must be %s, got %s" % (name, dtype_name, t.dtype))
    require(t.is_cuda, "%s must live on a ROCm device (got %s); there is no CPU path" % (name, t.device))
    return t.contiguous()


def f32(t, name):
    return _checked(t, name, torch.float32, "float32")


def i32(t, name):
    return _checked(t, name, torch.int32, "int32")


def out_or_empty(out, shape, dtype, dev, name="out"):
    """A caller-provided output buffer (the `out=` option of the operators: no allocation on the call path,
    ~1.7 us per tensor saved) or a fresh one. A provided buffer must match exactly; nothing is copied."""
    if out is None:
        return torch.empty(shape, dtype=dtype, device=dev)
    if not (isinstance(out, torch.Tensor) and out.dtype is dtype and tuple(out.shape) == tuple(shape) and out.device == dev
            and out.is_contiguous()):
        raise ValueError("%s must be a contiguous %s tensor of shape %s on %s" % (name, dtype, tuple(shape), dev))
    return out


def same_device(*ts):
    dev = ts[0].device
    for t in ts[1:]:
        if t.device != dev:
            raise ValueError("all tensors must be on the same device (%s vs %s)" % (dev, t.device))
    return dev


def ptr(t):
    return t.data_ptr() if t is not None else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device):
    """The current HIP stream of `device` as an integer handle (the raw getter skips building a
    torch.cuda.Stream object: ~0.3 us instead of ~2 us per operator call)."""
    if _raw_stream is not None:
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


class on_device:
    """Make `device` current for a launch, like torch.cuda.device(), but free when it already is
    (the common case): torch.cuda.device() alone costs several microseconds per operator call."""

    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx = device.index if device.index is not None else torch.cuda.current_device()
        self.prev = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            torch.cuda.set_device(self.idx)
            self.prev = cur
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


_deterministic = False


def set_deterministic(flag):
    """Route the scatter-add gradients (gather_point, group_point, three_interpolate) through the
    order-independent fixed-point kernels (pn2_*_grad_det): identical bits on every run, at about
    twice the accumulation traffic. Off by default, like the reference (fp32 atomics). Also switched
    on by torch.use_deterministic_algorithms(True)."""
    global _deterministic
    _deterministic = bool(flag)


def is_deterministic():
    return _deterministic or torch.are_deterministic_algorithms_enabled()


def det_workspace(lib, b, rows, c, device):
    """Scratch for one deterministic gradient call (int64 accumulators + header)."""
    nbytes = lib.pn2_det_grad_ws_bytes(b, rows, c)
    return torch.empty(((nbytes + 7) // 8,), dtype=torch.int64, device=device)


# When the scatter-add gradients run as a segmented reduction (csrc/seg_grad.hip) instead of atomics:
# always from 16 channels up; below that only when the index inversion can use the one-workgroup-per-
# cloud LDS path (>= 4 clouds, <= 24576 target rows), where it beats the atomics even at 3 channels.
SEG_GRAD_MIN_CHANNELS = 16


def use_segmented_grad(b, rows, c):
    return c >= SEG_GRAD_MIN_CHANNELS or (b >= 4 and rows <= 24576)


def seg_workspace(lib, b, rows, entries, device):
    nbytes = lib.pn2_seg_grad_ws_bytes(b, rows, entries)
    return torch.empty(((nbytes + 7) // 8,), dtype=torch.int64, device=device)
