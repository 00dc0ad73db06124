"""ctypes binding of the C-ABI HIP library (``libcddmsl_hip.so``, declared in ``include/cddmsl_hip.h``).

The library is the product: every hot-path op of the package goes through it.  There is no CPU
or eager-PyTorch fallback -- if the shared object is missing, or a call returns a non-zero
status, this module raises.  PyTorch only supplies device memory (``Tensor.data_ptr()``) and the
stream handle the kernels are enqueued on.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcddmsl_hip.so")

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    """Return the loaded library, loading it on first use.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  cddmsl_amd has no fallback path."
            )
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def stream_ptr():
    """The HIP stream torch is currently enqueuing on, as a void* (raw-stream query: ~1 us, this runs once per launch)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def to_device_async(cpu_tensor, device):
    """Host -> device copy that does NOT synchronise: staged through pinned memory and enqueued on the current stream
    (a pageable ``.to(device)`` blocks until everything already queued has run, i.e. acts as a full device sync)."""
    if torch.device(device).type == "cpu":
        return cpu_tensor
    if cpu_tensor.numel() == 0:
        return torch.empty(cpu_tensor.shape, dtype=cpu_tensor.dtype, device=device)
    return cpu_tensor.pin_memory().to(device, non_blocking=True)


def ptr(t):
    """Device pointer of a tensor (or NULL for None)."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def check(status, what):
    if status != 0:
        raise HipLibraryError(f"{what} failed with status {status}")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipLibraryError(
                "cddmsl_amd ops run only on the MI355X HIP path; got a CPU tensor "
                "(the CPU restatement lives in oracle/ and is test infrastructure only)"
            )


class Readback:
    """Device -> host copy whose wait covers ONLY the work enqueued before it: the copy goes to pinned memory on the current
    stream and an event is recorded right behind it; ``get()`` blocks on that event.  A ``tensor.cpu()`` / ``.tolist()`` at the
    point of use would instead wait for everything enqueued in between -- the idea here is to enqueue independent device work
    (the next stage, another branch) between issuing a readback and consuming it, so the device stays busy while the host
    does the data-dependent part (sampling permutations, list building)."""

    def __init__(self, t):
        if t.is_cuda:
            self.host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self.host.copy_(t, non_blocking=True)
            self.ev = torch.cuda.Event()
            self.ev.record()
        else:
            self.host, self.ev = t, None

    def get(self):
        if self.ev is not None:
            self.ev.synchronize()
        return self.host
