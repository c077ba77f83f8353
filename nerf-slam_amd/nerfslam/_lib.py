"""ctypes loader for libnerfslam_hip.so (the C-ABI of include/nerfslam_hip.h).

The product path has NO fallback: if the shared library is missing or an entry point is
absent, importing / calling raises.  torch must be imported first so that the library binds
to the HIP runtime torch already loaded (same soname libamdhip64.so.7) and can therefore use
torch's streams and device pointers.
"""
import ctypes as C
import os
import threading

import torch  # noqa: F401  (must precede CDLL: see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libnerfslam_hip.so")
# The same sources built with -DNS_TEST_VARIANTS (csrc/common.h, csrc/Makefile): the product's entry points PLUS the superseded /
# comparison kernels and the NS_* tuning switches that select them.  Only code that sets the master switch NS_VARIANTS -- the
# bit-identity tests and the A/B tools -- ever loads it; the product library has neither the kernels nor the switches.
VARIANTS_LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libnerfslam_hip_variants.so")

_lock = threading.Lock()
_libs = {}
# Held while a HIP graph is being captured; host threads that synchronise with the device (the tracker's read-backs under
# --parallel_run) take it around those calls: ROCm 7.2 answered a synchronising call made by one thread while another was
# capturing with hipErrorIllegalState even in thread-local capture mode.
capture_lock = threading.RLock()


NS_OK, NS_EINVAL, NS_ELAUNCH, NS_ENOSUP = 0, -1, -2, -3   # status codes of include/nerfslam_hip.h


class NerfSlamHipError(RuntimeError):
    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


def _load(path):
    if not os.path.exists(path):
        raise NerfSlamHipError(
            f"{path} is missing: build it with `make -C nerf-slam_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    L = C.CDLL(path)
    L.ns_last_error.restype = C.c_char_p
    L.ns_arch.restype = C.c_char_p
    for name in [n for n in ("ns_ba_plan_index_count", "ns_ba_workspace_bytes") if hasattr(L, n)]:
        getattr(L, name).restype = C.c_size_t
    L.ns_ngp_encode_backward_workspace_bytes.restype = C.c_long
    L.ns_ngp_encode_backward_fused_workspace_bytes.restype = C.c_size_t
    L.ns_ba_solve_large_workspace_bytes.restype = C.c_size_t
    L.ns_ngp_mlp_fragment_table_bytes.restype = C.c_size_t
    return L


def lib():
    """The C ABI: the product library -- or, while the master switch NS_VARIANTS is set in the environment (tests / A/B tools
    only; bench.py refuses to run with it), the variants build of the same sources."""
    path = LIB_PATH if os.environ.get("NS_VARIANTS") is None else VARIANTS_LIB_PATH
    L = _libs.get(path)
    if L is None:
        with _lock:
            L = _libs.get(path)
            if L is None:
                L = _libs[path] = _load(path)
    return L


def variant_env(name, default=None):
    """A/B switches of the tools (NS_* environment variables) are honoured ONLY together with the master switch NS_VARIANTS
    (csrc/common.h: ns_variant_env does the same for the library): a stray variable cannot silently change what the product runs."""
    if os.environ.get("NS_VARIANTS") is None:
        return default
    return os.environ.get(name, default)


def check(status, what):
    if status != 0:
        msg = lib().ns_last_error().decode("utf-8", "replace")
        raise NerfSlamHipError(f"{what} failed ({status}): {msg}", status)


def ptr(t):
    """Device (or host) address of a tensor as void*; None -> NULL."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    if not torch.cuda.is_available():
        raise NerfSlamHipError("no HIP device visible: the nerfslam kernels need an MI355X (gfx950)")
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NerfSlamHipError("nerfslam ops take device tensors; got a CPU tensor (there is no CPU fallback)")


def check_contiguous(**named):
    """The reference's only input validation: TORCH_CHECK(x.is_contiguous()) (src/droid.cpp:129-130)."""
    for name, t in named.items():
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")


import contextlib as _contextlib
import gc as _gc


_immortal_graphs = []
_capture_count_lock = threading.Lock()
_captures = 0
_gc_was_enabled = True


@_contextlib.contextmanager
def graph_capture(graph, **kw):
    """`torch.cuda.graph(graph, **kw)`, with two precautions the GPU suite (a dozen trainers created and dropped in one process)
    showed to be necessary on ROCm 7.2:
    * the cyclic garbage collector is OFF for the duration of the capture -- a collection in the middle of a capture (any Python
      allocation can trigger one) freed an unreachable trainer of earlier, and destroying its HIP graphs while a stream is
      capturing aborts the process (`Fatal Python error: Aborted ... Garbage-collecting`, one run in ~5);
    * a captured graph is never destroyed: destroying the graphs of an earlier trainer right BEFORE a capture (what the
      collection at the head of `torch.cuda.graph` does) made the first replay of the new graph segfault, every time.  The
      graphs of a dropped trainer therefore stay alive until the process ends (three small graphs per (re)capture)."""
    import torch
    global _captures, _gc_was_enabled
    with _capture_count_lock:            # (captures may overlap across threads: the collector comes back with the LAST one)
        if _captures == 0:
            _gc_was_enabled = _gc.isenabled()
            _gc.disable()
        _captures += 1
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        _immortal_graphs.append(graph)
        with _capture_count_lock:
            _captures -= 1
            if _captures == 0 and _gc_was_enabled:
                _gc.enable()
