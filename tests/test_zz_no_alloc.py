"""The one entry point that allocates nothing: mcmc_hip_eval_static (include/mcmc_hip.h) against
mcmc_hip_eval, through the C ABI.  The caller's device buffers come straight from the HIP runtime the
library itself links (hipMalloc / hipMemcpy through ctypes): plain pointers, as the ABI takes them.
Collected last on purpose: the round's end runs pytest -x."""
import ctypes as C

import numpy as np
import pytest

from conftest import make_cfg, perturbed
from mcmc_amd import hip as H

pytestmark = pytest.mark.gpu

_H2D, _D2H = 1, 2


class DeviceBuffer:
    """a caller-owned device allocation on GPU 0"""

    def __init__(self, nbytes):
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]
        assert self.rt.hipSetDevice(0) == 0
        self.ptr = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(self.ptr), nbytes) == 0 and self.ptr.value
        self.nbytes = nbytes

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes == self.nbytes
        assert self.rt.hipMemcpy(self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, _H2D) == 0

    def download(self, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes == self.nbytes
        assert self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, _D2H) == 0
        return out

    def free(self):
        if self.ptr.value:
            assert self.rt.hipFree(self.ptr) == 0
            self.ptr = C.c_void_p()


def _static_eval(s, xs):
    x_dev, out_dev = DeviceBuffer(xs.nbytes), DeviceBuffer(xs.shape[0] * 2 * 8)
    x_dev.upload(xs)
    out_dev.upload(np.full((xs.shape[0], 2), np.nan))
    s.eval_static(xs.shape[0], x_dev.ptr.value, out_dev.ptr.value)
    got = out_dev.download((xs.shape[0], 2))
    x_dev.free(); out_dev.free()
    return got


def test_eval_static_equals_eval_on_two_calls_in_a_row(R11, R10):
    """bit-exact (the bar of every FP64 parity test here) against the allocating entry point, on two
    calls in a row, the second with a smaller input -- the static array still holds the first
    call's histories then --, then on another model (fewer days, FMA contraction) on the same GPU."""
    cfg = make_cfg(200, R11)
    s = H.CovidSampler(cfg, R10)
    big = perturbed(cfg, 300, 0.08)
    small = perturbed(cfg, 70, 0.05, seq=3)
    assert s.eval_static_max() >= 300
    want_big, want_small = s.eval(big), s.eval(small)
    got_big = _static_eval(s, big)
    got_small = _static_eval(s, small)
    assert np.array_equal(got_big, want_big) and np.all(np.isfinite(got_big))
    assert np.array_equal(got_small, want_small)
    assert np.array_equal(_static_eval(s, big), want_big)          # and back up again
    # a second context (another model) shares the GPU's array
    cfg2 = make_cfg(45, R11, contract=1, max_delay=12, n_predict=5, fix_behavior_in_advance=5, initial_delay=7.0)
    s2 = H.CovidSampler(cfg2, R10)
    xs2 = perturbed(cfg2, 65, 0.05)
    assert np.array_equal(_static_eval(s2, xs2), s2.eval(xs2))
    assert np.array_equal(_static_eval(s, small), want_small)
    # chain state of a context is untouched by it
    s.set_chains(small, first_chain_id=1)
    before = (s.params(), s.loglike(), s.rng_state())
    _static_eval(s, big)
    after = (s.params(), s.loglike(), s.rng_state())
    assert all(np.array_equal(a, b) for a, b in zip(before, after))
    s.close()
    s2.close()


def test_eval_static_rejects_what_does_not_fit(R11, R10):
    cfg = make_cfg(200, R11)
    s = H.CovidSampler(cfg, R10)
    n_max = s.eval_static_max()
    assert 64 <= n_max < 10**6 and n_max % 64 == 0
    x_dev, out_dev = DeviceBuffer(8 * cfg["dim"] * 8), DeviceBuffer(8 * 2 * 8)
    with pytest.raises(H.McmcHipError, match="exceed the static scratch"):
        s.eval_static(n_max + 1, x_dev.ptr.value, out_dev.ptr.value)
    with pytest.raises(H.McmcHipError):
        s.eval_static(0, x_dev.ptr.value, out_dev.ptr.value)
    with pytest.raises(H.McmcHipError):
        s.eval_static(8, 0, out_dev.ptr.value)
    x_dev.free(); out_dev.free()
    s.close()
