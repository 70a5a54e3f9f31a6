"""The reference's SO(3) known-answer tests (src/common/utils_test.cc:5-21) against the DEVICE instantiation of
csrc/dmath.h — the code the kernels run — through the C-ABI self-test entry points."""
import ctypes as C

import numpy as np
import pytest

from test_host_kat import check_so3_kats, unpack_so3
from wildcat_slam_amd import records as R

pytestmark = pytest.mark.gpu


def so3_device(gpu, v):
    out = np.zeros(52)
    gpu._ck(gpu.lib.wc_selftest_so3(gpu.h, R.ptr(np.ascontiguousarray(v, float)), C.c_int(1), R.ptr(out)))
    return unpack_so3(out)


def test_dmath_so3_kats_on_the_device(gpu):
    check_so3_kats(lambda v: so3_device(gpu, v))


def test_device_and_host_instantiations_agree(gpu, oracle):
    rng = np.random.default_rng(11)
    for _ in range(30):
        v = rng.normal(size=3) * rng.choice([1e-11, 1e-2, 1.0])
        d = so3_device(gpu, v)
        out = np.zeros(52)
        assert gpu.lib.wc_selftest_so3(C.c_void_p(0), R.ptr(np.ascontiguousarray(v)), C.c_int(0), R.ptr(out)) == 0
        h = unpack_so3(out)
        for k in d:  # device libm (ocml) vs glibc: a few ulp
            assert np.allclose(d[k], h[k], rtol=0, atol=8e-16 * max(1.0, np.abs(h[k]).max())), k
        assert np.allclose(d["jl"], oracle.so3_jl(v), atol=1e-14)


def test_eig3_and_quaternions_on_the_device(gpu):
    rng = np.random.default_rng(4)
    for _ in range(100):
        a = rng.normal(size=(3, 3))
        a = a @ a.T * 10 ** rng.uniform(-6, 2)
        out = np.zeros(12)
        gpu._ck(gpu.lib.wc_selftest_eig3(gpu.h, R.ptr(np.ascontiguousarray(a)), C.c_int(1), R.ptr(out)))
        ev, v = out[:3], out[3:].reshape(3, 3)
        ev2 = np.linalg.eigvalsh(a)
        assert np.abs(ev - ev2).max() <= 4e-15 * ev2.max()
        assert np.abs(a @ v - v * ev).max() <= 1e-14 * ev2.max()
    for _ in range(50):
        qa, qb = rng.normal(size=4), rng.normal(size=4)
        qa, qb = qa / np.linalg.norm(qa), qb / np.linalg.norm(qb)
        f, p = rng.uniform(), rng.normal(size=3)
        inp = np.concatenate([qa, qb, [f], p])
        od, oh = np.zeros(11), np.zeros(11)
        gpu._ck(gpu.lib.wc_selftest_quat(gpu.h, R.ptr(inp), C.c_int(1), R.ptr(od)))
        assert gpu.lib.wc_selftest_quat(C.c_void_p(0), R.ptr(inp), C.c_int(0), R.ptr(oh)) == 0
        assert np.allclose(od, oh, atol=1e-15)
        # slerp end points and unit rotation
        w, x, y, z = qa
        Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                       [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        assert np.allclose(od[4:7], Rm @ p, atol=1e-14)


def test_fused_so3_forms_of_the_factor_kernels(gpu):
    """csrc/so3_fused.h (Exp + Jr from one sincos, Jr^-1 of a logarithm from the quaternion: what k_lin_imu / k_eval_imu
    evaluate instead of cost_functor.h:286-321's separate calls) against the dmath.h helpers the reference's KATs pin,
    on the device, from the series branch (|v| < 1e-10) to rotations close to pi.  The Jr^-1 coefficient 1 - th cot(th/2) / 2
    cancels for small th in BOTH forms (utils.h:38), hence the absolute tolerance on it."""
    rng = np.random.default_rng(23)
    vs = [np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0) * 3.0, np.zeros(3), np.array([3.1, 0.0, 0.0])]
    vs += [rng.normal(size=3) * s for s in (1e-12, 1e-9, 1e-6, 1e-3, 0.1, 1.0) for _ in range(8)]
    for v in vs:
        n = np.linalg.norm(v)
        if n > 3.1:
            v = v * (3.1 / n)
        d = so3_device(gpu, v)
        out = np.zeros(25)
        gpu._ck(gpu.lib.wc_selftest_so3_fused(gpu.h, R.ptr(np.ascontiguousarray(v, float)), R.ptr(out)))
        assert np.allclose(out[0:4], d["exp"], rtol=0, atol=4e-16), v
        # (utils.h:52 forms (1 - cos th) / th: zero below th = 1.5e-8 and ~1e-16 / th of rounding noise above; the fused form
        # 2 sin^2(th / 2) / th has neither, so the two differ by exactly that)
        assert np.allclose(out[4:13].reshape(3, 3), d["jr"], rtol=0, atol=2e-15 + (min(0.5 * n, 4e-16 / n) if n > 0 else 0.0)), v
        assert np.allclose(out[13:16], d["log_exp"], rtol=0, atol=1e-15 * max(1.0, n)), v
        assert np.allclose(out[16:25].reshape(3, 3), d["jr_inv"], rtol=0, atol=1e-9), v
        # Jr_inv(v) Jr(v) = I (utils_test.cc:5-12 with v -> -v)
        assert np.allclose(out[16:25].reshape(3, 3) @ out[4:13].reshape(3, 3), np.eye(3), atol=1e-9), v


def test_diagonal_block_factor_and_inverse(gpu):
    """the damped solve's 32 x 32 diagonal-block kernel on its own (csrc/window.hip: factor_inv32_blk, the critical path of every panel
    step): L and L^-1 of random SPD matrices against numpy, and a matrix that is not positive definite is reported"""
    rng = np.random.default_rng(5)
    for trial in range(4):
        b = rng.normal(size=(32, 40))
        a = b @ b.T + (1e-3 if trial == 0 else 0.5) * np.eye(32)
        L, X, clk, ok = gpu.selftest_factor32(a, 0, 2)
        Lr = np.linalg.cholesky(a)
        Xr = np.linalg.inv(Lr)
        assert ok and clk > 0
        assert np.abs(np.tril(L) - Lr).max() <= 1e-13 * np.abs(Lr).max()
        assert np.abs(np.tril(X) - Xr).max() <= 1e-12 * np.abs(Xr).max()
    assert not gpu.selftest_factor32(-np.eye(32), 0, 1)[3]
