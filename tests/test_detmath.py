"""Accuracy of the normative arithmetic (mulactseg_amd/csrc/detmath.h), evaluated through its plain-C
build (oracle/exact.c) against float64 numpy: exp and log within 1 ulp, fixed-point conversion exact,
softmax rows sum to one.  CPU-only."""
import numpy as np

from oracle import exact


def _ulp_err(y, ref):
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    return np.max(np.abs(y.astype(np.float64) - ref) / ulp)


def test_expf_within_one_ulp_and_edges():
    x = np.linspace(-103.9, 88.7, 1500001).astype(np.float32)
    ref = np.exp(x.astype(np.float64))
    keep = ref > 1.2e-38                         # normal results
    assert _ulp_err(exact.expf(x)[keep], ref[keep]) < 1.0
    edge = exact.expf(np.array([-np.inf, -200.0, -104.5, 0.0, -0.0, 89.0, np.inf, np.nan], dtype=np.float32))
    assert edge[0] == 0 and edge[1] == 0 and edge[2] == 0 and edge[3] == 1 and edge[4] == 1
    assert np.isinf(edge[5]) and np.isinf(edge[6]) and np.isnan(edge[7])
    sub = exact.expf(np.array([-95.0, -100.0], dtype=np.float32))        # subnormal results: one rounding
    assert np.all(np.abs(sub.astype(np.float64) - np.exp([-95.0, -100.0])) <= 1.5e-45)


def test_logf_within_one_ulp():
    rs = np.random.RandomState(0)
    x = np.exp(rs.uniform(-30, 3, 1000000)).astype(np.float32)
    assert _ulp_err(exact.logf(x), np.log(x.astype(np.float64))) < 1.0
    assert exact.logf(np.array([1.0], dtype=np.float32))[0] == 0.0
    tiny = np.array([1e-40, 1.4e-45], dtype=np.float32)
    assert np.allclose(exact.logf(tiny), np.log(tiny.astype(np.float64)), rtol=1e-6)


def test_fix_is_exact_floor():
    rs = np.random.RandomState(1)
    v = np.concatenate([rs.uniform(0, 2.0, 100000), [0.0, 1e-45, 1e-39, 1.0, 1.9999999, 1e-20, -1.0, -0.0]]).astype(np.float32)
    for frac in (24, 32, 40):
        ref = np.floor(np.maximum(v, 0).astype(np.float64) * 2.0 ** frac)
        ref[v < 1.1754944e-38] = 0               # zero, negative and subnormal inputs give 0
        assert np.array_equal(exact.fix(v, frac), ref.astype(np.uint64))
    big = np.array([18.4, 3.0], dtype=np.float32)   # loss values use 32 fractional bits
    assert np.array_equal(exact.fix(big, 32), np.floor(big.astype(np.float64) * 2.0 ** 32).astype(np.uint64))


def test_softmax_rows():
    rs = np.random.RandomState(2)
    z = rs.uniform(-1, 1, size=(5000, 20)).astype(np.float32)
    invT = exact.inv_temperature(0.1)
    p = exact.softmax_rows(z, invT)
    ref = np.exp((z.astype(np.float64) * float(invT)))
    ref /= ref.sum(axis=1, keepdims=True)
    assert np.max(np.abs(p - ref)) < 1e-6
    assert np.max(np.abs(p.sum(axis=1) - 1.0)) < 1e-6
    assert float(invT) == 10.0                   # float32(1 / float32(0.1))
