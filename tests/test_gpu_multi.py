"""One handle, several GPUs, one calling thread (sapca_multi_*, SURVEY.md 8b "Threading" / 8e).

A one-GPU box lists its device several times: the members then all-reduce through page-locked host memory inside the
process, everything else (row partition, one host thread per member, the three all-reduce sites of a fit, the shards'
rows of the projection written in place) is the product's multi-GPU path.  Reference: the same estimator on one handle."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import sapca
import sapca_oracle as O
from sapca import _lib as L
from sapca import synth
from sapca import PowerIterationNormalizer as PIN
from sapca import SVDMethod

pytestmark = pytest.mark.gpu


def _host(m, n, dens, k, seed, dtype, centred=True):
    p, i, v = synth.gapped_csr(m, n, dens, k, seed=seed, centred=centred, dtype=dtype)
    return sp.csr_matrix((v.numpy(), i.numpy().astype(np.int64), p.numpy().astype(np.int64)), shape=(m, n))


@pytest.mark.parametrize("ndev", [2, 3])
def test_randomized_fit_transform_sharded_inside_the_library(ndev):
    m, n, k, p, q = 9000, 1200, 10, 6, 3
    A = _host(m, n, 0.05, k, 3, torch.float32)
    om = synth.gaussian_panel(n, k + p, 5).numpy()
    make = lambda: (sapca.SparsePCABuilder.new().n_components(k).svd_method(SVDMethod.Random(p, q, PIN.QR)).build())
    one = make().set_omega(om)
    t1 = one.fit_transform(A)
    md = sapca.MultiDevice(make(), [0] * ndev).set_omega(om)
    assert not md.uses_rccl()                      # a device listed twice: the in-process transport
    t = md.fit_transform(A)
    np.testing.assert_allclose(md.singular_values_(np.float64), one.singular_values_(np.float64), rtol=2e-5)
    assert O.subspace_angle(md.components_(np.float64), one.components_(np.float64)) < 2e-5
    np.testing.assert_allclose(md.mean_(np.float64), one.mean_(np.float64), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(t, t1, atol=2e-4 * np.abs(t1).max())
    # replicated state: every member answers alike, bit for bit
    for i in range(1, ndev):
        assert np.array_equal(md.member(i).components_(np.float32), md.member(0).components_(np.float32))
    # a separate transform through the same handle (Q2: the per-column counts cross the shards)
    t2 = md.transform(A)
    np.testing.assert_allclose(t2, t, atol=1e-5 * np.abs(t).max())
    # against the oracle too
    ptr, idx, val = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64)
    want = O.fit(ptr, idx, val, m, n, n_components=k, n_oversamples=p, n_power_iterations=q, omega=om)
    assert O.subspace_angle(md.components_(np.float64), want.components) < 1e-4


def test_masked_lanczos_f64_sharded_inside_the_library():
    m, n, k = 6000, 900, 6
    A = _host(m, n, 0.06, k, 9, torch.float64, centred=False)
    mask = synth.bernoulli_mask(n, 0.6, 7).numpy()
    make = lambda: sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).svd_method(SVDMethod.Lanczos()).build()
    one = make()
    t1 = one.fit_transform(A)
    md = sapca.MultiDevice(make(), [0, 0])
    t = md.fit_transform(A)
    np.testing.assert_allclose(md.singular_values_(np.float64), one.singular_values_(np.float64), rtol=1e-8)
    assert O.subspace_angle(md.components_(np.float64), one.components_(np.float64)) < 1e-6
    c1, o1 = one.mask_index_maps()
    c2, o2 = md.mask_index_maps()
    assert np.array_equal(c1, c2) and np.array_equal(o1, o2)          # bit-exact index maps on every member
    np.testing.assert_allclose(t, t1, atol=1e-7 * np.abs(t1).max())


def test_a_failing_shard_fails_the_call_and_releases_its_peers():
    """shard 1 holds a column index past n: its upload refuses it; shard 0, already inside the first all-reduce, is released
    (no hang) and the call reports the shard that failed in its own right"""
    m, n, k = 4000, 500, 5
    A = _host(m, n, 0.05, k, 1, torch.float32)
    A.indices = A.indices.astype(np.int64)
    A.indices[-1] = n + 3
    md = sapca.MultiDevice(sapca.SparsePCABuilder.new().n_components(k).svd_method(SVDMethod.Random(4, 1, PIN.QR)).build(), [0, 0])
    with pytest.raises(L.SapcaError, match="shard 1") as e:
        md.fit(A)
    assert e.value.status == L.ERR_ARG
    # the handle is still usable
    B = _host(m, n, 0.05, k, 1, torch.float32)
    assert md.fit_transform(B).shape == (m, k)
    with pytest.raises(L.SapcaError, match="mask vector length"):
        sapca.MultiDevice(sapca.MaskedSparsePCABuilder.new().n_components(2).mask(np.ones(n + 1, bool))
                          .svd_method(SVDMethod.Random(3, 1)).build(), [0, 0]).fit(B)


def test_transposed_sweep_in_two_pieces_with_the_first_all_reduce_behind_the_second(debug_switches, monkeypatch):
    """the n x l panel of an A^T sweep is the one bandwidth-relevant collective of a sharded fit.  Where the
    operator allows it (DPP-fed sweep, rows in natural order) the sweep runs in two pieces of its output rows and the first
    piece's all-reduce runs on a side stream behind the second piece's sweep; the members agree on the cut (their own row
    blocks differ).  Same fit as with SAPCA_AT_OVERLAP=0 to f32 rounding, and the timings say which path ran."""
    m, n, k, p, q = 7000, 2600, 10, 6, 2
    A = _host(m, n, 0.05, k, 13, torch.float32)
    om = synth.gaussian_panel(n, k + p, 5).numpy()
    monkeypatch.setenv("SAPCA_NO_ROWSORT", "1")            # natural row order: a piece's rows are a contiguous range of the panel
    make = lambda: (sapca.SparsePCABuilder.new().n_components(k).spmm_variant(2).collect_timings(True)
                    .svd_method(SVDMethod.Random(p, q, PIN.QR)).build())
    res = {}
    for overlap in ("1", "0"):
        monkeypatch.setenv("SAPCA_AT_OVERLAP", overlap)
        md = sapca.MultiDevice(make(), [0, 0]).set_omega(om)
        t = md.fit_transform(A)
        pieces = [int(md.member(i).timings().at_sweep_pieces) for i in range(2)]
        assert pieces == ([2, 2] if overlap == "1" else [1, 1]), pieces
        res[overlap] = (md.singular_values_(np.float64), md.components_(np.float64), t)
    np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=1e-5)
    assert O.subspace_angle(res["1"][1], res["0"][1]) < 1e-5
    np.testing.assert_allclose(res["1"][2], res["0"][2], atol=1e-4 * np.abs(res["0"][2]).max())
    one = make().set_omega(om)
    one.fit(A)
    assert O.subspace_angle(res["1"][1], one.components_(np.float64)) < 2e-5


def test_resident_shards_are_uploaded_once_and_fitted_repeatedly():
    """sapca_multi_upload_csr_* + the *_resident calls (SURVEY.md 8f-1 for several devices): the shards cross PCIe once; fits
    with another k / seed and the projection reuse them.  Same numbers as the host-matrix calls of the same sapca_multi."""
    m, n, k, p, q = 8000, 1100, 8, 6, 2
    A = _host(m, n, 0.05, k, 17, torch.float32)
    om = synth.gaussian_panel(n, k + p, 5).numpy()
    make = lambda: sapca.SparsePCABuilder.new().n_components(k).svd_method(SVDMethod.Random(p, q, PIN.QR)).build()
    md = sapca.MultiDevice(make(), [0, 0, 0]).set_omega(om)
    t_host = md.fit_transform(A)
    comps_host = md.components_(np.float64).copy()
    md.upload(A)
    rows = [md.resident_shard(i) for i in range(3)]
    assert rows[0][0] == 0 and sum(r[1] for r in rows) == m and sum(r[2] for r in rows) == A.nnz
    assert all(rows[i][0] + rows[i][1] == rows[i + 1][0] for i in range(2))
    for _ in range(2):   # repeated fits of the resident shards
        t = md.fit_transform_resident()
        np.testing.assert_allclose(t, t_host, atol=1e-5 * np.abs(t_host).max())
        assert O.subspace_angle(md.components_(np.float64), comps_host) < 1e-6
    md.fit_resident()
    t2 = md.transform_resident()
    np.testing.assert_allclose(t2, t_host, atol=2e-4 * np.abs(t_host).max())
    # a host-matrix call drops the resident shards (the members' upload buffers were reused)
    md.fit(A)
    with pytest.raises(L.SapcaError, match="no resident matrix"):
        md.fit_transform_resident()
    # f64 shards on the same sapca_multi; the f32 call on them is refused
    A64 = A.astype(np.float64)
    md64 = sapca.MultiDevice(sapca.SparsePCABuilder.new().n_components(k).svd_method(SVDMethod.Lanczos()).build(), [0, 0])
    md64.upload(A64)
    t64 = md64.fit_transform_resident()
    one = sapca.SparsePCABuilder.new().n_components(k).svd_method(SVDMethod.Lanczos()).build()
    t1 = one.fit_transform(A64)
    np.testing.assert_allclose(md64.singular_values_(np.float64), one.singular_values_(np.float64), rtol=1e-8)
    np.testing.assert_allclose(t64, t1, atol=1e-6 * np.abs(t1).max())


def test_a_failing_resident_shard_releases_its_peers():
    """a member that fails inside a resident fit (here: n_components above its shard's... the mask of another width) ends the
    call for all members without a hang, and the sapca_multi stays usable"""
    m, n, k = 5000, 600, 5
    A = _host(m, n, 0.05, k, 2, torch.float32)
    md = sapca.MultiDevice(sapca.MaskedSparsePCABuilder.new().n_components(k).mask(np.ones(n, bool))
                           .svd_method(SVDMethod.Random(4, 1, PIN.QR)).build(), [0, 0])
    md.upload(A)
    assert md.fit_transform_resident().shape == (m, k)
    # break ONE member only: its mask no longer fits, it fails before its first collective while member 0 enters it
    lib = L.load()
    bad = np.ones(n + 1, np.uint8)
    import ctypes as C
    lib.sapca_set_mask(md.member(1)._h, bad.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(bad.size))
    with pytest.raises(L.SapcaError, match="shard 1.*mask vector length"):
        md.fit_transform_resident()
    good = np.ones(n, np.uint8)
    lib.sapca_set_mask(md.member(1)._h, good.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(good.size))
    assert md.fit_transform_resident().shape == (m, k)
