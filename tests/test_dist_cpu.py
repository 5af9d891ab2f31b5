"""World-size-2 tests on CPU (gloo): the row-sharded formulation of the randomized fit
reproduces the unsharded oracle, the library's row partitioner feeds it, and the all-reduce callback
adapter of sapca.dist moves data correctly through a C function pointer."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import sapca_oracle as O
from sapca import _lib as L
from sapca import dist as sdist
from sapca import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sharded_randomized_fit(A_local, m_global, n, k, p, q, omega):
    """Rank-local restatement of the fit with the three all-reduce sites of engine.cpp."""
    def allreduce(x):
        t = torch.from_numpy(np.ascontiguousarray(x))
        dist.all_reduce(t)
        return t.numpy()

    stats = allreduce(np.concatenate([np.asarray(A_local.sum(0)).ravel(), [A_local.shape[0]]]))   # site 1
    assert int(round(stats[-1])) == m_global
    mu = stats[:n] / m_global
    Q = omega.copy()

    def sweep_a(X):                                    # local rows only, no communication
        return A_local @ X - (mu @ X)[None, :]

    def sweep_at(Y):                                   # site 3: partial Z and 1^T Y summed over ranks
        z = allreduce(A_local.T @ Y)
        s = allreduce(Y.sum(0))
        return z - np.outer(mu, s)

    def qr_rows(Y):                                    # site 2: CholeskyQR2 with an all-reduced Gram
        for _ in range(2):
            G = allreduce(Y.T @ Y)
            R = np.linalg.cholesky(G).T
            Y = Y @ np.linalg.inv(R)
        return Y

    for _ in range(q):
        Y = qr_rows(sweep_a(Q))
        Q, _ = np.linalg.qr(sweep_at(Y))
    Y = qr_rows(sweep_a(Q))
    B = sweep_at(Y).T
    _, s, vt = np.linalg.svd(B, full_matrices=False)
    _, vt = O.svd_flip_v(None, vt[:k])
    return s[:k], vt, mu


def _worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n, k, p, q = 3000, 500, 6, 6, 3
        ptr, idx, val = (x.numpy() for x in synth.gapped_csr(m, n, 0.06, k, seed=17, dtype=torch.float64))
        A = sp.csr_matrix((val, idx.astype(np.int64), ptr), shape=(m, n))
        omega = synth.gaussian_panel(n, k + p, 3).numpy()
        (r0, r1) = sdist.shard_rows(ptr, world)[rank]
        bounds = sdist.shard_rows(ptr, world)
        assert bounds[0][0] == 0 and bounds[-1][1] == m and all(b[1] == c[0] for b, c in zip(bounds, bounds[1:]))
        nnz_per = [ptr[b] - ptr[a] for a, b in bounds]
        assert max(nnz_per) - min(nnz_per) <= 2 * np.diff(ptr).max()          # balanced by stored entries
        s, vt, mu = _sharded_randomized_fit(A[r0:r1], m, n, k, p, q, omega)
        want = O.fit(ptr, idx.astype(np.int64), val, m, n, n_components=k, n_oversamples=p, n_power_iterations=q, omega=omega)
        np.testing.assert_allclose(s, want.singular_values, rtol=1e-9)
        assert O.subspace_angle(vt, want.components) < 1e-8
        np.testing.assert_allclose(mu, want.mean, atol=1e-13)

        # the C-callable all-reduce adapter on host buffers (what sapca_comm_set_callback receives)
        cb = L.ALLREDUCE_FN(sdist.host_allreduce_callback())
        for dtype, npdt in ((0, np.float32), (1, np.float64)):
            buf = np.full(1000, rank + 1, dtype=npdt)
            rc = cb(None, buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size), C.c_int32(dtype), None)
            assert rc == 0 and np.all(buf == sum(range(1, world + 1)))
        with open(os.path.join(tmpdir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_row_sharded_fit_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
