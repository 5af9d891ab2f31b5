"""The library's RCCL mode with 2 and 4 ranks on ONE GPU, through a stand-in for librccl (-m gpu).

The real RCCL refuses ranks that share a device, so before this file `Comm::RCCL` -- ncclCommInitRank, the duplicate
communicator of the side stream (ncclCommSplit), the two-piece A^T sweep with its first all-reduce on that stream, the
abort path -- had only ever run on one-rank communicators.  tests/fake_rccl/fake_rccl.c implements the eight entry points
csrc/comm.cpp binds by name over POSIX shared memory; the -DSAPCA_DEBUG_SWITCHES build of the library opens it through
SAPCA_RCCL_LIBRARY (the release build reads no such variable and the file is never on a library path).  Everything
else is the product's: the shard logic, every all-reduce site, the vote on the cut, the watchdog contract of INTEGRATION.md §5.
"""
import os
import socket
import subprocess
import threading
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    out = tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "-x", "hip", "-o", str(out),
                           os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.c")])
    return str(out)


def _setup(rank, world, port, fake, extra_env=None):
    """every rank: the debug build of the library bound to the stand-in, a gloo group for the test's own exchanges"""
    import torch.distributed as dist
    from sapca import _lib as L
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SAPCA_RCCL_LIBRARY=fake, SAPCA_NO_ROWSORT="1",
                      FAKE_RCCL_TIMEOUT_S="40")
    os.environ.update(extra_env or {})
    L._lib = L.load_debug()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    return dist, L


def _shard(full, m, n, rank, world):
    import sapca
    from sapca import dist as sdist
    ptr = full[0].cpu().numpy()
    r0, r1 = sdist.shard_rows(ptr, world)[rank]
    lo, hi = int(ptr[r0]), int(ptr[r1])
    return r0, r1, sapca.DeviceCsr((full[0][r0:r1 + 1] - lo).contiguous(), full[1][lo:hi].contiguous(), full[2][lo:hi].contiguous(),
                                   (r1 - r0, n))


def _rccl_init(est, dist, L, world, rank):
    """what sapca.dist.init_comm does once it has decided for RCCL: rank 0's id to everyone, the collective init"""
    import ctypes as C
    uid = [None]
    if rank == 0:
        buf = (C.c_uint8 * 128)()
        assert L.load().sapca_comm_unique_id(buf) == L.OK
        uid[0] = bytes(buf)
    dist.broadcast_object_list(uid, src=0)
    est.comm_init_rank(world, rank, uid[0])


def _worker_overlap(rank, world, port, tmpdir, fake):
    dist, L = _setup(rank, world, port, fake)
    import sapca
    import sapca_oracle as O
    from sapca import synth
    try:
        m, n, k, p, q = 9000, 2600, 10, 6, 2
        full = synth.gapped_csr(m, n, 0.05, k, seed=13, dtype=torch.float32, device="cuda")
        r0, r1, shard = _shard(full, m, n, rank, world)
        om = synth.gaussian_panel(n, k + p, 5).numpy()
        make = lambda: (sapca.SparsePCABuilder.new().n_components(k).spmm_variant(2).collect_timings(True)
                        .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build().set_omega(om))
        res = {}
        for overlap in ("1", "0", None):
            if overlap is None:
                os.environ.pop("SAPCA_AT_OVERLAP")      # the default under RCCL: one piece (opt-in until it has run on > 1 GPU)
            else:
                os.environ["SAPCA_AT_OVERLAP"] = overlap
            est = make()
            _rccl_init(est, dist, L, world, rank)
            # the duplicate communicator of the side stream exists exactly where every rank asked for the two-piece sweep at init
            assert est.comm_has_side_lane() == (overlap == "1")
            t = est.fit_transform(shard)
            tm = est.timings()
            assert int(tm.at_sweep_pieces) == (2 if overlap == "1" else 1), (overlap, int(tm.at_sweep_pieces))
            assert tm.comm_ms >= 0
            res[overlap] = (est.singular_values_(np.float64), est.components_(np.float64), t.cpu().numpy())
            # replicated results are bitwise identical on every rank (the stand-in sums in rank order, like a fixed ring)
            gathered = [None] * world
            dist.all_gather_object(gathered, res[overlap][1].tobytes())
            assert all(g == gathered[0] for g in gathered), "components differ between ranks"
        np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=1e-5)
        assert O.subspace_angle(res["1"][1], res["0"][1]) < 1e-5
        np.testing.assert_allclose(res["1"][2], res["0"][2], atol=1e-4 * np.abs(res["0"][2]).max())
        assert res[None][1].tobytes() == res["0"][1].tobytes()      # default == explicit one piece, bit for bit
        # against the single-rank fit of the whole matrix
        one = make()
        t_ref = one.fit_transform(sapca.DeviceCsr(*full, (m, n))).cpu().numpy()
        np.testing.assert_allclose(res["1"][0], one.singular_values_(np.float64), rtol=2e-5)
        assert O.subspace_angle(res["1"][1], one.components_(np.float64)) < 1e-4
        np.testing.assert_allclose(res["1"][2], t_ref[r0:r1], atol=2e-3 * np.abs(t_ref).max())
        # Lanczos: the per-step vector all-reduce through the same communicator
        lz = sapca.SparsePCABuilder.new().n_components(k).svd_method(sapca.SVDMethod.Lanczos()).build()
        _rccl_init(lz, dist, L, world, rank)
        lz.fit(shard)
        lz1 = sapca.SparsePCABuilder.new().n_components(k).svd_method(sapca.SVDMethod.Lanczos()).build()
        lz1.fit(sapca.DeviceCsr(*full, (m, n)))
        np.testing.assert_allclose(lz.singular_values_(np.float64), lz1.singular_values_(np.float64), rtol=1e-5)
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_rccl_mode_with_several_ranks_overlapped_and_not(tmp_path, fake_rccl, world):
    mp.spawn(_worker_overlap, args=(world, _free_port(), str(tmp_path), fake_rccl), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _worker_abort(rank, world, port, tmpdir, fake):
    """Rank 1 fails outside any collective and never joins the one its peer waits in.  INTEGRATION.md §5: the failing rank
    aborts its communicator; a watchdog on the peer sees the asynchronous error and calls sapca_comm_abort from its own
    thread while the fit thread sits inside ncclAllReduce; the fit returns SAPCA_ERR_COMM; a new init brings both back."""
    dist, L = _setup(rank, world, port, fake, {"SAPCA_AT_OVERLAP": "1"})
    import sapca
    from sapca import synth
    try:
        m, n, k, p, q = 6000, 900, 8, 8, 2
        full = synth.gapped_csr(m, n, 0.05, k, seed=23, dtype=torch.float32, device="cuda")
        r0, r1, shard = _shard(full, m, n, rank, world)
        sm = sapca.SVDMethod.Random(p, q)
        est = sapca.SparsePCABuilder.new().n_components(k).svd_method(sm).build()
        _rccl_init(est, dist, L, world, rank)
        if rank == 1:
            with pytest.raises(L.SapcaError):          # k = 0: refused before any collective
                bad = sapca.SparsePCABuilder.new().n_components(0).svd_method(sm).build()
                bad.fit(shard)
            time.sleep(0.5)                            # (the peer is inside its first all-reduce by now)
            est.comm_abort()
            err = None
        else:
            stop = threading.Event()
            seen = []

            def watchdog():
                while not stop.is_set():
                    e = est.comm_async_error()
                    if e != 0:
                        seen.append(e)
                        est.comm_abort()               # from another thread, while the fit thread is inside the collective
                        return
                    time.sleep(0.01)
            th = threading.Thread(target=watchdog)
            th.start()
            t0 = time.time()
            with pytest.raises(L.SapcaError) as e:
                est.fit_transform(shard)
            stop.set()
            th.join()
            err = e.value
            assert err.status == L.ERR_COMM, err
            assert seen and time.time() - t0 < 20, (seen, time.time() - t0)
            with pytest.raises(L.SapcaError, match="communicator aborted"):   # and every later collective fails at once
                est.fit(shard)
        dist.barrier()
        # a new communicator: both ranks fit again
        _rccl_init(est, dist, L, world, rank)
        t = est.fit_transform(shard)
        assert bool(torch.isfinite(t).all())
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_a_rank_that_fails_outside_a_collective_and_the_watchdog_abort(tmp_path, fake_rccl):
    world = 2
    mp.spawn(_worker_abort, args=(world, _free_port(), str(tmp_path), fake_rccl), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _worker_split_init(rank, world, port, tmpdir, fake):
    """ncclCommInitRank fails on rank 1 only (the hook fails it after everyone has joined).  Rank 0's own ncclCommInitRank
    returns success -- a communicator whose peer is gone -- and the library finds out inside sapca_comm_init_rank, in the
    collectives it runs there (the duplicate for the side stream, the agreement on it): BOTH ranks get SAPCA_ERR_COMM, which
    is the outcome sapca.dist.init_comm can act on (all failed alike -> the callback transport)."""
    dist, L = _setup(rank, world, port, fake, {"FAKE_RCCL_FAIL_INIT_RANK": "1", "FAKE_RCCL_FAILFAST": "1"})
    import sapca
    from sapca import dist as sdist
    from sapca import synth
    try:
        m, n, k = 5000, 700, 6
        full = synth.gapped_csr(m, n, 0.05, k, seed=7, dtype=torch.float32, device="cuda")
        r0, r1, shard = _shard(full, m, n, rank, world)
        est = sapca.SparsePCABuilder.new().n_components(k).svd_method(sapca.SVDMethod.Random(6, 2)).build()
        with pytest.raises(L.SapcaError) as e:
            _rccl_init(est, dist, L, world, rank)
        assert e.value.status == L.ERR_COMM
        assert ("ncclCommInitRank" in str(e.value)) == (rank == 1), str(e.value)
        outcome = [None] * world
        dist.all_gather_object(outcome, str(e.value))
        assert all(outcome)
        est.comm_set_callback(world, rank, sdist.torch_allreduce_callback(None, True))   # (drops whatever RCCL state is left)
        t = est.fit_transform(shard)
        assert bool(torch.isfinite(t).all())
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_split_comm_init_outcome(tmp_path, fake_rccl):
    world = 2
    mp.spawn(_worker_split_init, args=(world, _free_port(), str(tmp_path), fake_rccl), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _worker_no_side_lane(rank, world, port, tmpdir, fake):
    """ncclCommSplit fails on ONE rank: the ranks agree at init that nobody has a side lane (a rank that kept its duplicate
    would sweep A^T in two pieces and issue collectives its peer never joins)."""
    dist, L = _setup(rank, world, port, fake, {"FAKE_RCCL_NO_SPLIT_RANK": "1", "SAPCA_AT_OVERLAP": "1"})
    import sapca
    from sapca import synth
    try:
        m, n, k, p, q = 9000, 2600, 10, 6, 2
        full = synth.gapped_csr(m, n, 0.05, k, seed=13, dtype=torch.float32, device="cuda")
        r0, r1, shard = _shard(full, m, n, rank, world)
        est = (sapca.SparsePCABuilder.new().n_components(k).spmm_variant(2).collect_timings(True)
               .svd_method(sapca.SVDMethod.Random(p, q)).build())
        _rccl_init(est, dist, L, world, rank)
        assert not est.comm_has_side_lane()
        t = est.fit_transform(shard)
        assert int(est.timings().at_sweep_pieces) == 1 and bool(torch.isfinite(t).all())
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_side_lane_is_agreed_across_ranks(tmp_path, fake_rccl):
    world = 2
    mp.spawn(_worker_no_side_lane, args=(world, _free_port(), str(tmp_path), fake_rccl), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
