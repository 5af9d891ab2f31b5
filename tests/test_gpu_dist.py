"""Row-sharded fit on the GPU with two ranks (-m gpu).  The test box has ONE GPU, so both ranks use it and
the all-reduce goes through the callback transport (gloo on host copies); the device kernels, the shard
logic and the three all-reduce sites are the product's.  RCCL itself is exercised by bench.py --gpus N."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir, method):
    import torch.distributed as dist
    import sapca
    from sapca import dist as sdist
    from sapca import synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        m, n, k, p, q = 6000, 900, 8, 8, 3
        centred = method.startswith("random")
        full = synth.gapped_csr(m, n, 0.05, k, seed=23, centred=centred, dtype=torch.float32, device="cuda")
        ptr = full[0].cpu().numpy()
        r0, r1 = sdist.shard_rows(ptr, world)[rank]
        lo, hi = int(ptr[r0]), int(ptr[r1])
        shard = sapca.DeviceCsr((full[0][r0:r1 + 1] - lo).contiguous(), full[1][lo:hi].contiguous(),
                                full[2][lo:hi].contiguous(), (r1 - r0, n))
        sm = sapca.SVDMethod.Random(p, q) if method.startswith("random") else sapca.SVDMethod.Lanczos()
        masked = method.endswith("masked")
        mask = synth.bernoulli_mask(n, 0.7, 3).numpy() if masked else None
        om = synth.gaussian_panel(int(mask.sum()) if masked else n, k + p, 5).numpy()
        new = (lambda: sapca.MaskedSparsePCABuilder.new().mask(mask)) if masked else sapca.SparsePCABuilder.new
        est = new().n_components(k).svd_method(sm).build().set_omega(om)
        assert sdist.init_comm(est, prefer="torch", stage_through_host=True) == "torch"
        t = est.fit_transform(shard)
        # single-rank reference on the full matrix, same Omega
        ref = new().n_components(k).svd_method(sm).build().set_omega(om)
        t_ref = ref.fit_transform(sapca.DeviceCsr(*full, (m, n)))
        np.testing.assert_allclose(est.singular_values_(np.float64), ref.singular_values_(np.float64), rtol=2e-5)
        np.testing.assert_allclose(est.mean_(np.float64), ref.mean_(np.float64), atol=1e-6)
        np.testing.assert_allclose(est.total_variance_(), ref.total_variance_(), rtol=1e-5)
        import sapca_oracle as O
        assert O.subspace_angle(est.components_(np.float64), ref.components_(np.float64)) < 1e-4
        scale = float(t_ref.abs().max())
        np.testing.assert_allclose(t.cpu().numpy(), t_ref[r0:r1].cpu().numpy(), atol=2e-3 * scale)
        with open(os.path.join(tmpdir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def _worker_straddle(rank, world, port, tmpdir):
    """l = 40 with one shard above the staged-sweep entry floor and one below it: the ranks pick different sweep
    kernels, and (before the panel leading dimension was pinned for sharded fits) different all-reduce sizes."""
    import torch.distributed as dist
    import sapca
    from sapca import _lib as L
    from sapca import dist as sdist
    from sapca import synth
    L._lib = L.load_debug()   # (the entry floor is a switch of the -DSAPCA_DEBUG_SWITCHES build: csrc/switches.h)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SAPCA_TILED_MIN_ENTRIES="400000")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        m, n, k, p, q = 12000, 1500, 30, 10, 2
        full = synth.gapped_csr(m, n, 0.05, k, seed=31, dtype=torch.float32, device="cuda")
        ptr = full[0].cpu().numpy()
        cut = int(0.7 * m)
        r0, r1 = (0, cut) if rank == 0 else (cut, m)
        lo, hi = int(ptr[r0]), int(ptr[r1])
        assert (hi - lo >= 400000) == (rank == 0)      # the shards straddle the floor
        shard = sapca.DeviceCsr((full[0][r0:r1 + 1] - lo).contiguous(), full[1][lo:hi].contiguous(),
                                full[2][lo:hi].contiguous(), (r1 - r0, n))
        sm = sapca.SVDMethod.Random(p, q)
        om = synth.gaussian_panel(n, k + p, 5).numpy()
        est = sapca.SparsePCABuilder.new().n_components(k).svd_method(sm).build().set_omega(om)
        assert sdist.init_comm(est, prefer="torch", stage_through_host=True) == "torch"
        t = est.fit_transform(shard)
        os.environ.pop("SAPCA_TILED_MIN_ENTRIES")
        ref = sapca.SparsePCABuilder.new().n_components(k).svd_method(sm).build().set_omega(om)
        t_ref = ref.fit_transform(sapca.DeviceCsr(*full, (m, n)))
        np.testing.assert_allclose(est.singular_values_(np.float64), ref.singular_values_(np.float64), rtol=2e-5)
        import sapca_oracle as O
        assert O.subspace_angle(est.components_(np.float64), ref.components_(np.float64)) < 1e-4
        scale = float(t_ref.abs().max())
        np.testing.assert_allclose(t.cpu().numpy(), t_ref[r0:r1].cpu().numpy(), atol=2e-3 * scale)
        with open(os.path.join(tmpdir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_either_side_of_the_staged_sweep_floor(tmp_path):
    world = 2
    mp.spawn(_worker_straddle, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.parametrize("method", ["random", "lanczos", "random_masked"])   # (masked: the masked-out columns' sums cross the all-reduce too)
def test_two_rank_row_sharded_fit(tmp_path, method):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), method), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.gpu
def test_rccl_binding_with_a_one_rank_communicator(debug_switches, monkeypatch):
    """One GPU cannot host two RCCL ranks, but a one-rank communicator still exercises the library's own
    RCCL binding (dlopen, ncclCommInitRank by-value id, ncclAllReduce on the library stream) at every
    all-reduce site of a fit; the result must equal the fit without a communicator."""
    import numpy as np
    import torch
    import sapca
    from sapca import synth
    from sapca import _lib as L
    import ctypes as C
    m, n, k, p, q = 5000, 800, 10, 6, 2
    ptr, idx, val = synth.gapped_csr(m, n, 0.05, k, seed=9, dtype=torch.float32, device="cuda")
    x = sapca.DeviceCsr(ptr, idx, val, (m, n))
    om = synth.gaussian_panel(n, k + p, 4).numpy()

    def build():
        return (sapca.SparsePCABuilder.new().n_components(k).random_seed(1)
                .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build().set_omega(om))

    ref = build()
    t_ref = ref.fit_transform(x).cpu().numpy()
    monkeypatch.setenv("SAPCA_COMM_FORCE_RCCL", "1")
    buf = (C.c_uint8 * 128)()
    assert L.load().sapca_comm_unique_id(buf) == L.OK
    est = build()
    est.comm_init_rank(1, 0, bytes(buf))
    t = est.fit_transform(x).cpu().numpy()
    # a fit with a communicator pins the panel leading dimension to 64 (rank-invariant collective sizes) where the
    # single-process fit of l = 16 uses 16: the same arithmetic in a different layout, equal to f32 rounding
    np.testing.assert_allclose(est.singular_values_(np.float64), ref.singular_values_(np.float64), rtol=1e-5)
    np.testing.assert_allclose(t, t_ref, rtol=1e-3, atol=1e-3 * np.abs(t_ref).max())
    # Lanczos path: the per-step vector all-reduce
    lz = (sapca.SparsePCABuilder.new().n_components(k).svd_method(sapca.SVDMethod.Lanczos()).build())
    lz_ref = (sapca.SparsePCABuilder.new().n_components(k).svd_method(sapca.SVDMethod.Lanczos()).build())
    buf2 = (C.c_uint8 * 128)()                 # a communicator id is one-shot
    assert L.load().sapca_comm_unique_id(buf2) == L.OK
    lz.comm_init_rank(1, 0, bytes(buf2))
    lz.fit(x)
    monkeypatch.delenv("SAPCA_COMM_FORCE_RCCL")
    lz_ref.fit(x)
    np.testing.assert_allclose(lz.singular_values_(np.float64), lz_ref.singular_values_(np.float64), rtol=1e-12)


def test_rccl_communicator_can_be_aborted_and_rebuilt(debug_switches, monkeypatch):
    """The built-in RCCL binding on a one-rank communicator (SAPCA_COMM_FORCE_RCCL=1 routes every all-reduce site through
    ncclAllReduce): the side stream gets its own communicator (ncclCommSplit), sapca_comm_abort ends the communicators --
    the next fit fails at once with SAPCA_ERR_COMM instead of entering a collective -- and a new sapca_comm_init_rank
    brings the handle back.  What a host program (or sapca_multi) does when a peer rank has failed."""
    import ctypes as C
    import sapca
    from sapca import _lib as L
    from sapca import synth
    lib = L.load()
    if not lib.sapca_comm_rccl_available():
        pytest.skip("librccl does not resolve")
    monkeypatch.setenv("SAPCA_COMM_FORCE_RCCL", "1")
    m, n, k = 3000, 500, 5
    dev = synth.gapped_csr(m, n, 0.06, k, seed=4, dtype=torch.float32, device="cuda")
    x = sapca.DeviceCsr(*dev, (m, n))
    est = sapca.SparsePCABuilder.new().n_components(k).svd_method(sapca.SVDMethod.Random(5, 2)).build()
    ident = (C.c_uint8 * 128)()
    assert lib.sapca_comm_unique_id(ident) == L.OK
    L.check(est._h, lib.sapca_comm_init_rank(est._h, C.c_uint32(1), C.c_uint32(0), ident))
    state = C.c_int32(7)
    assert lib.sapca_comm_async_error(est._h, C.byref(state)) == L.OK and state.value == 0
    lane = lib.sapca_comm_has_side_lane(est._h)
    assert lane in (0, 1)
    t0 = est.fit_transform(x).cpu().numpy()
    assert lib.sapca_comm_abort(est._h) == L.OK
    assert lib.sapca_comm_async_error(est._h, C.byref(state)) == L.OK and state.value == -1
    with pytest.raises(L.SapcaError, match="communicator aborted") as e:
        est.fit_transform(x)
    assert e.value.status == L.ERR_COMM
    assert lib.sapca_comm_unique_id(ident) == L.OK
    L.check(est._h, lib.sapca_comm_init_rank(est._h, C.c_uint32(1), C.c_uint32(0), ident))
    t1 = est.fit_transform(x).cpu().numpy()
    np.testing.assert_allclose(t1, t0, atol=1e-5 * np.abs(t0).max())
