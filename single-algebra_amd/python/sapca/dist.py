"""Multi-GPU glue: one process per GPU, rows of the CSR range-partitioned.

The library does the collectives itself (RCCL, resolved at run time) once every rank has the same
128-byte communicator id; this module only distributes that id over an existing
torch.distributed group.  If RCCL cannot be initialised from the library, the all-reduce is routed
through torch.distributed instead (same RCCL underneath with the nccl backend; gloo works too and
is what the CPU tests use).

All-reduce sites inside a fit (sapca/csrc/engine.cpp): column statistics (3n+1 f64, once), the
l x l Gram of every row-sharded panel (f64), and the n x l panel of every A^T sweep plus its
l-vector of column sums.  Nothing else crosses ranks: Omega, the n-side panels and the small SVD
are replicated and bitwise identical on every rank.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

_TYPESTR = {0: "<f4", 1: "<f8"}
_NP = {0: np.float32, 1: np.float64}


class _DevView:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, count, dtype):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": _TYPESTR[int(dtype)],
                                         "data": (int(ptr), False), "version": 2}


def torch_allreduce_callback(group=None, stage_through_host=False):
    """sapca_allreduce_fn backed by torch.distributed.all_reduce.

    stage_through_host=True copies the device buffer to the host and reduces there (gloo): only for
    tests where several ranks share one GPU and RCCL refuses duplicate devices."""
    import torch
    import torch.distributed as dist

    def cb(ctx, buf, count, dtype, stream):
        try:
            if count == 0:
                return 0
            t = torch.as_tensor(_DevView(buf, count, dtype), device="cuda")
            # everything is ordered on the stream the library is running on (it may be the library's
            # own stream, not torch's current one)
            ext = torch.cuda.ExternalStream(int(stream)) if stream else torch.cuda.current_stream()
            with torch.cuda.stream(ext):
                if stage_through_host:
                    ext.synchronize()
                    h = t.cpu()
                    dist.all_reduce(h, group=group)
                    t.copy_(h)
                    ext.synchronize()
                else:
                    dist.all_reduce(t, group=group)
            return 0
        except Exception as e:  # never raise through the C ABI
            print(f"sapca all-reduce callback failed: {e!r}", flush=True)
            return 1
    return cb


def host_allreduce_callback(group=None):
    """The same adapter for HOST buffers (CPU tests of the plumbing with the gloo backend)."""
    import torch
    import torch.distributed as dist

    def cb(ctx, buf, count, dtype, stream):
        try:
            arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_float if dtype == 0 else C.c_double)), shape=(int(count),))
            t = torch.from_numpy(arr)
            dist.all_reduce(t, group=group)
            return 0
        except Exception as e:
            print(f"sapca host all-reduce callback failed: {e!r}", flush=True)
            return 1
    return cb


def init_comm(estimator, group=None, prefer="rccl", stage_through_host=False):
    """Attach `estimator` (a SparsePCA / MaskedSparsePCA) to the ranks of a torch.distributed group.
    Returns "rccl" or "torch" (which transport the library will use).

    ncclCommInitRank is collective: a rank that cannot take part must say so BEFORE anyone enters it.  Every rank
    therefore first reports whether librccl resolves in its process and which device it drives; only if all can and
    no two ranks of a host share a device does rank 0 create the id.  After the collective init every rank reports its
    outcome: RCCL if all succeeded, the torch transport if all failed alike, an error on a split outcome."""
    import socket
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return "none"
    if prefer == "rccl":
        mine = (bool(L.load().sapca_comm_rccl_available()), socket.gethostname(),
                torch.cuda.current_device() if torch.cuda.is_available() else -1)
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=group)
        devices = [(h, d) for _, h, d in everyone]
        if all(ok for ok, _, _ in everyone) and len(set(devices)) == world and all(d >= 0 for _, d in devices):
            uid = [None]
            if rank == 0:
                buf = (C.c_uint8 * 128)()
                if L.load().sapca_comm_unique_id(buf) == L.OK:
                    uid[0] = bytes(buf)
            dist.broadcast_object_list(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            if uid[0] is not None:   # the same value on every rank: all enter the collective init, or none does
                try:
                    estimator.comm_init_rank(world, rank, uid[0])
                    err = None
                except L.SapcaError as e:
                    err = str(e)
                # every rank is out of the collective call at this point (a rank still inside it would hang either way):
                # all succeeded -> RCCL; all failed alike -> the torch transport; a split outcome cannot be repaired
                outcome = [None] * world
                dist.all_gather_object(outcome, err, group=group)
                if all(o is None for o in outcome):
                    return "rccl"
                if any(o is None for o in outcome):
                    raise RuntimeError(f"ncclCommInitRank succeeded on some ranks only: {outcome}")
                if rank == 0:
                    import sys
                    print(f"sapca: RCCL initialisation failed on every rank ({outcome[0]}); using torch.distributed", file=sys.stderr, flush=True)
    estimator.comm_set_callback(world, rank, torch_allreduce_callback(group, stage_through_host))
    return "torch"


def shard_rows(indptr, nparts):
    """nnz-balanced contiguous row ranges of a host CSR: [(r0, r1), ...] (sapca_partition_rows)."""
    from .ops import partition_rows
    b = partition_rows(indptr, nparts).astype(np.int64)
    return [(int(b[i]), int(b[i + 1])) for i in range(nparts)]
