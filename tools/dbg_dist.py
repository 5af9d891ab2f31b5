import os, sys, ctypes as C
sys.path.insert(0, "single-algebra_amd/python"); sys.path.insert(0, "oracle")
import numpy as np, torch, torch.multiprocessing as mp

def w(rank, world, port):
    import torch.distributed as dist, sapca
    from sapca import dist as sdist, _lib as L
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    est = sapca.SparsePCABuilder.new().n_components(4).svd_method(sapca.SVDMethod.Random(4, 1)).build()
    print(rank, sdist.init_comm(est, prefer="torch", stage_through_host=True), flush=True)
    for dt, code in ((torch.float32, 0), (torch.float64, 1)):
        t = torch.full((1000,), float(rank + 1), dtype=dt, device="cuda")
        torch.cuda.synchronize()
        st = L.load().sapca_comm_allreduce(est._h, C.c_void_p(t.data_ptr()), C.c_uint64(1000), C.c_int32(code))
        torch.cuda.synchronize()
        print(rank, dt, st, t[:3].tolist(), flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    mp.spawn(w, args=(2, 29533), nprocs=2, join=True)
