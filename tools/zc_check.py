import torch
class V:
    def __init__(s,p,n): s.__cuda_array_interface__={"shape":(n,),"typestr":"<f4","data":(p,False),"version":2}
a=torch.arange(8,dtype=torch.float32,device="cuda")
for dev in ("cuda","cuda:0",None):
    try:
        t=torch.as_tensor(V(a.data_ptr(),8),device=dev) if dev else torch.as_tensor(V(a.data_ptr(),8))
        print(dev, t.data_ptr()==a.data_ptr(), t.device)
    except Exception as e: print(dev,"ERR",e)
