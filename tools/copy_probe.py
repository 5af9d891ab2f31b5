import sys; sys.path.insert(0,'single-algebra_amd/python')
import torch, sapca
e=sapca.SparsePCABuilder.new().build()
for nb in (1<<28, 1<<30, 1<<31):
    print(nb>>20, "MiB:", round(e.measure_copy_gbs(nb, 5),1), "GB/s")
