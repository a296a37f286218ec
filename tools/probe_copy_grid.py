"""The streaming copy kernel of the link probe by grid size (workgroups per CU), next to hipMemcpyAsync: GB/s read + written."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xgnn_amd import lib
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.zeros(n, dtype=torch.uint8, device="cuda")
r = C.c_double(0)
for k in (0, 1, 2, 3, 4, 6, 8, 12, 16):
    assert lib().ggms_link_probe_copy(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n, 5, k, C.byref(r),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    print("hipMemcpyAsync" if k == 0 else f"kernel, {k} workgroups per CU", round(2 * r.value, 1), "GB/s")
