import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from hyperion_amd import _lib
from hyperion_amd.score_norm.adapt_s_norm import gather_stats, topk_rows
torch.manual_seed(0)
rows, nc, k = 4096, 20000, 300
X = torch.randn((rows, nc), device="cuda") * 5 - 8
src = torch.randn((10000, nc), device="cuda")
idx = topk_rows(src, k, 0)
lib = _lib.load()
for rep in range(2):
    gather_stats(X, None, idx, 10000, 1e-5, k)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 2048))()
lib.hypmi_dbg_stamps.restype = ctypes.c_int
lib.hypmi_dbg_stamps(buf)
a = np.array(buf, dtype=np.int64).reshape(2048, 8)
d = np.diff(a[:, :6], axis=1) / 100.0  # wall clock: 100 MHz -> us
print("us: loads+pivot-barrier, pivot->LDS written, setup, walk, drain barrier")
print("median", np.median(d, axis=0).round(2))
print("mean  ", d.mean(axis=0).round(2))
print("total per workgroup (median): %.1f us" % np.median((a[:, 5] - a[:, 0]) / 100.0))
# gap between consecutive workgroups on a CU is not visible here; launch time / (workgroups per CU) for comparison
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); gather_stats(X, None, idx, 10000, 1e-5, k); e1.record(); torch.cuda.synchronize()
print("launch %.3f ms -> %.1f us per workgroup slot" % (e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / (rows / 2 / 256)))
bufw = (ctypes.c_ulonglong * (32 * 2048))()
lib.hypmi_dbg_waves.restype = ctypes.c_int
lib.hypmi_dbg_waves(bufw)
w = np.array(bufw, dtype=np.int64).reshape(2048, 32)
dur = (w[:, :16] - w[:, 16:]) / 100.0
print("walk time per wave (median over workgroups), us:")
print(np.median(dur, axis=0).round(1))
print("by SIMD (wave % 4):", [float(np.median(dur[:, s::4]).round(1)) for s in range(4)])
