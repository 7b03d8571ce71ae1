"""Timing of the float64 Gram kernel (hypmi_gram_f64) against its matrix-core and read bounds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hyperion_amd import engine

for n, d in ((1_000_000, 200), (1_000_000, 512), (150_000, 200)):
    x = torch.randn((n, d), dtype=torch.float64, device="cuda")
    engine.gram64(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        engine.gram64(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    nb = (d + 63) // 64
    executed = 2.0 * n * (nb * (nb + 1) // 2) * 64 * 64
    print("%8d x %4d  %8.3f ms  %6.1f executed f64 TFLOP/s  (%5.1f algorithmic, x^T x in full)  %5.2f TB/s of x read once"
          % (n, d, ms, executed / ms / 1e9, 2.0 * n * d * d / ms / 1e9, 8.0 * n * d / ms / 1e9), flush=True)
