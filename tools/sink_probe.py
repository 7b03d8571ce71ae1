"""Timing of the score sink (hypmi_score_hist) against its HBM bound (4 bytes read per score)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyperion_amd.metrics import TrialHistogram

for ne, nt in ((10_000, 65_536), (50_000, 100_000)):
    g = torch.Generator(device="cuda").manual_seed(1)
    S = torch.randn((ne, nt), device="cuda", generator=g) * 8 - 14
    ke = torch.randint(0, 5000, (ne,), device="cuda", dtype=torch.int32)
    kt = torch.randint(0, 5000, (nt,), device="cuda", dtype=torch.int32)
    tar = (ke[:, None] == kt[None, :]).to(torch.uint8) if ne * nt <= 1e9 else None
    for name, kw in (("none", {}), ("keys", dict(key_enrol=ke, key_test=kt)), ("mask", dict(tar=tar))):
        if name == "mask" and tar is None:
            continue
        h = TrialHistogram(-60, 40, 4096)
        h.add(S, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            h.add(S, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%6d x %6d %-5s %7.3f ms  %6.2f TB/s of scores" % (ne, nt, name, ms, 4.0 * ne * nt / ms / 1e9), flush=True)
