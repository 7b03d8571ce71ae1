#!/usr/bin/env python
"""What bounds hypmi_gather_stats: the same launch with (a) real top-k style random lists, (b) every
entry of every list equal (LDS reads broadcast: no bank conflicts, same index stream), (c) consecutive
entries c, c+1, ... (conflict-free across lanes of a group when lists start at different c)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperion_amd.score_norm.adapt_s_norm import gather_stats  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10000)
ap.add_argument("--lists", type=int, default=65536)
ap.add_argument("--nc", type=int, default=20000)
ap.add_argument("--k", type=int, default=300)
a = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn((a.rows, a.nc), device="cuda", generator=g)
k8 = (a.k + 7) // 8


def timed(idx, name):
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gather_stats(X, None, idx, a.lists, 1e-5, a.k)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    el = 1.0 * a.rows * a.lists * a.k
    print("%-28s %8.3f ms  %6.2f elements/clk/CU  index stream %5.2f TB/s"
          % (name, ms, el / (ms * 1e-3) / 256 / 2.4e9, (a.rows / 2) * a.lists * k8 * 16 / (ms * 1e-3) / 1e12))


rnd = torch.randint(0, a.nc, (k8, a.lists, 8), device="cuda", generator=g, dtype=torch.int32)
rnd, _ = torch.sort(rnd.permute(1, 0, 2).reshape(a.lists, -1), dim=1)
rnd = rnd.reshape(a.lists, k8, 8).permute(1, 0, 2).contiguous()
timed(rnd.to(torch.int16), "random sorted lists")
timed(torch.full((k8, a.lists, 8), 1234, device="cuda", dtype=torch.int16), "all entries equal")
start = torch.randint(0, a.nc - 8 * k8, (1, a.lists, 1), device="cuda", generator=g, dtype=torch.int32)
seq = start + (torch.arange(k8, device="cuda", dtype=torch.int32).view(-1, 1, 1) * 8
               + torch.arange(8, device="cuda", dtype=torch.int32).view(1, 1, -1))
timed(seq.to(torch.int16), "consecutive from random c")

# the same random lists, each ordered by LDS bank pair (c mod 32 for 8-byte elements) and rotated by a
# lane-dependent offset, so that the 32 lanes of a half wave read 32 different bank pairs at each step
flat = torch.randint(0, a.nc, (a.lists, 8 * k8), device="cuda", generator=g, dtype=torch.int32)
timed(flat.reshape(a.lists, k8, 8).permute(1, 0, 2).contiguous().to(torch.int16), "random unsorted lists")
order = torch.argsort(flat % 32, dim=1, stable=True)
by_bank = torch.gather(flat, 1, order)[:, :a.k]
rot = (torch.arange(a.lists, device="cuda") % 32) * a.k // 32
pos = (torch.arange(a.k, device="cuda").view(1, -1) + rot.view(-1, 1)) % a.k
rotated = torch.gather(by_bank, 1, pos)
pad = torch.zeros((a.lists, 8 * k8 - a.k), device="cuda", dtype=torch.int32)
rotated = torch.cat([rotated, pad], dim=1)
timed(rotated.reshape(a.lists, k8, 8).permute(1, 0, 2).contiguous().to(torch.int16), "bank-ordered, lane-rotated")
