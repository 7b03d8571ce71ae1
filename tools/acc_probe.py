import sys; sys.path.insert(0, "/root/repo")
import torch
from hyperion_amd import engine
g = torch.Generator(device="cuda").manual_seed(1)
n1, n2, kp = 4096, 4096, 152
for name, rnd in (("fp16-exact inputs", True), ("fp32 inputs", False)):
    G1 = torch.randn((n1, kp), device="cuda", generator=g) * 3
    G2 = torch.randn((n2, kp), device="cuda", generator=g) * 0.7
    G1[:, kp-2] = torch.randn(n1, device="cuda", generator=g) * 500; G1[:, kp-1] = 1
    G2[:, kp-2] = 1; G2[:, kp-1] = torch.randn(n2, device="cuda", generator=g) * 500
    if rnd:
        G1, G2 = G1.half().float(), G2.half().float()
    ref = G1.double() @ G2.double().T
    bound = G1.double().abs() @ G2.double().abs().T
    for fmt in ("pair", "triple", False):
        S = engine.score(G1, G2, split=fmt).double()
        e = (S - ref) / bound
        print("%-18s %-7s max %.3e  mean(bias) %+.3e  rms %.3e   (2^-24 = 5.96e-8)" % (name, fmt, e.abs().max(), e.mean(), e.pow(2).mean().sqrt()))
