"""How much of the bound eps(S') = u (48 ||x_t||_2 + 8 max_k S'_tk + 14 S') the float32 STFT actually uses, on volume (GPU only).

csrc/bn_quant_in.h states the bound as empirical: 4 x what tools/stft_error_stats.py needs over 2.5e7 elements against the CPU oracle.
This tool measures the same ratio |S' - S| / eps(S') on the device for as many elements as one cares to wait for — S' from
``bn_stft_mag`` (the float32 kernel of the audio path), S from ``bn_stft_mag_exact`` (the float64 kernel, pinned to the oracle value for
value by tests/test_gpu_sweeps.py) — over signal families with RANDOM parameters per chunk.  The exactness of the INT8 bytes rests on
this ratio staying below 1.

    python tools/guard_margin.py [chunks per family = 2048] [seed = 1]
"""
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd")]
from birdnet_stm32 import _hip  # noqa: E402
from birdnet_stm32.models.runners import stft_device  # noqa: E402

T, W, SR, NFFT = 72000, 256, 24000, 512
HOP = T // W
U = 2.0**-24
KA, KB, KC = 48.0, 8.0, 14.0  # csrc/bn_quant_in.h: kGuardL2, kGuardPeak, kGuardRel


def families(n: int, g: torch.Generator, dev) -> dict:
    t = torch.arange(T, device=dev, dtype=torch.float64)[None, :] / SR
    r = lambda lo, hi: lo + (hi - lo) * torch.rand((n, 1), generator=g, device=dev, dtype=torch.float64)  # noqa: E731
    noise = lambda: torch.randn((n, T), generator=g, device=dev, dtype=torch.float64)  # noqa: E731
    tone = lambda f: torch.sin(2 * math.pi * f * t + r(0, 6.28))  # noqa: E731
    binw = SR / NFFT
    fam = {}
    fam["tone + noise"] = tone(r(50, 11900)) + 10 ** r(-4, 0) * noise()
    fam["pure tone"] = tone(r(20, 11990))
    fam["bin-centred tone"] = tone(binw * torch.randint(1, 255, (n, 1), generator=g, device=dev).double())
    fam["half-bin tone"] = tone(binw * (torch.randint(1, 255, (n, 1), generator=g, device=dev).double() + 0.5))
    fam["two tones"] = tone(r(100, 11000)) + 10 ** r(-3, 0) * tone(r(100, 11000))
    fam["gaussian noise"] = noise()
    fam["uniform noise"] = torch.rand((n, T), generator=g, device=dev, dtype=torch.float64) - 0.5
    f0, f1 = r(50, 6000), r(50, 11900)
    fam["chirp"] = torch.sin(2 * math.pi * (f0 * t + (f1 - f0) / 6.0 * t * t))
    fam["dc + noise"] = 1.0 + 10 ** r(-7, -1) * noise()
    fam["square"] = torch.sign(tone(r(40, 5000)))
    fam["impulses"] = (torch.rand((n, T), generator=g, device=dev) < 10 ** r(-4, -2)).double() * noise()
    h = torch.zeros((n, T), device=dev, dtype=torch.float64)
    fh = r(60, 400)
    for k in range(1, 30):
        h += torch.sin(2 * math.pi * fh * k * t) / k
    fam["harmonics"] = h + 10 ** r(-5, -2) * noise()
    fam["clipped"] = torch.clamp(r(1, 20) * (tone(r(100, 8000)) + 0.3 * noise()), -1, 1)
    fam["low tone"] = tone(r(1, 45))
    fam["near Nyquist"] = tone(r(11900, 11999.5))
    fam["onset"] = torch.where(t > r(0.05, 2.9), tone(r(200, 11000)), torch.zeros_like(t)) + 10 ** r(-8, -3) * noise()
    fam["am"] = (1 + r(0.1, 1.0) * torch.sin(2 * math.pi * r(1, 60) * t)) * tone(r(200, 11000))
    fam["fm"] = torch.sin(2 * math.pi * r(500, 9000) * t + r(1, 200) * torch.sin(2 * math.pi * r(1, 50) * t))
    fam["burst in silence"] = torch.where((t > r(0.2, 1.4)) & (t < r(1.5, 2.8)), noise(), torch.zeros_like(t))
    fam["tone >> noise"] = tone(r(100, 11000)) + 10 ** r(-9, -5) * noise()
    return fam


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    sub = 256
    ctx = _hip.Context(0, sub)
    worst_all, total = 0.0, 0
    print(f"bound eps(S') = u ({KA:g} ||x_t||_2 + {KB:g} max_k S'_tk + {KC:g} S'), u = 2^-24; {n} chunks per family, seed {seed}")
    print("| family | elements | largest |S' - S| / eps(S') | rms |S' - S| / (u ||x_t||_2) | largest |S' - S| / (u ||x_t||_2) |\n|---|---|---|---|---|")
    fam_names = None
    stats = {}
    for b0 in range(0, n, sub):
        nb = min(sub, n - b0)
        fam = families(nb, g, dev)
        fam_names = fam_names or list(fam)
        for name, x in fam.items():
            amp = 10 ** (-3 * torch.rand((nb, 1), generator=g, device=dev, dtype=torch.float64))  # peak between 1e-3 and 1
            x = (x / x.abs().amax(dim=1, keepdim=True).clamp_min(1e-300) * amp).float().contiguous()
            s32 = stft_device(ctx, x, NFFT, HOP, W, normalize=False).double()          # [nb, 257, W]
            s64 = stft_device(ctx, x, NFFT, HOP, W, normalize=False, exact=True).double()
            xp = torch.nn.functional.pad(x.double(), (NFFT // 2, NFFT // 2))
            l2 = xp.unfold(1, NFFT, HOP)[:, :W].pow(2).sum(-1).sqrt()[:, None, :]        # [nb, 1, W]
            peak = s32.amax(dim=1, keepdim=True)
            eps = U * (KA * l2 + KB * peak + KC * s32)
            err = (s32 - s64).abs()
            ratio = torch.where(eps > 0, err / eps.clamp_min(1e-300), torch.where(err > 0, torch.full_like(err, float("inf")), torch.zeros_like(err)))
            rel = torch.where(l2 > 0, err / (U * l2).clamp_min(1e-300), torch.zeros_like(err))
            st = stats.setdefault(name, [0, 0.0, 0.0, 0.0])
            st[0] += err.numel()
            st[1] = max(st[1], float(ratio.max()))
            st[2] += float(rel.pow(2).sum())
            st[3] = max(st[3], float(rel.max()))
    for name in fam_names:
        cnt, worst, ss, wrel = stats[name]
        worst_all = max(worst_all, worst)
        total += cnt
        print(f"| {name} | {cnt:.2e} | {worst:.3f} | {math.sqrt(ss / cnt):.2f} | {wrel:.1f} |", flush=True)
    print(f"\n{total:.3e} elements, largest |S' - S| / eps(S') = {worst_all:.3f} (the bytes are exact while this stays below 1)")


if __name__ == "__main__":
    main()
