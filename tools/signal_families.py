"""Twelve families of synthetic 3 s @ 24 kHz chunks with random parameters and levels, generated on the device: what tools/exact_soak.py soaks
the INT8 exactness pass with and what bench.py's ``hard_inputs`` block measures throughput on (the benchmark's own tone + noise input is
family 0; the noise-free families put whole chunks on the float64 route of csrc/bn_stft_exact.hip)."""
import numpy as np

T, SR = 72000, 24000
FAMILIES = ("tone + noise", "two tones", "white noise", "chirp", "amplitude-modulated tone + 1 % noise", "clipped tone + noise", "harmonic stack + 2 % noise",
            "onset behind digital silence", "sparse impulses + 0.1 % noise", "tone + 0.01 % noise", "brown noise", "square wave + 5 % noise")


def family_batch(torch, kind: int, B: int, g, dev):
    """``[B, T]`` float32 chunks of family ``kind`` (0..11), every chunk with its own frequencies / phases / level (four decades)."""
    t = torch.arange(T, device=dev, dtype=torch.float64) / SR

    def rnd(*shape, lo=0.0, hi=1.0):
        return lo + (hi - lo) * torch.rand(shape, generator=g, device=dev, dtype=torch.float64)

    f = rnd(B, 1, lo=60.0, hi=11500.0)
    tone = torch.sin(2 * np.pi * f * t[None, :] + rnd(B, 1, hi=6.28))
    noise = torch.randn((B, T), generator=g, device=dev, dtype=torch.float64)
    if kind == 0:
        x = rnd(B, 1, hi=1.0) * noise + tone
    elif kind == 1:
        x = tone + rnd(B, 1) * torch.sin(2 * np.pi * rnd(B, 1, lo=60.0, hi=11500.0) * t[None, :])
    elif kind == 2:
        x = noise
    elif kind == 3:
        x = torch.sin(2 * np.pi * (f * t[None, :] + rnd(B, 1, lo=-1500.0, hi=1500.0) * t[None, :] ** 2))
    elif kind == 4:
        x = (1 + 0.9 * torch.sin(2 * np.pi * rnd(B, 1, lo=1.0, hi=40.0) * t[None, :])) * tone + 0.01 * noise
    elif kind == 5:
        x = torch.clamp(3 * (0.3 * noise + tone), -1, 1)
    elif kind == 6:
        x = sum(torch.sin(2 * np.pi * (rnd(B, 1, lo=80.0, hi=400.0)) * h * t[None, :]) / h for h in range(1, 12)) + 0.02 * noise
    elif kind == 7:
        x = torch.where(t[None, :] > rnd(B, 1, hi=2.5), 0.2 * noise + tone, torch.zeros_like(tone))  # onset behind digital silence
    elif kind == 8:
        x = (torch.rand((B, T), generator=g, device=dev) < 2e-3).double() * noise + 1e-3 * noise
    elif kind == 9:
        x = 1e-4 * noise + tone  # almost noise-free
    elif kind == 10:
        x = torch.cumsum(noise, dim=1) / 50.0  # brown noise: strong low frequencies
    else:
        x = 0.05 * noise + torch.sign(tone)
    x = x * 10.0 ** rnd(B, 1, lo=-4.0, hi=0.0)  # any level: the normalisation is scale-free
    return x.to(torch.float32).contiguous()
