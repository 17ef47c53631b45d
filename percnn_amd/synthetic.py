"""Synthetic initial conditions for benchmarking (no datasets ship with the reference; its
``.mat`` files are Drive-hosted -- DataDrivenModeling/2d_gs_rd/train_2drd.py:604)."""
import torch


def gs_initial_state(shape, seed: int = 0, dtype=torch.float32) -> torch.Tensor:
    """Gray-Scott: u=1, v=0; a centred square/cube (half-width N/10 in 2D, N/8 in 3D) set to
    u=0.5, v=0.25; plus 0.01*randn from ``manual_seed(seed)``.  Returns [1,2,*shape] on CPU."""
    ndim = len(shape)
    h = torch.zeros((1, 2) + tuple(shape), dtype=dtype)
    h[:, 0] = 1.0
    box = tuple(slice(n // 2 - max(1, n // (10 if ndim == 2 else 8)), n // 2 + max(1, n // (10 if ndim == 2 else 8)))
                for n in shape)
    h[(slice(None), 0) + box] = 0.5
    h[(slice(None), 1) + box] = 0.25
    g = torch.Generator().manual_seed(seed)
    return h + 0.01 * torch.randn(h.shape, generator=g, dtype=dtype)


def lo_initial_state(n: int, dtype=torch.float64) -> torch.Tensor:
    """lambda-omega spiral wave: x=(i-N/2)*0.2; u=tanh(R)cos(theta-R), v=tanh(R)sin(theta-R)."""
    x = (torch.arange(n, dtype=torch.float64) - n / 2) * 0.2
    yy, xx = torch.meshgrid(x, x, indexing="ij")
    r = torch.sqrt(xx ** 2 + yy ** 2)
    th = torch.atan2(yy, xx)
    return torch.stack((torch.tanh(r) * torch.cos(th - r), torch.tanh(r) * torch.sin(th - r)))[None].to(dtype)
