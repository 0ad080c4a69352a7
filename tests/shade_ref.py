"""The reference's shading / orientation expressions as one torch function (network_grid.py:81-130, renderer.py:727-746),
used as the yardstick of csrc/shade.hip on the GPU and itself pinned, on the CPU, to tests/golden/shade_ref.npz — the
output of the reference's own NeRFNetwork.forward (tests/golden/make_goldens_from_reference.py)."""
import torch


def safe_normalize(x, eps=1e-20):
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps))


def torch_shade(sigma7, albedo, dirs, rays, rays_o, light_offset, ratio, shading, epsilon=1e-2):
    """sigma7 [7, M] (only the valid columns), albedo [M, 3], dirs [M, 3] raw. Returns color, normal, orient."""
    e = epsilon
    s = sigma7
    n = -torch.stack([0.5 * (s[1] - s[2]) / e, 0.5 * (s[3] - s[4]) / e, 0.5 * (s[5] - s[6]) / e], dim=-1)
    n = torch.nan_to_num(safe_normalize(n))
    ray_id = torch.repeat_interleave(torch.arange(rays.shape[0], device=rays.device), rays[:, 1].long())
    l = safe_normalize(rays_o + light_offset)[ray_id]
    lambertian = ratio + (1 - ratio) * (n * l).sum(-1).clamp(min=0)
    if shading == "textureless":
        color = lambertian.unsqueeze(-1).repeat(1, 3)
    elif shading == "normal":
        color = (n + 1) / 2
    else:
        color = albedo * lambertian.unsqueeze(-1)
    orient = (n * safe_normalize(dirs)).sum(-1).clamp(min=0) ** 2
    return color, n, orient
