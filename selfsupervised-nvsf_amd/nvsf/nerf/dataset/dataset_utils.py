"""Ray generation for MI355X: `get_lidar_rays` / `get_rays` with the signatures and result dictionaries
({rays_o, rays_d, inds}) of /root/reference/nvsf/nerf/dataset/dataset_utils.py:369-536, 539-687.

Pixel SELECTION (uniform random, error-map multinomial with jitter, patches) stays a handful of torch index
operations on N elements, as in the reference; ray DIRECTIONS are computed by one HIP kernel from the selected
indices instead of building the full H x W meshgrid and gathering from it on every step (csrc/raygen.hip).
B (frames per call) is 1 in the reference's loader; B > 1 loops."""
import torch

from nvsf import _hip


def _patch_dims(patch_size):
    if isinstance(patch_size, int):
        return patch_size, patch_size
    if len(patch_size) == 1:
        return patch_size[0], patch_size[0]
    return patch_size[0], patch_size[1]


def sample_pixel_indices(B, H, W, N, patch_size=1, error_map=None, use_error_map=False, device="cuda"):
    """[B, N] int64 pixel indices (row-major), drawn exactly like dataset_utils.py:407-503 / 573-667."""
    N = min(N, H * W)
    pH, pW = _patch_dims(patch_size)
    if pH > 1:
        num_patch = N // (pH * pW)
        if use_error_map:
            _, eH, eW = error_map.shape
            assert eH * eW >= num_patch, "Number of sampled pixels should be smaller than the error map size"
            s_w, s_h = W / eW, H / eH
            coarse = torch.multinomial(error_map.reshape(B, eH * eW).to(device), num_patch, replacement=False)
            ix, iy = coarse % eW, coarse // eW
            ix = (ix * s_w + torch.rand(B, num_patch, device=device) * s_w).long().clamp(max=W - pW)[0]
            iy = (iy * s_h + torch.rand(B, num_patch, device=device) * s_h).long().clamp(max=H - pH)[0]
        else:
            ix = torch.randint(0, W - pW, size=[num_patch], device=device)
            iy = torch.randint(0, H - pH, size=[num_patch], device=device)
        top_left = torch.stack([iy, ix], dim=-1)
        pi, pj = torch.meshgrid(torch.arange(pH, device=device), torch.arange(pW, device=device), indexing="ij")
        offsets = torch.stack([pi.reshape(-1), pj.reshape(-1)], dim=-1)
        cells = (top_left.unsqueeze(1) + offsets.unsqueeze(0)).view(-1, 2)
        inds = cells[:, 0] * W + cells[:, 1]
        return inds.expand([B, inds.shape[0]])
    if use_error_map:
        _, eH, eW = error_map.shape
        assert eH * eW >= N, "Number of sampled pixels should be smaller than the error map size"
        sx, sy = W / eW, H / eH
        coarse = torch.multinomial(error_map.reshape(B, eH * eW).to(device), N, replacement=False)
        ix, iy = coarse % eW, coarse // eW
        ix = (ix * sx + torch.rand(B, N, device=device) * sx).long().clamp(max=W - 1)
        iy = (iy * sy + torch.rand(B, N, device=device) * sy).long().clamp(max=H - 1)
        return iy * W + ix
    return torch.randint(0, H * W, size=[N], device=device).expand([B, N])


def _directions(kernel, poses, inds, n_total, extra):
    B = poses.shape[0]
    dev = poses.device
    n = inds.shape[1] if inds is not None else n_total
    rays_o = torch.empty(B, n, 3, dtype=torch.float32, device=dev)
    rays_d = torch.empty(B, n, 3, dtype=torch.float32, device=dev)
    poses = poses.float().contiguous()
    for b in range(B):
        ib = inds[b].long().contiguous() if inds is not None else None
        _hip.call(kernel, _hip.ptr(poses[b]), _hip.ptr(ib), n, *extra, _hip.ptr(rays_o[b]), _hip.ptr(rays_d[b]))
    return rays_o, rays_d


def get_lidar_rays(poses, intrinsics, intrinsics_hoz, H, W, N=-1, patch_size=1, error_map=None, use_error_map=False, inds=None):
    """poses [B,4,4] (sensor-to-world); intrinsics = (fov_up, fov) deg; intrinsics_hoz = (fov_hoz_up, fov_hoz) deg."""
    B, dev = poses.shape[0], poses.device
    if inds is None and N > 0:
        inds = sample_pixel_indices(B, H, W, N, patch_size, error_map, use_error_map, dev)
    fov_up, fov = float(intrinsics[0]), float(intrinsics[1])
    fov_hoz = float(intrinsics_hoz[1])
    rays_o, rays_d = _directions("nvsf_lidar_rays", poses, inds, H * W, (int(H), int(W), fov_up, fov, fov_hoz))
    if inds is None:
        inds = torch.arange(H * W, device=dev).expand([B, H * W])
    return {"rays_o": rays_o, "rays_d": rays_d, "inds": inds}


def get_rays(poses, intrinsics, H, W, N=-1, patch_size=1, error_map=None, use_error_map=False, inds=None):
    """poses [B,4,4] (camera-to-world); intrinsics: 3x3 K (fx, fy, cx, cy taken from it)."""
    B, dev = poses.shape[0], poses.device
    fx, fy, cx, cy = (float(intrinsics[0, 0]), float(intrinsics[1, 1]), float(intrinsics[0, 2]), float(intrinsics[1, 2]))
    if inds is None and N > 0:
        inds = sample_pixel_indices(B, H, W, N, patch_size, error_map, use_error_map, dev)
    rays_o, rays_d = _directions("nvsf_camera_rays", poses, inds, H * W, (int(W), fx, fy, cx, cy))
    if inds is None:
        inds = torch.arange(H * W, device=dev).expand([B, H * W])
    return {"rays_o": rays_o, "rays_d": rays_d, "inds": inds}
