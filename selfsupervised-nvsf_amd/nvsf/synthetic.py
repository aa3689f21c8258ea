"""Synthetic KITTI-360-shaped workloads (no dataset is available offline).

Host-side numpy only.  Scene constants are those of the reference's shipped configuration
(/root/reference/nvsf/configs/kitti360_1908.txt:5-10, nvsf/scripts/main_nvsf.py:35,167-169) and sensor
models restate the reference's ray generators:
  * LiDAR range image 66 x 1030, fov_up 2.0 deg, fov 26.9 deg, 360 deg horizontal
    (nvsf/scripts/preprocess_data.py:22-31; direction model nvsf/nerf/dataset/dataset_utils.py:369-536:
     beta = -(i - W/2)/W * fov_hoz, alpha = fov_up - j/H * fov, dir = (cos a cos b, cos a sin b, sin a));
  * pinhole camera 376 x 1408 (dataset_utils.py:539-687: pixel centre +0.5, dir = ((i-cx)/fx, (j-cy)/fy, 1)
     normalised, rotated by the pose).
"""
import numpy as np

SCALE = 0.010851959895748291          # configs/kitti360_1908.txt:9
BOUND = 2                             # main_nvsf.py:35
MIN_NEAR = 1.0 * SCALE                # main_nvsf.py:167-168 (min_near = min_near_lidar = 1.0 m)
LIDAR_MAX_DEPTH = 80.0 * SCALE        # main_nvsf.py:169
NUM_FRAMES = 64                       # configs/kitti360_1908.txt:4
LIDAR_HW = (66, 1030)
LIDAR_FOV = (2.0, 26.9, 360.0)        # fov_up, fov, fov_hoz [deg]
CAM_HW = (376, 1408)
CAM_K = (552.554261, 552.554261, 682.049453, 238.769549)  # fx, fy, cx, cy (KITTI-360 perspective cam_00)


def _random_rotation(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def lidar_rays(n, rng):
    """n random pixels of one LiDAR frame -> rays_o, rays_d fp32 [n,3] (one shared origin)."""
    H, W = LIDAR_HW
    fov_up, fov, fov_hoz = LIDAR_FOV
    pix = rng.integers(0, H * W, size=n)
    j, i = pix // W, pix % W
    beta = -(i - W / 2) / W * fov_hoz * np.pi / 180.0
    alpha = (fov_up - j / H * fov) * np.pi / 180.0
    d = np.stack([np.cos(alpha) * np.cos(beta), np.cos(alpha) * np.sin(beta), np.sin(alpha)], -1)
    R = _random_rotation(rng)
    d = d @ R.T
    o = np.broadcast_to(rng.uniform(-0.3, 0.3, size=3), (n, 3))
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)


def camera_rays(n, rng):
    """n random pixels of one camera frame -> rays_o, rays_d fp32 [n,3]."""
    H, W = CAM_HW
    fx, fy, cx, cy = CAM_K
    pix = rng.integers(0, H * W, size=n)
    j, i = pix // W + 0.5, pix % W + 0.5
    d = np.stack([(i - cx) / fx, (j - cy) / fy, np.ones_like(i)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    R = _random_rotation(rng)
    d = d @ R.T
    o = np.broadcast_to(rng.uniform(-0.3, 0.3, size=3), (n, 3))
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)


def _expand_bits(v):
    v = (v * 0x00010001) & 0xFF0000FF
    v = (v * 0x00000101) & 0x0F00F00F
    v = (v * 0x00000011) & 0xC30C30C3
    v = (v * 0x00000005) & 0x49249249
    return v


def boxes_density_grid(rng, cascades=2, H=128, n_boxes=64, bound=BOUND):
    """Procedural density grid [C, H^3] (Morton order within a cascade): union of random boxes, value 1 inside.
    Cascade c covers [-min(2^c, bound), +min(2^c, bound)]^3."""
    lo = rng.uniform(-bound, bound * 0.7, size=(n_boxes, 3))
    size = rng.uniform(0.1, 0.8, size=(n_boxes, 3))
    hi = np.minimum(lo + size, bound)
    grid = np.zeros((cascades, H ** 3), np.float32)
    idx = np.arange(H, dtype=np.uint32)
    X, Y, Z = np.meshgrid(idx, idx, idx, indexing="ij")
    morton = (_expand_bits(X) | (_expand_bits(Y) << 1) | (_expand_bits(Z) << 2)).reshape(-1)
    for c in range(cascades):
        mb = min(2.0 ** c, bound)
        centers = ((idx.astype(np.float64) + 0.5) / H * 2 - 1) * mb
        occ = np.zeros((H, H, H), bool)
        for b in range(n_boxes):
            mx = (centers >= lo[b, 0]) & (centers <= hi[b, 0])
            my = (centers >= lo[b, 1]) & (centers <= hi[b, 1])
            mz = (centers >= lo[b, 2]) & (centers <= hi[b, 2])
            occ |= mx[:, None, None] & my[None, :, None] & mz[None, None, :]
        grid[c, morton] = occ.reshape(-1).astype(np.float32)
    return grid
