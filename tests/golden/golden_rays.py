"""Fixture generator for the ray generators: runs the reference's own
nvsf/nerf/dataset/dataset_utils.py::get_rays / get_lidar_rays on CPU (import-only dependencies cv2, torch_ema,
trimesh, matplotlib, nvsf.lib are stubbed; neither function uses them) and stores poses, the pixel indices the
functions drew, and the resulting rays."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def _load_dataset_utils():
    for name in ("cv2", "trimesh", "matplotlib", "matplotlib.pyplot", "nvsf.lib", "nvsf.lib.convert", "nvsf.lib.tools"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    ema = types.ModuleType("torch_ema")
    ema.ExponentialMovingAverage = object
    sys.modules.setdefault("torch_ema", ema)
    sys.modules["nvsf.lib"].convert = sys.modules["nvsf.lib.convert"]
    sys.modules["nvsf.lib"].tools = sys.modules["nvsf.lib.tools"]
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    spec = importlib.util.spec_from_file_location("nvsf.nerf.dataset.dataset_utils", os.path.join(REF, "nvsf/nerf/dataset/dataset_utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _pose(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    P = np.eye(4)
    P[:3, :3] = R
    P[:3, 3] = rng.uniform(-0.5, 0.5, 3)
    return P.astype(np.float32)


def gen_rays(mods, out_dir):
    du = _load_dataset_utils()
    rng = np.random.default_rng(21)
    out = {}
    torch.manual_seed(5)
    # LiDAR: KITTI-360 range image 66 x 1030, fov (2.0, 26.9), horizontal (180, 360)  (scripts/preprocess_data.py:22-31, configs)
    pose = torch.from_numpy(_pose(rng))[None]
    r = du.get_lidar_rays(pose, [2.0, 26.9], [180.0, 360.0], 66, 1030, N=4096)
    out.update(lidar_pose=pose[0].numpy(), lidar_inds=r["inds"][0].numpy(), lidar_rays_o=r["rays_o"][0].numpy(), lidar_rays_d=r["rays_d"][0].numpy())
    r = du.get_lidar_rays(pose, [2.0, 26.9], [180.0, 360.0], 66, 1030, N=-1)  # full frame (evaluation)
    sel = np.arange(0, 66 * 1030, 97)
    out.update(lidar_full_sel=sel, lidar_full_rays_d=r["rays_d"][0].numpy()[sel], lidar_full_n=np.int64(r["rays_d"].shape[1]))
    r = du.get_lidar_rays(pose, [2.0, 26.9], [180.0, 360.0], 66, 1030, N=1024, patch_size=[2, 8])  # patch sampling
    out.update(lidar_patch_inds=r["inds"][0].numpy(), lidar_patch_rays_d=r["rays_d"][0].numpy())
    # camera: 376 x 1408 pinhole
    K = torch.tensor([[552.554261, 0, 682.049453], [0, 552.554261, 238.769549], [0, 0, 1]])
    pose_c = torch.from_numpy(_pose(rng))[None]
    r = du.get_rays(pose_c, K, 376, 1408, N=4096)
    out.update(cam_pose=pose_c[0].numpy(), cam_K=K.numpy(), cam_inds=r["inds"][0].numpy(), cam_rays_o=r["rays_o"][0].numpy(),
               cam_rays_d=r["rays_d"][0].numpy())
    np.savez_compressed(os.path.join(out_dir, "rays.npz"), **out)
    print("rays.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


GENERATORS = {"rays": gen_rays}
