"""On-disk formats either side of the render path and the per-step batch the Trainer feeds it.

  * `transforms_{sequence}_{split}.json` -- schema written by /root/reference/nvsf/preprocess/kitti360_to_nerf.py:152-187
    and read by nvsf/nerf/dataset/base_dataset.py:59-141: image / range-image sizes, pinhole intrinsics
    (fl_x, fl_y, cx, cy with the reference's fall-backs), frame_start / frame_end / num_frames and, per frame,
    frame_id, file_path, transform_matrix (camera-to-world), lidar_file_path, lidar2world.
  * range image `.npy` -- float [H_lidar, W_lidar, 3]: channel 1 intensity, channel 2 range in metres (channel 0 unused;
    generate_rangeview.py:185-217).  Ground truth per pixel = [raydrop, intensity, range * scale] with
    raydrop = (range != 0) (base_dataset.py:125-141).
  * `FrameSet.collate(index)` -- the training batch of base_dataset.py:303-407: time, camera / LiDAR rays for sampled
    pixels (device kernels, dataset_utils.py of this package) and the ground-truth pixels gathered at the same indices.

Image decoding uses PIL (the reference uses cv2, absent here); decoded arrays can be passed directly instead.
"""
import json
import os

import numpy as np
import torch

from nvsf.nerf.dataset import dataset_utils


def transforms_path(root_path, sequence_id, split):
    """base_dataset.py:60-62: <root>/train/<sequence>/transforms_<sequence>_<split>.json"""
    return os.path.join(root_path, "train", sequence_id, f"transforms_{sequence_id}_{split}.json")


def write_transforms(path, *, w, h, w_lidar, h_lidar, K, frame_start, frame_end, num_frames, frames, aabb_scale=2):
    """Writes the schema of kitti360_to_nerf.py:152-187.  frames: dicts with frame_id, file_path, transform_matrix [4,4],
    lidar_file_path, lidar2world [4,4]."""
    K = np.asarray(K, dtype=np.float64)
    doc = {"w": int(w), "h": int(h), "w_lidar": int(w_lidar), "h_lidar": int(h_lidar), "fl_x": float(K[0, 0]), "fl_y": float(K[1, 1]),
           "cx": float(K[0, 2]), "cy": float(K[1, 2]), "frame_start": int(frame_start), "frame_end": int(frame_end),
           "num_frames": int(num_frames), "num_frames_split": len(frames), "aabb_scale": aabb_scale,
           "frames": [{"frame_id": int(f["frame_id"]), "file_path": str(f["file_path"]),
                       "transform_matrix": np.asarray(f["transform_matrix"], dtype=np.float64).tolist(),
                       "lidar_file_path": str(f["lidar_file_path"]),
                       "lidar2world": np.asarray(f["lidar2world"], dtype=np.float64).tolist()} for f in frames]}
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=2)
    return doc


def load_transforms(path):
    """Parses a transforms file the way base_dataset.py:59-112,142-149 does.  Returns a dict:
    H, W, H_lidar, W_lidar (None when absent), intrinsics [3,3] float64, frame_start, frame_end, num_frames,
    frames (sorted by file_path), poses [F,4,4] fp32 (camera-to-world), poses_lidar [F,4,4] fp32, times [F] float64 =
    (frame_id - frame_start) / (frame_end - frame_start), frame_ids [F]."""
    with open(path, "r") as fh:
        t = json.load(fh)
    out = {"H": int(t["h"]) if "h" in t and "w" in t else None, "W": int(t["w"]) if "h" in t and "w" in t else None,
           "H_lidar": int(t["h_lidar"]) if "h_lidar" in t and "w_lidar" in t else None,
           "W_lidar": int(t["w_lidar"]) if "h_lidar" in t and "w_lidar" in t else None,
           "num_frames": t["num_frames"], "frame_start": t["frame_start"], "frame_end": t["frame_end"]}
    if "nr_returns" in t:
        out["nr_returns"] = int(t["nr_returns"])
    fl_x = t["fl_x"] if "fl_x" in t else t["fl_y"]
    fl_y = t["fl_y"] if "fl_y" in t else t["fl_x"]
    cx = t["cx"] if "cx" in t else out["W"] / 2
    cy = t["cy"] if "cy" in t else out["H"] / 2
    out["intrinsics"] = np.array([[fl_x, 0, cx], [0, fl_y, cy], [0, 0, 1]], dtype=np.float64)
    frames = sorted(t["frames"], key=lambda d: d["file_path"])
    out["frames"] = frames
    out["poses"] = np.stack([np.array(f["transform_matrix"], dtype=np.float32) for f in frames], axis=0)
    out["poses_lidar"] = np.stack([np.array(f["lidar2world"], dtype=np.float32) for f in frames], axis=0)
    span = out["frame_end"] - out["frame_start"]
    out["times"] = np.array([(f["frame_id"] - out["frame_start"]) / span for f in frames], dtype=np.float64)
    out["frame_ids"] = np.array([f["frame_id"] for f in frames])
    return out


def range_image_ground_truth(pc, scale, H_lidar=None, W_lidar=None):
    """pc: float [H, W, 3] range image (channel 1 intensity, channel 2 range) or the path of its .npy file.
    Returns [H, W, 3] = [raydrop, intensity, range * scale] (base_dataset.py:125-141), dtype of the stored array."""
    if isinstance(pc, (str, os.PathLike)):
        pc = np.load(pc)
    H = pc.shape[0] if H_lidar is None else H_lidar
    W = pc.shape[1] if W_lidar is None else W_lidar
    ray_drop = np.where(pc.reshape(-1, 3)[:, 2] == 0.0, 0.0, 1.0).reshape(H, W, 1)
    return np.concatenate([ray_drop, pc[:, :, 1, None], pc[:, :, 2, None] * scale], axis=-1)


def load_image(path, H=None, W=None):
    """RGB(A) image as float32 [H, W, 3/4] in [0, 1] (base_dataset.py:109-120).  `.npy` files hold the decoded array.
    A size mismatch is resolved by box-filter resampling (the reference: cv2.INTER_AREA)."""
    if str(path).endswith(".npy"):
        img = np.load(path)
        img = img.astype(np.float32) / 255 if img.dtype == np.uint8 else img.astype(np.float32)
    else:
        from PIL import Image
        im = Image.open(path)
        im = im.convert("RGBA" if im.mode in ("RGBA", "LA") else "RGB")
        if H is not None and (im.size[1] != H or im.size[0] != W):
            im = im.resize((W, H), Image.BOX)
        img = np.asarray(im, dtype=np.float32) / 255
    return img


def gather_pixels(images, inds):
    """images [B, H, W, C], inds [B, N] (row-major pixel indices) -> [B, N, C]  (base_dataset.py:371-399)."""
    B, C = images.shape[0], images.shape[-1]
    return torch.gather(images.reshape(B, -1, C), 1, torch.stack(C * [inds], -1))


class FrameSet:
    """Frames of one split resident on the device + the per-step batch (`collate`) of the reference's loader."""

    def __init__(self, root_path, sequence_id, split, scale, intrinsics_lidar=(2.0, 26.9), intrinsics_hoz_lidar=(180.0, 360.0),
                 num_rays=4096, num_rays_lidar=4096, patch_size=1, patch_size_lidar=1, device="cuda", training=True,
                 images=None, range_images=None):
        """images / range_images: optional pre-decoded lists (skips file reads, e.g. synthetic data)."""
        t = load_transforms(transforms_path(root_path, sequence_id, split))
        self.meta, self.device, self.training, self.scale = t, torch.device(device), training, scale
        self.H, self.W, self.H_lidar, self.W_lidar = t["H"], t["W"], t["H_lidar"], t["W_lidar"]
        self.intrinsics, self.intrinsics_lidar, self.intrinsics_hoz_lidar = t["intrinsics"], intrinsics_lidar, intrinsics_hoz_lidar
        self.num_rays = num_rays if training else -1
        self.num_rays_lidar = num_rays_lidar if training else -1
        self.patch_size, self.patch_size_lidar = patch_size, patch_size_lidar
        if images is None:
            images = [load_image(os.path.join(root_path, f["file_path"]), self.H, self.W) for f in t["frames"]]
        if range_images is None:
            range_images = [np.load(os.path.join(root_path, f["lidar_file_path"])) for f in t["frames"]]
        dev = self.device
        self.images = torch.from_numpy(np.stack(images, 0).astype(np.float32)).to(dev)
        self.images_lidar = torch.from_numpy(np.stack([range_image_ground_truth(pc, scale, self.H_lidar, self.W_lidar) for pc in range_images],
                                                      0).astype(np.float32)).to(dev)
        self.poses = torch.from_numpy(t["poses"]).to(dev)
        self.poses_lidar = torch.from_numpy(t["poses_lidar"]).to(dev)
        self.times = torch.from_numpy(t["times"].astype(np.float32)).view(-1, 1).to(dev)
        self.frame_ids = torch.from_numpy(t["frame_ids"]).view(-1, 1)
        self.error_map = self.error_map_rgb = None
        self.use_error_map = False  # set per epoch by RenderTrainStep.set_epoch (trainer.py:1056-1059)
        self._em_stats, self._em_owner = {}, {}

    def enable_error_maps(self):
        """The sampler's per-frame error maps as base_dataset.py:243-246 creates them: ones, [F, H_lidar / 2, W_lidar / 2] for the range
        image and [F, H / 4, W / 4] for the camera image (updated by RenderTrainStep.update_error_maps, trainer.py:552-630)."""
        n = len(self)
        self.error_map = torch.ones(n, int(self.H_lidar / 2), int(self.W_lidar / 2), dtype=torch.float32, device=self.device)
        self.error_map_rgb = torch.ones(n, int(self.H / 4), int(self.W / 4), dtype=torch.float32, device=self.device)
        return self

    def error_map_stats(self, device):
        """uint32 [2] = bit patterns of (+inf, 0): the running (min, max) the per-ray loss kernels reduce into."""
        fresh = self._em_stats.get("fresh")
        if fresh is None or fresh.device != device:
            fresh = self._em_stats["fresh"] = torch.tensor([0x7f800000, 0], dtype=torch.int64).to(torch.int32).to(device)
        return fresh.clone()

    def error_map_owner(self, device, n_cells):
        """uint32 [n_cells] scratch of nvsf_error_map_update (zero between calls: the kernel clears what it used)."""
        buf = self._em_owner.get(n_cells)
        if buf is None or buf.device != device:
            buf = self._em_owner[n_cells] = torch.zeros(n_cells, dtype=torch.int32, device=device)
        return buf

    def __len__(self):
        return self.poses_lidar.shape[0]

    def collate(self, index, use_error_map=None):
        """index: list with one frame index (batch_size 1, base_dataset.py:415-421).  Keys as in base_dataset.py:303-407."""
        use_error_map = self.use_error_map if use_error_map is None else use_error_map
        idx = torch.as_tensor(index, dtype=torch.long, device=self.device)
        B = idx.shape[0]
        res = {"index": index, "time": self.times[idx], "frame_id": self.frame_ids[idx.cpu()]}
        em = None if self.error_map is None else self.error_map[idx]
        em_rgb = None if self.error_map_rgb is None else self.error_map_rgb[idx]
        rays = dataset_utils.get_rays(self.poses[idx], self.intrinsics, self.H, self.W, self.num_rays, self.patch_size, em_rgb,
                                      use_error_map and em_rgb is not None)
        res.update({"H": self.H, "W": self.W, "rays_o": rays["rays_o"], "rays_d": rays["rays_d"], "rays_rgb_inds": rays["inds"],
                    "pose": self.poses[idx], "intrinsic_cam": self.intrinsics})
        rl = dataset_utils.get_lidar_rays(self.poses_lidar[idx], self.intrinsics_lidar, self.intrinsics_hoz_lidar, self.H_lidar, self.W_lidar,
                                          self.num_rays_lidar, self.patch_size_lidar, em, use_error_map and em is not None)
        res.update({"H_lidar": self.H_lidar, "W_lidar": self.W_lidar, "rays_o_lidar": rl["rays_o"], "rays_d_lidar": rl["rays_d"],
                    "rays_pano_inds": rl["inds"], "poses_lidar": self.poses_lidar[idx]})
        images, images_lidar = self.images[idx], self.images_lidar[idx]
        if self.training:
            images = gather_pixels(images, rays["inds"])
            images_lidar = gather_pixels(images_lidar, rl["inds"])
        res["images"], res["images_lidar"] = images, images_lidar
        res["pano_frame"] = self.images_lidar[idx]  # the whole ground-truth frame (base_dataset.py:403): the structural regulariser's masks
        return res

    def train_batch(self, index):
        """`collate` reshaped into the argument names of nvsf.nerf.train_step.RenderTrainStep.losses."""
        c = self.collate(index)
        gl = c["images_lidar"]  # [B, N, 3] = raydrop, intensity, range
        return {"rays_o_lidar": c["rays_o_lidar"], "rays_d_lidar": c["rays_d_lidar"], "rays_o": c["rays_o"], "rays_d": c["rays_d"],
                "time": c["time"], "gt_raydrop": gl[..., 0], "gt_intensity": gl[..., 1], "gt_depth": gl[..., 2], "gt_rgb": c["images"][..., :3],
                # what the structural regulariser and the error-map update read (trainer.py:386-391, 552-556, 588-590)
                "index": c["index"], "rays_pano_inds": c["rays_pano_inds"], "rays_rgb_inds": c["rays_rgb_inds"], "pano_frame": c["pano_frame"]}
