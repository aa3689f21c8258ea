"""End-to-end example of the multimodal training loop around the hot path (BASELINE config 4), one process per GPU:

    python tools/train_example.py --root /path/to/kitti360_nvsf --sequence 1908 [--dynamic] [--epochs 6] [--plain]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 tools/train_example.py ...

Data: the reference's on-disk formats (transforms_{seq}_{split}.json + range-image .npy + images; nvsf/nerf/dataset/formats.py).
Without --root a small synthetic data set in those formats is written to a temporary directory first.
Every step renders one frame per rank (frames are sharded over the ranks, nvsf/frame_shard.py), then ONE bucketed RCCL
all-reduce of the gradients, then Adam under the loss scaler (nvsf/nerf/train_step.py).  The loop is the shipped configuration's
(configs/kitti360_1908.txt: grad_loss, use_error_map; trainer.py:1035-1062): epochs over the frames, every second epoch samples 2 x 8 LiDAR
patches from the error map and adds the structural regularisation, every step writes its per-ray losses back into the frame's error maps,
one EMA update per epoch; --plain = random pixels and the default losses only.  Reports loss terms, PSNR, range RMSE, CD / F-score.
"""
import argparse
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))

import numpy as np
import torch


def synthetic_dataset(root, seq, n_frames=8, H=376, W=1408, Hl=66, Wl=1030, seed=0):  # KITTI-360's image / range-image sizes
    """A box-shaped toy scene in the reference's formats: constant-colour images, a range image of a sphere of radius 30 m."""
    from nvsf.nerf.dataset import formats as F
    rng = np.random.default_rng(seed)
    d = os.path.join(root, "train", seq)
    os.makedirs(d, exist_ok=True)
    frames = []
    for i in range(n_frames):
        pose = np.eye(4)
        pose[:3, 3] = [0.2 * i, 0.0, 0.0]
        img = np.full((H, W, 3), 120 + 10 * i, np.uint8)
        pc = np.zeros((Hl, Wl, 3), np.float32)
        pc[..., 1] = rng.random((Hl, Wl)) * 0.5
        pc[..., 2] = 30.0
        pc[rng.random((Hl, Wl)) < 0.1, 2] = 0.0
        np.save(os.path.join(d, f"img_{i:04d}.npy"), img)
        np.save(os.path.join(d, f"pano_{i:04d}.npy"), pc)
        frames.append({"frame_id": 1908 + i, "file_path": f"train/{seq}/img_{i:04d}.npy", "transform_matrix": pose,
                       "lidar_file_path": f"train/{seq}/pano_{i:04d}.npy", "lidar2world": pose})
    K = np.array([[552.55, 0, W / 2], [0, 552.55, H / 2], [0, 0, 1]])
    F.write_transforms(F.transforms_path(root, seq, "train"), w=W, h=H, w_lidar=Wl, h_lidar=Hl, K=K, frame_start=1908, frame_end=1908 + n_frames - 1,
                       num_frames=n_frames, frames=frames)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default=None)
    ap.add_argument("--sequence", default="1908")
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--plain", action="store_true", help="no structural regularisation / error maps / patch epochs")
    ap.add_argument("--num-rays", type=int, default=4096)
    ap.add_argument("--num-steps", type=int, default=768)
    ap.add_argument("--dynamic", action="store_true", help="the reference's space-time model instead of the static hash field")
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from nvsf import frame_shard, synthetic as S
    from nvsf.nerf.dataset.formats import FrameSet
    from nvsf.nerf.train_step import RenderTrainStep
    root = args.root
    if root is None:
        root = os.path.join(tempfile.gettempdir(), "nvsf_synthetic_376x1408")
        if rank == 0:
            synthetic_dataset(root, args.sequence)
        if world > 1:
            dist.barrier()
    scale = S.SCALE if hasattr(S, "SCALE") else 0.010851959895748291
    data = FrameSet(root, args.sequence, "train", scale, num_rays=args.num_rays, num_rays_lidar=args.num_rays, device=dev)
    torch.manual_seed(0)  # identical initial replicas
    if args.dynamic:
        from nvsf.nerf.models.network_dynamic import NeRFNetwork
        model = NeRFNetwork(time_resolution=8, num_frames=data.meta["num_frames"], bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR,
                            lidar_max_depth=S.LIDAR_MAX_DEPTH).to(dev)
    else:
        from nvsf.nerf.models.network_static import NeRFNetworkStatic
        model = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH,
                                  num_frames=data.meta["num_frames"]).to(dev)
    n = len(data)
    per_epoch = max(1, n // world)
    trainer = RenderTrainStep(model, iters=args.epochs * per_epoch, num_steps=args.num_steps, scale=scale, grad_loss=not args.plain,
                              use_error_map=not args.plain, change_patch_size_lidar=(1,) if args.plain else (2, 8))
    if not args.plain:
        trainer.attach_error_maps(data)
    it = 0
    for epoch in range(1, args.epochs + 1):
        sampler = trainer.set_epoch(epoch, data)  # "random" / "patch" (trainer.py:1035-1062)
        perm = np.random.default_rng(epoch).permutation(n)  # the same permutation on every rank
        sums, count = {}, 0
        for k in range(per_epoch):
            frame = int(perm[(k * world + rank) % n])
            loss, parts, n_coll = trainer.step(data.train_batch([frame]))
            for name, v in dict(parts, total=loss).items():
                sums[name] = sums.get(name, 0.0) + float(v)
            count += 1
            it += 1
        trainer.end_epoch()  # one EMA update per epoch (trainer.py:1420-1421)
        if rank == 0:
            line = f"epoch {epoch:3d}  sampler {sampler:6s}  " + "  ".join(f"{k} {v / count:.4f}" for k, v in sums.items()) + f"  all-reduces/step {n_coll}"
            if data.error_map is not None:
                em = data.error_map
                line += f"  error map: {int((em != 1).sum())} of {em.numel()} cells touched, max {float(em.max()):.1f}"
            print(line, flush=True)
    # whole-frame evaluation (Trainer.eval_step / evaluate_one_epoch): every frame rendered with the staged loop, its rays split over the ranks
    from nvsf.nerf.train_step import evaluate_frames
    whole = FrameSet(root, args.sequence, "train", scale, device=dev, training=False)
    res = evaluate_frames(model, whole, args.num_steps, indices=range(min(len(whole), 4)), ema=trainer.ema)
    if rank == 0:
        print(f"evaluation over {res['frames']} frames: loss {res['loss']:.4f}, PSNR {res['psnr']:.2f} dB, range RMSE {res['depth_rmse_m']:.2f} m, "
              f"chamfer distance {res['chamfer_distance']:.3f}, F-score {res['f_score']:.3f}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
