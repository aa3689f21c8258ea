"""oracle/torch_losses.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain-PyTorch (CPU) restatement of two pieces of the reference Trainer that the shipped configuration switches on
(/root/reference/nvsf/configs/kitti360_1908.txt:13-14 `grad_loss`, `use_error_map`):

  * structural_grad_loss   the `grad_loss` branch of the structural regularisation on LiDAR patches,
                           /root/reference/nvsf/nerf/trainer.py:296-470 (manual differences or --sobel_grad; criterion from
                           main_nvsf.py:204-221 `--depth_grad_loss`, the cosine form included);
  * error_map_update       the error-map write-back of trainer.py:552-583 (LiDAR) / 586-617 (camera).

Only tests/ import this module; the product path is csrc/losses.hip (nvsf_lidar_grad_loss_fwd/_bwd, nvsf_error_map_update).
PARITY STATUS: restated from the reference's source text (the Trainer method cannot be run in isolation: it needs the CUDA-only
model); pinned by nothing of the reference's own -- "parity unpinned", like the a1-a9 oracle.
"""
import torch


def _diff_padded(x, dim):
    """x[i] - x[i + 1] along `dim`, the last difference repeated so that the size is kept (trainer.py:340-343, 381-384, 395-404)."""
    n = x.shape[dim]
    d = x.narrow(dim, 0, n - 1) - x.narrow(dim, 1, n - 1)
    return torch.cat([d, d.narrow(dim, n - 2, 1)], dim=dim)


def _criterion(name, scale):
    return {"l1": torch.nn.L1Loss(reduction="none"), "mse": torch.nn.MSELoss(reduction="none"),
            "smoothl1": torch.nn.SmoothL1Loss(reduction="none", beta=0.1),
            "huber": torch.nn.HuberLoss(reduction="none", delta=0.2 * scale)}[name]  # main_nvsf.py:204-209


def _sobel(x):
    """F.conv2d with the two 3 x 3 kernels of trainer.py:317-328 / 368-380, padding 1, on [P, pH, pW] patches -> (grad_x, grad_y)."""
    import torch.nn.functional as F
    kx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=torch.float32)[None, None]
    ky = torch.tensor([[-1, -2, -1], [0, 0, 0], [1, 2, 1]], dtype=torch.float32)[None, None]
    x4 = x[:, None]
    return F.conv2d(x4, kx.to(x), padding=1)[:, 0], F.conv2d(x4, ky.to(x), padding=1)[:, 0]


def structural_grad_loss(pred_depth, gt_depth, gt_raydrop, pano_inds, pano_range, patch, scale, criterion="l1", alpha_grad=0.1, sobel=False):
    """pred_depth / gt_depth / gt_raydrop [N] (masked ranges in scene units), pano_inds [N] int64 = h * W + w, pano_range [H, W]
    (range x scale of the whole frame), patch = (pH, pW).  Returns the scalar added to the loss (trainer.py:456-458)."""
    pH, pW = patch
    H, W = pano_range.shape
    as_patches = lambda v: v.reshape(-1, pH, pW)                       # [P, pH, pW] (the reference adds a channel axis of size 1)
    pred = as_patches(pred_depth) / scale                              # :314-318
    gt = as_patches(gt_depth) / scale                                  # :364-367
    drop = as_patches(gt_raydrop)                                      # :368-370
    if sobel:                                                          # :316-328, :367-380
        (pred_gx, pred_gy), (gt_gx, gt_gy) = _sobel(pred), _sobel(gt)
    else:
        pred_gx, pred_gy = _diff_padded(pred, 2), _diff_padded(pred, 1)    # :340-343
        gt_gx, gt_gy = _diff_padded(gt, 2), _diff_padded(gt, 1)            # :381-384
    h, w = as_patches(pano_inds // W), as_patches(pano_inds % W)       # :387-394
    frame_gx = _diff_padded(pano_range, 1) / scale                     # :397-399: (a - b) / scale, padded
    frame_gy = _diff_padded(pano_range, 0) / scale                     # :400-402
    frame_gxx = _diff_padded(frame_gx.abs(), 1)                        # :404-405
    frame_gyy = _diff_padded(frame_gy.abs(), 0)                        # :406-407
    flat_x = (frame_gxx[h, w].abs() < 0.05).to(pred.dtype)             # :419-431 (the two gathers select the value at the pixel)
    flat_y = (frame_gyy[h, w].abs() < 0.05).to(pred.dtype)
    mx, my = drop * flat_x, drop * flat_y                              # :434-435
    if criterion == "cos":                                             # :442-452, criterion = torch.nn.CosineSimilarity() (main_nvsf.py:211)
        cos = torch.nn.CosineSimilarity()
        P = pred.shape[0]
        cx = cos((pred_gx * mx).reshape(P, -1), (gt_gx * mx).reshape(P, -1))
        cy = cos((pred_gy * my).reshape(P, -1), (gt_gy * my).reshape(P, -1))
        lx = (1 - cx).reshape(P, 1, 1).expand(P, pH, pW)
        ly = (1 - cy).reshape(P, 1, 1).expand(P, pH, pW)
        return (alpha_grad * (lx + ly)).sum()                          # :458, :462
    crit = _criterion(criterion, scale)
    lx = crit(pred_gx * mx, gt_gx * mx)                                # :449-450
    ly = crit(pred_gy * my, gt_gy * my)
    return (alpha_grad * (lx + ly)).sum()                              # :453, :458


def error_map_update(error_map, ray_loss, pixel_inds, H, W):
    """error_map [eH, eW] (modified in place and returned), ray_loss [N], pixel_inds [N]: trainer.py:566-583.  Rays that fall into
    one cell: written one after the other in ray order (each from the OLD map value), i.e. the last one stays."""
    eH, eW = error_map.shape
    err = ray_loss.detach().float()
    err = (err - err.min()) / (err.max() - err.min() + torch.finfo().eps)    # :570
    err = err * (1e3 - 1) + 1                                                # :573-574
    sh, sw = eH / H, eW / W                                                  # :578 (Python floats)
    ch = (pixel_inds // W * sh).long()                                       # :579
    cw = (pixel_inds % W * sw).long()                                        # :580
    new = 0.1 * error_map[ch, cw] + 0.9 * err                                # :583
    for k in range(new.shape[0]):                                            # :585, sequentially
        error_map[ch[k], cw[k]] = new[k]
    return error_map
