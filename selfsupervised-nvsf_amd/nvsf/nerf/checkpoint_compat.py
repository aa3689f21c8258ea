"""Reading the `ema` and `optimizer` entries of a checkpoint written by the reference's Trainer (nvsf/nerf/utils.py:622-648) into
this package's model, and writing the optimiser entry back in the reference's layout.

Both entries are lists ordered by PARAMETER POSITION, not by name, and the positions differ in two ways:
  * the reference's NeRFNetwork registers, before everything else, a `planes_encoder` and a `hash_encoder` it never uses and, last,
    a `unet` (network_dynamic.py:47-65, 192); torch_ema shadows `model.parameters()`, so `ema["shadow_params"]` carries their
    tensors too; the optimiser does not (get_params, network_dynamic.py:335-357);
  * a Planes4D is 6 x n_scales parameters `planes.<scale>.<pair>` [1, C, H, W] there and ONE channel-last parameter `planes_cl`
    here (planes_field.py).
The functions below translate by position: they walk this model's parameters in registration order -- which equals the
reference's for the modules both construct -- expand every `planes_cl` into its planes, skip the unused modules' entries (their
count is known: the unused encoders are constructed with the used ones' configuration; whatever follows the last used entry is
the U-Net's), and check every shape on the way.  A list that already has this package's layout is taken as it is.
"""
import torch

from nvsf.nerf.models.planes_field import Planes4D


def _layout(model):
    """[(parameter, owner Planes4D or None)] in registration order, and the number of reference entries the two unused leading
    encoders occupy (0 for a model without space-time encoders, e.g. NeRFNetworkStatic: such checkpoints are this package's own)."""
    owners = {}
    for mod in model.modules():
        if isinstance(mod, Planes4D):
            owners[mod.planes_cl] = mod
    rows = [(p, owners.get(p)) for p in model.parameters()]
    lead = 0
    pl, hs = getattr(model, "planes_encoder_lidar", None), getattr(model, "hash_encoder_lidar", None)
    if isinstance(pl, Planes4D) and hs is not None:
        lead = len(pl._layout) + len(list(hs.parameters()))
    return rows, lead


def _pack_planes(mod, tensors, like):
    """24 reference tensors [1, C, H, W] -> one flat channel-last tensor shaped like `planes_cl`."""
    flat = torch.empty_like(like)
    for (si, pi, off, C, H, W), t in zip(mod._layout, tensors):
        if tuple(t.shape) != (1, C, H, W):
            raise ValueError(f"planes.{si}.{pi}: {tuple(t.shape)} in the checkpoint, {(1, C, H, W)} in the model")
        flat[off:off + C * H * W] = t.to(flat.device, flat.dtype)[0].permute(1, 2, 0).reshape(-1)
    return flat


def _unpack_planes(mod, flat):
    return [mod._view(flat, si, pi).contiguous() for si, pi, *_ in mod._layout]


def shadows_for_model(shadow_params, model):
    """`ema["shadow_params"]` of a checkpoint -> one tensor per parameter of `model`, in `model.parameters()` order.
    Accepts this package's layout (one tensor per parameter) and the reference's (see the module docstring); raises ValueError
    for anything else."""
    rows, lead = _layout(model)
    shadow_params = list(shadow_params)
    if len(shadow_params) == len(rows) and all(s.numel() == p.numel() for s, (p, _) in zip(shadow_params, rows)):
        return [s.reshape(p.shape) for s, (p, _) in zip(shadow_params, rows)]
    need = sum(len(mod._layout) if mod is not None else 1 for _, mod in rows)
    if lead == 0 or len(shadow_params) < lead + need:
        raise ValueError(f"shadow_params: {len(shadow_params)} tensors; this model has {len(rows)} parameters "
                         f"({lead} + {need} + U-Net entries in the reference's layout)")
    out, at = [], lead
    for p, mod in rows:
        if mod is not None:
            out.append(_pack_planes(mod, shadow_params[at:at + len(mod._layout)], p.detach()))
            at += len(mod._layout)
        else:
            s = shadow_params[at]
            if s.numel() != p.numel():
                raise ValueError(f"shadow_params[{at}]: {tuple(s.shape)} in the checkpoint, {tuple(p.shape)} in the model")
            out.append(s.reshape(p.shape))
            at += 1
    return out


def _group_rows(model, optimizer):
    owners = {}
    for mod in model.modules():
        if isinstance(mod, Planes4D):
            owners[mod.planes_cl] = mod
    return [[(p, owners.get(p)) for p in g["params"]] for g in optimizer.param_groups]


def optimizer_state_for_model(state_dict, model, optimizer):
    """An Adam state_dict in the reference's layout (every plane a parameter of its own) -> the same state for `optimizer`, whose
    groups hold `planes_cl` instead: moments packed channel-last, the (common) step count kept.  A state_dict that already has
    this optimiser's layout is returned unchanged."""
    groups = _group_rows(model, optimizer)
    ref_groups = state_dict["param_groups"]
    if len(ref_groups) != len(groups):
        raise ValueError(f"optimizer: {len(ref_groups)} parameter groups in the checkpoint, {len(groups)} here")
    if all(len(rg["params"]) == len(g) for rg, g in zip(ref_groups, groups)):
        return state_dict
    new_state, new_groups, k = {}, [], 0
    for rg, g in zip(ref_groups, groups):
        ids = list(rg["params"])
        need = sum(len(mod._layout) if mod is not None else 1 for _, mod in g)
        if len(ids) != need:
            raise ValueError(f"optimizer group: {len(ids)} parameters in the checkpoint, {need} expected for this model")
        at, mine = 0, []
        for p, mod in g:
            n = len(mod._layout) if mod is not None else 1
            sts = [state_dict["state"].get(i) for i in ids[at:at + n]]
            at += n
            if all(st is not None for st in sts):
                if mod is None:
                    new_state[k] = sts[0]
                else:
                    steps = {float(st["step"]) for st in sts}
                    if len(steps) != 1:
                        raise ValueError("optimizer: the planes of one Planes4D carry different step counts")
                    merged = {key: _pack_planes(mod, [st[key] for st in sts], p.detach()) for key in ("exp_avg", "exp_avg_sq")}
                    merged["step"] = sts[0]["step"]
                    new_state[k] = merged
            elif any(st is not None for st in sts):
                raise ValueError("optimizer: only some planes of one Planes4D have a state")
            mine.append(k)
            k += 1
        new_groups.append(dict(rg, params=mine))
    return {"state": new_state, "param_groups": new_groups}


def optimizer_state_in_reference_layout(state_dict, model, optimizer):
    """The inverse of optimizer_state_for_model: `planes_cl` entries expanded into one entry per plane ([1, C, H, W] moments), ids
    renumbered -- the `optimizer` entry a reference Trainer loads (utils.py:728-733)."""
    groups = _group_rows(model, optimizer)
    new_state, new_groups, k = {}, [], 0
    for sg, g in zip(state_dict["param_groups"], groups):
        mine = []
        for i, (p, mod) in zip(sg["params"], g):
            st = state_dict["state"].get(i)
            n = len(mod._layout) if mod is not None else 1
            if st is not None:
                if mod is None:
                    new_state[k] = st
                else:
                    m1, m2 = _unpack_planes(mod, st["exp_avg"]), _unpack_planes(mod, st["exp_avg_sq"])
                    for j in range(n):
                        new_state[k + j] = {"step": st["step"].clone() if torch.is_tensor(st["step"]) else st["step"], "exp_avg": m1[j], "exp_avg_sq": m2[j]}
            mine += list(range(k, k + n))
            k += n
        new_groups.append(dict(sg, params=mine))
    return {"state": new_state, "param_groups": new_groups}
