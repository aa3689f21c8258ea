"""Checkpoint compatibility (checkpoint part): the parameter / buffer names and shapes of this repo's
NeRFNetwork equal those of the reference's model (schema recorded from the reference by tests/golden/golden_dynamic.py),
except for the three sub-modules the reference constructs but never uses (network_dynamic.py:47-65,192; excluded from its
optimiser at :337-338).  A reference checkpoint therefore loads with strict=False, as the reference itself loads them
(nvsf/nerf/utils.py:682-747)."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_state_dict_schema_matches_reference():
    import golden_dynamic as GD
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_dynamic import NeRFNetwork
    ref = json.load(open(os.path.join(HERE, "golden", "network_state_dict_keys.json")))
    m = NeRFNetwork(min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, **GD.SMALL)
    ours = {k: list(v.shape) for k, v in m.state_dict().items()}
    unused = ("planes_encoder.", "hash_encoder.", "unet.")
    expected = {k: v for k, v in ref.items() if not k.startswith(unused)}
    assert ours == expected
    assert {k.split(".")[0] for k in ref if k.startswith(unused)} == {"planes_encoder", "hash_encoder", "unet"}
    # a checkpoint written by the reference (all 226 entries) loads the way the reference loads it
    fake = {k: torch.full(v, 0.25) for k, v in ref.items()}
    missing, unexpected = m.load_state_dict(fake, strict=False)
    assert missing == [] and all(k.startswith(unused) for k in unexpected)
    assert float(m.hash_encoder_lidar.hash_static.params[0]) == 0.25
    assert float(m.planes_encoder_camera.plane(2, 4)[0, 0, 0, 0]) == 0.25
    groups = m.get_params(1e-2)
    assert len(groups) == 11 and [g["lr"] for g in groups][6] == 1e-3  # flow net at 0.1 x lr (network_dynamic.py:345)


def test_checkpoint_dict_round_trip(tmp_path):
    """save -> load through the reference's checkpoint dict (utils.py:622-747): same keys, parameters, optimiser moments,
    schedule position and step count; a bare state_dict and a `model_only` load behave as the reference's."""
    import golden_dynamic as GD
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_dynamic import NeRFNetwork
    from nvsf.nerf.train_step import RenderTrainStep

    def make(seed):
        torch.manual_seed(seed)
        m = NeRFNetwork(min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, **GD.SMALL)
        return RenderTrainStep(m, lr=1e-2, iters=100, fp16=False)

    a = make(1)
    for p in a.model.parameters():  # one optimiser step with synthetic gradients (no render on CPU)
        p.grad = torch.full_like(p, 1e-3)
    a.opt.step(); a.sched.step(); a.global_step = 7
    path = tmp_path / "run_ep0003.pth"
    a.save_checkpoint(path, epoch=3, stats={"results": [0.5], "checkpoints": [str(path)]})
    disk = torch.load(path, weights_only=False)
    assert set(disk) == {"epoch", "global_step", "stats", "optimizer", "lr_scheduler", "scaler", "model"}
    assert set(a.checkpoint_state(full=False)) == {"epoch", "global_step", "stats", "model"}

    b = make(2)
    missing, unexpected, epoch = b.load_checkpoint(str(path))
    assert missing == [] and unexpected == [] and epoch == 3 and b.global_step == 7
    for (k, v), (k2, v2) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)
    sa, sb = a.opt.state_dict()["state"], b.opt.state_dict()["state"]
    assert sa.keys() == sb.keys() and all(torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) for i in sa)
    assert b.sched.last_epoch == a.sched.last_epoch and b.opt.param_groups[0]["lr"] == a.opt.param_groups[0]["lr"]

    c = make(3)
    assert c.load_checkpoint(disk, model_only=True)[2] == 3 and c.global_step == 0
    c.load_checkpoint(a.model.state_dict())  # bare state_dict: strict
    assert torch.equal(c.model.sigma_net_lidar.params if hasattr(c.model, "sigma_net_lidar") else next(c.model.parameters()),
                       a.model.sigma_net_lidar.params if hasattr(a.model, "sigma_net_lidar") else next(a.model.parameters()))


def _reference_positional_entries(m, opt):
    """`ema["shadow_params"]` and `optimizer` of a checkpoint as the REFERENCE's Trainer writes them for this model configuration
    (position-ordered lists: network_dynamic.py:47-192 registers planes_encoder, hash_encoder first and unet last; every plane is a
    parameter; get_params :335-357 leaves the unused modules out of the optimiser).  Entry i is filled with the value i."""
    from nvsf.nerf.models.planes_field import Planes4D
    owners = {mod.planes_cl: mod for mod in m.modules() if isinstance(mod, Planes4D)}

    def expand(p):
        mod = owners.get(p)
        return [(1, C, H, W) for _, _, _, C, H, W in mod._layout] if mod is not None else [tuple(p.shape)]

    unused = [s for p in m.planes_encoder_lidar.parameters() for s in expand(p)] + [tuple(p.shape) for p in m.hash_encoder_lidar.parameters()]
    used = [s for p in m.parameters() for s in expand(p)]
    unet = [(32, 3, 1, 1), (32,), (64, 32, 3, 3)]  # any number of trailing tensors
    shadows = [torch.full(s, float(i)) for i, s in enumerate(unused + used + unet)]
    state, groups, k = {}, [], 0
    for g in opt.param_groups:
        ids = []
        for p in g["params"]:
            for s in expand(p):
                state[k] = {"step": torch.tensor(9.0), "exp_avg": torch.full(s, float(k)), "exp_avg_sq": torch.full(s, 1000.0 + k)}
                ids.append(k)
                k += 1
        groups.append(dict({key: v for key, v in g.items() if key != "params"}, params=ids))
    return shadows, len(unused), {"state": state, "param_groups": groups}


def test_reference_shaped_ema_and_optimizer_entries_load_by_position():
    """a reference dynamic checkpoint's `ema` (shadows of ALL the reference's parameters: three unused modules, 24 tensors
    per Planes4D) and `optimizer` (24 states per Planes4D) must restore into this model's one-parameter-per-Planes4D layout --
    before, both failed with a count mismatch, the shadows silently stayed at their construction-time (random) clones and
    evaluation under `ema=` overwrote the loaded weights with them."""
    import json
    import golden_dynamic as GD
    from nvsf import synthetic as S
    from nvsf.nerf import checkpoint_compat as compat
    from nvsf.nerf.ema import ExponentialMovingAverage
    from nvsf.nerf.models.network_dynamic import NeRFNetwork
    from nvsf.nerf.train_step import RenderTrainStep
    ref = json.load(open(os.path.join(HERE, "golden", "network_state_dict_keys.json")))
    torch.manual_seed(4)
    m = NeRFNetwork(min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, **GD.SMALL)
    step = RenderTrainStep(m, lr=1e-2, iters=100, fp16=False)
    step.ema = ExponentialMovingAverage(m.parameters(), decay=0.95)  # (constructed only on a HIP device by RenderTrainStep)
    shadows, lead, opt_state = _reference_positional_entries(m, step.opt)
    n_ref_params = sum(1 for k in ref if not k.endswith(("running_mean", "running_var", "num_batches_tracked")) and not k.startswith("aabb"))
    assert len(shadows) - 3 == n_ref_params - sum(1 for k in ref if k.startswith("unet.") and k.endswith(("weight", "bias")))  # same count as the reference minus its U-Net
    ckpt = {"epoch": 2, "global_step": 11, "model": {k: torch.full(v, 0.25) for k, v in ref.items()},
            "ema": {"decay": 0.95, "num_updates": 3, "shadow_params": shadows, "collected_params": None}, "optimizer": opt_state}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # nothing may fail to load
        step.load_checkpoint(ckpt)
    assert step.failed_to_load == [] and step.global_step == 11 and step.ema.num_updates == 3
    # every shadow / moment came from the entry at its REFERENCE position
    from nvsf.nerf.models.planes_field import Planes4D
    owners = {mod.planes_cl: mod for mod in m.modules() if isinstance(mod, Planes4D)}
    at = lead
    for p, s in zip(m.parameters(), step.ema.shadow_params):
        mod = owners.get(p)
        if mod is None:
            assert s.numel() == 0 or bool((s == float(at)).all())
            at += 1
        else:
            for j, (si, pi, *_rest) in enumerate(mod._layout):
                assert bool((mod._view(s, si, pi) == float(at + j)).all())
            at += len(mod._layout)
    k = 0
    for g in step.opt.param_groups:  # the optimiser counts through its groups (get_params order), not through the module tree
        for p in g["params"]:
            mod, st = owners.get(p), step.opt.state[p]
            assert float(st["step"]) == 9.0
            if mod is None:
                assert bool((st["exp_avg"] == float(k)).all()) and bool((st["exp_avg_sq"] == 1000.0 + k).all())
                k += 1
            else:
                for j, (si, pi, *_rest) in enumerate(mod._layout):
                    assert bool((mod._view(st["exp_avg"], si, pi) == float(k + j)).all()) and bool((mod._view(st["exp_avg_sq"], si, pi) == 1000.0 + k + j).all())
                k += len(mod._layout)
    # and back: the optimiser entry in the reference's layout is the one that came in
    back = compat.optimizer_state_in_reference_layout(step.opt.state_dict(), m, step.opt)
    assert [g["params"] for g in back["param_groups"]] == [g["params"] for g in opt_state["param_groups"]]
    assert back["state"].keys() == opt_state["state"].keys()
    assert all(torch.equal(back["state"][i]["exp_avg"], opt_state["state"][i]["exp_avg"]) and
               torch.equal(back["state"][i]["exp_avg_sq"], opt_state["state"][i]["exp_avg_sq"]) for i in opt_state["state"])
    # this package's own layout loads as before
    own = step.checkpoint_state(epoch=1)
    step.load_checkpoint(own)
    assert step.failed_to_load == []


def test_unreadable_ema_entry_restarts_the_average_from_the_loaded_weights():
    import golden_dynamic as GD
    import pytest
    from nvsf import synthetic as S
    from nvsf.nerf.ema import ExponentialMovingAverage
    from nvsf.nerf.models.network_dynamic import NeRFNetwork
    from nvsf.nerf.train_step import RenderTrainStep
    torch.manual_seed(5)
    m = NeRFNetwork(min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, **GD.SMALL)
    step = RenderTrainStep(m, lr=1e-2, iters=100, fp16=False)
    step.ema = ExponentialMovingAverage(m.parameters(), decay=0.95)
    state = {k: torch.full_like(v, 0.125) for k, v in m.state_dict().items()}
    with pytest.warns(UserWarning, match="Failed to load ema"):
        step.load_checkpoint({"model": state, "ema": {"decay": 0.9, "num_updates": 5, "shadow_params": [torch.zeros(3)], "collected_params": None}},
                             model_only=True)
    assert step.failed_to_load == ["ema"] and step.ema.num_updates == 0
    for p, s in zip(m.parameters(), step.ema.shadow_params):
        assert torch.equal(p.detach(), s)  # not the construction-time clones
    assert float(m.sigma_net.params[0]) == 0.125
