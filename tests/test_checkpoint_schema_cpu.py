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
