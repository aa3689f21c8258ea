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
    assert float(m.planes_encoder_camera.planes[2][4][0, 0, 0, 0]) == 0.25
    groups = m.get_params(1e-2)
    assert len(groups) == 11 and [g["lr"] for g in groups][6] == 1e-3  # flow net at 0.1 x lr (network_dynamic.py:345)
