"""Scene-flow field (`FlowField`) for MI355X: 3-D hash grid (L16 F8, HIP kernel) -> cubic Lagrange reduction over
time -> bias-free MLP 32 -> 64 -> 64 -> 6 (forward and backward flow).  Same constructor, parameter names
(`grid_enc.params`, `mlp.<i>.weight`) and arithmetic as /root/reference/nvsf/nerf/models/flow_field.py:41-133.
The three small dense layers are plain GEMMs and go through torch (rocBLAS/hipBLASLt); their last layer is
initialised N(0, 1e-3) as in the reference (:103)."""
import numpy as np
import torch
import torch.nn as nn

import tinycudann as tcnn
from nvsf.nerf.models.hash_field import lagrange_reduce


class FlowField(nn.Module):
    def __init__(self, input_dim=4, num_layers=3, hidden_dim=64, use_freq=False, num_freqs=6, use_grid=True, num_basis=4,
                 n_levels=16, n_features_per_level=8, base_resolution=32, max_resolution=8192, log2_hashmap_size=18):
        super().__init__()
        if use_freq or not use_grid:
            raise NotImplementedError("FlowField: the reference configuration (hash grid, no frequency embedding) is implemented")
        self.use_freq, self.use_grid = use_freq, use_grid
        per_level_scale = np.exp2(np.log2(max_resolution / base_resolution) / (n_levels - 1))
        self.grid_enc = tcnn.Encoding(n_input_dims=3, encoding_config={
            "otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": n_features_per_level,
            "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution, "per_level_scale": per_level_scale})
        self.n_levels, self.n_features_per_level, self.num_basis = n_levels, n_features_per_level, num_basis
        self.input_dim = self.grid_enc.n_output_dims // num_basis
        layers = []
        for l in range(num_layers):
            last = l == num_layers - 1
            layers.append(nn.Linear(self.input_dim if l == 0 else hidden_dim, 6 if last else hidden_dim, bias=False))
            if not last:
                layers.append(nn.ReLU())
        self.mlp = nn.Sequential(*layers)
        torch.nn.init.normal_(self.mlp[-1].weight.data, 0, 0.001)

    def forward(self, xt):
        """xt: [N, 4] = (x, y, z, t) in [0, 1]; all rows share one t (the reference reads xt[0, 3], :125)."""
        t = xt[0, 3]
        feat = self.grid_enc.encode_columns(xt, (0, 1, 2)).float()
        return self.mlp(lagrange_reduce(feat, t, self.n_levels, self.n_features_per_level, self.num_basis))
