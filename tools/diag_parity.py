"""Diagnostic (GPU): where do the fused density kernel and the oracle differ?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O
from nvsf import field_ops as ops, synthetic as S
from test_render_static_gpu import _model, _oracle, _t
dev = torch.device("cuda:0")
for std in (None, 0.1, 0.5):
    m = _model(dev, std)
    rng = np.random.default_rng(5)
    N, T = 64, 80
    o, d = S.lidar_rays(N, rng)
    ref = _oracle(m, o, d, True, T)
    nears = torch.full((N,), float(m.min_near_lidar), device=dev); fars = torch.full((N,), float(m.lidar_max_depth), device=dev)
    enc = m.hash_encoder_lidar
    z, sig, geo = ops.density_uniform(_t(o, dev), _t(d, dev), nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), enc.spec, m.sigma_net.weights_f16())
    geo = geo.cpu().numpy(); g_ref = ref["geo"]
    eq = (geo[..., :15] == g_ref)
    diff = np.abs(geo[..., :15].astype(np.float32) - g_ref.astype(np.float32))
    print(f"std={std}: geo exact {eq.mean():.4f} max|diff| {diff.max():.3e} max|ref| {np.abs(g_ref.astype(np.float32)).max():.3f}")
    s, sr = sig.cpu().numpy(), ref["sigmas"]
    print(f"   sigma exact {(s == sr).mean():.4f} max rel {np.abs(s / sr - 1).max():.3e} sigma range {sr.min():.3e}..{sr.max():.3e}")
    # feature-level: standalone hashgrid kernel vs oracle, and standalone MLP on oracle features
    zz, xyz = O.uniform_samples(o, d, nears.cpu().numpy(), fars.cpu().numpy(), torch.linspace(0, 1, T).numpy(), None, np.array([-2, -2, -2, 2, 2, 2], np.float32))
    x01 = ((xyz.reshape(-1, 3) + np.float32(2)) * np.float32(0.25)).astype(np.float32)
    table = enc.params.detach().cpu().numpy().astype(np.float16)
    feat_r = O.hashgrid_fwd(x01, (0, 1, 2), table, enc.spec)
    feat_g = ops.hashgrid_forward(_t(x01, dev), (0, 1, 2), _t(table, dev), enc.spec).cpu().numpy()
    print(f"   hashgrid standalone exact {(feat_r.view(np.uint16) == feat_g.view(np.uint16)).mean():.4f}")
    w = m.sigma_net.params.detach().cpu().numpy().astype(np.float16)
    h_r, hid_r = O.mlp_fwd(feat_r, w, 32, 32, 1, want_hidden=True)
    h_g = ops.mlp_forward(_t(feat_r, dev), _t(w, dev), m.sigma_net.spec).cpu().numpy()
    print(f"   mlp standalone (same feats) exact {(h_r == h_g).mean():.4f} max|diff| {np.abs(h_r.astype(np.float32) - h_g.astype(np.float32)).max():.3e}")
    # emulate fp32-sequential accumulation to see whether the device matches that better than fp64
    wf = w.astype(np.float32); W0 = wf[:64 * 32].reshape(64, 32); W1 = wf[64 * 32:].reshape(16, 64)
    x = feat_r.astype(np.float32)
    hid32 = np.maximum(x @ W0.T, 0).astype(np.float16).astype(np.float32)
    out32 = (hid32 @ W1.T).astype(np.float16)
    print(f"   numpy fp32 matmul vs device exact {(out32 == h_g).mean():.4f}; vs oracle exact {(out32 == h_r).mean():.4f}")
