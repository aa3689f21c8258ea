"""`nvsf.nerf.raymarching.raymarching` for MI355X: the nine operators of the reference module
(/root/reference/nvsf/nerf/raymarching/raymarching.py:48,82,108,133,164,289,360,460,510) with the same
names, positional signatures, defaults and allocation rules, backed by the HIP kernels of
libnvsf_hip.so through its C ABI (include/nvsf_hip.h) instead of a pybind/ATen extension.

Host-side contract kept from the reference wrappers:
  * float inputs are cast to fp32 under autocast (custom_fwd(cast_inputs=float32));
  * rays are flattened to [N, 3] and made contiguous; CPU inputs are moved to the device;
  * every output tensor is allocated here (torch caching allocator) and handed to the kernel;
  * march_rays_train: outputs zero-initialised, M = N*max_steps or the aligned mean_count,
    result sliced to the aligned live count when force_all_rays or mean_count <= 0 (one D2H read);
  * composite_rays_train has a backward for (sigmas, rgbs); depth receives no gradient.
Differences: kernels run on PyTorch's *current* stream (the reference uses the legacy default stream),
launch failures raise, and the packed sample order is deterministic (ray order).
"""
import warnings

import torch
from torch.autograd import Function

from nvsf import _hip

_fwd32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = torch.amp.custom_bwd(device_type="cuda")


def _dev(t):
    return t if t.is_cuda else t.cuda()


def _rays(t):
    return _dev(t).contiguous().view(-1, 3)


# ------------------------------------------------------------------------------------------------
# utils
# ------------------------------------------------------------------------------------------------
class _near_far_from_aabb(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        """rays_o, rays_d [N,3]; aabb [6] (xmin,ymin,zmin,xmax,ymax,zmax) -> nears, fars [N]."""
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        aabb = _dev(aabb).contiguous()
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        _hip.call("nvsf_near_far_from_aabb", _hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(aabb), N, float(min_near),
                  _hip.ptr(nears), _hip.ptr(fars))
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, radius):
        """Far intersection with the sphere of `radius` as (theta, phi) in [-1,1]^2 -> coords [N,2]."""
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
        _hip.call("nvsf_sph_from_ray", _hip.ptr(rays_o), _hip.ptr(rays_d), float(radius), N, _hip.ptr(coords))
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    @staticmethod
    def forward(ctx, coords):
        """coords int32 [N,3] -> Morton indices int32 [N]."""
        coords = _dev(coords).int().contiguous()
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        _hip.call("nvsf_morton3D", _hip.ptr(coords), N, _hip.ptr(indices))
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    @staticmethod
    def forward(ctx, indices):
        """indices int32 [N] -> coords int32 [N,3]."""
        indices = _dev(indices).int().contiguous()
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        _hip.call("nvsf_morton3D_invert", _hip.ptr(indices), N, _hip.ptr(coords))
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, grid, thresh, bitfield=None):
        """grid fp32 [C, H^3] -> uint8 bitfield [C*H^3/8]; bit i of byte n = grid[8n+i] > thresh."""
        grid = _dev(grid).contiguous()
        C, H3 = grid.shape[0], grid.shape[1]
        N = C * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        _hip.call("nvsf_packbits", _hip.ptr(grid), N, float(thresh), _hip.ptr(bitfield))
        return bitfield


packbits = _packbits.apply


# ------------------------------------------------------------------------------------------------
# training
# ------------------------------------------------------------------------------------------------
# nvsf_march_rays_train_ws reports a bounded inter-workgroup wait that expired through counter[1] < 0 (include/nvsf_hip.h); its
# sample arrays are then incomplete.  Calls that read the counter back anyway check the flag in that same read; calls that do not
# (mean_count > 0: no device->host read, as in the reference) leave an asynchronous copy of the flag here, which the next
# march_rays_train call -- or check_march_status() -- looks at once its copy event has completed.
class _StatusRing:
    """Per device: a few pinned int32 slots with their copy events, allocated once and handed out round-robin (a pinned allocation
    and an event per march_rays_train call were on the hot path).  A slot is reused only after its copy has landed and been looked
    at; when every slot is still in flight the oldest one is waited for."""
    SLOTS = 8

    def __init__(self):
        self.host = torch.empty(self.SLOTS, dtype=torch.int32).pin_memory()
        self.events = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self.busy = [False] * self.SLOTS
        self.next = 0
        self.failed = False

    def note(self, flag):
        i = self.next
        if self.busy[i]:
            self.events[i].synchronize()
            self._look(i)
        self.host[i:i + 1].copy_(flag, non_blocking=True)
        self.events[i].record()
        self.busy[i] = True
        self.next = (i + 1) % self.SLOTS

    def _look(self, i):
        self.failed = self.failed or int(self.host[i]) < 0
        self.busy[i] = False

    def poll(self, wait):
        for i in range(self.SLOTS):
            if self.busy[i]:
                if wait:
                    self.events[i].synchronize()
                if self.events[i].query():
                    self._look(i)
        failed, self.failed = self.failed, False
        return failed


_status_rings = {}  # device index -> _StatusRing


def _note_status(step_counter):
    ring = _status_rings.get(step_counter.device.index)
    if ring is None:
        ring = _status_rings[step_counter.device.index] = _StatusRing()
    ring.note(step_counter[1:2])


def check_march_status(wait=False):
    """Raises NvsfHipError if a march_rays_train launch whose counter was not read back reported an expired wait.
    wait=True synchronises with the outstanding launches first (RenderTrainStep.step does, once per step, BEFORE the optimiser:
    an expired launch hands empty ranges to the compositor, and its step must not be applied)."""
    failed = False
    for ring in _status_rings.values():
        failed = ring.poll(wait) or failed
    if failed:
        raise _hip.NvsfHipError("nvsf_march_rays_train_ws: a bounded inter-workgroup wait expired (counter[1] < 0); "
                                "the samples of that call are invalid")


def march_status_pending():
    return any(any(r.busy) for r in _status_rings.values())


class _march_rays_train(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024, _entry="ws", _spin_limit=0):
        """Occupancy-grid sample generation.  Returns xyzs [M,3], dirs [M,3], deltas [M,2], rays int32 [N,3]
        with rays[n] = (ray id, first sample, sample count).
        `_entry`, `_spin_limit` (not in the reference's signature) are for tests: "ref" selects the three-launch entry point
        (nvsf_march_rays_train_passes), "c" the reference-shaped C entry (nvsf_march_rays_train: the one-launch kernel on a scratch
        block from the stream-ordered pool); a spin limit of 1 forces the one-launch kernel's expiry path."""
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        density_bitfield = _dev(density_bitfield).contiguous()
        nears, fars = _dev(nears).contiguous(), _dev(fars).contiguous()
        dev, dt = rays_o.device, rays_o.dtype
        N = rays_o.shape[0]
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        # The reference zero-fills the three M-row outputs (raymarching.py:235-237).  When the result is sliced to the
        # sample count anyway (force_all_rays / no mean_count), M = N * max_steps rows (134 MB at 4096 x 1024) would be
        # cleared only to be discarded: allocate them uninitialised and clear just the rows the caller keeps beyond the
        # written ones.  Same returned tensors.
        sliced = force_all_rays or mean_count <= 0
        alloc = torch.empty if sliced else torch.zeros
        xyzs = alloc(M, 3, dtype=dt, device=dev)
        dirs = alloc(M, 3, dtype=dt, device=dev)
        deltas = alloc(M, 2, dtype=dt, device=dev)
        # `rays` is written for every ray by a launch that completes; zeros, so that a launch that gives up (counter[1] < 0) on the
        # path without a host read leaves (offset 0, count 0) rows -- rays without samples -- not garbage ranges, for the compositor
        rays = (torch.empty if sliced else torch.zeros)(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        noises = torch.rand(N, dtype=dt, device=dev) if perturb else torch.zeros(N, dtype=dt, device=dev)

        # one-launch form (nvsf_march_rays_train_ws: counts once, ranges from a scanner wave inside the launch) on a scratch tensor of
        # torch's allocator; the three-launch entry point nvsf_march_rays_train_passes gives the same outputs bit for bit
        use_ws = N > 0 and _entry == "ws"
        check_march_status()
        ws_bytes = _hip.march_ws_bytes(N) if use_ws else 0
        workspace = torch.empty(ws_bytes // 8, dtype=torch.int64, device=dev) if use_ws else None

        # the read-back path can recover from an expired wait of the one-launch kernel (below): it needs the counter as it was
        one_launch = use_ws or (N > 0 and _entry == "c")
        before = step_counter.clone() if (one_launch and sliced) else None

        def launch(ws=use_ws):
            if ws:
                _hip.call("nvsf_march_rays_train_ws", _hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(density_bitfield), float(bound),
                          float(dt_gamma), int(max_steps), N, int(C), int(H), M, _hip.ptr(nears), _hip.ptr(fars), _hip.ptr(xyzs),
                          _hip.ptr(dirs), _hip.ptr(deltas), _hip.ptr(rays), _hip.ptr(step_counter), _hip.ptr(noises), _hip.ptr(workspace),
                          ws_bytes, int(_spin_limit))
                return
            _hip.call("nvsf_march_rays_train" if (_entry == "c" and ws is not None) else "nvsf_march_rays_train_passes", _hip.ptr(rays_o),
                      _hip.ptr(rays_d), _hip.ptr(density_bitfield), float(bound),
                      float(dt_gamma), int(max_steps), N, int(C), int(H), M, _hip.ptr(nears), _hip.ptr(fars), _hip.ptr(xyzs),
                      _hip.ptr(dirs), _hip.ptr(deltas), _hip.ptr(rays), _hip.ptr(step_counter), _hip.ptr(noises))
        launch()
        if sliced:
            # the one device->host read of the reference (raymarching.py:277); rays[0,1] = counter value before the call
            m, status, first = torch.stack([step_counter[0], step_counter[1], rays[0, 1]]).tolist() if N > 0 else (int(step_counter[0].item()), 0, 0)
            if status < 0 and before is not None and int(before[1]) >= 0:
                # the one-launch kernel gave up on an inter-workgroup wait (a bounded spin; never seen outside the test that forces
                # it): nothing of this call is usable, but the three-launch entry point -- no waits between
                # workgroups, same outputs bit for bit -- can redo it instead of failing the training run here
                warnings.warn("nvsf_march_rays_train_ws: a bounded inter-workgroup wait expired; the call is repeated through "
                              "nvsf_march_rays_train_passes (three launches)")
                step_counter.copy_(before)
                if not sliced:
                    rays.zero_()
                launch(ws=None)
                m, status, first = torch.stack([step_counter[0], step_counter[1], rays[0, 1]]).tolist()
            if status < 0:
                raise _hip.NvsfHipError("nvsf_march_rays_train_ws: a bounded inter-workgroup wait expired (counter[1] < 0); "
                                        "the samples of this call are invalid")
            if first != 0 or m > M:
                # a pre-loaded counter or rays dropped for lack of room: the written rows are not one prefix -- redo on
                # cleared buffers (never the case for the renderer, which passes a zeroed counter and M = N * max_steps)
                for t in (xyzs, dirs, deltas):
                    t.zero_()
                step_counter[0] -= m - first
                step_counter[1] -= N
                launch()
                written = M
            else:
                written = m
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
            if written < xyzs.shape[0]:
                xyzs[written:].zero_()
                dirs[written:].zero_()
                deltas[written:].zero_()
        elif one_launch:
            _note_status(step_counter)
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class _composite_rays_train(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        """sigmas [M], rgbs [M,3], deltas [M,2], rays [N,3] -> weights_sum [N], depth [N], image [N,3]."""
        sigmas, rgbs, deltas, rays = sigmas.contiguous(), rgbs.contiguous(), deltas.contiguous(), rays.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        depth = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
        _hip.call("nvsf_composite_rays_train_forward", _hip.ptr(sigmas), _hip.ptr(rgbs), _hip.ptr(deltas), _hip.ptr(rays), M, N,
                  float(T_thresh), _hip.ptr(weights_sum), _hip.ptr(depth), _hip.ptr(image))
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = [M, N, T_thresh]
        return weights_sum, depth, image

    @staticmethod
    @_bwd
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        # as in the reference, grad_depth is dropped (raymarching.py:330)
        grad_weights_sum, grad_image = grad_weights_sum.contiguous(), grad_image.contiguous()
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        _hip.call("nvsf_composite_rays_train_backward", _hip.ptr(grad_weights_sum), _hip.ptr(grad_image), _hip.ptr(sigmas),
                  _hip.ptr(rgbs), _hip.ptr(deltas), _hip.ptr(rays), _hip.ptr(weights_sum), _hip.ptr(image), M, N,
                  float(T_thresh), _hip.ptr(grad_sigmas), _hip.ptr(grad_rgbs))
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _composite_rays_train.apply


# ------------------------------------------------------------------------------------------------
# inference
# ------------------------------------------------------------------------------------------------
class _march_rays(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                align=-1, perturb=False, dt_gamma=0, max_steps=1024):
        """At most n_step samples for each of the first n_alive rays of rays_alive, starting at rays_t.
        Returns xyzs, dirs [n_alive*n_step (+pad), 3] and deltas [.., 2]; unfilled slots stay zero."""
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        dev, dt = rays_o.device, rays_o.dtype
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        xyzs = torch.zeros(M, 3, dtype=dt, device=dev)
        dirs = torch.zeros(M, 3, dtype=dt, device=dev)
        deltas = torch.zeros(M, 2, dtype=dt, device=dev)
        noises = torch.rand(n_alive, dtype=dt, device=dev) if perturb else torch.zeros(n_alive, dtype=dt, device=dev)
        _hip.call("nvsf_march_rays", int(n_alive), int(n_step), _hip.ptr(rays_alive), _hip.ptr(rays_t), _hip.ptr(rays_o),
                  _hip.ptr(rays_d), float(bound), float(dt_gamma), int(max_steps), int(C), int(H),
                  _hip.ptr(density_bitfield.contiguous()), _hip.ptr(near), _hip.ptr(far), _hip.ptr(xyzs), _hip.ptr(dirs),
                  _hip.ptr(deltas), _hip.ptr(noises))
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


class _composite_rays(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
        """In-place accumulation into weights_sum, depth, image (and rays_alive / rays_t).  Returns ()."""
        _hip.call("nvsf_composite_rays", int(n_alive), int(n_step), float(T_thresh), _hip.ptr(rays_alive), _hip.ptr(rays_t),
                  _hip.ptr(sigmas.contiguous()), _hip.ptr(rgbs.contiguous()), _hip.ptr(deltas.contiguous()),
                  _hip.ptr(weights_sum), _hip.ptr(depth), _hip.ptr(image))
        return tuple()


composite_rays = _composite_rays.apply
