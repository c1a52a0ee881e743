"""Host-side mirror of the reference's DJL block `nerfBlock` and of the trainer calls made on it
(reference nerf/nerfBlock.scala:146-157, nerf/runNerf.scala:94-106,181-204), on top of libnerfhip.so.

The reference's toolchain (Scala 2.11 / JVM / DJL) is absent from this environment, so this mirror is
Python over ctypes; INTEGRATION.md shows the JNI binding a maintainer would add on the Scala side.  The
same four tensors go in (raysO, raysD, bounds, viewDir, all [n][.]) and the same (n,3) RGB comes out.
Host arrays are numpy float32; the *_dev methods take device addresses (or torch CUDA tensors, used purely
as device-memory owners) and enqueue on the handle's stream.
"""
import ctypes as C

import numpy as np

from . import _lib
from .config import NerfConfig


def _ptr(x):
    """address of a numpy array (host) or of anything with data_ptr() (torch tensor, device) or an int"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


class NerfBlock:
    """nerfBlock(config) + the trainer state DJL keeps for it (parameters, Adam moments, step)."""

    def __init__(self, config):
        if isinstance(config, dict):
            config = NerfConfig.from_dict(config)
        self.config = config
        self.N = config.NSamples
        self.NI = config.NImportance if config.useHierarchical else 0
        self.S = self.N + self.NI
        self._lib = _lib.load()
        self._h = C.c_void_p()
        cc = config.to_c()
        rc = self._lib.nerfhip_create(C.byref(cc), C.byref(self._h))
        if rc != 0:
            msg = self._lib.nerfhip_last_error(None)
            self._h = None
            raise _lib.NerfHipError(f"nerfhip_create failed (code {rc}): {msg.decode() if msg else '?'}")

    # ---- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._lib.nerfhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        _lib.check(self._lib, self._h, rc, what)

    def set_stream(self, stream_ptr):
        self._chk(self._lib.nerfhip_set_stream(self._h, stream_ptr), "set_stream")

    def synchronize(self):
        self._chk(self._lib.nerfhip_synchronize(self._h), "synchronize")

    # ---- parameters (flat fp32, DJL child order)
    def param_count(self):
        return int(self._lib.nerfhip_param_count(self._h))

    def set_params(self, flat):
        flat = _f32(flat, (self.param_count(),))
        self._chk(self._lib.nerfhip_set_params(self._h, _ptr(flat), flat.size), "set_params")

    def get_params(self):
        out = np.empty(self.param_count(), np.float32)
        self._chk(self._lib.nerfhip_get_params(self._h, _ptr(out), out.size), "get_params")
        return out

    def init_params(self, seed=0):
        self._chk(self._lib.nerfhip_init_params(self._h, seed), "init_params")

    def get_opt_state(self):
        n = self.param_count()
        m = np.empty(n, np.float32); v = np.empty(n, np.float32); step = C.c_int64()
        self._chk(self._lib.nerfhip_get_opt_state(self._h, _ptr(m), _ptr(v), n, C.byref(step)), "get_opt_state")
        return m, v, step.value

    def set_opt_state(self, m, v, step):
        n = self.param_count()
        m = _f32(m, (n,)); v = _f32(v, (n,))
        self._chk(self._lib.nerfhip_set_opt_state(self._h, _ptr(m), _ptr(v), n, int(step)), "set_opt_state")

    # ---- checkpoints (the host owns the files; the reference only ever saves weights, runNerf.scala:125-131)
    def save_checkpoint(self, path):
        """weights + Adam moments + update count as one .npz (loadable with numpy, no pickle): true resume."""
        m, v, step = self.get_opt_state()
        np.savez(path, params=self.get_params(), adam_m=m, adam_v=v, step=np.int64(step),
                 layout=np.array("flat fp32, DJL child order (include/nerfhip.h)"))

    def load_checkpoint(self, path, weights_only=False):
        with np.load(path, allow_pickle=False) as z:
            self.set_params(z["params"])
            if not weights_only:
                self.set_opt_state(z["adam_m"], z["adam_v"], int(z["step"]))

    # ---- forward (host arrays)
    def _aux(self, n, rand, debug, training, keep):
        if not rand and not debug:
            return None, {}
        a = _lib.Aux()
        shapes = {"t_rand": (n, self.N), "u_rand": (n, max(self.NI, 1)), "noise_c": (n, self.N), "noise_f": (n, self.S)}
        for k, shp in shapes.items():
            v = (rand or {}).get(k)
            if v is not None:
                v = _f32(v, shp)
                keep.append(v)
                setattr(a, k, _ptr(v))
        out = {}
        if debug:
            hier = self.config.useHierarchical
            out["z_coarse"] = np.zeros((n, self.N), np.float32)
            out["z_fine"] = np.zeros((n, self.S), np.float32)
            out["w_fine"] = np.zeros((n, self.S), np.float32)
            out["sigma_f"] = np.zeros((n, self.S), np.float32)
            out["craw_f"] = np.zeros((n, self.S, 3), np.float32)
            out["depth"] = np.zeros(n, np.float32)
            out["acc"] = np.zeros(n, np.float32)
            if hier:
                out["w_coarse"] = np.zeros((n, self.N), np.float32)
                out["sigma_c"] = np.zeros((n, self.N), np.float32)
                if training:
                    out["craw_c"] = np.zeros((n, self.N, 3), np.float32)
            for k, v in out.items():
                setattr(a, k, _ptr(v))
        return a, out

    def forward(self, rays_o, rays_d, bounds, view_dir, training=False, rand=None, debug=False):
        """nerfBlock.forwardInternal(NDList(raysO, raysD, bounds, viewDir), training).
        Returns (fineRgb, coarseRgb or None[, debug dict])."""
        rays_o = _f32(rays_o); n = rays_o.shape[0]
        rays_d = _f32(rays_d, (n, 3)); bounds = _f32(bounds, (n, 2)); view_dir = _f32(view_dir, (n, 3))
        keep = []
        aux, dbg = self._aux(n, rand, debug, training, keep)
        auxp = C.byref(aux) if aux is not None else None
        fine = np.empty((n, 3), np.float32)
        coarse = None
        if training:
            coarse = np.empty((n, 3), np.float32) if self.config.useHierarchical else None
            rc = self._lib.nerfhip_forward_train(self._h, _ptr(rays_o), _ptr(rays_d), _ptr(bounds), _ptr(view_dir), n,
                                                 _ptr(fine), _ptr(coarse), auxp)
            self._chk(rc, "forward_train")
        else:
            rc = self._lib.nerfhip_render(self._h, _ptr(rays_o), _ptr(rays_d), _ptr(bounds), _ptr(view_dir), n,
                                          _ptr(fine), auxp)
            self._chk(rc, "render")
        return (fine, coarse, dbg) if debug else (fine, coarse)

    def evaluate(self, rays_o, rays_d, bounds, view_dir):
        """trainer.evaluate(netInput).get(0) (runNerf.scala:194)."""
        return self.forward(rays_o, rays_d, bounds, view_dir, training=False)[0]

    def render_u8(self, rays_o, rays_d, bounds, view_dir):
        rays_o = _f32(rays_o); n = rays_o.shape[0]
        rays_d = _f32(rays_d, (n, 3)); bounds = _f32(bounds, (n, 2)); view_dir = _f32(view_dir, (n, 3))
        out = np.empty((n, 3), np.uint8)
        rc = self._lib.nerfhip_render_u8(self._h, _ptr(rays_o), _ptr(rays_d), _ptr(bounds), _ptr(view_dir), n, _ptr(out))
        self._chk(rc, "render_u8")
        return out

    def render_to_image(self, rays_o, rays_d, bounds):
        """runNerf.renderToImage (runNerf.scala:181-204) for inputs shaped (P, H, W, .): the reference
        evaluates one image ROW per call; here the whole image goes down in one call, pixel (j, i) stays at
        [j, i] (row-major), viewDir = raysD (:192).  Returns uint8 (P, H, W, 3)."""
        rays_o = _f32(rays_o); P, H, W = rays_o.shape[:3]
        flat = lambda a, k: _f32(a, (P, H, W, k)).reshape(-1, k)
        o, d, b = flat(rays_o, 3), flat(rays_d, 3), flat(bounds, 2)
        return self.render_u8(o, d, b, d).reshape(P, H, W, 3)

    def render_pose(self, H, W, focal, c2w, near, far, ndc=False):
        """One image from its camera pose, rays generated on the device (nerfhip_render_pose_u8): uint8 (H, W, 3)."""
        c2w = _f32(np.asarray(c2w)[:3, :4], (3, 4))
        out = np.empty((H, W, 3), np.uint8)
        rc = self._lib.nerfhip_render_pose_u8(self._h, int(H), int(W), float(focal), _ptr(c2w), float(near), float(far),
                                              int(ndc), _ptr(out))
        self._chk(rc, "render_pose_u8")
        return out

    def render_pose_rows(self, H, W, focal, c2w, near, far, ndc, row0, row1):
        """Rows [row0, row1) of that image (nerfhip_render_pose_rows_u8): one rank's block of a frame whose rows are
        sharded over the GPUs; uint8 (row1 - row0, W, 3)."""
        c2w = _f32(np.asarray(c2w)[:3, :4], (3, 4))
        out = np.empty((max(row1 - row0, 0), W, 3), np.uint8)
        rc = self._lib.nerfhip_render_pose_rows_u8(self._h, int(H), int(W), float(focal), _ptr(c2w), float(near),
                                                   float(far), int(ndc), int(row0), int(row1), _ptr(out))
        self._chk(rc, "render_pose_rows_u8")
        return out

    # ---- training (host arrays)
    def train_backward(self, rays_o, rays_d, bounds, view_dir, label, rand=None, debug=False):
        """trainer.forward + calculateLoss + collector.backward (runNerf.scala:100-103).
        Returns (loss_fine, loss_coarse[, debug])."""
        rays_o = _f32(rays_o); n = rays_o.shape[0]
        rays_d = _f32(rays_d, (n, 3)); bounds = _f32(bounds, (n, 2)); view_dir = _f32(view_dir, (n, 3))
        label = _f32(label, (n, 3))
        keep = []
        aux, dbg = self._aux(n, rand, debug, True, keep)
        loss = np.zeros(2, np.float32)
        rc = self._lib.nerfhip_train_backward(self._h, _ptr(rays_o), _ptr(rays_d), _ptr(bounds), _ptr(view_dir),
                                              _ptr(label), n, _ptr(loss), C.byref(aux) if aux is not None else None)
        self._chk(rc, "train_backward")
        return (float(loss[0]), float(loss[1]), dbg) if debug else (float(loss[0]), float(loss[1]))

    def get_grads(self):
        out = np.empty(self.param_count(), np.float32)
        self._chk(self._lib.nerfhip_get_grads(self._h, _ptr(out), out.size), "get_grads")
        return out

    def grad_buffer(self):
        """(device address, count) of the flat fp32 gradient buffer: what data parallelism all-reduces."""
        p = C.c_void_p(); n = C.c_int64()
        self._chk(self._lib.nerfhip_grad_buffer(self._h, C.byref(p), C.byref(n)), "grad_buffer")
        return p.value, n.value

    def zero_grads(self):
        """a rank whose shard of the batch is empty contributes zeros to the all-reduce (nerfhip_zero_grads)"""
        self._chk(self._lib.nerfhip_zero_grads(self._h), "zero_grads")

    def optimizer_step(self, grad_scale=1.0):
        """trainer.step() (runNerf.scala:106)."""
        self._chk(self._lib.nerfhip_optimizer_step(self._h, float(grad_scale)), "optimizer_step")

    def train_step(self, rays_o, rays_d, bounds, view_dir, label):
        rays_o = _f32(rays_o); n = rays_o.shape[0]
        rays_d = _f32(rays_d, (n, 3)); bounds = _f32(bounds, (n, 2)); view_dir = _f32(view_dir, (n, 3))
        label = _f32(label, (n, 3))
        loss = np.zeros(2, np.float32)
        rc = self._lib.nerfhip_train_step(self._h, _ptr(rays_o), _ptr(rays_d), _ptr(bounds), _ptr(view_dir), _ptr(label),
                                          n, _ptr(loss))
        self._chk(rc, "train_step")
        return float(loss[0]), float(loss[1])

    # ---- device-pointer entry points (inputs already resident in HBM; no synchronisation)
    def render_dev(self, rays_o, rays_d, bounds, view_dir, n, rgb_out):
        rc = self._lib.nerfhip_render_dev(self._h, _ptr(rays_o), _ptr(rays_d), _ptr(bounds), _ptr(view_dir), n,
                                          _ptr(rgb_out), None)
        self._chk(rc, "render_dev")

    def render_dev_static(self, rays_o, rays_d, bounds, view_dir, n, rgb_out):
        """render_dev without any allocation: intermediates in the library's static device scratch (n bounded by
        render_static_max_rays(), not by maxRays)"""
        rc = self._lib.nerfhip_render_dev_static(self._h, _ptr(rays_o), _ptr(rays_d), _ptr(bounds), _ptr(view_dir), n,
                                                 _ptr(rgb_out))
        self._chk(rc, "render_dev_static")

    def render_static_max_rays(self):
        return int(self._lib.nerfhip_render_static_max_rays(self._h))

    def train_backward_dev(self, rays_o, rays_d, bounds, view_dir, label, n, loss_host=None):
        rc = self._lib.nerfhip_train_backward_dev(self._h, _ptr(rays_o), _ptr(rays_d), _ptr(bounds), _ptr(view_dir),
                                                  _ptr(label), n, _ptr(loss_host), None)
        self._chk(rc, "train_backward_dev")

    def train_step_dev(self, rays_o, rays_d, bounds, view_dir, label, n, loss_host=None):
        rc = self._lib.nerfhip_train_step_dev(self._h, _ptr(rays_o), _ptr(rays_d), _ptr(bounds), _ptr(view_dir),
                                              _ptr(label), n, _ptr(loss_host))
        self._chk(rc, "train_step_dev")

    # ---- data parallelism, native RCCL behind the C ABI (replaces DJL's multi-device trainer.step, runNerf.scala:95-106)
    @staticmethod
    def dp_unique_id():
        """128 rendezvous bytes made by rank 0 (ncclGetUniqueId); the host carries them to the other ranks."""
        lib = _lib.load()
        buf = (C.c_uint8 * 128)()
        _lib.check(lib, None, lib.nerfhip_dp_unique_id(buf), "dp_unique_id")
        return bytes(buf)

    def dp_init(self, rank, world, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._chk(self._lib.nerfhip_dp_init(self._h, int(rank), int(world), buf), "dp_init")

    def dp_world(self):
        r, w = C.c_int32(), C.c_int32()
        self._chk(self._lib.nerfhip_dp_world(self._h, C.byref(r), C.byref(w)), "dp_world")
        return r.value, w.value

    def dp_broadcast_state(self, root=0):
        self._chk(self._lib.nerfhip_dp_broadcast_state(self._h, int(root)), "dp_broadcast_state")

    def dp_allreduce_grads(self):
        self._chk(self._lib.nerfhip_dp_allreduce_grads(self._h), "dp_allreduce_grads")

    def dp_train_step_dev(self, rays_o, rays_d, bounds, view_dir, label, n, loss_host=None):
        rc = self._lib.nerfhip_dp_train_step_dev(self._h, _ptr(rays_o), _ptr(rays_d), _ptr(bounds), _ptr(view_dir),
                                                 _ptr(label), n, _ptr(loss_host))
        self._chk(rc, "dp_train_step_dev")

    def dp_destroy(self):
        self._chk(self._lib.nerfhip_dp_destroy(self._h), "dp_destroy")

    # ---- single kernels on device pointers
    def k_stratified(self, bounds, t_rand, n, z_out):
        self._chk(self._lib.nerfhip_k_stratified(self._h, _ptr(bounds), _ptr(t_rand), n, _ptr(z_out)), "k_stratified")

    def k_composite_fwd(self, sigma, noise, craw, z, n, S, w, rgb, depth, acc):
        rc = self._lib.nerfhip_k_composite_fwd(self._h, _ptr(sigma), _ptr(noise), _ptr(craw), _ptr(z), n, S, _ptr(w),
                                               _ptr(rgb), _ptr(depth), _ptr(acc))
        self._chk(rc, "k_composite_fwd")

    def k_composite_bwd(self, sigma, noise, craw, z, d_rgb, n, S, d_sigma, d_craw):
        rc = self._lib.nerfhip_k_composite_bwd(self._h, _ptr(sigma), _ptr(noise), _ptr(craw), _ptr(z), _ptr(d_rgb), n, S,
                                               _ptr(d_sigma), _ptr(d_craw))
        self._chk(rc, "k_composite_bwd")

    def k_sample_pdf(self, w, z, u_rand, n, cdf, ind, z_samples, z_fine):
        rc = self._lib.nerfhip_k_sample_pdf(self._h, _ptr(w), _ptr(z), _ptr(u_rand), n, _ptr(cdf), _ptr(ind),
                                            _ptr(z_samples), _ptr(z_fine))
        self._chk(rc, "k_sample_pdf")

    def k_mlp_forward(self, net, rays_o, rays_d, view_dir, z, n, S, sigma, craw):
        rc = self._lib.nerfhip_k_mlp_forward(self._h, net, _ptr(rays_o), _ptr(rays_d), _ptr(view_dir), _ptr(z), n, S,
                                             _ptr(sigma), _ptr(craw))
        self._chk(rc, "k_mlp_forward")

    def k_gen_rays(self, H, W, focal, c2w, near, far, ndc, rays_o, rays_d, bounds):
        c2w = _f32(np.asarray(c2w)[:3, :4], (3, 4))
        rc = self._lib.nerfhip_k_gen_rays(self._h, H, W, float(focal), _ptr(c2w), float(near), float(far), int(ndc),
                                          _ptr(rays_o), _ptr(rays_d), _ptr(bounds))
        self._chk(rc, "k_gen_rays")

    def k_random(self, stream, seed, step, n, out):
        """n values of one of the library's Philox streams (1 perturb, 2 pdf: U(0,1); 3/4 raw noise, 5 init: N(0,1))"""
        self._chk(self._lib.nerfhip_k_random(self._h, int(stream), int(seed), int(step), int(n), _ptr(out)), "k_random")

    def rng_step(self):
        """(step the last forward pass used, step the next one will use)"""
        a, b = C.c_uint32(), C.c_uint32()
        self._chk(self._lib.nerfhip_rng_step(self._h, C.byref(a), C.byref(b)), "rng_step")
        return a.value, b.value

    # ---- profiling
    def set_profiling(self, on=True):
        self._chk(self._lib.nerfhip_set_profiling(self._h, int(on)), "set_profiling")

    def get_profile(self):
        p = _lib.Prof()
        self._chk(self._lib.nerfhip_get_profile(self._h, C.byref(p)), "get_profile")
        return {k: getattr(p, k) for k, _ in _lib.Prof._fields_ if not k.startswith("reserved")}
