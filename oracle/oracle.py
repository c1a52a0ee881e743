"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package nerf_amd never does.  See oracle/nerf_oracle.h for scope and citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_SKIPS = 8


class OrcConfig(C.Structure):
    _fields_ = [
        ("use_dir", C.c_int32), ("use_sh", C.c_int32), ("use_hierarchical", C.c_int32),
        ("pos_l", C.c_int32), ("dir_l", C.c_int32), ("D", C.c_int32), ("W", C.c_int32),
        ("n_skips", C.c_int32), ("skips", C.c_int32 * MAX_SKIPS),
        ("n_samples", C.c_int32), ("n_importance", C.c_int32),
        ("raw_noise_std", C.c_float), ("white_bkgd", C.c_int32), ("lin_disp", C.c_int32),
        ("perturb", C.c_int32), ("lrate", C.c_float), ("lrate_decay", C.c_int32),
    ]


_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int32)


class OrcIO(C.Structure):
    _fields_ = [(n, _FP) for n in ("t_rand", "u_rand", "noise_c", "noise_f", "rgb_fine", "rgb_coarse",
                                   "depth", "acc", "z_coarse", "w_coarse", "cdf")] + \
               [("ind", _IP)] + \
               [(n, _FP) for n in ("z_samples", "z_fine", "w_fine", "sigma_c", "sigma_f", "craw_c", "craw_f", "z_fine_in")]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("nerf_oracle.cpp", "nerf_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            so = build()
        _LIB = C.CDLL(so)
        _LIB.orc_param_count.restype = C.c_int64
        _LIB.orc_net_param_count.restype = C.c_int64
        _LIB.orc_lr.restype = C.c_float
        _LIB.orc_pos_width.restype = C.c_int32
        _LIB.orc_num_threads.restype = C.c_int32
    return _LIB


def make_config(cfg):
    """cfg: any object/dict with the nerfConfig-style fields (see nerf_amd.NerfConfig)."""
    g = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
    c = OrcConfig()
    c.use_dir = int(g("useDir")); c.use_sh = int(g("useSH")); c.use_hierarchical = int(g("useHierarchical"))
    c.pos_l = g("posL"); c.dir_l = g("dirL"); c.D = g("D"); c.W = g("W")
    skips = list(g("skips"))
    c.n_skips = len(skips)
    for i, s in enumerate(skips):
        c.skips[i] = s
    c.n_samples = g("NSamples"); c.n_importance = g("NImportance")
    c.raw_noise_std = float(g("rawNoiseStd")); c.white_bkgd = int(g("whiteBkgd"))
    c.lin_disp = int(g("linDisp")); c.perturb = int(g("perturb"))
    c.lrate = float(g("lrate", 5e-4)); c.lrate_decay = int(g("lrateDecay", 250))
    return c


def _f(a):
    return None if a is None else a.ctypes.data_as(_FP)


def _arr(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


class Oracle:
    """Thin object wrapper: one oracle per configuration."""

    def __init__(self, cfg):
        self.c = make_config(cfg)
        self.N = self.c.n_samples
        self.NI = self.c.n_importance
        self.hier = bool(self.c.use_hierarchical)
        self.S = self.N + self.NI if self.hier else self.N
        self.L = lib()

    def param_count(self):
        return int(self.L.orc_param_count(C.byref(self.c)))

    def net_param_count(self):
        return int(self.L.orc_net_param_count(C.byref(self.c)))

    def init_params(self, seed=0):
        p = np.zeros(self.param_count(), np.float32)
        self.L.orc_init_params(C.byref(self.c), C.c_uint64(seed), _f(p))
        return p

    def linspace(self, n):
        o = np.zeros(n, np.float32)
        self.L.orc_linspace(C.c_int32(n), _f(o))
        return o

    def position_code(self, x, L):
        x = _arr(x).reshape(-1, 3)
        o = np.zeros((x.shape[0], 3 * (1 + 2 * L)), np.float32)
        self.L.orc_position_code(_f(x), C.c_int64(x.shape[0]), C.c_int32(L), _f(o))
        return o

    def sh2(self, d):
        d = _arr(d).reshape(-1, 3)
        o = np.zeros((d.shape[0], 9), np.float32)
        self.L.orc_sh2(_f(d), C.c_int64(d.shape[0]), _f(o))
        return o

    def stratified(self, bounds, t_rand=None):
        bounds = _arr(bounds)
        B = bounds.shape[0]
        t_rand = _arr(t_rand, (B, self.N))
        z = np.zeros((B, self.N), np.float32)
        self.L.orc_stratified(C.byref(self.c), C.c_int64(B), _f(bounds), _f(t_rand), _f(z))
        return z

    def composite(self, sigma, craw, z, noise=None):
        z = _arr(z)
        B, S = z.shape
        sigma = _arr(sigma, (B, S)); craw = _arr(craw, (B, S, 3)); noise = _arr(noise, (B, S))
        w = np.zeros((B, S), np.float32); rgb = np.zeros((B, 3), np.float32)
        depth = np.zeros(B, np.float32); acc = np.zeros(B, np.float32)
        self.L.orc_composite(C.byref(self.c), C.c_int64(B), C.c_int32(S), _f(sigma), _f(noise), _f(craw), _f(z),
                             _f(w), _f(rgb), _f(depth), _f(acc))
        return w, rgb, depth, acc

    def composite_bwd(self, sigma, craw, z, w, d_rgb, noise=None):
        z = _arr(z)
        B, S = z.shape
        sigma = _arr(sigma, (B, S)); craw = _arr(craw, (B, S, 3)); noise = _arr(noise, (B, S))
        w = _arr(w, (B, S)); d_rgb = _arr(d_rgb, (B, 3))
        ds = np.zeros((B, S), np.float32); dc = np.zeros((B, S, 3), np.float32)
        self.L.orc_composite_bwd(C.byref(self.c), C.c_int64(B), C.c_int32(S), _f(sigma), _f(noise), _f(craw),
                                 _f(z), _f(w), _f(d_rgb), _f(ds), _f(dc))
        return ds, dc

    def sample_pdf(self, w, z, u_rand=None):
        w = _arr(w); z = _arr(z)
        B = w.shape[0]
        N, NI = self.N, self.NI
        assert w.shape == (B, N) and z.shape == (B, N)
        u_rand = _arr(u_rand, (B, NI))
        cdf = np.zeros((B, N - 1), np.float32); ind = np.zeros((B, NI), np.int32)
        zs = np.zeros((B, NI), np.float32); zf = np.zeros((B, N + NI), np.float32)
        self.L.orc_sample_pdf(C.byref(self.c), C.c_int64(B), _f(w), _f(z), _f(u_rand), _f(cdf),
                              ind.ctypes.data_as(_IP), _f(zs), _f(zf))
        return cdf, ind, zs, zf

    def mlp_forward(self, net_params, is_coarse, training, rays_o, rays_d, viewdir, z):
        z = _arr(z)
        B, S = z.shape
        net_params = _arr(net_params); rays_o = _arr(rays_o, (B, 3)); rays_d = _arr(rays_d, (B, 3))
        viewdir = _arr(viewdir, (B, 3))
        sigma = np.zeros((B, S), np.float32); craw = np.zeros((B, S, 3), np.float32)
        self.L.orc_mlp_forward(C.byref(self.c), _f(net_params), C.c_int32(int(is_coarse)), C.c_int32(int(training)),
                               C.c_int64(B), C.c_int32(S), _f(rays_o), _f(rays_d), _f(viewdir), _f(z),
                               _f(sigma), _f(craw))
        return sigma, craw

    def mlp_forward_korder(self, net_params, want_rgb, rays_o, rays_d, viewdir, z):
        """the network in the summation order of the fp32 MFMA kernels (orc_mlp_forward_korder)"""
        z = _arr(z)
        B, S = z.shape
        net_params = _arr(net_params); rays_o = _arr(rays_o, (B, 3)); rays_d = _arr(rays_d, (B, 3))
        viewdir = _arr(viewdir, (B, 3))
        sigma = np.zeros((B, S), np.float32); craw = np.zeros((B, S, 3), np.float32)
        self.L.orc_mlp_forward_korder(C.byref(self.c), _f(net_params), C.c_int32(int(want_rgb)), C.c_int64(B),
                                      C.c_int32(S), _f(rays_o), _f(rays_d), _f(viewdir), _f(z), _f(sigma), _f(craw))
        return sigma, craw

    def _io(self, B, rand):
        N, NI, S = self.N, self.NI, self.S
        out = {
            "rgb_fine": np.zeros((B, 3), np.float32), "rgb_coarse": np.zeros((B, 3), np.float32),
            "depth": np.zeros(B, np.float32), "acc": np.zeros(B, np.float32),
            "z_coarse": np.zeros((B, N), np.float32), "w_coarse": np.zeros((B, N), np.float32),
            "cdf": np.zeros((B, max(N - 1, 1)), np.float32), "ind": np.zeros((B, max(NI, 1)), np.int32),
            "z_samples": np.zeros((B, max(NI, 1)), np.float32), "z_fine": np.zeros((B, S), np.float32),
            "w_fine": np.zeros((B, S), np.float32), "sigma_c": np.zeros((B, N), np.float32),
            "sigma_f": np.zeros((B, S), np.float32), "craw_c": np.zeros((B, N, 3), np.float32),
            "craw_f": np.zeros((B, S, 3), np.float32),
        }
        io = OrcIO()
        keep = []
        rand = rand or {}
        shapes = {"t_rand": (B, N), "u_rand": (B, NI), "noise_c": (B, N), "noise_f": (B, S)}
        for k, shp in shapes.items():
            a = _arr(rand.get(k), shp) if rand.get(k) is not None else None
            keep.append(a)
            setattr(io, k, _f(a) if a is not None else None)
        for k, a in out.items():
            setattr(io, k, a.ctypes.data_as(_IP) if k == "ind" else _f(a))
        zin = rand.get("z_fine_in")   # test hook (nerf_oracle.h): the rest of the pass at THESE fine depths
        if zin is not None:
            zin = _arr(zin, (B, S))
            keep.append(zin)
            io.z_fine_in = _f(zin)
        return io, out, keep

    def forward(self, params, rays_o, rays_d, bounds, viewdir, training=False, rand=None):
        rays_o = _arr(rays_o); B = rays_o.shape[0]
        rays_d = _arr(rays_d, (B, 3)); bounds = _arr(bounds, (B, 2)); viewdir = _arr(viewdir, (B, 3))
        params = _arr(params, (self.param_count(),))
        io, out, keep = self._io(B, rand)
        self.L.orc_forward(C.byref(self.c), _f(params), C.c_int32(int(training)), C.c_int64(B), _f(rays_o),
                           _f(rays_d), _f(bounds), _f(viewdir), C.byref(io))
        return out

    def train_grads(self, params, rays_o, rays_d, bounds, viewdir, label, rand=None):
        rays_o = _arr(rays_o); B = rays_o.shape[0]
        rays_d = _arr(rays_d, (B, 3)); bounds = _arr(bounds, (B, 2)); viewdir = _arr(viewdir, (B, 3))
        label = _arr(label, (B, 3))
        params = _arr(params, (self.param_count(),))
        io, out, keep = self._io(B, rand)
        lf = C.c_float(); lc = C.c_float()
        g = np.zeros(self.param_count(), np.float32)
        self.L.orc_train_grads(C.byref(self.c), _f(params), C.c_int64(B), _f(rays_o), _f(rays_d), _f(bounds),
                               _f(viewdir), _f(label), C.byref(io), C.byref(lf), C.byref(lc), _f(g))
        out["loss_fine"] = lf.value
        out["loss_coarse"] = lc.value
        out["grads"] = g
        return out

    def lr(self, t):
        return float(self.L.orc_lr(C.byref(self.c), C.c_int64(t)))

    def adam(self, p, g, m, v, t, grad_scale=1.0):
        """In-place on p, m, v (float32 contiguous)."""
        for a in (p, m, v):
            assert a.dtype == np.float32 and a.flags.c_contiguous
        g = _arr(g)
        self.L.orc_adam(C.byref(self.c), C.c_int64(p.size), _f(p), _f(g), _f(m), _f(v), C.c_int64(t),
                        C.c_float(grad_scale))

    def gen_rays(self, H, W, focal, c2w, near, far, ndc=False):
        c2w = _arr(np.asarray(c2w)[:3, :4], (3, 4))
        o = np.zeros((H * W, 3), np.float32); d = np.zeros((H * W, 3), np.float32)
        b = np.zeros((H * W, 2), np.float32)
        self.L.orc_gen_rays(C.c_int32(H), C.c_int32(W), C.c_float(focal), _f(c2w), C.c_float(near), C.c_float(far),
                            C.c_int32(int(ndc)), _f(o), _f(d), _f(b))
        return o, d, b

    def to_u8(self, rgb):
        rgb = _arr(rgb)
        out = np.zeros(rgb.shape, np.uint8)
        self.L.orc_to_u8(_f(rgb), C.c_int64(rgb.size), out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out

    def num_threads(self):
        return int(self.L.orc_num_threads())
