"""ctypes binding of libnerfhip.so (the C ABI declared in include/nerfhip.h).

The shared library is the product; this module only marshals pointers.  It fails loudly when the
library is missing or cannot be loaded -- there is no CPU or PyTorch fallback for any entry point.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnerfhip.so")

MAX_SKIPS = 8
F32, BF16, F16X3 = 0, 1, 2


class Config(C.Structure):
    """struct nerfhip_config (include/nerfhip.h)."""
    _fields_ = [
        ("use_dir", C.c_int32), ("use_sh", C.c_int32), ("use_hierarchical", C.c_int32),
        ("pos_l", C.c_int32), ("dir_l", C.c_int32), ("D", C.c_int32), ("W", C.c_int32),
        ("n_skips", C.c_int32), ("skips", C.c_int32 * MAX_SKIPS),
        ("n_samples", C.c_int32), ("n_importance", C.c_int32),
        ("raw_noise_std", C.c_float), ("white_bkgd", C.c_int32), ("lin_disp", C.c_int32), ("perturb", C.c_int32),
        ("lrate", C.c_float), ("lrate_decay", C.c_int32),
        ("device", C.c_int32), ("dtype", C.c_int32), ("seed", C.c_uint64), ("max_rays", C.c_int32),
        ("reserved", C.c_int32),
    ]


_P = C.c_void_p


class Aux(C.Structure):
    """struct nerfhip_aux: raw addresses (host or device, matching the entry point)."""
    _fields_ = [(n, _P) for n in ("t_rand", "u_rand", "noise_c", "noise_f", "z_coarse", "w_coarse", "z_fine",
                                  "w_fine", "sigma_c", "sigma_f", "craw_c", "craw_f", "depth", "acc")]


class Prof(C.Structure):
    _fields_ = [("mlp_fwd_ms", C.c_float), ("mlp_bwd_ms", C.c_float), ("mlp_wgrad_ms", C.c_float),
                ("composite_ms", C.c_float), ("sampling_ms", C.c_float), ("other_ms", C.c_float),
                ("mlp_fwd_launches", C.c_int32), ("reserved", C.c_int32),
                ("mlp_fwd_flop", C.c_double), ("mlp_bwd_flop", C.c_double), ("mlp_wgrad_flop", C.c_double),
                ("comm_ms", C.c_float), ("reserved2", C.c_float)]


# every symbol include/nerfhip.h declares: (name, restype, argtypes)
_I64, _I32, _F = C.c_int64, C.c_int32, C.c_float
SYMBOLS = {
    "nerfhip_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "nerfhip_destroy": (None, [_P]),
    "nerfhip_last_error": (C.c_char_p, [_P]),
    "nerfhip_set_stream": (C.c_int, [_P, _P]),
    "nerfhip_synchronize": (C.c_int, [_P]),
    "nerfhip_param_count": (_I64, [_P]),
    "nerfhip_set_params": (C.c_int, [_P, _P, _I64]),
    "nerfhip_get_params": (C.c_int, [_P, _P, _I64]),
    "nerfhip_init_params": (C.c_int, [_P, C.c_uint64]),
    "nerfhip_get_opt_state": (C.c_int, [_P, _P, _P, _I64, C.POINTER(_I64)]),
    "nerfhip_set_opt_state": (C.c_int, [_P, _P, _P, _I64, _I64]),
    "nerfhip_render": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, C.POINTER(Aux)]),
    "nerfhip_render_dev": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, C.POINTER(Aux)]),
    "nerfhip_render_dev_static": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P]),
    "nerfhip_render_static_max_rays": (_I64, [_P]),
    "nerfhip_render_u8": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P]),
    "nerfhip_render_pose_u8": (C.c_int, [_P, _I32, _I32, _F, _P, _F, _F, _I32, _P]),
    "nerfhip_render_pose_rows_u8": (C.c_int, [_P, _I32, _I32, _F, _P, _F, _F, _I32, _I32, _I32, _P]),
    "nerfhip_forward_train": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P, C.POINTER(Aux)]),
    "nerfhip_train_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P, C.POINTER(Aux)]),
    "nerfhip_train_backward_dev": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P, C.POINTER(Aux)]),
    "nerfhip_grad_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I64)]),
    "nerfhip_get_grads": (C.c_int, [_P, _P, _I64]),
    "nerfhip_zero_grads": (C.c_int, [_P]),
    "nerfhip_optimizer_step": (C.c_int, [_P, _F]),
    "nerfhip_train_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P]),
    "nerfhip_train_step_dev": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P]),
    "nerfhip_dp_unique_id": (C.c_int, [_P]),
    "nerfhip_dp_init": (C.c_int, [_P, _I32, _I32, _P]),
    "nerfhip_dp_init_all": (C.c_int, [C.POINTER(_P), _I32]),
    "nerfhip_dp_world": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32)]),
    "nerfhip_dp_broadcast_state": (C.c_int, [_P, _I32]),
    "nerfhip_dp_allreduce_grads": (C.c_int, [_P]),
    "nerfhip_dp_allreduce_grads_all": (C.c_int, [C.POINTER(_P), _I32]),
    "nerfhip_dp_train_step_dev": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P]),
    "nerfhip_dp_destroy": (C.c_int, [_P]),
    "nerfhip_k_stratified": (C.c_int, [_P, _P, _P, _I64, _P]),
    "nerfhip_k_composite_fwd": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P]),
    "nerfhip_k_composite_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P]),
    "nerfhip_k_sample_pdf": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "nerfhip_k_mlp_forward": (C.c_int, [_P, _I32, _P, _P, _P, _P, _I64, _I32, _P, _P]),
    "nerfhip_k_gen_rays": (C.c_int, [_P, _I32, _I32, _F, _P, _F, _F, _I32, _P, _P, _P]),
    "nerfhip_k_random": (C.c_int, [_P, _I32, C.c_uint64, C.c_uint32, _I64, _P]),
    "nerfhip_rng_step": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "nerfhip_set_profiling": (C.c_int, [_P, _I32]),
    "nerfhip_get_profile": (C.c_int, [_P, C.POINTER(Prof)]),
    "nerfhip_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Loads libnerfhip.so and types every entry point.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C nerf_amd/csrc` "
                           "(or __graft_entry__.build()); nerf_amd has no fallback path")
    # One HIP runtime per process: the PyTorch wheel carries its own libamdhip64.so, and whichever copy is mapped
    # first serves every later user of that soname.  If this library came first it would bring in /opt/rocm's
    # runtime, and torch, loaded afterwards on top of it, then finds no GPU ("No HIP GPUs are available").  Device
    # tensors of the callers are torch's, so torch goes first when it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class NerfHipError(RuntimeError):
    pass


def check(lib, handle, rc, what):
    if rc != 0:
        msg = lib.nerfhip_last_error(handle)
        raise NerfHipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
