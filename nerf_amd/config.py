"""NerfConfig: the arithmetic-relevant fields of the reference's case class nerfConfig
(reference nerf/nerfConfig.scala:6-55), same names, same meaning, plus the build-side extensions
(device ordinal, MFMA dtype, RNG seed, workspace size)."""
from dataclasses import dataclass, field
from typing import List

from . import _lib


@dataclass
class NerfConfig:
    useDir: bool = True              # nerfConfig.scala:20
    useSH: bool = False              # :21
    useHierarchical: bool = True     # :22
    posL: int = 10                   # :24
    dirL: int = 4                    # :25
    D: int = 8                       # :27 (depth; the reference's comments swap D and W, the code does not)
    W: int = 256                     # :28
    skips: List[int] = field(default_factory=lambda: [4])  # :29
    NSamples: int = 64               # :31
    NImportance: int = 128           # :32
    rawNoiseStd: float = 0.0         # :34
    whiteBkgd: bool = True           # :35
    linDisp: bool = False            # :36
    perturb: bool = False            # :37
    ndc: bool = False                # :38 (caller side: ray generation)
    batchNum: int = 1024             # :40
    lrate: float = 5e-4              # :42
    lrateDecay: int = 250            # :43
    # ---- build extensions
    device: int = 0                  # nerfConfig.device (:7) -> HIP ordinal of THIS process' GPU
    dtype: str = "f32"               # "f32" | "bf16" | "f16x3" (fp32-grade split-f16 render, fp32 training)
    seed: int = 0
    maxRays: int = 4096              # workspace is sized once for this many rays per call

    @staticmethod
    def from_dict(d):
        c = NerfConfig()
        for k, v in d.items():
            if hasattr(c, k):
                setattr(c, k, list(v) if k == "skips" else v)
        return c

    @property
    def samples_per_ray(self):
        return self.NSamples + (self.NImportance if self.useHierarchical else 0)

    def to_c(self):
        c = _lib.Config()
        c.use_dir, c.use_sh, c.use_hierarchical = int(self.useDir), int(self.useSH), int(self.useHierarchical)
        c.pos_l, c.dir_l, c.D, c.W = self.posL, self.dirL, self.D, self.W
        if len(self.skips) > _lib.MAX_SKIPS:
            raise ValueError("too many skips")
        c.n_skips = len(self.skips)
        for i, s in enumerate(self.skips):
            c.skips[i] = s
        c.n_samples, c.n_importance = self.NSamples, self.NImportance
        c.raw_noise_std = float(self.rawNoiseStd)
        c.white_bkgd, c.lin_disp, c.perturb = int(self.whiteBkgd), int(self.linDisp), int(self.perturb)
        c.lrate, c.lrate_decay = float(self.lrate), int(self.lrateDecay)
        c.device = self.device
        c.dtype = {"f32": _lib.F32, "bf16": _lib.BF16, "f16x3": _lib.F16X3}[self.dtype]
        c.seed = self.seed
        c.max_rays = self.maxRays
        return c
