"""In-process stand-ins for the third-party modules the reference imports but that are
absent from this container (diffusers, videox_fun, torchvision ...).

Used ONLY by tests/golden/make_golden.py, in the build container, to import the reference's
own model files from /root/reference and record golden input/output vectors.  Nothing here
ships in the product path and nothing here is imported on the GPU box (the reference does
not exist there).

What is real and what is a stand-in
-----------------------------------
* Real (the reference's own code, executed as-is): everything in
  versecrafter/models/wan_transformer3d.py and wan_transformer3d_versecrafter.py.
* Stand-in (third-party, un-vendored submodule -> "parity unpinned" at this boundary):
  - videox_fun.models.attention_utils.attention: softmax(q k^T / sqrt(D)) v on [B,L,N,D],
    keys >= k_lens[b] masked (flash-attn varlen semantics as used by upstream Wan2.1).
  - videox_fun.models.cache_utils.TeaCache: field/method contract as used by
    wan_transformer3d.py:205-245, 828-839.
  - videox_fun.utils.cfg_skip: identity decorator (CLI default cfg_skip_ratio=0).
  - diffusers mixins: plain classes with a register_to_config that records kwargs.
"""
import functools
import inspect
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        if not hasattr(self, "_vc_config"):
            object.__setattr__(self, "_vc_config", _Cfg())
        self._vc_config.update(cfg)
        init(self, *args, **kwargs)
    return inner


class ConfigMixin:
    @property
    def config(self):
        return self._vc_config

    @classmethod
    def from_config(cls, config, **kwargs):
        params = inspect.signature(cls.__init__).parameters
        kw = {k: v for k, v in dict(config).items() if k in params}
        kw.update({k: v for k, v in kwargs.items() if k in params})
        return cls(**kw)


class ModelMixin(nn.Module):
    pass


class FromOriginalModelMixin:
    pass


def attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0.0, softmax_scale=None,
              q_scale=None, causal=False, window_size=(-1, -1), deterministic=False,
              dtype=torch.bfloat16, **kw):
    """[B,L,N,D] attention; keys at index >= k_lens[b] are masked out."""
    b, lq, n, d = q.shape
    lk = k.shape[1]
    qh, kh, vh = (t.transpose(1, 2).float() for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / (d ** 0.5)
    if k_lens is not None:
        mask = torch.arange(lk)[None, :] >= torch.as_tensor(k_lens)[:, None]  # [B, Lk]
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    o = torch.softmax(s, dim=-1) @ vh
    return o.transpose(1, 2).contiguous().to(q.dtype)


class TeaCache:
    """Contract from wan_transformer3d.py:205-245, 828-839 (third-party class, restated)."""

    def __init__(self, coefficients, num_steps, rel_l1_thresh=0.0, num_skip_start_steps=0,
                 offload=True):
        self.coefficients = coefficients
        self.num_steps = num_steps
        self.rel_l1_thresh = rel_l1_thresh
        self.num_skip_start_steps = num_skip_start_steps
        self.offload = offload
        self.rescale_func = np.poly1d(coefficients)
        self.reset()

    @staticmethod
    def compute_rel_l1_distance(prev, cur):
        return ((cur - prev).abs().mean() / prev.abs().mean()).cpu().item()

    def reset(self):
        self.cnt = 0
        self.should_calc = True
        self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = None
        self.previous_residual = None
        self.previous_residual_cond = None
        self.previous_residual_uncond = None


def cfg_skip():
    def deco(fn):
        return fn
    return deco


def install():
    d = _mod("diffusers")
    d.AutoencoderKL = object
    d.__version__ = "0.0.0-stub"
    cu = _mod("diffusers.configuration_utils")
    cu.ConfigMixin = ConfigMixin
    cu.register_to_config = register_to_config
    _mod("diffusers.loaders")
    sf = _mod("diffusers.loaders.single_file_model")
    sf.FromOriginalModelMixin = FromOriginalModelMixin
    _mod("diffusers.models")
    mu = _mod("diffusers.models.modeling_utils")
    mu.ModelMixin = ModelMixin
    du = _mod("diffusers.utils")
    du.is_torch_version = lambda *a, **k: True
    du.logging = types.SimpleNamespace(get_logger=lambda *a, **k: types.SimpleNamespace(
        warning=print, info=lambda *a, **k: None))

    vf = _mod("videox_fun")
    dist = _mod("videox_fun.dist")
    dist.get_sequence_parallel_rank = lambda: 0
    dist.get_sequence_parallel_world_size = lambda: 1
    dist.get_sp_group = lambda: None
    dist.usp_attn_forward = None
    dist.xFuserLongContextAttention = None
    ut = _mod("videox_fun.utils")
    ut.cfg_skip = cfg_skip
    vm = _mod("videox_fun.models")
    au = _mod("videox_fun.models.attention_utils")
    au.attention = attention
    cu2 = _mod("videox_fun.models.cache_utils")
    cu2.TeaCache = TeaCache
    ca = _mod("videox_fun.models.wan_camera_adapter")
    ca.SimpleAdapter = object
    vf.dist, vf.utils, vf.models = dist, ut, vm
    vm.attention_utils, vm.cache_utils, vm.wan_camera_adapter = au, cu2, ca

    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")


# ---- additional stand-ins needed to import versecrafter/pipeline/pipeline_wan_versecrafter.py -------------------------------
# Real: the reference's own pipeline file, executed as it lies.  Stand-ins (third-party, un-vendored / not installed):
#   diffusers.DiffusionPipeline      : register_modules = setattr, progress_bar = no-op context, _execution_device = cpu
#   diffusers.VaeImageProcessor      : preprocess(tensor [N,C,H,W] in [0,1]) -> 2x-1 when do_normalize (default) else x; no
#                                      resize (inputs already have height x width) -- diffusers' behaviour for tensor inputs
#   diffusers.randn_tensor           : torch.randn with a generator
#   videox_fun.utils.fm_solvers_unipc.FlowUniPCMultistepScheduler : this repo's restatement (isinstance target)
#   torchvision, comfy, PIL users    : empty modules (never called on the recorded path)
class DiffusionPipeline:
    def __init__(self):
        pass

    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def _execution_device(self):
        return torch.device("cpu")

    def progress_bar(self, iterable=None, total=None):
        import contextlib

        class _Bar:
            def update(self, n=1):
                pass

        @contextlib.contextmanager
        def cm():
            yield _Bar()
        return cm()

    def maybe_free_model_hooks(self):
        pass


class VaeImageProcessor:
    def __init__(self, vae_scale_factor=8, do_normalize=True, do_binarize=False, do_convert_grayscale=False, **kw):
        self.do_normalize, self.do_binarize = do_normalize, do_binarize

    def preprocess(self, image, height=None, width=None):
        assert torch.is_tensor(image) and image.dim() == 4
        assert (height is None or image.shape[2] == height) and (width is None or image.shape[3] == width)
        out = image
        if self.do_normalize:
            out = 2.0 * out - 1.0
        if self.do_binarize:
            out = (out >= 0.5).to(out.dtype)
        return out


def install_pipeline_stubs():
    import importlib.machinery
    import transformers  # noqa: F401  (real; must be imported before the empty torchvision module exists)
    from transformers import T5Tokenizer  # noqa: F401
    d = sys.modules["diffusers"]

    class FlowMatchEulerDiscreteScheduler:      # isinstance target only
        pass

    d.FlowMatchEulerDiscreteScheduler = FlowMatchEulerDiscreteScheduler
    cb = _mod("diffusers.callbacks")
    cb.MultiPipelineCallbacks = type("MultiPipelineCallbacks", (), {})
    cb.PipelineCallback = type("PipelineCallback", (), {})
    ip = _mod("diffusers.image_processor")
    ip.VaeImageProcessor = VaeImageProcessor
    emb = _mod("diffusers.models.embeddings")
    emb.get_1d_rotary_pos_embed = None
    _mod("diffusers.pipelines")
    pu = _mod("diffusers.pipelines.pipeline_utils")
    pu.DiffusionPipeline = DiffusionPipeline
    sch = _mod("diffusers.schedulers")
    sch.FlowMatchEulerDiscreteScheduler = FlowMatchEulerDiscreteScheduler
    du = sys.modules["diffusers.utils"]
    du.BaseOutput = object
    du.replace_example_docstring = lambda doc: (lambda fn: fn)
    tu = _mod("diffusers.utils.torch_utils")
    tu.randn_tensor = lambda shape, generator=None, device=None, dtype=None: torch.randn(
        shape, generator=generator, dtype=dtype).to(device)
    vp = _mod("diffusers.video_processor")
    vp.VideoProcessor = type("VideoProcessor", (), {"__init__": lambda self, **kw: None})
    tv = _mod("torchvision")
    tvt = _mod("torchvision.transforms")
    tvf = _mod("torchvision.transforms.functional")
    tv.transforms, tvt.functional = tvt, tvf
    for m in (tv, tvt, tvf):
        m.__spec__ = importlib.machinery.ModuleSpec(m.__name__, None)

    vm = sys.modules["videox_fun.models"]
    vm.AutoencoderKLWan = vm.AutoTokenizer = vm.WanT5EncoderModel = object
    fm = _mod("videox_fun.utils.fm_solvers_unipc")
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(
        __import__("os").path.abspath(__file__))))
    if root not in sys.path:
        sys.path.insert(0, root)
    from versecrafter_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    fm.FlowUniPCMultistepScheduler = FlowUniPCMultistepScheduler
    sys.modules["videox_fun.utils"].fm_solvers_unipc = fm
