"""MI355X-native volumetric rendering path of NeRF-SOS.

Host-side mirror of the reference's interface for this path (models/nerf_net.py, models/nerf_mlp.py,
models/sampler.py, models/renderer.py) over the C ABI in ``include/nerf_sos_hip.h``.  There is no CPU or
eager-PyTorch fallback: every op raises if the HIP library is missing or the tensors are not on a GPU.
"""
from .nerf_net import MLP, NeRFMLP, NeRFNet, export_density  # noqa: F401
from . import ops  # noqa: F401
from . import sharding  # noqa: F401
from . import losses  # noqa: F401
from . import io  # noqa: F401
from . import synthetic  # noqa: F401
from . import quality  # noqa: F401
from .graphs import GraphedPatchStep, GraphedRender  # noqa: F401
from .losses import CorrelationLoss, GeoCorrelationLoss, NeRFContrastive  # noqa: F401

__all__ = ["NeRFNet", "NeRFMLP", "MLP", "export_density", "ops", "sharding", "losses", "io", "synthetic", "quality", "CorrelationLoss", "GeoCorrelationLoss", "NeRFContrastive", "GraphedRender", "GraphedPatchStep"]
