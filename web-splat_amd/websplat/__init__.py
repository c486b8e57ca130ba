"""websplat -- Python binding of libwebsplat_hip.so, the MI355X drop-in for web-splat's render path.

Importing this package loads the HIP library; there is no CPU fallback (ImportError if it is not built).
"""
from ._lib import (LIB_PATH, WebSplatError, lib, check, ws_gaussian_quantization, ws_quantization)  # noqa: F401
from .api import (Aabb, Context, GaussianRenderer, GenericGaussianPointCloud, GPURSSorter, PerspectiveCamera,  # noqa: F401
                  PointCloud, SplattingArgs, pointcloud_stats, FORMATS, Scene, SceneCamera, read_npz, read_ply, write_png,
                  render_views, measure, ViewBatch, config_from_env, stage_splat, footprint_tiles, packed_rect, binning_decision)
from . import synth  # noqa: F401
