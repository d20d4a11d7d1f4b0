"""nice_slam_amd -- MI355X-native (gfx950, HIP) implementation of the NICE-SLAM volume-rendering hot
path, drop-in behind the reference's Renderer / decoder / get_samples call surface.

    from nice_slam_amd import Renderer, NICE, get_samples, grid_init, load_bound

No CPU / PyTorch fallback exists: every arithmetic entry point goes through libnsr.so.
"""
from .common import aabb_keep, get_camera_from_tensor, get_samples, get_rays, grid_init, load_bound, to_channels_last  # noqa: F401
from .decoders import NICE, MLP, MLP_no_xyz  # noqa: F401
from .renderer import Renderer  # noqa: F401
from .optim import FlatAdam, MaskedGridAdam  # noqa: F401
from .frustum import FrustumSelector  # noqa: F401
from .mapping import backward, get_samples_window, mapping_loss, seed_pixel_draws, tracking_loss  # noqa: F401
from . import graphs  # noqa: F401

__all__ = ["Renderer", "NICE", "MLP", "MLP_no_xyz", "get_samples", "get_rays", "grid_init", "load_bound",
           "to_channels_last", "MaskedGridAdam", "FlatAdam", "FrustumSelector", "aabb_keep", "get_samples_window", "mapping_loss", "tracking_loss", "seed_pixel_draws", "get_camera_from_tensor", "backward"]
