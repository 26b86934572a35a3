"""robosimgs_amd -- MI355X-native 3D Gaussian Splatting render path for RoboSimGS scenes.

Host-side API (stays Python): Camera, Gaussians, load_ply / save_ply, synthetic_scene.
Device path (hand-written HIP for gfx950 behind the C ABI in include/mgs.h): the operators
in `ops` and `rasterization` / `render`.  Importing the device path needs torch and the
built libmgs.so; the host-side API needs only numpy.
"""
from .camera import (Camera, camera_ring, cameras_from_camera_params_json,
                     cameras_from_transforms_json, depth_to_distance, distance_to_depth,
                     unproject_point)
from .gaussians import (Gaussians, load_dataparser_transforms, load_ply, save_ply,
                        synthetic_scene, synthetic_scene_heavy_tailed)

__version__ = "0.1.0"

_DEVICE_API = {"rasterization", "render", "check_isect_status", "fully_fused_projection",
               "spherical_harmonics", "isect_tiles", "isect_offset_encode",
               "rasterize_to_pixels", "render_sharded", "gather_frames"}


def __getattr__(name):  # lazy: keeps `import robosimgs_amd` torch-free for host-only use
    if name in ("rasterization", "render", "check_isect_status"):
        from . import rendering
        return getattr(rendering, name)
    if name in ("fully_fused_projection", "spherical_harmonics", "isect_tiles",
                "isect_offset_encode", "rasterize_to_pixels"):
        from . import ops
        return getattr(ops, name)
    if name in ("composite_over", "frame_to_u8"):
        from . import compositing
        return getattr(compositing, name)
    if name in ("frame_to_dataset", "DatasetWriter", "read_dataset_frame", "unnormalize_points"):
        from . import dataset
        return getattr(dataset, name)
    if name == "transform_gaussians":
        from . import transform
        return transform.transform_gaussians
    if name in ("l1_loss", "unit_gradient"):
        from . import losses
        return getattr(losses, name)
    if name in ("FrameRenderer", "locality_order"):
        from . import pipeline
        return getattr(pipeline, name)
    if name in ("reorder_parameters", "Trainer"):
        from . import training
        return getattr(training, name)
    if name in ("render_sharded", "gather_frames", "shard_cameras"):
        from . import distributed
        return getattr(distributed, name)
    raise AttributeError(name)
