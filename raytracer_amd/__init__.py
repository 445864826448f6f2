"""raytracer_amd: the wallisc/RayTracer per-pixel render path on MI355X (HIP, gfx950).

The product is libraytracer_amd.so (C ABI: include/raytracer_amd.h, built from
raytracer_amd/csrc by __graft_entry__.build()); this package is the thin Python
mirror of that interface.
"""
from . import abi, tiles
from .api import (COOK_TORRANCE, PHONG, MultiScene, RaytracerError, Scene, Sequence, device_count, launch_kernel, load_bmp, make_params, parse_pov,
                  tile_count, write_tga)
from .scene import SceneData, default_material, identity_modifier

__all__ = ["abi", "tiles", "COOK_TORRANCE", "PHONG", "MultiScene", "RaytracerError", "Scene", "SceneData", "Sequence", "default_material",
           "device_count", "identity_modifier", "launch_kernel", "load_bmp", "make_params", "parse_pov", "tile_count", "write_tga"]
