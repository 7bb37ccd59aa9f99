"""raytracing_amd -- MI355X-native primary-ray tracer for the voxel world of
Solvir777/Raytracing: HIP kernels behind the C ABI of include/vrt.h, plus the thin
host-side mirror of the reference's GraphicsCore / Player interface for this path."""
from . import _lib
from ._lib import (MEM_DEVICE, MEM_HOST, TRACE_AUTO, TRACE_LITERAL, TRACE_SKIP, VOXEL_PACKED_U8,
                   VOXEL_REF_U16U16, VrtError)
from .graphics_core import GraphicsCore, GraphicsSettings
from .player import Player, PushConstants
from . import scenes

__all__ = ["GraphicsCore", "GraphicsSettings", "Player", "PushConstants", "scenes", "VrtError", "_lib",
           "MEM_DEVICE", "MEM_HOST", "TRACE_AUTO", "TRACE_LITERAL", "TRACE_SKIP", "VOXEL_PACKED_U8",
           "VOXEL_REF_U16U16"]
