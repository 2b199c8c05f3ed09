"""MI355X-native watershed engine: drop-in for the segmenting / merging transform path of
smups/rustronomy-watershed.  The directory name carries a hyphen, so import it
through `__graft_entry__.load_package()` (registers it as `rustronomy_watershed_amd`)."""
from . import _ffi
from .api import (ALWAYS_FILL, ENGINE_AUTO, ENGINE_FUSED, ENGINE_SWEEP, NEVER_FILL, NORMAL_MAX, UNCOLOURED, BuildErr,
                  Context, HookCtx, MaxToHigh, MaxToLow, MergingWatershed, SeedOutOfBounds, SegmentingWatershed,
                  TransformBuilder, WatershedError, WatershedUtils, default_context)

__all__ = ["ALWAYS_FILL", "ENGINE_AUTO", "ENGINE_FUSED", "ENGINE_SWEEP", "NEVER_FILL", "NORMAL_MAX", "UNCOLOURED",
           "BuildErr", "Context", "HookCtx", "MaxToHigh", "MaxToLow", "MergingWatershed", "SeedOutOfBounds",
           "SegmentingWatershed", "TransformBuilder", "WatershedError", "WatershedUtils", "default_context", "_ffi"]
