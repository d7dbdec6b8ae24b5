"""MI355X-native bundle-adjustment hot path (HIP kernels behind a C ABI).

Public surface mirrors the reference's solver interface:
FullBundleAdjustmentSolver, PoseOnlyBundleAdjustmentSolver, Options, Summary.
The numerical path lives in libba_hip.so (include/ba_hip.h); importing this
package does not touch the GPU, creating a solver does.
"""
from .solver import (BaProblem, Camera, FullBundleAdjustmentSolver,  # noqa
                     FullBundleAdjustmentSolverRefactor,
                     IterationStatus, OptimizationInfo, Options,
                     PoseOnlyBundleAdjustmentSolver, SolverType, Summary)
from . import scenes  # noqa
from . import scene_io  # noqa

__all__ = ["BaProblem", "Camera", "FullBundleAdjustmentSolver",
           "FullBundleAdjustmentSolverRefactor",
           "IterationStatus", "OptimizationInfo", "Options",
           "PoseOnlyBundleAdjustmentSolver", "SolverType", "Summary",
           "scenes", "scene_io"]
