"""totsu_amd -- MI355X (gfx950) backend for the Totsu first-order conic solver.

Host-side mirror of the reference's operator / plugin interface for the hot path (totsu_core + totsu problem
builders) over the C ABI of libtotsu_f32hip.so.  No CPU fallback: the HIP library must be built
(`__graft_entry__.build()`), and using any op without a GPU raises.
"""
from .matop import MatOp, MatType
from .solver import Solver, SolverError, SolverParam
from .linalg import F32HIP, F32HIPSlice, splitm
from .cone import ConeZero, ConeRPos, ConeSOC, ConeRotSOC, ConePSD
from .matbuild import MatBuild
from .problem import ProbLP, ProbSOCP, ProbSDP, ProbQP, ProbQCQP
from .fused import FusedSolver, DeviceBuffer, Bf16Matrix
from .parallel import ShardedSolver, TorchComm, shard_segments
from .sparse import SparseMatOp, SpTile

__all__ = ["MatOp", "MatType", "Solver", "SolverError", "SolverParam", "F32HIP", "F32HIPSlice", "splitm",
           "ConeZero", "ConeRPos", "ConeSOC", "ConeRotSOC", "ConePSD", "MatBuild", "ProbLP", "ProbSOCP",
           "ProbSDP", "ProbQP", "ProbQCQP", "FusedSolver", "DeviceBuffer", "Bf16Matrix", "ShardedSolver", "TorchComm",
           "shard_segments", "SparseMatOp", "SpTile"]
