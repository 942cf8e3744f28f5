"""mrhyde_amd -- MI355X-native element-local assembly path of MrHyDE.

The product is the C-ABI shared library built from mrhyde_amd/csrc (see
include/mrhyde_amd.h).  This package is a thin ctypes binding used by the
tests, bench.py and __graft_entry__.py: it only moves pointers across the
boundary.  Device memory, streams and torch.distributed come from PyTorch
(plumbing); every computation runs in the HIP kernels.  There is no CPU
fallback: creating a Block without a GPU raises.
"""
from .api import (Sparse3D, MASS_ON_THE_FLY, MASS_LOCAL, MASS_DATABASE, MASS_DATABASE_SPARSE, batched_condense, BASIS_HDIV, BASIS_HGRAD, BASIS_HVOL, BC_NEUMANN, BC_WEAK_DIRICHLET, BC_FLUX, BC_INTERFACE, Newton, NEWTON_SOLVE, NEWTON_BACKTRACKED, NEWTON_DONE, PATH_POINT_ENGINE, PATH_ROW_GATHER, Block, MhaError, ScatterPlan, PATH_AUTO, PATH_ELEMENT_ATOMIC, PATH_LOCAL_THEN_SCATTER, PATH_ROW_OWNER,
                  device_count, lib_path, load_library, mesh_structured, mesh_multi, row_partition, block_patterns_host_apply, swhdg_eigendecomp,
                  swhdg_side_terms, version)

__all__ = ["Sparse3D", "MASS_ON_THE_FLY", "MASS_LOCAL", "MASS_DATABASE", "MASS_DATABASE_SPARSE", "batched_condense", "BASIS_HDIV", "BASIS_HGRAD", "BASIS_HVOL", "PATH_POINT_ENGINE", "PATH_ROW_GATHER", "BC_NEUMANN", "BC_WEAK_DIRICHLET", "BC_FLUX", "BC_INTERFACE", "Newton", "NEWTON_SOLVE", "NEWTON_BACKTRACKED", "NEWTON_DONE", "Block", "MhaError", "ScatterPlan", "PATH_AUTO", "PATH_ELEMENT_ATOMIC", "PATH_LOCAL_THEN_SCATTER", "PATH_ROW_OWNER",
           "device_count", "lib_path", "load_library", "mesh_structured", "mesh_multi", "row_partition", "block_patterns_host_apply", "swhdg_eigendecomp", "swhdg_side_terms", "version"]
