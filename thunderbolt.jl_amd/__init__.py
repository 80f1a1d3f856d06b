"""thunderbolt.jl_amd — MI355X (gfx950) backend for Thunderbolt.jl's per-cell FE integration and
pointwise reaction hot path.  The compute lives in libtbhip.so (hand-written HIP, C ABI in
include/tbhip.h); this package is the host-side mirror of the reference's operator API.
Import as ``import thunderbolt_jl_amd`` (shim at the repository root)."""
from . import _lib
from ._lib import TBError, build_library, lib  # noqa: F401
from .api import *  # noqa: F401,F403
from . import distributed  # noqa: F401
from . import meshio  # noqa: F401
from . import coordinates  # noqa: F401
from .coordinates import (compute_lv_coordinate_system, compute_midmyocardial_section_coordinate_system, create_microstructure_model,  # noqa: F401
                          create_lumped_microstructure_model, ODB25LTMicrostructureParameters, evaluate_coordinate_axes, evaluate_coordinate,
                          wrap_rotational, apicobasal_from_laplace)
from .meshgen import generate_ring_mesh, generate_open_ring_mesh, generate_ideal_lv_mesh_hex, ideal_lv_microstructure, uniform_refinement  # noqa: F401
