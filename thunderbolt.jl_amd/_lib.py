"""ctypes declarations for libtbhip.so (include/tbhip.h).  Loading fails loudly when the HIP
library has not been built: there is no CPU fallback anywhere in this package."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TB_LIBTBHIP") or os.path.join(_HERE, "libtbhip.so")   # TB_LIBTBHIP: a profiling build (make ablation)

# enums of include/tbhip.h
TB_OK = 0
TB_ERR_BAD_ARG, TB_ERR_HIP, TB_ERR_NEG_DETJ, TB_ERR_PATTERN, TB_ERR_UNSUPPORTED, TB_ERR_NOMEM = -1, -2, -3, -4, -5, -6
TB_QUAD4, TB_HEX8, TB_TET4, TB_HEX27 = 2, 3, 4, 5
TB_STRATEGY_ATOMIC, TB_STRATEGY_PER_COLOR, TB_STRATEGY_ELEMENT, TB_STRATEGY_PATCH = 0, 1, 2, 3
TB_FORM_MASS, TB_FORM_DIFFUSION, TB_FORM_SOURCE, TB_FORM_HYPERELASTIC = 0, 1, 2, 3
TB_MATERIAL_HOLZAPFEL_OGDEN_2009 = 0
TB_BC_ROBIN, TB_BC_NORMAL_SPRING, TB_BC_PRESSURE, TB_BC_BENDING_SPRING, TB_BC_PRESSURE_FIELD = 0, 1, 2, 3, 4
TB_COEF_CONST_SCALAR, TB_COEF_CONST_TENSOR, TB_COEF_FIELD_SCALAR = 0, 1, 2
TB_COEF_SPECTRAL_CONST, TB_COEF_SPECTRAL_FIELD, TB_COEF_TRANSVERSE_CONST = 3, 4, 5
TB_SRC_CONST, TB_SRC_NORM_PLUS_T, TB_SRC_COS_EXP, TB_SRC_TABULATED = 0, 1, 2, 3
TB_CELL_FHN, TB_CELL_ALIEV_PANFILOV, TB_CELL_PCG2019, TB_CELL_TT06, TB_CELL_FHN_HETEROGENEOUS, TB_CELL_ORD11 = 0, 1, 2, 3, 4, 5
TB_LAYOUT_SOA, TB_LAYOUT_AOS = 0, 1

c_dp = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
vp = C.c_void_p


class tb_coef(C.Structure):
    _fields_ = [("kind", C.c_int32), ("wrap", C.c_int32), ("Cm", C.c_double), ("chi", C.c_double),
                ("p", C.c_double * 16), ("field", c_dp), ("field_len", C.c_int64)]


class tb_material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("p", C.c_double * 16), ("f", C.c_double * 3),
                ("s", C.c_double * 3), ("n", C.c_double * 3), ("fsn_field", c_dp), ("fsn_field_len", C.c_int64)]


class tb_hill(C.Structure):
    _fields_ = [("framework", C.c_int32), ("active_energy", C.c_int32), ("active_penalty", C.c_int32), ("adg_kind", C.c_int32),
                ("sarcomere_kind", C.c_int32), ("active_p", C.c_double * 12), ("sheetlet_part", C.c_double), ("sarcomere_p", C.c_double * 2)]


TB_PRECOND_NONE, TB_PRECOND_JACOBI, TB_PRECOND_L1GS, TB_PRECOND_CHEBYSHEV = 0, 1, 2, 3
TB_SWEEP_FORWARD, TB_SWEEP_SYMMETRIC = 0, 2
TB_LOCAL_SUCCESS, TB_LOCAL_LINEAR_SOLVE_FAILED, TB_LOCAL_MAX_ITERS, TB_LOCAL_CONVERGENCE_FAILURE, TB_LOCAL_INFEASIBLE = 0, 1, 2, 3, 4
TB_HILL_NONE, TB_HILL_GENERALIZED, TB_HILL_EXTENDED = 0, 1, 2
TB_ACTIVE_SIMPLE_SPRING = 100
TB_ADG_GMK, TB_ADG_GMK_INCOMPRESSIBLE, TB_ADG_RLRSQ = 0, 1, 2
TB_SARCOMERE_PELCE_SUN_LANGEVELD_1995, TB_SARCOMERE_CONSTANT_STRETCH, TB_SARCOMERE_RDQ20MF = 0, 1, 2

# name -> (restype, argtypes): every symbol include/tbhip.h declares
SIGNATURES = {
    "tb_last_error_string": (C.c_char_p, []),
    "tb_last_kernel_name": (C.c_char_p, []),
    "tb_version": (C.c_char_p, []),
    "tb_abi_revision": (C.c_int, []),
    "tb_device_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "tb_device_destroy": (C.c_int, [vp]),
    "tb_device_set_stream": (C.c_int, [vp, vp]),
    "tb_device_use_null_stream": (C.c_int, [vp]),
    "tb_device_synchronize": (C.c_int, [vp]),
    "tb_device_defer_status": (C.c_int, [vp, C.c_int]),
    "tb_device_poll_status": (C.c_int, [vp]),
    "tb_device_info": (C.c_int, [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "tb_malloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "tb_free": (C.c_int, [vp, vp]),
    "tb_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "tb_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "tb_memcpy_d2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "tb_memset": (C.c_int, [vp, vp, C.c_int, C.c_size_t]),
    "tb_event_create": (C.c_int, [vp, C.POINTER(vp)]),
    "tb_event_record": (C.c_int, [vp, vp]),
    "tb_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    "tb_event_destroy": (C.c_int, [vp]),
    "tb_mesh_create": (C.c_int, [vp, C.c_int, C.c_int64, c_dp, C.c_int64, c_i32p, C.c_int, C.c_int, c_i32p, C.c_int64,
                                 C.c_int, C.POINTER(vp)]),
    "tb_mesh_destroy": (C.c_int, [vp]),
    "tb_mesh_ncells": (C.c_int64, [vp]),
    "tb_mesh_ndofs": (C.c_int64, [vp]),
    "tb_pattern_create": (C.c_int, [vp, C.c_int64, c_i64p, c_i32p, C.c_int, C.POINTER(vp)]),
    "tb_pattern_destroy": (C.c_int, [vp]),
    "tb_pattern_nnz": (C.c_int64, [vp]),
    "tb_pattern_rowptr_device": (vp, [vp]),
    "tb_pattern_colidx_device": (vp, [vp]),
    "tb_form_create": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(tb_coef), C.POINTER(vp)]),
    "tb_form_destroy": (C.c_int, [vp]),
    "tb_form_set_table": (C.c_int, [vp, c_dp, C.c_int64]),
    "tb_assemble_matrix": (C.c_int, [vp, vp, C.c_int, C.c_double, vp]),
    "tb_assemble_matrix_pair": (C.c_int, [vp, vp, vp, C.c_int, C.c_double, vp, vp]),
    "tb_assemble_vector": (C.c_int, [vp, C.c_int, C.c_double, vp]),
    "tb_hyperelastic_create": (C.c_int, [vp, C.c_int, C.POINTER(tb_material), C.POINTER(vp)]),
    "tb_residual": (C.c_int, [vp, C.c_int, vp, C.c_double, vp]),
    "tb_linearize": (C.c_int, [vp, vp, C.c_int, vp, C.c_double, vp, vp]),
    "tb_hyperelastic_set_active_tension": (C.c_int, [vp, C.c_double, c_dp, C.c_int64]),
    "tb_sarcomere_model_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tb_sarcomere_step": (C.c_int, [vp, C.c_int, c_dp, C.c_int, vp, C.c_int64, vp, vp, vp, C.c_double, C.c_double, C.c_double, C.c_double,
                                    C.c_double, C.c_int, C.c_int, vp, vp]),
    "tb_host_sarcomere_eval": (C.c_int, [C.c_int, c_dp, C.c_int, c_dp, C.c_double, C.c_double, C.c_double, c_dp, c_dp, c_dp]),
    "tb_sarcomere_implicit_step": (C.c_int, [vp, C.c_int, c_dp, C.c_int, vp, vp, C.c_int64, vp, vp, vp, C.c_double, C.c_double, C.c_double, C.c_double,
                                             C.c_double, C.c_int, vp, vp, vp, C.POINTER(C.c_int64)]),
    "tb_host_sarcomere_derivatives": (C.c_int, [C.c_int, c_dp, C.c_int, c_dp, C.c_double, C.c_double, C.c_double, C.c_int, c_dp, c_dp, c_dp, c_dp]),
    "tb_host_sarcomere_local_solve": (C.c_int, [C.c_int, c_dp, C.c_int, c_dp, c_dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                                c_dp, c_dp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "tb_hyperelastic_set_previous_solution": (C.c_int, [vp, vp]),
    "tb_hyperelastic_set_prestress": (C.c_int, [vp, c_dp]),
    "tb_form_set_cellset": (C.c_int, [vp, c_i32p, C.c_int64, C.c_int]),
    "tb_form_clear_cellset": (C.c_int, [vp]),
    "tb_form_set_accumulate": (C.c_int, [vp, C.c_int]),
    "tb_hyperelastic_set_condensation": (C.c_int, [vp, C.c_int, c_dp, C.c_int, C.c_double, C.c_double, C.c_int]),
    "tb_hyperelastic_n_quadrature_points": (C.c_int, [vp, C.POINTER(C.c_int64)]),
    "tb_hyperelastic_set_internal_state": (C.c_int, [vp, vp, vp, C.c_double]),
    "tb_hyperelastic_local_solve_report": (C.c_int, [vp, C.POINTER(C.c_int64), vp, C.c_int64]),
    "tb_hyperelastic_set_hill": (C.c_int, [vp, vp]),
    "tb_host_material_eval_hill": (C.c_int, [vp, vp, C.c_double, c_dp, c_dp, c_dp, c_dp]),
    "tb_facet_form_create": (C.c_int, [vp, C.c_int, C.c_double, C.c_int, c_i32p, C.c_int64, C.c_int, C.POINTER(vp)]),
    "tb_facet_form_set_field": (C.c_int, [vp, c_dp, C.c_int64]),
    "tb_facet_form_set_param": (C.c_int, [vp, C.c_double]),
    "tb_facet_assemble": (C.c_int, [vp, vp, vp, C.c_double, vp, vp]),
    "tb_host_material_eval": (C.c_int, [C.POINTER(tb_material), c_dp, c_dp, c_dp, c_dp]),
    "tb_reaction_step": (C.c_int, [vp, C.c_int, c_dp, C.c_int, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_double,
                                   C.c_double, C.c_int, C.c_double]),
    "tb_reaction_step_x": (C.c_int, [vp, C.c_int, c_dp, C.c_int, vp, vp, C.c_int64, C.c_int, C.c_int, vp, C.c_int, C.c_double,
                                     C.c_double, C.c_int, C.c_double]),
    "tb_reaction_step_rtc": (C.c_int, [vp, C.c_int, c_dp, C.c_int, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_double,
                                       C.c_double, C.c_int, C.c_double, c_dp]),
    "tb_reaction_step_rl": (C.c_int, [vp, C.c_int, c_dp, C.c_int, vp, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double]),
    "tb_cell_model_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tb_cell_model_defaults": (C.c_int, [C.c_int, c_dp, c_dp]),
    "tb_heat_matrix": (C.c_int, [vp, C.c_int64, vp, vp, C.c_double, vp]),
    "tb_spmv_csr": (C.c_int, [vp, vp, vp, C.c_double, C.c_double, vp]),
    "tb_cg_solve": (C.c_int, [vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "tb_cg_solve_from_residual": (C.c_int, [vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "tb_pcg_solve": (C.c_int, [vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "tb_l1gs_apply": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "tb_gmres_solve": (C.c_int, [vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "tb_solver_last_tolerance": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "tb_axpy": (C.c_int, [vp, C.c_int64, C.c_double, vp, vp]),
    "tb_absmax": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.POINTER(C.c_double)]),
    "tb_dot": (C.c_int, [vp, C.c_int64, vp, vp, C.POINTER(C.c_double)]),
    "tb_cgd_dot": (C.c_int, [vp, C.c_int64, vp, vp, vp, vp]),
    "tb_cgd_update": (C.c_int, [vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "tb_cgd_direction": (C.c_int, [vp, C.c_int64, vp, vp, vp, vp, vp]),
    "tb_cgd_rotate": (C.c_int, [vp, vp]),
    "tb_gather_indexed": (C.c_int, [vp, C.c_int64, vp, vp, vp]),
    "tb_convert_f64_to_f32": (C.c_int, [vp, C.c_int64, vp, vp]),
    "tb_convert_f32_to_f64": (C.c_int, [vp, C.c_int64, vp, vp]),
    "tb_assemble_matrix_f32": (C.c_int, [vp, vp, C.c_int, C.c_double, vp]),
    "tb_assemble_matrix_pair_f32": (C.c_int, [vp, vp, vp, C.c_int, C.c_double, vp, vp]),
    "tb_assemble_vector_f32": (C.c_int, [vp, C.c_int, C.c_double, vp]),
    "tb_reaction_step_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int64, C.c_int, C.c_int, vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double]),
    "tb_spmv_csr_f32": (C.c_int, [vp, vp, vp, C.c_double, C.c_double, vp]),
    "tb_heat_matrix_f32": (C.c_int, [vp, C.c_int64, vp, vp, C.c_double, vp]),
    "tb_axpy_f32": (C.c_int, [vp, C.c_int64, C.c_double, vp, vp]),
    "tb_cg_solve_f32": (C.c_int, [vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, C.c_int, vp, vp]),
    "tb_extract_diagonal": (C.c_int, [vp, vp, vp]),
    "tb_pattern_patch_stats": (C.c_int, [vp, vp]),
    "tb_comm_unique_id": (C.c_int, [vp]),
    "tb_comm_create": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "tb_comm_destroy": (C.c_int, [vp]),
    "tb_comm_rank_size": (C.c_int, [vp, vp, vp]),
    "tb_comm_exchange": (C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
    "tb_comm_allreduce": (C.c_int, [vp, vp, C.c_int64, C.c_int]),
    "tb_comm_exchange_begin": (C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
    "tb_cgd_iteration": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "tb_graph_begin": (C.c_int, [vp]),
    "tb_graph_end": (C.c_int, [vp, C.POINTER(vp)]),
    "tb_graph_launch": (C.c_int, [vp, C.c_double]),
    "tb_graph_node_count": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "tb_graph_destroy": (C.c_int, [vp]),
    "tb_comm_exchange_end": (C.c_int, [vp]),
    "tb_pattern_spmv_plan": (C.c_int, [vp, vp]),
    "tb_scatter_add_indexed": (C.c_int, [vp, C.c_int64, vp, vp, vp]),
    "tb_scatter_indexed": (C.c_int, [vp, C.c_int64, vp, vp, vp]),
    "tb_spmv_csr_rows": (C.c_int, [vp, vp, vp, C.c_int64, vp, vp]),
    "tb_spmv_csr_dot": (C.c_int, [vp, vp, vp, vp, vp]),
    "tb_spmv_mirror": (C.c_int, [vp, vp]),
    "tb_apply_zero_csr": (C.c_int, [vp, vp, vp, vp, C.c_double]),
    "tb_meandiag": (C.c_int, [vp, vp, C.POINTER(C.c_double)]),
    "tb_max": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.POINTER(C.c_double)]),
    "tb_host_generate_grid_hex": (C.c_int, [C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_i32p]),
    "tb_host_generate_grid_quad": (C.c_int, [C.c_int, C.c_int, c_dp, c_dp, c_dp, c_i32p]),
    "tb_host_perturb_nodes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, c_dp]),
    "tb_host_close_dofs": (C.c_int64, [C.c_int, C.c_int, C.c_int64, C.c_int64, c_i32p, c_i32p]),
    "tb_host_build_pattern": (C.c_int64, [C.c_int64, C.c_int, c_i32p, C.c_int64, c_i64p, c_i32p]),
    "tb_host_locality_permutation": (C.c_int, [C.c_int, C.c_int64, c_dp, C.c_int64, c_i32p, C.c_int, c_i32p, C.c_int64, C.c_int, c_i32p, c_i32p, c_i32p]),
}


class TBError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libtbhip error %d: %s" % (code, msg))
        self.code = code


def build_library(force=False):
    """Compile thunderbolt.jl_amd/csrc for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", csrc, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


TB_ABI_REVISION = 6   # include/tbhip.h: TB_ABI_REVISION


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "thunderbolt.jl_amd: %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C thunderbolt.jl_amd/csrc). There is no CPU fallback." % LIB_PATH)
        try:  # share one HIP runtime with the host framework when it is present (same soname)
            import torch  # noqa: F401
        except Exception:  # pragma: no cover
            pass
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        if _lib.tb_abi_revision() != TB_ABI_REVISION:
            raise ImportError("thunderbolt.jl_amd: %s has ABI revision %d, this binding was written against %d — rebuild (make -C thunderbolt.jl_amd/csrc)"
                              % (LIB_PATH, _lib.tb_abi_revision(), TB_ABI_REVISION))
    return _lib


def check(rc):
    if rc != TB_OK:
        raise TBError(rc, lib().tb_last_error_string().decode("utf-8", "replace"))
    return rc
