"""ctypes binding of the C-ABI in include/vggsfm_amd.h (libvggsfm_amd.so, built for gfx950).

There is NO fallback: if the shared library is missing or was built for another target, importing
a product function that needs it raises.  torch is imported first so that the library resolves
``libamdhip64.so.7`` to the HIP runtime torch already loaded (same streams, same allocations).
"""
import ctypes
import os

import torch  # noqa: F401  (must be loaded before the HIP library, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VGGSFM_AMD_LIB: an alternative build of the same library, for A/B measurements of kernel variants)
LIB_PATH = os.environ.get("VGGSFM_AMD_LIB") or os.path.join(_HERE, "libvggsfm_amd.so")

VGG_OK = 0
ABI_VERSION = 2       # VGG_ABI_VERSION of include/vggsfm_amd.h this binding was written against
_ERRORS = {-1: "invalid argument", -2: "HIP runtime error", -3: "workspace too small / missing",
           -4: "unsupported configuration"}


class BAProblem(ctypes.Structure):
    _fields_ = [("num_cams", ctypes.c_int32), ("num_pts", ctypes.c_int32), ("num_obs", ctypes.c_int32),
                ("num_intr", ctypes.c_int32), ("camera_model", ctypes.c_int32), ("refine_focal", ctypes.c_int32),
                ("refine_extra", ctypes.c_int32), ("loss", ctypes.c_int32), ("loss_scale", ctypes.c_double),
                ("cam_q", ctypes.c_void_p), ("cam_t", ctypes.c_void_p), ("intr", ctypes.c_void_p),
                ("pts", ctypes.c_void_p), ("row_ptr", ctypes.c_void_p), ("obs_cam", ctypes.c_void_p),
                ("obs_uv", ctypes.c_void_p), ("col_ptr", ctypes.c_void_p), ("cobs_pt", ctypes.c_void_p),
                ("cobs_uv", ctypes.c_void_p), ("cam_const", ctypes.c_void_p), ("intr_const", ctypes.c_void_p),
                ("pt_const", ctypes.c_void_p), ("num_chunks", ctypes.c_int32), ("num_tile_batches", ctypes.c_int32),
                ("chunk_desc", ctypes.c_void_p),
                ("entries", ctypes.c_void_p), ("num_segments", ctypes.c_int32), ("obs_slot", ctypes.c_void_p),
                ("num_tiles", ctypes.c_int32), ("tile_desc", ctypes.c_void_p), ("tile_batches", ctypes.c_void_p),
                ("chol_split_a", ctypes.c_int32), ("chol_split_b", ctypes.c_int32), ("chol_first_blk", ctypes.c_void_p), ("merged_tile_launch", ctypes.c_int32)]


class BAOptions(ctypes.Structure):
    _fields_ = [("max_num_iterations", ctypes.c_int32), ("max_num_consecutive_invalid_steps", ctypes.c_int32),
                ("jacobi_scaling", ctypes.c_int32), ("function_tolerance", ctypes.c_double),
                ("gradient_tolerance", ctypes.c_double), ("parameter_tolerance", ctypes.c_double),
                ("initial_trust_region_radius", ctypes.c_double), ("max_trust_region_radius", ctypes.c_double),
                ("min_trust_region_radius", ctypes.c_double), ("min_lm_diagonal", ctypes.c_double),
                ("max_lm_diagonal", ctypes.c_double), ("min_relative_decrease", ctypes.c_double),
                ("overlap_factorization", ctypes.c_int32)]


class BAIteration(ctypes.Structure):
    _fields_ = [("iteration", ctypes.c_int32), ("successful", ctypes.c_int32), ("cost", ctypes.c_double),
                ("cost_change", ctypes.c_double), ("gradient_max_norm", ctypes.c_double),
                ("step_norm", ctypes.c_double), ("relative_decrease", ctypes.c_double), ("radius", ctypes.c_double)]


class BASummary(ctypes.Structure):
    _fields_ = [("initial_cost", ctypes.c_double), ("final_cost", ctypes.c_double),
                ("num_iterations", ctypes.c_int32), ("num_successful_steps", ctypes.c_int32),
                ("num_unsuccessful_steps", ctypes.c_int32), ("termination", ctypes.c_int32),
                ("n_reduced", ctypes.c_int32), ("num_log", ctypes.c_int32)]


# every symbol include/vggsfm_amd.h declares (tests check the library exports all of them)
EXPORTED = ["vgg_build_arch", "vgg_abi_version", "vgg_abi_sizeof", "vgg_project_points", "vgg_filter_points_workspace_bytes",
            "vgg_filter_points", "vgg_cam_from_img_workspace_bytes", "vgg_cam_from_img",
            "vgg_triangulate_workspace_bytes", "vgg_triangulate_tracks", "vgg_triangulate_by_pair", "vgg_triangulate_chunks_workspace_bytes",
            "vgg_triangulate_tracks_chunks", "vgg_ba_workspace_bytes", "vgg_ba_solve",
            "vgg_ba_begin", "vgg_ba_phase", "vgg_ba_reduce_buffer", "vgg_ba_finish", "vgg_cholesky_solve",
            "vgg_ba_profile", "vgg_ba_profile_read", "vgg_cholesky_workspace_bytes", "vgg_pose_refine",
            "vgg_p3p_ransac_workspace_bytes", "vgg_p3p_ransac", "vgg_fmat_seven_point", "vgg_fmat_score",
            "vgg_fmat_eight_point", "vgg_fmat_residuals", "vgg_cholesky_solve_split", "vgg_ba_poll_done", "vgg_ba_tuning", "vgg_cholesky_solve_envelope", "vgg_ba_set_tile_rhs", "vgg_ba_set_step_from_factors", "vgg_triangulate_tracks_chunks_enqueue", "vgg_ba_set_tile_dma"]

_lib = None


def lib():
    """Load the HIP library or fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the gfx950 HIP library has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C vggsfm_amd/csrc`). "
            "vggsfm_amd has no CPU or eager fallback.")
    L = ctypes.CDLL(LIB_PATH)
    L.vgg_build_arch.restype = ctypes.c_char_p
    arch = L.vgg_build_arch().decode()
    if arch != "gfx950":
        raise RuntimeError(f"libvggsfm_amd.so was built for {arch}, expected gfx950")
    for name in ("vgg_filter_points_workspace_bytes", "vgg_cam_from_img_workspace_bytes",
                 "vgg_triangulate_workspace_bytes", "vgg_triangulate_chunks_workspace_bytes", "vgg_ba_workspace_bytes",
                 "vgg_cholesky_workspace_bytes", "vgg_p3p_ransac_workspace_bytes"):
        getattr(L, name).restype = ctypes.c_size_t
    L.vgg_ba_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    # a stale or variant build (VGGSFM_AMD_LIB) with another struct layout would be driven with shifted pointers: refuse it
    if not hasattr(L, "vgg_abi_sizeof") or L.vgg_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: ABI version {L.vgg_abi_version()} but this binding speaks {ABI_VERSION}; rebuild "
                           "the library (`make -C vggsfm_amd/csrc`)")
    L.vgg_abi_sizeof.restype = ctypes.c_size_t
    for which, st in enumerate((BAProblem, BAOptions, BAIteration, BASummary)):
        if int(L.vgg_abi_sizeof(which)) != ctypes.sizeof(st):
            raise RuntimeError(f"{LIB_PATH}: sizeof({st.__name__}) is {int(L.vgg_abi_sizeof(which))} in the library and "
                               f"{ctypes.sizeof(st)} in the binding -- header and binding are out of step")
    _lib = L
    return L


def check(rc, what):
    if rc != VGG_OK:
        raise RuntimeError(f"{what} failed: {_ERRORS.get(rc, rc)}")


def ptr(t):
    """Device (or host) address of a tensor as void*; None -> NULL."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vggsfm_amd kernels need tensors on an MI355X (cuda) device; there is no CPU path")
