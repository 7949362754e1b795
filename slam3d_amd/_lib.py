"""Loader of the C-ABI library (slam3d_amd/lib/libparticlefilter.so).

There is no Python or CPU implementation behind this package: if the library has not been built
(`make -C slam3d_amd/csrc`, or `python -c "import __graft_entry__ as g; g.build()"`) importing the
product modules fails, and creating a filter without a visible gfx950 device raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PFX_LIB_PATH: another build of the same library (kernel experiments, tools/dev); the product path is the default
LIB_PATH = os.environ.get("PFX_LIB_PATH") or os.path.join(_HERE, "lib", "libparticlefilter.so")

_lib = None


class PfxError(RuntimeError):
    pass


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C slam3d_amd/csrc` "
                          "(there is no fallback implementation)")
    # PyTorch's ROCm wheel ships its own copies of the HIP / HSA runtime libraries.  A process can hold only one of
    # them: if this library (linked against /opt/rocm) initialises the GPU first, torch.cuda later finds "no HIP GPUs".
    # Loading torch first makes both share torch's copies (same SONAMEs), which is the order bench.py and the sharded
    # filter have always used.  Set PFX_PRELOAD_TORCH=0 in processes that will never touch torch.cuda.
    if os.environ.get("PFX_PRELOAD_TORCH", "1") != "0":
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, f, d, i, i64, u32 = C.c_void_p, C.c_float, C.c_double, C.c_int, C.c_int64, C.c_uint32
    pf_, pd_ = C.POINTER(C.c_float), C.POINTER(C.c_double)

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("pfx_version", C.c_char_p)
    sig("pfx_last_error", C.c_char_p)
    sig("pfx_device_count", i)
    sig("pfx_set_abort_on_error", None, i)
    sig("pfx_error_count", C.c_longlong)
    sig("pfx_loc_create", vp, i64, u32, u32, i)
    sig("pfx_loc_destroy", None, vp)
    sig("pfx_loc_clone", vp, vp)
    sig("pfx_loc_reset", i, vp, u32, u32)
    sig("pfx_loc_n", i64, vp)
    sig("pfx_loc_device", i, vp)
    sig("pfx_loc_seed", u32, vp)
    sig("pfx_loc_stream_id", u32, vp)
    sig("pfx_loc_set_stream", i, vp, vp)
    sig("pfx_loc_sync", i, vp)
    sig("pfx_loc_set_force_branch", i, vp, i)
    sig("pfx_loc_set_deferred", i, vp, i)
    sig("pfx_loc_deposit_vio", None, vp, d, f, f, f, f)
    sig("pfx_loc_deposit_range", i, vp, f, f, f, f, f)
    sig("pfx_loc_deposit_rssi", i, vp, f, f, f, i)
    sig("pfx_loc_get_tag_loc", i, vp, pd_, pf_, pf_, pf_, pf_)
    sig("pfx_loc_upload", i, vp, vp, vp, vp, vp, vp, i64, i)
    sig("pfx_loc_download", i, vp, vp, vp, vp, vp, vp, i64)
    sig("pfx_loc_initialized", i, vp)
    sig("pfx_loc_epoch", u32, vp)
    sig("pfx_loc_set_epoch", None, vp, u32)
    sig("pfx_loc_device_arrays", i, vp, C.POINTER(vp), pf_)
    sig("pfx_loc_apply_vio", i, vp, f, f, f, f, f)
    sig("pfx_loc_apply_range", i, vp, f, f, f, f, f)
    sig("pfx_loc_resample", i, vp, f, f, f, f, f)
    sig("pfx_loc_spawn_all", i, vp, f, f, f, f, f)
    sig("pfx_loc_last_stats", i, vp, pd_)
    sig("pfx_loc_record_ancestors", i, vp, i)
    sig("pfx_loc_last_ancestors", i, vp, vp, i64)
    sig("pfx_resample_indices_f32", i, vp, i64, f, f, vp, i)
    sig("pfx_resample_indices_f64", i, vp, i64, d, d, vp, i)
    sig("pfx_loc_profile_begin", i, vp, i)
    sig("pfx_loc_profile_mask", i, vp, C.c_uint)
    sig("pfx_loc_profile_stride", i, vp, i)
    sig("pfx_loc_profile_read", i, vp, pd_, C.POINTER(i64))
    sig("pfx_loc_profile_records", i64, vp, vp, vp, i64)
    sig("pfx_loc_branch_log", i64, vp, vp, i64, C.POINTER(i64))
    sig("pfx_loc_shard_config", i, vp, i, i, i64, i64, i64)
    sig("pfx_loc_shard_begin", i, vp, f, f, f, f, f)
    sig("pfx_loc_shard_deposit", i, vp)
    sig("pfx_loc_shard_stage", i, vp)
    sig("pfx_loc_shard_local_ptr", vp, vp)
    sig("pfx_loc_shard_all_ptr", vp, vp)
    sig("pfx_loc_shard_row_doubles", i64, vp)
    sig("pfx_loc_shard_decide", i, vp)
    sig("pfx_loc_shard_plan", i, vp, C.POINTER(i), C.POINTER(i64))
    sig("pfx_loc_shard_pending", i, vp)
    sig("pfx_loc_shard_produce", i, vp)
    sig("pfx_loc_shard_land", i, vp, i, i64, i64)
    sig("pfx_loc_shard_finish", i, vp, i, i64, i64)
    sig("pfx_loc_shard_ptrs_both", i, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp))
    sig("pfx_loc_shard_cur", i, vp)
    sig("pfx_loc_shard_pose_sums", i, vp, pd_)
    sig("pfx_loc_pose_from_sums", i, vp, pd_, pd_, pf_, pf_, pf_, pf_)
    sig("pfx_rccl_unique_id", i, vp)
    sig("pfx_loc_shard_comm_init", i, vp, vp)
    sig("pfx_loc_shard_comm_active", i, vp)
    sig("pfx_loc_shard_comm_destroy", i, vp)
    sig("pfx_loc_shard_deposit_range", i, vp, f, f, f, f, f)
    sig("pfx_loc_shard_settle", i, vp)
    sig("pfx_loc_shard_get_tag_loc", i, vp, pd_, pf_, pf_, pf_, pf_)
    sig("pfx_loc_shard_exchanged", i64, vp)
    sig("pfx_shard_exchange_plan", i, vp, i64, i, i, vp, vp, vp, vp, C.POINTER(i64), C.POINTER(i64))
    sig("pfx_pose_sums_noalloc", i, vp, vp, vp, vp, vp, i64, f, vp, vp, i)
    sig("pfx_loc_of", vp, vp)
    sig("pfx_loc_sync_host", i, vp)
    sig("pfx_loc_release", None, vp)
    # reference-layout entry points (include/particleFilter.h)
    sig("particleFilterSeed_set", None, C.c_uint)
    sig("particleFilterLoc_init", None, vp)
    sig("particleFilterLoc_depositVio", None, vp, d, f, f, f, f)
    sig("particleFilterLoc_depositRange", None, vp, f, f, f, f, f)
    sig("particleFilterLoc_depositRssi", None, vp, f, f, f, i)
    sig("particleFilterLoc_getTagLoc", C.c_uint8, vp, pd_, pf_, pf_, pf_, pf_)
    # many Loc filters in one set of launches (pfx_loc_batch_*)
    sig("pfx_loc_batch_create", vp, i, i64, vp, u32, vp, i)
    sig("pfx_loc_batch_destroy", None, vp)
    sig("pfx_loc_batch_num_filters", i, vp)
    sig("pfx_loc_batch_n", i64, vp)
    sig("pfx_loc_batch_seed", u32, vp, i)
    sig("pfx_loc_batch_stream_id", u32, vp, i)
    sig("pfx_loc_batch_epoch", u32, vp, i)
    sig("pfx_loc_batch_initialized", i, vp, i)
    sig("pfx_loc_batch_sync", i, vp)
    sig("pfx_loc_batch_set_force_branch", i, vp, i)
    sig("pfx_loc_batch_deposit_vio", i, vp, vp, vp, vp, vp, vp, vp)
    sig("pfx_loc_batch_deposit_range", i, vp, vp, vp, vp, vp, vp, vp)
    sig("pfx_loc_batch_deposit_rssi", i, vp, vp, vp, vp, vp, vp)
    sig("pfx_loc_batch_get_tag_loc", i, vp, vp, vp, vp, vp, vp, vp)
    sig("pfx_loc_batch_download", i, vp, i, vp, vp, vp, vp, vp, i64)
    sig("pfx_loc_batch_last_stats", i, vp, i, pd_)
    sig("pfx_loc_batch_branch_counts", i, vp, C.POINTER(i64), C.POINTER(i64), vp, vp)
    sig("pfx_gate_bounds_both", i, vp, vp, i, vp, vp, vp, vp, i)
    # the Loc stage functions of the reference's internal headers (include/pfStages.h)
    sig("pfRandom_init", None)
    sig("pfInit_initTagLoc", None, vp, f, f, f, f, f)
    sig("pfMeasurement_applyVioLoc", None, vp, f, f, f, f, f)
    sig("pfMeasurement_applyRangeLoc", None, vp, f, f, f, f, f)
    sig("pfResample_resampleLoc", None, vp, f, f, f, f, f)
    # batched SLAM filters (pfx_slam_*)
    u8p = C.POINTER(C.c_uint8)
    sig("pfx_slam_create", vp, i, i, i, i, vp, u32, i)
    sig("pfx_slam_destroy", None, vp)
    sig("pfx_slam_num_filters", i, vp)
    sig("pfx_slam_n_tag", i, vp)
    sig("pfx_slam_n_bcn", i, vp)
    sig("pfx_slam_max_bcns", i, vp)
    sig("pfx_slam_sync", i, vp)
    sig("pfx_slam_reset", i, vp, i, u32)
    sig("pfx_slam_add_bcn", i, vp, i)
    sig("pfx_slam_bcn_initialized", i, vp, i, i)
    sig("pfx_slam_epoch", u32, vp, i)
    sig("pfx_slam_set_epoch", None, vp, i, u32)
    sig("pfx_slam_seed", u32, vp, i)
    sig("pfx_slam_deposit_tag_vio", i, vp, i, d, f, f, f, f)
    sig("pfx_slam_deposit_bcn_vio", i, vp, i, i, d, f, f, f, f)
    sig("pfx_slam_deposit_range", i, vp, i, i, f, f)
    sig("pfx_slam_deposit_rssi", i, vp, i, i, i)
    sig("pfx_slam_get_tag_loc", i, vp, i, pd_, pf_, pf_, pf_, pf_)
    sig("pfx_slam_get_bcn_loc", i, vp, i, i, pd_, pf_, pf_, pf_, pf_)
    sig("pfx_slam_last_tag_branch", i, vp, i)
    sig("pfx_slam_deposit_tag_vio_batch", i, vp, vp, vp, vp, vp, vp)
    sig("pfx_slam_deposit_range_batch", i, vp, vp, vp, vp)
    sig("pfx_slam_get_tag_loc_batch", i, vp, vp, vp, vp, vp, vp)
    sig("pfx_slam_get_bcn_loc_batch", i, vp, vp, vp, vp, vp, vp, vp, vp)
    sig("pfx_slam_device_arrays", i, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int32))
    sig("pfx_slam_row_census", i, vp, C.POINTER(i64))
    sig("pfx_slam_download_tag", i, vp, i, vp, vp, vp, vp, vp)
    sig("pfx_slam_upload_tag", i, vp, i, vp, vp, vp, vp, vp)
    sig("pfx_slam_download_bcn", i, vp, i, i, vp, vp, vp, vp, vp)
    sig("pfx_slam_upload_bcn", i, vp, i, i, vp, vp, vp, vp, vp, i)
    sig("pfx_slam_of", vp, vp)
    sig("pfx_slam_slot_of", i, vp, vp)
    sig("pfx_slam_sync_host", i, vp, vp)
    sig("pfx_slam_release", None, vp)
    sig("particleFilterSlam_init", None, vp)
    sig("particleFilterSlam_addBcn", None, vp)
    sig("particleFilterSlam_depositTagVio", None, vp, d, f, f, f, f)
    sig("particleFilterSlam_depositBcnVio", None, vp, d, f, f, f, f)
    sig("particleFilterSlam_depositRange", None, vp, vp, f, f, vp, i)
    sig("particleFilterSlam_depositRssi", None, vp, vp, i, vp, i)
    sig("particleFilterSlam_getTagLoc", C.c_uint8, vp, pd_, pf_, pf_, pf_, pf_)
    sig("particleFilterSlam_getBcnLoc", C.c_uint8, vp, vp, pd_, pf_, pf_, pf_, pf_)
    _lib = L
    return L


def last_error() -> str:
    return lib().pfx_last_error().decode()


def check(rc: int, what: str) -> int:
    if rc < 0:
        raise PfxError(f"{what}: {last_error()}")
    return rc
