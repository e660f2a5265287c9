"""ctypes binding of libmadrl_hip.so (include/madrl_hip.h).

`import torch` happens BEFORE the library is dlopen'ed: PyTorch-ROCm bundles its own
libamdhip64.so and the loader must reuse that already-mapped runtime, otherwise tensor
data_ptr()s and torch's stream handles would not be valid inside our kernels
("HIP runtime loading rule").

There is no fallback: if the shared library is missing this module raises, and every
environment class in madrl_amd fails loudly with it.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("MADRL_HIP_LIB") or os.path.join(_HERE, "libmadrl_hip.so")  # override: profiling variants only
ABI_VERSION = 7
POLICY_COUNTER_WORDS = 32 * 65   # MADRL_POLICY_COUNTER_WORDS of include/madrl_hip.h

KERNEL_AUTO, KERNEL_GENERIC, KERNEL_WAVE = 0, 1, 2

_lib = None


class MadrlError(RuntimeError):
    pass


class PursuitConfig(C.Structure):
    """mirror of madrl_pursuit_config (include/madrl_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "xs", "ys", "n_pursuers", "n_evaders", "obs_range", "n_catch", "surround",
        "flatten", "include_id", "reward_global", "sample_maps", "n_maps", "max_steps",
        "auto_reset", "max_opponents", "control_evaders", "reserved0")] + [(n, C.c_double) for n in (
            "catchr", "term_pursuit", "urgency_reward", "layer_norm", "constraint_window")] + [
                ("seed", C.c_uint64), ("env_id_base", C.c_int64)]


class WaterworldConfig(C.Structure):
    """mirror of madrl_waterworld_config (include/madrl_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "n_pursuers", "n_evaders", "n_coop", "n_poison", "n_sensors", "addid",
        "speed_features", "reward_global", "obstacle_fixed", "max_steps", "auto_reset", "reserved0")] + [
            (n, C.c_double) for n in (
                "radius", "obstacle_radius", "ev_speed", "poison_speed", "sensor_range", "action_scale",
                "poison_reward", "food_reward", "encounter_reward", "control_penalty")] + [
                    ("obstacle_loc", C.c_double * 2), ("seed", C.c_uint64), ("env_id_base", C.c_int64)]


class HostageConfig(C.Structure):
    """mirror of madrl_hostage_config (include/madrl_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "n_good", "n_hostages", "n_bad", "n_coop_save", "n_coop_avoid", "n_sensors", "addid", "reward_global",
        "key_fixed", "max_steps", "auto_reset")] + [(n, C.c_double) for n in (
            "radius", "bad_speed", "sensor_range", "action_scale", "save_reward", "hit_reward", "encounter_reward", "not_saved_reward",
            "bomb_reward", "bomb_radius", "key_radius", "control_penalty")] + [
                ("key_loc", C.c_double * 2), ("seed", C.c_uint64), ("env_id_base", C.c_int64)]


class MultiWalkerConfig(C.Structure):
    """mirror of madrl_multiwalker_config (include/madrl_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "n_walkers", "reward_global", "terminate_on_fall", "one_hot", "max_steps", "auto_reset",
        "discrete_only", "polygon_revision", "reserved0")] + [(n, C.c_double) for n in (
            "position_noise", "angle_noise", "forward_reward", "fall_reward", "drop_reward")] + [
                ("seed", C.c_uint64), ("env_id_base", C.c_int64)]


class PursuitShardIO(C.Structure):
    """mirror of madrl_pursuit_shard_io (include/madrl_hip.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("actions", "inj_evader_actions", "obs", "rew", "done", "removed", "stream")]


class StandardizeArgs(C.Structure):
    """mirror of madrl_standardize_args (include/madrl_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in ("struct_size", "enable_obsnorm", "enable_rewnorm", "reserved0")] + [
        (n, C.c_double) for n in ("obs_alpha", "rew_alpha", "eps", "scale_reward")] + [
        (n, C.c_void_p) for n in ("obs_mean", "obs_var", "obs_out", "rew_mean", "rew_var", "rew_out")]


_vp = C.c_void_p

# name -> (restype, argtypes); this table is also what tests use to check that the library
# exports every symbol the header declares.
SIGNATURES = {
    "madrl_abi_version": (C.c_int, []),
    "madrl_last_error": (C.c_char_p, []),
    "madrl_philox4x32_10": (None, [_vp, _vp, _vp]),
    "madrl_pursuit_obs_dim": (C.c_int, [_vp, _vp]),
    "madrl_pursuit_state_bytes": (C.c_int, [_vp, C.c_int64, _vp]),
    "madrl_pursuit_record_bytes": (C.c_int, [_vp, _vp]),
    "madrl_pursuit_flags_offset": (C.c_int, [_vp, C.c_int64, _vp]),
    "madrl_pursuit_invalidate_obs": (C.c_int, [_vp]),
    "madrl_pursuit_declare_obs_zero": (C.c_int, [_vp, _vp, _vp]),
    "madrl_pursuit_set_params": (C.c_int, [_vp, C.c_double, C.c_double]),
    "madrl_pursuit_set_curriculum": (C.c_int, [_vp, _vp, _vp]),
    "madrl_pursuit_create": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, _vp, _vp]),
    "madrl_pursuit_destroy": (None, [_vp]),
    "madrl_pursuit_set_launch": (C.c_int, [_vp, C.c_int32, C.c_int64]),
    "madrl_pursuit_set_kernel": (C.c_int, [_vp, C.c_int32]),
    "madrl_pursuit_kernel_kind": (C.c_int, [_vp, _vp]),
    "madrl_pursuit_reset": (C.c_int, [_vp] * 6),
    "madrl_pursuit_step": (C.c_int, [_vp] * 8),
    "madrl_pursuit_get_state": (C.c_int, [_vp] * 10),
    "madrl_pursuit_set_state": (C.c_int, [_vp] * 10),
    "madrl_waterworld_obs_dim": (C.c_int, [_vp, _vp]),
    "madrl_waterworld_state_bytes": (C.c_int, [_vp, C.c_int64, _vp]),
    "madrl_waterworld_create": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, _vp, _vp]),
    "madrl_waterworld_destroy": (None, [_vp]),
    "madrl_waterworld_set_launch": (C.c_int, [_vp, C.c_int64]),
    "madrl_waterworld_set_standardize": (C.c_int, [_vp, _vp]),
    "madrl_waterworld_reset": (C.c_int, [_vp] * 4),
    "madrl_waterworld_step": (C.c_int, [_vp] * 8),
    "madrl_waterworld_get_state": (C.c_int, [_vp] * 7),
    "madrl_waterworld_set_state": (C.c_int, [_vp] * 7),
    "madrl_hostage_obs_dim": (C.c_int, [_vp, _vp]),
    "madrl_hostage_state_bytes": (C.c_int, [_vp, C.c_int64, _vp]),
    "madrl_hostage_create": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, _vp, _vp]),
    "madrl_hostage_destroy": (None, [_vp]),
    "madrl_hostage_set_launch": (C.c_int, [_vp, C.c_int64]),
    "madrl_hostage_reset": (C.c_int, [_vp] * 4),
    "madrl_hostage_step": (C.c_int, [_vp] * 8),
    "madrl_hostage_get_state": (C.c_int, [_vp] * 10),
    "madrl_hostage_set_state": (C.c_int, [_vp] * 10),
    "madrl_pursuit_step_sharded": (C.c_int, [_vp, _vp, C.c_int32, _vp, C.c_int32, C.c_int32]),
    "madrl_multiwalker_obs_dim": (C.c_int, [_vp, _vp]),
    "madrl_multiwalker_state_bytes": (C.c_int, [_vp, C.c_int64, _vp]),
    "madrl_multiwalker_create": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, _vp]),
    "madrl_multiwalker_destroy": (None, [_vp]),
    "madrl_multiwalker_set_mode": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "madrl_pursuit_set_walk": (C.c_int, [_vp, C.c_int32]),
    "madrl_multiwalker_dims": (C.c_int, [_vp, _vp, _vp]),
    "madrl_multiwalker_lanes": (C.c_int, [_vp, _vp, _vp]),
    "madrl_multiwalker_record_bytes": (C.c_int, [_vp, _vp, _vp]),
    "madrl_multiwalker_reset": (C.c_int, [_vp] * 4),
    "madrl_multiwalker_step": (C.c_int, [_vp] * 6),
    "madrl_multiwalker_get_bodies": (C.c_int, [_vp] * 5),
    "madrl_multiwalker_get_state": (C.c_int, [_vp] * 7),
    "madrl_multiwalker_set_state": (C.c_int, [_vp] * 4),
    "madrl_multiwalker_reset_with": (C.c_int, [_vp] * 6),
    "madrl_wrap_obsnorm": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, _vp, C.c_double, C.c_double, _vp]),
    "madrl_wrap_rewnorm": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, _vp, C.c_double, C.c_double, C.c_double, C.c_int32, _vp]),
    "madrl_wrap_obsbuffer": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _vp, _vp, _vp]),
    "madrl_wrap_diagnostics": (C.c_int, [_vp] * 6 + [C.c_int64, C.c_int32, C.c_double, C.c_int32] + [_vp] * 5),
    "madrl_heuristic_pursuit": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _vp, C.c_uint64, C.c_int64, C.c_uint32, _vp, _vp, _vp]),
    "madrl_heuristic_waterworld": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, _vp, _vp]),
    "madrl_heuristic_multiwalker": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, _vp]),
    "madrl_rollout_gae": (C.c_int, [_vp] * 3 + [C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_double] + [_vp] * 3),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise MadrlError(
                "libmadrl_hip.so not found at %s -- build it with `python -m madrl_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % SO_PATH)
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        if L.madrl_abi_version() != ABI_VERSION:
            raise MadrlError("libmadrl_hip.so ABI %d != expected %d" % (L.madrl_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().madrl_last_error()
        raise MadrlError("madrl error %d: %s" % (rc, msg.decode() if msg else "?"))


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor passed across the C ABI must be contiguous"
    return C.c_void_p(t.data_ptr())


def current_stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
