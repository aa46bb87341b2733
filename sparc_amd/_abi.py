"""ctypes mirror of ``include/wedm_hip.h`` (ABI version = ``ABI_VERSION`` below, cross-checked against the library by the loader).

Pure declarations — no library is loaded here, so this module imports on a
GPU-less box.  Field order and types must match the header exactly; the loader
(`sparc_amd._lib`) cross-checks ``sizeof(wedm_params)`` against the library.
"""
from __future__ import annotations

import ctypes as C
import enum

ABI_VERSION = 4
MAX_MODE = 19


def t_quads(n_seg_max: int) -> int:
    """``WEDM_T_QUADS``: 16-byte words (rows of the quad-interleaved T block) a wire of n_seg_max segments occupies."""
    return (int(n_seg_max) + 3) >> 2


# status codes -----------------------------------------------------------------
OK = 0
ERR_BAD_ARG = -1
ERR_NOT_BOUND = -2
ERR_HIP = -3
ERR_NO_DEVICE = -4
ERR_BAD_MODE = -5
ERR_UNSUPPORTED = -6

STATUS_NAMES = {
    OK: "WEDM_OK",
    ERR_BAD_ARG: "WEDM_ERR_BAD_ARG",
    ERR_NOT_BOUND: "WEDM_ERR_NOT_BOUND",
    ERR_HIP: "WEDM_ERR_HIP",
    ERR_NO_DEVICE: "WEDM_ERR_NO_DEVICE",
    ERR_BAD_MODE: "WEDM_ERR_BAD_MODE",
    ERR_UNSUPPORTED: "WEDM_ERR_UNSUPPORTED",
}


class F64(enum.IntEnum):
    """Rows of the float64 state block (``enum wedm_f64_field``)."""

    WORKPIECE_POS = 0
    WIRE_POS = 1
    WIRE_VEL = 2
    PREV_ACCEL = 3
    DEBRIS_VOLUME = 4
    DEBRIS_DENSITY = 5
    FLOW = 6
    LAST_GAP = 7
    LAST_DENSITY = 8
    WIRE_LAST_FLOW = 9
    VOLTAGE = 10
    CURRENT = 11
    SPARK_Y = 12
    LAST_CRATER = 13
    CAVITY = 14
    TARGET_DELTA = 15
    TARGET_VOLTAGE = 16
    ON_TIME = 17
    OFF_TIME = 18
    TARGET_POS = 19
    UNWIND_VEL = 20
    H_BASE = 21
    H_ZONE = 22
    TMAX = 23
    VOLT_ACC = 24
    VOLT_SUM = 25


F64_COUNT = 26


class I32(enum.IntEnum):
    """Rows of the int32 state block (``enum wedm_i32_field``)."""

    TIME = 0
    SINCE_SERVO = 1
    SINCE_OPEN_V = 2
    SINCE_IGNITION = 3
    SINCE_SPARK_END = 4
    SPARK_DUR = 5
    RANDOM_SHORT_REM = 6
    DEBRIS_SHORT_REM = 7
    TIME_CRITICAL = 8
    CURRENT_MODE = 9
    EPISODE = 10
    KEY_LO = 11
    KEY_HI = 12
    SPARK_COUNT = 13
    TIME_HI = 14


I32_COUNT = 15


class I8(enum.IntEnum):
    """Rows of the int8 state block (``enum wedm_i8_field``)."""

    SPARK_STATE = 0
    IS_SHORT = 1
    WIRE_BROKEN = 2
    TARGET_REACHED = 3
    DONE = 4
    CTRL_STEP = 5
    ERROR = 6
    MODE_CACHED = 7


I8_COUNT = 8


class STAT(enum.IntEnum):
    """Rows of the optional statistics block (``enum wedm_stat_field``)."""

    CRATER_SUM = 0
    CRATER_SUMSQ = 1
    CRATER_MIN = 2
    CRATER_MAX = 3


STAT_COUNT = 4


class GF64(enum.IntEnum):
    """Rows of the per-environment geometry float64 block."""

    HEIGHT = 0
    KERF_BASE = 1
    CAVITY_COEFF = 2
    K_COND = 3
    TUF = 4
    A_SURF = 5
    S_AREA = 6
    JOULE_GEOM = 7


GEOM_F64_COUNT = 8


class GI32(enum.IntEnum):
    """Rows of the per-environment geometry int32 block."""

    N_SEG = 0
    ZONE_START = 1
    AZ_START = 2
    AZ_END = 3
    CONTACT_BOTTOM = 4
    CONTACT_TOP = 5


GEOM_I32_COUNT = 6

REPLAY_SLOTS = 5  # enum wedm_replay_slot: debris roll, random-short roll, ignition roll, spark y [mm], crater volume [um^3]

OBS_DIM = 8
OBS_NAMES = ("gap", "wire_velocity", "voltage", "current", "spark_state", "debris_density", "flow_rate", "tmax")

_d = C.c_double
_i = C.c_int32
_tab = _d * (MAX_MODE + 1)
_itab = _i * (MAX_MODE + 1)


class Params(C.Structure):
    """``struct wedm_params``."""

    _fields_ = [
        ("servo_interval", _i), ("dt_us", _i), ("control_mode", _i), ("per_env_geometry", _i),
        ("initial_gap", _d), ("target_cutting_distance", _d),
        ("n_seg", _i), ("zone_start", _i), ("az_start", _i), ("az_end", _i),
        ("contact_bottom", _i), ("contact_top", _i),
        ("workpiece_height", _d), ("kerf_base", _d), ("cavity_coeff", _d),
        ("k_cond", _d), ("tuf", _d), ("a_surf", _d), ("s_area", _d), ("joule_geom", _d),
        ("segment_len", _d),
        ("spool_T", _d), ("temp_ref", _d), ("rho_elec", _d), ("alpha_rho", _d), ("rho_c", _d),
        ("plasma_efficiency", _d), ("base_convection", _d),
        ("convection_velocity_factor", _d), ("convection_flow_enhancement", _d),
        ("critical_temperature", _d), ("breaking_temperature", _d), ("dielectric_temperature", _d),
        ("base_critical_density", _d), ("gap_coefficient", _d), ("max_critical_density", _d),
        ("hard_short_gap", _d), ("sigmoid_steepness", _d),
        ("debris_short_duration", _i), ("random_short_duration", _i),
        ("random_short_min_gap", _d), ("random_short_max_gap", _d), ("random_short_max_probability", _d),
        ("ignition_a", _d), ("ignition_b", _d), ("ignition_c", _d), ("ln2", _d),
        ("default_target_voltage", _d), ("default_on_time", _d), ("default_off_time", _d),
        ("default_current", _d), ("spark_voltage_factor", _d),
        ("reference_gap", _d), ("debris_obstruction_coeff", _d), ("debris_removal_per_us", _d),
        ("dt_s", _d), ("damping_coeff", _d), ("stiffness_coeff", _d), ("omega_n", _d),
        ("max_acceleration", _d), ("max_jerk_dt", _d), ("max_speed", _d),
        ("mode_current", _tab), ("crater_mean", _tab), ("crater_std", _tab), ("crater_depth", _tab),
        ("crater_valid", _itab),
        ("env_id_offset", C.c_uint32), ("obs_dim", _i),
        ("disable_ignition", _i), ("autoreset", _i), ("reward_mode", _i), ("stencil_mode", _i),
        ("reset_semantics", _i), ("keep_stepping_terminated", _i),
        ("reward_break_penalty", _d),
    ]


class StatePtrs(C.Structure):
    """``struct wedm_state_ptrs``."""

    _fields_ = [
        ("f64", C.c_void_p), ("i32", C.c_void_p), ("i8", C.c_void_p), ("T", C.c_void_p),
        ("obs", C.c_void_p), ("stride", C.c_int64), ("stats", C.c_void_p), ("reward", C.c_void_p),
        ("crater_log", C.c_void_p), ("crater_log_capacity", C.c_int64),
    ]


class GeomPtrs(C.Structure):
    """``struct wedm_geom_ptrs``."""

    _fields_ = [("f64", C.c_void_p), ("i32", C.c_void_p)]


class ActionPtrs(C.Structure):
    """``struct wedm_action_ptrs``."""

    _fields_ = [
        ("servo", C.c_void_p), ("target_voltage", C.c_void_p), ("on_time", C.c_void_p),
        ("off_time", C.c_void_p), ("current_mode", C.c_void_p),
    ]


class TraceDesc(C.Structure):
    """``struct wedm_trace_desc``."""

    _fields_ = [
        ("f64", C.c_void_p), ("i32", C.c_void_p), ("i8", C.c_void_p), ("T", C.c_void_p),
        ("f64_mask", C.c_uint32), ("i32_mask", C.c_uint32), ("i8_mask", C.c_uint32),
        ("env_lo", C.c_int32), ("env_count", C.c_int32), ("every", C.c_int32), ("capacity", C.c_int32),
        ("reserved0", C.c_int32),
    ]
