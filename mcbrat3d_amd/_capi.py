"""ctypes binding of include/mcbrat.h (the C ABI of the HIP library).

There is no CPU fallback: if libmcbrat_hip.so is missing or no HIP device is
usable, importing the symbols or creating a context raises."""
import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCBRAT_LIB", os.path.join(_HERE, "libmcbrat_hip.so"))  # MCBRAT_LIB: A/B builds
_lib = None


class McbratError(RuntimeError):
    """A failure status from the integrator (the reference's stateIsFailure(status))."""


class Fate(C.Structure):
    _fields_ = [("fate", C.c_int32), ("ix", C.c_int32), ("iy", C.c_int32), ("iz", C.c_int32),
                ("nScatter", C.c_int32), ("nEvents", C.c_int32), ("weight", C.c_float)]


FATE_DTYPE = np.dtype([("fate", "<i4"), ("ix", "<i4"), ("iy", "<i4"), ("iz", "<i4"),
                       ("nScatter", "<i4"), ("nEvents", "<i4"), ("weight", "<f4")])


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("legs", "crossings", "collisions", "absorbEvents", "topExits",
                                         "surfaceHits", "rouletteKills", "rouletteSurvivals", "walkIterations",
                                         "walkLanes", "eventPhases", "eventLanes", "launchPhases", "surfacePhases", "badPhotons")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


# every symbol include/mcbrat.h declares: (restype, argtypes)
_vp, _i32, _i64, _u64, _f, _d = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double
SYMBOLS = {
    "mcbrat_abi_version": (C.c_int, []),
    "mcbrat_create": (_vp, [C.c_int]),
    "mcbrat_destroy": (None, [_vp]),
    "mcbrat_last_error": (C.c_char_p, [_vp]),
    "mcbrat_set_grid": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mcbrat_set_optics": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _d]),
    "mcbrat_set_inverse_table": (C.c_int, [_vp, _i32, _i32, _i32, _vp]),
    "mcbrat_specify_parameters": (C.c_int, [_vp, _i32, _i32, _f]),
    "mcbrat_set_source_solar": (C.c_int, [_vp, _f, _f]),
    "mcbrat_set_source_emission": (C.c_int, [_vp, _vp, _d]),
    "mcbrat_compute_radiative_transfer": (C.c_int, [_vp, _u64, _u64, _i64, _i32, _vp]),
    "mcbrat_report_results": (C.c_int, [_vp] * 9),
    "mcbrat_moments_length": (_i64, [_vp]),
    "mcbrat_bind_moments": (C.c_int, [_vp, _vp]),
    "mcbrat_reset_moments": (C.c_int, [_vp]),
    "mcbrat_get_moments": (C.c_int, [_vp, _vp]),
    "mcbrat_enable_counters": (C.c_int, [_vp, _i32]),
    "mcbrat_get_counters": (C.c_int, [_vp, _vp]),
    "mcbrat_last_trace_ms": (_f, [_vp]),
    "mcbrat_specify_intensity": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _f, _i32, _i32, _i32, _f]),
    "mcbrat_set_forward_table": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "mcbrat_report_intensity": (C.c_int, [_vp, _vp, _vp]),
    "mcbrat_forward_table_legendre": (C.c_int, [_i32, _vp, _i32, _vp]),
    "mcbrat_forward_table_tabulated": (C.c_int, [_i32, _vp, _vp, _i32, _vp]),
    "mcbrat_hybrid_phase_functions": (C.c_int, [_i32, _i32, _vp, _f, _vp]),
    "mcbrat_get_event_threshold": (C.c_int, [_vp]),
    "mcbrat_set_async": (C.c_int, [_vp, _i32]),
    "mcbrat_synchronize": (C.c_int, [_vp]),
    "mcbrat_stream_wait_done": (C.c_int, [_vp, _vp]),
    "mcbrat_wait_stream": (C.c_int, [_vp, _vp]),
    "mcbrat_chain_after": (C.c_int, [_vp, _vp]),
    "mcbrat_set_tuning": (C.c_int, [_vp] + [_i32] * 8),
    "mcbrat_set_walk_options": (C.c_int, [_vp, _i32, _i32]),
    "mcbrat_set_option": (C.c_int, [_vp, C.c_char_p, _i32]),
    "mcbrat_first_drop": (C.c_char_p, [_vp]),
    "mcbrat_get_walk_mode": (C.c_int, [_vp]),
    "mcbrat_moments_device_pointer": (_vp, [_vp]),
    "mcbrat_frequency_distribution": (C.c_int, [_vp, _u64, _u64, _i32, _vp, _i64, _vp]),
    "mcbrat_set_surface_description": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "mcbrat_trace_fates": (C.c_int, [_vp, _u64, _u64, _i64, _vp]),
    "mcbrat_inverse_table_legendre": (C.c_int, [_i32, _vp, _i32, _vp]),
    "mcbrat_inverse_table_tabulated": (C.c_int, [_i32, _vp, _vp, _i32, _vp]),
    "mcbrat_block_decomposition": (C.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mcbrat_flight_tables": (C.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mcbrat_emission_weighting": (C.c_int, [_i32] * 4 + [_vp] * 7 + [_d] * 4 + [_vp] * 3),
}


ABI_VERSION = 3  # MCBRAT_ABI_VERSION of include/mcbrat.h this binding was written against (badPhotons since 2, mcbrat_set_option since 3)


def hip_runtimes():
    """Distinct HIP runtime libraries (libamdhip64) mapped into this process, as paths.

    The product links ROCm's runtime by SONAME (libamdhip64.so.7, RUNPATH /opt/rocm); PyTorch ships a copy of its own and asks
    for it by another name (libamdhip64.so, RPATH $ORIGIN).  Loaded after PyTorch the product binds to PyTorch's copy (same
    SONAME, already loaded): ONE runtime.  Loaded before it, PyTorch maps its copy beside ROCm's: TWO runtimes in one
    process, and a stream or device pointer of one handed to the other (bench.py, driver.run: RCCL on the moment buffer) is
    a handle of a different library.  INTEGRATION.md, "One HIP runtime"."""
    found = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                m = re.search(r"(/\S*libamdhip64[^\s]*)", line)
                if m:
                    found.add(os.path.realpath(m.group(1)))
    except OSError:
        pass
    return sorted(found)


def assert_single_hip_runtime():
    """Raises unless exactly one HIP runtime is mapped; returns its path.  Called when the library is loaded, and again by
    everything that passes PyTorch objects (streams, tensors) to it."""
    rts = hip_runtimes()
    if len(rts) > 1:
        raise McbratError("two HIP runtimes in one process (%s): import torch BEFORE the first call into mcbrat3d_amd, so that "
                          "libmcbrat_hip.so binds to the runtime PyTorch brings (INTEGRATION.md, 'One HIP runtime')" % ", ".join(rts))
    return rts[0] if rts else None


def lib():
    """The loaded native library; raises if it has not been built, if its ABI is not the one this binding expects, or if
    loading it has left two HIP runtimes in the process."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise McbratError("%s is missing: build it with `python -m mcbrat3d_amd.build` "
                              "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        old = bool(os.environ.get("MCBRAT_LIB_OLD"))  # A/B runs against a library built from an earlier round's sources: MCBRAT_LIB=... MCBRAT_LIB_OLD=1
        for name, (res, args) in SYMBOLS.items():
            if old and not hasattr(L, name):
                continue
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        # (a stale or foreign build: the size of mcbrat_counters and the meaning of its last field depend on the version --
        # an ABI-1 library would leave badPhotons unwritten and every "nothing was dropped" check would pass vacuously)
        got = L.mcbrat_abi_version()
        if got != ABI_VERSION and not old:
            raise McbratError("%s has ABI version %d, this binding expects %d: rebuild it (`python -m mcbrat3d_amd.build --force`), "
                              "or set MCBRAT_LIB_OLD=1 for an A/B run against an older library" % (LIB_PATH, got, ABI_VERSION))
        L._mcbrat_abi = got
        _lib = L
        assert_single_hip_runtime()
    return _lib


def lib_abi():
    return lib()._mcbrat_abi


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def check(ctx, rc):
    if rc != 0:
        msg = lib().mcbrat_last_error(ctx)
        raise McbratError(msg.decode() if msg else "mcbrat: failure %d" % rc)
