"""MI355X path-tracing backend for the raytracing-in-one-weekend scene model.

host     — config / camera / sampler / scene builders / flatten (librtiow_host.so, C++)
backend  — the HIP render path behind include/rt_api.h (librtiow_hip.so)
"""
from . import _ffi, host, backend, configs, sharding  # noqa: F401
from ._ffi import RtError  # noqa: F401
