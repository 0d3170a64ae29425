"""Key-point configuration records for derivative interpolation.

Drop-in for the two record types of the reference's utils_derivs_interpolation module
(/root/reference/utils_derivs_interpolation.py:3-14): same class names, same field names,
same positional construction order — callers such as acrobot.py:115 do
``derivs_interpolation(keypoint_method, minN, maxN, jerk_threshold, iterative_error_threshold)``.

keypoint_method is one of 'setInterval' | 'adaptiveJerk' | 'iterativeError'
(/root/reference/ilqr.py:396-400); the solver maps it onto MI_KP_* of include/mi_ilqr.h.
"""
import dataclasses as _dc

_KP_FIELDS = (
    ("keypoint_method", str),            # which key-point selection rule
    ("minN", int),                       # minimum spacing between key-points
    ("maxN", int),                       # maximum spacing (adaptiveJerk only)
    ("jerk_threshold", float),           # adaptiveJerk trigger
    ("iterative_error_threshold", float),  # iterativeError trigger
)
_SPAN_FIELDS = (("start_index", int), ("end_index", int))

derivs_interpolation = _dc.make_dataclass("derivs_interpolation", _KP_FIELDS)
index_tuple = _dc.make_dataclass("index_tuple", _SPAN_FIELDS)
for _cls in (derivs_interpolation, index_tuple):
    _cls.__module__ = __name__

KEYPOINT_METHODS = ("setInterval", "adaptiveJerk", "iterativeError")
