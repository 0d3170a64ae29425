"""Key-point configuration records for derivative interpolation.

Drop-in for the two record types of the reference's utils_derivs_interpolation module
(/root/reference/utils_derivs_interpolation.py:3-14): same class names, same field names,
same positional construction order — callers such as acrobot.py:115 do
``derivs_interpolation(keypoint_method, minN, maxN, jerk_threshold, iterative_error_threshold)``.

keypoint_method is one of 'setInterval' | 'adaptiveJerk' | 'iterativeError'
(/root/reference/ilqr.py:396-400); the solver maps it onto MI_KP_* of include/mi_ilqr.h.
"""
from dataclasses import dataclass

KEYPOINT_METHODS = ("setInterval", "adaptiveJerk", "iterativeError")


@dataclass
class derivs_interpolation:
    keypoint_method: str               # which key-point selection rule (KEYPOINT_METHODS)
    minN: int                          # minimum spacing between key-points
    maxN: int                          # maximum spacing (adaptiveJerk only)
    jerk_threshold: float              # adaptiveJerk trigger
    iterative_error_threshold: float   # iterativeError trigger


@dataclass
class index_tuple:
    start_index: int
    end_index: int
