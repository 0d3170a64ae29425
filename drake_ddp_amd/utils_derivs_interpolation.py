"""Key-point configuration types — same names, field order and positional
construction as /root/reference/utils_derivs_interpolation.py:3-14 so callers
such as acrobot.py:115 (`derivs_interpolation(keypoint_method, minN, maxN,
jerk_threshold, iterative_error_threshold)`) work unchanged."""
from dataclasses import dataclass


@dataclass
class derivs_interpolation:
    keypoint_method: str
    minN: int
    maxN: int
    jerk_threshold: float
    iterative_error_threshold: float


@dataclass
class index_tuple:
    start_index: int
    end_index: int
