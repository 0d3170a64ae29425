"""Workload definitions are shared with the product (pure-NumPy data, no solver
code): re-exported here so oracle scripts read ``oracle.problems``."""
from drake_ddp_amd.workloads import *  # noqa: F401,F403
