"""ORACLE — test infrastructure only.

CPU restatement of the iLQR hot path of vincekurtz/drake_ddp (reference
/root/reference/ilqr.py) plus the harness that pins it against the reference.
Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may
import anything from here, and only as the checker / reported baseline — never
as the thing measured or shipped.  drake_ddp_amd/ must not import this package.
"""
