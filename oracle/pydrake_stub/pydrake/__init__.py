"""Stub of the two free functions /root/reference/ilqr.py takes from pydrake
(SURVEY.md F2/F4).  Used ONLY inside this container by oracle/gen_golden.py to
import the unmodified reference; never shipped, never on the product path."""
