"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see mp2p_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from .oracle import *  # noqa: F401,F403
