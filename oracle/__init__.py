"""CPU oracle for the set-abstraction hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package. pointnet2_amd (the product) never does.
"""
from .oracle import *  # noqa: F401,F403
