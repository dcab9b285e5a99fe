"""CPU oracle (TEST INFRASTRUCTURE ONLY -- never imported by inferflow_amd/).

ctypes bindings over oracle/libifa_oracle.so (plain-C restatement of the
reference's quantized decode path, see ifa_oracle.h) and, when present,
oracle/_ref/libifa_ref_quant.so (the reference's own block codecs compiled
from /root/reference; see ref_quant_wrap.cc).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.
"""
from .oracle import *  # noqa: F401,F403
