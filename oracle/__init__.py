"""TEST INFRASTRUCTURE ONLY.

CPU oracle for the NTPoly hot path: `ntpoly_oracle.c` is a plain-C restatement of the
reference algorithm, `_ref/` (git-ignored, built by build_ref.py where /root/reference
exists) is the real reference.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (ntpoly_amd/) never does.
"""
