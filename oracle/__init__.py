"""CPU restatement of the reference's algorithm for the rollout path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; nothing under kinpoly_amd/ does
(tests/test_host_cpu.py checks it).  np_oracle.py / np_spd.py (numpy fp64, pinned to fixtures generated from the reference),
kp_oracle.c + kpo.py (C fp64 physics; the MuJoCo-side arithmetic is UNPINNED).
"""
