"""CPU oracle for the RWKV-7 decode hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package,
and only as the checker / timed CPU baseline.  chirrup_amd (the product) never does.
"""
