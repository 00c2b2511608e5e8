"""CPU oracle for the DirectVoxGO ray-marching hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package;
directvoxgo_amd never does.  See oracle/dvgo_oracle.c for scope and pinning status.
"""
