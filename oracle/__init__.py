"""CPU oracle for the zignal image hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (zignal_amd, libzignal_hip.so) never does.
"""
