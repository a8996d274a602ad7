"""CPU parity oracle -- TEST INFRASTRUCTURE ONLY (see oracle/rt_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product path (raytrace-miniapp_amd) never does."""
