"""Test-only checker for the content-addressing path (see oracle/vit_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (vit_amd/) never does.
"""
