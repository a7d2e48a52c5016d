"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see oracle/bx_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package.
The product package (boundless_amd/) must never import it.
"""
