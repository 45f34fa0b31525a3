"""CPU oracle for the reforge render-graph hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package, and only as the checker.  The product
package (reforge_amd) never imports it.

PARITY UNPINNED for the authored nodes: see oracle/rf_oracle.h.
"""
