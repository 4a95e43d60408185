"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (longsom_amd/) never does.  See oracle/README.md for what each file restates.
"""
