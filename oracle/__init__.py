"""ORACLE -- test infrastructure only (see oracle/dana_oracle.c header). Never imported by the
product package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it."""
