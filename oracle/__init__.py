"""ORACLE -- test infrastructure only.  CPU restatement of the reference's algorithm for the hot
path; imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg and by nothing
else.  See oracle/minigtn.py for the pinning status."""
