"""CPU oracle for the redux hot path -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, and from
nowhere else.  See oracle/redux_oracle.h for what the oracle restates and what pins it.
"""
