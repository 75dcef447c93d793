"""CPU oracle for the dropEst Estimation hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under dropest_amd/ imports it (tests/test_cabi_cpu.py enforces that).
"""
from .binding import Oracle, OracleConfig, build_oracle, oracle_lib  # noqa: F401
