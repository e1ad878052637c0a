"""CPU oracle for the TREW tandem-repeat scan -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (trew_amd/) never does.  See oracle/trew_oracle.h.
"""
from .pyoracle import (  # noqa: F401
    OracleParams,
    TABLE_NAMES,
    build,
    check_ans_seq,
    code,
    four_to_int,
    int_to_four,
    repeat_check,
    revcomp,
    rot_seq,
    run_long,
    run_pair,
    run_short,
    run_short_mt_timed,
    segment_check,
    segment_stats,
)
from .output_oracle import fold_tables, format_sections, putative_trm  # noqa: F401
