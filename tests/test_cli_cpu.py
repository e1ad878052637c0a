"""CLI surface of the `trew` binary that needs no GPU: argument validation, messages, exit codes
(reference: trew.cpp:143-376)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TREW = os.path.join(ROOT, "trew_amd", "bin", "trew")
FQ = os.path.join(ROOT, "tests", "golden", "test.fastq")


def run(*args):
    return subprocess.run([TREW, *args], capture_output=True, text=True, timeout=60)


@pytest.mark.skipif(not os.path.exists(TREW), reason="binary not built")
@pytest.mark.parametrize(
    "args,msg",
    [
        (["short", "6", "5", FQ], "MIN_MER must not be greater than MAX_MER."),
        (["short", "2", "5", FQ], "MIN_MER must be greater than or equal to 3."),
        (["short", "5", "65", FQ], "MAX_MER must be less than or equal to 64."),
        (["short", "5", "32", FQ, "-m", "16"], "TABLE_MAX_MER must be less than or equal to 15."),
        (["long", "5", "32", FQ, "-s", "63"], "SLICE_LENGTH must be greater than or equal to twice of MAX_MER."),
        (["short", "5", "32", FQ, "-q", "3"], "QUEUE_SIZE must be -1 (unlimited) or greater than or equal to 4."),
        (["short", "5", "32", FQ, "-t", "0"], "number of threads must be positive."),
        (["short", "5", "32", FQ, "-L", "0"], "Baseline must be in range 0 to 1."),
        (["short", "5", "32", FQ, "-L", "0.9", "-H", "0.8"], "Low baseline must be smaller than high baseline."),
        (["short", "5", "32", FQ, "-t", "1"], "You must use at least two threads."),
        (["short", "5", "32"], "SHORT_FASTQ is required in single-end mode."),
        (["short", "5", "32", FQ, "--fq1", FQ], "--fq1 and --fq2 should not be used in single-end mode."),
        (["short", "5", "32", "--paired_end", "--fq1", FQ], "--fq1 and --fq2 are required in paired-end mode."),
        (["short", "5", "32", "--paired_end", "--fq1", FQ, FQ, "--fq2", FQ], "--fq1 and --fq2 must have the same number of files."),
        (["short", "5", "32", "/nonexistent.fastq"], "/nonexistent.fastq : file not found"),
        # options of this build (stderr only; stdout stays the reference's CSV)
        (["short", "5", "32", FQ, "--table_log2_slots", "5"], "table_log2_slots must be in range 12 to 30."),
        (["short", "5", "32", FQ, "--batch_mib", "0"], "Usage: short"),
        (["short", "5", "32", FQ, "--devices", "0,x"], "Usage: short"),
    ],
)
def test_argument_errors(args, msg):
    r = run(*args)
    assert r.returncode == 1
    assert msg in r.stderr
    assert r.stdout == ""


@pytest.mark.skipif(not os.path.exists(TREW), reason="binary not built")
def test_usage_and_version():
    assert run().returncode == 1
    assert run("bogus").returncode == 1
    r = run("--version")
    assert r.returncode == 0 and r.stdout.strip() == "0.5.0"
