"""No byte outside the caller's buffers: every device buffer handed to the C-ABI lies against unmapped addresses
(tests/guard_memory.py: HIP virtual-memory calls, nothing mapped before or after), so that a read or a write one byte out of
bounds is a memory access fault instead of a silent success.  The reference has no such test - safe Rust indexes slices
(image_buffer.rs:40-98) and its fuzz targets ask "does not panic"; a C-ABI over raw device pointers has to prove it.

The scenarios run in child processes (tests/guard_runner.py): a fault ends the process, the parent reads its exit status and
its last line.  `selfcheck` first shows that the guard does fault on this box when a kernel is told to read past its buffer.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(what, timeout=900):
    env = dict(os.environ, PYTHONUNBUFFERED="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "guard_runner.py"), what], capture_output=True, text=True, timeout=timeout, env=env)
    return p.returncode, p.stdout.strip().splitlines(), p.stderr


def test_the_guard_faults_when_a_kernel_reads_past_its_buffer():
    rc, lines, err = run("selfcheck")
    assert rc != 0 and "survived" not in lines, (rc, lines[-3:], err[-400:])
    assert any(l.startswith("granularity") for l in lines), (lines, err[-400:])       # (it got as far as the encode)


@pytest.mark.parametrize("what,at_least", [("pixels", 99), ("planes", 81), ("raw", 10)])
def test_nothing_is_read_or_written_outside_the_callers_buffers(what, at_least):
    rc, lines, err = run(what)
    assert rc == 0, (rc, lines[-2:], err[-1500:])
    assert lines[-1].startswith("done") and int(lines[-1].split()[1]) >= at_least, lines[-2:]
