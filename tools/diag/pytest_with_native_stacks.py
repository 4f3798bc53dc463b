#!/usr/bin/env python3
"""pytest with tools/diag/stackprof.c's crash handler installed (SIGSEGV / SIGBUS / SIGABRT print the faulting thread's NATIVE stack):
    python3 tools/diag/pytest_with_native_stacks.py <pytest arguments>"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = "/tmp/libstackprof.so"
if not os.path.exists(so):
    subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "diag", "stackprof.c"), "-ldl"])
ctypes.CDLL(so).stackprof_install_crash_handler()
import pytest  # noqa: E402
sys.exit(pytest.main(["-p", "no:faulthandler", "-s"] + sys.argv[1:]))      # (-s: the handler writes to the real stderr)
