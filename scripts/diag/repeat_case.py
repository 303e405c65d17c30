#!/usr/bin/env python
"""Run one pytest selection N times in THIS process and report every failure's assertion line (flakiness of a tolerance, not of
the device).   usage: python scripts/diag/repeat_case.py N <pytest args...>"""
import sys, io, contextlib, re
import pytest
n = int(sys.argv[1])
fails = 0
for i in range(n):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = pytest.main(['-q', '-x', '-p', 'no:cacheprovider'] + sys.argv[2:])
    out = buf.getvalue()
    if rc != 0:
        fails += 1
        print(f'--- run {i}: rc {rc}')
        for line in out.splitlines():
            if line.startswith('E ') or 'FAILED' in line:
                print(line[:240])
print(f'{fails} of {n} runs failed')
