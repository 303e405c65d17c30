"""CPU: the C-ABI library loads and exports every symbol include/dis_hip.h declares (no compute calls)."""
import os
import re
import pytest


def header_symbols(root):
    src = open(os.path.join(root, 'include', 'dis_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|long)\s+(dis_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    from depthinspace_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        g.build()
    syms = header_symbols(g.ROOT)
    assert len(syms) >= 40
    assert set(syms) == set(lib.SIGS.keys()), set(syms) ^ set(lib.SIGS.keys())
    L = lib.load()
    for s in syms:
        assert hasattr(L, s), s
    assert lib.check_all_symbols() == len(syms)
    assert lib.fn('dis_abi_version')() >= 1


def test_argument_counts_match_header():
    import __graft_entry__ as g
    from depthinspace_amd import lib
    src = open(os.path.join(g.ROOT, 'include', 'dis_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    for name, args in re.findall(r'\b(?:int|long)\s+(dis_[a-z0-9_]+)\s*\(([^)]*)\)', src):
        args = args.strip()
        n = 0 if args in ('', 'void') else len(args.split(','))
        assert n == len(lib.SIGS[name]), (name, n, len(lib.SIGS[name]))
        for decl, code in zip([] if n == 0 else args.split(','), lib.SIGS[name]):
            decl = decl.strip()
            kind = 'p' if '*' in decl else ('f' if decl.startswith('float') else ('d' if decl.startswith('double') else ('l' if decl.startswith('long') else 'i')))
            assert kind == code, (name, decl, code)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from depthinspace_amd import lib
    monkeypatch.setattr(lib, '_lib', None)
    monkeypatch.setattr(lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError):
        lib.load()


def test_cpu_tensors_are_rejected():
    import torch
    from depthinspace_amd import ops
    with pytest.raises(RuntimeError):
        ops.lcn(torch.zeros(1, 1, 16, 16))
