"""CPU: the HOST side of every C-ABI entry point under AddressSanitizer + UBSan (sanitizer build of the boundary).

`make -C depthinspace_amd/csrc asan` compiles the host pass of the nine .hip files with -fsanitize=address,undefined (seconds; the
device images are empty stubs, nothing can launch).  A driver generated from include/dis_hip.h calls every entry point
  1. with every pointer NULL and sizes 1            -> must refuse (non-zero status) without touching anything,
  2. with valid host pointers and sizes 0 / -1      -> must refuse a bad shape before it dereferences or launches,
  3. (size queries) with every extent 64 and 1024    -> the workspace arithmetic must not overflow (five extents: 2^50),
and exits non-zero on the first sanitizer report.  No GPU is needed: argument checks, launch geometry and workspace sizes are host code.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'depthinspace_amd', 'csrc')
LLVM = os.environ.get('LLVM_BIN', '/opt/rocm/lib/llvm/bin')


def _protos():
    src = open(os.path.join(ROOT, 'include', 'dis_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = []
    for ret, name, args in re.findall(r'\b(int|long)\s+(dis_[a-z0-9_]+)\s*\(([^)]*)\)', src):
        args = args.strip()
        al = [] if args in ('', 'void') else [a.strip() for a in args.split(',')]
        kinds = []
        for a in al:
            if '*' in a:
                kinds.append('s' if re.search(r'\bstream$', a) else 'p')
            elif a.startswith('float'):
                kinds.append('f')
            elif a.startswith('double'):
                kinds.append('d')
            elif a.startswith('long'):
                kinds.append('l')
            else:
                kinds.append('i')
        out.append((ret, name, kinds))
    return out


def _call(name, kinds, ptr, ival):
    vals = {'p': ptr, 's': 'nullptr', 'f': '1.0f', 'd': '1.0', 'l': f'(long)({ival})', 'i': f'(int)({ival})'}
    return f'{name}({", ".join(vals[k] for k in kinds)})'


def test_entry_points_under_host_sanitizers(tmp_path):
    if not (os.path.exists('/opt/rocm/bin/hipcc') and os.path.exists(os.path.join(LLVM, 'clang++'))):
        pytest.skip('no hipcc / clang++')
    r = subprocess.run(['make', '-C', CSRC, '-j4', 'asan'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = os.path.join(ROOT, 'depthinspace_amd', 'libdis_hip_asan.so')
    assert os.path.exists(lib)
    protos = _protos()
    assert len(protos) >= 100
    lines = ['#include <cstdio>', '#include <cstring>', f'#include "{os.path.join(ROOT, "include", "dis_hip.h")}"',
             'alignas(64) static char buf[1 << 16];',
             'struct P { template <class T> operator T*() const { return (T*)buf; } };   // any typed pointer into the host buffer',
             'int main() {', '  long bad = 0, n = 0;', '  std::memset(buf, 0, sizeof buf);']
    nptr = 0
    for ret, name, kinds in protos:
        has_ptr = 'p' in kinds
        is_query = ret == 'long'
        # 1. NULL pointers
        lines.append(f'  {{ long rc = (long){_call(name, kinds, "nullptr", 1)}; ++n;')
        if has_ptr and not is_query:
            nptr += 1
            lines.append(f'    if (rc == 0) {{ std::printf("ACCEPTED-NULL {name}\\n"); ++bad; }}')
        lines.append('  }')
        # 2. host pointers, degenerate sizes
        if any(k in kinds for k in 'il'):
            for iv in (0, -1):
                lines.append(f'  {{ long rc = (long){_call(name, kinds, "P()", iv)}; ++n; (void)rc; }}')
        # 3. size queries: large but sane extents
        if is_query and kinds:
            for iv in (64, 1 << 10):
                lines.append(f'  {{ long rc = (long){_call(name, kinds, "nullptr", iv)}; ++n; (void)rc; }}')
    lines += ['  std::printf("calls %ld refused-null-missing %ld\\n", n, bad);', '  return bad ? 3 : 0;', '}']
    src = tmp_path / 'drv.cpp'
    text = '\n'.join(lines)
    src.write_text(text)
    exe = tmp_path / 'drv'
    r = subprocess.run([os.path.join(LLVM, 'clang++'), '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
                        '-o', str(exe), str(src), f'-L{os.path.dirname(lib)}', '-ldis_hip_asan', f'-Wl,-rpath,{os.path.dirname(lib)}',
                        '-L/opt/rocm/lib', '-lamdhip64', '-Wl,-rpath,/opt/rocm/lib'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=23', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    tail = (r.stdout[-1500:] + '\n' + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, tail
    m = re.search(r'calls (\d+) refused-null-missing (\d+)', r.stdout)
    assert m and int(m.group(1)) >= 3 * len(protos) // 2 and int(m.group(2)) == 0, tail
    assert nptr >= 80
