"""ASan + UBSan build and run of the CPU side: the oracle sources together with a driver that walks their API on edge-case inputs
(tests/cpp/oracle_sanitize.cpp), and the C++ host header program (tests/cpp/host_mirror.cpp, no-device path) -- SURVEY section 5 asks for
sanitizer runs of the host code; GPU sanitizers are not available on the pool, so this covers what runs on the CPU."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, 'tests', 'cpp', 'build')
SAN = ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer', '-g', '-O1']
ENV = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0:exitcode=23', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')


def test_oracle_under_address_and_ub_sanitizers():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, 'oracle_sanitize')
    srcs = [f for f in sorted(glob.glob(os.path.join(ROOT, 'oracle', '*.cpp'))) if not f.endswith('ref_shim.cpp')]
    cmd = ['g++', '-std=c++17', '-ffp-contract=off', '-Wall', '-Wno-unused-variable'] + SAN + [os.path.join(ROOT, 'tests', 'cpp', 'oracle_sanitize.cpp')] + srcs + ['-o', exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=ENV)
    assert r.returncode == 0 and 'sanitize run ok' in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-3000:]


def test_host_header_program_under_sanitizers(pkg):
    """include/dvbs2gpu_host.hpp compiled with the sanitizers; on a box without a GPU the program must leave through the
    'no CPU fallback' exception without a report (construction, error path and destruction of the mirror classes)"""
    import torch
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, 'host_mirror_san')
    pkg_dir = os.path.join(ROOT, 'sdrpp-dvbs-demodulator_amd')
    cmd = ['g++', '-std=c++17', '-Wall', '-Wextra'] + SAN + ['-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'cpp', 'host_mirror.cpp'),
                                                             '-o', exe, '-L' + pkg_dir, '-ldvbs2gpu', '-Wl,-rpath,' + pkg_dir, '-pthread']
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    if torch.cuda.is_available():
        pytest.skip('GPU present: the library would initialise HIP under ASan (not supported on the pool)')
    tmp = os.path.join(BUILD, 'san_in.bin')
    open(tmp, 'wb').write(b'\0' * 4096)
    for args in (['s2', tmp, tmp + '.o', '14', '1', '0', '1000'], ['bbts', tmp, tmp + '.o', '14232', '4'], ['dvbs', tmp, tmp + '.o', '1000']):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=dict(ENV, ASAN_OPTIONS=ENV['ASAN_OPTIONS'] + ':detect_leaks=0'))
        assert r.returncode == 3 and 'no CPU fallback' in r.stderr, (r.returncode, r.stderr[-2000:])
        assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-3000:]
