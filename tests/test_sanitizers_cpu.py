"""Host sanitizer pass (SURVEY section 5 "race detection / sanitizers": the reference has -Wall -Werror only,
make/Makefile.linux-x86_64:31).  The host sources of the library -- C API (capi.cpp), driver (context.cpp), tokenizer,
DPM solver and the graph engine with its .sdodw loader -- are rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer
(`make asan`, host code only) and the host test files run against that build in a subprocess with the runtime preloaded.
Any heap overflow / use-after-free / UB in those paths aborts the subprocess.  CPU box only: GPU ASan is unavailable."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'stable-diffusion-on-device_amd')


def _runtime():
    hits = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
    return hits[0] if hits else None


@pytest.mark.skipif(_runtime() is None, reason='clang AddressSanitizer runtime not found')
def test_host_tests_pass_under_asan_and_ubsan():
    subprocess.check_call(['make', '-C', PKG, '-j8', 'all', 'asan'], stdout=subprocess.DEVNULL)
    env = dict(os.environ, SDOD_LIBSDOD='libsdod_asan.so', LD_PRELOAD=_runtime(),
               ASAN_OPTIONS='detect_leaks=0:halt_on_error=1:abort_on_error=1',      # CPython itself "leaks" at exit
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    probe = subprocess.run([sys.executable, '-c', 'import sys; sys.path.insert(0, %r); from sdod.amd import _lib; '
                            'l = _lib.load("libsdod.so"); print(l._name); print(open("/proc/self/maps").read().count("libclang_rt.asan"))' % PKG],
                           env=env, capture_output=True, text=True, timeout=120)
    assert probe.returncode == 0, probe.stderr[-2000:]
    name, mapped = probe.stdout.split()[:2]
    assert name.endswith('libsdod_asan.so') and int(mapped) > 0, probe.stdout        # the instrumented build is what ran
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_host_cabi.py', 'tests/test_formats_cpu.py', '-x', '-q', '-m', 'not gpu',
                        '-p', 'no:cacheprovider'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and ' passed' in r.stdout and 'AddressSanitizer' not in tail and 'runtime error' not in tail, tail
