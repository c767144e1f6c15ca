"""CPU: the host-side native code (csrc/host.cpp: sedt_hungarian_batch, the batched assignment solver behind the reference-style host
matching, sedt/matcher.py:95) built with -fsanitize=address,undefined and driven over random and degenerate problems - sizes with
more targets than queries, empty clips, ties, a non-finite cost (must be refused, not read out of bounds).  The sanitized build is
a stand-alone executable (g++; no HIP runtime needed: host.cpp has no device code), so the interpreter is not run under ASan.
SURVEY.md section 5 lists this run; GPU AddressSanitizer is not available on the pool."""
import os
import shutil
import subprocess

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from conftest import ROOT

DRIVER = r'''
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <vector>
#include <stdint.h>
namespace sedt { void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); } }
extern "C" int sedt_hungarian_batch(const float* cost, int nlayers, int nclips, int Q, int Nt, const int32_t* col_off,
                                    const int32_t* ncols, int32_t* assign);
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  int32_t hdr[4];
  if (!f || fread(hdr, 4, 4, f) != 4) return 3;
  const int L = hdr[0], B = hdr[1], Q = hdr[2], Nt = hdr[3];
  std::vector<int32_t> off(B), nc(B);
  if (fread(off.data(), 4, B, f) != (size_t)B || fread(nc.data(), 4, B, f) != (size_t)B) return 3;
  std::vector<float> cost((size_t)L * B * Q * Nt);               // exactly sized: an out-of-bounds read is an ASan report
  if (!cost.empty() && fread(cost.data(), 4, cost.size(), f) != cost.size()) return 3;
  fclose(f);
  std::vector<int32_t> assign((size_t)L * B * Q, -7);
  const int rc = sedt_hungarian_batch(cost.data(), L, B, Q, Nt, off.data(), nc.data(), assign.data());
  printf("%d", rc);
  for (size_t i = 0; i < assign.size(); ++i) printf(" %d", assign[i]);
  printf("\n");
  return 0;
}
'''


@pytest.fixture(scope='module')
def asan_driver(tmp_path_factory):
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('g++ not found')
    d = tmp_path_factory.mktemp('asan')
    src = d / 'driver.cpp'
    src.write_text(DRIVER)
    exe = d / 'hungarian_asan'
    host = os.path.join(ROOT, 'sound_event_detection_transformer_amd', 'csrc', 'host.cpp')
    r = subprocess.run([gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-fno-omit-frame-pointer',
                        str(src), host, '-o', str(exe)], capture_output=True, text=True)
    if r.returncode != 0 and 'sanitize' in r.stderr and 'cannot find' in r.stderr:
        pytest.skip('libasan / libubsan not installed: ' + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    return str(exe), str(d)


def _run(exe, d, cost, sizes, L, B, Q):
    Nt = int(sum(sizes))
    off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32) if B else np.zeros(0, np.int32)
    path = os.path.join(d, 'case.bin')
    with open(path, 'wb') as f:
        np.asarray([L, B, Q, Nt], np.int32).tofile(f)
        off.tofile(f)
        np.asarray(sizes, np.int32).tofile(f)
        np.ascontiguousarray(cost, np.float32).tofile(f)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([exe, path], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]
    vals = [int(v) for v in r.stdout.split()]
    return vals[0], np.asarray(vals[1:], np.int32).reshape(L, B, Q), r.stderr


@pytest.mark.parametrize('seed,L,B,Q,nmax', [(0, 3, 8, 10, 9), (1, 1, 5, 4, 12), (2, 2, 6, 21, 20), (3, 3, 1, 1, 1)])
def test_hungarian_batch_under_asan_ubsan_matches_scipy(asan_driver, seed, L, B, Q, nmax):
    exe, d = asan_driver
    rng = np.random.RandomState(seed)
    sizes = rng.randint(0, nmax + 1, size=B)
    sizes[0] = nmax                                         # (the largest case incl. more targets than queries when nmax > Q)
    if B > 1:
        sizes[1] = 0                                        # an empty clip
    Nt = int(sizes.sum())
    cost = rng.randn(L, B, Q, Nt).astype(np.float32)
    if seed == 2:
        cost = np.round(cost * 2) / 2                       # many exact ties: any optimal assignment has the same total cost
    rc, assign, _ = _run(exe, d, cost, sizes, L, B, Q)
    assert rc == 0
    off = np.concatenate([[0], np.cumsum(sizes)])
    for l in range(L):
        for b in range(B):
            n = int(sizes[b])
            a = assign[l, b]
            if n == 0:
                assert (a == -1).all()
                continue
            c = cost[l, b][:, off[b]:off[b] + n].astype(np.float64)
            qi, ti = linear_sum_assignment(c)
            matched = np.nonzero(a >= 0)[0]
            assert len(matched) == min(Q, n) and len(set(a[matched].tolist())) == len(matched)
            assert abs(c[matched, a[matched]].sum() - c[qi, ti].sum()) <= 1e-9 * max(1.0, abs(c[qi, ti].sum()))
            if seed != 2:
                np.testing.assert_array_equal(matched, qi)
                np.testing.assert_array_equal(a[matched], ti)


def test_hungarian_batch_refuses_nonfinite_costs_without_touching_memory(asan_driver):
    exe, d = asan_driver
    cost = np.zeros((1, 2, 3, 4), np.float32)
    cost[0, 1, 2, 3] = np.inf
    rc, assign, err = _run(exe, d, cost, [2, 2], 1, 2, 3)
    assert rc == 2 and 'non-finite' in err
