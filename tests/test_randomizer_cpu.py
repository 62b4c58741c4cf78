"""CPU: the shuffle order of RandomizerMask::Generate is std::random_shuffle seeded by srand
(nnet-randomizer.cc:33-44).  The library writes that algorithm out (it left the standard in
C++17); this test pins it, bit for bit, to the C++ standard library that the reference itself
would be linked against, by compiling a five-line program that calls std::random_shuffle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROG = r"""
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv) {
  int seed = atoi(argv[1]), n = atoi(argv[2]), rounds = atoi(argv[3]);
  srand(seed);
  for (int r = 0; r < rounds; r++) {
    std::vector<int> v(n);
    for (int i = 0; i < n; i++) v[i] = i;
    std::random_shuffle(v.begin(), v.end());
    for (int i = 0; i < n; i++) printf("%d ", v[i]);
    printf("\n");
  }
}
"""


@pytest.fixture(scope="module")
def stdlib_shuffle(tmp_path_factory):
    d = tmp_path_factory.mktemp("shuf")
    src = d / "shuf.cc"
    src.write_text(PROG)
    exe = d / "shuf"
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-w", str(src), "-o", str(exe)])

    def run(seed, n, rounds):
        out = subprocess.check_output([str(exe), str(seed), str(n), str(rounds)]).decode().strip().split("\n")
        return [np.array(l.split(), np.int32) for l in out]
    return run


def lib_mask():
    lib = C.CDLL(os.path.join(ROOT, "kaldi-aslp_amd", "libaslp_hip.so"))
    fn = lib.aslp_randomizer_mask_generate
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")]

    def gen(seed, n):
        out = np.empty(n, np.int32)
        assert fn(seed, n, out) == 0
        return out
    return gen


@pytest.mark.parametrize("seed,n", [(777, 5), (777, 1111), (1, 32768), (12345, 2)])
def test_mask_order_matches_std_random_shuffle(stdlib_shuffle, seed, n):
    gen = lib_mask()
    want = stdlib_shuffle(seed, n, 3)
    got = [gen(seed, n), gen(-1, n), gen(-1, n)]  # consecutive masks continue the same rand() stream
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
