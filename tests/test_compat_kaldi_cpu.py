"""Seam B4 as a source-level drop-in: the REFERENCE's own tool sources compile UNCHANGED against include/aslp_compat_kaldi.h --
`namespace kaldi::aslp_nnet`, `CuMatrix<BaseFloat>`, `KALDI_LOG`, `trn_opts.Register(&po)`, the reference's header paths through the
forwarding headers of include/kaldi_compat/.  Development container only: the reference tree does not travel, so the test skips where
/root/reference is absent (the linked binaries, kaldi-aslp_amd/bin_ref/, do travel and run in tests/test_tools_gpu.py)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/aslp-nnetbin"
HIPCC = "/opt/rocm/bin/hipcc"
# every main() of the reference's aslp-nnetbin/ (23 of them; the Makefile's `refmains` target links them all into kaldi-aslp_amd/bin_ref/)
MAINS = sorted(f[:-3] for f in os.listdir(REF) if f.endswith(".cc")) if os.path.isdir(REF) else []


def test_compat_layer_has_a_forwarding_header_for_every_include_of_the_served_mains():
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present (GPU box)")
    assert len(MAINS) == 23
    assert len(PAR_MAINS) == 5
    for m in MAINS + PAR_MAINS:
        for inc in re.findall(r'#include\s+"([^"]+)"', open(os.path.join(REF, m + ".cc")).read()):
            path = os.path.join(ROOT, "include", "kaldi_compat", inc)
            assert os.path.exists(path), "%s includes %s: no forwarding header" % (m, inc)
            assert '#include "aslp_compat_kaldi' in open(path).read()


# ... and the reference's own unit test of the randomizers (linked as bin_ref/nnet-randomizer-test, run by tests/test_randomizer_gpu.py)
UNIT_TESTS = ["../aslp-nnet/nnet-randomizer-test"]
# ... and the five mains of src/aslp-parallelbin/ (include/aslp_compat_kaldi_parallel.h: the workers / servers with the reference's
# constructors over one communicator per process; run beside the engine's tools by tests/test_ref_mains_diff_gpu.py and tests/test_parallel_gpu.py)
PAR = "/root/reference/src/aslp-parallelbin"
PAR_MAINS = sorted("../aslp-parallelbin/" + f[:-3] for f in os.listdir(PAR) if f.endswith(".cc")) if os.path.isdir(PAR) else []


@pytest.mark.parametrize("main", MAINS + UNIT_TESTS + PAR_MAINS)
def test_reference_main_compiles_unchanged(main):
    src = os.path.join(REF, main + ".cc")
    if not os.path.exists(src) or not os.path.exists(HIPCC):
        pytest.skip("reference tree or hipcc not present")
    inc = ["-I" + os.path.join(ROOT, "include", "kaldi_compat"), "-I" + os.path.join(ROOT, "include")]
    inc += ["-I" + os.path.join(ROOT, "kaldi-aslp_amd", d) for d in ("nnet", "util", "csrc", "parallel")]
    # (host pass only: a main() has no device code of its own, and the device pass of the same headers is what `make refmains` runs)
    p = subprocess.run([HIPCC, "-x", "hip", "--offload-arch=gfx950", "--cuda-host-only", "-std=c++17", "-fsyntax-only"] + inc + [src], capture_output=True,
                       timeout=900)
    errs = [ln for ln in p.stderr.decode().splitlines() if "error" in ln]
    assert p.returncode == 0 and not errs, "\n".join(errs[:10])


def test_nothing_of_the_reference_is_copied_into_the_compat_layer():
    """the forwarding headers are two lines each; the compat header declares aliases and using-declarations only"""
    base = os.path.join(ROOT, "include", "kaldi_compat")
    for d, _, files in os.walk(base):
        for f in files:
            assert len(open(os.path.join(d, f)).read().splitlines()) <= 3
    # the compat headers themselves: aliases, using-declarations, the one reader class over the engine's two, OpenFst's symbol-table text format
    # and host matrices over the engine's host types -- written here, none of it the reference's text (the copy detector's job); what this test
    # pins is that they stay small
    for h, most in (("aslp_compat_kaldi.h", 280), ("aslp_compat_kaldi_matrix.h", 260), ("aslp_compat_kaldi_parallel.h", 90)):
        assert len(open(os.path.join(ROOT, "include", h)).read().splitlines()) <= most
