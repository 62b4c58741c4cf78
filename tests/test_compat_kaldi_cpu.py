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
MAINS = ["aslp-nnet-train-frame", "aslp-nnet-info"]   # (the Makefile's `refmains` target links these two and aslp-nnet-copy / -init)


def test_compat_layer_has_a_forwarding_header_for_every_include_of_the_served_mains():
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present (GPU box)")
    for m in MAINS + ["aslp-nnet-copy", "aslp-nnet-init"]:
        for inc in re.findall(r'#include\s+"([^"]+)"', open(os.path.join(REF, m + ".cc")).read()):
            path = os.path.join(ROOT, "include", "kaldi_compat", inc)
            assert os.path.exists(path), "%s includes %s: no forwarding header" % (m, inc)
            assert '#include "aslp_compat_kaldi.h"' in open(path).read()


@pytest.mark.parametrize("main", MAINS)
def test_reference_main_compiles_unchanged(main):
    src = os.path.join(REF, main + ".cc")
    if not os.path.exists(src) or not os.path.exists(HIPCC):
        pytest.skip("reference tree or hipcc not present")
    inc = ["-I" + os.path.join(ROOT, "include", "kaldi_compat"), "-I" + os.path.join(ROOT, "include")]
    inc += ["-I" + os.path.join(ROOT, "kaldi-aslp_amd", d) for d in ("nnet", "util", "csrc")]
    p = subprocess.run([HIPCC, "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only"] + inc + [src], capture_output=True, timeout=900)
    errs = [ln for ln in p.stderr.decode().splitlines() if "error" in ln]
    assert p.returncode == 0 and not errs, "\n".join(errs[:10])


def test_nothing_of_the_reference_is_copied_into_the_compat_layer():
    """the forwarding headers are two lines each; the compat header declares aliases and using-declarations only"""
    base = os.path.join(ROOT, "include", "kaldi_compat")
    for d, _, files in os.walk(base):
        for f in files:
            assert len(open(os.path.join(d, f)).read().splitlines()) <= 3
    text = open(os.path.join(ROOT, "include", "aslp_compat_kaldi.h")).read()
    assert "class " not in text.split("#ifndef ASLP_COMPAT_KALDI_H_")[1].replace("template <typename Real> using", "")
