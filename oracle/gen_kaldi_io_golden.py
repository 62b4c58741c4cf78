#!/usr/bin/env python3
"""Generates tests/golden/kaldi_io/*: table files written by the REFERENCE's own tools (oracle/_ref/copy-feats, copy-vector,
copy-int-vector: `make -C oracle ref`, compiled from /root/reference/src where it lies).  TEST INFRASTRUCTURE.  The source
archives (src_*.ark) are written by tests/kaldi_formats.py from seeded numpy data; everything named ref_* is reference output:
binary and text archives, compressed feature archives (CM for > 8 rows, CM2 for <= 8 rows) and the reference's own decoding of
them, an archive + script pair with byte offsets.  Run from the repo root inside the development container."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kaldi_formats as kf  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "kaldi_io")
REF = os.path.join(ROOT, "oracle", "_ref")
os.makedirs(OUT, exist_ok=True)
os.chdir(OUT)  # script files hold the archive name as given on the command line: keep it relative


def run(tool, *args):
    subprocess.run([os.path.join(REF, tool), "--print-args=false"] + list(args), check=True, capture_output=True)


rng = np.random.default_rng(2024)
shapes = [(5, 13), (40, 13), (1, 13), (8, 13), (9, 13), (120, 13)]
feats = [("utt-%02d" % i, (rng.standard_normal(s) * rng.uniform(0.5, 20.0) + rng.uniform(-5, 5)).astype(np.float32)) for i, s in enumerate(shapes)]
feats[2][1][0, :4] = [0.0, -0.0, 1e-30, 3.4e38]  # extremes in the 1-row matrix
open("src_feats.ark", "wb").write(kf.archive([(k, kf.matrix_bin(m)) for k, m in feats]))
run("copy-feats", "ark:src_feats.ark", "ark:ref_feats_bin.ark")
run("copy-feats", "ark:src_feats.ark", "ark,t:ref_feats_txt.ark")
run("copy-feats", "--compress=true", "ark:src_feats.ark", "ark:ref_feats_cm.ark")
run("copy-feats", "ark:ref_feats_cm.ark", "ark:ref_feats_cm_decoded.ark")
run("copy-feats", "ark:src_feats.ark", "ark,scp:ref_feats_scp.ark,ref_feats.scp")

vecs = [("v%d" % i, (rng.standard_normal(n) * 3).astype(np.float32)) for i, n in enumerate((1, 7, 300))]
open("src_vec.ark", "wb").write(kf.archive([(k, kf.vector_bin(v)) for k, v in vecs]))
run("copy-vector", "ark:src_vec.ark", "ark:ref_vec_bin.ark")
run("copy-vector", "ark:src_vec.ark", "ark,t:ref_vec_txt.ark")

ints = [("a%d" % i, [int(x) for x in rng.integers(-5, 3000, n)]) for i, n in enumerate((0, 1, 25, 600))]
open("src_int.ark", "wb").write(kf.archive([(k, kf.int32vec_bin(v)) for k, v in ints]))
run("copy-int-vector", "ark:src_int.ark", "ark:ref_int_bin.ark")
run("copy-int-vector", "ark:src_int.ark", "ark,t:ref_int_txt.ark")
print("wrote", sorted(os.listdir(OUT)))
