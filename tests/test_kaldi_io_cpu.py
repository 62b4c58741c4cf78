"""Kaldi stream formats and Table readers / writers of the host-only I/O layer (kaldi-aslp_amd/util, nnet/host-io.cpp),
driven through bin/aslp-table-copy (no GPU) and checked against tests/kaldi_formats.py, an independent statement of the
formats.  Covers SURVEY 8(f) N2: ark / scp / pipes / offsets for BaseFloatMatrix (FM, DM, CM, CM2, text), BaseFloatVector,
Posterior, Int32Vector; rspecifier / wspecifier options; error behaviour."""
import os
import subprocess

import numpy as np
import pytest

import kaldi_formats as kf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "kaldi-aslp_amd", "bin", "aslp-table-copy")


def run(*args, ok=True, stdin=None):
    p = subprocess.run([TOOL, "--print-args=false"] + list(args), input=stdin, capture_output=True, timeout=60)
    if ok:
        assert p.returncode == 0, p.stderr.decode()
    return p


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-C", os.path.join(ROOT, "kaldi-aslp_amd"), os.path.join(ROOT, "kaldi-aslp_amd", "bin", "aslp-table-copy")], check=True)


def mats(rng, n=4):
    return [("utt%d" % i, rng.standard_normal((int(rng.integers(1, 9)), 5)).astype(np.float32)) for i in range(n)]


def test_matrix_archive_binary_text_double_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    ms = mats(rng) + [("empty", np.zeros((0, 0), np.float32))]
    src = tmp_path / "in.ark"
    # a mix of float / double binary objects and a text object in ONE archive: every object carries its own header
    ents = []
    for i, (k, m) in enumerate(ms):
        ents.append((k, kf.matrix_bin(m, double=(i % 2 == 1)) if i != 2 else kf.matrix_txt(m)))
    src.write_bytes(kf.archive(ents))
    out = tmp_path / "out.ark"
    run("--type=matrix", "ark:%s" % src, "ark:%s" % out)
    got = kf.parse_bin_archive(out.read_bytes(), "matrix")
    assert [k for k, _ in got] == [k for k, _ in ms]
    for (k, a), (_, b) in zip(got, ms):
        if b.size == 0:
            assert a.size == 0
        else:
            np.testing.assert_allclose(a, b, rtol=0 if k != "utt2" else 1e-7, atol=0)
    # text out, then back in: bit-exact after one text round trip at the stream precision the writer sets? no: text is
    # 7 significant digits like the reference's (InitKaldiOutputStream precision), so compare to 1e-6 relative
    txt = tmp_path / "out.txt"
    run("--type=matrix", "ark:%s" % out, "ark,t:%s" % txt)
    lines = txt.read_text()
    assert lines.startswith("utt0  [\n  ")
    back = tmp_path / "back.ark"
    run("--type=matrix", "ark,t:%s" % txt, "ark:%s" % back)
    for (k, a), (_, b) in zip(kf.parse_bin_archive(back.read_bytes(), "matrix"), ms):
        if b.size:
            np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-7)


def test_compressed_matrices_decode_like_the_reference(tmp_path):
    rng = np.random.default_rng(1)
    rows, cols = 7, 6
    hdr = np.sort(rng.integers(0, 65536, (cols, 4)), axis=1).astype(np.uint16)
    data = rng.integers(0, 256, (cols, rows)).astype(np.uint8)
    data[0, :3] = [0, 64, 65]
    data[1, :3] = [192, 193, 255]
    d2 = rng.integers(0, 65536, (rows, cols)).astype(np.uint16)
    src = tmp_path / "c.ark"
    src.write_bytes(kf.archive([("a", kf.compressed_cm(-3.5, 11.25, hdr, data)), ("b", kf.compressed_cm2(0.25, 100.0, d2)),
                                ("c", kf.compressed_cm(0.0, 1.0, np.zeros((0, 4), np.uint16), np.zeros((0, 0), np.uint8)))]))
    out = tmp_path / "o.ark"
    run("--type=matrix", "ark:%s" % src, "ark:%s" % out)
    got = dict(kf.parse_bin_archive(out.read_bytes(), "matrix"))
    assert np.array_equal(got["a"], kf.decode_cm(-3.5, 11.25, hdr, data))
    assert np.array_equal(got["b"], kf.decode_cm2(0.25, 100.0, d2))
    assert got["c"].size == 0


def test_posterior_int_vector_and_ali_to_post(tmp_path):
    post = [[(3, 1.0)], [(5, 0.25), (7, 0.75)], [], [(0, 0.5)]]
    ali = [4, 4, 9, 0, 2999]
    (tmp_path / "p.ark").write_bytes(kf.archive([("u1", kf.posterior_bin(post)), ("u2", kf.posterior_txt(post[:2]))]))
    run("--type=posterior", "ark:%s" % (tmp_path / "p.ark"), "ark:%s" % (tmp_path / "p2.ark"))
    got = kf.parse_bin_archive((tmp_path / "p2.ark").read_bytes(), "posterior")
    assert got == [("u1", post), ("u2", post[:2])]
    run("--type=posterior", "ark:%s" % (tmp_path / "p2.ark"), "ark,t:%s" % (tmp_path / "p.txt"))
    assert (tmp_path / "p.txt").read_text().splitlines()[0] == "u1 [ 3 1 ] [ 5 0.25 7 0.75 ] [ ] [ 0 0.5 ] "
    (tmp_path / "a.ark").write_bytes(kf.archive([("u1", kf.int32vec_bin(ali)), ("u2", kf.int32vec_txt(ali[:3])), ("u3", kf.int32vec_txt([]))]))
    run("--type=int32-vector", "ark:%s" % (tmp_path / "a.ark"), "ark:%s" % (tmp_path / "a2.ark"))
    assert kf.parse_bin_archive((tmp_path / "a2.ark").read_bytes(), "int32-vector") == [("u1", ali), ("u2", ali[:3]), ("u3", [])]
    run("--type=int32-vector", "ark:%s" % (tmp_path / "a2.ark"), "ark,t:%s" % (tmp_path / "a.txt"))
    assert (tmp_path / "a.txt").read_text() == "u1 4 4 9 0 2999 \nu2 4 4 9 \nu3 \n"
    run("--type=ali-to-post", "ark:%s" % (tmp_path / "a.ark"), "ark:%s" % (tmp_path / "ap.ark"))
    got = kf.parse_bin_archive((tmp_path / "ap.ark").read_bytes(), "posterior")
    assert got[0] == ("u1", [[(a, 1.0)] for a in ali])
    # alignments where posteriors are expected: the reference's hint (hmm/posterior.cc:97-100)
    p = run("--type=posterior", "ark,t:%s" % (tmp_path / "a.txt"), "ark:/dev/null", ok=False)
    assert p.returncode != 0 and b"did you provide alignments instead of posteriors?" in p.stderr


def test_vectors(tmp_path):
    v = np.arange(5, dtype=np.float32) * 0.5 - 1
    (tmp_path / "v.ark").write_bytes(kf.archive([("a", kf.vector_bin(v)), ("b", kf.vector_bin(v, double=True)), ("c", b" [ 1 2.5 -3 ]\n")]))
    run("--type=vector", "ark:%s" % (tmp_path / "v.ark"), "ark:%s" % (tmp_path / "v2.ark"))
    got = kf.parse_bin_archive((tmp_path / "v2.ark").read_bytes(), "vector")
    assert np.array_equal(got[0][1], v) and np.array_equal(got[1][1], v) and np.array_equal(got[2][1], np.float32([1, 2.5, -3]))


def test_scp_offsets_pipes_and_random_access(tmp_path):
    rng = np.random.default_rng(2)
    ms = mats(rng, 6)
    ark, scp = tmp_path / "f.ark", tmp_path / "f.scp"
    src = tmp_path / "src.ark"
    src.write_bytes(kf.archive([(k, kf.matrix_bin(m)) for k, m in ms]))
    # ark,scp: the script file points into the archive with byte offsets (kaldi-table-inl.h:905-940)
    run("--type=matrix", "ark:%s" % src, "ark,scp:%s,%s" % (ark, scp))
    lines = scp.read_text().splitlines()
    assert len(lines) == 6
    raw = ark.read_bytes()
    for line, (k, m) in zip(lines, ms):
        key, rx = line.split(" ", 1)
        assert key == k
        path, off = rx.rsplit(":", 1)
        assert path == str(ark) and raw[int(off):int(off) + 5] == b"\0BFM "
    # read back through the scp in a shuffled order (sequential scp reader seeks inside the open file)
    shuffled = tmp_path / "sh.scp"
    order = [3, 0, 5, 1, 4, 2]
    shuffled.write_text("".join(lines[i] + "\n" for i in order))
    out = tmp_path / "o.ark"
    run("--type=matrix", "scp:%s" % shuffled, "ark:%s" % out)
    got = kf.parse_bin_archive(out.read_bytes(), "matrix")
    assert [k for k, _ in got] == [ms[i][0] for i in order]
    for (k, a), i in zip(got, order):
        assert np.array_equal(a, ms[i][1])
    # pipes on both sides + stdin/stdout
    p = run("--type=matrix", "ark:cat %s |" % src, "ark:| cat > %s" % (tmp_path / "piped.ark"))
    assert (tmp_path / "piped.ark").read_bytes() == src.read_bytes()
    p = run("--type=matrix", "ark:-", "ark:-", stdin=src.read_bytes())
    assert p.stdout == src.read_bytes()
    # scp entries that are commands
    cmd_scp = tmp_path / "cmd.scp"
    cmd_scp.write_text("k1 cat %s |\n" % (tmp_path / "one.mat"))
    (tmp_path / "one.mat").write_bytes(kf.matrix_bin(ms[0][1]))
    run("--type=matrix", "scp:%s" % cmd_scp, "ark:%s" % out)
    assert np.array_equal(kf.parse_bin_archive(out.read_bytes(), "matrix")[0][1], ms[0][1])
    # random access: archive (reads forward, keeps what it passed) and script
    for spec in ("ark:%s" % src, "scp:%s" % scp, "ark,s,cs:%s" % src):
        run("--type=matrix", "--random-access=true", spec, "ark:%s" % out)
        got = kf.parse_bin_archive(out.read_bytes(), "matrix")
        assert all(np.array_equal(a, m) for (_, a), (_, m) in zip(got, ms)) and len(got) == 6


def test_errors_and_permissive(tmp_path):
    rng = np.random.default_rng(3)
    m = rng.standard_normal((3, 4)).astype(np.float32)
    good = kf.matrix_bin(m)
    (tmp_path / "trunc.ark").write_bytes(kf.archive([("a", good), ("b", good[:-5])]))
    p = run("--type=matrix", "ark:%s" % (tmp_path / "trunc.ark"), "ark:/dev/null", ok=False)
    assert p.returncode != 0
    # scp with a missing file: error, unless permissive (p) -> skipped (kaldi-table-inl.h:150-200)
    (tmp_path / "g.mat").write_bytes(good)
    (tmp_path / "m.scp").write_text("a %s\nb %s\nc %s\n" % (tmp_path / "g.mat", tmp_path / "missing.mat", tmp_path / "g.mat"))
    assert run("--type=matrix", "scp:%s" % (tmp_path / "m.scp"), "ark:/dev/null", ok=False).returncode != 0
    run("--type=matrix", "scp,p:%s" % (tmp_path / "m.scp"), "ark:%s" % (tmp_path / "o.ark"))
    assert [k for k, _ in kf.parse_bin_archive((tmp_path / "o.ark").read_bytes(), "matrix")] == ["a", "c"]
    # malformed specifiers / options
    for bad in (["ark,scp:x", "ark:/dev/null"], ["ark:%s" % (tmp_path / "g.mat"), "scp,ark:a,b"], ["ark:/nonexistent/x", "ark:/dev/null"]):
        assert run("--type=matrix", *bad, ok=False).returncode != 0
    p = run("--no-such-option=1", "ark:-", "ark:-", ok=False)
    assert p.returncode != 0 and b"Invalid option --no-such-option=1" in p.stderr
    p = subprocess.run([TOOL], capture_output=True)
    assert p.returncode == 1 and b"Usage:  aslp-table-copy" in p.stderr and b"--random-access" in p.stderr
    # inconsistent text matrix
    (tmp_path / "bad.txt").write_text("a  [\n  1 2 3\n  4 5 ]\n")
    assert run("--type=matrix", "ark,t:%s" % (tmp_path / "bad.txt"), "ark:/dev/null", ok=False).returncode != 0


def test_config_file_and_option_forms(tmp_path):
    (tmp_path / "c.conf").write_text("# comment\n--type=int32-vector   # trailing comment\n")
    (tmp_path / "a.txt").write_text("u 1 2 3\n")
    p = run("--config=%s" % (tmp_path / "c.conf"), "--random_access", "ark:%s" % (tmp_path / "a.txt"), "ark,t:-")
    assert p.stdout == b"u 1 2 3 \n"
    # command line overrides the config file; "--" ends the options
    (tmp_path / "p.txt").write_text("u [ 1 0.5 ]\n")
    p = run("--config=%s" % (tmp_path / "c.conf"), "--type=posterior", "--", "ark:%s" % (tmp_path / "p.txt"), "ark,t:-")
    assert p.stdout == b"u [ 1 0.5 ] \n"
