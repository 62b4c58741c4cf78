"""End-to-end frames/s of the command-line path on BASELINE cfg2: aslp-nnet-init -> aslp-nnet-train-frame reading a
feature archive and a posterior archive from disk (page cache), randomizer 32768, minibatch 1024.  SURVEY 8(d): the
reference's own fps line counts I/O; this is that number for the tool."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kaldi_formats as kf
BIN = os.path.join(ROOT, "kaldi-aslp_amd", "bin")
tmp = sys.argv[1] if len(sys.argv) > 1 else "/tmp/e2e"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
os.makedirs(tmp, exist_ok=True)
proto = ["<NnetProto>"]
d = 440
for _ in range(5):
    proto += ["<AffineTransform> <InputDim> %d <OutputDim> 2048 <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.04" % d,
              "<BatchNormalization> <InputDim> 2048 <OutputDim> 2048", "<Sigmoid> <InputDim> 2048 <OutputDim> 2048"]
    d = 2048
proto += ["<AffineTransform> <InputDim> 2048 <OutputDim> 3000 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04",
          "<Softmax> <InputDim> 3000 <OutputDim> 3000", "</NnetProto>"]
open(tmp + "/nnet.proto", "w").write("\n".join(proto) + "\n")
rng = np.random.default_rng(0)
n_utt = frames // 500
with open(tmp + "/feats.ark", "wb") as ff, open(tmp + "/post.ark", "wb") as pf:
    for i in range(n_utt):
        T = 500
        f = rng.standard_normal((T, 440), dtype=np.float32)
        ff.write(("utt%05d " % i).encode() + kf.matrix_bin(f))
        lab = rng.integers(0, 3000, T)
        pf.write(("utt%05d " % i).encode() + kf.posterior_bin([[(int(l), 1.0)] for l in lab]))
subprocess.run([BIN + "/aslp-nnet-init", "--print-args=false", tmp + "/nnet.proto", tmp + "/nnet.init"], check=True, capture_output=True)
t0 = time.time()
p = subprocess.run([BIN + "/aslp-nnet-train-frame", "--print-args=false", "--learn-rate=0.00001", "--minibatch-size=1024", "--randomizer-size=32768",
                    "ark:%s/feats.ark" % tmp, "ark:%s/post.ark" % tmp, tmp + "/nnet.init", tmp + "/nnet.out"], capture_output=True)
wall = time.time() - t0
err = p.stderr.decode()
print([l for l in err.splitlines() if "fps" in l or "AvgLoss" in l or "ERROR" in l][-3:])
if os.environ.get("PROFILE") == "1":
    print("\n".join(l for l in err.splitlines() if "profile" in l.lower() or "\t" in l or "Time" in l)[-3000:])
print("frames %d, process wall %.2f s -> %.0f frames/s incl. start-up, model read / write" % (n_utt * 500, wall, n_utt * 500 / wall))
