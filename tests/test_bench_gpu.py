"""bench.py's N > 1 flow end to end on ONE GPU: with ASLP_COMM_TRANSPORT=shm the ranks are separate processes sharing the device
(parallel/comm.cpp ShmComm), so the launcher / rendezvous / BSP sync / max-over-ranks timing / cfg3_bsp block / exit status that the
driver's 2-, 4-, 8-GPU runs go through are exercised where only one GPU exists.  Both launch forms: bench.py's own launcher and
`python -m torch.distributed.run` (the driver's command line)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLEAN = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ASLP_COMM_FILE", "ASLP_COMM_TOKEN", "MASTER_ADDR", "MASTER_PORT")


def check_line(out, n):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert d["metric"] == "frames/sec (aslp-nnet-train)" and d["unit"] == "frames/sec"
    assert "ShmComm" in d["config"]["sync"] and d["config"]["parallelism"] == "bsp-dp%d" % n
    assert d["value"] == pytest.approx(d["config"]["global_batch"] * 1000.0 / d["ms_per_step"], rel=1e-3)   # whole-job frames over the max-over-ranks time
    bsp = d.get("cfg3_bsp")
    assert bsp and bsp["n_gpus"] == n and bsp["valid_frames_per_sec"] > 0
    # BASELINE's own multi-GPU configurations: cfg4 (cfg1 net, minibatch 256/GPU, BSP) and cfg5 (LC-BLSTM + Warp-CTC, EASGD server + workers)
    c4, c5 = d.get("cfg4_bsp"), d.get("cfg5_easgd")
    assert c4 and "error" not in c4 and c4["n_gpus"] == n and c4["frames_per_sec"] > 0 and c4["sync_ms"] > 0, c4
    assert c5 and "error" not in c5 and c5["workers"] == n - 1 and c5["valid_frames_per_sec"] > 0 and c5["sync_ms"] > 0, c5
    import math
    assert math.isfinite(c4["avg_xent_per_frame_rank0"]) and 0.0 < c4["avg_xent_per_frame_rank0"] < 20.0, c4     # the loss survived the exchanges
    assert math.isfinite(c5["avg_ctc_obj_per_sequence"]) and c5["avg_ctc_obj_per_sequence"] > 0.0, c5
    assert d["comm"]["ranks_seen"] == n and d["comm"]["transport"] == "shm"     # the top-level copy of the transport's own count
    for blk in (bsp, c4, c5, d["config"]["comm"]):
        assert blk["transport"] == "shm" and blk["ranks_seen"] == n and blk["scaling_measured"] is False   # says so when nothing was scaled
    return d


def test_bench_two_ranks_own_launcher():
    env = {k: v for k, v in os.environ.items() if k not in CLEAN}
    env["ASLP_COMM_TRANSPORT"] = "shm"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, "\n".join(l for l in p.stderr.decode().splitlines() if "ResetLstmStreams" not in l)[-4000:]
    check_line(p.stdout.decode(), 2)


def test_bench_two_ranks_under_torch_distributed_run():
    env = {k: v for k, v in os.environ.items() if k not in CLEAN}
    env["ASLP_COMM_TRANSPORT"] = "shm"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, "\n".join(l for l in p.stderr.decode().splitlines() if "ResetLstmStreams" not in l)[-4000:]
    check_line(p.stdout.decode(), 2)


def test_bench_three_ranks_server_and_two_workers():
    """`bench.py --gpus 3` as the driver would start it for N = 3: cfg4's BSP over three ranks, cfg5's EASGD server on rank 0 with TWO
    workers exchanging with it (the smallest run in which worker-server traffic interleaves), all on one GPU over shared memory --
    so the first real SCALE record cannot fail on plumbing."""
    env = {k: v for k, v in os.environ.items() if k not in CLEAN}
    env["ASLP_COMM_TRANSPORT"] = "shm"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "6", "--warmup", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, "\n".join(l for l in p.stderr.decode().splitlines() if "ResetLstmStreams" not in l)[-4000:]
    d = check_line(p.stdout.decode(), 3)
    assert d["cfg5_easgd"]["workers"] == 2 and d["cfg5_easgd"]["ranks_seen"] == 3 and d["cfg4_bsp"]["ranks_seen"] == 3



def test_single_gpu_line_quotes_every_fraction_against_the_instruction_it_issued():
    """`python bench.py` (N = 1, short): every `frac` in the line is algorithmic TFLOP/s over the peak of the instruction the
    kernels ISSUE (split-fp16 products: 2516 / 3 TF-equivalent), the ratio to the fp32 instruction's peak lives under `ratio_to_fp32_mfma_peak`;
    the headline carries `roofline` (+ the operand-feed block), `cpu_baseline` (thread probe named) and `cfg1_cpu_baseline` with the
    GPU / CPU ratio on the configuration the >= 30x target is stated on."""
    env = {k: v for k, v in os.environ.items() if k not in CLEAN}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--prewarm-steps", "50", "--no-e2e-tool"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    peak16 = 2516.0 / 3.0
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == pytest.approx(peak16) and r["frac"] == pytest.approx(r["achieved"] / peak16, rel=1e-6)
    assert "ratio_to_fp32_mfma_peak" in r and "frac_of_fp32_mfma_peak" not in r
    f = r["feed"]
    assert f["unit"] == "TB/s" and f["frac"] == pytest.approx(f["achieved"] / f["peak"], rel=1e-6) and 0.0 < f["frac"] < 1.0
    assert f["achieved"] == pytest.approx(f["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12, rel=1e-6)

    def walk(x, path=""):
        if isinstance(x, dict):
            if "frac" in x and "tflops" in x and "product_accuracy" not in path:
                assert x["frac"] <= x["tflops"] / peak16 * (1 + 1e-6) or x.get("peak_tflops") == 157.3, (path, x["frac"], x["tflops"])
            assert "frac_of_mfma_peak" not in x, path
            for k, v in x.items():
                walk(v, path + "/" + k)
    walk(d)
    for key in ("chunked_xent", "whole_utterance_warpctc"):
        blk = d["cfg3"][key]
        assert blk["frac"] == pytest.approx(blk["tflops"] / peak16, rel=1e-6) and blk["ratio_to_fp32_mfma_peak"] == pytest.approx(blk["tflops"] / 157.3, rel=1e-6)
    c1 = d["cfg1_gpu"]
    assert c1["frac"] == pytest.approx(c1["tflops"] / peak16, rel=1e-6)
    assert d["fp32_instruction"]["peak"] == 157.3          # (that block runs the fp32 instruction: its fraction is against 157.3)
    cb, cb1 = d["cpu_baseline"], d["cfg1_cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1
    if cb["kind"] == "reference":
        assert str(cb["cores"]) in cb["thread_probe_frames_per_sec"] and "thread_choice" in cb
    assert cb1["value"] > 0 and "WITHOUT BatchNormalization" in cb1["sample"]
    assert c1["vs_cfg1_cpu_baseline"] == pytest.approx(c1["frames_per_sec"] / cb1["value"], rel=1e-6) and c1["vs_cfg1_cpu_baseline"] > 30.0
