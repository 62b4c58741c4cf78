#!/usr/bin/env python3
"""bench.py -- frames/sec of the aslp-nnet training step on MI355X (BASELINE.json metric).

Workload (config 2 of BASELINE.json): 5 x 2048 sigmoid DNN with BatchNormalization after every
hidden AffineTransform, 440-dim input (40-dim fbank x 11 splice), 3000 pdf targets, minibatch
1024 per GPU; one step = Nnet::Propagate -> Xent::Eval -> Nnet::Backpropagate (+ Update of every
layer) on a synthetic minibatch already resident in HBM.  With N > 1 every rank trains its own
replica on its own shard and the replicas are averaged BSP-style (aslp-parallel/bsp-worker.cc)
with one RCCL all-reduce every `sync_period` frames (default 25600, as the reference).

The layer products run on the fp16 matrix instruction with every fp32 operand carried as two fp16 pieces (csrc/gemm_split16.hip; results
at least as close to float64 as the fp32 instruction's, `product_accuracy`); `fp32_instruction` is the same step with ASLP_GEMM_SPLIT_F16=0.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the roofline / cpu_baseline fields.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IN_DIM, HID, NH, OUT_DIM, MB = 440, 2048, 5, 3000, 1024
FLOP_PER_FRAME = 141131776.0   # 2W fwd + 2(W-W1) bwd-data + 2W wgrad
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2516.0  # MI355X_MICROARCH.md: dense fp16 MFMA (v_mfma_f32_32x32x16_f16)
SPLIT_PEAK_TF_EQUIV = F16_MFMA_PEAK_TFLOPS / 3.0   # three fp16 instructions per fp32-equivalent product: 839 TF-equivalent
L2_PEAK_TBS = 34.5   # MI355X_MICROARCH.md: L2 aggregate bandwidth (8 XCDs x 16 channels x 128 B/clk at 2.1 GHz fabric-side)
CFG2_LEARN_RATE = 0.008      # run_bn_dnn.sh:81-83 / SURVEY 8d (the kept weight planes' bound depends on it; rounds 1-4 timed at 1e-5)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
ARITHMETIC = "fp32 operands as 2xfp16 pieces behind power-of-two scales, fp32 accumulate"


def proto():
    lines = ["<NnetProto>"]
    d = IN_DIM
    for _ in range(NH):
        lines.append("<AffineTransform> <InputDim> %d <OutputDim> %d <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.04" % (d, HID))
        lines.append("<BatchNormalization> <InputDim> %d <OutputDim> %d" % (HID, HID))
        lines.append("<Sigmoid> <InputDim> %d <OutputDim> %d" % (HID, HID))
        d = HID
    lines.append("<AffineTransform> <InputDim> %d <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04" % (d, OUT_DIM))
    lines.append("<Softmax> <InputDim> %d <OutputDim> %d" % (OUT_DIM, OUT_DIM))
    lines.append("</NnetProto>")
    return "\n".join(lines) + "\n"


def cpu_baseline(seconds_budget=20.0):
    """The oracle's C chain (a port of the reference CPU path: cache-blocked AVX2 sgemm + OpenMP,
    oracle/aslp_oracle.c) timed on this host: same net, same minibatch size, a bounded number of
    steps.  The thread count is the best of a short probe over {all, half, 64, 32, 16} hardware
    threads (SMT siblings and many-socket hosts do not always help a 1024-row minibatch)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as oracle
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    d = oracle.lib.orc_dnn_create(IN_DIM, HID, NH, OUT_DIM, 1, MB, 777)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((MB, IN_DIM)).astype(np.float32)
    lab = rng.integers(0, OUT_DIM, MB).astype(np.int32)
    oracle.lib.orc_set_num_threads(avail)
    oracle.lib.orc_dnn_train_step(d, x, lab, 1e-5, 0.0)  # warm-up (page-in)
    best, cores = None, avail
    for th in sorted({avail, max(1, avail // 2), 64, 32, 16}, reverse=True):
        if th > avail:
            continue
        oracle.lib.orc_set_num_threads(th)
        t0 = time.time()
        oracle.lib.orc_dnn_train_step(d, x, lab, 1e-5, 0.0)
        el = time.time() - t0
        if best is None or el < best:
            best, cores = el, th
    oracle.lib.orc_set_num_threads(cores)
    t0 = time.time()
    steps = 0
    while True:
        oracle.lib.orc_dnn_train_step(d, x, lab, 1e-5, 0.0)
        steps += 1
        el = time.time() - t0
        if el > seconds_budget or steps >= 20:
            break
    oracle.lib.orc_dnn_destroy(d)
    return {"value": steps * MB / el, "unit": "frames/sec", "cores": cores, "kind": "port",
            "sample": "%d steps of minibatch %d of the same 5x2048+BN DNN (oracle/aslp_oracle.c: blocked AVX2 sgemm + OpenMP, "
                      "%d of %d hardware threads)" % (steps, MB, cores, avail)}


def cpu_baseline_reference(seconds_budget=15.0, cfg1=False):
    """The same training step on the REFERENCE's own CuMatrix / CuVector CPU code over OpenBLAS (oracle/_ref/ref_dnn_bench, built by
    `make -C oracle ref` in the development container from the reference sources where they lie; it travels with the snapshot and binds
    to the OpenBLAS of the image's scipy wheel, which the GPU box has too).  cfg1: the net WITHOUT BatchNormalization at minibatch 256, lr
    0.008 -- the configuration the >= 30x target names (aslp-nnetbin/aslp-nnet-train-frame.cc:109-131).  Thread count: best of a short probe,
    reported with every probed count.  Returns None when the binary is absent or does not run here (the caller then reports the port alone)."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_dnn_bench")
    if not os.path.exists(exe):
        return None
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def run(threads, seconds, steps):
        env = dict(os.environ, OPENBLAS_NUM_THREADS=str(threads), OMP_NUM_THREADS=str(threads))
        out = subprocess.run([exe, str(seconds), str(steps)] + (["cfg1"] if cfg1 else []), env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=240)
        return json.loads(out.stdout.decode().strip().splitlines()[-1])

    try:
        best, probe = None, {}
        probe_steps = 8 if cfg1 else 3   # (a probe is >= ~0.5 s of CPU work per thread count)
        for th in sorted({min(avail, 128), min(avail, 64), min(avail, 32), min(avail, 16), min(avail, 8)}, reverse=True):
            r = run(th, 1.5, probe_steps)
            probe[str(th)] = round(r["frames_per_sec"], 1)
            if best is None or r["frames_per_sec"] > best[1]:
                best = (th, r["frames_per_sec"])
        r = run(best[0], seconds_budget, 80 if cfg1 else 20)
    except Exception:   # noqa: BLE001 -- a baseline that cannot run is reported as absent, the port below still is
        return None
    if cfg1 and not (r.get("batch_norm") == 0 and r.get("minibatch") == 256):
        return None   # (an older binary that ignores the third argument)
    what = "5x2048 DNN WITHOUT BatchNormalization (BASELINE cfg1)" if cfg1 else "same 5x2048+BN DNN"
    return {"value": r["frames_per_sec"], "unit": "frames/sec", "cores": r["threads"], "kind": "reference",
            "thread_probe_frames_per_sec": probe,
            "thread_choice": "best of a %d-step probe per count; OpenBLAS threads only speed the sgemm calls up, the reference's element-wise kaldi-matrix passes "
                             "(sigmoid, softmax, BatchNormalization's vector ops, the SGD axpy) are single-threaded, and past the point where the products stop "
                             "dominating more threads only add OpenBLAS's fork / join and oversubscription cost -- which is why a small count wins on a "
                             "%d-thread host" % (probe_steps, avail),
            "sample": "%d steps of minibatch %d of the %s on the reference's own CuMatrix CPU code + OpenBLAS (oracle/_ref/ref_dnn_bench: "
                      "aslp-cudamatrix / matrix sources compiled where they lie, operations issued in Nnet::Propagate / Backpropagate order by "
                      "oracle/ref_dnn_bench.cpp), %d OpenBLAS threads of %d hardware threads" % (r["steps"], r["minibatch"], what, r["threads"], avail)}


def ctc_rel_err(aslp, dev):
    """Second half of BASELINE.json's metric ("CTC-loss fp32 rel-err"): the HIP forward-backward against the outputs
    the REFERENCE's own CPU code produced for tests/golden/ctc_a128_t200.bin (data file, generator oracle/gen_ctc_golden.cpp)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctc_golden
    g = ctc_golden.load("a128_t200")
    acts = torch.from_numpy(g["acts"].reshape(g["maxT"] * g["mb"], g["A"])).to(dev)
    labels, o = [], 0
    for l in g["label_lengths"]:
        labels.append([int(v) for v in g["flat_labels"][o:o + l]])
        o += l
    costs, grads = aslp.ops.ctc_loss(acts, labels, g["input_lengths"])
    fin = np.isfinite(g["costs"])
    rel_cost = float(np.max(np.abs(costs[fin] - g["costs"][fin]) / np.maximum(np.abs(g["costs"][fin]), 1e-6)))
    gr = grads.cpu().numpy().reshape(-1).astype(np.float64)
    rel_grad = float(np.linalg.norm(gr - g["grads"]) / np.linalg.norm(g["grads"]))
    return {"max_rel_err_cost": rel_cost, "rel_frobenius_err_grad": rel_grad, "tolerance": 1e-4,
            "fixture": "tests/golden/ctc_a128_t200.bin (alphabet 128, %d utterances, T <= %d; reference CPU output)" % (g["mb"], g["maxT"])}


LC_FLOP_PER_ROW = 70.25e6  # 4 x BLstmProjectedStreamsLC (C 512, R 256, in 40) + Affine 512 -> 128, fwd + bwd + wgrad


def cfg3_block(aslp, dev):
    """BASELINE.json configs[2] beside the headline (extra key `cfg3`): 4 x BLstmProjectedStreamsLC (C 512, R 256, in 40) +
    AffineTransform 512 -> 128, S = 32 streams, lr 1e-5, momentum 0.9, synthetic data resident in HBM.
      (i)  chunked: chunk 40 + 20 frames of right context (T = 60), Softmax + Xent on the chunk frames, the streams' state
           carried from chunk to chunk -- the loop body of aslp-nnet-train-blstm-streams-lc.cc;
      (ii) whole utterances of U(200, 800) frames padded to the longest, Warp-CTC on the pre-softmax activations (L = T / 4
           labels) -- the loop body of aslp-nnet-train-warp-ctc-streams.cc:158-223, with ResetLstmStreams telling the LC
           component its streams (Nnet::SetSeqLengths does not reach it in the reference, nnet-nnet.cc:498-530).
    Times are wall clock around K steps (synchronised both sides); the recurrence / CTC shares come from a second pass with
    HIP-event region timers on the launch stream (include/aslp_kernels.h aslp_region_*)."""
    import numpy as np
    import torch
    S, CHUNK, RIGHT, A = 32, 40, 20, 128

    def proto(softmax):
        lines, d = ["<NnetProto>"], 40
        for _ in range(4):
            lines.append("<BLstmProjectedStreamsLC> <InputDim> %d <OutputDim> 512 <CellDim> 512 <ParamScale> 0.02 <ClipGradient> 5.0" % d)
            d = 512
        lines.append("<AffineTransform> <InputDim> 512 <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04" % A)
        if softmax:
            lines.append("<Softmax> <InputDim> %d <OutputDim> %d" % (A, A))
        return "\n".join(lines + ["</NnetProto>"]) + "\n"

    def regions(names):
        out = {}
        for n in names:
            ms = C.c_double()
            cnt = aslp.lib.aslp_region_get(n.encode(), C.byref(ms))
            out[n] = (int(cnt), ms.value)
        return out

    def timed(step, warm, k, k_prof, names):
        for i in range(warm):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            step(warm + i)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / k
        aslp.lib.aslp_region_reset()
        aslp.lib.aslp_region_profile(1)
        for i in range(k_prof):
            step(warm + k + i)
        torch.cuda.synchronize()
        aslp.lib.aslp_region_profile(0)
        r = {n: v[1] / k_prof for n, v in regions(names).items()}
        aslp.lib.aslp_region_reset()
        return el, r

    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    out = {"model": "4 x BLstmProjectedStreamsLC (C 512, R 256, in 40) + AffineTransform 512->128", "streams": S, "dtype": "f32",
           "flop_per_row": LC_FLOP_PER_ROW, "peak_tflops": SPLIT_PEAK_TF_EQUIV,
           "peak_note": "the instruction issued is v_mfma_f32_*_f16 x 3 per fp32-equivalent product (2516 / 3 = 839 TF-equivalent): every `frac` "
                        "below is algorithmic fp32-equivalent TFLOP/s over THAT peak; `ratio_to_fp32_mfma_peak` is the same rate over the fp32 "
                        "instruction's 157.3 TFLOP/s -- a ratio of rates, not a utilisation (the persistent recurrences spend FOUR fp16 multiplies "
                        "per fp32 multiply, hi / lo rows x hi / lo columns, so for their share even 839 is generous)",
           "recurrence": "persistent kernels, one launch per layer and pass (csrc/rnn_persistent.hip); their products on v_mfma_f32_16x16x32_f16 "
                         "with every fp32 operand as two fp16 pieces behind power-of-two scales, fp32 accumulation (error below an fp32 fma chain's: "
                         "devtools/micro/f16_split.hip; ASLP_LSTM_SPLIT_F16=0 = the fp32 instruction); weight gradients on the side stream beside the "
                         "recurrence below (ASLP_LSTM_SIDE_GRADS=0 = one stream)"}
    # (i) chunked + Xent
    T = CHUNK + RIGHT
    net = aslp.Nnet.Init(proto(True), seed=777)
    net.SetTrainOptions(learn_rate=1e-5, momentum=0.9)
    net.SetChunkSize(CHUNK)
    xent = aslp.Xent()
    x = torch.randn(T * S, 40, device=dev, generator=g)
    labels = torch.randint(0, A, (T * S,), device=dev, generator=g, dtype=torch.int32)
    fw = torch.ones(T * S, device=dev)
    fw.view(T, S)[CHUNK:] = 0   # right-context frames carry no loss

    def step_x(i):
        net.ResetLstmStreams([1] * S if i == 0 else [0] * S)
        net.TrainStepXent(xent, x, labels, fw)

    el, r = timed(step_x, 5, 30, 10, ("lstm_recurrence_fwd", "lstm_recurrence_bwd"))
    rec = r["lstm_recurrence_fwd"] + r["lstm_recurrence_bwd"]
    tf = LC_FLOP_PER_ROW * T * S / el / 1e12
    out["chunked_xent"] = {"chunk": CHUNK, "right_context": RIGHT, "rows_per_step": T * S, "ms_per_step": el * 1e3,
                           "valid_frames_per_sec": CHUNK * S / el, "tflops": tf, "frac": tf / SPLIT_PEAK_TF_EQUIV,
                           "ratio_to_fp32_mfma_peak": tf / F32_MFMA_PEAK_TFLOPS,
                           "recurrence_ms_per_step": rec, "recurrence_share": rec / (el * 1e3),
                           "recurrent_launches_per_layer_per_pass": 1}
    # the same step with the recurrent products on the fp32 matrix instruction (aslp_lstm_split16(0)): beside the default, for the reader who
    # wants the figure without fp16 multiplies anywhere
    try:
        aslp.lib.aslp_lstm_split16(0)
        el32, _ = timed(step_x, 5, 30, 0, ())
        out["chunked_xent"]["fp32_instruction_recurrence"] = {"ms_per_step": el32 * 1e3, "valid_frames_per_sec": CHUNK * S / el32,
                                                              "ratio_to_fp32_mfma_peak": LC_FLOP_PER_ROW * T * S / el32 / 1e12 / F32_MFMA_PEAK_TFLOPS,
                                                              "note": "only the recurrences' products are on the fp32 instruction here; the layers' batched "
                                                                      "products stay on the split-fp16 kernels, so no single instruction peak applies"}
    finally:
        aslp.lib.aslp_lstm_split16(-1)
    del net
    # (ii) whole utterances + Warp-CTC
    rng = np.random.default_rng(99)
    lens = rng.integers(200, 801, S).astype(np.int32)
    lens[0] = 800
    Tm = int(lens.max())
    lab = [[int(v) for v in rng.integers(1, A, max(1, int(t) // 4))] for t in lens]
    netc = aslp.Nnet.Init(proto(False), seed=777)
    netc.SetTrainOptions(learn_rate=1e-5 / float(lens.sum()), momentum=0.9)   # the tool's per-frame normalisation of the step
    ctc = aslp.WarpCtc()
    xc = torch.randn(Tm * S, 40, device=dev, generator=g)

    def step_c(i):
        netc.ResetLstmStreams([1] * S)
        netc.TrainStepWarpCtc(ctc, xc, lens, lab)

    el, r = timed(step_c, 1, 3, 2, ("lstm_recurrence_fwd", "lstm_recurrence_bwd", "ctc_loss"))
    rec = r["lstm_recurrence_fwd"] + r["lstm_recurrence_bwd"]
    tf = LC_FLOP_PER_ROW * Tm * S / el / 1e12
    st = ctc.GetStats()
    out["whole_utterance_warpctc"] = {"max_frames": Tm, "valid_frames": int(lens.sum()), "rows_per_step": Tm * S, "labels_per_utt": "T/4",
                                      "ms_per_step": el * 1e3, "valid_frames_per_sec": float(lens.sum()) / el, "rows_per_sec": Tm * S / el,
                                      "tflops": tf, "frac": tf / SPLIT_PEAK_TF_EQUIV, "ratio_to_fp32_mfma_peak": tf / F32_MFMA_PEAK_TFLOPS,
                                      "recurrence_ms_per_step": rec,
                                      "recurrence_share": rec / (el * 1e3), "ctc_ms_per_step": r["ctc_loss"],
                                      "ctc_share": r["ctc_loss"] / (el * 1e3), "avg_ctc_obj_per_sequence": st["obj"] / max(st["sequences"], 1.0)}
    return out


def recurrent_family_block(aslp, dev):
    """One recurrent layer (512 in, H / C = 512, S = 32 streams, T = 60 frames) + Affine 512 -> 128 + Softmax + Xent per family member of
    SURVEY 8(a) rows a8-a10: train-step time through the engine, reported as microseconds per timestep (forward + backward).
    Extra key `recurrent_layers` (N = 1 only); a few hundredths of a second each."""
    import torch
    S, T, A = 32, 60, 128
    cases = [("GruStreams", "<GruStreams> <InputDim> 512 <OutputDim> 512 <ParamScale> 0.02 <ClipGradient> 5.0", 512),
             ("LstmProjectedStreams", "<LstmProjectedStreams> <InputDim> 512 <OutputDim> 256 <CellDim> 512 <ParamScale> 0.02 <ClipGradient> 5.0", 256),
             ("BLstmProjectedStreams", "<BLstmProjectedStreams> <InputDim> 512 <OutputDim> 512 <CellDim> 512 <ParamScale> 0.02 <ClipGradient> 5.0", 512),
             ("Lstm", "<Lstm> <InputDim> 512 <OutputDim> 512 <ParamScale> 0.01 <ClipGradient> 5.0", 512)]
    out = {"streams": S, "frames": T, "unit": "us per timestep, forward + backward, one layer"}
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    x = torch.randn(T * S, 512, device=dev, generator=g)
    lab = torch.randint(0, A, (T * S,), device=dev, generator=g, dtype=torch.int32)
    for name, line, od in cases:
        proto = ("<NnetProto>\n%s\n<AffineTransform> <InputDim> %d <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04\n"
                 "<Softmax> <InputDim> %d <OutputDim> %d\n</NnetProto>\n" % (line, od, A, A, A))
        net = aslp.Nnet.Init(proto, seed=1)
        net.SetTrainOptions(learn_rate=1e-5, momentum=0.9)
        xent = aslp.Xent()
        net.SetSeqLengths([T] * S)
        for i in range(5):
            net.ResetLstmStreams([1] * S if i == 0 else [0] * S)
            net.TrainStepXent(xent, x, lab)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        for i in range(n):
            net.ResetLstmStreams([0] * S)
            net.TrainStepXent(xent, x, lab)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / n * 1e6 / T
    return out


def cfg1_gpu_block(aslp, dev):
    """BASELINE.json configs[0] / the per-GPU leg of configs[3] on the GPU (extra key `cfg1_gpu`, N = 1): the 5 x 2048 sigmoid DNN WITHOUT
    BatchNormalization, minibatch 256, learn rate 0.008, no momentum (run_dnn.sh:60-101; the frame tool's loop body,
    aslp-nnet-train-frame.cc:109-131).  256 rows are a quarter of cfg2's minibatch: a 2048 x 2048 layer product has 128 tiles of 64 x 64 for
    256 CUs, so this is the small-batch regime of the same kernels."""
    import torch
    mb = 256
    lines, d = ["<NnetProto>"], IN_DIM
    for _ in range(NH):
        lines.append("<AffineTransform> <InputDim> %d <OutputDim> %d <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.04" % (d, HID))
        lines.append("<Sigmoid> <InputDim> %d <OutputDim> %d" % (HID, HID))
        d = HID
    lines += ["<AffineTransform> <InputDim> %d <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04" % (d, OUT_DIM),
              "<Softmax> <InputDim> %d <OutputDim> %d" % (OUT_DIM, OUT_DIM), "</NnetProto>"]
    net = aslp.Nnet.Init("\n".join(lines) + "\n", seed=777)
    net.SetTrainOptions(learn_rate=0.008, momentum=0.0)
    xent = aslp.Xent()
    g = torch.Generator(device=dev)
    g.manual_seed(2468)
    x = torch.randn(mb, IN_DIM, device=dev, generator=g)
    labels = torch.randint(0, OUT_DIM, (mb,), device=dev, generator=g, dtype=torch.int32)
    warm, k = 100, 400
    for _ in range(warm):
        net.TrainStepXent(xent, x, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        net.TrainStepXent(xent, x, labels)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / k
    st = xent.GetStats()
    tf = FLOP_PER_FRAME * mb / el / 1e12
    return {"workload": "cfg1 net on the GPU: 5x2048 sigmoid DNN, no BatchNorm, minibatch 256, lr 0.008, Propagate + Xent + Backpropagate + SGD update",
            "steps": k, "warmup": warm, "ms_per_step": el * 1e3, "frames_per_sec": mb / el, "tflops": tf, "frac": tf / SPLIT_PEAK_TF_EQUIV, "peak_tflops": SPLIT_PEAK_TF_EQUIV,
            "ratio_to_fp32_mfma_peak": tf / F32_MFMA_PEAK_TFLOPS,
            "peak_note": "split-fp16 products: 2516 / 3 TF-equivalent is the peak of the instruction issued; the ratio to the fp32 instruction's 157.3 is a ratio of rates",
            "avg_xent_per_frame": (st["loss"] - st["entropy"]) / max(st["frames"], 1.0)}


def fp32_instruction_block(aslp, dev, steps, warmup):
    """Extra key `fp32_instruction` (N = 1): the SAME cfg2 step (same net, options, data, step counts) with every product on the fp32 matrix
    instruction v_mfma_f32_32x32x2_f32 (aslp_gemm_split16(0) = ASLP_GEMM_SPLIT_F16=0) -- the figure without an fp16 multiply anywhere."""
    import torch
    aslp.lib.aslp_gemm_split16(0)
    try:
        net = aslp.Nnet.Init(proto(), seed=777)
        net.SetTrainOptions(learn_rate=CFG2_LEARN_RATE, momentum=0.0)
        xent = aslp.Xent()
        g = torch.Generator(device=dev)
        g.manual_seed(1234)
        x = torch.randn(MB, IN_DIM, device=dev, generator=g)
        labels = torch.randint(0, OUT_DIM, (MB,), device=dev, generator=g, dtype=torch.int32)
        for _ in range(warmup):
            net.TrainStepXent(xent, x, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net.TrainStepXent(xent, x, labels)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        st = xent.GetStats()
    finally:
        aslp.lib.aslp_gemm_split16(-1)
    tf = FLOP_PER_FRAME * MB / el / 1e12
    return {"value": MB / el, "unit": "frames/sec", "ms_per_step": el * 1e3, "steps": steps, "warmup": warmup, "algorithmic_tflops": tf,
            "frac": tf / F32_MFMA_PEAK_TFLOPS, "peak": F32_MFMA_PEAK_TFLOPS,
            "avg_xent_per_frame": (st["loss"] - st["entropy"]) / max(st["frames"], 1.0),
            "instruction": "v_mfma_f32_32x32x2_f32 (aslp_gemm_split16(0) / ASLP_GEMM_SPLIT_F16=0)"}


def product_accuracy_block(aslp, dev):
    """Extra key `product_accuracy` (N = 1): for every product shape of BASELINE's configs -- K = 40, 256, 440, 512, 2048 (and the minibatch-long
    reductions of the weight gradients), all three operand layouts -- max |C - C_float64| / sum |a||b| on either instruction.  The split path is the
    default because it is at least as close to float64 as the fp32 instruction on every one of them (`split_le_fp32`)."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    shapes = []
    for K in (40, 256, 440, 512, 2048):
        rows = 1920 if K in (40, 256, 512) else 1024      # cfg3 (T S = 1920) / cfg2 (minibatch 1024)
        shapes += [("NT", 0, 1, rows, 2048, K), ("NN", 0, 0, rows, 2048, K), ("TN", 1, 0, 2048, 2048 if K == 2048 else 512, K)]
    shapes += [("TN", 1, 0, 2048, 2048, 1024), ("TN", 1, 0, 3000, 2048, 1024), ("TN", 1, 0, 2048, 512, 1920), ("NT", 0, 1, 1024, 3000, 2048),
               ("NN", 0, 0, 1024, 2048, 3000), ("NT", 0, 1, 256, 2048, 2048), ("TN", 1, 0, 2048, 2048, 256)]
    rows_out, all_le = [], True
    try:
        for name, tA, tB, M, N, K in shapes:
            A = torch.randn((K, M) if tA else (M, K), device=dev, generator=g)
            B = torch.randn((N, K) if tB else (K, N), device=dev, generator=g) * 0.05
            opA, opB = (A.t() if tA else A).double(), (B.t() if tB else B).double()
            ref, mag = opA @ opB, opA.abs() @ opB.abs()
            err = {}
            served = False
            for key, on in (("fp32", 0), ("split", 1)):
                aslp.lib.aslp_gemm_split16(on)
                C_ = torch.zeros(M, N, device=dev)
                aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C_)
                if on:
                    served = aslp.lib.aslp_gemm_last_tile() in (304, 308, 311, 328, 351)
                err[key] = ((C_.double() - ref).abs() / mag).max().item()
            le = err["split"] <= err["fp32"] * 1.0000001
            all_le = all_le and le
            rows_out.append({"layout": name, "M": M, "N": N, "K": K, "split": err["split"], "fp32_instruction": err["fp32"],
                             "on_fp16_instruction": bool(served), "split_le_fp32": bool(le)})
    finally:
        aslp.lib.aslp_gemm_split16(-1)
    return {"metric": "max |C - C_f64| / sum |a||b|", "split_le_fp32": bool(all_le), "shapes": rows_out,
            "note": "on_fp16_instruction false: the shape is below the split kernels' floor (K < 64 or an extent < 128) and runs the fp32 instruction either way"}


def hbm_kernels_block(aslp, dev):
    """Extra key `hbm_kernels` (N = 1): the bandwidth-bound kernels of the path, each timed alone with HIP events on the launch stream:
    algorithmic bytes (every tensor read once and written once per pass, as the reference's own launches would move them) / time
    = GB/s and the fraction of the 8 TB/s HBM peak.  cfg2 shapes (1024 x 2048, 1024 x 3000), the randomizer cache (32768 x 440), cfg5's
    CompactFsmn / RowConvolution swaps."""
    import numpy as np
    import torch

    def timed_us(fn, n=50, warm=5, batches=21):
        # warm, then the MEDIAN of `batches` back-to-back groups of max(1, n // 5) calls: one slow group (a lazy code-object load, a clock
        # dip) no longer sets the figure
        for _ in range(max(warm, 10)):
            fn()
        reps = max(1, n // 5)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(batches + 1)]
        torch.cuda.synchronize()
        ev[0].record()
        for b in range(batches):
            for _ in range(reps):
                fn()
            ev[b + 1].record()
        torch.cuda.synchronize()
        return float(np.median([ev[b].elapsed_time(ev[b + 1]) * 1e3 / reps for b in range(batches)]))

    out = {}

    def rec(name, us, nbytes, what):
        out[name] = {"us": us, "algorithmic_bytes": nbytes, "gb_per_s": nbytes / us / 1e3, "frac_of_hbm_peak": nbytes / us / 1e3 / HBM_PEAK_GBS, "what": what}

    g = torch.Generator(device=dev)
    g.manual_seed(7)
    R, D = MB, HID
    x = torch.randn(R, D, device=dev, generator=g)
    dy = torch.randn(R, D, device=dev, generator=g) * 1e-2
    o, xh, idf = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    scale, shift = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    mean, istd, dsc, dsh = (torch.zeros(D, device=dev) for _ in range(4))
    rec("bn_forward", timed_us(lambda: aslp.ops.bn_forward(x, o, xh, scale, shift, mean, istd)), 12 * R * D,
        "BatchNormalization forward 1024 x 2048 (statistics + normalise, two reads + one write per element in the reference's passes)")
    rec("bn_backward", timed_us(lambda: aslp.ops.bn_backward(x, dy, xh, scale, mean, istd, dsc, dsh, 0.0, idf)), 20 * R * D,
        "BatchNormalization backward 1024 x 2048")
    acts = torch.randn(R, OUT_DIM, device=dev, generator=g)
    post = torch.softmax(acts, 1).contiguous()
    lab = torch.randint(0, OUT_DIM, (R,), device=dev, generator=g, dtype=torch.int32)
    diff, fw, stats = torch.empty_like(acts), torch.ones(R, device=dev), torch.zeros(5, device=dev, dtype=torch.float64)
    rec("xent", timed_us(lambda: aslp.ops.xent_eval(post, fw, diff, stats, labels=lab)), 8 * R * OUT_DIM, "Xent::Eval 1024 x 3000 (read posteriors, write diff)")
    # ... and as the engine issues it since round 5: the rows' launch per evaluation, the sum into the accumulators once per 32 evaluations
    room = torch.empty(32, R, 5, device=dev, dtype=torch.float64)
    rows_us = timed_us(lambda: aslp.ops.xent_eval_rows(post, fw, diff, room[0], lab))
    sum_us = timed_us(lambda: aslp.ops.xent_sum_rowstats(room, R, 32, stats), 20)
    rec("xent_per_step", rows_us + sum_us / 32, 8 * R * OUT_DIM,
        "Xent::Eval 1024 x 3000 as the training step pays for it: aslp_xent_eval_rows per evaluation + 1/32 of aslp_xent_sum_rowstats over 32 evaluations")
    out["xent_per_step"]["rows_launch_us"] = rows_us
    out["xent_per_step"]["sum_of_32_us"] = sum_us
    sm = torch.empty_like(acts)
    rec("softmax", timed_us(lambda: aslp.ops.softmax(sm, acts)), 8 * R * OUT_DIM, "Softmax 1024 x 3000")
    rows = 32768
    feats = torch.randn(rows, 40, device=dev, generator=g)
    spl = torch.empty(rows, 440, device=dev)
    offs = torch.arange(-5, 6, device=dev, dtype=torch.int32)
    rec("splice", timed_us(lambda: aslp.ops.splice(spl, feats, offs), 20), 4 * rows * (40 + 440), "Splice -5..5, 32768 x 40 -> 32768 x 440")
    cache = torch.randn(rows, 440, device=dev, generator=g)
    shuf = torch.empty_like(cache)
    mask = torch.randperm(rows, device=dev, generator=g).to(torch.int32)
    rec("randomize", timed_us(lambda: aslp.ops.randomize(shuf, cache, mask), 20), 8 * rows * 440, "MatrixRandomizer row gather, 32768 x 440")
    sg, sy = torch.empty_like(x), torch.empty_like(x)
    rec("sigmoid", timed_us(lambda: aslp.ops.sigmoid(sy, x)), 8 * R * D, "Sigmoid forward 1024 x 2048")
    rec("diff_sigmoid", timed_us(lambda: aslp.ops.diff_sigmoid(sg, sy, dy)), 12 * R * D, "Sigmoid backward 1024 x 2048")
    # cfg5 swaps: the component's passes at the kernel level (the launches a training step makes for it), and the same step through the
    # engine's Python entry points (which add an input copy, an output copy and two diff copies around a one-component net)
    T, Dm, P, F = 800, 512, 30, 30
    xf = torch.randn(T, Dm, device=dev, generator=g)
    odf = torch.randn(T, Dm, device=dev, generator=g) * 0.01
    coef = torch.randn(P + F + 1, Dm, device=dev, generator=g) * 0.05
    corr, of, idff = torch.zeros_like(coef), torch.empty_like(xf), torch.empty_like(xf)

    def fsmn():
        aslp.ops.fsmn_forward(of, xf, coef, P, F)
        aslp.ops.fsmn_backward(idff, corr, coef, xf, odf, P, F, 0.0, 1e-5)
    rec("compact_fsmn", timed_us(fsmn, 30), 4 * Dm * T * 7,
        "CompactFsmn 512, 30 + 30 taps, T = 800: forward; in-diff + tap gradients; their sum + update (3 launches; 2 + 5 tensor passes)")
    net = aslp.Nnet.Init("<NnetProto>\n<CompactFsmn> <InputDim> 512 <OutputDim> 512 <PastContext> 30 <FutureContext> 30 <LearnRateCoef> 1.0\n</NnetProto>\n")
    net.SetTrainOptions(learn_rate=1e-5)

    def fsmn_engine():
        net.Propagate(xf)
        net.Backpropagate(odf, want_in_diff=True)
    out["compact_fsmn"]["through_engine_us"] = timed_us(fsmn_engine, 30)
    S, K = 32, 20
    lens = np.random.default_rng(0).integers(T // 2, T + 1, S)
    lens[0] = T
    x2 = torch.randn(T * S, Dm, device=dev, generator=g)
    od2 = torch.randn(T * S, Dm, device=dev, generator=g) * 0.01
    w = torch.randn(Dm * (K + 1), device=dev, generator=g) * 0.1
    wd, wc, o2, idf2 = torch.zeros_like(w), torch.zeros_like(w), torch.empty_like(x2), torch.empty_like(x2)
    sl = torch.from_numpy(lens.astype(np.int32)).to(dev)

    def rowconv():
        aslp.ops.rowconv_forward(o2, x2, w, sl, K)
        aslp.ops.rowconv_backward(idf2, wd, x2, od2, w, sl, K, wc, 0.9, 1e-5)
    rec("row_convolution", timed_us(rowconv, 10), 4 * Dm * T * S * 7,
        "RowConvolution 512, FutureContext 20, T = 800, S = 32: forward; in-diff + tap partials in one pass; their sum + momentum + update (3 launches; 7 tensor passes)")
    net2 = aslp.Nnet.Init("<NnetProto>\n<RowConvolution> <InputDim> 512 <OutputDim> 512 <FutureContext> 20\n</NnetProto>\n")
    net2.SetTrainOptions(learn_rate=1e-5, momentum=0.9)
    net2.SetSeqLengths(lens)

    def rowconv_engine():
        net2.Propagate(x2)
        net2.Backpropagate(od2, want_in_diff=True)
    out["row_convolution"]["through_engine_us"] = timed_us(rowconv_engine, 10)
    return {"peak_gb_per_s": HBM_PEAK_GBS, "timing": "HIP events (torch.cuda.Event on the launch stream): >= 10 warm-up calls, then the median over 21 back-to-back groups of calls", "kernels": out}


def e2e_tool_block(frames=1024000, bindir="bin"):
    """SURVEY 8(d) asks for the end-to-end figure beside the compute-only one (extra key `e2e_tool`, N = 1): the cfg2 net trained by the
    command-line tool itself -- aslp-nnet-init, then aslp-nnet-train-frame reading a feature archive and a posterior archive (page
    cache), randomizer 32768, minibatch 1024 -- and the tool's OWN `fps` figure, whose timer spans archive parsing, the randomizer,
    uploads, every step and the model write like the reference's (aslp-nnet-train-frame.cc:99-139).  A child process: this one's GPU
    state is untouched.  Returns an {"error": ...} record instead of raising."""
    import re
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import kaldi_formats as kf
    train_bindir = os.path.join(ROOT, "kaldi-aslp_amd", bindir)   # "bin_ref": the REFERENCE's own main() built on the engine
    bindir = os.path.join(ROOT, "kaldi-aslp_amd", "bin")
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 6 * frames * IN_DIM else None
    tmp = tempfile.mkdtemp(prefix="aslp_e2e_", dir=base)
    try:
        with open(os.path.join(tmp, "nnet.proto"), "w") as f:
            f.write(proto())
        rng = np.random.default_rng(0)
        utt = 1000
        n_utt = frames // utt
        with open(os.path.join(tmp, "feats.ark"), "wb") as ff, open(os.path.join(tmp, "post.ark"), "wb") as pf:
            for i in range(n_utt):
                ff.write(("utt%05d " % i).encode() + kf.matrix_bin(rng.standard_normal((utt, IN_DIM), dtype=np.float32)))
                pf.write(("utt%05d " % i).encode() + kf.posterior_bin([[(int(l), 1.0)] for l in rng.integers(0, OUT_DIM, utt)]))
        subprocess.run([os.path.join(bindir, "aslp-nnet-init"), "--print-args=false", os.path.join(tmp, "nnet.proto"), os.path.join(tmp, "nnet.init")],
                       check=True, capture_output=True, timeout=600)
        t0 = time.time()
        p = subprocess.run([os.path.join(train_bindir, "aslp-nnet-train-frame"), "--print-args=false", "--learn-rate=%g" % CFG2_LEARN_RATE, "--minibatch-size=%d" % MB,
                            "--randomizer-size=32768", "ark:%s/feats.ark" % tmp, "ark:%s/post.ark" % tmp, os.path.join(tmp, "nnet.init"),
                            os.path.join(tmp, "nnet.out")], capture_output=True, timeout=900)
        wall = time.time() - t0
        err = p.stderr.decode(errors="replace")
        m = re.findall(r"fps\s*([0-9.eE+]+)", err)
        if p.returncode != 0 or not m:
            return {"error": "aslp-nnet-train-frame rc %d: %s" % (p.returncode, err[-300:])}
        return {"tool": "aslp-nnet-init + aslp-nnet-train-frame (randomizer 32768, minibatch %d) on binary archives in the page cache" % MB,
                "frames": n_utt * utt, "frames_per_sec": float(m[-1]), "process_wall_s": wall,
                "timer": "the tool's own fps line: archive parsing, randomizer, uploads, every step and the model write are inside it"}
    except Exception as e:   # noqa: BLE001 -- an extra block never takes the headline down
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cfg3_bsp_block(aslp, dev, native_parallel, comm, rank, world, sync_period):
    """N > 1 only (extra key `cfg3_bsp`): every rank trains its own replica of the cfg3 LC-BLSTM (chunked, Xent) on its own synthetic shard and
    the replicas are averaged BSP-style by the native BspWorker every `sync_period` valid frames -- the configuration BASELINE.json's 8-GPU
    target is quoted on.  Whole-job valid frames/s, barrier + synchronise on both sides, max over ranks."""
    import torch
    S, CHUNK, RIGHT, A = 32, 40, 20, 128
    T = CHUNK + RIGHT
    lines, d = ["<NnetProto>"], 40
    for _ in range(4):
        lines.append("<BLstmProjectedStreamsLC> <InputDim> %d <OutputDim> 512 <CellDim> 512 <ParamScale> 0.02 <ClipGradient> 5.0" % d)
        d = 512
    lines += ["<AffineTransform> <InputDim> 512 <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04" % A,
              "<Softmax> <InputDim> %d <OutputDim> %d" % (A, A), "</NnetProto>"]
    net = aslp.Nnet.Init("\n".join(lines) + "\n", seed=777)
    net.SetTrainOptions(learn_rate=1e-5, momentum=0.9)
    net.SetChunkSize(CHUNK)
    xent = aslp.Xent()
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    x = torch.randn(T * S, 40, device=dev, generator=g)
    labels = torch.randint(0, A, (T * S,), device=dev, generator=g, dtype=torch.int32)
    fw = torch.ones(T * S, device=dev)
    fw.view(T, S)[CHUNK:] = 0
    worker = native_parallel.BspWorker(comm)
    worker.InitParam(net)
    since = 0
    steps, warm = 100, 10

    def step(i):
        nonlocal since
        net.ResetLstmStreams([1] * S if i == 0 else [0] * S)
        net.TrainStepXent(xent, x, labels, fw)
        since += CHUNK * S
        if since >= sync_period:
            worker.Synchronize(since)
            since = 0

    for i in range(warm):
        step(i)
    worker.Synchronize(max(since, 1))
    since = 0
    comm.Barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    torch.cuda.synchronize()
    comm.Barrier()
    torch.cuda.synchronize()
    el = comm.MaxOverRanks(time.perf_counter() - t0)
    worker.close()
    out = {"workload": "cfg3 LC-BLSTM (4 x 512-cell, chunk 40 + 20, S = 32) + Xent, BSP every %d valid frames" % sync_period, "n_gpus": world,
           "steps": steps, "ms_per_step": el * 1e3 / steps, "valid_frames_per_sec": world * steps * CHUNK * S / el, "scaling": "weak"}
    out.update(_comm_facts(comm))
    return out


def _comm_facts(comm):
    """what the transport itself says about the group: proves which transport carried the run and that every rank joined it"""
    t = comm.Transport()
    return {"transport": t, "ranks_seen": comm.RanksSeen(), "ranks_seen_source": "ncclCommCount" if t == "rccl" else "ranks that joined the shared-memory segment",
            "scaling_measured": t == "rccl",
            "note": None if t == "rccl" else "ranks share GPUs and stage tensors through host shared memory: a functional run of the N > 1 flow, NOT a scaling measurement"}


def cfg4_bsp_block(aslp, dev, native_parallel, comm, rank, world, sync_period):
    """N > 1 only (extra key `cfg4_bsp`), BASELINE.json configs[3]: the cfg1 net (5 x 2048 sigmoid DNN, no BatchNormalization), minibatch 256 per
    GPU, learn rate 0.008, every rank on its own shard, BSP model averaging every `sync_period` frames (run_parallel.sh;
    aslp-nnet-train-frame-worker.cc:109-188; bsp-worker.cc:33-65).  Whole-job frames/s (barrier + synchronise on both sides, max over ranks) and
    the time the ranks spend inside Synchronize."""
    import torch
    mb = 256
    lines, d = ["<NnetProto>"], IN_DIM
    for _ in range(NH):
        lines.append("<AffineTransform> <InputDim> %d <OutputDim> %d <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.04" % (d, HID))
        lines.append("<Sigmoid> <InputDim> %d <OutputDim> %d" % (HID, HID))
        d = HID
    lines += ["<AffineTransform> <InputDim> %d <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04" % (d, OUT_DIM),
              "<Softmax> <InputDim> %d <OutputDim> %d" % (OUT_DIM, OUT_DIM), "</NnetProto>"]
    net = aslp.Nnet.Init("\n".join(lines) + "\n", seed=777)
    net.SetTrainOptions(learn_rate=0.008, momentum=0.0)
    xent = aslp.Xent()
    g = torch.Generator(device=dev)
    g.manual_seed(4321 + rank)
    x = torch.randn(mb, IN_DIM, device=dev, generator=g)
    labels = torch.randint(0, OUT_DIM, (mb,), device=dev, generator=g, dtype=torch.int32)
    worker = native_parallel.BspWorker(comm)
    worker.InitParam(net)
    steps, warm = 400, 100
    since, sync_s, syncs = 0, 0.0, 0

    def step(timed):
        nonlocal since, sync_s, syncs
        net.TrainStepXent(xent, x, labels)
        since += mb
        if since >= sync_period:
            if timed:
                torch.cuda.synchronize()
                t = time.perf_counter()
            worker.Synchronize(since)
            if timed:
                torch.cuda.synchronize()
                sync_s += time.perf_counter() - t
                syncs += 1
            since = 0

    for _ in range(warm):
        step(False)
    worker.Synchronize(max(since, 1))
    since = 0
    comm.Barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    comm.Barrier()
    torch.cuda.synchronize()
    el = comm.MaxOverRanks(time.perf_counter() - t0)
    sync_ms = comm.MaxOverRanks(sync_s * 1e3 / max(syncs, 1))
    worker.close()
    st = xent.GetStats()   # (rank 0's shard: also proves the loss stayed finite through the exchanges)
    out = {"workload": "cfg4: 5x2048 sigmoid DNN (no BatchNorm), minibatch 256/GPU, lr 0.008, BSP every %d frames, distinct shards" % sync_period,
           "n_gpus": world, "steps": steps, "ms_per_step": el * 1e3 / steps, "frames_per_sec": world * steps * mb / el, "syncs_timed": syncs,
           "sync_ms": sync_ms, "scaling": "weak", "avg_xent_per_frame_rank0": (st["loss"] - st["entropy"]) / max(st["frames"], 1.0)}
    out.update(_comm_facts(comm))
    return out


def cfg5_easgd_block(aslp, dev, native_parallel, comm, rank, world, sync_period):
    """N > 1 only (extra key `cfg5_easgd`), BASELINE.json configs[4]: the cfg3 LC-BLSTM on whole utterances with Warp-CTC, rank 0 the EASGD
    parameter server (alpha = 0.5: easgd-server.cc:37-86), ranks 1 .. N-1 the workers that exchange their model with it every `sync_period`
    valid frames (easgd-worker.cc:37-67).  Whole-job valid frames/s of the N - 1 workers; the server's time is the workers' (it serves until
    the last one has stopped)."""
    import numpy as np
    import torch
    S, A = 32, 128
    lines, d = ["<NnetProto>"], 40
    for _ in range(4):
        lines.append("<BLstmProjectedStreamsLC> <InputDim> %d <OutputDim> 512 <CellDim> 512 <ParamScale> 0.02 <ClipGradient> 5.0" % d)
        d = 512
    lines += ["<AffineTransform> <InputDim> 512 <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04" % A, "</NnetProto>"]
    net = aslp.Nnet.Init("\n".join(lines) + "\n", seed=777)
    rng = np.random.default_rng(99 + rank)
    lens = rng.integers(200, 801, S).astype(np.int32)
    lens[0] = 800
    Tm = int(lens.max())
    lab = [[int(v) for v in rng.integers(1, A, max(1, int(t) // 4))] for t in lens]
    net.SetTrainOptions(learn_rate=1e-5 / float(lens.sum()), momentum=0.9)
    steps, warm = 6, 1
    comm.Barrier()
    if rank == 0:
        class _Ptr:   # a raw device pointer as a torch tensor (the server works on the model's own tensors)
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}
        params = [torch.as_tensor(_Ptr(p_, n_), device=dev) for p_, n_ in net.GetGpuParams(writers_announce=True) if n_ > 0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        native_parallel.ServerRun(comm, "easgd", params, p0=0.5)
        torch.cuda.synchronize()
        el_local, frames_local, sync_ms_local = time.perf_counter() - t0, 0.0, 0.0
    else:
        ctc = aslp.WarpCtc()
        g = torch.Generator(device=dev)
        g.manual_seed(99 + rank)
        xc = torch.randn(Tm * S, 40, device=dev, generator=g)
        worker = native_parallel.EasgdWorker(comm, 0.5)
        worker.InitParam(net)
        since, sync_s, syncs = 0, 0.0, 0
        for _ in range(warm):
            net.ResetLstmStreams([1] * S)
            net.TrainStepWarpCtc(ctc, xc, lens, lab)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net.ResetLstmStreams([1] * S)
            net.TrainStepWarpCtc(ctc, xc, lens, lab)
            since += int(lens.sum())
            if since >= sync_period:
                torch.cuda.synchronize()
                t = time.perf_counter()
                worker.Synchronize(since)
                torch.cuda.synchronize()
                sync_s += time.perf_counter() - t
                syncs += 1
                since = 0
        torch.cuda.synchronize()
        el_local = time.perf_counter() - t0
        worker.Stop()
        worker.close()
        frames_local, sync_ms_local = float(steps * int(lens.sum())), sync_s * 1e3 / max(syncs, 1)
        cst = ctc.GetStats()
        obj_local = cst["obj"] / max(cst["sequences"], 1.0)
    comm.Barrier()
    el = comm.MaxOverRanks(el_local if rank != 0 else 0.0)
    frames = sum(comm.AllReduceHostDouble([frames_local]))
    sync_ms = comm.MaxOverRanks(sync_ms_local)
    obj = sum(comm.AllReduceHostDouble([obj_local if rank != 0 else 0.0])) / max(world - 1, 1)   # (a non-finite worker objective makes this non-finite)
    out = {"workload": "cfg5: 4 x BLstmProjectedStreamsLC (C 512) + Warp-CTC on whole utterances (S = 32, T <= 800), EASGD alpha 0.5, server on rank 0, "
                       "%d worker(s), exchange every %d valid frames" % (world - 1, sync_period),
           "n_gpus": world, "workers": world - 1, "steps_per_worker": steps, "valid_frames_per_sec": frames / el if el > 0 else 0.0,
           "ms_per_step": el * 1e3 / steps, "sync_ms": sync_ms, "scaling": "weak", "avg_ctc_obj_per_sequence": obj}
    out.update(_comm_facts(comm))
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` from a bare shell (no launcher environment): start N ranks, one process per GPU, and hand
    back rank 0's JSON line.  This parent never touches the GPU (no torch.cuda call, no HIP call): the children are ordinary
    child processes, nothing is exec'ed over a process that has initialised the device.  The ranks meet through a
    rendezvous file carrying a per-launch token (kaldi-aslp_amd/parallel/comm.cpp) -- no MPI, no torch.distributed."""
    import secrets
    import subprocess
    import tempfile
    token = secrets.token_hex(8)
    comm_file = os.path.join(tempfile.gettempdir(), "aslp_bench_comm_" + token)
    argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), ASLP_COMM_FILE=comm_file, ASLP_COMM_TOKEN=token)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on this driver
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = b""
    rc = 0
    import threading
    got = {}
    reader = threading.Thread(target=lambda: got.setdefault("out", procs[0].stdout.read()), daemon=True)   # rank 0 prints the line; it ends after the last collective
    reader.start()
    t_end = time.time() + args.launch_timeout
    try:
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            bad = [c for c in codes if c not in (None, 0)]
            if bad:                              # a rank has failed (e.g. its communicator did not come up): the others would wait for it in a
                rc = bad[0]                      # collective until their own time-outs -- end them now, with the failing rank's status
                break
            if time.time() > t_end:
                rc = 124
                break
            time.sleep(0.1)
    finally:
        for p in procs:                        # exactly the processes started here, never a pattern
            if p.poll() is None:
                p.kill()
                p.wait()
                rc = rc or 124
        for p in procs:
            rc = rc or (p.returncode or 0)
        reader.join(timeout=10.0)
        out0 = got.get("out", b"") or b""
        for f in (comm_file, comm_file + ".ctl"):
            try:
                os.unlink(f)
            except OSError:
                pass
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return rc


def comm_failed(rank, world, err):
    """A rank whose communicator does not come up ends the run: one line on stderr with the transport's own error string, exit status 4.
    Nothing is retried and nothing is re-executed -- the launcher (bench.py's own, or torch.distributed.run) sees the status, ends the other
    ranks and returns non-zero; a bench line is never printed for a job that did not have all its ranks."""
    sys.stderr.write("bench.py: rank %d of %d: communicator did not come up: %s\n" % (rank, world, err))
    sys.stderr.flush()
    os._exit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--prewarm-steps", type=int, default=400, help="untimed steps in front of the warm-up when fewer than 400 steps are timed (clock ramp)")
    ap.add_argument("--sync-period", type=int, default=25600, help="frames between BSP model syncs (N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-profile", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--no-update-overlap", action="store_true",
                    help="keep the weight-gradient GEMMs on the main stream (per-kernel profiles: every kernel alone on the chip)")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the LC-BLSTM (BASELINE cfg3) block of the JSON line")
    ap.add_argument("--cfg3-bsp-timeout", type=int, default=420, help="N > 1: seconds the extra cfg4_bsp / cfg3_bsp / cfg5_easgd blocks may take before they are dropped")
    ap.add_argument("--launch-timeout", type=int, default=3000, help="N > 1 from a bare shell: seconds the parent waits for rank 0")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the cfg2 step itself (profiler passes: nothing but the timed workload's kernels in the trace); implies the --no-* flags")
    ap.add_argument("--no-e2e-tool", action="store_true", help="skip the end-to-end command-line block of the JSON line (extra key e2e_tool)")
    ap.add_argument("--e2e-frames", type=int, default=1024000)
    ap.add_argument("--dry-run-ranks", action="store_true", help=argparse.SUPPRESS)   # launcher plumbing test (no GPU): tests/test_bench_cpu.py
    args = ap.parse_args()
    if args.headline_only:
        args.no_cfg3 = args.no_e2e_tool = args.no_cpu_baseline = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        sys.exit(launch_ranks(args))          # before anything touches the GPU
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if args.dry_run_ranks:
        if os.environ.get("ASLP_BENCH_DRYRUN_FAIL_RANK") == str(rank):
            raise SystemExit(3)
        if os.environ.get("ASLP_BENCH_DRYRUN_COMM_ERROR") and os.environ.get("ASLP_BENCH_DRYRUN_COMM_ERROR_RANK", "0") == str(rank):
            # (launcher test: the path a failed ncclCommInitRank takes, without a GPU)
            comm_failed(rank, world, RuntimeError(os.environ["ASLP_BENCH_DRYRUN_COMM_ERROR"]))
        if rank == 0:
            print(json.dumps({"n_gpus": world, "dry_run": True, "comm_file": os.environ.get("ASLP_COMM_FILE"), "token": os.environ.get("ASLP_COMM_TOKEN")}))
        return

    import torch
    if os.environ.get("ASLP_COMM_TRANSPORT") == "shm":
        # ranks as separate processes that may SHARE GPUs (parallel/comm.cpp ShmComm): lets `bench.py --gpus N` exercise the whole N > 1 flow --
        # launcher, rendezvous, BSP sync, max-over-ranks timing, the cfg3_bsp block -- on a one-GPU box.  Not a scaling measurement.
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ASLP_BENCH_FORCE_SYNC=1 exercises the sync path on a single GPU as well (RCCL all-reduce over a group of one)
    force_sync = os.environ.get("ASLP_BENCH_FORCE_SYNC") == "1"
    # The communicator comes up BEFORE the model is allocated and the first kernel runs: initialising RCCL afterwards
    # leaves every later step ~1 ms slower on this stack (measured, devtools/dbg_sync2.py: 1.44 vs 2.39 ms/step).
    # It is the product's own RcclComm (kaldi-aslp_amd/parallel/comm.cpp through libaslp_parallel.so), not torch.distributed.
    comm = None
    if world > 1 or force_sync:
        import aslp_import
        aslp_import.load()
        from kaldi_aslp_amd import native_parallel
        comm_file = os.environ.get("ASLP_COMM_FILE")
        token = os.environ.get("ASLP_COMM_TOKEN")
        if world > 1 and not comm_file:
            # started by torch.distributed.run (the driver's N > 1 command): every rank sees the same MASTER_PORT and the
            # same parent (the elastic agent), which together name this launch
            import tempfile
            token = "%s-%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
            comm_file = os.path.join(tempfile.gettempdir(), "aslp_bench_comm_" + token)
        try:
            comm = native_parallel.ProcessComm(comm_file, rank=rank, num_nodes=world, token=token, timeout_s=900)   # RcclComm unless ASLP_COMM_TRANSPORT=shm
        except Exception as e:   # noqa: BLE001 -- ncclGetUniqueId / ncclCommInitRank failed or timed out (the native message carries RCCL's own error string)
            comm_failed(rank, world, e)
        seen = comm.RanksSeen()
        if seen != world:
            comm_failed(rank, world, RuntimeError("the communicator counts %d ranks (ncclCommCount / joined segment), the launcher started %d" % (seen, world)))

    import aslp_import
    aslp = aslp_import.load()   # raises if libaslp_hip.so is missing (no fallback)
    aslp.ops.use_torch_stream()
    net = aslp.Nnet.Init(proto(), seed=777)            # same init on every rank (like one aslp-nnet-init model)
    net.SetTrainOptions(learn_rate=CFG2_LEARN_RATE, momentum=0.0)   # run_bn_dnn.sh:81-83,99-101
    if args.no_update_overlap:
        net.SetUpdateOverlap(False)
    xent = aslp.Xent()
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)                          # every rank its own shard
    x = torch.randn(MB, IN_DIM, device=dev, generator=g)
    labels = torch.randint(0, OUT_DIM, (MB,), device=dev, generator=g, dtype=torch.int32)
    worker = None
    if comm is not None:                                # bsp-worker.cc:33-65 as reimplemented (parallel/workers.cpp), RCCL all-reduce
        worker = native_parallel.BspWorker(comm)
        worker.InitParam(net)

    frames_since_sync = 0

    def step():
        nonlocal frames_since_sync
        net.TrainStepXent(xent, x, labels)
        frames_since_sync += MB
        if worker is not None and frames_since_sync >= args.sync_period:
            worker.Synchronize(frames_since_sync)
            frames_since_sync = 0

    # Clocks first: after idle the chip needs several hundred milliseconds under load before its clocks settle (a 20-step run from a cold
    # chip measures 5 % less than the same 20 steps half a second later: devtools/bench_gemm.py).  A fixed number of untimed
    # steps -- the same on every rank, so the sync schedule stays aligned -- precedes the W warm-up steps when the run itself is short.
    prewarm = max(0, args.prewarm_steps - args.warmup) if args.steps < 400 else 0
    cold_value = None
    if prewarm > 0 and worker is None:
        # the same W warm-up + K timed steps WITHOUT the pre-warm, first thing on the cold chip (extra key `cold_value`): how much of the
        # headline depends on the pre-warm is then visible in the line itself.  Its steps count towards the pre-warm.
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        cold_value = world * args.steps * MB / (time.perf_counter() - tc)
        prewarm_left = max(0, prewarm - args.warmup - args.steps)
    else:
        prewarm_left = prewarm
    for _ in range(prewarm_left):
        step()
    for _ in range(args.warmup):
        step()
    if worker is not None:
        # part of the warm-up: one model sync, so that the communicator's buffers and the pack / scale / unpack kernels are
        # loaded before the clock starts (a first sync inside K = 50 steps costs more than the 50 steps)
        worker.Synchronize(max(frames_since_sync, 1))
        frames_since_sync = 0
    if comm is not None:
        comm.Barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if comm is not None:
        comm.Barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.MaxOverRanks(elapsed)
    # Five more windows of the same K steps (N = 1): `value` above is the driver-comparable one; the spread of short windows on this box
    # -- minimum, median, maximum -- goes into the line beside it (extra key `windows`)
    windows = []
    if comm is None:
        for _ in range(5):
            torch.cuda.synchronize()
            tw = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            windows.append(world * args.steps * MB / (time.perf_counter() - tw))

    # Per-kernel durations for the roofline: HIP events around every GEMM launch on the launch stream, over the SAME
    # K steps run once more -- two event records per GEMM inside the timed region cost ~7 % of `value` (measured),
    # so `value` above is timed without them and the kernel timings come from this second, identical pass.
    # The weight-gradient GEMMs normally run on a side stream beside the rest of the backward pass; for the per-kernel
    # figure they are put back in line (SetUpdateOverlap(False)), so each duration is that kernel alone on the chip.
    if not args.no_gemm_profile:
        net.SetUpdateOverlap(False)
        aslp.lib.aslp_gemm_profile(1)
        aslp.lib.aslp_gemm_profile_reset()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        aslp.lib.aslp_gemm_profile(0)
        net.SetUpdateOverlap(not args.no_update_overlap)
    else:
        aslp.lib.aslp_gemm_profile_reset()

    # per-variant GEMM statistics of the timed region (HIP events on the launch stream)
    gemm = {}
    for vi, name in enumerate(("NT", "NN", "TN", "TT")):
        fl, ms = C.c_double(), C.c_double()
        n = aslp.lib.aslp_gemm_profile_get(vi, C.byref(fl), C.byref(ms))
        if n > 0:
            gemm[name] = {"launches": int(n), "flop_per_launch": fl.value / n, "avg_us": ms.value * 1e3 / n if ms.value > 0 else None,
                          "tflops": fl.value / ms.value / 1e9 if ms.value > 0 else None}
    st = xent.GetStats()  # also proves the loss stayed finite

    if rank == 0:
        total_frames = world * args.steps * MB
        value = total_frames / elapsed
        out = {
            "metric": "frames/sec (aslp-nnet-train)", "value": value, "unit": "frames/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "prewarm_steps": prewarm, "cold_value": cold_value, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "arithmetic": ARITHMETIC, "data": "synthetic",
            "config": {"workload": "cfg2: 5x2048 sigmoid DNN + BatchNorm, 440 in (40 fbank x 11 splice), 3000 pdfs, minibatch 1024/GPU, "
                                   "Propagate + Xent + Backpropagate + SGD update",
                       "global_batch": world * MB, "parallelism": "bsp-dp%d" % world,
                       "sync": ("native BspWorker on ShmComm (ranks sharing GPUs, tensors staged through shared memory: a functional run, not a scaling measurement)"
                                if os.environ.get("ASLP_COMM_TRANSPORT") == "shm" else
                                "native BspWorker on RcclComm (libaslp_parallel.so: ncclAllReduce over the parameter tensors in HBM)") if comm is not None else None,
                       "sync_period_frames": args.sync_period, "comm": _comm_facts(comm) if comm is not None else None,
                       "learn_rate": CFG2_LEARN_RATE, "avg_xent_per_frame": (st["loss"] - st["entropy"]) / max(st["frames"], 1.0)},
        }
        if comm is not None:
            out["comm"] = _comm_facts(comm)   # (also under config.comm): transport and the number of ranks the transport itself counts
        if windows:
            ws = sorted([value] + windows)
            out["windows"] = {"steps_each": args.steps, "count": len(ws), "min": ws[0], "median": ws[len(ws) // 2], "max": ws[-1],
                              "note": "`value` and five further windows of the same K steps, back to back on this box"}
        timed = {k: v for k, v in gemm.items() if v["tflops"]}
        if timed:
            dom = max(timed, key=lambda k: timed[k]["avg_us"] * timed[k]["launches"])
            d = timed[dom]
            traffic, traffic_src = None, None
            try:  # per-launch L2<->fabric bytes of this kernel from the committed PMC passes of the same command
                with open(os.path.join(ROOT, "profiles", "dnn_cfg2_pmc.json")) as f:   # (refreshed per round; the entry names the tile it was taken with)
                    pmc = json.load(f)[dom]
                traffic = (2.0 * pmc["fetch_kb"] + pmc["write_kb"]) * 1024.0
                traffic_src = "NOT measured in this run: committed PMC pass profiles/dnn_cfg2_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 gfx950 correction)"
            except (OSError, KeyError, ValueError):
                pass
            tile_buf = C.create_string_buffer(256)
            tile_cfg = aslp.lib.aslp_gemm_profile_tile(("NT", "NN", "TN", "TT").index(dom), tile_buf, 256)
            tile_name = tile_buf.value.decode()
            if traffic is not None and pmc.get("cfg") not in (None, tile_cfg):
                traffic, traffic_src = None, "committed PMC pass was taken with tile cfg %s, this run used %d: omitted" % (pmc.get("cfg"), tile_cfg)
            split = tile_cfg in (304, 308, 311, 328, 351)   # the product ran on the fp16 instruction (three per fp32-equivalent multiply)
            peak = SPLIT_PEAK_TF_EQUIV if split else F32_MFMA_PEAK_TFLOPS
            out["roofline"] = {"bound": "mfma", "kernel": "aslp_sgemm<%s> (%s; cfg %d)" % (dom, tile_name, tile_cfg), "achieved": d["tflops"], "peak": peak,
                               "unit": "TFLOP/s (fp32-equivalent)" if split else "TFLOP/s", "frac": d["tflops"] / peak,
                               "peak_note": ("dense fp16 MFMA peak %.0f TFLOP/s / 3 instructions per fp32-equivalent product" % F16_MFMA_PEAK_TFLOPS) if split
                                            else "fp32 MFMA peak",
                               "ratio_to_fp32_mfma_peak": d["tflops"] / F32_MFMA_PEAK_TFLOPS,
                               "ratio_to_fp32_mfma_peak_note": ("a ratio of rates (this kernel's fp32-equivalent rate over the fp32 instruction's peak), NOT a utilisation: the "
                                                               "fp32 instruction is not what runs; the matrix pipe's busy fraction is `matrix_pipe_busy`") if split else None,
                               "matrix_pipe_busy": (pmc.get("mfma_busy") if traffic is not None else None),
                               "matrix_pipe_busy_source": ("NOT measured in this run: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) of the committed PMC pass "
                                                           "(profiles/dnn_cfg2_pmc.json)") if traffic is not None else None,
                               "what_bounds_it": ("round-6 counters and ablations (profiles/r06_gemm_*; DESIGN 7 'Round 6'): INSIDE the K loop the L2 is ~80 % busy "
                                                  "(TCC_BUSY) feeding LDS-DMA -- a 64 x 128 tile per CU pulls (BM + BN) x K x 4 B through L2 -> LDS, 402 MB per "
                                                  "1024 x 2048 x 2048 product -- while the matrix pipe is 45-50 % busy and the LDS 25 %; with the matrix "
                                                  "instructions running the shader clock sits ~20-25 % below the clock of the same kernel without them, and the "
                                                  "launch takes what its LDS-DMA traffic alone takes at that clock (ablations: DMA + MFMA without LDS reads = the full "
                                                  "kernel; every request an L2 hit = the full kernel).  Around the loop ~8 us per launch are fixed (dispatch, first "
                                                  "tile, C stores, kernel end) and the weight gradients add an HBM-bound epilogue (W, gradient and the new W's planes: "
                                                  "64-80 MB); `feed` below quotes the operand traffic against the L2's peak") if split else None,
                               "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                               "flop_per_launch": d["flop_per_launch"], "avg_launch_us": d["avg_us"], "launches": d["launches"],
                               "launch_mix": {"NT": "per step: the 440-input layer, four 2048 x 2048 hidden layers and the 3000-column output layer, all on the named tile",
                                              "NN": "per step: four hidden layers' in-diff products and the output layer's (K = 3000), all on the named tile",
                                              "TN": "per step SIX weight-gradient launches on TWO kernel templates: four 2048 x 2048 x 1024 on gemm_s16_ks128<true> (128 x 128; the "
                                                    "named one) and the 440-input and 3000-output layers' on gemm_s16_glds<64,128,2,2,3,false,false,0,true>; `achieved` and "
                                                    "`avg_launch_us` average all six (flop_per_launch = their mean)"}.get(dom),
                               "timing": "HIP events on the launch stream, second pass over the same K steps"}
            if split:
                # operand bytes every launch pulls through L2 -> LDS, from the tile geometry the library uses for cfg2's shapes (64 x 128 tiles;
                # 128 x 128 for the 2048 x 2048 weight gradients): ceil(M / BM) ceil(N / BN) (BM + BN) pad64(K) 4 B
                def feed(M, N, K, bm, bn):
                    return -(-M // bm) * -(-N // bn) * (bm + bn) * (-(-K // 64) * 64) * 4.0
                per_step = {"NT": [feed(MB, HID, IN_DIM, 64, 128)] + [feed(MB, HID, HID, 64, 128)] * (NH - 1) + [feed(MB, OUT_DIM, HID, 64, 128)],
                            "NN": [feed(MB, HID, HID, 64, 128)] * (NH - 1) + [feed(MB, HID, OUT_DIM, 64, 128)],
                            "TN": [feed(HID, IN_DIM, MB, 64, 128)] + [feed(HID, HID, MB, 128, 128)] * (NH - 1) + [feed(OUT_DIM, HID, MB, 64, 128)]}.get(dom)
                if per_step and abs(len(per_step) * args.steps - d["launches"]) <= len(per_step):
                    bpl = sum(per_step) / len(per_step)
                    tbs = bpl / (d["avg_us"] * 1e-6) / 1e12
                    out["roofline"]["feed"] = {
                        "path": "L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), the operand tiles of the K loop", "bytes_per_launch": bpl,
                        "achieved": tbs, "peak": L2_PEAK_TBS, "unit": "TB/s", "frac": tbs / L2_PEAK_TBS,
                        "peak_note": "L2 aggregate ~34.5 TB/s (MI355X_MICROARCH.md; devtools/micro/lds_feed.hip reaches 33-35 TB/s with every request a hit)",
                        "in_loop_l2_busy": 0.80, "in_loop_matrix_pipe_busy": 0.49,
                        "in_loop_source": "NOT measured in this run: TCC_BUSY_sum / (128 L2 channel instances x cycles) and SQ_VALU_MFMA_BUSY_CYCLES of the committed "
                                          "passes profiles/r06_gemm_split16_pmc_*.txt, the launch's fixed ~8 us taken out",
                        "note": "`achieved` is over the WHOLE launch (fixed part and, for the weight gradients, the HBM-bound update epilogue included), like roofline.achieved"}
            tot_fl = sum(v["flop_per_launch"] * v["launches"] for v in timed.values())
            tot_ms = sum(v["avg_us"] * v["launches"] for v in timed.values()) / 1e3
            step_tf = FLOP_PER_FRAME * args.steps * MB / elapsed / 1e12
            out["gemm_all"] = {"variants": gemm, "tflops": tot_fl / tot_ms / 1e9, "frac_of_step_time": tot_ms / (elapsed * 1e3),
                               "algorithmic_tflops_whole_step": step_tf, "whole_step_frac_of_split_peak": step_tf / SPLIT_PEAK_TF_EQUIV,
                               "whole_step_ratio_to_fp32_mfma_peak": step_tf / F32_MFMA_PEAK_TFLOPS}
        else:
            out["roofline"] = {"bound": "mfma", "achieved": FLOP_PER_FRAME * args.steps * MB / elapsed / 1e12, "peak": F32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": FLOP_PER_FRAME * args.steps * MB / elapsed / 1e12 / F32_MFMA_PEAK_TFLOPS,
                               "traffic": None, "note": "whole-step algorithmic flops (per-kernel events disabled)"}
        if world == 1 and not args.headline_only:
            out["ctc_loss_fp32_rel_err"] = ctc_rel_err(aslp, dev)
        if world == 1 and not args.no_cfg3:
            net = None
            torch.cuda.empty_cache()
            out["cfg3"] = cfg3_block(aslp, dev)
            out["recurrent_layers"] = recurrent_family_block(aslp, dev)
            out["cfg1_gpu"] = cfg1_gpu_block(aslp, dev)
        if world == 1 and not args.headline_only:
            net = None
            torch.cuda.empty_cache()
            # the same step on the fp32 matrix instruction, the accuracy of every product shape on both instructions, the bandwidth-bound kernels
            out["fp32_instruction"] = fp32_instruction_block(aslp, dev, min(args.steps, 200), min(args.warmup, 50))
            out["fp32_instruction"]["vs_default"] = out["fp32_instruction"]["value"] / value
            out["product_accuracy"] = product_accuracy_block(aslp, dev)
            out["hbm_kernels"] = hbm_kernels_block(aslp, dev)
        if world == 1 and not args.no_e2e_tool:
            out["e2e_tool"] = e2e_tool_block(args.e2e_frames)
            if "frames_per_sec" in out["e2e_tool"]:
                out["e2e_tool"]["of_compute_only"] = out["e2e_tool"]["frames_per_sec"] / value
                # the same tool on three times the frames: what the run pays ONCE (first cache fill with nothing to overlap it, first-step
                # allocations, the model write: all inside the tool's timer, as in the reference) separated from what it pays per step
                longer = e2e_tool_block(3 * args.e2e_frames)
                if "frames_per_sec" in longer:
                    f1, f3 = out["e2e_tool"]["frames"], longer["frames"]
                    t1, t3 = f1 / out["e2e_tool"]["frames_per_sec"], f3 / longer["frames_per_sec"]
                    marg = (f3 - f1) / max(t3 - t1, 1e-9)
                    out["e2e_tool"]["longer_run"] = {"frames": f3, "frames_per_sec": longer["frames_per_sec"], "of_compute_only": longer["frames_per_sec"] / value}
                    out["e2e_tool"]["steady_state"] = {"frames_per_sec": marg, "of_compute_only": marg / value, "fixed_cost_s": t1 - f1 / marg,
                                                       "how": "(frames_3x - frames_1x) / (seconds_3x - seconds_1x) of the tool's own fps timer; fixed_cost_s = "
                                                              "what the 1x run takes beyond frames_1x at that rate"}
                # the REFERENCE's own aslp-nnet-train-frame.cc, compiled unchanged against the engine (kaldi-aslp_amd/bin_ref/, built where the
                # reference tree is; the binary travels): what a caller written against the reference gets without touching its source
                if os.path.exists(os.path.join(ROOT, "kaldi-aslp_amd", "bin_ref", "aslp-nnet-train-frame")):
                    refmain = e2e_tool_block(args.e2e_frames, bindir="bin_ref")
                    if "frames_per_sec" in refmain:
                        out["e2e_tool"]["reference_main_unchanged"] = {
                            "frames_per_sec": refmain["frames_per_sec"], "of_the_engines_tool": refmain["frames_per_sec"] / out["e2e_tool"]["frames_per_sec"],
                            "what": "src/aslp-nnetbin/aslp-nnet-train-frame.cc of the reference built against include/aslp_compat_kaldi.h and linked with "
                                    "libaslp_hip.so, same archives and flags, its own fps line"}
        if world == 1 and not args.no_cpu_baseline:
            port = cpu_baseline()
            ref = cpu_baseline_reference()
            out["cpu_baseline"] = ref if ref is not None else port
            if ref is not None:
                out["cpu_baseline_port"] = port   # the oracle's own C chain beside it (round 1's figure)
            # the denominator of north_star's ">= 30x the host-CPU frames/sec on a 5x2048 DNN at 1 GPU": cfg1 on the reference's CPU path, and
            # the ratio against cfg1 on the GPU (`cfg1_gpu`), both measured in this run on this box
            ref1 = cpu_baseline_reference(10.0, cfg1=True)
            if ref1 is not None:
                out["cfg1_cpu_baseline"] = ref1
                if "cfg1_gpu" in out:
                    out["cfg1_gpu"]["vs_cfg1_cpu_baseline"] = out["cfg1_gpu"]["frames_per_sec"] / ref1["value"]
                    out["cfg1_gpu"]["target"] = ">= 30x cfg1_cpu_baseline (north_star)"
    if comm is not None and not args.no_cfg3:   # N > 1 (or ASLP_BENCH_FORCE_SYNC=1): every rank takes part; rank 0 reports
        if worker is not None:
            worker.close()   # it aliases the cfg2 net's parameter tensors
            worker = None
        net = None
        torch.cuda.empty_cache()
        # The headline above is already measured: this extra block must not be able to take it down.  A watchdog ends the
        # process with the line printed if the block has not come back (a rank that died inside it leaves the others in a
        # collective); an exception on this rank is recorded instead of raised.
        import threading
        printed = threading.Lock()   # the line goes out once, whoever gets there first (watchdog thread or main thread)

        def bail():
            if not printed.acquire(blocking=False):
                return
            if rank == 0:
                out["multi_gpu_blocks"] = {"error": "cfg4_bsp / cfg3_bsp / cfg5_easgd did not finish within %d s; dropped" % args.cfg3_bsp_timeout}
                print(json.dumps(out), flush=True)
            os._exit(3)   # the headline is on stdout, but the run did not complete: non-zero

        dog = threading.Timer(args.cfg3_bsp_timeout, bail)
        dog.daemon = True
        dog.start()
        blk = {}
        extra = {}
        for key, fn in (("cfg4_bsp", cfg4_bsp_block), ("cfg3_bsp", cfg3_bsp_block), ("cfg5_easgd", cfg5_easgd_block)):
            try:
                b_ = fn(aslp, dev, native_parallel, comm, rank, world, args.sync_period)
            except Exception as e:   # noqa: BLE001 -- reported in the line, never silent
                b_ = {"error": "%s: %s" % (type(e).__name__, e)}
            extra[key] = b_
            if "error" in b_:
                blk = b_         # (the other ranks may be stuck in a collective: stop here, the watchdog ends this rank too)
                break
            torch.cuda.empty_cache()
        if "error" not in blk:
            dog.cancel()
        if rank == 0:
            out.update(extra)
        if "error" in blk:
            if printed.acquire(blocking=False):
                if rank == 0:
                    print(json.dumps(out), flush=True)
                os._exit(3)
            time.sleep(3600)   # the watchdog is printing: it ends the process
    if comm is not None:
        comm.Barrier()
        if worker is not None:
            worker.close()
        comm.close()
    if rank == 0:
        if comm is not None and not args.no_cfg3:
            dog.cancel()
            if not printed.acquire(blocking=False):
                time.sleep(3600)   # the watchdog fired at the very end and is printing
        print(json.dumps(out))   # the last thing on stdout


if __name__ == "__main__":
    main()
