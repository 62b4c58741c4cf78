// capi.cpp -- C handle API (include/aslp_nnet.h) over Nnet / Xent.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <utility>
#include <vector>

#include "aslp_nnet.h"
#include "ctc-loss.h"
#include "nnet-loss.h"
#include "nnet-nnet.h"
#include "nnet-randomizer.h"
#include "warp-ctc.h"

using namespace aslp;

struct aslp_nnet_s {
  Nnet nnet;
  CuMatrix out, diff, in_diff;
  CuVector fw;
};
struct aslp_xent_s {
  Xent xent;
};
struct aslp_matrix_randomizer_s {
  ~aslp_matrix_randomizer_s() { ReleaseStaged(); }
  void ReleaseStaged() {
    for (auto &b : staged) { b.second->Wait(); PinnedFree(b.first); }
    staged.clear();
  }
  std::vector<std::pair<float *, std::unique_ptr<StreamMarker>>> staged;  // pinned copies of stage_add()'s rows, until the lane has sent them
  MatrixRandomizer r;
};
struct aslp_eesenctc_s {
  Ctc ctc;
  CuMatrix diff;
};
struct aslp_warpctc_s {
  WarpCtc ctc;
  CuMatrix diff;
};
static void SplitLabels(const int32_t *flat, const int32_t *lens, int n, std::vector<std::vector<int32>> *out) {
  out->resize(n);
  int off = 0;
  for (int i = 0; i < n; i++) {
    (*out)[i].assign(flat + off, flat + off + lens[i]);
    off += lens[i];
  }
}

static thread_local std::string t_err;
#define API_BEGIN try {
#define API_END                      \
  }                                  \
  catch (const std::exception &e) {  \
    t_err = e.what();                \
    return 1;                        \
  }                                  \
  return 0;

static int CopyStr(const std::string &s, char *buf, int n) {
  if (!buf || n <= 0) return (int)s.size();
  std::strncpy(buf, s.c_str(), n - 1);
  buf[n - 1] = 0;
  return (int)s.size();
}

extern "C" {

const char *aslp_nnet_last_error(void) { return t_err.c_str(); }
void aslp_set_verbose(int level) { g_verbose_level = level; }

int aslp_nnet_init_from_proto(const char *proto_text, unsigned seed, aslp_nnet_t *out) {
  API_BEGIN
  SRand(seed);
  aslp_nnet_s *h = new aslp_nnet_s();
  try { h->nnet.InitFromString(proto_text); } catch (...) { delete h; throw; }
  *out = h;
  API_END
}
int aslp_nnet_read(const char *path, aslp_nnet_t *out) {
  API_BEGIN
  aslp_nnet_s *h = new aslp_nnet_s();
  try { h->nnet.Read(path); } catch (...) { delete h; throw; }
  *out = h;
  API_END
}
int aslp_nnet_write(aslp_nnet_t n, const char *path, int binary) { API_BEGIN n->nnet.Write(path, binary != 0); API_END }
int aslp_nnet_copy(aslp_nnet_t n, aslp_nnet_t *out) {
  API_BEGIN
  aslp_nnet_s *h = new aslp_nnet_s();
  try { h->nnet = n->nnet; } catch (...) { delete h; throw; }
  *out = h;
  API_END
}
void aslp_nnet_free(aslp_nnet_t n) { delete n; }

int aslp_nnet_set_train_options(aslp_nnet_t n, float lr, float mmt, float l2, float l1) {
  API_BEGIN
  NnetTrainOptions o;
  o.learn_rate = lr; o.momentum = mmt; o.l2_penalty = l2; o.l1_penalty = l1;
  n->nnet.SetTrainOptions(o);
  API_END
}
int aslp_nnet_input_dim(aslp_nnet_t n) { try { return n->nnet.InputDim(); } catch (...) { return -1; } }
int aslp_nnet_output_dim(aslp_nnet_t n) { try { return n->nnet.OutputDim(); } catch (...) { return -1; } }
int aslp_nnet_num_components(aslp_nnet_t n) { return n->nnet.NumComponents(); }
int aslp_nnet_num_params(aslp_nnet_t n) { try { return n->nnet.NumParams(); } catch (...) { return -1; } }
int aslp_nnet_component_marker(aslp_nnet_t n, int c, char *buf, int buflen) {
  API_BEGIN CopyStr(Component::TypeToMarker(n->nnet.GetComponent(c).GetType()), buf, buflen); API_END
}
int aslp_nnet_info(aslp_nnet_t n, char *buf, int buflen) { API_BEGIN CopyStr(n->nnet.Info(), buf, buflen); API_END }
int aslp_nnet_set_link_aliasing(aslp_nnet_t n, int on) { API_BEGIN n->nnet.SetLinkAliasing(on != 0); API_END }
int aslp_nnet_set_layer_fusion(aslp_nnet_t n, int on) { API_BEGIN n->nnet.SetLayerFusion(on != 0); API_END }
int aslp_nnet_set_update_overlap(aslp_nnet_t n, int on) { API_BEGIN n->nnet.SetUpdateOverlap(on != 0); API_END }

int aslp_nnet_propagate(aslp_nnet_t n, const float *in, int rows, int cols, int stride, float *out, int out_stride) {
  API_BEGIN
  CuSubMatrix inm(const_cast<float *>(in), rows, cols, stride);
  n->nnet.Propagate(inm, &n->out);
  if (out) {
    CuSubMatrix o(out, n->out.NumRows(), n->out.NumCols(), out_stride);
    o.CopyFromMat(n->out);
  }
  API_END
}
int aslp_nnet_feedforward(aslp_nnet_t n, const float *in, int rows, int cols, int stride, float *out, int out_stride) {
  API_BEGIN
  CuSubMatrix inm(const_cast<float *>(in), rows, cols, stride);
  n->nnet.Feedforward(inm, &n->out);
  if (out) {
    CuSubMatrix o(out, n->out.NumRows(), n->out.NumCols(), out_stride);
    o.CopyFromMat(n->out);
  }
  API_END
}
int aslp_nnet_backpropagate(aslp_nnet_t n, const float *out_diff, int rows, int cols, int stride, float *in_diff, int in_diff_stride) {
  API_BEGIN
  CuSubMatrix od(const_cast<float *>(out_diff), rows, cols, stride);
  if (in_diff) {
    n->nnet.Backpropagate(od, &n->in_diff);
    CuSubMatrix o(in_diff, n->in_diff.NumRows(), n->in_diff.NumCols(), in_diff_stride);
    o.CopyFromMat(n->in_diff);
  } else {
    n->nnet.Backpropagate(od, NULL);
  }
  API_END
}
int aslp_nnet_reset_lstm_streams(aslp_nnet_t n, const int32_t *flags, int ns) {
  API_BEGIN n->nnet.ResetLstmStreams(std::vector<int32>(flags, flags + ns)); API_END
}
int aslp_nnet_set_seq_lengths(aslp_nnet_t n, const int32_t *lens, int ns) {
  API_BEGIN n->nnet.SetSeqLengths(std::vector<int32>(lens, lens + ns)); API_END
}
int aslp_nnet_set_chunk_size(aslp_nnet_t n, int chunk_size) { API_BEGIN n->nnet.SetChunkSize(chunk_size); API_END }

int aslp_nnet_get_params(aslp_nnet_t n, float *host_buf, int buf_len) {
  API_BEGIN
  std::vector<BaseFloat> w;
  n->nnet.GetParams(&w);
  if ((int)w.size() > buf_len) ASLP_ERR << "buffer too small: need " << w.size();
  std::memcpy(host_buf, w.data(), sizeof(float) * w.size());
  API_END
}
int aslp_nnet_get_gpu_params(aslp_nnet_t n, float **ptrs, int *sizes, int max_n) {
  try {
    std::vector<std::pair<BaseFloat *, int>> p;
    n->nnet.GetGpuParams(&p);
    for (int i = 0; i < (int)p.size() && i < max_n; i++) { ptrs[i] = p[i].first; sizes[i] = p[i].second; }
    return (int)p.size();
  } catch (const std::exception &e) { t_err = e.what(); return -1; }
}
int aslp_nnet_param_writers_announce(aslp_nnet_t n) { API_BEGIN n->nnet.ParamWritersAnnounce(); API_END }
int aslp_nnet_get_acc_stats(aslp_nnet_t n, double **dev_ptrs, int *sizes, int max_n, double **counts_host, int max_bn, int *num_bn) {
  try {
    std::vector<double *> acc;
    std::vector<std::pair<double *, int>> data;
    n->nnet.GetAccStats(&acc, &data);
    for (int i = 0; i < (int)data.size() && i < max_n; i++) { dev_ptrs[i] = data[i].first; sizes[i] = data[i].second; }
    for (int i = 0; i < (int)acc.size() && i < max_bn; i++) counts_host[i] = acc[i];
    if (num_bn) *num_bn = (int)acc.size();
    return (int)data.size();
  } catch (const std::exception &e) { t_err = e.what(); return -1; }
}
int aslp_nnet_component_output(aslp_nnet_t n, int c, float *host_dst, int rows, int cols) {
  API_BEGIN
  const CuMatrixBase &m = n->nnet.OutputBuffer(c);
  if (m.NumRows() != rows || m.NumCols() != cols) ASLP_ERR << "component " << c << " output is " << m.NumRows() << " x " << m.NumCols();
  m.CopyToHost(host_dst, cols);
  API_END
}
int aslp_nnet_component_out_diff(aslp_nnet_t n, int c, float *host_dst, int rows, int cols) {
  API_BEGIN
  const CuMatrixBase &m = n->nnet.OutputDiffBuffer(c);
  if (m.NumRows() != rows || m.NumCols() != cols) ASLP_ERR << "component " << c << " out-diff is " << m.NumRows() << " x " << m.NumCols();
  m.CopyToHost(host_dst, cols);
  API_END
}

int aslp_xent_create(aslp_xent_t *out) { API_BEGIN *out = new aslp_xent_s(); API_END }
void aslp_xent_free(aslp_xent_t x) { delete x; }
int aslp_xent_eval_batch(aslp_xent_t x, const float *net_out, int rows, int cols, int stride, const float *targets, int tgt_stride,
                   const int32_t *labels, const float *frame_weights, float *diff, int diff_stride) {
  API_BEGIN
  // thin path straight onto the fused kernel; statistics accumulate inside x
  CuSubMatrix y(const_cast<float *>(net_out), rows, cols, stride);
  CuMatrix d;
  if (targets) {
    CuSubMatrix t(const_cast<float *>(targets), rows, cols, tgt_stride);
    std::vector<float> fw(rows);
    DeviceToHost(fw.data(), frame_weights, sizeof(float) * rows);
    x->xent.Eval(fw, y, t, &d);
  } else {
    CuSubVector fwv(const_cast<float *>(frame_weights), rows);
    CuArray<int32> lab;
    lab.Resize(rows);
    DeviceToDevice(lab.Data(), labels, sizeof(int32) * rows);
    x->xent.EvalLabels(fwv, y, lab, &d);
  }
  CuSubMatrix o(diff, rows, cols, diff_stride);
  o.CopyFromMat(d);
  API_END
}
int aslp_xent_report(aslp_xent_t x, char *buf, int buflen) { API_BEGIN CopyStr(x->xent.Report(), buf, buflen); API_END }
int aslp_xent_get_stats(aslp_xent_t x, double stats[5]) { API_BEGIN x->xent.GetStats(stats); API_END }

int aslp_nnet_train_step_xent(aslp_nnet_t n, aslp_xent_t x, const float *in, int rows, int cols, int stride, const int32_t *labels,
                              const float *frame_weights) {
  API_BEGIN
  CuSubMatrix inm(const_cast<float *>(in), rows, cols, stride);
  n->nnet.PropagateForLoss(inm, true);
  float fw_max = -1.0f;   // unknown for weights given in device memory
  if (frame_weights == NULL) {
    if (n->fw.Dim() != rows) { n->fw.Resize(rows, kUndefined); n->fw.Set(1.0f); }
    frame_weights = n->fw.Data();
    fw_max = 1.0f;
  }
  CuSubVector fwv(const_cast<float *>(frame_weights), rows);
  if (n->nnet.LossInputIsPreSoftmax()) x->xent.EvalLabelsPreSoftmax(fwv, n->nnet.LossInput(), labels, n->nnet.LossDiff(rows), fw_max);
  else x->xent.EvalLabels(fwv, n->nnet.LossInput(), labels, n->nnet.LossDiff(rows), fw_max);
  n->nnet.BackpropagateFromLossDiff();
  API_END
}

int aslp_warpctc_create(aslp_warpctc_t *out) { API_BEGIN *out = new aslp_warpctc_s(); API_END }
void aslp_warpctc_free(aslp_warpctc_t w) { delete w; }
int aslp_warpctc_eval(aslp_warpctc_t w, const int32_t *frame_num_utt, int num_utt, const float *net_out, int rows, int cols, int stride,
                      const int32_t *flat_labels, const int32_t *label_lengths, float *diff, int diff_stride, float *costs_host) {
  API_BEGIN
  CuSubMatrix y(const_cast<float *>(net_out), rows, cols, stride);
  std::vector<int32> frames(frame_num_utt, frame_num_utt + num_utt);
  std::vector<std::vector<int32>> labels;
  SplitLabels(flat_labels, label_lengths, num_utt, &labels);
  std::vector<std::string> utt(num_utt);
  for (int i = 0; i < num_utt; i++) utt[i] = "utt" + std::to_string(i);
  w->ctc.Eval(utt, frames, y, labels, &w->diff);
  CuSubMatrix o(diff, rows, cols, diff_stride);
  o.CopyFromMat(w->diff);
  if (costs_host) std::memcpy(costs_host, w->ctc.LastCosts().data(), sizeof(float) * num_utt);
  API_END
}
int aslp_warpctc_error_rate(aslp_warpctc_t w, const int32_t *frame_num_utt, int num_utt, const float *net_out, int rows, int cols, int stride,
                            const int32_t *flat_labels, const int32_t *label_lengths) {
  API_BEGIN
  CuSubMatrix y(const_cast<float *>(net_out), rows, cols, stride);
  std::vector<int32> frames(frame_num_utt, frame_num_utt + num_utt);
  std::vector<std::vector<int32>> labels;
  SplitLabels(flat_labels, label_lengths, num_utt, &labels);
  w->ctc.ErrorRate(frames, y, labels);
  API_END
}
int aslp_warpctc_report(aslp_warpctc_t w, char *buf, int buflen) { API_BEGIN CopyStr(w->ctc.Report(), buf, buflen); API_END }
int aslp_warpctc_get_stats(aslp_warpctc_t w, double stats[5]) {
  API_BEGIN
  stats[0] = w->ctc.Obj(); stats[1] = w->ctc.Frames(); stats[2] = w->ctc.Sequences();
  stats[3] = w->ctc.NumErrorTokens(); stats[4] = w->ctc.NumRefTokens();
  API_END
}
int aslp_nnet_train_step_warpctc(aslp_nnet_t n, aslp_warpctc_t w, const float *in, int rows, int cols, int stride,
                                 const int32_t *frame_num_utt, int num_utt, const int32_t *flat_labels, const int32_t *label_lengths) {
  API_BEGIN
  CuSubMatrix inm(const_cast<float *>(in), rows, cols, stride);
  std::vector<int32> frames(frame_num_utt, frame_num_utt + num_utt);
  std::vector<std::vector<int32>> labels;
  SplitLabels(flat_labels, label_lengths, num_utt, &labels);
  std::vector<std::string> utt(num_utt);
  for (int i = 0; i < num_utt; i++) utt[i] = "utt" + std::to_string(i);
  n->nnet.SetSeqLengths(frames);  // aslp-nnet-train-warp-ctc-streams.cc: whole-utterance batches
  n->nnet.Propagate(inm, &n->out);
  w->ctc.Eval(utt, frames, n->out, labels, &n->diff);
  w->ctc.ErrorRate(frames, n->out, labels);
  n->nnet.Backpropagate(n->diff, NULL);
  API_END
}

int aslp_eesenctc_create(aslp_eesenctc_t *out) { API_BEGIN *out = new aslp_eesenctc_s(); API_END }
void aslp_eesenctc_free(aslp_eesenctc_t c) { delete c; }
int aslp_eesenctc_eval(aslp_eesenctc_t c, const int32_t *frame_num_utt, int num_utt, const float *net_out, int rows, int cols, int stride,
                       const int32_t *flat_labels, const int32_t *label_lengths, float *diff, int diff_stride, float *costs_host) {
  API_BEGIN
  CuSubMatrix y(const_cast<float *>(net_out), rows, cols, stride);
  std::vector<std::vector<int32>> labels;
  SplitLabels(flat_labels, label_lengths, num_utt, &labels);
  if (frame_num_utt == NULL) {
    ASLP_ASSERT(num_utt == 1);
    c->ctc.Eval(y, labels[0], &c->diff);
  } else {
    std::vector<int32> frames(frame_num_utt, frame_num_utt + num_utt);
    std::vector<std::string> utt(num_utt);
    for (int i = 0; i < num_utt; i++) utt[i] = "utt" + std::to_string(i);
    c->ctc.EvalParallel(utt, frames, y, labels, &c->diff);
  }
  CuSubMatrix o(diff, rows, cols, diff_stride);
  o.CopyFromMat(c->diff);
  if (costs_host) std::memcpy(costs_host, c->ctc.LastCosts().data(), sizeof(float) * num_utt);
  API_END
}
int aslp_eesenctc_error_rate(aslp_eesenctc_t c, const int32_t *frame_num_utt, int num_utt, const float *net_out, int rows, int cols, int stride,
                             const int32_t *flat_labels, const int32_t *label_lengths) {
  API_BEGIN
  CuSubMatrix y(const_cast<float *>(net_out), rows, cols, stride);
  std::vector<std::vector<int32>> labels;
  SplitLabels(flat_labels, label_lengths, num_utt, &labels);
  if (frame_num_utt == NULL) {
    float er;
    std::vector<int32> hyp;
    c->ctc.ErrorRate(y, labels[0], &er, &hyp);
  } else {
    std::vector<int32> frames(frame_num_utt, frame_num_utt + num_utt);
    c->ctc.ErrorRateMSeq(frames, y, labels);
  }
  API_END
}
int aslp_eesenctc_report(aslp_eesenctc_t c, char *buf, int buflen) { API_BEGIN CopyStr(c->ctc.Report(), buf, buflen); API_END }
int aslp_eesenctc_get_stats(aslp_eesenctc_t c, double stats[5]) {
  API_BEGIN
  stats[0] = c->ctc.Obj(); stats[1] = c->ctc.Frames(); stats[2] = c->ctc.Sequences();
  stats[3] = c->ctc.NumErrorTokens(); stats[4] = c->ctc.NumRefTokens();
  API_END
}

int aslp_randomizer_mask_generate(int seed, int size, int32_t *mask_host) {
  API_BEGIN
  RandomizerMask m;
  if (seed >= 0) SRand(seed);
  const std::vector<int32> &v = m.Generate(size);
  std::memcpy(mask_host, v.data(), sizeof(int32) * size);
  API_END
}
int aslp_matrix_randomizer_create(int randomizer_size, int minibatch_size, aslp_matrix_randomizer_t *out) {
  API_BEGIN
  NnetDataRandomizerOptions c;
  c.randomizer_size = randomizer_size;
  c.minibatch_size = minibatch_size;
  aslp_matrix_randomizer_s *h = new aslp_matrix_randomizer_s();
  h->r.Init(c);
  *out = h;
  API_END
}
void aslp_matrix_randomizer_free(aslp_matrix_randomizer_t r) { delete r; }
int aslp_matrix_randomizer_add_data(aslp_matrix_randomizer_t r, const float *dev, int rows, int cols, int stride) {
  API_BEGIN
  r->r.AddData(CuSubMatrix(const_cast<float *>(dev), rows, cols, stride));
  API_END
}
int aslp_matrix_randomizer_stage_begin(aslp_matrix_randomizer_t r) { API_BEGIN r->r.StageBegin(); API_END }
int aslp_matrix_randomizer_stage_add(aslp_matrix_randomizer_t r, const float *host, int rows, int cols) {
  API_BEGIN
  float *blk = static_cast<float *>(PinnedAlloc(sizeof(float) * (size_t)rows * cols));
  std::memcpy(blk, host, sizeof(float) * (size_t)rows * cols);
  std::unique_ptr<StreamMarker> done(new StreamMarker);
  r->r.StageAddPinned(blk, rows, cols, done.get());
  r->staged.push_back(std::make_pair(blk, std::move(done)));
  API_END
}
int aslp_matrix_randomizer_stage_commit(aslp_matrix_randomizer_t r) {
  API_BEGIN
  r->r.StageCommit();
  r->ReleaseStaged();
  API_END
}
int aslp_matrix_randomizer_stage_state(aslp_matrix_randomizer_t r, int state[2]) {
  API_BEGIN
  state[0] = r->r.StageFull(); state[1] = r->r.StageFrames();
  API_END
}
int aslp_matrix_randomizer_randomize(aslp_matrix_randomizer_t r, const int32_t *mask_host, int n) {
  API_BEGIN
  r->r.Randomize(std::vector<int32>(mask_host, mask_host + n));
  API_END
}
int aslp_matrix_randomizer_next(aslp_matrix_randomizer_t r) { API_BEGIN r->r.Next(); API_END }
int aslp_matrix_randomizer_state(aslp_matrix_randomizer_t r, int state[3]) {
  API_BEGIN
  state[0] = r->r.IsFull(); state[1] = r->r.Done(); state[2] = r->r.NumFrames();
  API_END
}
int aslp_matrix_randomizer_value(aslp_matrix_randomizer_t r, const float **dev, int *rows, int *cols, int *stride) {
  API_BEGIN
  const CuMatrixBase &m = r->r.Value();
  *dev = m.Data(); *rows = m.NumRows(); *cols = m.NumCols(); *stride = m.Stride();
  API_END
}

}  // extern "C"
