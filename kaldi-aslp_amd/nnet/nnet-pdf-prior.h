// nnet-pdf-prior.h -- PdfPrior (src/aslp-nnet/nnet-pdf-prior.{h,cc}): log-priors from class frame counts, subtracted from
// the log-posteriors / pre-softmax activations that aslp-nnet-forward writes for the decoder.
#pragma once
#include <cfloat>
#include <cmath>
#include <string>

#include "cu-matrix.h"
#include "kaldi-io.h"
#include "parse-options.h"

namespace aslp {

struct PdfPriorOptions {  // nnet-pdf-prior.h:36-55
  std::string class_frame_counts;
  BaseFloat prior_scale, prior_floor;
  PdfPriorOptions() : class_frame_counts(""), prior_scale(1.0), prior_floor(1e-10) {}
  void Register(OptionsItf *opts) {
    opts->Register("class-frame-counts", &class_frame_counts,
                   "Vector with frame-counts of pdfs to compute log-priors. (priors are typically subtracted from log-posteriors or pre-softmax activations)");
    opts->Register("prior-scale", &prior_scale, "Scaling factor to be applied on pdf-log-priors");
    opts->Register("prior-floor", &prior_floor, "Flooring constatnt for prior probability (i.e. label rel. frequency)");
  }
};

class PdfPrior {
 public:
  explicit PdfPrior(const PdfPriorOptions &opts) : prior_scale_(opts.prior_scale) {  // nnet-pdf-prior.cc:27-71
    if (opts.class_frame_counts == "") return;  // deactivated (e.g. bottleneck features)
    ASLP_LOG << "Computing pdf-priors from : " << opts.class_frame_counts;
    HostVectorD frame_counts;
    {
      Input in(opts.class_frame_counts);
      frame_counts.Read(in.Stream(), false);
    }
    double sum = 0.0;
    for (double c : frame_counts.data) sum += c;
    HostVector log_priors(frame_counts.Dim());
    int32 num_floored = 0;
    double check = 0.0;
    for (int32 i = 0; i < frame_counts.Dim(); i++) {
      const double rel_freq = frame_counts.data[i] * (1.0 / sum);
      double lp = std::log(rel_freq + 1e-20);
      if (rel_freq < opts.prior_floor) { lp = std::sqrt(FLT_MAX); num_floored++; }  // zero likelihood without NaNs downstream
      log_priors.data[i] = (BaseFloat)lp;
      check += lp;
    }
    ASLP_LOG << "Floored " << num_floored << " pdf-priors (hard-set to " << std::sqrt(FLT_MAX) << ", which disables DNN output when decoding)";
    ASLP_ASSERT(std::isfinite(check));
    log_priors_ = log_priors;
  }
  void SubtractOnLogpost(CuMatrixBase *llk) {  // :74-86
    if (log_priors_.Dim() == 0) ASLP_ERR << "--class-frame-counts is empty: Cannot initialize priors without the counts.";
    if (log_priors_.Dim() != llk->NumCols())
      ASLP_ERR << "Dimensionality mismatch, class_frame_counts " << log_priors_.Dim() << " pdf_output_llk " << llk->NumCols();
    llk->AddVecToRows(-prior_scale_, log_priors_);
  }

 private:
  BaseFloat prior_scale_;
  CuVector log_priors_;
};

}  // namespace aslp
