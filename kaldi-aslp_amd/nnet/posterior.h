// posterior.h -- hmm/posterior.h's Posterior type and its stream format (hmm/posterior.cc:29-125).  Host-only.
#pragma once
#include <iostream>
#include <utility>
#include <vector>

#include "base.h"

namespace aslp {

// one vector of (pdf-id, weight) pairs per frame
typedef std::vector<std::vector<std::pair<int32, BaseFloat>>> Posterior;

void WritePosterior(std::ostream &os, bool binary, const Posterior &post);
void ReadPosterior(std::istream &is, bool binary, Posterior *post);
// ali-to-post: one (id, 1.0) pair per frame (hmm/posterior.cc AlignmentToPosterior)
void AlignmentToPosterior(const std::vector<int32> &ali, Posterior *post);

}  // namespace aslp
