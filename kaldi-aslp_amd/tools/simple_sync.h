// simple_sync.h -- the seam between the train-simple loop (frame_tools.cpp) and a model synchroniser that rides in it:
// aslp-nnet-train-simple runs with none, aslp-nnet-train-simple-mpi (tools/parallel/worker_tools.cpp) with the two-rank pairwise
// averaging of src/aslp-parallel/nnet-mpi-sync.cc.
#pragma once
#include <string>

#include "nnet-nnet.h"
#include "parse-options.h"

struct SimpleSync {
  virtual ~SimpleSync() {}
  virtual void Register(aslp::ParseOptions *po) = 0;                        // its command-line options
  virtual void Connect() = 0;                                               // after the options are read, before the model exists
  virtual void Init(aslp::Nnet *nnet, std::string *feature_rspecifier) = 0; // model read; may rewrite the feature table (JOB -> rank)
  virtual void AfterMinibatch() = 0;
  virtual void Finish() = 0;                                                // this rank's data is exhausted
  virtual bool WritesModel() const = 0;
};

// the body of aslp-nnet-train-simple; sync may be NULL
int TrainSimpleWithSync(int argc, char *argv[], SimpleSync *sync);
