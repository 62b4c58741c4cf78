// aslp-nnet-copy -- src/aslp-nnetbin/aslp-nnet-copy.cc: read a model (text or binary), write it (text or binary).
#include "nnet-nnet.h"
#include "parse-options.h"

int main(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Initialize Neural Network parameters according to a prototype (aslp_nnet).\n"
        "Usage:  aslp-nnet-copy [options] <nnet-in> <nnet-out>\n"
        "e.g.:\n"
        " aslp-nnet-copy --binary=false nnet.in nnet.out\n";
    g_verbose_level = 1;
    ParseOptions po(usage);
    bool binary_write = true;
    po.Register("binary", &binary_write, "Write output in binary mode");
    int32 seed = 777;
    po.Register("seed", &seed, "Seed for random number generator");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) { po.PrintUsage(); exit(1); }
    std::string nnet_in_filename = po.GetArg(1), nnet_out_filename = po.GetArg(2);
    Nnet nnet;
    nnet.Read(nnet_in_filename);
    nnet.Write(nnet_out_filename, binary_write);
    ASLP_LOG << "Written model to " << nnet_out_filename;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
