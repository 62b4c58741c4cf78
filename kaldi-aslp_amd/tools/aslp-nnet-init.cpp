// aslp-nnet-init -- src/aslp-nnetbin/aslp-nnet-init.cc: <NnetProto> -> initialised model (libc rand seeded with --seed).
#include "kaldi-io.h"
#include "nnet-nnet.h"
#include "parse-options.h"

int main(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Initialize Neural Network parameters according to a prototype (aslp_nnet).\n"
        "Usage:  aslp-nnet-initialize [options] <nnet-prototype-in> <nnet-out>\n"
        "e.g.:\n"
        " aslp-nnet-initialize --binary=false nnet.proto nnet.init\n";
    g_verbose_level = 1;  // be verbose by default
    ParseOptions po(usage);
    bool binary_write = true;
    po.Register("binary", &binary_write, "Write output in binary mode");
    int32 seed = 777;
    po.Register("seed", &seed, "Seed for random number generator");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) { po.PrintUsage(); exit(1); }
    std::string nnet_config_in_filename = po.GetArg(1), nnet_out_filename = po.GetArg(2);
    SRand(seed);  // the engine's private copy of the libc generator (base.h): same sequence as srand(seed); rand()
    Nnet nnet;
    nnet.Init(nnet_config_in_filename);
    nnet.Write(nnet_out_filename, binary_write);
    ASLP_LOG << "Written initialized model to " << nnet_out_filename;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
