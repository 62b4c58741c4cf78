// aslp-nnet-info -- src/aslp-nnetbin/aslp-nnet-info.cc: topology and weight statistics to stdout.
#include "nnet-nnet.h"
#include "parse-options.h"

int main(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Print human-readable information about the neural network.\n"
        "(topology, various weight statistics, etc.) It prints to stdout.\n"
        "Usage:  aslp-nnet-info [options] <nnet-in>\n"
        "e.g.:\n"
        " aslp-nnet-info 1.nnet\n";
    ParseOptions po(usage);
    po.Read(argc, argv);
    if (po.NumArgs() != 1) { po.PrintUsage(); exit(1); }
    std::string nnet_rxfilename = po.GetArg(1);
    Nnet nnet;
    nnet.Read(nnet_rxfilename);
    std::cout << nnet.Info();
    ASLP_LOG << "Printed info about " << nnet_rxfilename;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
