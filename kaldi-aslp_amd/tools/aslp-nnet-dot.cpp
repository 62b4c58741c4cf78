// aslp-nnet-dot -- src/aslp-nnetbin/aslp-nnet-dot.cc: the component graph as a Graphviz file (Nnet::WriteDotFile).
#include <fstream>

#include "nnet-nnet.h"
#include "parse-options.h"

int main(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Generate dot file about the neural network.\n"
        "Usage:  aslp-nnet-generate-graph [options] <nnet-in> <dot-out>\n"
        "e.g.:\n"
        " aslp-nnet-info 1.nnet 1.dot\n";
    ParseOptions po(usage);
    po.Read(argc, argv);
    if (po.NumArgs() != 2) { po.PrintUsage(); exit(1); }
    std::string nnet_rxfilename = po.GetArg(1), dot_wxfilename = po.GetArg(2);
    Nnet nnet;
    nnet.Read(nnet_rxfilename);
    std::ofstream ko(dot_wxfilename.c_str());
    nnet.WriteDotFile(ko);
    ASLP_LOG << "Generate dot file for " << nnet_rxfilename;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
