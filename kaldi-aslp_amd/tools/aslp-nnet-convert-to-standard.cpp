// aslp-nnet-convert-to-standard -- src/aslp-nnetbin/aslp-nnet-convert-to-standard.cc: drops the ASLP graph header fields
// (and the Input / Output layers) so that plain Kaldi nnet1 tools can read the model (Nnet::WriteStandard).
#include "kaldi-io.h"
#include "nnet-nnet.h"
#include "parse-options.h"

int main(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Convert aslp nnet to standard kaldi nnet1\n"
        "Usage:  aslp-nnet-convert-to-standard [options] <nnet-in> <nnet-out>\n"
        "e.g.:\n"
        " aslp-nnet-convert-to-standard --binary=false nnet.in nnet.out\n";
    g_verbose_level = 1;
    ParseOptions po(usage);
    bool binary_write = true;
    po.Register("binary", &binary_write, "Write output in binary mode");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) { po.PrintUsage(); exit(1); }
    std::string nnet_in_filename = po.GetArg(1), nnet_out_filename = po.GetArg(2);
    Nnet nnet;
    nnet.Read(nnet_in_filename);
    {
      Output ko(nnet_out_filename, binary_write);
      nnet.WriteStandard(ko.Stream(), binary_write);
    }
    ASLP_LOG << "Written model to " << nnet_out_filename;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
