// model_tools.cpp -- the model-file tools of src/aslp-nnetbin (init, copy, info, dot, convert-to-standard, insert): one entry
// function per tool (Main_<tool name with _ for ->), linked behind tools/main_stub.cpp into bin/<tool name>.
#include <cmath>
#include <fstream>

#include "kaldi-io.h"
#include "nnet-basic.h"
#include "nnet-nnet.h"
#include "parse-options.h"

// ======================================================================================================================
// aslp-nnet-init -- src/aslp-nnetbin/aslp-nnet-init.cc: <NnetProto> -> initialised model (libc rand seeded with --seed).
int Main_aslp_nnet_init(int argc, char *argv[]) {
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

// ======================================================================================================================
// aslp-nnet-copy -- src/aslp-nnetbin/aslp-nnet-copy.cc: read a model (text or binary), write it (text or binary).
int Main_aslp_nnet_copy(int argc, char *argv[]) {
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

// ======================================================================================================================
// aslp-nnet-info -- src/aslp-nnetbin/aslp-nnet-info.cc: topology and weight statistics to stdout.
int Main_aslp_nnet_info(int argc, char *argv[]) {
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

// ======================================================================================================================
// aslp-nnet-dot -- src/aslp-nnetbin/aslp-nnet-dot.cc: the component graph as a Graphviz file (Nnet::WriteDotFile).
int Main_aslp_nnet_dot(int argc, char *argv[]) {
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

// ======================================================================================================================
// aslp-nnet-convert-to-standard -- src/aslp-nnetbin/aslp-nnet-convert-to-standard.cc: drops the ASLP graph header fields
// (and the Input / Output layers) so that plain Kaldi nnet1 tools can read the model (Nnet::WriteStandard).
int Main_aslp_nnet_convert_to_standard(int argc, char *argv[]) {
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

// ======================================================================================================================
// aslp-nnet-insert -- src/aslp-nnetbin/aslp-nnet-insert.cc: layer-wise pretraining helper.  Inserts the hidden components of
// a second net (its Input / Output layers dropped) before a given component -- by default before the last updatable one --
// and, unless told otherwise, re-randomizes the AffineTransform that follows them (stddev-factor / sqrt(input dim)).
namespace aslp {
static int32 IndexOfLastUpdatableComponent(const Nnet &nnet) {  // :20-31
  int32 index = -1;
  for (int32 c = 0; c < nnet.NumComponents(); c++)
    if (nnet.GetComponent(c).IsUpdatable()) index = c;
  return index;
}
static void InsertComponents(const Nnet &src_nnet, int32 c_to_insert, Nnet *dest_nnet) {  // :33-52
  ASLP_ASSERT(c_to_insert >= 0 && c_to_insert <= dest_nnet->NumComponents());
  const int32 c_tot = dest_nnet->NumComponents() + src_nnet.NumComponents() - 2;
  std::vector<Component *> components(c_tot);
  for (int32 c = 0; c < c_to_insert; c++) components[c] = dest_nnet->GetComponent(c).Copy();
  for (int32 c = 0; c < src_nnet.NumComponents() - 2; c++) components[c + c_to_insert] = src_nnet.GetComponent(c + 1).Copy();
  for (int32 c = c_to_insert; c < dest_nnet->NumComponents(); c++)
    components[c + src_nnet.NumComponents() - 2] = dest_nnet->GetComponent(c).Copy();
  dest_nnet->Destroy();
  for (size_t c = 0; c < components.size(); c++) dest_nnet->AppendComponent(components[c]);
  dest_nnet->Check();
}
}  // namespace aslp

int Main_aslp_nnet_insert(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Insert components into a neural network-based acoustic model.\n"
        "Usage:  aslp-nnet-insert [options] <model-in1> <...> <model-inN> <model-out>\n"
        "e.g.:\n"
        " aslp-nnet-insert 1.nnet \"aslp-nnet-init hidden_layer.config -| \" 2.nnet\n";
    ParseOptions po(usage);
    bool binary_write = true, randomize_next_component = true;
    int32 insert_at = -1, insert_offset = 0, srand_seed = 0;
    BaseFloat stddev_factor = 0.1;
    po.Register("binary", &binary_write, "Write output in binary mode");
    po.Register("randomize-next-component", &randomize_next_component,
                "If true, randomize the parameters of the next component after what we insert (which must be updatable).");
    po.Register("insert-at", &insert_at, "Inserts new components before the specified component (note: indexes are zero-based).  If <0, "
                "inserts before the last updatable component(typically before the softmax).");
    po.Register("insert_offset", &insert_offset, "if insert-at = -1, assume the ID of the last updatable component"
                "(typically before the softmax) is k, inserts before the k-offset component.");
    po.Register("stddev-factor", &stddev_factor, "Factor on the standard deviation when randomizing next component (only relevant if "
                "--randomize-next-component=true");
    po.Register("srand", &srand_seed, "Seed for random number generator");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) { po.PrintUsage(); exit(1); }
    std::string nnet_rxfilename = po.GetArg(1), raw_nnet_rxfilename = po.GetArg(2), nnet_wxfilename = po.GetArg(3);
    Nnet nnet, src_nnet;
    nnet.Read(nnet_rxfilename);
    src_nnet.Read(raw_nnet_rxfilename);
    if (insert_at == -1) {
      if ((insert_at = IndexOfLastUpdatableComponent(nnet)) == -1)
        ASLP_ERR << "We don't know where to insert the new components: the neural net doesn't have exactly one softmax component, "
                    "and you didn't use the --insert-at option.";
      insert_at = insert_at - insert_offset;
    }
    InsertComponents(src_nnet, insert_at, &nnet);
    ASLP_LOG << "Inserted " << src_nnet.NumComponents() - 2 << " components at " << "position " << insert_at;
    if (randomize_next_component) {
      const int32 c = insert_at + src_nnet.NumComponents() - 2;
      AffineTransform *uc = dynamic_cast<AffineTransform *>(&nnet.GetComponent(c));
      if (!uc) ASLP_ERR << "You have --randomize-next-component=true, but the component to randomize is not updatable: " << nnet.GetComponent(c).Info();
      const int32 out_dim = uc->OutputDim(), in_dim = uc->InputDim();
      const BaseFloat stddev = stddev_factor / std::sqrt(static_cast<BaseFloat>(in_dim));
      // the reference draws these on the device (CuRand::RandGaussian, seeded elsewhere); here: the engine's seeded host
      // generator -- a different stream of normals with the same distribution
      SRand(srand_seed);
      HostMatrix w(out_dim, in_dim);
      for (float &v : w.data) v = stddev * RandGauss();
      HostVector b(out_dim);
      for (float &v : b.data) v = stddev * RandGauss();
      CuMatrix cw;
      cw = w;
      CuVector cb;
      cb = b;
      uc->SetLinearity(cw);
      uc->SetBias(cb);
      ASLP_LOG << "Randomized component index " << c << " with stddev " << stddev;
    }
    nnet.Write(nnet_wxfilename, binary_write);
    ASLP_LOG << "Write neural-net acoustic model to " << nnet_wxfilename;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
