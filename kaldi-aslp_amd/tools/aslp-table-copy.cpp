// aslp-table-copy -- copies a Kaldi table from an rspecifier to a wspecifier (what copy-feats / copy-vector / copy-post /
// copy-int-vector / ali-to-post do for their types), through the same readers and writers the training tools use.
// Host-only: runs without a GPU; the CPU tests use it to check every stream format against independently written files.
#include "kaldi-table.h"
#include "parse-options.h"

using namespace aslp;

template <class Holder>
static int Copy(const std::string &rspecifier, const std::string &wspecifier, bool random_access) {
  TableWriter<Holder> writer(wspecifier);
  int n = 0;
  if (!random_access) {
    SequentialTableReader<Holder> reader(rspecifier);
    for (; !reader.Done(); reader.Next(), n++) writer.Write(reader.Key(), reader.Value());
  } else {  // same result, but every object is fetched by key through the random-access reader
    std::vector<std::string> keys;
    {
      SequentialTableReader<Holder> reader(rspecifier);
      for (; !reader.Done(); reader.Next()) keys.push_back(reader.Key());
    }
    RandomAccessTableReader<Holder> ra(rspecifier);
    RspecifierOptions opts;
    ClassifyRspecifier(rspecifier, NULL, &opts);
    if (!opts.called_sorted) {  // any order is allowed: ask backwards first, and for a key that is not there
      for (auto it = keys.rbegin(); it != keys.rend(); ++it)
        if (!ra.HasKey(*it)) ASLP_ERR << "key " << *it << " not found by the random-access reader";
      if (ra.HasKey("<no-such-key>")) ASLP_ERR << "random-access reader found a key that is not there";
    }
    for (const std::string &k : keys) {  // "cs": keys are asked for in sorted order, as promised
      if (!ra.HasKey(k)) ASLP_ERR << "key " << k << " not found by the random-access reader";
      writer.Write(k, ra.Value(k));
      n++;
    }
  }
  if (!writer.Close()) ASLP_ERR << "error closing " << wspecifier;
  return n;
}

int main(int argc, char *argv[]) {
  try {
    const char *usage =
        "Copy a table of matrices, vectors, posteriors or integer vectors.\n"
        "Usage:  aslp-table-copy [options] <rspecifier> <wspecifier>\n"
        "e.g.:\n"
        " aslp-table-copy --type=matrix ark:feats.ark ark,t:-\n"
        " aslp-table-copy --type=ali-to-post ark:ali.ark ark:post.ark\n";
    ParseOptions po(usage);
    std::string type = "matrix";
    bool random_access = false;
    po.Register("type", &type, "matrix|vector|posterior|int32-vector|ali-to-post");
    po.Register("random-access", &random_access, "Fetch every object by key through the random-access reader");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) { po.PrintUsage(); return 1; }
    const std::string r = po.GetArg(1), w = po.GetArg(2);
    int n;
    if (type == "matrix") n = Copy<BaseFloatMatrixHolder>(r, w, random_access);
    else if (type == "vector") n = Copy<BaseFloatVectorHolder>(r, w, random_access);
    else if (type == "posterior") n = Copy<PosteriorHolder>(r, w, random_access);
    else if (type == "int32-vector") n = Copy<BasicVectorHolder<int32>>(r, w, random_access);
    else if (type == "ali-to-post") {
      SequentialInt32VectorReader reader(r);
      PosteriorWriter writer(w);
      n = 0;
      for (; !reader.Done(); reader.Next(), n++) {
        Posterior post;
        AlignmentToPosterior(reader.Value(), &post);
        writer.Write(reader.Key(), post);
      }
    } else {
      ASLP_ERR << "unknown --type " << type;
      return 1;
    }
    ASLP_LOG << "Copied " << n << " objects.";
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
