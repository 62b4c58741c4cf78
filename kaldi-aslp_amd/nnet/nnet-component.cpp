// nnet-component.cpp -- marker table, factory, config-line Init and nnet-file Read/Write.
// Follows src/aslp-nnet/nnet-component.cc:45-350.
#include "nnet-component.h"

#include <algorithm>

#include "nnet-basic.h"
#include "nnet-conv.h"
#include "nnet-recurrent.h"
#include "nnet-temporal.h"

namespace aslp {

const struct Component::key_value Component::kMarkerMap[] = {  // nnet-component.cc:45-80
    {Component::kSoftmax, "<Softmax>"},
    {Component::kBlockSoftmax, "<BlockSoftmax>"},
    {Component::kSigmoid, "<Sigmoid>"},
    {Component::kTanh, "<Tanh>"},
    {Component::kDropout, "<Dropout>"},
    {Component::kReLU, "<ReLU>"},
    {Component::kLengthNormComponent, "<LengthNormComponent>"},
    {Component::kSplice, "<Splice>"},
    {Component::kCopy, "<Copy>"},
    {Component::kAddShift, "<AddShift>"},
    {Component::kRescale, "<Rescale>"},
    {Component::kAffineTransform, "<AffineTransform>"},
    {Component::kLinearTransform, "<LinearTransform>"},
    {Component::kConvolutionalComponent, "<ConvolutionalComponent>"},
    {Component::kMaxPoolingComponent, "<MaxPoolingComponent>"},
    {Component::kLstmProjectedStreams, "<LstmProjectedStreams>"},
    {Component::kBLstmProjectedStreams, "<BLstmProjectedStreams>"},
    {Component::kBatchNormalization, "<BatchNormalization>"},
    {Component::kInputLayer, "<InputLayer>"},
    {Component::kOutputLayer, "<OutputLayer>"},
    {Component::kScaleLayer, "<ScaleLayer>"},
    {Component::kLstm, "<Lstm>"},
    {Component::kBLstm, "<BLstm>"},
    {Component::kRowConvolution, "<RowConvolution>"},
    {Component::kBLstmProjectedStreamsLC, "<BLstmProjectedStreamsLC>"},
    {Component::kGruStreams, "<GruStreams>"},
    {Component::kLstmCifgProjectedStreams, "<LstmCifgProjectedStreams>"},
    {Component::kCompactFsmn, "<CompactFsmn>"},
    {Component::kPnormComponent, "<Pnorm>"},
    {Component::kPnormComponent, "<Maxout>"},  // sic: the reference maps <Maxout> to kPnormComponent (:79-80)
};

const char *Component::TypeToMarker(ComponentType t) {
  int32 N = sizeof(kMarkerMap) / sizeof(kMarkerMap[0]);
  for (int i = 0; i < N; i++)
    if (kMarkerMap[i].key == t) return kMarkerMap[i].value;
  ASLP_ERR << "Unknown type" << t;
  return NULL;
}

Component::ComponentType Component::MarkerToType(const std::string &s) {
  std::string s_lowercase(s);
  std::transform(s.begin(), s.end(), s_lowercase.begin(), ::tolower);
  int32 N = sizeof(kMarkerMap) / sizeof(kMarkerMap[0]);
  for (int i = 0; i < N; i++) {
    std::string m(kMarkerMap[i].value);
    std::transform(m.begin(), m.end(), m.begin(), ::tolower);
    if (s_lowercase == m) return kMarkerMap[i].key;
  }
  ASLP_ERR << "Unknown marker : '" << s << "'";
  return kUnknown;
}

Component *Component::NewComponentOfType(ComponentType comp_type, int32 input_dim, int32 output_dim) {
  Component *ans = NULL;
  switch (comp_type) {
    case kAffineTransform: ans = new AffineTransform(input_dim, output_dim); break;
    case kSoftmax: ans = new Softmax(input_dim, output_dim); break;
    case kBlockSoftmax: ans = new BlockSoftmax(input_dim, output_dim); break;
    case kSigmoid: ans = new Sigmoid(input_dim, output_dim); break;
    case kTanh: ans = new Tanh(input_dim, output_dim); break;
    case kReLU: ans = new ReLU(input_dim, output_dim); break;
    case kDropout: ans = new Dropout(input_dim, output_dim); break;
    case kSplice: ans = new Splice(input_dim, output_dim); break;
    case kCopy: ans = new CopyComponent(input_dim, output_dim); break;
    case kAddShift: ans = new AddShift(input_dim, output_dim); break;
    case kRescale: ans = new Rescale(input_dim, output_dim); break;
    case kBatchNormalization: ans = new BatchNormalization(input_dim, output_dim); break;
    case kInputLayer: ans = new InputLayer(input_dim, output_dim); break;
    case kOutputLayer: ans = new OutputLayer(input_dim, output_dim); break;
    case kScaleLayer: ans = new ScaleLayer(input_dim, output_dim); break;
    case kBLstmProjectedStreamsLC: ans = new BLstmProjectedStreamsLC(input_dim, output_dim); break;
    case kLstmProjectedStreams: ans = new LstmProjectedStreams(input_dim, output_dim); break;
    case kBLstmProjectedStreams: ans = new BLstmProjectedStreams(input_dim, output_dim); break;
    case kLstmCifgProjectedStreams: ans = new LstmCifgProjectedStreams(input_dim, output_dim); break;
    case kLstm: ans = new Lstm(input_dim, output_dim); break;
    case kBLstm: ans = new BLstm(input_dim, output_dim); break;
    case kGruStreams: ans = new GruStreams(input_dim, output_dim); break;
    case kRowConvolution: ans = new RowConvolution(input_dim, output_dim); break;
    case kCompactFsmn: ans = new CompactFsmn(input_dim, output_dim); break;
    case kLinearTransform: ans = new LinearTransform(input_dim, output_dim); break;
    case kConvolutionalComponent: ans = new ConvolutionalComponent(input_dim, output_dim); break;
    case kMaxPoolingComponent: ans = new MaxPoolingComponent(input_dim, output_dim); break;
    case kLengthNormComponent: ans = new LengthNormComponent(input_dim, output_dim); break;
    case kPnormComponent: ans = new PnormComponent(input_dim, output_dim); break;
    case kMaxoutComponent: ans = new MaxoutComponent(input_dim, output_dim); break;
    case kUnknown:
    default:
      ASLP_ERR << "Missing type: " << TypeToMarker(comp_type);   // nnet-component.cc:203-205
  }
  return ans;
}

// ---- the two textual forms of a component ---------------------------------------------------------------------------------------
// A prototype line:  <Marker> <InputDim> i <OutputDim> o [<Name> n <Input> a[:off],b[:off],...] <component options...>
// A stored record:   <Marker> o i [<Name> n] id [ inputs ] [ offsets ] <component data...>
// (the formats are the reference's, nnet-component.cc:211-342; the parsing below is this repo's)
namespace {

// "a:3,b,c:10" -> names {a, b, c}, offsets {3, 0, 10}: the producers a graph node reads and the column each lands at
struct LinkList {
  std::vector<std::string> names;
  std::vector<int32> offsets;
  explicit LinkList(const std::string &spec) {
    size_t pos = 0;
    while (pos <= spec.size()) {
      const size_t comma = std::min(spec.find(',', pos), spec.size());
      const std::string item = spec.substr(pos, comma - pos);
      pos = comma + 1;
      if (item.empty()) continue;
      const size_t colon = item.find(':');
      ASLP_ASSERT(colon == std::string::npos || item.find(':', colon + 1) == std::string::npos);   // name or name:offset
      int32 off = 0;
      if (colon != std::string::npos) ConvertStringToInteger(item.substr(colon + 1), &off);
      names.push_back(item.substr(0, colon));
      offsets.push_back(off);
    }
  }
};

// the fixed part of a stored record, in file order
struct RecordHeader {
  std::string marker, name;
  int32 dim_out = 0, dim_in = 0, id = 0;
  std::vector<int32> input, offset;
  // false at the end of the net (end of stream or </Nnet>)
  bool Read(std::istream &is, bool binary) {
    if (Peek(is, binary) == EOF) return false;
    ReadToken(is, binary, &marker);
    if (marker == "<Nnet>") ReadToken(is, binary, &marker);   // the opening tag may sit in front of the first record
    if (marker == "</Nnet>") return false;
    ReadBasicType(is, binary, &dim_out);
    ReadBasicType(is, binary, &dim_in);
    if (Peek(is, binary) == '<') {   // optional, graph nets only
      ExpectToken(is, binary, "<Name>");
      ReadToken(is, binary, &name);
    }
    ReadBasicType(is, binary, &id);
    ReadIntegerVector(is, binary, &input);
    ReadIntegerVector(is, binary, &offset);
    ASLP_ASSERT(input.size() == offset.size());
    return true;
  }
  void Write(std::ostream &os, bool binary, bool with_links) const {
    WriteToken(os, binary, marker);
    WriteBasicType(os, binary, dim_out);
    WriteBasicType(os, binary, dim_in);
    if (with_links) {
      if (!name.empty()) { WriteToken(os, binary, "<Name>"); WriteToken(os, binary, name); }
      WriteBasicType(os, binary, id);
      WriteIntegerVector(os, binary, input);
      WriteIntegerVector(os, binary, offset);
    }
    if (!binary) os << "\n";
  }
};

}  // namespace

Component *Component::Init(const std::string &conf_line) {
  std::istringstream is(conf_line);
  std::string marker;
  ReadToken(is, false, &marker);
  int32 dims[2] = {0, 0};
  const char *const dim_tokens[2] = {"<InputDim>", "<OutputDim>"};
  for (int i = 0; i < 2; i++) {
    ExpectToken(is, false, dim_tokens[i]);
    ReadBasicType(is, false, &dims[i]);
  }
  Component *comp = NewComponentOfType(MarkerToType(marker), dims[0], dims[1]);
  if (conf_line.find("<Name>") != std::string::npos) {   // a graph node: its own name and the producers it reads
    std::string name, inputs;
    ExpectToken(is, false, "<Name>");
    ReadToken(is, false, &name);
    ExpectToken(is, false, "<Input>");
    ReadToken(is, false, &inputs);
    const LinkList links(inputs);
    comp->SetName(name);
    comp->SetInputName(links.names);
    comp->SetOffset(links.offsets);
  }
  is >> std::ws;
  comp->InitData(is);   // whatever is left belongs to the component
  return comp;
}

Component *Component::Read(std::istream &is, bool binary) {
  RecordHeader h;
  if (!h.Read(is, binary)) return NULL;
  Component *comp = NewComponentOfType(MarkerToType(h.marker), h.dim_in, h.dim_out);
  comp->ReadData(is, binary);
  comp->SetName(h.name);
  comp->SetId(h.id);
  comp->SetInput(h.input);
  comp->SetOffset(h.offset);
  return comp;
}

static RecordHeader HeaderOf(const Component &c, const std::string &name, int32 id, const std::vector<int32> &input, const std::vector<int32> &offset) {
  RecordHeader h;
  h.marker = Component::TypeToMarker(c.GetType());
  h.dim_out = c.OutputDim();
  h.dim_in = c.InputDim();
  h.name = name;
  h.id = id;
  h.input = input;
  h.offset = offset;
  return h;
}

void Component::Write(std::ostream &os, bool binary) const {
  HeaderOf(*this, name_, id_, input_, offset_).Write(os, binary, true);
  this->WriteData(os, binary);
}

void Component::WriteStandard(std::ostream &os, bool binary) const {   // the upstream-Kaldi record: no name, id or links
  HeaderOf(*this, name_, id_, input_, offset_).Write(os, binary, false);
  this->WriteData(os, binary);
}


void Component::Feedforward(const CuMatrixBase &in, CuMatrix *out) {  // nnet-component.h:286-296
  if (input_dim_ != in.NumCols())
    ASLP_ERR << "Non-matching dims! " << TypeToMarker(GetType()) << " input-dim : " << input_dim_ << " data : " << in.NumCols();
  out->Resize(in.NumRows(), output_dim_, kSetZero);
  FeedforwardFnc(in, out);
}

void Component::Propagate(const CuMatrixBase &in, CuMatrix *out) {  // :303-314
  if (input_dim_ != in.NumCols())
    ASLP_ERR << "Non-matching dims! " << TypeToMarker(GetType()) << " input-dim : " << input_dim_ << " data : " << in.NumCols();
  // The reference zeroes `out` (kSetZero); every PropagateFnc here overwrites all of it,
  // so the zero-fill pass over HBM is skipped.
  out->Resize(in.NumRows(), output_dim_, kUndefined);
  PropagateFnc(in, out);
}

void Component::Backpropagate(const CuMatrixBase &in, const CuMatrixBase &out, const CuMatrixBase &out_diff, CuMatrix *in_diff) {
  if (output_dim_ != out_diff.NumCols())  // :317-347
    ASLP_ERR << "Non-matching output dims, component:" << output_dim_ << " data:" << out_diff.NumCols();
  if (in_diff == NULL) return;  // no nested-nnet components in this library
  in_diff->Resize(out_diff.NumRows(), input_dim_, BackpropOverwritesInDiff() ? kUndefined : kSetZero);
  ASLP_ASSERT((in.NumRows() == out.NumRows()) && (in.NumRows() == out_diff.NumRows()) && (in.NumRows() == in_diff->NumRows()));
  ASLP_ASSERT(in.NumCols() == in_diff->NumCols());
  ASLP_ASSERT(out.NumCols() == out_diff.NumCols());
  BackpropagateFnc(in, out, out_diff, in_diff);
}

void AppendRowMajor(const CuMatrixBase &m, std::vector<BaseFloat> *out) {
  HostMatrix h;
  m.CopyToMat(&h);
  out->insert(out->end(), h.data.begin(), h.data.end());
}
void AppendVector(const CuVectorBase &v, std::vector<BaseFloat> *out) {
  std::vector<float> h(v.Dim());
  v.CopyToHost(h.data());
  out->insert(out->end(), h.begin(), h.end());
}
void InitMatParamUniform(CuMatrix &m, float scale) {
  HostMatrix h(m.NumRows(), m.NumCols());
  for (auto &x : h.data) x = (RandUniform() - 0.5f) * 2 * scale;
  m.CopyFromMat(h);
}
void InitVecParamUniform(CuVector &v, float scale) {
  HostVector h(v.Dim());
  for (auto &x : h.data) x = (RandUniform() - 0.5) * 2 * scale;
  v = h;
}

}  // namespace aslp
