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

Component *Component::Init(const std::string &conf_line) {  // nnet-component.cc:211-285
  std::istringstream is(conf_line);
  std::string component_type_string;
  int32 input_dim, output_dim;
  ReadToken(is, false, &component_type_string);
  ComponentType component_type = MarkerToType(component_type_string);
  ExpectToken(is, false, "<InputDim>");
  ReadBasicType(is, false, &input_dim);
  ExpectToken(is, false, "<OutputDim>");
  ReadBasicType(is, false, &output_dim);
  Component *ans = NewComponentOfType(component_type, input_dim, output_dim);
  if (conf_line.find("<Name>") != std::string::npos) {
    std::string name;
    ExpectToken(is, false, "<Name>");
    ReadToken(is, false, &name);
    std::string input_string;
    ExpectToken(is, false, "<Input>");
    ReadToken(is, false, &input_string);
    std::vector<std::string> sub_input_string;
    SplitStringToVector(input_string, ",", true, &sub_input_string);
    int32 num_input = sub_input_string.size();
    std::vector<std::string> input_name;
    std::vector<int32> offset(num_input, 0);
    for (int i = 0; i < num_input; i++) {
      std::vector<std::string> field;
      SplitStringToVector(sub_input_string[i], ":", true, &field);
      ASLP_ASSERT(field.size() >= 1);
      ASLP_ASSERT(field.size() <= 2);
      if (field.size() == 2) ConvertStringToInteger(field[1], &offset[i]);
      input_name.push_back(field[0]);
    }
    ans->SetInputName(input_name);
    ans->SetName(name);
    ans->SetOffset(offset);
  }
  is >> std::ws;
  ans->InitData(is);
  return ans;
}

Component *Component::Read(std::istream &is, bool binary) {  // nnet-component.cc:288-325
  int32 dim_out, dim_in;
  std::string token;
  int first_char = Peek(is, binary);
  if (first_char == EOF) return NULL;
  ReadToken(is, binary, &token);
  if (token == "<Nnet>") ReadToken(is, binary, &token);
  if (token == "</Nnet>") return NULL;
  ReadBasicType(is, binary, &dim_out);
  ReadBasicType(is, binary, &dim_in);
  std::string name;
  int32 id;
  std::vector<int32> input, offset;
  if (Peek(is, binary) == '<') {
    ExpectToken(is, binary, "<Name>");
    ReadToken(is, binary, &name);
  }
  ReadBasicType(is, binary, &id);
  ReadIntegerVector(is, binary, &input);
  ReadIntegerVector(is, binary, &offset);
  ASLP_ASSERT(input.size() == offset.size());
  Component *ans = NewComponentOfType(MarkerToType(token), dim_in, dim_out);
  ans->ReadData(is, binary);
  ans->SetName(name);
  ans->SetId(id);
  ans->SetInput(input);
  ans->SetOffset(offset);
  return ans;
}

void Component::Write(std::ostream &os, bool binary) const {  // nnet-component.cc:328-342
  WriteToken(os, binary, Component::TypeToMarker(GetType()));
  WriteBasicType(os, binary, OutputDim());
  WriteBasicType(os, binary, InputDim());
  if (!name_.empty()) {
    WriteToken(os, binary, "<Name>");
    WriteToken(os, binary, name_);
  }
  WriteBasicType(os, binary, id_);
  WriteIntegerVector(os, binary, input_);
  WriteIntegerVector(os, binary, offset_);
  if (!binary) os << "\n";
  this->WriteData(os, binary);
}

void Component::WriteStandard(std::ostream &os, bool binary) const {
  WriteToken(os, binary, Component::TypeToMarker(GetType()));
  WriteBasicType(os, binary, OutputDim());
  WriteBasicType(os, binary, InputDim());
  if (!binary) os << "\n";
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
