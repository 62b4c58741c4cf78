// host-io.cpp -- host-only half of the substrate: Kaldi matrix / vector stream formats, text utilities, libc-rand helpers.
// No HIP in this file: the table / archive tools that never touch the GPU link it without the device library.
#include <cmath>
#include <cstdlib>

#include "host-matrix.h"
#include "posterior.h"

namespace aslp {

int g_verbose_level = 0;

static std::string g_program_name;
const char *ProgramName() { return g_program_name.c_str(); }
void SetProgramName(const char *argv0) {
  const char *c = std::strrchr(argv0, '/');
  g_program_name = std::string(c ? c + 1 : argv0) + ":";
}

// ---- host matrix / vector I/O (matrix/kaldi-matrix.cc:1201-1430, kaldi-vector.cc:1094-1228) ------
void HostMatrix::Write(std::ostream &os, bool binary) const {
  if (!os.good()) ASLP_ERR << "Failed to write matrix to stream: stream not good";
  if (binary) {
    WriteToken(os, binary, "FM");
    WriteBasicType(os, binary, (int32)rows);
    WriteBasicType(os, binary, (int32)cols);
    os.write(reinterpret_cast<const char *>(data.data()), sizeof(float) * data.size());
  } else {
    if (cols == 0) {
      os << " [ ]\n";
    } else {
      os << " [";
      for (int i = 0; i < rows; i++) {
        os << "\n  ";
        for (int j = 0; j < cols; j++) os << (*this)(i, j) << " ";
      }
      os << "]\n";
    }
  }
  if (!os.good()) ASLP_ERR << "Failed to write matrix to stream";
}

static bool ParseSpecialFloat(const std::string &s, float *out) {
  std::string l;
  for (char c : s) l.push_back(::tolower(c));
  if (l == "inf" || l == "infinity") { *out = std::numeric_limits<float>::infinity(); return true; }
  if (l == "-inf" || l == "-infinity") { *out = -std::numeric_limits<float>::infinity(); return true; }
  if (l == "nan" || l == "-nan") { *out = std::numeric_limits<float>::quiet_NaN(); return true; }
  return false;
}

// Compressed feature matrices (what copy-feats --compress=true writes): matrix/compressed-matrix.cc:438-528.
// "CM": per-column header of four uint16 percentiles + one byte per element, column-major; "CM2": uint16 per
// element, row-major.  The global header on disk is {min_value, range, num_rows, num_cols} (the in-memory `format`
// field is not written, :454-455).  Dequantisation follows Uint16ToFloat (:245-251) and CharToFloat (:364-374),
// including the double-precision interpolation of the latter.
static void ReadCompressedMatrix(std::istream &is, const std::string &token, HostMatrix *m) {
  int format = 0;
  if (token == "CM") format = 1;
  else if (token == "CM2") format = 2;
  else ASLP_ERR << "Unexpected token " << token << ", expecting CM or CM2.";
  struct { float min_value, range; int32 num_rows, num_cols; } h;
  is.read(reinterpret_cast<char *>(&h), sizeof(h));
  if (is.fail()) ASLP_ERR << "Failed to read header";
  if (h.num_cols == 0) { m->Resize(0, 0); return; }
  if (h.num_rows < 0 || h.num_cols < 0) ASLP_ERR << "Compressed matrix: negative dimensions";
  m->Resize(h.num_rows, h.num_cols);
  auto u16 = [&](uint16_t v) { return h.min_value + h.range * 1.52590218966964e-05F * v; };
  if (format == 1) {
    std::vector<uint16_t> pch((size_t)h.num_cols * 4);
    is.read(reinterpret_cast<char *>(pch.data()), pch.size() * sizeof(uint16_t));
    std::vector<unsigned char> bytes((size_t)h.num_cols * h.num_rows);
    is.read(reinterpret_cast<char *>(bytes.data()), bytes.size());
    if (is.fail()) ASLP_ERR << "Failed to read data.";
    for (int c = 0; c < h.num_cols; c++) {
      const float p0 = u16(pch[4 * c]), p25 = u16(pch[4 * c + 1]), p75 = u16(pch[4 * c + 2]), p100 = u16(pch[4 * c + 3]);
      const unsigned char *col = bytes.data() + (size_t)c * h.num_rows;
      for (int r = 0; r < h.num_rows; r++) {
        const unsigned char v = col[r];
        float f;
        if (v <= 64) f = p0 + (p25 - p0) * v * (1 / 64.0);
        else if (v <= 192) f = p25 + (p75 - p25) * (v - 64) * (1 / 128.0);
        else f = p75 + (p100 - p75) * (v - 192) * (1 / 63.0);
        (*m)(r, c) = f;
      }
    }
  } else {
    std::vector<uint16_t> d((size_t)h.num_rows * h.num_cols);
    is.read(reinterpret_cast<char *>(d.data()), d.size() * sizeof(uint16_t));
    if (is.fail()) ASLP_ERR << "Failed to read data.";
    for (size_t i = 0; i < d.size(); i++) m->data[i] = u16(d[i]);
  }
}

HostMatrixSink &host_matrix_sink() {
  static thread_local HostMatrixSink s;
  return s;
}

void HostMatrix::Read(std::istream &is, bool binary) {
  if (binary) {
    int peekval = Peek(is, binary);
    std::string token;
    ReadToken(is, binary, &token);
    if (peekval == 'C') { ReadCompressedMatrix(is, token, this); return; }
    if (peekval == 'D') {
      if (token != "DM") ASLP_ERR << "Failed to read matrix from stream: expected token DM, got " << token;
      int32 r, c;
      ReadBasicType(is, binary, &r);
      ReadBasicType(is, binary, &c);
      std::vector<double> tmp((size_t)r * c);
      is.read(reinterpret_cast<char *>(tmp.data()), sizeof(double) * tmp.size());
      Resize(r, c);
      for (size_t i = 0; i < tmp.size(); i++) data[i] = (float)tmp[i];
    } else {
      if (token != "FM") ASLP_ERR << "Failed to read matrix from stream: Expected token FM, got " << token;
      int32 r, c;
      ReadBasicType(is, binary, &r);
      ReadBasicType(is, binary, &c);
      HostMatrixSink &sink = host_matrix_sink();
      if (sink.take != nullptr && r > 0 && c > 0) {
        rows = r; cols = c;
        data.clear();
        is.read(reinterpret_cast<char *>(sink.take(sink.ctx, r, c)), sizeof(float) * (size_t)r * c);
      } else {
        Resize(r, c);
        is.read(reinterpret_cast<char *>(data.data()), sizeof(float) * data.size());
      }
    }
    if (is.fail()) ASLP_ERR << "Failed to read matrix from stream (binary, truncated?)";
    return;
  }
  std::string str;
  is >> str;
  if (is.fail()) ASLP_ERR << "Failed to read matrix from stream: Expected \"[\", got EOF";
  if (str == "[]") { Resize(0, 0); return; }
  if (str != "[") ASLP_ERR << "Failed to read matrix from stream: Expected \"[\", got \"" << str << '"';
  std::vector<std::vector<float>> rows_v;
  std::vector<float> cur;
  while (true) {
    int i = is.peek();
    if (i == -1) ASLP_ERR << "Failed to read matrix from stream: got EOF while reading matrix data";
    char ch = static_cast<char>(i);
    if (ch == ']') {
      is.get();
      i = is.peek();
      if (static_cast<char>(i) == '\r') { is.get(); is.get(); }
      else if (static_cast<char>(i) == '\n') { is.get(); }
      if (!cur.empty()) rows_v.push_back(cur);
      if (rows_v.empty()) { Resize(0, 0); return; }
      int nr = rows_v.size(), nc = rows_v[0].size();
      Resize(nr, nc);
      for (int r = 0; r < nr; r++) {
        if ((int)rows_v[r].size() != nc)
          ASLP_ERR << "Failed to read matrix from stream: Matrix has inconsistent #cols: " << nc << " vs." << rows_v[r].size()
                   << " (processing row" << r << ")";
        for (int c = 0; c < nc; c++) (*this)(r, c) = rows_v[r][c];
      }
      return;
    } else if (ch == '\n' || ch == ';') {
      is.get();
      if (!cur.empty()) { rows_v.push_back(cur); cur.clear(); }
    } else if ((i >= '0' && i <= '9') || i == '-') {
      float r;
      is >> r;
      if (is.fail()) {
        // libstdc++ refuses "-inf"/"-nan": re-read as a word
        is.clear();
        std::string w;
        is >> w;
        if (!ParseSpecialFloat(w, &r)) ASLP_ERR << "Failed to read matrix from stream: stream failure/EOF while reading matrix data.";
      }
      cur.push_back(r);
    } else if (isspace(i)) {
      is.get();
    } else {
      std::string w;
      is >> w;
      float r;
      if (!ParseSpecialFloat(w, &r)) ASLP_ERR << "Failed to read matrix from stream: Expecting numeric matrix data, got " << w;
      cur.push_back(r);
    }
  }
}

void HostVector::Write(std::ostream &os, bool binary) const {
  if (!os.good()) ASLP_ERR << "Failed to write vector to stream: stream not good";
  if (binary) {
    WriteToken(os, binary, "FV");
    WriteBasicType(os, binary, (int32)data.size());
    os.write(reinterpret_cast<const char *>(data.data()), sizeof(float) * data.size());
  } else {
    os << " [ ";
    for (float v : data) os << v << " ";
    os << "]\n";
  }
  if (!os.good()) ASLP_ERR << "Failed to write vector to stream";
}

template <class Real>
static void ReadVectorImpl(std::istream &is, bool binary, std::vector<Real> *out) {
  if (binary) {
    int peekval = Peek(is, binary);
    std::string token;
    ReadToken(is, binary, &token);
    int32 size;
    if (peekval == 'D') {
      if (token != "DV") ASLP_ERR << "Failed to read vector from stream: expected token DV, got " << token;
      ReadBasicType(is, binary, &size);
      std::vector<double> tmp(size);
      if (size > 0) is.read(reinterpret_cast<char *>(tmp.data()), sizeof(double) * size);
      out->resize(size);
      for (int i = 0; i < size; i++) (*out)[i] = (Real)tmp[i];
    } else {
      if (token != "FV") ASLP_ERR << "Failed to read vector from stream: Expected token FV, got " << token;
      ReadBasicType(is, binary, &size);
      std::vector<float> tmp(size);
      if (size > 0) is.read(reinterpret_cast<char *>(tmp.data()), sizeof(float) * size);
      out->resize(size);
      for (int i = 0; i < size; i++) (*out)[i] = (Real)tmp[i];
    }
    if (is.fail()) ASLP_ERR << "Failed to read vector from stream: error reading vector data (binary mode); truncated stream?";
    return;
  }
  std::string s;
  is >> s;
  if (is.fail()) ASLP_ERR << "Failed to read vector from stream: EOF while trying to read vector.";
  if (s == "[]") { out->clear(); return; }
  if (s != "[") ASLP_ERR << "Failed to read vector from stream: Expected \"[\" but got " << s;
  std::vector<Real> data;
  while (true) {
    int i = is.peek();
    if (i == '-' || (i >= '0' && i <= '9')) {
      Real r;
      is >> r;
      if (is.fail()) {
        is.clear();
        std::string w;
        is >> w;
        float f;
        if (!ParseSpecialFloat(w, &f)) ASLP_ERR << "Failed to read vector from stream: failed to read number.";
        r = f;
      }
      data.push_back(r);
    } else if (i == ' ' || i == '\t') {
      is.get();
    } else if (i == ']') {
      is.get();
      *out = data;
      i = is.peek();
      if (static_cast<char>(i) == '\r') { is.get(); is.get(); }
      else if (static_cast<char>(i) == '\n') { is.get(); }
      return;
    } else if (i == -1) {
      ASLP_ERR << "Failed to read vector from stream: EOF while reading vector data.";
    } else if (i == '\n' || i == '\r') {
      ASLP_ERR << "Failed to read vector from stream: newline found while reading vector (maybe it's a matrix?)";
    } else {
      is >> s;
      float f;
      if (!ParseSpecialFloat(s, &f)) ASLP_ERR << "Failed to read vector from stream: Expecting numeric vector data, got " << s;
      data.push_back(f);
    }
  }
}
void HostVector::Read(std::istream &is, bool binary) { ReadVectorImpl(is, binary, &data); }
void HostVectorD::Read(std::istream &is, bool binary) { ReadVectorImpl(is, binary, &data); }
void HostVectorD::Write(std::ostream &os, bool binary) const {
  if (binary) {
    WriteToken(os, binary, "DV");
    WriteBasicType(os, binary, (int32)data.size());
    os.write(reinterpret_cast<const char *>(data.data()), sizeof(double) * data.size());
  } else {
    os << " [ ";
    for (double v : data) os << v << " ";
    os << "]\n";
  }
  if (!os.good()) ASLP_ERR << "Failed to write vector to stream";
}

// ---- text utils ---------------------------------------------------------------------------------
void SplitStringToVector(const std::string &full, const char *delim, bool omit_empty_strings, std::vector<std::string> *out) {
  size_t start = 0, found = 0, end = full.size();
  out->clear();
  while (found != std::string::npos) {
    found = full.find_first_of(delim, start);
    if (!omit_empty_strings || (found != start && start != end)) out->push_back(full.substr(start, found - start));
    start = found + 1;
  }
}
bool ConvertStringToInteger(const std::string &str, int32 *out) {
  const char *this_str = str.c_str();
  char *end = nullptr;
  errno = 0;
  long long i = strtoll(this_str, &end, 10);
  if (end != this_str) while (isspace(*end)) end++;
  if (end == this_str || *end != '\0' || errno != 0) return false;
  if (i > std::numeric_limits<int32>::max() || i < std::numeric_limits<int32>::min()) return false;
  *out = (int32)i;
  return true;
}
bool ConvertStringToReal(const std::string &str, float *out) {
  const char *this_str = str.c_str();
  char *end = nullptr;
  errno = 0;
  double d = strtod(this_str, &end);
  if (end != this_str) while (isspace(*end)) end++;
  if (end == this_str || *end != '\0' || errno != 0) return false;
  *out = (float)d;
  return true;
}
bool SplitStringToIntegers(const std::string &full, const char *delim, bool omit_empty_strings, std::vector<int32> *out) {
  if (*(full.c_str()) == '\0') { out->clear(); return true; }
  std::vector<std::string> split;
  SplitStringToVector(full, delim, omit_empty_strings, &split);
  out->resize(split.size());
  for (size_t i = 0; i < split.size(); i++)
    if (!ConvertStringToInteger(split[i], &(*out)[i])) return false;
  return true;
}
static char g_rand_statebuf[128];
static struct random_data g_rand_data;
static bool g_rand_seeded = false;
void SRand(unsigned seed) {
  std::memset(&g_rand_data, 0, sizeof(g_rand_data));
  std::memset(g_rand_statebuf, 0, sizeof(g_rand_statebuf));
  initstate_r(seed, g_rand_statebuf, sizeof(g_rand_statebuf), &g_rand_data);
  g_rand_seeded = true;
}
int Rand() {
  if (!g_rand_seeded) SRand(1);  // an unseeded rand() behaves as srand(1)
  int32_t r;
  random_r(&g_rand_data, &r);
  return r;
}
float RandUniform() { return (float)((Rand() + 1.0) / (RAND_MAX + 2.0)); }
float RandGauss() {
  // base/kaldi-math.h:104-107 writes both draws in one expression (order unspecified in C++); fixed here: radius first
  const float u1 = RandUniform(), u2 = RandUniform();
  return (float)(sqrtf(-2 * logf(u1)) * cosf(2 * M_PI * u2));
}

// ---- Posterior (hmm/posterior.cc:29-125) ---------------------------------------------------------------
void WritePosterior(std::ostream &os, bool binary, const Posterior &post) {
  if (binary) {
    WriteBasicType(os, binary, (int32)post.size());
    for (const auto &frame : post) {
      WriteBasicType(os, binary, (int32)frame.size());
      for (const auto &pr : frame) {
        WriteBasicType(os, binary, pr.first);
        WriteBasicType(os, binary, pr.second);
      }
    }
  } else {  // [ 1235 0.6 12 0.4 ] [ 34 1 ] ... terminated by a newline
    for (const auto &frame : post) {
      os << "[ ";
      for (const auto &pr : frame) os << pr.first << ' ' << pr.second << ' ';
      os << "] ";
    }
    os << '\n';
  }
  if (!os.good()) ASLP_ERR << "Output stream error writing Posterior.";
}

void ReadPosterior(std::istream &is, bool binary, Posterior *post) {
  post->clear();
  if (binary) {
    int32 sz;
    ReadBasicType(is, true, &sz);
    if (sz < 0 || sz > 10000000) ASLP_ERR << "Reading posterior: got negative or improbably large size" << sz;
    post->resize(sz);
    // (hmm/posterior.cc:56-72 reads every number with its own ReadBasicType; a million frames per second of training want fewer trips
    //  through the stream: the 5-byte frame header and the frame's 10-byte pairs are taken from the stream buffer in one piece each)
    std::streambuf *sb = is.rdbuf();
    char buf[10 * 64];
    for (auto &frame : *post) {
      if (sb->sgetn(buf, 5) != 5 || buf[0] != 4) { is.setstate(std::ios::failbit); ASLP_ERR << "ReadBasicType: failed reading a frame's size of a Posterior"; }
      int32 sz2;
      std::memcpy(&sz2, buf + 1, 4);
      if (sz2 < 0) ASLP_ERR << "Reading posteriors: got negative size";
      frame.resize(sz2);
      for (int32 at = 0; at < sz2; at += 64) {
        const int32 n = std::min<int32>(64, sz2 - at);
        if (sb->sgetn(buf, 10 * n) != 10 * n) { is.setstate(std::ios::failbit); ASLP_ERR << "ReadBasicType: failed reading the pairs of a Posterior frame"; }
        for (int32 i = 0; i < n; i++) {
          const char *q = buf + 10 * i;
          if (q[0] != 4 || q[5] != 4) ASLP_ERR << "ReadBasicType: did not get expected integer type, " << (int)q[0] << " vs. 4.  You can change this code to successfully read it later, if needed.";
          std::memcpy(&frame[at + i].first, q + 1, 4);
          std::memcpy(&frame[at + i].second, q + 6, 4);
        }
      }
    }
    return;
  }
  std::string line;
  std::getline(is, line);
  if (is.fail()) ASLP_ERR << "holder of Posterior: error reading line " << (is.eof() ? "[eof]" : "");
  std::istringstream line_is(line);
  while (true) {
    std::string str;
    line_is >> std::ws;
    if (line_is.eof()) break;
    line_is >> str;
    if (str != "[") {
      int32 str_int;
      ASLP_ERR << "Reading Posterior object: expecting [, got '" << str
               << (ConvertStringToInteger(str, &str_int) ? "': did you provide alignments instead of posteriors?" : "'.");
    }
    std::vector<std::pair<int32, BaseFloat>> this_vec;
    while (true) {
      line_is >> std::ws;
      if (line_is.peek() == ']') { line_is.get(); break; }
      int32 i;
      BaseFloat p;
      line_is >> i >> p;
      if (line_is.fail()) ASLP_ERR << "Error reading Posterior object (could not get data after \"[\");";
      this_vec.push_back(std::make_pair(i, p));
    }
    post->push_back(this_vec);
  }
}

void AlignmentToPosterior(const std::vector<int32> &ali, Posterior *post) {
  post->clear();
  post->resize(ali.size());
  for (size_t i = 0; i < ali.size(); i++) (*post)[i].push_back(std::make_pair(ali[i], (BaseFloat)1.0));
}

}  // namespace aslp
