// kaldi-io.cpp -- see kaldi-io.h.  Host-only.
#include "kaldi-io.h"

#include <ext/stdio_filebuf.h>

#include <cctype>
#include <cstdlib>
#include <cstring>

namespace aslp {

InputType ClassifyRxfilename(const std::string &filename) {  // kaldi-io.cc:118-153
  const char *c = filename.c_str();
  if (*c == '\0' || (*c == '-' && c[1] == '\0')) return kStandardInput;
  if (*c == '|') return kNoInput;  // an output pipe
  if (isspace(*c) || isspace(c[filename.length() - 1])) return kNoInput;
  if ((*c == 't' || *c == 'b') && c[1] == ',') return kNoInput;  // an rspecifier where a filename was expected
  const char *d = c + filename.length() - 1;
  if (*d == '|') return kPipeInput;
  if (isdigit(*d)) {
    while (isdigit(*d) && d > c) d--;
    return *d == ':' ? kOffsetFileInput : kFileInput;
  }
  if (strchr(c, '|') != NULL) {
    ASLP_WARN << "Trying to classify rxfilename with pipe symbol in the wrong place (pipe without | at the end?): " << filename;
    return kNoInput;
  }
  return kFileInput;
}

OutputType ClassifyWxfilename(const std::string &filename) {  // kaldi-io.cc:77-116
  const char *c = filename.c_str();
  if (*c == '\0' || (*c == '-' && c[1] == '\0')) return kStandardOutput;
  if (*c == '|') return kPipeOutput;
  if (isspace(*c) || isspace(c[filename.length() - 1])) return kNoOutput;
  if ((*c == 't' || *c == 'b') && c[1] == ',') return kNoOutput;
  const char *d = c + filename.length() - 1;
  if (*d == '|') return kNoOutput;  // an input pipe
  if (isdigit(*d)) {
    while (isdigit(*d) && d > c) d--;
    if (*d == ':') return kNoOutput;  // file:offset is not writable
    return kFileOutput;
  }
  if (strchr(c, '|') != NULL) {
    ASLP_WARN << "Trying to classify wxfilename with pipe symbol in the wrong place (pipe without | at the beginning?): " << filename;
    return kNoOutput;
  }
  return kFileOutput;
}

std::string PrintableRxfilename(const std::string &rx) { return (rx == "" || rx == "-") ? "standard input" : rx; }
std::string PrintableWxfilename(const std::string &wx) { return (wx == "" || wx == "-") ? "standard output" : wx; }

struct Input::PipeBuf {
  FILE *f = nullptr;
  std::unique_ptr<__gnu_cxx::stdio_filebuf<char>> buf;
};
struct Output::PipeBuf {
  FILE *f = nullptr;
  std::unique_ptr<__gnu_cxx::stdio_filebuf<char>> buf;
};

Input::Input() {}
Input::~Input() { Close(); }
Input::Input(const std::string &rxfilename, bool *contents_binary) {
  if (!Open(rxfilename, contents_binary)) ASLP_ERR << "Error opening input stream " << PrintableRxfilename(rxfilename);
}

std::istream &Input::Stream() {
  if (!stream_) ASLP_ERR << "Input::Stream(), not open.";
  return *stream_;
}

int Input::Close() {
  int status = 0;
  stream_ = nullptr;
  open_name_.clear();
  file_.reset();
  if (pipe_) {
    pipe_stream_.reset();
    pipe_->buf.reset();
    if (pipe_->f) status = pclose(pipe_->f);
    pipe_.reset();
  }
  return status;
}

bool Input::Open(const std::string &rxfilename, bool *contents_binary) {
  const InputType type = ClassifyRxfilename(rxfilename);
  if (type == kOffsetFileInput) {
    const size_t pos = rxfilename.find_last_of(':');
    const std::string fname(rxfilename, 0, pos);
    const long offset = atol(rxfilename.c_str() + pos + 1);
    if (!(file_ && open_name_ == fname)) {  // a further offset into the file already open only seeks (kaldi-io.cc:711-722)
      Close();
      file_.reset(new std::ifstream(fname.c_str(), std::ios_base::in | std::ios_base::binary));
      if (!file_->is_open()) { file_.reset(); return false; }
      open_name_ = fname;
    }
    file_->clear();
    file_->seekg(offset, std::ios_base::beg);
    if (file_->fail()) { Close(); return false; }
    stream_ = file_.get();
  } else {
    Close();
    if (type == kFileInput) {
      file_.reset(new std::ifstream(rxfilename.c_str(), std::ios_base::in | std::ios_base::binary));
      if (!file_->is_open()) { file_.reset(); return false; }
      stream_ = file_.get();
    } else if (type == kStandardInput) {
      stream_ = &std::cin;
    } else if (type == kPipeInput) {
      const std::string cmd(rxfilename, 0, rxfilename.length() - 1);  // without the trailing '|'
      pipe_.reset(new PipeBuf);
      pipe_->f = popen(cmd.c_str(), "r");
      if (!pipe_->f) { pipe_.reset(); ASLP_WARN << "Failed opening pipe for reading, command is: " << cmd; return false; }
      pipe_->buf.reset(new __gnu_cxx::stdio_filebuf<char>(pipe_->f, std::ios_base::in | std::ios_base::binary));
      pipe_stream_.reset(new std::istream(pipe_->buf.get()));
      stream_ = pipe_stream_.get();
    } else {
      ASLP_WARN << "Invalid input filename format " << PrintableRxfilename(rxfilename);
      return false;
    }
  }
  if (contents_binary != NULL && !InitKaldiInputStream(*stream_, contents_binary)) { Close(); return false; }
  return true;
}

Output::Output() {}
Output::Output(const std::string &wxfilename, bool binary, bool write_header) {
  if (!Open(wxfilename, binary, write_header)) ASLP_ERR << "Error opening output stream " << PrintableWxfilename(wxfilename);
}
Output::~Output() {
  if (IsOpen() && !Close()) std::cerr << "WARNING: error closing output file " << PrintableWxfilename(name_) << std::endl;
}
std::ostream &Output::Stream() {
  if (!stream_) ASLP_ERR << "Output::Stream() called but not open.";
  return *stream_;
}

bool Output::Close() {
  if (!stream_) return false;
  bool ok = true;
  stream_->flush();
  if (stream_->fail()) ok = false;
  stream_ = nullptr;
  if (file_) { file_->close(); if (file_->fail()) ok = false; file_.reset(); }
  if (pipe_) {
    pipe_stream_.reset();
    pipe_->buf.reset();
    if (pipe_->f && pclose(pipe_->f) != 0) ok = false;
    pipe_.reset();
  }
  return ok;
}

bool Output::Open(const std::string &wxfilename, bool binary, bool write_header) {
  if (IsOpen() && !Close()) ASLP_ERR << "Output::Open(), failed to close output stream: " << PrintableWxfilename(name_);
  name_ = wxfilename;
  const OutputType type = ClassifyWxfilename(wxfilename);
  if (type == kFileOutput) {
    file_.reset(new std::ofstream(wxfilename.c_str(), std::ios_base::out | std::ios_base::binary));
    if (!file_->is_open()) { file_.reset(); return false; }
    stream_ = file_.get();
  } else if (type == kStandardOutput) {
    stream_ = &std::cout;
  } else if (type == kPipeOutput) {
    const std::string cmd(wxfilename, 1);
    pipe_.reset(new PipeBuf);
    pipe_->f = popen(cmd.c_str(), "w");
    if (!pipe_->f) { pipe_.reset(); ASLP_WARN << "Failed opening pipe for writing, command is: " << cmd; return false; }
    pipe_->buf.reset(new __gnu_cxx::stdio_filebuf<char>(pipe_->f, std::ios_base::out | std::ios_base::binary));
    pipe_stream_.reset(new std::ostream(pipe_->buf.get()));
    stream_ = pipe_stream_.get();
  } else {
    ASLP_WARN << "Invalid output filename format " << PrintableWxfilename(wxfilename);
    return false;
  }
  if (write_header) {
    InitKaldiOutputStream(*stream_, binary);
    if (!stream_->good()) { Close(); return false; }
  }
  return true;
}

}  // namespace aslp
