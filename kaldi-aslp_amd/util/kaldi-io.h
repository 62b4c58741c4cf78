// kaldi-io.h -- extended filenames: files, "-" (stdin/stdout), pipes ("cmd |", "| cmd") and "file:offset".
// Host-only.  Follows src/util/kaldi-io.{h,cc}: ClassifyRxfilename (:118-153), ClassifyWxfilename (:77-116),
// Input::Open with the "\0B" binary-header probe (:700-760), Output::Open (:640-690).
#pragma once
#include <cstdio>
#include <fstream>
#include <memory>
#include <string>

#include "base.h"

namespace aslp {

enum InputType { kNoInput, kFileInput, kStandardInput, kOffsetFileInput, kPipeInput };
enum OutputType { kNoOutput, kFileOutput, kStandardOutput, kPipeOutput };
InputType ClassifyRxfilename(const std::string &rxfilename);
OutputType ClassifyWxfilename(const std::string &wxfilename);
std::string PrintableRxfilename(const std::string &rxfilename);
std::string PrintableWxfilename(const std::string &wxfilename);

class Input {
 public:
  Input();
  // throws if it cannot be opened
  Input(const std::string &rxfilename, bool *contents_binary = NULL);
  ~Input();
  // contents_binary != NULL: the Kaldi header is consumed and its flag returned; returns false on failure
  bool Open(const std::string &rxfilename, bool *contents_binary = NULL);
  bool IsOpen() const { return stream_ != nullptr; }
  std::istream &Stream();
  // exit status of a pipe (0 for files); the stream is gone afterwards
  int Close();

 private:
  struct PipeBuf;
  std::unique_ptr<std::ifstream> file_;
  std::unique_ptr<PipeBuf> pipe_;
  std::unique_ptr<std::istream> pipe_stream_;
  std::istream *stream_ = nullptr;
  std::string open_name_;  // file part of the last "file:offset" open: a further offset into it only seeks
  Input(const Input &) = delete;
  Input &operator=(const Input &) = delete;
};

class Output {
 public:
  Output();
  Output(const std::string &wxfilename, bool binary, bool write_header = true);
  ~Output();
  bool Open(const std::string &wxfilename, bool binary, bool write_header);
  bool IsOpen() const { return stream_ != nullptr; }
  std::ostream &Stream();
  bool Close();  // false if the stream went bad or a pipe returned non-zero

 private:
  struct PipeBuf;
  std::unique_ptr<std::ofstream> file_;
  std::unique_ptr<PipeBuf> pipe_;
  std::unique_ptr<std::ostream> pipe_stream_;
  std::ostream *stream_ = nullptr;
  std::string name_;
  Output(const Output &) = delete;
  Output &operator=(const Output &) = delete;
};

}  // namespace aslp
