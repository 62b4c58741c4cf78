// parse-options.h -- the command-line convention of every aslp-nnet-* tool (src/util/parse-options.{h,cc}):
// "--name=value" options (a bare "--flag" sets a bool), "--config=file", "--help", "--print-args", "--verbose",
// "--" ends the options, the rest are positional.  Host-only.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "base.h"

namespace aslp {

class OptionsItf {  // itf/options-itf.h
 public:
  virtual void Register(const std::string &name, bool *ptr, const std::string &doc) = 0;
  virtual void Register(const std::string &name, int32 *ptr, const std::string &doc) = 0;
  virtual void Register(const std::string &name, uint32_t *ptr, const std::string &doc) = 0;
  virtual void Register(const std::string &name, float *ptr, const std::string &doc) = 0;
  virtual void Register(const std::string &name, double *ptr, const std::string &doc) = 0;
  virtual void Register(const std::string &name, std::string *ptr, const std::string &doc) = 0;
  virtual ~OptionsItf() {}
};

class ParseOptions : public OptionsItf {
 public:
  explicit ParseOptions(const char *usage);
  void Register(const std::string &name, bool *ptr, const std::string &doc) { Reg(name, kBool, ptr, doc, false); }
  void Register(const std::string &name, int32 *ptr, const std::string &doc) { Reg(name, kInt, ptr, doc, false); }
  void Register(const std::string &name, uint32_t *ptr, const std::string &doc) { Reg(name, kUint, ptr, doc, false); }
  void Register(const std::string &name, float *ptr, const std::string &doc) { Reg(name, kFloat, ptr, doc, false); }
  void Register(const std::string &name, double *ptr, const std::string &doc) { Reg(name, kDouble, ptr, doc, false); }
  void Register(const std::string &name, std::string *ptr, const std::string &doc) { Reg(name, kString, ptr, doc, false); }
  // parses argv; "--help" prints the usage and exits 0; an unknown / malformed option prints the usage and throws
  int Read(int argc, const char *const *argv);
  void PrintUsage(bool print_command_line = false);
  void PrintConfig(std::ostream &os);
  int NumArgs() const { return (int)positional_.size(); }
  std::string GetArg(int i) const;  // 1-based
  std::string GetOptArg(int i) const { return i <= NumArgs() ? GetArg(i) : ""; }
  static std::string Escape(const std::string &str);

 private:
  enum Kind { kBool, kInt, kUint, kFloat, kDouble, kString };
  struct Opt { Kind kind; void *ptr; std::string name, doc; bool standard; };
  void Reg(const std::string &name, Kind k, void *ptr, const std::string &doc, bool standard);
  bool SetOption(const std::string &key, const std::string &value, bool has_equal_sign);
  void ReadConfigFile(const std::string &filename);
  static void SplitLongArg(const std::string &in, std::string *key, std::string *value, bool *has_equal_sign);
  static void NormalizeArgName(std::string *str);
  bool ToBool(std::string str);
  std::map<std::string, Opt> opts_;  // ordered by normalised name, like the reference's doc map
  std::vector<std::string> positional_;
  const char *usage_;
  bool print_args_ = true, help_ = false;
  std::string config_;
  int argc_ = 0;
  const char *const *argv_ = NULL;
};

}  // namespace aslp
