// parse-options.cpp -- see parse-options.h (src/util/parse-options.cc).  Host-only.
#include "parse-options.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <iomanip>

namespace aslp {

ParseOptions::ParseOptions(const char *usage) : usage_(usage) {
  Reg("config", kString, &config_, "Configuration file to read (this option may be repeated)", true);
  Reg("print-args", kBool, &print_args_, "Print the command line arguments (to stderr)", true);
  Reg("help", kBool, &help_, "Print out usage message", true);
  Reg("verbose", kInt, &g_verbose_level, "Verbose level (higher->more logging)", true);
}

void ParseOptions::NormalizeArgName(std::string *str) {  // lower case, '_' -> '-'
  std::string out;
  for (char c : *str) out.push_back(c == '_' ? '-' : (char)std::tolower((unsigned char)c));
  *str = out;
}

static std::string NumToString(double v) { std::ostringstream s; s << v; return s.str(); }

void ParseOptions::Reg(const std::string &name, Kind k, void *ptr, const std::string &doc, bool standard) {
  std::string idx = name;
  NormalizeArgName(&idx);
  if (opts_.count(idx)) { ASLP_WARN << "Registering option twice, ignoring second time: " << name; }
  std::string d = doc;
  switch (k) {  // parse-options.cc:120-180: the default goes into the usage text at registration time
    case kBool: d += std::string(" (bool, default = ") + (*static_cast<bool *>(ptr) ? "true)" : "false)"); break;
    case kInt: d += " (int, default = " + std::to_string(*static_cast<int32 *>(ptr)) + ")"; break;
    case kUint: d += " (uint, default = " + std::to_string(*static_cast<uint32_t *>(ptr)) + ")"; break;
    case kFloat: d += " (float, default = " + NumToString(*static_cast<float *>(ptr)) + ")"; break;
    case kDouble: d += " (double, default = " + NumToString(*static_cast<double *>(ptr)) + ")"; break;
    case kString: d += " (string, default = \"" + *static_cast<std::string *>(ptr) + "\")"; break;
  }
  opts_[idx] = Opt{k, ptr, name, d, standard};
}

std::string ParseOptions::GetArg(int i) const {
  if (i < 1 || i > (int)positional_.size()) ASLP_ERR << "ParseOptions::GetArg, invalid index " << i;
  return positional_[i - 1];
}

std::string ParseOptions::Escape(const std::string &str) {  // parse-options.cc:250-300: quote what a shell would split
  const char *ok_chars = "[]~#^_-+=:.,/";
  bool ok = !str.empty();
  for (unsigned char c : str)
    if (!isalnum(c) && !strchr(ok_chars, c)) ok = false;
  if (ok) return str;
  char quote = '\'';
  const char *escape = "'\\''";
  if (strchr(str.c_str(), '\'') && !strpbrk(str.c_str(), "\"`$\\")) { quote = '"'; escape = "\\\""; }
  std::string out(1, quote);
  for (char c : str) {
    if (c == quote) out += escape;
    else out.push_back(c);
  }
  out.push_back(quote);
  return out;
}

void ParseOptions::SplitLongArg(const std::string &in, std::string *key, std::string *value, bool *has_equal_sign) {
  size_t pos = in.find('=');
  if (pos == std::string::npos) {  // "--flag"
    *key = std::string(in, 2);
    value->clear();
    *has_equal_sign = false;
  } else if (pos == 2) {
    ASLP_ERR << "Invalid option (no key): " << in;
  } else {
    *key = std::string(in, 2, pos - 2);
    *value = std::string(in, pos + 1);
    *has_equal_sign = true;
  }
}

static void Trim(std::string *s) {
  const char *white = " \t\n\r\f\v";
  size_t a = s->find_first_not_of(white), b = s->find_last_not_of(white);
  if (a == std::string::npos) s->clear();
  else *s = s->substr(a, b - a + 1);
}

bool ParseOptions::ToBool(std::string str) {
  std::transform(str.begin(), str.end(), str.begin(), ::tolower);
  if (str == "true" || str == "t" || str == "1" || str == "") return true;
  if (str == "false" || str == "f" || str == "0") return false;
  PrintUsage(true);
  ASLP_ERR << "Invalid format for boolean argument [expected true or false]: " << str;
  return false;
}

bool ParseOptions::SetOption(const std::string &key, const std::string &value, bool has_equal_sign) {
  auto it = opts_.find(key);
  if (it == opts_.end()) return false;
  Opt &o = it->second;
  switch (o.kind) {
    case kBool:
      if (has_equal_sign && value == "") ASLP_ERR << "Invalid option --" << key << "=";
      *static_cast<bool *>(o.ptr) = ToBool(value);
      break;
    case kInt: {
      int32 v;
      if (!ConvertStringToInteger(value, &v)) ASLP_ERR << "Invalid integer option \"" << value << "\"";
      *static_cast<int32 *>(o.ptr) = v;
      break;
    }
    case kUint: {
      char *end;
      unsigned long v = strtoul(value.c_str(), &end, 10);
      if (value.empty() || *end != '\0' || value[0] == '-') ASLP_ERR << "Invalid integer option \"" << value << "\"";
      *static_cast<uint32_t *>(o.ptr) = (uint32_t)v;
      break;
    }
    case kFloat: {
      float v;
      if (!ConvertStringToReal(value, &v)) ASLP_ERR << "Invalid floating-point option \"" << value << "\"";
      *static_cast<float *>(o.ptr) = v;
      break;
    }
    case kDouble: {
      char *end;
      double v = strtod(value.c_str(), &end);
      if (value.empty() || *end != '\0') ASLP_ERR << "Invalid floating-point option \"" << value << "\"";
      *static_cast<double *>(o.ptr) = v;
      break;
    }
    case kString:
      if (!has_equal_sign) ASLP_ERR << "Invalid option --" << key;
      *static_cast<std::string *>(o.ptr) = value;
      break;
  }
  return true;
}

void ParseOptions::ReadConfigFile(const std::string &filename) {  // parse-options.cc:466-505: one "--name=value" per line, # comments
  std::ifstream is(filename.c_str());
  if (!is.good()) ASLP_ERR << "Cannot open config file: " << filename;
  std::string line, key, value;
  while (std::getline(is, line)) {
    size_t pos = line.find('#');
    if (pos != std::string::npos) line.erase(pos);
    Trim(&line);
    if (line.empty()) continue;
    if (line.substr(0, 2) != "--")
      ASLP_ERR << "Reading config file " << filename << ": line does not begin with -- (should be of the form --x=y): " << line;
    bool has_equal_sign;
    SplitLongArg(line, &key, &value, &has_equal_sign);
    NormalizeArgName(&key);
    Trim(&value);
    if (!SetOption(key, value, has_equal_sign)) {
      PrintUsage(true);
      ASLP_ERR << "Invalid option " << line << " in config file " << filename;
    }
  }
}

int ParseOptions::Read(int argc, const char *const *argv) {
  argc_ = argc;
  argv_ = argv;
  if (argc > 0) SetProgramName(argv[0]);
  std::string key, value;
  int i;
  for (i = 1; i < argc; i++) {  // first pass: config files and --help
    if (std::strncmp(argv[i], "--", 2) == 0) {
      if (std::strcmp(argv[i], "--") == 0) break;
      bool has_equal_sign;
      SplitLongArg(argv[i], &key, &value, &has_equal_sign);
      NormalizeArgName(&key);
      Trim(&value);
      if (key == "config") ReadConfigFile(value);
      if (key == "help") { PrintUsage(); exit(0); }
    }
  }
  bool double_dash_seen = false;
  for (i = 1; i < argc; i++) {  // second pass: the command line overrides the config files
    if (std::strncmp(argv[i], "--", 2) == 0) {
      if (std::strcmp(argv[i], "--") == 0) { i += 1; double_dash_seen = true; break; }
      bool has_equal_sign;
      SplitLongArg(argv[i], &key, &value, &has_equal_sign);
      NormalizeArgName(&key);
      Trim(&value);
      if (!SetOption(key, value, has_equal_sign)) {
        PrintUsage(true);
        ASLP_ERR << "Invalid option " << argv[i];
      }
    } else {
      break;
    }
  }
  for (; i < argc; i++) {
    if (std::strcmp(argv[i], "--") == 0 && !double_dash_seen) double_dash_seen = true;
    else positional_.push_back(std::string(argv[i]));
  }
  if (print_args_) {
    std::ostringstream strm;
    for (int j = 0; j < argc; j++) strm << Escape(argv[j]) << " ";
    strm << '\n';
    std::cerr << strm.str() << std::flush;
  }
  return i;
}

void ParseOptions::PrintUsage(bool print_command_line) {
  std::cerr << '\n' << usage_ << '\n';
  bool header = false;
  for (auto &kv : opts_) {
    if (kv.second.standard) continue;
    if (!header) { std::cerr << "Options:" << '\n'; header = true; }
    std::cerr << "  --" << std::setw(25) << std::left << kv.second.name << " : " << kv.second.doc << '\n';
  }
  if (header) std::cerr << '\n';
  std::cerr << "Standard options:" << '\n';
  for (auto &kv : opts_)
    if (kv.second.standard) std::cerr << "  --" << std::setw(25) << std::left << kv.second.name << " : " << kv.second.doc << '\n';
  std::cerr << '\n';
  if (print_command_line) {
    std::ostringstream strm;
    strm << "Command line was: ";
    for (int j = 0; j < argc_; j++) strm << Escape(argv_[j]) << " ";
    strm << '\n';
    std::cerr << strm.str() << std::flush;
  }
}

void ParseOptions::PrintConfig(std::ostream &os) {
  os << '\n' << "[[ Configuration of UI-Registered options ]]" << '\n';
  for (auto &kv : opts_) {
    const Opt &o = kv.second;
    os << o.name << " = ";
    switch (o.kind) {
      case kBool: os << (*static_cast<bool *>(o.ptr) ? "true" : "false"); break;
      case kInt: os << *static_cast<int32 *>(o.ptr); break;
      case kUint: os << *static_cast<uint32_t *>(o.ptr); break;
      case kFloat: os << *static_cast<float *>(o.ptr); break;
      case kDouble: os << *static_cast<double *>(o.ptr); break;
      case kString: os << "'" << *static_cast<std::string *>(o.ptr) << "'"; break;
    }
    os << '\n';
  }
  os << '\n';
}

}  // namespace aslp
