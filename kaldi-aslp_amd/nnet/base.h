// base.h -- error/log macros and the Kaldi stream formats the nnet files use.
//
// Mirrors the slice of src/base/{kaldi-error.h, io-funcs.h, io-funcs-inl.h, io-funcs.cc} the
// aslp-nnet model files depend on: tokens end with a space in both modes;
// binary basic types carry a size byte; integer vectors are "[ 1 2 3 ]\n" in text and
// size-byte + int32 count + raw data in binary; binary streams start with "\0B".
// Errors throw std::runtime_error like KALDI_ERR (base/kaldi-error.h).
#pragma once
#include <chrono>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace aslp {

typedef float BaseFloat;
typedef int32_t int32;

// "name:" of the running tool (empty inside the library / Python); set by ParseOptions::Read like Kaldi's g_program_name
const char *ProgramName();
void SetProgramName(const char *argv0);

class MessageLogger {
 public:
  MessageLogger(const char *sev, const char *func, const char *file, int line, bool fatal) : fatal_(fatal) {
    const char *b = std::strrchr(file, '/');
    // "LOG (program:function():file:line) message", the program part once ParseOptions::Read has seen argv[0]
    ss_ << sev << " (" << ProgramName() << func << "():" << (b ? b + 1 : file) << ':' << line << ") ";
  }
  ~MessageLogger() noexcept(false) {
    if (fatal_) throw std::runtime_error(ss_.str());
    std::cerr << ss_.str() << std::endl;
  }
  std::ostream &stream() { return ss_; }

 private:
  std::ostringstream ss_;
  bool fatal_;
};

extern int g_verbose_level;

#define ASLP_ERR ::aslp::MessageLogger("ERROR", __func__, __FILE__, __LINE__, true).stream()
#define ASLP_WARN ::aslp::MessageLogger("WARNING", __func__, __FILE__, __LINE__, false).stream()
#define ASLP_LOG ::aslp::MessageLogger("LOG", __func__, __FILE__, __LINE__, false).stream()
#define ASLP_VLOG(v) \
  if ((v) <= ::aslp::g_verbose_level) ::aslp::MessageLogger("VLOG", __func__, __FILE__, __LINE__, false).stream()
#define ASLP_ASSERT(cond)                                                   \
  do {                                                                      \
    if (!(cond)) ASLP_ERR << "Assertion failed: (" << #cond << ")";         \
  } while (0)

struct Timer {  // base/timer.h
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void Reset() { t0 = std::chrono::steady_clock::now(); }
  double Elapsed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// ---- io-funcs ------------------------------------------------------------------------------
inline void InitKaldiOutputStream(std::ostream &os, bool binary) {
  if (binary) {
    os.put('\0');
    os.put('B');
  }
  if (os.precision() < 7) os.precision(7);
}
inline bool InitKaldiInputStream(std::istream &is, bool *binary) {
  if (is.peek() == '\0') {
    is.get();
    if (is.peek() != 'B') return false;
    is.get();
    *binary = true;
  } else {
    *binary = false;
  }
  return true;
}

inline int Peek(std::istream &is, bool binary) {
  if (!binary) is >> std::ws;
  return is.peek();
}

inline void CheckToken(const char *token) {
  if (*token == '\0') ASLP_ERR << "Token is empty (not a valid token)";
  for (const char *p = token; *p; ++p)
    if (::isspace(*p)) ASLP_ERR << "Token is not a valid token (contains space): '" << token << "'";
}
inline void WriteToken(std::ostream &os, bool, const std::string &token) {
  CheckToken(token.c_str());
  os << token << " ";
  if (os.fail()) throw std::runtime_error("Write failure in WriteToken.");
}
inline void ReadToken(std::istream &is, bool binary, std::string *str) {
  if (!binary) is >> std::ws;
  is >> *str;
  if (is.fail()) ASLP_ERR << "ReadToken, failed to read token at file position " << is.tellg();
  if (!isspace(is.peek()))
    ASLP_ERR << "ReadToken, expected space after token, saw instead " << static_cast<char>(is.peek())
             << ", at file position " << is.tellg();
  is.get();
}
inline void ExpectToken(std::istream &is, bool binary, const std::string &token) {
  long pos_at_start = is.tellg();
  if (!binary) is >> std::ws;
  std::string str;
  is >> str;
  is.get();
  if (is.fail()) ASLP_ERR << "Failed to read token [started at file position " << pos_at_start << "], expected " << token;
  if (str != token) ASLP_ERR << "Expected token \"" << token << "\", got instead \"" << str << "\".";
}

template <class T>
inline void WriteBasicType(std::ostream &os, bool binary, T t) {
  static_assert(std::numeric_limits<T>::is_integer, "integer overload");
  if (binary) {
    char len_c = (std::numeric_limits<T>::is_signed ? 1 : -1) * static_cast<char>(sizeof(t));
    os.put(len_c);
    os.write(reinterpret_cast<const char *>(&t), sizeof(t));
  } else {
    if (sizeof(t) == 1) os << static_cast<int16_t>(t) << " ";
    else os << t << " ";
  }
  if (os.fail()) throw std::runtime_error("Write failure in WriteBasicType.");
}
template <>
inline void WriteBasicType<bool>(std::ostream &os, bool, bool b) {
  os << (b ? "T" : "F") << " ";
}
template <>
inline void WriteBasicType<float>(std::ostream &os, bool binary, float f) {
  if (binary) {
    os.put(static_cast<char>(sizeof(f)));
    os.write(reinterpret_cast<const char *>(&f), sizeof(f));
  } else {
    os << f << " ";
  }
}
template <>
inline void WriteBasicType<double>(std::ostream &os, bool binary, double f) {
  if (binary) {
    os.put(static_cast<char>(sizeof(f)));
    os.write(reinterpret_cast<const char *>(&f), sizeof(f));
  } else {
    os << f << " ";
  }
}

template <class T>
inline void ReadBasicType(std::istream &is, bool binary, T *t) {
  static_assert(std::numeric_limits<T>::is_integer, "integer overload");
  if (binary) {
    int len_c_in = is.get();
    if (len_c_in == -1) ASLP_ERR << "ReadBasicType: encountered end of stream.";
    char len_c = static_cast<char>(len_c_in),
         len_c_expected = (std::numeric_limits<T>::is_signed ? 1 : -1) * static_cast<char>(sizeof(*t));
    if (len_c != len_c_expected)
      ASLP_ERR << "ReadBasicType: did not get expected integer type, " << static_cast<int>(len_c) << " vs. "
               << static_cast<int>(len_c_expected);
    is.read(reinterpret_cast<char *>(t), sizeof(*t));
  } else {
    if (sizeof(*t) == 1) {
      int16_t i;
      is >> i;
      *t = i;
    } else {
      is >> *t;
    }
  }
  if (is.fail()) ASLP_ERR << "Read failure in ReadBasicType, file position is " << is.tellg();
}
template <>
inline void ReadBasicType<double>(std::istream &is, bool binary, double *d);
template <>
inline void ReadBasicType<float>(std::istream &is, bool binary, float *f) {
  if (binary) {
    int c = is.peek();
    if (c == sizeof(*f)) {
      is.get();
      is.read(reinterpret_cast<char *>(f), sizeof(*f));
    } else if (c == sizeof(double)) {
      double d;
      is.get();
      is.read(reinterpret_cast<char *>(&d), sizeof(d));
      *f = d;
    } else {
      ASLP_ERR << "ReadBasicType: expected float, saw " << is.peek() << ", at file position " << is.tellg();
    }
  } else {
    is >> *f;
  }
  if (is.fail()) ASLP_ERR << "ReadBasicType: failed to read, at file position " << is.tellg();
}
template <>
inline void ReadBasicType<double>(std::istream &is, bool binary, double *d) {
  if (binary) {
    int c = is.peek();
    if (c == sizeof(*d)) {
      is.get();
      is.read(reinterpret_cast<char *>(d), sizeof(*d));
    } else if (c == sizeof(float)) {
      float f;
      is.get();
      is.read(reinterpret_cast<char *>(&f), sizeof(f));
      *d = f;
    } else {
      ASLP_ERR << "ReadBasicType: expected float, saw " << is.peek() << ", at file position " << is.tellg();
    }
  } else {
    is >> *d;
  }
  if (is.fail()) ASLP_ERR << "ReadBasicType: failed to read, at file position " << is.tellg();
}

template <class T>
inline void WriteIntegerVector(std::ostream &os, bool binary, const std::vector<T> &v) {
  if (binary) {
    char sz = sizeof(T);
    os.write(&sz, 1);
    int32 vecsz = static_cast<int32>(v.size());
    os.write(reinterpret_cast<const char *>(&vecsz), sizeof(vecsz));
    if (vecsz != 0) os.write(reinterpret_cast<const char *>(&(v[0])), sizeof(T) * vecsz);
  } else {
    os << "[ ";
    for (const T &x : v) os << x << " ";
    os << "]\n";
  }
  if (os.fail()) throw std::runtime_error("Write failure in WriteIntegerType.");
}
template <class T>
inline void ReadIntegerVector(std::istream &is, bool binary, std::vector<T> *v) {
  if (binary) {
    int sz = is.peek();
    if (sz == sizeof(T)) is.get();
    else ASLP_ERR << "ReadIntegerVector: expected to see type of size " << sizeof(T) << ", saw instead " << sz
                  << ", at file position " << is.tellg();
    int32 vecsz;
    is.read(reinterpret_cast<char *>(&vecsz), sizeof(vecsz));
    if (is.fail() || vecsz < 0) ASLP_ERR << "ReadIntegerVector: read failure at file position " << is.tellg();
    v->resize(vecsz);
    if (vecsz > 0) is.read(reinterpret_cast<char *>(&((*v)[0])), sizeof(T) * vecsz);
  } else {
    std::vector<T> tmp_v;
    is >> std::ws;
    if (is.peek() != static_cast<int>('['))
      ASLP_ERR << "ReadIntegerVector: expected to see [, saw " << is.peek() << ", at file position " << is.tellg();
    is.get();
    is >> std::ws;
    while (is.peek() != static_cast<int>(']')) {
      T next_t;
      is >> next_t >> std::ws;
      if (is.fail()) ASLP_ERR << "ReadIntegerVector: read failure at file position " << is.tellg();
      tmp_v.push_back(next_t);
    }
    is.get();
    *v = tmp_v;
  }
  if (is.fail()) ASLP_ERR << "ReadIntegerVector: read failure at file position " << is.tellg();
}

// text-utils.h subset
void SplitStringToVector(const std::string &full, const char *delim, bool omit_empty_strings, std::vector<std::string> *out);
bool SplitStringToIntegers(const std::string &full, const char *delim, bool omit_empty_strings, std::vector<int32> *out);
bool ConvertStringToInteger(const std::string &str, int32 *out);
bool ConvertStringToReal(const std::string &str, float *out);

// libc-rand based helpers with Kaldi's formulas (base/kaldi-math.h: RandUniform = (Rand()+1)/(RAND_MAX+2),
// RandGauss = sqrt(-2 log U1) cos(2 pi U2)).  The reference seeds with srand(seed) in aslp-nnet-init.
// The engine's private copy of the C library generator: the reference draws parameters and shuffle masks from libc
// rand() (seeded by srand(--seed) / srand(--randomizer-seed)), but inside a HIP process the global rand() state is not
// ours alone (the runtime draws from it while it loads code objects, measured), so the same glibc generator -- random_r
// on a 128-byte state, bit-identical sequence to srand(seed); rand() -- is kept in a state only the engine touches.
void SRand(unsigned seed);
int Rand();
float RandUniform();
float RandGauss();

}  // namespace aslp
