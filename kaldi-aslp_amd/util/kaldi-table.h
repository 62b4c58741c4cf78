// kaldi-table.h -- Kaldi Table readers / writers (ark, scp, pipes) for the types the training tools read and write:
// BaseFloatMatrix, BaseFloatVector, Posterior, Int32Vector.  Host-only, header-only on top of kaldi-io.
//
// Follows src/util/kaldi-table.{h,cc}, kaldi-table-inl.h, kaldi-holder-inl.h:
//   rspecifier  [opts,]ark:rxfilename | [opts,]scp:rxfilename     opts b t o no p np s ns cs ncs (kaldi-table.cc:212-293)
//   wspecifier  [opts,]ark:wx | [opts,]scp:wx | [opts,]ark,scp:ark_wx,scp_wx    opts b t f nf p np (kaldi-table.cc:132-210)
//   archive     "<key> " + object, objects carry their own "\0B" header in binary mode (kaldi-table-inl.h:358-403)
//   script      "<key> <rxfilename>" per line (kaldi-table.cc:49-78)
#pragma once
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "host-matrix.h"
#include "kaldi-io.h"
#include "posterior.h"

namespace aslp {

// ---- holders (kaldi-holder-inl.h) ----------------------------------------------------------------------
template <class Obj>
struct KaldiObjectHolder {  // :36-96 objects with Read(is, binary) / Write(os, binary)
  typedef Obj T;
  static bool Write(std::ostream &os, bool binary, const T &t) {
    InitKaldiOutputStream(os, binary);
    try { t.Write(os, binary); return os.good(); } catch (const std::exception &e) { ASLP_WARN << "Exception caught writing Table object. " << e.what(); return false; }
  }
  bool Read(std::istream &is) {
    bool binary;
    if (!InitKaldiInputStream(is, &binary)) { ASLP_WARN << "Reading Table object, failed reading binary header"; return false; }
    try { t_.Read(is, binary); return true; } catch (const std::exception &e) { ASLP_WARN << "Exception caught reading Table object. " << e.what(); return false; }
  }
  T &Value() { return t_; }
  void Clear() { t_ = T(); }
  T t_;
};
struct PosteriorHolder {  // hmm/posterior.cc:127-160
  typedef Posterior T;
  static bool Write(std::ostream &os, bool binary, const T &t) {
    InitKaldiOutputStream(os, binary);
    try { WritePosterior(os, binary, t); return true; } catch (const std::exception &e) { ASLP_WARN << "Exception caught writing table of posteriors. " << e.what(); return false; }
  }
  bool Read(std::istream &is) {
    t_.clear();
    bool binary;
    if (!InitKaldiInputStream(is, &binary)) { ASLP_WARN << "Reading Table object, failed reading binary header"; return false; }
    try { ReadPosterior(is, binary, &t_); return true; } catch (const std::exception &e) { ASLP_WARN << "Exception caught reading table of posteriors. " << e.what(); t_.clear(); return false; }
  }
  T &Value() { return t_; }
  void Clear() { t_.clear(); }
  T t_;
};
template <class B>
struct BasicVectorHolder {  // :191-288: binary = int32 count + every element as a basic type; text = "1 2 3\n"
  typedef std::vector<B> T;
  static bool Write(std::ostream &os, bool binary, const T &t) {
    InitKaldiOutputStream(os, binary);
    try {
      if (binary) WriteBasicType(os, binary, (int32)t.size());
      for (const B &b : t) WriteBasicType(os, binary, b);
      if (!binary) os << '\n';
      return os.good();
    } catch (const std::exception &e) { ASLP_WARN << "Exception caught writing Table object (BasicVector). " << e.what(); return false; }
  }
  bool Read(std::istream &is) {
    t_.clear();
    bool binary;
    if (!InitKaldiInputStream(is, &binary)) { ASLP_WARN << "Reading Table object [integer type], failed reading binary header"; return false; }
    try {
      if (!binary) {
        std::string line;
        std::getline(is, line);
        if (is.fail()) { ASLP_WARN << "BasicVectorHolder::Read, error reading line " << (is.eof() ? "[eof]" : ""); return false; }
        std::istringstream ls(line);
        while (true) {
          ls >> std::ws;
          if (ls.eof()) break;
          B b;
          ReadBasicType(ls, false, &b);
          t_.push_back(b);
        }
      } else {
        int32 size;
        ReadBasicType(is, true, &size);
        t_.resize(size);
        for (B &b : t_) ReadBasicType(is, true, &b);
      }
      return true;
    } catch (const std::exception &e) { ASLP_WARN << "BasicVectorHolder::Read, read error or unexpected data. " << e.what(); return false; }
  }
  T &Value() { return t_; }
  void Clear() { t_.clear(); }
  T t_;
};

template <class B>
struct BasicHolder {  // kaldi-holder-inl.h:99-189: one basic type; text form is "<value>\n"
  typedef B T;
  static bool Write(std::ostream &os, bool binary, const T &t) {
    InitKaldiOutputStream(os, binary);
    try { WriteBasicType(os, binary, t); if (!binary) os << '\n'; return os.good(); } catch (const std::exception &e) { ASLP_WARN << "Exception caught writing Table object. " << e.what(); return false; }
  }
  bool Read(std::istream &is) {
    bool binary;
    if (!InitKaldiInputStream(is, &binary)) { ASLP_WARN << "Reading Table object [integer type], failed reading binary header"; return false; }
    try {
      int c;
      while (isspace((c = is.peek())) && c != static_cast<int>('\n')) is.get();
      if (is.peek() == '\n') { ASLP_WARN << "Found newline but expected basic type."; return false; }
      ReadBasicType(is, binary, &t_);
      while (isspace((c = is.peek())) && c != static_cast<int>('\n')) is.get();
      if (is.peek() == '\n') is.get();
      else if (!binary) { ASLP_WARN << "BasicHolder::Read, expected newline, got " << is.peek(); return false; }
      return true;
    } catch (const std::exception &e) { ASLP_WARN << "Exception caught reading Table object. " << e.what(); return false; }
  }
  T &Value() { return t_; }
  void Clear() {}
  T t_ = T();
};

// ---- specifiers -------------------------------------------------------------------------------------------
enum RspecifierType { kNoRspecifier, kArchiveRspecifier, kScriptRspecifier };
struct RspecifierOptions { bool once = false, sorted = false, called_sorted = false, permissive = false; };
enum WspecifierType { kNoWspecifier, kArchiveWspecifier, kScriptWspecifier, kBothWspecifier };
struct WspecifierOptions { bool binary = true, flush = false, permissive = false; };
RspecifierType ClassifyRspecifier(const std::string &rspecifier, std::string *rxfilename, RspecifierOptions *opts);
WspecifierType ClassifyWspecifier(const std::string &wspecifier, std::string *archive_wxfilename, std::string *script_wxfilename,
                                  WspecifierOptions *opts);
bool ReadScriptFile(const std::string &rxfilename, bool warn, std::vector<std::pair<std::string, std::string>> *script_out);
bool IsToken(const std::string &token);  // non-empty, printable, no whitespace (text-utils.cc)

// ---- SequentialTableReader (kaldi-table-inl.h: archive :247-530, script :65-245) ------------------------
template <class Holder>
class SequentialTableReader {
 public:
  typedef typename Holder::T T;
  SequentialTableReader() {}
  explicit SequentialTableReader(const std::string &rspecifier) {
    if (!Open(rspecifier)) ASLP_ERR << "Error constructing TableReader: rspecifier is " << rspecifier;
  }
  bool Open(const std::string &rspecifier) {
    rspecifier_ = rspecifier;
    type_ = ClassifyRspecifier(rspecifier, &rxfilename_, &opts_);
    if (type_ == kNoRspecifier) { ASLP_WARN << "Invalid rspecifier " << rspecifier; return false; }
    if (type_ == kArchiveRspecifier) {
      if (!input_.Open(rxfilename_)) { ASLP_WARN << "Failed to open stream " << PrintableRxfilename(rxfilename_); return false; }
    } else {
      if (!ReadScriptFile(rxfilename_, true, &script_)) return false;
      pos_ = (size_t)-1;
    }
    open_ = true;
    Next();
    return !error_ || opts_.permissive;
  }
  bool IsOpen() const { return open_; }
  bool Done() const { return !have_; }
  const std::string &Key() const { if (!have_) ASLP_ERR << "Key() called when there is no current object"; return key_; }
  T &Value() { if (!have_) ASLP_ERR << "Value() called when there is no current object"; return holder_.Value(); }
  void Next() {
    have_ = false;
    holder_.Clear();
    if (type_ == kArchiveRspecifier) NextArchive();
    else NextScript();
    if (error_ && !opts_.permissive) ASLP_ERR << "TableReader: error reading " << rspecifier_;
  }
  // false if reading stopped on an error (or a pipe returned non-zero)
  bool Close() {
    int status = input_.Close();
    open_ = false;
    have_ = false;
    return !error_ && status == 0;
  }

 private:
  void NextArchive() {
    std::istream &is = input_.Stream();
    is.clear();
    is >> key_;
    if (is.eof()) return;
    if (is.fail()) { ASLP_WARN << "Error reading archive " << PrintableRxfilename(rxfilename_); error_ = true; return; }
    int c = is.peek();
    if (c != ' ' && c != '\t' && c != '\n') {
      ASLP_WARN << "Invalid archive file format: expected space after key " << key_ << ", reading " << PrintableRxfilename(rxfilename_);
      error_ = true;
      return;
    }
    if (c != '\n') is.get();
    if (holder_.Read(is)) have_ = true;
    else { ASLP_WARN << "Object read failed, reading archive " << PrintableRxfilename(rxfilename_); error_ = true; }
  }
  void NextScript() {
    while (true) {
      pos_++;
      if (pos_ >= script_.size()) return;
      key_ = script_[pos_].first;
      const std::string &rx = script_[pos_].second;
      if (!input_.Open(rx)) {
        ASLP_WARN << "Failed to open file " << PrintableRxfilename(rx);
        if (opts_.permissive) continue;
        error_ = true;
        return;
      }
      if (holder_.Read(input_.Stream())) { have_ = true; return; }
      ASLP_WARN << "Failed to load object from " << PrintableRxfilename(rx);
      if (opts_.permissive) continue;
      error_ = true;
      return;
    }
  }
  std::string rspecifier_, rxfilename_, key_;
  RspecifierType type_ = kNoRspecifier;
  RspecifierOptions opts_;
  Input input_;
  std::vector<std::pair<std::string, std::string>> script_;
  size_t pos_ = 0;
  Holder holder_;
  bool open_ = false, have_ = false, error_ = false;
};

// ---- RandomAccessTableReader -----------------------------------------------------------------------------
// Script: key -> rxfilename map, objects opened on demand (kaldi-table-inl.h:1090-1330).  Archive: the unsorted
// implementation (:1830-2050): reads forward until the key turns up and keeps everything it passed.
template <class Holder>
class RandomAccessTableReader {
 public:
  typedef typename Holder::T T;
  RandomAccessTableReader() {}
  explicit RandomAccessTableReader(const std::string &rspecifier) {
    if (!Open(rspecifier)) ASLP_ERR << "Error opening RandomAccessTableReader object  (rspecifier is: " << rspecifier << ")";
  }
  bool Open(const std::string &rspecifier) {
    rspecifier_ = rspecifier;
    type_ = ClassifyRspecifier(rspecifier, &rxfilename_, &opts_);
    if (type_ == kNoRspecifier) { ASLP_WARN << "Invalid rspecifier: " << rspecifier; return false; }
    if (type_ == kArchiveRspecifier) {
      if (!input_.Open(rxfilename_)) { ASLP_WARN << "Failed to open stream " << PrintableRxfilename(rxfilename_); return false; }
    } else {
      std::vector<std::pair<std::string, std::string>> script;
      if (!ReadScriptFile(rxfilename_, true, &script)) return false;
      for (auto &kv : script) script_map_[kv.first] = kv.second;
    }
    open_ = true;
    return true;
  }
  bool IsOpen() const { return open_; }
  bool HasKey(const std::string &key) {
    if (type_ == kScriptRspecifier) {
      auto it = script_map_.find(key);
      if (it == script_map_.end()) return false;
      if (opts_.permissive) return LoadScriptObject(key, it->second);  // permissive: a broken entry counts as absent
      return true;
    }
    return FindInArchive(key);
  }
  const T &Value(const std::string &key) {
    if (type_ == kScriptRspecifier) {
      auto it = script_map_.find(key);
      if (it == script_map_.end()) ASLP_ERR << "Value() called but no such key " << key << " in " << rspecifier_;
      if (!(cur_valid_ && cur_key_ == key) && !LoadScriptObject(key, it->second))
        ASLP_ERR << "Failed to load object from " << PrintableRxfilename(it->second) << " (to suppress this error, add the permissive (p, ) option to the rspecifier.";
      return cur_.Value();
    }
    if (!FindInArchive(key)) ASLP_ERR << "Value() called but no such key " << key << " in archive " << PrintableRxfilename(rxfilename_);
    return *map_[key];
  }
  bool Close() { int st = input_.Close(); open_ = false; return st == 0 && !error_; }

 private:
  bool LoadScriptObject(const std::string &key, const std::string &rx) {
    if (cur_valid_ && cur_key_ == key) return true;
    cur_valid_ = false;
    if (!input_.Open(rx)) { ASLP_WARN << "Failed to open file " << PrintableRxfilename(rx); return false; }
    if (!cur_.Read(input_.Stream())) { ASLP_WARN << "Failed to load object from " << PrintableRxfilename(rx); return false; }
    cur_key_ = key;
    cur_valid_ = true;
    return true;
  }
  bool FindInArchive(const std::string &key) {
    // "s, cs" (archive sorted by key, keys asked for in sorted order: kaldi-table.h's sorted random-access mode): nothing
    // before the key being asked for can be asked for again, so it is released instead of kept for the whole run
    if (opts_.sorted && opts_.called_sorted && !map_.empty()) {
      for (auto it = map_.begin(); it != map_.end();) {
        if (it->first < key) it = map_.erase(it);
        else ++it;
      }
    }
    if (map_.count(key)) return true;
    if (opts_.sorted && !last_key_read_.empty() && key < last_key_read_) return false;  // sorted archive: already past it
    while (!eof_) {
      std::istream &is = input_.Stream();
      std::string k;
      is.clear();
      is >> k;
      if (is.eof()) { eof_ = true; break; }
      if (is.fail()) { error_ = true; ASLP_ERR << "Error reading archive " << PrintableRxfilename(rxfilename_); }
      int c = is.peek();
      if (c != ' ' && c != '\t' && c != '\n') { error_ = true; ASLP_ERR << "Invalid archive file format: expected space after key " << k << ", reading archive " << PrintableRxfilename(rxfilename_); }
      if (c != '\n') is.get();
      Holder h;
      if (!h.Read(is)) {
        error_ = true;
        if (opts_.permissive) { ASLP_WARN << "Object read failed, reading archive " << PrintableRxfilename(rxfilename_) << "; stopping there (permissive)"; eof_ = true; break; }
        ASLP_ERR << "Object read failed, reading archive " << PrintableRxfilename(rxfilename_);
      }
      if (map_.count(k)) ASLP_ERR << "Error in RandomAccessTableReader: duplicate key " << k << " in archive " << PrintableRxfilename(rxfilename_);
      map_[k].reset(new T(std::move(h.Value())));
      last_key_read_ = k;
      if (k == key) return true;
    }
    return false;
  }
  std::string rspecifier_, rxfilename_;
  RspecifierType type_ = kNoRspecifier;
  RspecifierOptions opts_;
  Input input_;
  std::unordered_map<std::string, std::string> script_map_;
  std::unordered_map<std::string, std::unique_ptr<T>> map_;
  Holder cur_;
  std::string cur_key_, last_key_read_;
  bool cur_valid_ = false, open_ = false, eof_ = false, error_ = false;
};

// ---- TableWriter (kaldi-table-inl.h:535-1040) ---------------------------------------------------------------
template <class Holder>
class TableWriter {
 public:
  typedef typename Holder::T T;
  TableWriter() {}
  explicit TableWriter(const std::string &wspecifier) {
    if (!Open(wspecifier)) ASLP_ERR << "TableWriter: failed to write to " << wspecifier;
  }
  bool Open(const std::string &wspecifier) {
    wspecifier_ = wspecifier;
    type_ = ClassifyWspecifier(wspecifier, &archive_wx_, &script_wx_, &opts_);
    if (type_ == kNoWspecifier) { ASLP_WARN << "Invalid wspecifier " << wspecifier; return false; }
    if (type_ == kScriptWspecifier) {
      std::vector<std::pair<std::string, std::string>> script;
      if (!ReadScriptFile(script_wx_, true, &script)) return false;  // scp: the list of files to write to
      for (auto &kv : script) script_map_[kv.first] = kv.second;
      return open_ = true;
    }
    if (type_ == kBothWspecifier && ClassifyWxfilename(archive_wx_) != kFileOutput) {
      ASLP_WARN << "When writing to both archive and script, archive must be a real file (cannot give offsets into a pipe)";
      return false;
    }
    if (!archive_.Open(archive_wx_, opts_.binary, false)) { ASLP_WARN << "Failed to open stream " << PrintableWxfilename(archive_wx_); return false; }
    if (type_ == kBothWspecifier && !script_.Open(script_wx_, false, false)) {
      ASLP_WARN << "Failed to open script file " << PrintableWxfilename(script_wx_);
      return false;
    }
    return open_ = true;
  }
  bool IsOpen() const { return open_; }
  void Write(const std::string &key, const T &value) {
    if (!open_) ASLP_ERR << "TableWriter: Write called on invalid stream";
    if (!IsToken(key)) ASLP_ERR << "Using invalid key " << key;
    bool ok;
    if (type_ == kScriptWspecifier) {
      auto it = script_map_.find(key);
      if (it == script_map_.end()) { ASLP_WARN << "TableWriter: key " << key << " not in the script file"; ok = false; }
      else { Output out; ok = out.Open(it->second, opts_.binary, false) && Holder::Write(out.Stream(), opts_.binary, value) && out.Close(); }
    } else {
      std::ostream &os = archive_.Stream();
      os << key << ' ';
      if (type_ == kBothWspecifier) {
        const std::streampos off = os.tellp();
        script_.Stream() << key << ' ' << archive_wx_ << ':' << (long long)off << '\n';
      }
      ok = Holder::Write(os, opts_.binary, value);
      if (opts_.flush) { os.flush(); if (type_ == kBothWspecifier) script_.Stream().flush(); }
    }
    if (!ok) {
      if (opts_.permissive) ASLP_WARN << "Write failure to " << wspecifier_ << " (ignored: permissive)";
      else ASLP_ERR << "Write failure to " << wspecifier_;
    }
  }
  void Flush() { if (archive_.IsOpen()) archive_.Stream().flush(); }
  bool Close() {
    bool ok = true;
    if (archive_.IsOpen()) ok = archive_.Close() && ok;
    if (script_.IsOpen()) ok = script_.Close() && ok;
    open_ = false;
    return ok;
  }
  ~TableWriter() { if (open_ && !Close()) std::cerr << "WARNING: error closing TableWriter " << wspecifier_ << std::endl; }

 private:
  std::string wspecifier_, archive_wx_, script_wx_;
  WspecifierType type_ = kNoWspecifier;
  WspecifierOptions opts_;
  Output archive_, script_;
  std::unordered_map<std::string, std::string> script_map_;
  bool open_ = false;
};

// util/table-types.h
typedef KaldiObjectHolder<HostMatrix> BaseFloatMatrixHolder;
typedef KaldiObjectHolder<HostVector> BaseFloatVectorHolder;
typedef SequentialTableReader<BaseFloatMatrixHolder> SequentialBaseFloatMatrixReader;
typedef RandomAccessTableReader<BaseFloatMatrixHolder> RandomAccessBaseFloatMatrixReader;
typedef TableWriter<BaseFloatMatrixHolder> BaseFloatMatrixWriter;
typedef SequentialTableReader<BaseFloatVectorHolder> SequentialBaseFloatVectorReader;
typedef RandomAccessTableReader<BaseFloatVectorHolder> RandomAccessBaseFloatVectorReader;
typedef TableWriter<BaseFloatVectorHolder> BaseFloatVectorWriter;
typedef RandomAccessTableReader<BasicHolder<BaseFloat>> RandomAccessBaseFloatReader;
typedef SequentialTableReader<PosteriorHolder> SequentialPosteriorReader;
typedef RandomAccessTableReader<PosteriorHolder> RandomAccessPosteriorReader;
typedef TableWriter<PosteriorHolder> PosteriorWriter;
typedef SequentialTableReader<BasicVectorHolder<int32>> SequentialInt32VectorReader;
typedef RandomAccessTableReader<BasicVectorHolder<int32>> RandomAccessInt32VectorReader;
typedef TableWriter<BasicVectorHolder<int32>> Int32VectorWriter;

}  // namespace aslp
