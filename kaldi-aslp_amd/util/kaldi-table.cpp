// kaldi-table.cpp -- specifier parsing and script files (src/util/kaldi-table.cc).  Host-only.
#include "kaldi-table.h"

#include <cctype>
#include <cstring>

namespace aslp {

bool IsToken(const std::string &token) {  // text-utils.cc: non-empty, all printable, no space
  if (token.empty()) return false;
  for (unsigned char c : token)
    if (!isprint(c) || isspace(c)) return false;
  return true;
}

static void SplitStringOnFirstSpace(const std::string &str, std::string *first, std::string *rest) {  // text-utils.cc
  const char *white = " \t\n\r\f\v";
  size_t first_nonwhite = str.find_first_not_of(white);
  if (first_nonwhite == std::string::npos) { first->clear(); rest->clear(); return; }
  size_t next_white = str.find_first_of(white, first_nonwhite);
  if (next_white == std::string::npos) { *first = std::string(str, first_nonwhite); rest->clear(); return; }
  size_t next_nonwhite = str.find_first_not_of(white, next_white);
  if (next_nonwhite == std::string::npos) { *first = std::string(str, first_nonwhite, next_white - first_nonwhite); rest->clear(); return; }
  size_t last_nonwhite = str.find_last_not_of(white);
  *first = std::string(str, first_nonwhite, next_white - first_nonwhite);
  *rest = std::string(str, next_nonwhite, last_nonwhite + 1 - next_nonwhite);
}

bool ReadScriptFile(const std::string &rxfilename, bool warn, std::vector<std::pair<std::string, std::string>> *script_out) {
  bool is_binary;
  Input input;
  if (!input.Open(rxfilename, &is_binary)) {
    if (warn) ASLP_WARN << "Error opening script file: " << PrintableRxfilename(rxfilename);
    return false;
  }
  if (is_binary) {
    if (warn) ASLP_WARN << "Error: script file appears to be binary: " << PrintableRxfilename(rxfilename);
    return false;
  }
  std::string line;
  int line_number = 0;
  while (std::getline(input.Stream(), line)) {
    line_number++;
    if (line.empty()) {
      if (warn) ASLP_WARN << "Empty " << line_number << "'th line in script file " << PrintableRxfilename(rxfilename);
      return false;
    }
    std::string key, rest;
    SplitStringOnFirstSpace(line, &key, &rest);
    if (key.empty() || rest.empty()) {
      if (warn) ASLP_WARN << "Invalid " << line_number << "'th line in script file:\"" << line << '"';
      return false;
    }
    script_out->push_back(std::make_pair(key, rest));
  }
  return true;
}

RspecifierType ClassifyRspecifier(const std::string &rspecifier, std::string *rxfilename, RspecifierOptions *opts) {
  if (rxfilename) rxfilename->clear();
  if (opts) *opts = RspecifierOptions();
  size_t pos = rspecifier.find(':');
  if (pos == std::string::npos) return kNoRspecifier;
  if (isspace(*(rspecifier.rbegin()))) return kNoRspecifier;
  std::string before_colon(rspecifier, 0, pos), after_colon(rspecifier, pos + 1);
  std::vector<std::string> parts;
  SplitStringToVector(before_colon, ", ", false, &parts);
  RspecifierType rs = kNoRspecifier;
  for (const std::string &str : parts) {
    const char *c = str.c_str();
    if (!strcmp(c, "b") || !strcmp(c, "t")) {
    } else if (!strcmp(c, "o")) { if (opts) opts->once = true;
    } else if (!strcmp(c, "no")) { if (opts) opts->once = false;
    } else if (!strcmp(c, "p")) { if (opts) opts->permissive = true;
    } else if (!strcmp(c, "np")) { if (opts) opts->permissive = false;
    } else if (!strcmp(c, "s")) { if (opts) opts->sorted = true;
    } else if (!strcmp(c, "ns")) { if (opts) opts->sorted = false;
    } else if (!strcmp(c, "cs")) { if (opts) opts->called_sorted = true;
    } else if (!strcmp(c, "ncs")) { if (opts) opts->called_sorted = false;
    } else if (!strcmp(c, "ark")) { if (rs == kNoRspecifier) rs = kArchiveRspecifier; else return kNoRspecifier;
    } else if (!strcmp(c, "scp")) { if (rs == kNoRspecifier) rs = kScriptRspecifier; else return kNoRspecifier;
    } else return kNoRspecifier;
  }
  if (rs != kNoRspecifier && rxfilename) *rxfilename = after_colon;
  return rs;
}

WspecifierType ClassifyWspecifier(const std::string &wspecifier, std::string *archive_wxfilename, std::string *script_wxfilename,
                                  WspecifierOptions *opts) {  // kaldi-table.cc:132-210
  if (archive_wxfilename) archive_wxfilename->clear();
  if (script_wxfilename) script_wxfilename->clear();
  if (opts) *opts = WspecifierOptions();
  size_t pos = wspecifier.find(':');
  if (pos == std::string::npos) return kNoWspecifier;
  if (isspace(*(wspecifier.rbegin()))) return kNoWspecifier;
  std::string before_colon(wspecifier, 0, pos), after_colon(wspecifier, pos + 1);
  std::vector<std::string> parts;
  SplitStringToVector(before_colon, ", ", false, &parts);
  WspecifierType ws = kNoWspecifier;
  for (const std::string &str : parts) {
    const char *c = str.c_str();
    if (!strcmp(c, "b")) { if (opts) opts->binary = true;
    } else if (!strcmp(c, "f")) { if (opts) opts->flush = true;
    } else if (!strcmp(c, "nf")) { if (opts) opts->flush = false;
    } else if (!strcmp(c, "t")) { if (opts) opts->binary = false;
    } else if (!strcmp(c, "p")) { if (opts) opts->permissive = true;
    } else if (!strcmp(c, "ark")) { if (ws == kNoWspecifier) ws = kArchiveWspecifier; else return kNoWspecifier;  // "scp,ark" is not allowed
    } else if (!strcmp(c, "scp")) {
      if (ws == kNoWspecifier) ws = kScriptWspecifier;
      else if (ws == kArchiveWspecifier) ws = kBothWspecifier;
      else return kNoWspecifier;
    } else return kNoWspecifier;
  }
  switch (ws) {
    case kArchiveWspecifier: if (archive_wxfilename) *archive_wxfilename = after_colon; break;
    case kScriptWspecifier: if (script_wxfilename) *script_wxfilename = after_colon; break;
    case kBothWspecifier: {
      pos = after_colon.find(',');
      if (pos == std::string::npos) return kNoWspecifier;
      if (archive_wxfilename) *archive_wxfilename = std::string(after_colon, 0, pos);
      if (script_wxfilename) *script_wxfilename = std::string(after_colon, pos + 1);
      break;
    }
    default: break;
  }
  return ws;
}

}  // namespace aslp
