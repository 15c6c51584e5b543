// Host front-end: Go regexp/syntax semantics (Parse(Perl) -> Simplify -> Compile) producing the
// same syntax.Prog the reference hands to its emitters (/root/reference/regengo.go:92-104).
// regexp/syntax is Go stdlib (not in the reference tree); this is a from-scratch C++ statement of
// its published behaviour, pinned by tests/golden/progs.json (Progs recovered from the reference's
// checked-in generated matchers).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace rgx {

constexpr int32_t kMaxRune = 0x10FFFF;

enum Flags : uint32_t {
  kFoldCase = 1, kLiteral = 2, kClassNL = 4, kDotNL = 8, kOneLine = 16, kNonGreedy = 32, kPerlX = 64,
  kUnicodeGroups = 128, kWasDollar = 256, kSimple = 512,
  kPerl = kClassNL | kOneLine | kPerlX | kUnicodeGroups,
};

enum Op : int {
  OpNoMatch = 1, OpEmptyMatch, OpLiteral, OpCharClass, OpAnyCharNotNL, OpAnyChar, OpBeginLine, OpEndLine,
  OpBeginText, OpEndText, OpWordBoundary, OpNoWordBoundary, OpCapture, OpStar, OpPlus, OpQuest, OpRepeat,
  OpConcat, OpAlternate,
  OpPseudo = 128, OpLeftParen = 128, OpVerticalBar = 129,
};

struct Regexp;
using RegexpPtr = std::shared_ptr<Regexp>;
struct Regexp {
  int op = 0;
  uint32_t flags = 0;
  std::vector<RegexpPtr> sub;
  std::vector<int32_t> rune;
  int min = 0, max = 0, cap = 0;
  std::string name;
  bool Equal(const Regexp* y) const;
  std::string Dump() const;
};

enum InstOp : uint8_t {
  InstAlt = 0, InstAltMatch, InstCapture, InstEmptyWidth, InstMatch, InstFail, InstNop, InstRune, InstRune1,
  InstRuneAny, InstRuneAnyNotNL,
};
enum EmptyOp : uint32_t {
  EmptyBeginLine = 1, EmptyEndLine = 2, EmptyBeginText = 4, EmptyEndText = 8, EmptyWordBoundary = 16,
  EmptyNoWordBoundary = 32,
};

struct Inst {
  InstOp op = InstFail;
  uint32_t out = 0, arg = 0;
  std::vector<int32_t> rune;
};
struct Prog {
  std::vector<Inst> inst;
  int start = 0;
  int numcap = 2;
  bool ascii_text = false;     // set by the table builders for kFlagAsciiText (rgx_dfa.h): classes are cut down to their bytes < 0x80
  std::string Dump() const;
};

struct SyntaxError {
  std::string msg;
};

RegexpPtr Parse(const std::string& pattern_utf8, uint32_t flags);  // throws SyntaxError
RegexpPtr Simplify(const RegexpPtr& re);
Prog Compile(const RegexpPtr& re);
std::vector<std::string> CaptureNames(const RegexpPtr& re);

// analysis_match_len.go:34-251
int MinMatchLen(const Regexp* re);
int MaxMatchLen(const Regexp* re);
// analysis.go:335-369, 168-207, 316-330, 117-124
bool DetectNestedQuantifiers(const Regexp* re, int depth = 0);
bool DetectComplexity(const Prog& p);
bool HasEndAnchor(const Prog& p);
bool IsAnchored(const Prog& p);

int32_t SimpleFold(int32_t r);
void SimpleFoldTable(std::vector<int32_t>* out);   // (r, SimpleFold(r)) pairs of every non-trivial orbit, sorted by r
// \p{name} range table (pairs), false for unknown names; UCD version 0xMMmmpp of the tables (rgx_unicode_tables.inc)
bool UnicodeTable(const std::string& name, std::vector<int32_t>* out);
int UnicodeVersion();
int RuneLen(int32_t r);
int EncodeRune(int32_t r, uint8_t out[4]);

}  // namespace rgx
