// Native byte-pair-encoding core of the CLIP tokenizer (host C++, no GPU work): the merge loop of
// SimpleTokenizer.bpe (jclip/simple_tokenizer.py:88-129) over integer symbol ids.
//
// Vocabulary ids (simple_tokenizer.py:72-78): [0, 256) the byte alphabet in bytes_to_unicode() order, [256, 512) the same
// symbols with the end-of-word marker "</w>", 512 + r the symbol created by merge r, then the two specials.  A word
// arrives as its raw UTF-8 bytes (the Python side does the clean-up, lower-casing and the Unicode-class regex split);
// it starts as one symbol per byte, the last one carrying "</w>", and the adjacent pair with the lowest merge rank is
// merged -- every occurrence, left to right -- until no ranked pair is left.  Output: the ids, ready for clip.tokenize.
#include "common.h"

#include <stdint.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Bpe {
  int byte_id[256];                               // byte value -> id of its plain symbol
  std::unordered_map<uint64_t, int> rank;        // (left id << 32 | right id) -> merge rank
  int n_merges = 0;
};

// the byte alphabet order of bytes_to_unicode(): printable latin-1 bytes first, then the remaining 68
void byte_order(int (&id_of)[256], std::vector<std::string>& sym) {
  bool printable[256] = {false};
  for (int b = 0x21; b < 0x7F; ++b) printable[b] = true;
  for (int b = 0xA1; b < 0xAD; ++b) printable[b] = true;
  for (int b = 0xAE; b < 0x100; ++b) printable[b] = true;
  auto utf8 = [](int cp) {
    std::string s;
    if (cp < 0x80) {
      s += (char)cp;
    } else if (cp < 0x800) {
      s += (char)(0xC0 | (cp >> 6));
      s += (char)(0x80 | (cp & 0x3F));
    } else {
      s += (char)(0xE0 | (cp >> 12));
      s += (char)(0x80 | ((cp >> 6) & 0x3F));
      s += (char)(0x80 | (cp & 0x3F));
    }
    return s;
  };
  int next = 0;
  for (int b = 0; b < 256; ++b)
    if (printable[b]) {
      id_of[b] = next++;
      sym.push_back(utf8(b));
    }
  int shift = 0;
  for (int b = 0; b < 256; ++b)
    if (!printable[b]) {
      id_of[b] = next++;
      sym.push_back(utf8(256 + shift++));
    }
}

}  // namespace

// merges: the text of the merges file AFTER its header line ("left right\n" per merge), n_merges lines are used.
// Returns an opaque handle (NULL on a malformed file; clipfs_last_error says why).
extern "C" void* clipfs_bpe_create(const char* merges, size_t n_bytes, int n_merges) {
  if (!merges || n_merges <= 0) {
    clipfs::set_error("bpe_create: no merges");
    return nullptr;
  }
  Bpe* t = new Bpe();
  std::vector<std::string> sym;
  byte_order(t->byte_id, sym);
  std::unordered_map<std::string, int> id_of;
  id_of.reserve((size_t)n_merges * 2 + 1024);
  for (int i = 0; i < 256; ++i) {
    id_of[sym[i]] = i;
    id_of[sym[i] + "</w>"] = 256 + i;
  }
  size_t pos = 0;
  for (int r = 0; r < n_merges; ++r) {
    size_t eol = pos;
    while (eol < n_bytes && merges[eol] != '\n') ++eol;
    std::string line(merges + pos, eol - pos);
    pos = eol + 1;
    const size_t sp = line.find(' ');
    if (sp == std::string::npos || sp == 0 || sp + 1 >= line.size()) {
      clipfs::set_error("bpe_create: merge line %d is not 'left right'", r);
      delete t;
      return nullptr;
    }
    const std::string a = line.substr(0, sp), b = line.substr(sp + 1);
    auto ia = id_of.find(a), ib = id_of.find(b);
    if (ia == id_of.end() || ib == id_of.end()) {
      clipfs::set_error("bpe_create: merge %d uses a symbol no earlier merge produced", r);
      delete t;
      return nullptr;
    }
    t->rank[((uint64_t)(uint32_t)ia->second << 32) | (uint32_t)ib->second] = r;
    id_of[a + b] = 512 + r;
  }
  t->n_merges = n_merges;
  return t;
}

extern "C" void clipfs_bpe_destroy(void* handle) { delete static_cast<Bpe*>(handle); }

// words: concatenated UTF-8 bytes, word w = bytes [offsets[w], offsets[w+1]).  ids: capacity `cap` in total; counts[w] =
// ids of word w (written consecutively).  Returns the total number of ids, or -1 when `cap` is too small / bad arguments.
extern "C" long clipfs_bpe_encode(const void* handle, const uint8_t* words, const int32_t* offsets, int n_words, int32_t* ids,
                                  int32_t* counts, long cap) {
  const Bpe* t = static_cast<const Bpe*>(handle);
  if (!t || !words || !offsets || !ids || !counts || n_words < 0) {
    clipfs::set_error("bpe_encode: bad arguments");
    return -1;
  }
  long total = 0;
  std::vector<int> s, o;
  for (int w = 0; w < n_words; ++w) {
    const int lo = offsets[w], hi = offsets[w + 1];
    if (hi <= lo) {
      counts[w] = 0;
      continue;
    }
    s.clear();
    for (int i = lo; i < hi; ++i) s.push_back(t->byte_id[words[i]]);
    s.back() += 256;  // the last byte carries "</w>"
    while (s.size() > 1) {
      int best = t->n_merges, ba = 0, bb = 0;
      for (size_t i = 0; i + 1 < s.size(); ++i) {
        auto it = t->rank.find(((uint64_t)(uint32_t)s[i] << 32) | (uint32_t)s[i + 1]);
        if (it != t->rank.end() && it->second < best) {
          best = it->second;
          ba = s[i];
          bb = s[i + 1];
        }
      }
      if (best == t->n_merges) break;
      o.clear();
      for (size_t i = 0; i < s.size();) {
        if (i + 1 < s.size() && s[i] == ba && s[i + 1] == bb) {
          o.push_back(512 + best);
          i += 2;
        } else {
          o.push_back(s[i]);
          i += 1;
        }
      }
      s.swap(o);
    }
    if (total + (long)s.size() > cap) {
      clipfs::set_error("bpe_encode: output capacity %ld too small", cap);
      return -1;
    }
    for (size_t i = 0; i < s.size(); ++i) ids[total + (long)i] = s[i];
    counts[w] = (int32_t)s.size();
    total += (long)s.size();
  }
  return total;
}
