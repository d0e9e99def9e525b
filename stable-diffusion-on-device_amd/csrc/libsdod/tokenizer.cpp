#include "tokenizer.h"

#include <fstream>
#include <limits>
#include <stdexcept>

namespace sdod {
namespace {

void append_utf8(std::string& s, uint32_t cp) {
    if (cp < 0x80) {
        s.push_back((char)cp);
    } else if (cp < 0x800) {
        s.push_back((char)(0xC0 | (cp >> 6)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
        s.push_back((char)(0xE0 | (cp >> 12)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        s.push_back((char)(0xF0 | (cp >> 18)));
        s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

// strict UTF-8 decoder; throws on malformed input (the reference reports "Invalid UTF-8 string", tokenizer.cpp:77)
std::vector<uint32_t> decode_utf8(const std::string& s) {
    std::vector<uint32_t> out;
    out.reserve(s.size());
    size_t i = 0;
    while (i < s.size()) {
        const unsigned char c = (unsigned char)s[i];
        uint32_t cp;
        int extra;
        if (c < 0x80) { cp = c; extra = 0; }
        else if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; extra = 1; }
        else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; extra = 2; }
        else if ((c & 0xF8) == 0xF0) { cp = c & 0x07; extra = 3; }
        else throw std::invalid_argument("Invalid UTF-8 string");
        if (i + extra >= s.size() + (extra ? 0 : 1)) throw std::invalid_argument("Invalid UTF-8 string");
        for (int k = 1; k <= extra; ++k) {
            const unsigned char cc = (unsigned char)s[i + k];
            if ((cc & 0xC0) != 0x80) throw std::invalid_argument("Invalid UTF-8 string");
            cp = (cp << 6) | (cc & 0x3F);
        }
        out.push_back(cp);
        i += extra + 1;
    }
    return out;
}

bool is_space(uint32_t c) {
    return c == ' ' || (c >= 0x09 && c <= 0x0D) || c == 0x1C || c == 0x1D || c == 0x1E || c == 0x1F || c == 0x85 || c == 0xA0 ||
           c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

bool is_digit(uint32_t c) {
    if (c >= '0' && c <= '9') return true;
    return c == 0xB2 || c == 0xB3 || c == 0xB9 || (c >= 0xBC && c <= 0xBE) || (c >= 0x0660 && c <= 0x0669) ||
           (c >= 0x06F0 && c <= 0x06F9) || (c >= 0x0966 && c <= 0x096F) || (c >= 0xFF10 && c <= 0xFF19);
}

// letters: ASCII exactly; beyond ASCII every code point that is not whitespace, a digit, or in a block of
// punctuation / symbols counts as a letter (approximation of \p{L}; exact for Latin, Greek, Cyrillic, CJK text)
bool is_letter(uint32_t c) {
    if (c < 0x80) return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
    if (is_space(c) || is_digit(c)) return false;
    if (c <= 0xBF) return c == 0xAA || c == 0xB5 || c == 0xBA;
    if (c == 0xD7 || c == 0xF7) return false;
    if (c >= 0x2000 && c <= 0x2BFF) return false; // punctuation, arrows, math, symbols
    if (c >= 0x3000 && c <= 0x303F) return false;
    if (c >= 0xFE30 && c <= 0xFE6F) return false;
    if ((c >= 0xFF00 && c <= 0xFF0F) || (c >= 0xFF1A && c <= 0xFF20) || (c >= 0xFF3B && c <= 0xFF40) || (c >= 0xFF5B && c <= 0xFF65))
        return false;
    if (c >= 0x1F000 && c <= 0x1FAFF) return false; // emoji & pictographs
    return true;
}

uint32_t to_lower(uint32_t c) {
    if (c >= 'A' && c <= 'Z') return c + 32;
    if (c < 0x80) return c;
    if (c >= 0xC0 && c <= 0xDE && c != 0xD7) return c + 32;
    if (c >= 0x100 && c <= 0x17F) { // Latin Extended-A: mostly even = upper, odd = lower
        if ((c >= 0x139 && c <= 0x148) || (c >= 0x179 && c <= 0x17E)) return (c & 1) ? c + 1 : c;
        if (c == 0x130 || c == 0x131 || c == 0x138 || c == 0x149 || c == 0x17F) return c;
        if (c == 0x178) return 0xFF;
        return (c & 1) ? c : c + 1;
    }
    if (c >= 0x391 && c <= 0x3A9 && c != 0x3A2) return c + 32;
    if (c >= 0x410 && c <= 0x42F) return c + 32;
    if (c >= 0x400 && c <= 0x40F) return c + 80;
    return c;
}

} // namespace

Tokenizer::Tokenizer(const std::string& bpe_file) {
    std::ifstream in(bpe_file, std::ios::binary);
    if (!in) throw std::invalid_argument("Tokenizer file " + bpe_file + " does not exist");
    // bytes_to_unicode (tokenizer.cpp:22-53 / gen_tokenizer_file.py:5-24)
    {
        int extra = 0;
        for (int b = 0; b < 256; ++b) {
            const bool printable = (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174 && b <= 255);
            byte_symbol_[b].clear();
            append_utf8(byte_symbol_[b], printable ? (uint32_t)b : 256u + (uint32_t)extra++);
        }
    }
    // ids follow line order; a "first second" line is both a token (first+second) and a merge rule (tokenizer.cpp:239-251)
    std::string line;
    unsigned next = 0, rank = 0;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        const size_t sp = line.find(' ');
        if (next >= std::numeric_limits<token_type>::max() - 1) throw std::invalid_argument("tokenizer file has too many entries");
        if (sp == std::string::npos) {
            tokens_.emplace(line, (token_type)next++);
        } else {
            const std::string first = line.substr(0, sp), second = line.substr(sp + 1);
            tokens_.emplace(first + second, (token_type)next++);
            ranks_.emplace(first + '\x01' + second, rank++);
        }
    }
    start_ = (token_type)next++;
    end_ = (token_type)next++;
}

void Tokenizer::bpe(std::vector<token_type>& out, const std::vector<std::string>& symbols, unsigned max_len) const {
    if (out.size() >= max_len || symbols.empty()) return;
    std::vector<std::string> word = symbols;
    word.back() += "</w>";
    auto id_of = [&](const std::string& s) {
        auto it = tokens_.find(s);
        if (it == tokens_.end()) throw std::invalid_argument("symbol not in the tokenizer vocabulary: " + s);
        return it->second;
    };
    while (word.size() > 1) {
        unsigned best = std::numeric_limits<unsigned>::max();
        size_t best_i = 0;
        for (size_t i = 0; i + 1 < word.size(); ++i) {
            auto it = ranks_.find(word[i] + '\x01' + word[i + 1]);
            if (it != ranks_.end() && it->second < best) {
                best = it->second;
                best_i = i;
            }
        }
        if (best == std::numeric_limits<unsigned>::max()) break;
        const std::string first = word[best_i], second = word[best_i + 1];
        std::vector<std::string> merged;
        merged.reserve(word.size());
        for (size_t i = 0; i < word.size();) {
            if (i + 1 < word.size() && word[i] == first && word[i + 1] == second) {
                merged.push_back(first + second);
                i += 2;
            } else {
                merged.push_back(word[i]);
                i += 1;
            }
        }
        word.swap(merged);
    }
    for (const auto& w : word) {
        out.push_back(id_of(w));
        if (out.size() >= max_len) return;
    }
}

std::vector<Tokenizer::token_type> Tokenizer::tokenize(const std::string& text, unsigned context_len) const {
    if (context_len < 2) throw std::invalid_argument("context_len must be at least 2");
    std::vector<token_type> out;
    out.reserve(context_len);
    out.push_back(start_);

    // clean-up (tokenizer.cpp:55-108): trim, collapse whitespace runs to one space, lowercase
    std::vector<uint32_t> cps;
    {
        const std::vector<uint32_t> raw = decode_utf8(text);
        cps.reserve(raw.size());
        bool pending_space = false;
        for (uint32_t c : raw) {
            if (is_space(c)) {
                pending_space = !cps.empty();
            } else {
                if (pending_space) cps.push_back(' ');
                pending_space = false;
                cps.push_back(to_lower(c));
            }
        }
    }
    // 's|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+   (tokenizer.cpp:111, :113-222)
    const size_t n = cps.size();
    size_t i = 0;
    auto emit = [&](size_t b, size_t e) {
        std::string utf8;
        for (size_t k = b; k < e; ++k) append_utf8(utf8, cps[k]);
        std::vector<std::string> symbols;
        symbols.reserve(utf8.size());
        for (unsigned char byte : utf8) symbols.push_back(byte_symbol_[byte]);
        bpe(out, symbols, context_len - 1);
    };
    while (i < n && out.size() < context_len - 1) {
        const uint32_t c = cps[i];
        if (c == '\'' && i + 1 < n) {
            const uint32_t c1 = cps[i + 1];
            if (c1 == 's' || c1 == 't' || c1 == 'm' || c1 == 'd') { emit(i, i + 2); i += 2; continue; }
            if (i + 2 < n) {
                const uint32_t c2 = cps[i + 2];
                if ((c1 == 'r' && c2 == 'e') || (c1 == 'v' && c2 == 'e') || (c1 == 'l' && c2 == 'l')) { emit(i, i + 3); i += 3; continue; }
            }
        }
        if (is_digit(c)) { emit(i, i + 1); i += 1; continue; }
        if (is_letter(c)) {
            size_t j = i + 1;
            while (j < n && is_letter(cps[j])) ++j;
            emit(i, j); i = j; continue;
        }
        if (!is_space(c)) {
            size_t j = i + 1;
            while (j < n && !is_space(cps[j]) && !is_letter(cps[j]) && !is_digit(cps[j])) ++j;
            emit(i, j); i = j; continue;
        }
        ++i;
    }
    while (out.size() < context_len) out.push_back(end_);
    return out;
}

} // namespace sdod
