// CLIP byte-level BPE tokenizer.  Same role, input file and output as the reference's libsdod::Tokenizer
// (csrc/libsdod/src/tokenizer.h:17-41, tokenizer.cpp:228-369; file produced by gen_tokenizer_file.py:27-42):
// SOT, BPE ids, truncation to context_len-1, EOT padding to context_len, uint16 ids.
// Differences, on purpose: the merge scan is the canonical CLIP one (the reference's
// hangs on [a,a,b] with merge (a,b)); UTF-8 is decoded here, no process locale is touched, so it is
// thread-safe and non-ASCII prompts work; all Unicode whitespace separates tokens (canonical CLIP \s).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace sdod {

class Tokenizer {
public:
    using token_type = uint16_t;
    explicit Tokenizer(const std::string& bpe_file);

    std::vector<token_type> tokenize(const std::string& text, unsigned context_len = 77) const;
    token_type start_token() const { return start_; }
    token_type end_token() const { return end_; }
    size_t vocab_size() const { return tokens_.size() + 2; }

private:
    std::unordered_map<std::string, token_type> tokens_;
    std::unordered_map<std::string, unsigned> ranks_; // key = first + '\x01' + second
    token_type start_ = 0, end_ = 0;
    std::string byte_symbol_[256];                    // bytes_to_unicode, UTF-8 encoded

    void bpe(std::vector<token_type>& out, const std::vector<std::string>& symbols, unsigned max_len) const;
};

} // namespace sdod
