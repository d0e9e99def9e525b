"""oracle/tokenizer_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

Pure-Python restatement of the reference's CLIP byte-level BPE tokenizer
(/root/reference/csrc/libsdod/src/tokenizer.cpp) and of its vocabulary file
format (/root/reference/gen_tokenizer_file.py:27-42).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Pinning: checked token-for-token against the reference's own tokenizer.cpp
compiled in place (oracle/_ref/libref.so) on a synthetic vocabulary, and against
tests/golden/tokenizer_synthetic.json generated from it (oracle/gen_golden.py).
The real CLIP vocabulary (bpe_simple_vocab_16e6.txt.gz) is absent offline, so
parity with real CLIP ids is "unpinned" until a vocab file is supplied.

Deliberate divergences from the reference (all documented in DESIGN.md):
  * Q3 (tokenizer.cpp:335-362): the reference's merge scan never merges `(a,b)`
    when the symbol before the pair equals `a` ([a,a,b]) and then loops forever.
    This oracle (and the product) implement the canonical CLIP merge scan.
  * Q4 (tokenizer.cpp:259-261): the reference needs the process locale
    en_US.utf8; without it any non-ASCII byte throws.  Here UTF-8 is decoded
    directly; letters / digits / lowercase follow Python's unicode tables.
  * whitespace: the reference collapses only iswblank (space, tab); canonical
    CLIP collapses all of \\s.  `canonical_ws=True` (default) follows CLIP,
    `canonical_ws=False` reproduces the reference for comparison runs.
"""
from __future__ import annotations


def bytes_to_unicode():
    """gen_tokenizer_file.py:5-24 / tokenizer.cpp:22-53 (bytes_translate)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(2 ** 8):
        if b not in bs:
            bs.append(b)
            cs.append(2 ** 8 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def write_ctokenizer(path, merges):
    """gen_tokenizer_file.py:33-42: 256 byte symbols, 256 '</w>' variants, then `first second` lines."""
    vocab = list(bytes_to_unicode().values())
    vocab = vocab + [v + "</w>" for v in vocab]
    with open(path, "wb") as f:
        for v in vocab:
            f.write((v + "\n").encode("utf-8"))
        for a, b in merges:
            f.write((a + " " + b + "\n").encode("utf-8"))


# A small hand-made merge table that exercises multi-level merges, '</w>' merges,
# apostrophe contractions, digits and punctuation runs.
SYNTHETIC_MERGES = [
    ("t", "h"), ("i", "n"), ("a", "n"), ("e", "r"), ("th", "e</w>"), ("o", "n"), ("r", "e"),
    ("h", "o"), ("r", "s"), ("ho", "rs"), ("hors", "e</w>"), ("a", "s"), ("t", "r"), ("as", "tr"),
    ("o", "f</w>"), ("p", "h"), ("o", "t"), ("ph", "ot"), ("o", "g"), ("phot", "og"), ("r", "a"),
    ("photog", "ra"), ("photogra", "ph</w>"), ("a", "u"), ("astr", "on"), ("astron", "au"),
    ("astronau", "t</w>"), ("r", "i"), ("d", "in"), ("ri", "din"), ("ridin", "g</w>"),
    ("a", "b"), ("ab", "c</w>"), ("'", "s</w>"), ("'", "t</w>"), ("!", "!</w>"), ("1", "2"),
    ("l", "l"), ("'", "ll</w>"), ("Ã", "©"), ("e", "e"), ("ee", "e</w>"),
]


class TokenizerOracle:
    def __init__(self, path, canonical_ws=True):
        # tokenizer.cpp:228-255: ids by line order; merged token id continues the count
        self.tokens = {}
        self.ranks = {}
        nxt = 0
        with open(path, "rb") as f:
            for raw in f.read().split(b"\n"):
                line = raw.decode("utf-8")
                if not line:
                    continue
                sp = line.find(" ")
                if sp < 0:
                    self.tokens.setdefault(line, nxt)
                    nxt += 1
                else:
                    first, second = line[:sp], line[sp + 1:]
                    self.tokens.setdefault(first + second, nxt)
                    nxt += 1
                    self.ranks.setdefault((first, second), len(self.ranks))
        self.start_token = nxt
        self.end_token = nxt + 1
        self.b2u = bytes_to_unicode()
        self.canonical_ws = canonical_ws

    # tokenizer.cpp:55-108
    def sanitize(self, s):
        is_blank = (lambda c: c.isspace()) if self.canonical_ws else (lambda c: c in " \t")
        out = []
        found = False
        last_blank = False
        for ch in s:
            b = is_blank(ch)
            if not b:
                found = True
                out.append(ch.lower() if len(ch.lower()) == 1 else ch)
            elif found and not last_blank:
                out.append(" ")
            last_blank = b
        if found and last_blank:
            out.pop()
        return "".join(out)

    # tokenizer.cpp:113-222: 's|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+
    def split(self, s):
        is_space = (lambda c: c.isspace()) if self.canonical_ws else (lambda c: c in " \t")
        toks = []
        i, n = 0, len(s)
        while i < n:
            if s[i] == "'" and i + 1 < n:
                if s[i + 1] in "stmd":
                    toks.append(s[i:i + 2]); i += 2; continue
                if i + 2 < n and s[i + 1:i + 3] in ("re", "ve", "ll"):
                    toks.append(s[i:i + 3]); i += 3; continue
            c = s[i]
            if c.isdigit():
                toks.append(c); i += 1; continue
            if c.isalpha():
                j = i + 1
                while j < n and s[j].isalpha():
                    j += 1
                toks.append(s[i:j]); i = j; continue
            if not is_space(c):
                j = i + 1
                while j < n and not (s[j].isdigit() or s[j].isalpha() or is_space(s[j])):
                    j += 1
                toks.append(s[i:j]); i = j; continue
            i += 1
        return toks

    # tokenizer.cpp:279-369 with the canonical merge scan (see module docstring, Q3)
    def bpe(self, out, token, max_len):
        if len(out) >= max_len:
            return
        word = list(token)
        word[-1] = word[-1] + "</w>"
        if len(word) == 1:
            out.append(self.tokens[word[0]])
            return
        while True:
            pairs = [(word[k], word[k + 1]) for k in range(len(word) - 1)]
            best = None
            for p in pairs:
                r = self.ranks.get(p)
                if r is not None and (best is None or r < best[0]):
                    best = (r, p)
            if best is None:
                break
            first, second = best[1]
            new_word = []
            k = 0
            while k < len(word):
                if word[k] == first and k + 1 < len(word) and word[k + 1] == second:
                    new_word.append(first + second); k += 2
                else:
                    new_word.append(word[k]); k += 1
            word = new_word
            if len(word) == 1:
                break
        for w in word:
            out.append(self.tokens[w])
            if len(out) >= max_len:
                return

    @staticmethod
    def _reference_scan(word, first, second):
        """The reference's merge scan, tokenizer.cpp:339-357, restated to detect Q3 divergence."""
        new_word = []
        prev_first = False
        for w in word:
            if prev_first:
                if w == second:
                    new_word.append(first + second)
                else:
                    new_word.append(first)
                    new_word.append(w)
                prev_first = False
            elif w == first:
                prev_first = True
            else:
                new_word.append(w)
        return new_word

    def diverges_from_reference(self, text):
        """True if the reference's merge scan (Q3) would differ from the canonical one on `text`
        (the reference then hangs or drops a symbol); such inputs are never sent to oracle/_ref."""
        for tok in self.split(self.sanitize(text)):
            translated = "".join(self.b2u[b] for b in tok.encode("utf-8"))
            word = list(translated)
            word[-1] += "</w>"
            while len(word) > 1:
                pairs = [(word[k], word[k + 1]) for k in range(len(word) - 1)]
                best = None
                for p in pairs:
                    r = self.ranks.get(p)
                    if r is not None and (best is None or r < best[0]):
                        best = (r, p)
                if best is None:
                    break
                first, second = best[1]
                canon = []
                k = 0
                while k < len(word):
                    if word[k] == first and k + 1 < len(word) and word[k + 1] == second:
                        canon.append(first + second); k += 2
                    else:
                        canon.append(word[k]); k += 1
                if self._reference_scan(word, first, second) != canon:
                    return True
                word = canon
        return False

    # tokenizer.cpp:258-276
    def tokenize(self, text, context_len=77):
        out = [self.start_token]
        for tok in self.split(self.sanitize(text)):
            translated = "".join(self.b2u[b] for b in tok.encode("utf-8"))
            self.bpe(out, translated, context_len - 1)
        while len(out) < context_len:
            out.append(self.end_token)
        return out
