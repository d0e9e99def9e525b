"""Producer of `ctokenizer.txt`, the on-disk vocabulary libsdod's tokenizer reads.

Format (what tokenizer.cpp:228-255 parses and gen_tokenizer_file.py:33-42 writes): UTF-8, '\\n'-separated;
512 symbol lines -- the 256 byte symbols of the GPT-2/CLIP byte<->unicode table, then the same 256 with '</w>' appended --
followed by one `first second` line per BPE merge.  Token id = line index; a merge line's token is first+second; the
tokenizer appends <|startoftext|> and <|endoftext|> after the last line (49406 / 49407 with the full CLIP table).

Sources accepted (none ships with the reference or this repository; there is no network here):
  * OpenAI CLIP's `bpe_simple_vocab_16e6.txt.gz` (line 0 is a header; CLIP uses merges 1 .. 49152-256-2), or
  * a directory holding HuggingFace `merges.txt` (+ optional `vocab.json`, then used to cross-check every id).
"""
import gzip
import json
import os

CLIP_MERGES = 49152 - 256 - 2


def byte_symbols():
    """the 256 printable stand-ins for byte values, in vocabulary order: bytes that are printable latin-1 characters
    stand for themselves ('!'..'~', U+00A1..U+00AC, U+00AE..U+00FF, in that order); the 68 others take U+0100 upward in
    ascending byte order"""
    printable = list(range(0x21, 0x7F)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    taken = set(printable)
    rest = [b for b in range(256) if b not in taken]
    return [chr(b) for b in printable] + [chr(0x100 + i) for i in range(len(rest))]


def read_merges(src, limit=CLIP_MERGES):
    """-> [(first, second), ...] from a CLIP .txt.gz / .txt merge list or a HF directory (merges.txt)"""
    if os.path.isdir(src):
        src = os.path.join(src, 'merges.txt')
    opener = gzip.open if src.endswith('.gz') else open
    with opener(src, 'rb') as f:
        lines = f.read().decode('utf-8').split('\n')
    body = lines[1:]                                   # line 0: '"bpe_simple_vocab_16e6.txt#version: 0.2' / '#version: 0.2'
    merges = []
    for ln in body:
        if len(merges) == limit:
            break
        parts = ln.split()
        if not parts:
            continue
        if len(parts) != 2:
            raise ValueError(f'malformed merge line {len(merges) + 1}: {ln!r}')
        merges.append((parts[0], parts[1]))
    return merges


def vocabulary(merges):
    """token strings in id order, including the two specials the tokenizer appends"""
    sym = byte_symbols()
    return sym + [s + '</w>' for s in sym] + [a + b for a, b in merges] + ['<|startoftext|>', '<|endoftext|>']


def write(path, merges):
    sym = byte_symbols()
    with open(path, 'wb') as f:
        for s in sym:
            f.write((s + '\n').encode('utf-8'))
        for s in sym:
            f.write((s + '</w>\n').encode('utf-8'))
        for a, b in merges:
            if any(c in x for x in (a, b) for c in ' \n'):
                raise ValueError(f'merge symbols must not contain blanks or newlines: {(a, b)!r}')
            f.write((a + ' ' + b + '\n').encode('utf-8'))
    return path


def generate(src, out_path, limit=CLIP_MERGES):
    """src -> out_path; with a HF directory that also has vocab.json, every token id is checked against it"""
    merges = read_merges(src, limit)
    vj = os.path.join(src, 'vocab.json') if os.path.isdir(src) else None
    if vj and os.path.exists(vj):
        with open(vj, 'r', encoding='utf-8') as f:
            want = json.load(f)
        for i, tok in enumerate(vocabulary(merges)):
            if want.get(tok) != i:
                raise ValueError(f'vocab.json disagrees at id {i}: {tok!r} -> {want.get(tok)}')
    return write(out_path, merges)
