"""RWKV "world" tokenizer: greedy longest-match over a byte trie.

Host-side string code next to the hot path (the worker needs ids <-> text, chirrup/worker.py:19,
:221, :492).  Same behaviour as the reference's TRIE_TOKENIZER (Albatross/utils.py:104-159): vocab
file lines are ``<id> <python literal> <byte length>``; id 0 is "<|endoftext|>" and is never produced
by encode; encoding repeatedly takes the longest vocabulary entry that prefixes the remaining bytes.
"""
import ast
from typing import Dict, List


class TRIE_TOKENIZER:
    def __init__(self, file_name: str):
        self.idx2token: Dict[int, bytes] = {0: "<|endoftext|>".encode("utf-8")}
        with open(file_name, "r", encoding="utf-8") as f:
            for line in f:
                if not line.strip():
                    continue
                first, last = line.index(" "), line.rindex(" ")
                tok = ast.literal_eval(line[first:last].strip())
                tok = tok.encode("utf-8") if isinstance(tok, str) else tok
                if not isinstance(tok, bytes) or len(tok) != int(line[last:]):
                    raise ValueError(f"bad vocabulary line: {line!r}")
                self.idx2token[int(line[:first])] = tok
        self.token2idx = {t: i for i, t in self.idx2token.items() if i != 0}
        # trie as nested dicts keyed by byte value; key -1 holds the token id that ends at the node
        self._root: dict = {}
        for tok, i in self.token2idx.items():
            node = self._root
            for byte in tok:
                node = node.setdefault(byte, {})
            node[-1] = i

    def encodeBytes(self, src: bytes) -> List[int]:
        out: List[int] = []
        pos, n = 0, len(src)
        while pos < n:
            node, best_id, best_end, j = self._root, None, pos, pos
            while j < n:
                node = node.get(src[j])
                if node is None:
                    break
                j += 1
                if -1 in node:
                    best_id, best_end = node[-1], j
            if best_id is None:
                raise ValueError(f"byte {src[pos]!r} at offset {pos} is not covered by the vocabulary")
            out.append(best_id)
            pos = best_end
        return out

    def decodeBytes(self, tokens) -> bytes:
        return b"".join(self.idx2token[int(i)] for i in tokens)

    def encode(self, src: str) -> List[int]:
        return self.encodeBytes(src.encode("utf-8"))

    def decode(self, tokens, utf8_errors: str = "strict") -> str:
        return self.decodeBytes(tokens).decode("utf-8", errors=utf8_errors)
