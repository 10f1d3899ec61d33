"""Host glue between text-side phone symbols and phone ids: the k2-format symbol table reader and
`TokensCollector.phone2token` of the reference (utils/symbol_table.py:77-125,280-287 and
modules/datamodule.py:30-35,65-69), so that `Megatts(..., symbol_table)` needs nothing from the reference
tree (its `modules.datamodule` drags in lhotse / lightning for this one dictionary lookup).

Reference behaviour kept exactly (it is part of a trained model's contract - the embedding row of a phone):
  * the table file has one `<symbol> <integer id>` pair per line, fields separated by blanks / tabs; empty lines are
    skipped, anything else than two fields, a duplicated symbol or a duplicated id is an error; id 0 is the null
    symbol (`<eps>` unless the file names another one) and is ADDED when the file does not list it;
  * `SymbolTable.symbols` is the list of symbols sorted as STRINGS - not by id - and `TokensCollector` numbers that
    sorted list: a phone's token id is its rank among the sorted symbols (`<eps>` included), whatever ids the file gives.
G2P (text -> phone symbols: pypinyin + MFA dictionary, modules/tokenizer.py:41-98) stays outside (SURVEY: out of scope).
"""
from __future__ import annotations

from typing import Dict, Iterable, List


class SymbolTable:
    """utils/symbol_table.py (k2's SymbolTable): id <-> symbol maps of a `.k2symbols` file."""

    def __init__(self, id2sym: Dict[int, str], sym2id: Dict[str, int], eps: str = "<eps>"):
        self._id2sym, self._sym2id, self.eps = dict(id2sym), dict(sym2id), eps
        for idx, sym in self._id2sym.items():                       # __post_init__, :57-73
            assert self._sym2id[sym] == idx and idx >= 0
        for sym, idx in self._sym2id.items():
            assert idx >= 0 and self._id2sym[idx] == sym
        if 0 not in self._id2sym:
            self._id2sym[0] = self.eps
            self._sym2id[self.eps] = 0
        else:
            assert self._id2sym[0] == self.eps and self._sym2id[self.eps] == 0

    @staticmethod
    def from_str(s: str) -> "SymbolTable":                          # :77-106
        id2sym: Dict[int, str] = {}
        sym2id: Dict[str, int] = {}
        for line in s.split("\n"):
            fields = line.split()
            if len(fields) == 0:
                continue
            assert len(fields) == 2, f"Expect a line with 2 fields. Given: {len(fields)}"
            sym, idx = fields[0], int(fields[1])
            assert sym not in sym2id, f"Duplicated symbol {sym}"
            assert idx not in id2sym, f"Duplicated id {idx}"
            id2sym[idx] = sym
            sym2id[sym] = idx
        return SymbolTable(id2sym, sym2id, id2sym.get(0, "<eps>"))

    @staticmethod
    def from_file(filename: str) -> "SymbolTable":                  # :108-125
        with open(filename, "r", encoding="utf-8") as f:
            return SymbolTable.from_str(f.read().strip())

    @property
    def symbols(self) -> List[str]:                                 # :280-287: sorted as strings
        return sorted(self._sym2id.keys())

    @property
    def ids(self) -> List[int]:
        return sorted(self._id2sym.keys())

    def __len__(self) -> int:
        return len(self._sym2id)

    def __getitem__(self, k):
        return self._id2sym[k] if isinstance(k, int) else self._sym2id[k]

    def __contains__(self, k) -> bool:
        return k in self._id2sym if isinstance(k, int) else k in self._sym2id


class TokensCollector:
    """modules/datamodule.py:30-35,65-69: phone symbol -> token id = rank of the symbol in the sorted symbol list."""

    def __init__(self, symbols_table: str) -> None:
        unique_tokens = SymbolTable.from_file(symbols_table).symbols
        self.token2idx = {token: idx for idx, token in enumerate(unique_tokens)}

    def phone2token(self, phone: Iterable[str]):
        """list of phone symbols -> int64 tensor of token ids; an unknown symbol raises KeyError (as the reference)."""
        import torch
        return torch.tensor([self.token2idx[token] for token in phone], dtype=torch.int64)

    @property
    def vocab_size(self) -> int:
        return len(self.token2idx)
