"""Host glue between text-side phone symbols and phone ids.

`Megatts(..., symbol_table)` takes a k2-format symbol table (reference: utils/symbol_table.py, k2's SymbolTable,
Apache-2.0) and maps phone symbols to embedding rows the way `TokensCollector.phone2token` does
(modules/datamodule.py:30-35,65-69).  This module is an independent reader of that FILE FORMAT plus the two
behaviours a trained model depends on; nothing is imported from the reference tree (its `modules.datamodule`
drags in lhotse / lightning for one dictionary lookup).

The contract, as observed on the reference classes (tests/test_cpu_host.py checks it against the live ones):
  * one `<symbol> <integer id>` pair per line, blank-separated, empty lines ignored; a line with another field
    count, a repeated symbol or a repeated id is rejected;
  * id 0 is the null symbol: whatever the file calls it, `<eps>` if the file does not list id 0 (it is then added);
  * a phone's TOKEN id is the rank of its symbol among all symbols sorted as strings (null symbol included) -
    not the id written in the file.
G2P (text -> phone symbols: pypinyin + MFA dictionary, modules/tokenizer.py:41-98) stays outside (out of scope).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Tuple, Union


class SymbolTableError(ValueError, AssertionError):
    """A malformed symbol table.  Also an AssertionError: the reference signals these cases with `assert`, and a
    caller written against it catches that."""


def _parse_pairs(text: str) -> List[Tuple[str, int]]:
    rows = [(no, ln.split()) for no, ln in enumerate(text.splitlines(), 1)]
    bad = [(no, f) for no, f in rows if f and len(f) != 2]
    if bad:
        no, f = bad[0]
        raise SymbolTableError(f"symbol table line {no}: expected '<symbol> <id>', found {len(f)} fields")
    try:
        return [(f[0], int(f[1])) for _, f in rows if f]
    except ValueError as e:
        raise SymbolTableError(f"symbol table: id is not an integer ({e})") from None


def _first_repeat(items: Sequence) -> object:
    seen = set()
    for it in items:
        if it in seen:
            return it
        seen.add(it)
    return None


class SymbolTable:
    """Two-way map of a `.k2symbols` file.  `table[3]` -> symbol, `table["zh"]` -> id, `in` works for both."""

    DEFAULT_NULL = "<eps>"

    def __init__(self, pairs: Iterable[Tuple[str, int]], eps: str = DEFAULT_NULL):
        pairs = list(pairs)
        for what, col in (("symbol", [s for s, _ in pairs]), ("id", [i for _, i in pairs])):
            rep = _first_repeat(col)
            if rep is not None:
                raise SymbolTableError(f"symbol table: repeated {what} {rep!r}")
        if any(i < 0 for _, i in pairs):
            raise SymbolTableError("symbol table: negative id")
        self._by_sym: Dict[str, int] = {s: i for s, i in pairs}
        self._by_id: Dict[int, str] = {i: s for s, i in pairs}
        null_in_file = self._by_id.get(0)
        if null_in_file is None:                # the null symbol joins the table (and therefore the ranking)
            # a file that lists the null symbol with a NON-zero id and has no id 0 is accepted as the reference accepts it
            # (utils/symbol_table.py:66-68): the symbol is re-mapped to 0, its old id keeps pointing at it, and the
            # string-sorted ranking phone2token uses is unchanged
            self._by_sym[eps], self._by_id[0] = 0, eps
            null_in_file = eps
        self.eps = null_in_file

    @classmethod
    def from_str(cls, text: str) -> "SymbolTable":
        return cls(_parse_pairs(text))

    @classmethod
    def from_file(cls, filename: str) -> "SymbolTable":
        with open(filename, "r", encoding="utf-8") as f:
            return cls.from_str(f.read())

    @property
    def symbols(self) -> List[str]:
        """All symbols in STRING order (not id order) - the order phone2token numbers them in."""
        return sorted(self._by_sym)

    @property
    def ids(self) -> List[int]:
        return sorted(self._by_id)

    def __len__(self) -> int:
        return len(self._by_sym)

    def __getitem__(self, key: Union[int, str]):
        return self._by_id[key] if isinstance(key, int) else self._by_sym[key]

    def __contains__(self, key: Union[int, str]) -> bool:
        return key in (self._by_id if isinstance(key, int) else self._by_sym)


class TokensCollector:
    """Phone symbol -> token id = rank of the symbol in the string-sorted symbol list (modules/datamodule.py:30-35,65-69)."""

    def __init__(self, symbols_table: str) -> None:
        self.token2idx = {sym: rank for rank, sym in enumerate(SymbolTable.from_file(symbols_table).symbols)}

    def phone2token(self, phone: Iterable[str]):
        """list of phone symbols -> int64 tensor of token ids; an unknown symbol raises KeyError (as the reference)."""
        import torch
        return torch.tensor([self.token2idx[p] for p in phone], dtype=torch.int64)

    @property
    def vocab_size(self) -> int:
        return len(self.token2idx)
