"""
Minimal FITS image reader / writer (numpy only) for `Observation(path)`,
`Observation.save_observation` and `save_mapped_observation`
(`planetmapper/observation.py:228-250, 1184-1474` use astropy.io.fits for this; astropy is
not a dependency of this package).

Scope: primary HDU + IMAGE extensions, BITPIX 8/16/32/64/-32/-64 with the unsigned-integer
BZERO convention, fixed- and free-format value cards, `HIERARCH` keywords (ESO convention, as
written by astropy for the `HIERARCH PLANMAP ...` metadata), COMMENT / HISTORY cards and
CONTINUE long strings. Tables and other extension types are skipped on read.
"""

from __future__ import annotations

import math
import os
from typing import Any, Iterable, Iterator

import numpy as np

BLOCK = 2880
CARD = 80
COMMENTARY = ('COMMENT', 'HISTORY', '')
_STRUCTURAL = ('SIMPLE', 'BITPIX', 'NAXIS', 'EXTEND', 'XTENSION', 'PCOUNT', 'GCOUNT', 'END')

_BITPIX_DTYPE = {8: '>u1', 16: '>i2', 32: '>i4', 64: '>i8', -32: '>f4', -64: '>f8'}
# numpy dtype -> (BITPIX, BZERO) ; unsigned types use the FITS offset convention
_DTYPE_BITPIX = {
    'u1': (8, None), 'i2': (16, None), 'i4': (32, None), 'i8': (64, None),
    'f4': (-32, None), 'f8': (-64, None),
    'u2': (16, 32768), 'u4': (32, 2147483648), 'i1': (8, -128),
}  # fmt: skip


def _norm(keyword: str) -> str:
    k = str(keyword).strip().upper()
    if k.startswith('HIERARCH '):
        k = k[9:].strip()
    return ' '.join(k.split())


class Header:
    """Ordered FITS header: a list of `[keyword, value, comment]` cards with dict-style access
    (first matching card), like the subset of `astropy.io.fits.Header` the reference uses."""

    def __init__(self, cards: Iterable | dict | None = None) -> None:
        self.cards: list[list] = []
        if isinstance(cards, dict):
            cards = list(cards.items())
        for c in cards or ():
            c = tuple(c)
            self.append(c[0], c[1] if len(c) > 1 else '', c[2] if len(c) > 2 else '')

    def copy(self) -> 'Header':
        h = Header()
        h.cards = [list(c) for c in self.cards]
        return h

    def __len__(self) -> int:
        return len(self.cards)

    def __iter__(self) -> Iterator[str]:
        return iter(c[0] for c in self.cards)

    def keys(self) -> list[str]:
        return [c[0] for c in self.cards]

    def items(self) -> list[tuple[str, Any]]:
        return [(c[0], c[1]) for c in self.cards]

    def _find(self, keyword: str) -> int:
        k = _norm(keyword)
        for i, c in enumerate(self.cards):
            if c[0] == k:
                return i
        return -1

    def __contains__(self, keyword: str) -> bool:
        return self._find(keyword) >= 0

    def __getitem__(self, keyword: str) -> Any:
        i = self._find(keyword)
        if i < 0:
            raise KeyError(f"Keyword {keyword!r} not found.")
        return self.cards[i][1]

    def get(self, keyword: str, default: Any = None) -> Any:
        i = self._find(keyword)
        return default if i < 0 else self.cards[i][1]

    def comment(self, keyword: str) -> str:
        return self.cards[self._find(keyword)][2]

    def __setitem__(self, keyword: str, value: Any) -> None:
        comment = None
        if isinstance(value, tuple):
            value, comment = value
        i = self._find(keyword)
        if i < 0 or _norm(keyword) in COMMENTARY:
            self.append(keyword, value, comment or '')
        else:
            self.cards[i][1] = value
            if comment is not None:
                self.cards[i][2] = comment

    def append(self, keyword: str, value: Any = '', comment: str = '') -> None:
        self.cards.append([_norm(keyword), value, '' if comment is None else str(comment)])

    def add_comment(self, text: str) -> None:
        self.append('COMMENT', str(text))

    def add_history(self, text: str) -> None:
        self.append('HISTORY', str(text))

    def remove(self, keyword: str, ignore_missing: bool = False, remove_all: bool = False) -> None:
        k = _norm(keyword)
        idx = [i for i, c in enumerate(self.cards) if c[0] == k]
        if not idx:
            if ignore_missing:
                return
            raise KeyError(f"Keyword '{keyword}' not found.")
        for i in reversed(idx if remove_all else idx[:1]):
            del self.cards[i]

    def update(self, other: 'Header | dict') -> None:
        for k, v in other.items() if not isinstance(other, Header) else other.items():
            if _norm(k) in COMMENTARY:
                self.append(k, v)
            else:
                self[k] = v
        if isinstance(other, Header):
            for k, _, c in other.cards:
                if c and k not in COMMENTARY:
                    self.cards[self._find(k)][2] = c

    def __repr__(self) -> str:
        return '\n'.join(s.rstrip() for c in self.cards for s in format_card(*c))


# ---------------------------------------------------------------------------- card <-> text
def _format_value(value: Any) -> tuple[str, bool]:
    """FITS text of a value and whether it is a string"""
    if isinstance(value, (bool, np.bool_)):
        return ('T' if value else 'F'), False
    if isinstance(value, (int, np.integer)):
        return str(int(value)), False
    if isinstance(value, (float, np.floating)):
        v = float(value)
        if not math.isfinite(v):
            raise ValueError(f'Floating point {v!r} values are not allowed in FITS headers.')
        s = repr(v).upper()  # shortest round-trip form, exponent marker 'E'
        if '.' not in s and 'E' not in s:
            s += '.0'
        return s, False
    if value is None:
        return '', False
    s = str(value).replace("'", "''")
    return f"'{s:8}'", True


def is_hierarch(keyword: str) -> bool:
    return len(keyword) > 8 or ' ' in keyword or not all(ch.isalnum() or ch in '-_' for ch in keyword)


def format_card(keyword: str, value: Any = '', comment: str = '') -> list[str]:
    """One logical card -> one or more 80-character card images."""
    keyword = _norm(keyword)
    if keyword in COMMENTARY:
        text = str(value)
        chunks = [text[i : i + 72] for i in range(0, len(text), 72)] or ['']
        return [f'{keyword:<8}{c}'.ljust(CARD) for c in chunks]
    val, is_str = _format_value(value)
    com = f' / {comment}' if comment else ''
    if is_hierarch(keyword):
        head = f'HIERARCH {keyword} = {val}'
        if len(head) > CARD:
            head = f'HIERARCH {keyword}= {val}'  # the space before '=' is optional
        if len(head) > CARD:
            raise ValueError(f'The header keyword {keyword!r} with its value is too long')
        return [(head + com)[:CARD].ljust(CARD)]
    if is_str and len(val) > CARD - 10:
        # long string: CONTINUE convention
        body = str(value).replace("'", "''")
        parts = [body[i : i + 67] for i in range(0, len(body), 67)]
        cards = []
        for i, part in enumerate(parts):
            amp = '&' if i < len(parts) - 1 else ''
            if i == 0:
                cards.append(f"{keyword:<8}= '{part}{amp}'".ljust(CARD))
            else:
                cards.append(f"CONTINUE  '{part}{amp}'".ljust(CARD))
        if comment:
            last = cards[-1].rstrip()
            if len(last) + len(com) <= CARD:
                cards[-1] = (last + com).ljust(CARD)
        return cards
    field = f'{val:<20}' if is_str else f'{val:>20}'
    return [(f'{keyword:<8}= {field}' + com)[:CARD].ljust(CARD)]


def _parse_value(text: str) -> tuple[Any, str]:
    """value field (after '=') -> (value, comment)"""
    t = text.strip()
    if t.startswith("'"):
        out = []
        i = 1
        while i < len(t):
            if t[i] == "'":
                if i + 1 < len(t) and t[i + 1] == "'":
                    out.append("'")
                    i += 2
                    continue
                break
            out.append(t[i])
            i += 1
        rest = t[i + 1 :]
        comment = rest.split('/', 1)[1].strip() if '/' in rest else ''
        return ''.join(out).rstrip(), comment
    val, _, comment = t.partition('/')
    val = val.strip()
    comment = comment.strip()
    if val == 'T':
        return True, comment
    if val == 'F':
        return False, comment
    if val == '':
        return None, comment
    try:
        return int(val), comment
    except ValueError:
        pass
    try:
        return float(val.upper().replace('D', 'E')), comment
    except ValueError:
        return val, comment


def parse_card(card: str) -> tuple[str, Any, str] | None:
    """80-character card image -> (keyword, value, comment); None for END / blank padding"""
    key = card[:8].strip().upper()
    if key == 'END' and card[8:].strip() == '':
        return None
    if key == 'HIERARCH' and '=' in card:
        left, _, right = card[8:].partition('=')
        value, comment = _parse_value(right)
        return _norm(left), value, comment
    if card[8:10] == '= ':
        value, comment = _parse_value(card[10:])
        return key, value, comment
    if key == 'CONTINUE':
        value, comment = _parse_value(card[8:])
        return 'CONTINUE', value, comment
    return key, card[8:].rstrip(), ''


# ---------------------------------------------------------------------------- HDUs
class HDU:
    """One header-data unit: `header` (user + structural cards), `data` (native byte order, or
    None) and `name` (EXTNAME, 'PRIMARY' for the first)."""

    def __init__(self, data: np.ndarray | None = None, header: Header | None = None, name: str | None = None) -> None:
        self.data = None if data is None else np.asarray(data)
        self.header = header.copy() if header is not None else Header()
        self.name = name if name is not None else self.header.get('EXTNAME', '')


def _read_header(buf: bytes, pos: int) -> tuple[Header, int]:
    hdr = Header()
    while True:
        block = buf[pos : pos + BLOCK]
        if len(block) < BLOCK:
            raise ValueError('truncated FITS header')
        pos += BLOCK
        for i in range(BLOCK // CARD):
            card = block[i * CARD : (i + 1) * CARD].decode('ascii', errors='replace')
            parsed = parse_card(card)
            if parsed is None:
                return hdr, pos
            key, value, comment = parsed
            if key == 'CONTINUE' and hdr.cards and isinstance(hdr.cards[-1][1], str):
                prev = hdr.cards[-1]
                prev[1] = (prev[1][:-1] if prev[1].endswith('&') else prev[1]) + str(value)
                if comment:
                    prev[2] = (prev[2] + ' ' + comment).strip()
                continue
            if key == '' and value == '':
                continue
            hdr.append(key, value, comment)


def read(path: str | os.PathLike) -> list[HDU]:
    """All HDUs of a FITS file (data of non-image extensions is None)."""
    path = os.fspath(path)
    with open(path, 'rb') as f:
        buf = f.read()
    if buf[:2] == b'\x1f\x8b':  # gzip magic, whatever the suffix says
        import gzip

        buf = gzip.decompress(buf)
    pos = 0
    hdus: list[HDU] = []
    while pos < len(buf):
        if buf[pos : pos + BLOCK].strip(b' \0') == b'':
            pos += BLOCK
            continue
        hdr, pos = _read_header(buf, pos)
        if not hdus and hdr.get('SIMPLE') is not True:
            raise ValueError('not a FITS file (SIMPLE = T missing)')
        naxis = int(hdr.get('NAXIS', 0))
        shape = [int(hdr[f'NAXIS{i}']) for i in range(naxis, 0, -1)]
        bitpix = int(hdr.get('BITPIX', 8))
        count = int(np.prod(shape)) if shape else 0
        nbytes = abs(bitpix) // 8 * int(hdr.get('GCOUNT', 1)) * (int(hdr.get('PCOUNT', 0)) + count)
        data = None
        if count and hdr.get('XTENSION', 'IMAGE').strip() == 'IMAGE':
            raw = np.frombuffer(buf, dtype=_BITPIX_DTYPE[bitpix], count=count, offset=pos).reshape(shape)
            bzero, bscale = hdr.get('BZERO', 0), hdr.get('BSCALE', 1)
            if bscale == 1 and bzero == 0:
                data = raw.astype(raw.dtype.newbyteorder('='))
            elif bscale == 1 and (bitpix, bzero) in ((16, 32768), (32, 2147483648), (8, -128)):
                target = {16: np.uint16, 32: np.uint32, 8: np.int8}[bitpix]
                data = (raw.astype(np.int64) + int(bzero)).astype(target)
            else:
                data = raw.astype(np.float64) * bscale + bzero
        pos += (nbytes + BLOCK - 1) // BLOCK * BLOCK
        name = 'PRIMARY' if not hdus else str(hdr.get('EXTNAME', '')).strip()
        hdus.append(HDU(data, hdr, name))
    if not hdus:
        raise ValueError('empty FITS file')
    return hdus


def _structural_cards(data: np.ndarray | None, primary: bool, name: str) -> tuple[list[tuple], np.ndarray | None]:
    cards: list[tuple] = []
    payload = None
    bzero = None
    if data is None:
        bitpix, shape = 8, ()
    else:
        if data.dtype == np.bool_:
            data = data.astype(np.uint8)
        code = data.dtype.kind + str(data.dtype.itemsize)
        if code not in _DTYPE_BITPIX:
            raise TypeError(f'unsupported FITS data type {data.dtype}')
        bitpix, bzero = _DTYPE_BITPIX[code]
        shape = data.shape
        if bzero is None:
            payload = np.ascontiguousarray(data, dtype=_BITPIX_DTYPE[bitpix])
        else:
            payload = np.ascontiguousarray(data.astype(np.int64) - bzero, dtype=_BITPIX_DTYPE[bitpix])
    if primary:
        cards.append(('SIMPLE', True, 'conforms to FITS standard'))
    else:
        cards.append(('XTENSION', 'IMAGE', 'Image extension'))
    cards.append(('BITPIX', bitpix, 'array data type'))
    cards.append(('NAXIS', len(shape), 'number of array dimensions'))
    for i, n in enumerate(reversed(shape)):
        cards.append((f'NAXIS{i + 1}', int(n), ''))
    if primary:
        cards.append(('EXTEND', True, ''))
    else:
        cards.append(('PCOUNT', 0, 'number of parameters'))
        cards.append(('GCOUNT', 1, 'number of groups'))
    if bzero is not None:
        cards.append(('BSCALE', 1, ''))
        cards.append(('BZERO', bzero, ''))
    if not primary and name:
        cards.append(('EXTNAME', name, 'extension name'))
    return cards, payload


def write(path: str | os.PathLike, hdus: list[HDU], overwrite: bool = True) -> None:
    """Write HDUs (the first becomes the primary HDU, the others IMAGE extensions)."""
    path = os.fspath(path)
    if not overwrite and os.path.exists(path):
        raise OSError(f'File {path!r} already exists.')
    out = bytearray()
    for i, hdu in enumerate(hdus):
        cards, payload = _structural_cards(hdu.data, i == 0, hdu.name)
        skip = set(_STRUCTURAL) | {'BSCALE', 'BZERO', 'EXTNAME'} | {f'NAXIS{k}' for k in range(1, 10)}
        if i == 0:
            skip.discard('EXTNAME')
        text = []
        for c in cards:
            text += format_card(*c)
        for k, v, c in hdu.header.cards:
            if k in skip or k == 'CONTINUE':
                continue
            text += format_card(k, v, c)
        text.append('END'.ljust(CARD))
        blob = ''.join(text).encode('ascii', errors='replace')
        out += blob + b' ' * (-len(blob) % BLOCK)
        if payload is not None:
            raw = payload.tobytes()
            out += raw + b'\0' * (-len(raw) % BLOCK)
    tmp = path + '.tmp~'
    if path.lower().endswith('.gz'):
        # astropy (the reference's writer) compresses by suffix; read() decompresses by suffix
        import gzip

        with gzip.open(tmp, 'wb', compresslevel=6) as f:
            f.write(out)
    else:
        with open(tmp, 'wb') as f:
            f.write(out)
    os.replace(tmp, path)


__all__ = ['Header', 'HDU', 'read', 'write', 'format_card', 'parse_card']
