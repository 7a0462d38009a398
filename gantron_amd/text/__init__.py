"""text_to_sequence / sequence_to_text (reference text/__init__.py:15-74)."""
import re

from . import cleaners
from .symbols import symbols

_symbol_to_id = {s: i for i, s in enumerate(symbols)}
_id_to_symbol = dict(enumerate(symbols))
_curly_re = re.compile(r'(.*?)\{(.+?)\}(.*)')


def _keep(s):
    return s in _symbol_to_id and s != '_' and s != '~'


def _ids(syms):
    return [_symbol_to_id[s] for s in syms if _keep(s)]


def _clean(text, cleaner_names):
    for name in cleaner_names:
        fn = getattr(cleaners, name, None)
        if fn is None:
            raise Exception('Unknown cleaner: %s' % name)
        text = fn(text)
    return text


def text_to_sequence(text, cleaner_names):
    """Ids of the cleaned text; `{HH AW1 S}` spans are read as ARPAbet phones."""
    sequence = []
    while len(text):
        m = _curly_re.match(text)
        if not m:
            sequence += _ids(_clean(text, cleaner_names))
            break
        sequence += _ids(_clean(m.group(1), cleaner_names))
        sequence += _ids(['@' + s for s in m.group(2).split()])
        text = m.group(3)
    return sequence


def sequence_to_text(sequence):
    out = ''
    for i in sequence:
        s = _id_to_symbol.get(int(i))
        if s is None:
            continue
        out += '{%s}' % s[1:] if (len(s) > 1 and s[0] == '@') else s
    return out.replace('}{', ' ')
