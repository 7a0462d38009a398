"""Text cleaners (reference text/cleaners.py:16-90).  `convert_to_ascii` restates what the
reference obtains from `unidecode` (absent here) for the Latin range: NFKD decomposition with
combining marks dropped plus a few typographic replacements.  PARITY UNPINNED beyond ASCII."""
import re
import unicodedata

from .numbers import normalize_numbers

_whitespace_re = re.compile(r'\s+')
_ABBREVIATIONS = [(re.compile('\\b%s\\.' % a, re.IGNORECASE), b) for a, b in (
    ('mrs', 'misess'), ('mr', 'mister'), ('dr', 'doctor'), ('st', 'saint'), ('co', 'company'), ('jr', 'junior'),
    ('maj', 'major'), ('gen', 'general'), ('drs', 'doctors'), ('rev', 'reverend'), ('lt', 'lieutenant'),
    ('hon', 'honorable'), ('sgt', 'sergeant'), ('capt', 'captain'), ('esq', 'esquire'), ('ltd', 'limited'),
    ('col', 'colonel'), ('ft', 'fort'))]
_TYPOGRAPHIC = {'‘': "'", '’': "'", '“': '"', '”': '"', '–': '-', '—': '--',
                '…': '...', 'ß': 'ss', 'æ': 'ae', 'ø': 'o', 'œ': 'oe'}


def convert_to_ascii(text):
    text = ''.join(_TYPOGRAPHIC.get(ch, ch) for ch in text)
    text = unicodedata.normalize('NFKD', text)
    return ''.join(ch for ch in text if ord(ch) < 128)


def expand_abbreviations(text):
    for regex, replacement in _ABBREVIATIONS:
        text = re.sub(regex, replacement, text)
    return text


def collapse_whitespace(text):
    return re.sub(_whitespace_re, ' ', text)


def basic_cleaners(text):
    return collapse_whitespace(text.lower())


def transliteration_cleaners(text):
    return collapse_whitespace(convert_to_ascii(text).lower())


def english_cleaners(text):
    text = convert_to_ascii(text).lower()
    text = normalize_numbers(text)
    text = expand_abbreviations(text)
    return collapse_whitespace(text)
