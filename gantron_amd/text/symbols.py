"""Symbol inventory of the text front-end (reference: text/symbols.py:9-18, text/cmudict.py:6-14).

148 entries: pad, '-', punctuation, ASCII letters and '@'-prefixed ARPAbet phones.  The
embedding table size (hparams.n_symbols) is the length of this list.
"""

PAD = '_'
SPECIAL = '-'
PUNCTUATION = "!'(),.:;? "
LETTERS = ''.join(chr(c) for c in range(ord('A'), ord('Z') + 1)) + \
          ''.join(chr(c) for c in range(ord('a'), ord('z') + 1))

# ARPAbet: vowels carry an optional stress digit 0/1/2, consonants do not.
_VOWELS = ('AA', 'AE', 'AH', 'AO', 'AW', 'AY', 'EH', 'ER', 'EY', 'IH', 'IY', 'OW', 'OY', 'UH', 'UW')
_CONSONANTS = ('B', 'CH', 'D', 'DH', 'F', 'G', 'HH', 'JH', 'K', 'L', 'M', 'N', 'NG', 'P', 'R', 'S',
               'SH', 'T', 'TH', 'V', 'W', 'Y', 'Z', 'ZH')


def _arpabet():
    phones = list(_CONSONANTS)
    for v in _VOWELS:
        phones.extend([v, v + '0', v + '1', v + '2'])
    return sorted(phones)   # the reference list is in plain lexicographic order


ARPABET = _arpabet()
symbols = [PAD] + list(SPECIAL) + list(PUNCTUATION) + list(LETTERS) + ['@' + p for p in ARPABET]
