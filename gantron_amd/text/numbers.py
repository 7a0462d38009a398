"""Number normalisation of the English cleaners (reference text/numbers.py:9-71).

The reference delegates spelling to `inflect` (not part of its tree, absent here): this module
restates the few behaviours it relies on -- cardinals without "and", ordinals, and the pairwise
reading of years ("1984" -> "nineteen eighty-four", zero read as "oh").  PARITY UNPINNED: no
reference fixture covers this boundary."""
import re

_ONES = ['zero', 'one', 'two', 'three', 'four', 'five', 'six', 'seven', 'eight', 'nine', 'ten', 'eleven', 'twelve',
         'thirteen', 'fourteen', 'fifteen', 'sixteen', 'seventeen', 'eighteen', 'nineteen']
_TENS = ['', '', 'twenty', 'thirty', 'forty', 'fifty', 'sixty', 'seventy', 'eighty', 'ninety']
_SCALES = [(10 ** 12, 'trillion'), (10 ** 9, 'billion'), (10 ** 6, 'million'), (1000, 'thousand')]
_ORD_IRREG = {'one': 'first', 'two': 'second', 'three': 'third', 'five': 'fifth', 'eight': 'eighth', 'nine': 'ninth',
              'twelve': 'twelfth'}


def _below_100(n, zero='zero'):
    if n < 20:
        return _ONES[n] if n else zero
    t, o = divmod(n, 10)
    return _TENS[t] + ('-' + _ONES[o] if o else '')


def _below_1000(n):
    h, r = divmod(n, 100)
    parts = []
    if h:
        parts.append(_ONES[h] + ' hundred')
    if r:
        parts.append(_below_100(r))
    return ' '.join(parts)


def cardinal(n):
    """number_to_words(n, andword='')"""
    if n == 0:
        return 'zero'
    parts = []
    for scale, name in _SCALES:
        q, n = divmod(n, scale)
        if q:
            parts.append(cardinal(q) + ' ' + name)
    if n:
        parts.append(_below_1000(n))
    return ', '.join(parts) if len(parts) > 1 else parts[0]


def pairs(n):
    """number_to_words(n, andword='', zero='oh', group=2) with ', ' removed: digits read two by two."""
    s = str(n)
    groups = [s[max(i - 2, 0):i] for i in range(len(s), 0, -2)][::-1]
    out = []
    for gdigits in groups:
        if len(gdigits) == 2 and gdigits[0] == '0':
            out.append('oh ' + _below_100(int(gdigits[1]), zero='oh'))
        else:
            out.append(_below_100(int(gdigits), zero='oh'))
    return ' '.join(out)


def ordinal(n):
    words = cardinal(n)
    head, sep, last = words.rpartition(' ')
    if '-' in last:
        h2, _, l2 = last.rpartition('-')
        head, sep, last = head + sep + h2, '-', l2
    if last in _ORD_IRREG:
        last = _ORD_IRREG[last]
    elif last.endswith('y'):
        last = last[:-1] + 'ieth'
    else:
        last += 'th'
    return head + sep + last


_comma_number_re = re.compile(r'([0-9][0-9\,]+[0-9])')
_decimal_number_re = re.compile(r'([0-9]+\.[0-9]+)')
_pounds_re = re.compile(r'£([0-9\,]*[0-9]+)')
_dollars_re = re.compile(r'\$([0-9\.\,]*[0-9]+)')
_ordinal_re = re.compile(r'[0-9]+(st|nd|rd|th)')
_number_re = re.compile(r'[0-9]+')


def _dollars(m):
    match = m.group(1)
    parts = match.split('.')
    if len(parts) > 2:
        return match + ' dollars'
    dollars = int(parts[0]) if parts[0] else 0
    cents = int(parts[1]) if len(parts) > 1 and parts[1] else 0
    d = '%s %s' % (dollars, 'dollar' if dollars == 1 else 'dollars')
    c = '%s %s' % (cents, 'cent' if cents == 1 else 'cents')
    if dollars and cents:
        return d + ', ' + c
    return d if dollars else (c if cents else 'zero dollars')


def _number(m):
    num = int(m.group(0))
    if 1000 < num < 3000:
        if num == 2000:
            return 'two thousand'
        if 2000 < num < 2010:
            return 'two thousand ' + cardinal(num % 100)
        if num % 100 == 0:
            return cardinal(num // 100) + ' hundred'
        return pairs(num)
    return cardinal(num)


def normalize_numbers(text):
    text = re.sub(_comma_number_re, lambda m: m.group(1).replace(',', ''), text)
    text = re.sub(_pounds_re, r'\1 pounds', text)
    text = re.sub(_dollars_re, _dollars, text)
    text = re.sub(_decimal_number_re, lambda m: m.group(1).replace('.', ' point '), text)
    text = re.sub(_ordinal_re, lambda m: ordinal(int(re.match(r'[0-9]+', m.group(0)).group(0))), text)
    text = re.sub(_number_re, _number, text)
    return text
