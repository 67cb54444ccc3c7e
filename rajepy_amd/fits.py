"""Minimal FITS primary-HDU writer/reader for the map products.

The reference writes its EM / Tau / Flux products with astropy.io.fits
(classes.py:1588-1650); astropy is not a dependency here.  This writer emits the same bytes
astropy 4.3.1 does for the same header assignments: 80-column cards, fixed-format values,
astropy's float formatting ('%.16G', exponent padded to two digits, clipped to 20 columns),
long HISTORY text folded at 72 columns, big-endian float64 data, 2880-byte blocks.
tests/test_host_logic.py::test_fits_products_byte_identical_to_reference compares whole files
with the reference's output (SHA-256 of the files astropy wrote for the reference run,
tests/golden/pipeline_cfg1.json).
"""
import numpy as np

BLOCK = 2880
CARD = 80


def format_float(value):
    s = '{:.16G}'.format(float(value))
    if '.' not in s and 'E' not in s:
        s += '.0'
    elif 'E' in s:
        mant, exp = s.split('E')
        sign = ''
        if exp[0] in '+-':
            sign, exp = exp[0], exp[1:]
        s = '{}E{}{:02d}'.format(mant, sign, int(exp))
    if len(s) > 20:
        i = s.find('E')
        s = s[:20] if i < 0 else s[:20 - (len(s) - i)] + s[i:]
    return s


def _format_value(value):
    if isinstance(value, (bool, np.bool_)):
        return '{:>20}'.format('T' if value else 'F')
    if isinstance(value, (int, np.integer)):
        return '{:>20d}'.format(int(value))
    if isinstance(value, (float, np.floating)):
        return '{:>20}'.format(format_float(value))
    if isinstance(value, str):
        return "'{:8}'".format(value.replace("'", "''"))
    raise TypeError("unsupported FITS value type {}".format(type(value)))


def card(key, value, comment=None):
    body = '{:8}= {}'.format(key, _format_value(value))
    if isinstance(value, str):
        body = '{:30}'.format(body)
    if comment:
        body += ' / ' + comment
    if len(body) > CARD:
        body = body[:CARD]
    return '{:80}'.format(body)


def history_cards(text):
    """A long HISTORY string is folded into 72-character commentary cards; trailing blanks of
    the whole text are dropped (astropy Header._add_commentary / Card commentary folding)."""
    text = text.rstrip()
    return ['{:80}'.format('HISTORY ' + text[i:i + 72].rstrip()
                           if text[i:i + 72].strip() else 'HISTORY')
            for i in range(0, max(len(text), 1), 72)]


class Header:
    """Ordered list of (key, value, comment) plus HISTORY lines, primary-HDU keywords first."""

    def __init__(self):
        self.cards = []
        self.history = []

    def set(self, key, value, comment=None):
        for i, (k, _, c) in enumerate(self.cards):
            if k == key:
                self.cards[i] = (key, value, comment if comment is not None else c)
                return
        self.cards.append((key, value, comment))

    def add_history(self, text):
        self.history.append(text)

    def render(self, data):
        shape = data if isinstance(data, tuple) else data.shape
        out = [card('SIMPLE', True, 'conforms to FITS standard'),
               card('BITPIX', -64, 'array data type'),
               card('NAXIS', len(shape), 'number of array dimensions')]
        for i, n in enumerate(reversed(shape)):
            out.append(card('NAXIS%d' % (i + 1), int(n)))
        out.append(card('EXTEND', True))
        out += [card(k, v, c) for k, v, c in self.cards]
        for h in self.history:
            out += history_cards(h)
        out.append('{:80}'.format('END'))
        txt = ''.join(out)
        txt += ' ' * (-len(txt) % BLOCK)
        return txt.encode('ascii')


class BigEndian:
    """Image data already in FITS order and byte order: `raw` = the big-endian float64 bytes
    (any C-contiguous bytes-like object), `shape` = the array shape they stand for.  JetModel
    builds large products this way on the GPU (transpose + byte swap at HBM speed)."""

    def __init__(self, shape, raw):
        self.shape = tuple(int(n) for n in shape)
        self.raw = raw
        if len(memoryview(raw).cast('B')) != 8 * int(np.prod(self.shape)):
            raise ValueError("payload size does not match its shape")


def writeto(filename, data, header):
    if isinstance(data, BigEndian):
        shape, raw = data.shape, memoryview(data.raw).cast('B')
    else:
        data = np.asarray(data, dtype=np.float64)
        shape = data.shape
        # one pass: layout (a transposed view is fine) and byte order together; written
        # straight from the array's buffer
        raw = memoryview(np.ascontiguousarray(data, dtype='>f8')).cast('B')
    with open(filename, 'wb') as f:
        f.write(header.render(shape))
        f.write(raw)
        f.write(b'\0' * (-len(raw) % BLOCK))


def read(filename):
    """-> (data float64, list of 80-char header cards).  Primary HDU, BITPIX=-64 only."""
    with open(filename, 'rb') as f:
        raw = f.read()
    cards, pos, done = [], 0, False
    while not done:
        block = raw[pos:pos + BLOCK].decode('ascii')
        pos += BLOCK
        for i in range(0, BLOCK, CARD):
            c = block[i:i + CARD]
            if c.startswith('END') and c[3:].strip() == '':
                done = True
                break
            cards.append(c)
    kv = {c[:8].strip(): c[10:30].strip() for c in cards if c[8:10] == '= '}
    if int(kv['BITPIX']) != -64:
        raise ValueError("only BITPIX = -64 products are supported")
    shape = tuple(int(kv['NAXIS%d' % i]) for i in range(int(kv['NAXIS']), 0, -1))
    n = int(np.prod(shape))
    data = np.frombuffer(raw, dtype='>f8', count=n, offset=pos).reshape(shape)
    return data.astype(np.float64), cards
