"""Minimal FITS binary-table I/O for the mean-function file ("meanify" output).

The reference reads it with fitsio (``treegp/gp_interp.py:97-102``: one row, columns COORDS0
``(2*n)D`` with TDIM ``(2, n)`` and PARAMS0 ``nD``) and writes it in ``treegp/meanify.py:139-165``.
fitsio is not part of this image, and the wire format is small enough to handle directly:
2880-byte blocks, 80-character header cards, big-endian IEEE doubles.
"""
import numpy as np

_BLOCK = 2880


def _parse_header(raw, pos):
    cards = {}
    while True:
        blk = raw[pos:pos + _BLOCK]
        if len(blk) < _BLOCK:
            raise ValueError("truncated FITS header")
        pos += _BLOCK
        done = False
        for i in range(0, _BLOCK, 80):
            card = blk[i:i + 80].decode("ascii")
            key = card[:8].strip()
            if key == "END":
                done = True
                break
            if card[8:10] == "= ":
                val = card[10:]
                if val.lstrip().startswith("'"):
                    val = val.lstrip()[1:]
                    val = val[:val.index("'")].rstrip()
                else:
                    val = val.split("/")[0].strip()
                cards[key] = val
        if done:
            return cards, pos


def read_bintable_row(path, row=0):
    """Columns of one row of the first BINTABLE extension as float64 arrays, shaped by TDIMn
    (FITS lists the fastest axis first, so TDIM '(2,2500)' becomes shape (2500, 2))."""
    with open(path, "rb") as fh:
        raw = fh.read()
    cards, pos = _parse_header(raw, 0)
    naxis = int(cards.get("NAXIS", 0))
    nbytes = 0
    if naxis > 0:
        nbytes = abs(int(cards["BITPIX"])) // 8
        for a in range(1, naxis + 1):
            nbytes *= int(cards["NAXIS%d" % a])
    pos += (nbytes + _BLOCK - 1) // _BLOCK * _BLOCK
    cards, pos = _parse_header(raw, pos)
    if cards.get("XTENSION") != "BINTABLE":
        raise ValueError("first extension of %s is not a BINTABLE" % path)
    width = int(cards["NAXIS1"])
    off = pos + row * width
    out = {}
    for i in range(1, int(cards["TFIELDS"]) + 1):
        form = cards["TFORM%d" % i]
        if not form.endswith("D"):
            raise ValueError("only 'nD' (float64) columns are supported, got %r" % form)
        count = int(form[:-1] or 1)
        arr = np.frombuffer(raw, dtype=">f8", count=count, offset=off).astype(np.float64)
        off += 8 * count
        tdim = cards.get("TDIM%d" % i)
        if tdim:
            dims = tuple(int(t) for t in tdim.strip("()").split(","))
            arr = arr.reshape(dims[::-1])
        out[cards["TTYPE%d" % i]] = arr
    return out


def _card(key, value, comment=""):
    if isinstance(value, bool):
        v = "%20s" % ("T" if value else "F")
    elif isinstance(value, int):
        v = "%20d" % value
    else:
        v = "%-20s" % ("'%-8s'" % value)
    s = "%-8s= %s" % (key, v)
    if comment:
        s += " / " + comment
    return s[:80].ljust(80)


def write_bintable_row(path, columns, extname="average_solution"):
    """One-row BINTABLE with float64 columns; ``columns`` maps name -> ndarray."""
    prim = [_card("SIMPLE", True), _card("BITPIX", 16), _card("NAXIS", 0), _card("EXTEND", True), "END".ljust(80)]
    width = sum(8 * int(np.asarray(v).size) for v in columns.values())
    hdr = [_card("XTENSION", "BINTABLE"), _card("BITPIX", 8), _card("NAXIS", 2), _card("NAXIS1", width),
           _card("NAXIS2", 1), _card("PCOUNT", 0), _card("GCOUNT", 1), _card("TFIELDS", len(columns))]
    data = b""
    for i, (name, v) in enumerate(columns.items(), start=1):
        v = np.asarray(v, dtype=np.float64)
        hdr.append(_card("TTYPE%d" % i, name))
        hdr.append(_card("TFORM%d" % i, "%dD" % v.size))
        if v.ndim > 1:
            hdr.append(_card("TDIM%d" % i, "(" + ",".join(str(d) for d in v.shape[::-1]) + ")"))
        data += v.astype(">f8").tobytes()
    hdr.append(_card("EXTNAME", extname))
    hdr.append("END".ljust(80))

    def pad(b, fill):
        return b + fill * ((-len(b)) % _BLOCK)
    with open(path, "wb") as fh:
        fh.write(pad("".join(prim).encode("ascii"), b" "))
        fh.write(pad("".join(hdr).encode("ascii"), b" "))
        fh.write(pad(data, b"\0"))
