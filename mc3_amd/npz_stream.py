"""A ``.npz`` checkpoint that grows in place.

The reference's hub rewrites the whole output dictionary with ``np.savez`` at every progress
report (mc3/mcmc_driver.py:321-324): at 1e7 samples that is ~0.5 GB per report, ten times.
Here the arrays that grow with the run -- the rows of Z and what is kept per row -- are
*appended*: each lives in a region of the file sized for the whole run, a report writes only
the rows that are new, then the small arrays and a fresh directory behind the regions.  The
file is an ordinary (stored, ZIP64) zip of ``.npy`` members at every report: ``np.load``
reads it, and so does the reference's ``resume``.

Layout::

    [local header | .npy header (fixed size) | rows ... room for max_rows]    per grown array
    [small members + directory, report 1][... report 2] ...                 appended per report

A report never touches bytes an earlier directory describes, except the few hundred bytes of
the grown members' own headers (shape, CRC) at its very end: until that moment the file on
disk is the previous report's, consistent; a crash inside that last step can leave grown
members of different lengths, which ``np.load`` shows as such.  CRCs are kept current without
re-reading the file: the CRC of a member is combined from the CRC of its header and the
running CRC of its rows (``crc32_combine``, the GF(2) construction of zlib).
"""
import io
import os
import struct
import time
import zlib

import numpy as np

_NPY_HEADER = 256          # bytes of every grown member's .npy header (magic + dict, padded)
_ZIP64_EXTRA_LOCAL = 20    # id, size, usize, csize
_VERSION = 45


def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def _gf2_square(mat):
    return [_gf2_times(mat, mat[n]) for n in range(32)]


def crc32_combine(crc1, crc2, len2):
    """CRC-32 of A + B from crc32(A), crc32(B) and len(B) (zlib's crc32_combine)."""
    if len2 <= 0:
        return crc1
    odd = [0xedb88320] + [1 << n for n in range(31)]      # the operator for one zero bit
    even = _gf2_square(odd)                              # two
    odd = _gf2_square(even)                              # four
    while True:
        even = _gf2_square(odd)                          # (first pass: one zero byte)
        if len2 & 1:
            crc1 = _gf2_times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = _gf2_square(even)
        if len2 & 1:
            crc1 = _gf2_times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def _npy_header(dtype, shape):
    """A version-1.0 .npy header of exactly _NPY_HEADER bytes (the format pads with spaces)."""
    d = "{'descr': %r, 'fortran_order': False, 'shape': %r, }" % (
        np.lib.format.dtype_to_descr(np.dtype(dtype)), tuple(int(s) for s in shape))
    body = d.encode('latin1')
    pad = _NPY_HEADER - 10 - len(body) - 1
    if pad < 0:
        raise ValueError('array description does not fit the fixed .npy header')
    body += b' '*pad + b'\n'
    return b'\x93NUMPY\x01\x00' + struct.pack('<H', len(body)) + body


def _dos_time():
    t = time.localtime()
    return ((t.tm_year - 1980) << 9 | t.tm_mon << 5 | t.tm_mday,
            t.tm_hour << 11 | t.tm_min << 5 | t.tm_sec//2)


def _local_header(name, crc, size):
    date, tm = _dos_time()
    extra = struct.pack('<HHQQ', 1, 16, size, size)
    return struct.pack('<IHHHHHIIIHH', 0x04034b50, _VERSION, 0, 0, tm, date, crc,
                       0xffffffff, 0xffffffff, len(name), len(extra)) + name + extra


def _central_header(name, crc, size, offset):
    date, tm = _dos_time()
    extra = struct.pack('<HHQQQ', 1, 24, size, size, offset)
    return struct.pack('<IHHHHHHIIIHHHHHII', 0x02014b50, _VERSION, _VERSION, 0, 0, tm, date,
                       crc, 0xffffffff, 0xffffffff, len(name), len(extra), 0, 0, 0,
                       0o600 << 16, 0xffffffff) + name + extra


def _end_records(ncd, cd_size, cd_offset):
    z64 = struct.pack('<IQHHIIQQQQ', 0x06064b50, 44, _VERSION, _VERSION, 0, 0, ncd, ncd,
                      cd_size, cd_offset)
    loc = struct.pack('<IIQI', 0x07064b50, 0, cd_offset + cd_size, 1)
    end = struct.pack('<IHHHHIIH', 0x06054b50, 0xffff, 0xffff, 0xffff, 0xffff, 0xffffffff,
                      0xffffffff, 0)
    return z64 + loc + end


class _Grown:
    """One member that grows by rows."""

    def __init__(self, name, dtype, row_shape, max_rows, offset):
        self.name = (name + '.npy').encode()
        self.dtype = np.dtype(dtype)
        self.row_shape = tuple(int(s) for s in row_shape)
        self.row_bytes = self.dtype.itemsize*int(np.prod(self.row_shape, dtype=np.int64))
        self.max_rows = int(max_rows)
        self.offset = offset                               # of the local header
        self.header_bytes = 30 + len(self.name) + _ZIP64_EXTRA_LOCAL
        self.data_offset = offset + self.header_bytes + _NPY_HEADER
        self.end = self.data_offset + self.max_rows*self.row_bytes
        self.rows = 0
        self.data_crc = 0
        self.last = None                                   # the last row written

    def size(self):
        return _NPY_HEADER + self.rows*self.row_bytes

    def headers(self):
        npy = _npy_header(self.dtype, (self.rows,) + self.row_shape)
        crc = crc32_combine(zlib.crc32(npy), self.data_crc, self.rows*self.row_bytes)
        return crc, _local_header(self.name, crc, self.size()) + npy


class StreamingNpz:
    """``grown``: {key: (dtype, shape of one row, most rows it will ever have)}.  Every
    ``write(output)`` must pass, for each grown key, an array whose leading rows are the
    rows passed before (it is a run's history: rows are only ever appended)."""

    def __init__(self, path, grown):
        self.path = path if str(path).endswith('.npz') else str(path) + '.npz'
        self.grown = {}
        at = 0
        for key, (dtype, row_shape, max_rows) in grown.items():
            m = _Grown(key, dtype, row_shape, max_rows, at)
            self.grown[key] = m
            at = m.end
        self.tail0 = at                   # where the first generation of small members starts
        self.tail = at
        self.fd = os.open(self.path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
        self.bytes_written = 0            # (what a report cost: for the tests and the docs)

    def _pwrite(self, data, offset):
        view = memoryview(data).cast('B')
        done = 0
        while done < len(view):
            done += os.pwrite(self.fd, view[done:done + (1 << 30)], offset + done)
        self.bytes_written += len(view)

    def write(self, output):
        """The file becomes a snapshot of ``output``."""
        staged = []
        for key, m in self.grown.items():
            if key not in output:             # (not there yet: e.g. zmask before the burn-in)
                if m.rows:
                    raise ValueError(f"'{key}' was written before and is missing now")
                continue
            arr = np.asarray(output[key])
            if arr.dtype != m.dtype or arr.shape[1:] != m.row_shape or len(arr) > m.max_rows \
                    or len(arr) < m.rows:
                raise ValueError(f"'{key}': {arr.dtype}{arr.shape} does not continue the "
                                 f"{m.dtype}{(m.rows,) + m.row_shape} already written "
                                 f"(room for {m.max_rows} rows)")
            if m.rows and not np.array_equal(arr[m.rows - 1], m.last, equal_nan=True):
                m.rows, m.data_crc = 0, 0     # not a continuation after all: from the start
            new = np.ascontiguousarray(arr[m.rows:])
            if len(arr):
                m.last = np.array(arr[-1])
            if len(new):
                self._pwrite(new, m.data_offset + m.rows*m.row_bytes)
                m.data_crc = zlib.crc32(memoryview(new).cast('B'), m.data_crc)
            staged.append((m, len(arr)))
        # the small members of this generation, behind everything an earlier directory names
        start = self.tail
        blob = io.BytesIO()
        entries = []
        for key, val in output.items():
            if key in self.grown:
                continue
            buf = io.BytesIO()
            np.lib.format.write_array(buf, np.asanyarray(val), allow_pickle=True)
            data = buf.getvalue()
            name = (key + '.npy').encode()
            crc = zlib.crc32(data)
            entries.append((name, crc, len(data), start + blob.tell()))
            blob.write(_local_header(name, crc, len(data)) + data)
        headers = []
        for m, nrows in staged:
            m.rows = nrows
            crc, head = m.headers()
            headers.append((m, head))
            entries.append((m.name, crc, m.size(), m.offset))
        cd_offset = start + blob.tell()
        cd = b''.join(_central_header(*e) for e in entries)
        blob.write(cd + _end_records(len(entries), len(cd), cd_offset))
        # behind the previous generation (whose directory stays valid until the end of the file
        # moves), then -- last -- the grown members' own headers
        self._pwrite(blob.getvalue(), start)
        os.fsync(self.fd)
        for m, head in headers:
            self._pwrite(head, m.offset)
        os.fsync(self.fd)
        self.tail = start + blob.tell()

    def close(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
