"""FarmHash Fingerprint64 (farmhashna::Hash64) in pure Python integers.  Oracle: tests only.

TensorFlow's ``categorical_column_with_hash_bucket`` (reference ``trainers/ml_100k.py:19,20,29,30``)
maps a value to ``Fingerprint64(bytes) mod hash_bucket_size``.  TensorFlow and FarmHash
are not in the reference tree; this follows FarmHash's published algorithm (farmhash.cc,
namespace farmhashna, v1.1).  Written independently of ``csrc/host_ids.cpp`` in style (explicit
masking, slices) so that the two restatements check each other.

Pinned by external known answers (tests/test_fingerprint.py): the 1-3 byte branch by TensorFlow's own op test ('a'..'d') and
pyfarmhash's README ('abc'); the 4-7 and 8-16 byte branches — every MovieLens item id, zipcode and occupation — by the three
full 64-bit FARM_FINGERPRINT values of BigQuery's reference page (round 5); 17 bytes and more: unpinned (C and Python
restatements agree with each other only; no ML-100k feature is that long).
"""
M64 = (1 << 64) - 1
K0 = 0xC3A5C85C97CB3127
K1 = 0xB492B66FBE98F273
K2 = 0x9AE16A3B2F90404F


def _f64(s, i):
    return int.from_bytes(s[i:i + 8], "little")


def _f32(s, i):
    return int.from_bytes(s[i:i + 4], "little")


def _rot(v, s):
    return v if s == 0 else ((v >> s) | (v << (64 - s))) & M64


def _smix(v):
    return v ^ (v >> 47)


def _h16(u, v, mul):
    a = ((u ^ v) * mul) & M64
    a ^= a >> 47
    b = ((v ^ a) * mul) & M64
    b ^= b >> 47
    return (b * mul) & M64


def _len0to16(s):
    n = len(s)
    if n >= 8:
        mul = (K2 + n * 2) & M64
        a = (_f64(s, 0) + K2) & M64
        b = _f64(s, n - 8)
        c = (_rot(b, 37) * mul + a) & M64
        d = ((_rot(a, 25) + b) * mul) & M64
        return _h16(c, d, mul)
    if n >= 4:
        mul = (K2 + n * 2) & M64
        a = _f32(s, 0)
        return _h16((n + (a << 3)) & M64, _f32(s, n - 4), mul)
    if n > 0:
        a, b, c = s[0], s[n >> 1], s[n - 1]
        y = (a + (b << 8)) & 0xFFFFFFFF
        z = (n + (c << 2)) & 0xFFFFFFFF
        return (_smix(((y * K2) & M64) ^ ((z * K0) & M64)) * K2) & M64
    return K2


def _len17to32(s):
    n = len(s)
    mul = (K2 + n * 2) & M64
    a = (_f64(s, 0) * K1) & M64
    b = _f64(s, 8)
    c = (_f64(s, n - 8) * mul) & M64
    d = (_f64(s, n - 16) * K2) & M64
    return _h16((_rot((a + b) & M64, 43) + _rot(c, 30) + d) & M64,
                (a + _rot((b + K2) & M64, 18) + c) & M64, mul)


def _len33to64(s):
    n = len(s)
    mul = (K2 + n * 2) & M64
    a = (_f64(s, 0) * K2) & M64
    b = _f64(s, 8)
    c = (_f64(s, n - 8) * mul) & M64
    d = (_f64(s, n - 16) * K2) & M64
    y = (_rot((a + b) & M64, 43) + _rot(c, 30) + d) & M64
    z = _h16(y, (a + _rot((b + K2) & M64, 18) + c) & M64, mul)
    e = (_f64(s, 16) * mul) & M64
    f = _f64(s, 24)
    g = ((y + _f64(s, n - 32)) * mul) & M64
    h = ((z + _f64(s, n - 24)) * mul) & M64
    return _h16((_rot((e + f) & M64, 43) + _rot(g, 30) + h) & M64,
                (e + _rot((f + a) & M64, 18) + g) & M64, mul)


def _weak32(s, i, a, b):
    w, x, y, z = _f64(s, i), _f64(s, i + 8), _f64(s, i + 16), _f64(s, i + 24)
    a = (a + w) & M64
    b = _rot((b + a + z) & M64, 21)
    c = a
    a = (a + x + y) & M64
    b = (b + _rot(a, 44)) & M64
    return (a + z) & M64, (b + c) & M64


def fingerprint64(data):
    """farmhashna::Hash64(data) for ``bytes``/``str`` (utf-8)."""
    s = data.encode("utf-8") if isinstance(data, str) else bytes(data)
    n = len(s)
    if n <= 16:
        return _len0to16(s)
    if n <= 32:
        return _len17to32(s)
    if n <= 64:
        return _len33to64(s)
    seed = 81
    x = seed
    y = (seed * K1 + 113) & M64
    z = (_smix((y * K2 + 113) & M64) * K2) & M64
    v = (0, 0)
    w = (0, 0)
    x = (x * K2 + _f64(s, 0)) & M64
    end = ((n - 1) // 64) * 64
    last64 = end + ((n - 1) & 63) - 63
    p = 0
    while True:
        x = (_rot((x + y + v[0] + _f64(s, p + 8)) & M64, 37) * K1) & M64
        y = (_rot((y + v[1] + _f64(s, p + 48)) & M64, 42) * K1) & M64
        x ^= w[1]
        y = (y + v[0] + _f64(s, p + 40)) & M64
        z = (_rot((z + w[0]) & M64, 33) * K1) & M64
        v = _weak32(s, p, (v[1] * K1) & M64, (x + w[0]) & M64)
        w = _weak32(s, p + 32, (z + w[1]) & M64, (y + _f64(s, p + 16)) & M64)
        z, x = x, z
        p += 64
        if p == end:
            break
    mul = (K1 + ((z & 0xFF) << 1)) & M64
    p = last64
    w = ((w[0] + ((n - 1) & 63)) & M64, w[1])
    v = ((v[0] + w[0]) & M64, v[1])
    w = ((w[0] + v[0]) & M64, w[1])
    x = (_rot((x + y + v[0] + _f64(s, p + 8)) & M64, 37) * mul) & M64
    y = (_rot((y + v[1] + _f64(s, p + 48)) & M64, 42) * mul) & M64
    x ^= (w[1] * 9) & M64
    y = (y + v[0] * 9 + _f64(s, p + 40)) & M64
    z = (_rot((z + w[0]) & M64, 33) * mul) & M64
    v = _weak32(s, p, (v[1] * mul) & M64, (x + w[0]) & M64)
    w = _weak32(s, p + 32, (z + w[1]) & M64, (y + _f64(s, p + 16)) & M64)
    z, x = x, z
    return _h16((_h16(v[0], w[0], mul) + ((_smix(y) * K0) & M64) + z) & M64,
                (_h16(v[1], w[1], mul) + x) & M64, mul)
