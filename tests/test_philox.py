"""Known-answer tests of the random stream: Philox4x32-10 (Random123 kat_vectors), which seeds every
(pixel, sample), and xorshift128 (Marsaglia 2003, xor128), which produces the sample's draws --
for the product's generator and the checker's independent implementation."""
import ctypes as C

KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
     [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


def test_product_philox_kat(rtmi):
    for ctr, key, want in KAT:
        assert rtmi.philox4x32_10(ctr, key) == want


def test_checker_philox_kat(rtcheck):
    lib = rtcheck.oracle_lib()
    for ctr, key, want in KAT:
        out = (C.c_uint32 * 4)()
        lib.rto_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
        assert list(out) == want


def test_product_and_checker_agree_on_random_counters(rtmi, rtcheck):
    import random
    rnd = random.Random(1)
    lib = rtcheck.oracle_lib()
    for _ in range(200):
        ctr = [rnd.getrandbits(32) for _ in range(4)]
        key = [rnd.getrandbits(32) for _ in range(2)]
        out = (C.c_uint32 * 4)()
        lib.rto_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
        assert list(out) == rtmi.philox4x32_10(ctr, key)


def _xor128_python(x, y, z, w, n):
    out = []
    for _ in range(n):
        t = (x ^ (x << 11)) & 0xFFFFFFFF
        x, y, z = y, z, w
        w = (w ^ (w >> 19) ^ (t ^ (t >> 8))) & 0xFFFFFFFF
        out.append(w)
    return out


def test_xor128_known_answer():
    # the sequence printed in Marsaglia's paper for the default seed
    assert _xor128_python(123456789, 362436069, 521288629, 88675123, 5) == [
        3701687786, 458299110, 2500872618, 3633119408, 516391518]


def test_sample_stream_is_philox_seeded_xor128(rtmi, rtcheck):
    lib = rtcheck.oracle_lib()
    for seed, pixel, sample in ((2023, 0, 0), (2023, 123456, 7), ((5 << 32) | 9, 2073599, 1023), (0, 0, 0)):
        state = rtmi.philox4x32_10([pixel, sample, 0, 0], [seed & 0xFFFFFFFF, seed >> 32])
        want = _xor128_python(*state, 64)
        assert rtmi.sample_stream(seed, pixel, sample, 64) == want
        out = (C.c_uint32 * 64)()
        lib.rto_sample_stream(seed, pixel, sample, out, 64)
        assert list(out) == want
    # streams of neighbouring pixels / samples are unrelated
    a = rtmi.sample_stream(1, 10, 3, 8)
    assert a != rtmi.sample_stream(1, 11, 3, 8) and a != rtmi.sample_stream(1, 10, 4, 8) and a != rtmi.sample_stream(2, 10, 3, 8)
