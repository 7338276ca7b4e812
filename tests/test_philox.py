"""Philox4x32-10 known-answer tests (Random123 kat_vectors) for the product's generator and
the checker's independent implementation."""
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
