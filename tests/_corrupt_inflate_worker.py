"""Worker of tests/test_inflate.py::test_inflate_survives_corrupted_payloads (own process: a GPU fault or hang is an exit code /
a timeout). Random bit flips, truncations and garbage in raw-deflate payloads: the GPU decoder agrees with zlib whenever zlib
decodes the payload to exactly the announced size, and otherwise reports the member instead of writing outside it. Usage: SEED N"""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from volcanosv_amd.abi import VsvError  # noqa: E402
from volcanosv_amd.engine import Engine  # noqa: E402

if __name__ == "__main__":
    rng = np.random.default_rng(int(sys.argv[1]))
    agree = rejected = zlib_only = 0
    with Engine(0) as eng:
        for it in range(int(sys.argv[2])):
            kind = int(rng.integers(0, 3))
            n = int(rng.integers(1, 65000))
            if kind == 0:
                data = rng.integers(0, 4, n).astype(np.uint8).tobytes()                    # compressible
            elif kind == 1:
                data = (b"ACGTTGCA" * (n // 8 + 1))[:n]                                    # long matches
            else:
                data = rng.integers(0, 256, n).astype(np.uint8).tobytes()                  # literals / stored
            c = zlib.compressobj(int(rng.choice([0, 1, 6, 9])), zlib.DEFLATED, -15, 9, int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY])))
            payload = bytearray(c.compress(data) + c.flush())
            mode = int(rng.integers(0, 4))
            if mode == 0:
                for _ in range(int(rng.integers(1, 4))):
                    k = int(rng.integers(0, len(payload)))
                    payload[k] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:
                del payload[int(rng.integers(0, len(payload))):]
                payload += b"\0"
            elif mode == 2:
                k = int(rng.integers(0, len(payload)))
                payload[k:k + 8] = bytes(rng.integers(0, 256, 8).astype(np.uint8))
            isize = n if rng.random() < 0.8 else int(rng.integers(0, 65536))              # sometimes the announced size lies
            try:
                d = zlib.decompressobj(-15)
                want = d.decompress(bytes(payload))
                z_ok = d.eof and len(want) == isize
            except zlib.error:
                z_ok = False
            try:
                got = eng.bgzf_inflate([bytes(payload)], [isize])[0]
                g_ok = True
            except VsvError:
                g_ok = False
            if z_ok:
                assert g_ok and got == want, (it, mode, len(payload), isize)
                agree += 1
            elif g_ok:
                zlib_only += 1        # zlib stops at trailing garbage / a missing end; the GPU filled exactly isize bytes: tolerated
            else:
                rejected += 1
    print("agree %d rejected %d gpu_accepts_what_zlib_rejects %d" % (agree, rejected, zlib_only))
