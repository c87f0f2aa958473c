"""SURVEY §8 f4 — encryption-hook ordering.  The reference encrypts AFTER the codec (DeflaterOutputStream.EncryptBlock on every
block Deflate() returned, CS/DeflaterOutputStream.cs:227,256,111) and decrypts BEFORE it (InflaterInputBuffer.CryptoTransform,
CS/InflaterInputStream.cs:131-138,276-305).  The hooks stay on the host, unchanged; these tests run a PKZIP-classic stream
cipher (S/Encryption/PkzipClassic.cs: the test double below) through the mirrors over the DEVICE codec and check that the
ciphertext is exactly encrypt(compressed bytes of the reference Deflater), block order included."""
import io
import zlib

import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu


class PkzipClassic:
    """PKZIP classic stream cipher (APPNOTE 6.1; S/Encryption/PkzipClassic.cs:354-460), as an ICryptoTransform test double."""

    def __init__(self, password, encrypt):
        self.k = [0x12345678, 0x23456789, 0x34567890]
        self.encrypt = encrypt
        for c in password:
            self._update(c)

    def _update(self, ch):
        k = self.k
        k[0] = zlib.crc32(bytes([ch]), k[0] ^ 0xFFFFFFFF) ^ 0xFFFFFFFF
        k[1] = (k[1] + (k[0] & 0xFF)) & 0xFFFFFFFF
        k[1] = (k[1] * 134775813 + 1) & 0xFFFFFFFF
        k[2] = zlib.crc32(bytes([k[1] >> 24]), k[2] ^ 0xFFFFFFFF) ^ 0xFFFFFFFF

    def _byte(self):
        t = (self.k[2] & 0xFFFF) | 2
        return ((t * (t ^ 1)) >> 8) & 0xFF

    def TransformBlock(self, inb, inoff, count, outb, outoff):
        for i in range(count):
            v = int(inb[inoff + i])
            if self.encrypt:
                outb[outoff + i] = v ^ self._byte()
                self._update(v)
            else:
                p = v ^ self._byte()
                outb[outoff + i] = p
                self._update(p)
        return count


@pytest.mark.parametrize("read_ahead", [0, None])
def test_encrypt_after_deflate_decrypt_before_inflate(read_ahead):
    from sharpziplib_amd.deflater import Deflater
    from sharpziplib_amd.inflater import Inflater
    from sharpziplib_amd.streams import DeflaterOutputStream, InflaterInputStream
    data = C.generate("dickens", 0xF4, 0, 200000)
    pw = b"correct horse"
    sink = io.BytesIO()
    sink.close = lambda: None
    dos = DeflaterOutputStream(sink, Deflater(6, True), 512)
    dos.cryptoTransform_ = PkzipClassic(pw, True)
    for off in range(0, data.size, 30000):
        dos.Write(data[off:off + 30000])
    dos.Finish()
    cipher = sink.getvalue()
    plain = O.deflate(data, 6)                                     # what the reference Deflater emits for this input
    ref = np.zeros(len(plain), np.uint8)
    PkzipClassic(pw, True).TransformBlock(np.frombuffer(plain, np.uint8), 0, len(plain), ref, 0)
    assert cipher == ref.tobytes()                                 # same bytes, same order: the hook saw exactly the codec's output
    assert cipher != plain and dos.cryptoTransform_ is None
    # read side: the hook decrypts each filled buffer before SetInput
    iis = InflaterInputStream(io.BytesIO(cipher), Inflater(True), 4096, readAhead=read_ahead)
    iis.inputBuffer.SetCryptoTransform(PkzipClassic(pw, False))
    out = np.zeros(data.size, np.uint8)
    got = 0
    while got < data.size:
        n = iis.Read(out, got, data.size - got)
        if n <= 0:
            break
        got += n
    assert got == data.size and out.tobytes() == data.tobytes()
