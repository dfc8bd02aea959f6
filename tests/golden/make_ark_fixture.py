#!/usr/bin/env python3
"""An x-vector archive the build did NOT write: hand-assembled here from Kaldi's documented byte layout, with `struct`
only (nothing of neuralplda_amd is imported), and committed as data next to this script.

Layout (Kaldi I/O docs, "Kaldi I/O mechanisms" / kaldi-vector.cc Vector<float>::Write, as kaldi_io.read_vec_flt parses
it — the call the reference makes per utterance, /root/reference/dataprep_sre.py:152-167):
  binary archive : for every entry   <key> <space> \\0 B  F V <space>  \\x04 <int32 little-endian dim>  dim x float32 LE
  script file    : <key> <space> <path>:<offset>\\n   with offset = byte position of the \\0B marker (just after the blank)
  text archive   : <key> <space><space>[ v1 v2 ... ]\\n
Values are a closed formula plus a few awkward bit patterns (negative zero, a denormal, the largest finite float, a value
needing all 9 significant digits), so the test can recompute them without any reader.

    python tests/golden/make_ark_fixture.py      # rewrites tests/golden/xvec_fixture.{ark,scp,txt.ark}
"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
DIM = 8
KEYS = ["id10001-utt_a", "id10001-utt_b", "sw_4021-B_0003", "x", "spk99-long.key-with.dots_and-dashes"]
SPECIAL = {  # (entry, component) -> float32 bit pattern
    (1, 0): 0x80000000,  # -0.0
    (2, 3): 0x00000001,  # smallest denormal
    (3, 7): 0x7F7FFFFF,  # largest finite
    (4, 2): 0x3DFCD6EA,  # 0.12345679
}


def value_bits(i, j):
    if (i, j) in SPECIAL:
        return SPECIAL[(i, j)]
    return struct.unpack("<I", struct.pack("<f", (i + 1) * 0.25 - j * 0.125 + (1 if (i + j) % 3 == 0 else -1) * 1e-3 * (i * DIM + j)))[0]


def main():
    ark = b""
    scp = ""
    txt = ""
    for i, key in enumerate(KEYS):
        ark += key.encode("ascii") + b" "
        off = len(ark)
        ark += b"\0B" + b"FV " + b"\x04" + struct.pack("<i", DIM)
        vals = []
        for j in range(DIM):
            bits = value_bits(i, j)
            ark += struct.pack("<I", bits)
            vals.append(struct.unpack("<f", struct.pack("<I", bits))[0])
        scp += f"{key} xvec_fixture.ark:{off}\n"
        txt += key + "  [ " + " ".join(repr(v) if v == v else "nan" for v in (float(struct.unpack('<f', struct.pack('<f', v))[0]) for v in vals)) + " ]\n"
    with open(os.path.join(HERE, "xvec_fixture.ark"), "wb") as fh:
        fh.write(ark)
    with open(os.path.join(HERE, "xvec_fixture.scp"), "w") as fh:
        fh.write(scp)
    with open(os.path.join(HERE, "xvec_fixture.txt.ark"), "w") as fh:
        fh.write(txt)
    print(f"wrote {len(KEYS)} vectors of dim {DIM}: {len(ark)} bytes of archive")


if __name__ == "__main__":
    main()
