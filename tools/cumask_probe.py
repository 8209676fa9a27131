"""which physical CUs a CU-masked stream uses (tools/native/cumask_probe.hip)"""
import ctypes, os, sys, collections
import numpy as np
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "cumask_probe.so"))
def run(bits, nwg=4096, spin=20):
    out = np.zeros(2 * nwg, dtype=np.uint32)
    if bits is None:
        rc = lib.cumask_probe(None, 0, nwg, spin, out.ctypes.data_as(ctypes.c_void_p))
    else:
        words = 8
        mask = np.zeros(words, dtype=np.uint32)
        for b in bits:
            mask[b // 32] |= np.uint32(1 << (b % 32))
        rc = lib.cumask_probe(mask.ctypes.data_as(ctypes.c_void_p), words, nwg, spin, out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    hw, xcc = out[0::2], out[1::2] & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5)
    per_xcc = collections.Counter()
    for x, c in set(zip(xcc.tolist(), cu.tolist())):
        per_xcc[x] += 1
    return sum(per_xcc.values()), dict(sorted(per_xcc.items()))
print("no mask:", run(None))
print("bits 0..63:", run(range(64)))
print("bits 0..127:", run(range(128)))
print("bits 0..191:", run(range(192)))
print("bits 64..255:", run(range(64, 256)))
print("every 4th bit (64 of 256):", run(range(0, 256, 4)))
print("bits with (b % 32) < 24 (192):", run([b for b in range(256) if b % 32 < 24]))
