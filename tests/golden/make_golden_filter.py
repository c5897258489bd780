#!/usr/bin/env python3
"""Generate tests/golden/golden_filter.json: what the UNMODIFIED reference's filter_inplace
(/root/reference/src/cr-filter.c, compiled into oracle/_ref/libcomprop_ref.so) does to seeded PE / ELF /
BMP / plain inputs, block by block, and the size + SHA-256 of `comprop -q -F e` output for a mixed file.

The filters keep their state in function statics, so every case runs in a fresh process. Only expected
OUTPUTS are committed; the inputs come from the generators in tests/crlib.py.
    python tests/golden/make_golden_filter.py
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import crlib  # noqa: E402

REF = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libcomprop_ref.so")


def cases():
    """name -> (list of generator specs concatenated into one stream, block size the stream is cut into)"""
    c = {}
    c["pe_one_block"] = ([("gen_pe", 60000, 5)], 1 << 20)
    c["pe_three_blocks"] = ([("gen_pe", 150000, 11)], 65536)
    c["pe_not_i386_dll"] = ([("gen_pe", 20000, 5, 0x8664, 3, 0x80, 0x2000)], 1 << 20)
    c["pe_not_i386_exe"] = ([("gen_pe", 20000, 5, 0x8664, 3, 0x80, 0x0002)], 1 << 20)
    c["elf_one_block"] = ([("gen_elf", 50000, 6)], 1 << 20)
    c["elf_two_blocks"] = ([("gen_elf", 100000, 9)], 65536)
    c["elf_not_386"] = ([("gen_elf", 20000, 6, 62)], 1 << 20)
    c["bmp24"] = ([("gen_bmp", 200, 120, 24, 7, b"tail" * 100)], 1 << 20)
    c["bmp32_blocks"] = ([("gen_bmp", 150, 100, 32, 8, b"")], 16384)
    c["bmp24_blocks_odd_width"] = ([("gen_bmp", 97, 83, 24, 9, b"xyz")], 10000)
    c["bmp_no_size_field"] = ([("gen_bmp", 64, 48, 24, 10, b"", False)], 1 << 20)
    c["bmp_too_small"] = ([("gen_bmp", 3, 50, 24, 7, b"")], 1 << 20)
    c["plain_text"] = ([("gen_text", 70000, 4)], 65536)
    c["mixed"] = ([("gen_text", 5000, 4), ("gen_pe", 40000, 21), ("gen_text", 3000, 5), ("gen_bmp", 80, 60, 24, 3, b""),
                   ("gen_elf", 30000, 22), ("gen_text", 2000, 6), ("gen_pe", 12000, 23)], 32768)
    # the reference never resets its ELF byte counter: the second image is converted lossily (dec_restores is false)
    c["two_elf"] = ([("gen_elf", 30000, 22), ("gen_text", 2000, 6), ("gen_elf", 12000, 23)], 1 << 20)
    # what Silesia's tarballs (mozilla, samba, ooffice) look like: many images of all kinds in one stream
    c["tar_like"] = ([("gen_text", 3000, 31), ("gen_elf", 40000, 32), ("gen_pe", 50000, 33), ("gen_bmp", 120, 90, 24, 34, b""),
                      ("gen_text", 1500, 35), ("gen_elf", 25000, 36), ("gen_bmp", 64, 64, 32, 37, b"pad"), ("gen_pe", 30000, 38),
                      ("gen_elf", 70000, 39), ("gen_text", 4000, 40)], 65536)
    return c


def build(specs):
    return b"".join(getattr(crlib, s[0])(*s[1:]) for s in specs)


CHILD = r'''
import ctypes, sys, json, hashlib
lib = ctypes.CDLL(sys.argv[1])
lib.filter_inplace.restype = ctypes.c_int
lib.filter_inplace.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int]
data = open(sys.argv[2], "rb").read()
block = int(sys.argv[3]); mode = int(sys.argv[4])
out = bytearray(); rets = []
for i in range(0, max(len(data), 1), block):
    b = data[i:i + block]
    buf = ctypes.create_string_buffer(b, len(b) + 256)      # slack: the reference reads a little past some blocks
    rets.append(lib.filter_inplace(buf, len(b), mode))
    out += buf.raw[:len(b)]
open(sys.argv[5], "wb").write(out)
print(json.dumps(rets))
'''


def run_ref(data, block, mode, tmp):
    src, dst = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    open(src, "wb").write(data)
    p = subprocess.run([sys.executable, "-c", CHILD, REF, src, str(block), str(mode), dst], capture_output=True, text=True, check=True)
    return json.loads(p.stdout.strip().splitlines()[-1]), open(dst, "rb").read()


def main():
    import tempfile
    gold = {"_about": "reference filter_inplace per block (fresh process per case) — see make_golden_filter.py", "cases": {}}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (specs, block) in cases().items():
            data = build(specs)
            rets, enc = run_ref(data, block, 0, tmp)
            rets_d, dec = run_ref(enc, block, 1, tmp)
            gold["cases"][name] = {"specs": [list(s[:1]) + [x.hex() if isinstance(x, bytes) else x for x in s[1:]] for s in specs],
                                   "block": block, "n": len(data), "in_sha256": crlib.sha(data), "returns": rets,
                                   "enc_sha256": crlib.sha(enc), "changed_bytes": sum(a != b for a, b in zip(data, enc)),
                                   "dec_returns": rets_d, "dec_sha256": crlib.sha(dec), "dec_restores": dec == data}
        # the whole tool: comprop -q -F e on the mixed and the tar-like stream (16 MiB default blocks -> one block)
        child = ("import ctypes,sys\nlib=ctypes.CDLL(sys.argv[1])\nargs=[b'comprop',b'-q',b'-F',b'e',sys.argv[2].encode(),sys.argv[3].encode()]\n"
                 "argv=(ctypes.c_char_p*(len(args)+1))(*args,None)\nsys.exit(lib.main(len(args),argv)&255)\n")
        for case in ("mixed", "tar_like"):
            data = build(cases()[case][0])
            src, dst = os.path.join(tmp, case + ".bin"), os.path.join(tmp, case + ".crop")
            open(src, "wb").write(data)
            subprocess.run([sys.executable, "-c", child, REF, src, dst], check=True)
            out = open(dst, "rb").read()
            gold[f"cli_{case}_F"] = {"n": len(data), "in_sha256": crlib.sha(data), "size": len(out), "sha256": crlib.sha(out)}
    with open(os.path.join(HERE, "golden_filter.json"), "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
    for k, v in gold["cases"].items():
        print(f"{k:26s} n={v['n']:7d} returns={v['returns']} changed={v['changed_bytes']} dec_restores={v['dec_restores']}")
    print("cli", gold["cli_mixed_F"], gold["cli_tar_like_F"])


if __name__ == "__main__":
    main()
