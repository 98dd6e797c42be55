"""What machine code the built library holds, kernel by kernel (no GPU, no external tool).

    python -m parakeet_slam_amd.codeobj [--json] [path/to/libparakeet_slam.so]

``kernel_hashes(so)`` -> {mangled kernel symbol: first 16 hex digits of the SHA-256 of its instructions}, read from the gfx950 code
objects (ELF images, e_machine EM_AMDGPU) embedded in the shared library.  bench.py replays counter values measured in earlier
rocprofv3 passes (profiles/*/pmc_*.json: HBM traffic, instruction counts); each of those files records the hashes of the kernels
it measured, and the bench line carries a replayed value only while the library it loaded still holds exactly those instructions
("bench.py does not check that the replayed file's git matches the library it loaded")."""
from __future__ import annotations

import hashlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libparakeet_slam.so")
EM_AMDGPU = 224


def _code_objects(data):
    idx = 0
    while True:
        i = data.find(b"\x7fELF", idx)
        if i < 0:
            return
        idx = i + 4
        if len(data) < i + 0x40 or data[i + 4] != 2 or struct.unpack_from("<H", data, i + 18)[0] != EM_AMDGPU:
            continue
        e_shoff = struct.unpack_from("<Q", data, i + 0x28)[0]
        e_shentsize, e_shnum, e_shstrndx = struct.unpack_from("<HHH", data, i + 0x3A)
        yield data[i:i + e_shoff + e_shentsize * e_shnum], e_shoff, e_shentsize, e_shnum, e_shstrndx


def kernel_hashes(so=LIB):
    """{kernel symbol: sha256 of its instruction bytes, 16 hex digits} over every gfx950 code object in `so`."""
    data = open(so, "rb").read()
    out = {}
    for img, shoff, shentsize, shnum, _strndx in _code_objects(data):
        secs = []
        for k in range(shnum):
            name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", img, shoff + k * shentsize)
            secs.append(dict(type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
        for sec in secs:
            if sec["type"] != 2:  # SHT_SYMTAB
                continue
            strtab = secs[sec["link"]]
            n = sec["size"] // max(sec["entsize"], 24)
            for k in range(n):
                st_name, st_info, _other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", img, sec["off"] + 24 * k)
                if (st_info & 0xF) != 2 or st_size == 0 or st_shndx == 0 or st_shndx >= len(secs):  # STT_FUNC
                    continue
                end = img.index(b"\0", strtab["off"] + st_name)
                sym = img[strtab["off"] + st_name:end].decode("ascii", "replace")
                text = secs[st_shndx]
                o = text["off"] + (st_value - text["addr"])
                out[sym] = hashlib.sha256(img[o:o + st_size]).hexdigest()[:16]
    return out


def combined(hashes, needles):
    """One 16-digit figure for the kernels whose symbol contains any of `needles` (sorted by symbol), or None when there is none."""
    picked = sorted((k, v) for k, v in hashes.items() if any(n in k for n in needles))
    if not picked:
        return None
    h = hashlib.sha256()
    for k, v in picked:
        h.update(("%s=%s;" % (k, v)).encode())
    return h.hexdigest()[:16]


# the kernels behind the counter files' keys (profiles/*/pmc_*.json) and bench.py's routes
KERNEL_SYMBOLS = {
    "k_step_pub<2, 512>": ["k_step_pubILi2ELi512E"],
    "k_step_pub<1, 256>": ["k_step_pubILi1ELi256E"],
    "k_step_pub_big": ["k_step_pub_bigILi"],
    "k_step_pub_duo": ["k_step_pub_duoILi"],
    "k_step_fused": ["k_step_fused"],
    "k_step_regs": ["k_step_regs"],
    "k_observe": ["9k_observeILb"],
}


def figures(so=LIB):
    h = kernel_hashes(so)
    return {k: combined(h, v) for k, v in KERNEL_SYMBOLS.items()}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    so = args[0] if args else LIB
    if "--json" in sys.argv:
        print(json.dumps(figures(so)))
    else:
        for k, v in sorted(kernel_hashes(so).items()):
            print(v, k)
