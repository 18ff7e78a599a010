"""Register footprint of every kernel in a built library, from the code object's own metadata (no compiler run):
    python tools/kernel_registers.py [lib.so] [--over N]
Walks the .so for embedded AMDGPU ELF code objects, reads their NT_AMDGPU_METADATA note (msgpack) and prints
`registers (of which AGPRs) name` per kernel -- on gfx950's unified register file `.vgpr_count` is the wave's whole allocation,
VGPRs and AGPRs together.  Why it matters: the one kernel of this library that needed more than 256 registers
(VGPR + AGPR: layernorm_bwd at 256 + 91) returned wrong rows whenever waves of another hardware queue shared its SIMDs
(DESIGN.md section 6); tests/test_hygiene_cpu.py holds every kernel to <= 256."""
import struct, sys

import msgpack


def code_objects(blob):
    """(offset, bytes) of every ELF64 image with e_machine == EM_AMDGPU (224) inside `blob`."""
    out, pos = [], 0
    while True:
        pos = blob.find(b"\x7fELF", pos)
        if pos < 0:
            return out
        if blob[pos + 4] == 2 and struct.unpack_from("<H", blob, pos + 18)[0] == 224:
            shoff, = struct.unpack_from("<Q", blob, pos + 40)
            shentsize, shnum = struct.unpack_from("<HH", blob, pos + 58)
            out.append((pos, blob[pos:pos + shoff + shentsize * shnum]))
        pos += 4


def kernels(elf):
    shoff, = struct.unpack_from("<Q", elf, 40)
    shentsize, shnum = struct.unpack_from("<HH", elf, 58)
    found = []
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        off, size = struct.unpack_from("<QQ", elf, sh + 24)
        if sh_type != 7:                                   # SHT_NOTE
            continue
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz].rstrip(b"\0")
            d0 = p + 12 + (namesz + 3) // 4 * 4
            if name == b"AMDGPU" and ntype == 32:          # NT_AMDGPU_METADATA
                md = msgpack.unpackb(elf[d0:d0 + descsz], raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    found.append((k.get(".vgpr_count", 0), k.get(".agpr_count", 0), k.get(".name", "?")))
            p = d0 + (descsz + 3) // 4 * 4
    return found


def library_kernels(path):
    blob = open(path, "rb").read()
    out = []
    for _, elf in code_objects(blob):
        out += kernels(elf)
    return out


if __name__ == "__main__":
    path = next((a for a in sys.argv[1:] if not a.startswith("--")), "sign-language-nlp_amd/lib/libslnlp.so")
    over = int(sys.argv[sys.argv.index("--over") + 1]) if "--over" in sys.argv else -1
    ks = sorted(library_kernels(path), key=lambda k: -k[0])
    for v, a, n in ks:
        if v > over:
            print(f"{v:4d} ({a:3d} AGPRs)  {n[:110]}")
    print(f"{len(ks)} kernels, largest register allocation = {max(v for v, _, _ in ks)}")
