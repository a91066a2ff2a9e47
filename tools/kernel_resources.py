# Register / scratch budget of every kernel in the BUILT library, read from the gfx950 code objects inside libemdr2_hip.so (no GPU needed):
#   python tools/kernel_resources.py [path/to/libemdr2_hip.so] [--count v_mov_b64 --in attention_fwd]
# One line per kernel: VGPRs, AGPRs, SGPRs, spills, scratch bytes, static LDS.  tests/test_isa_budget.py pins the numbers the design rests on
# (occupancy of the attention kernels, no scratch anywhere, no register copies around the attention forward's PV product: DESIGN 5.6).
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "emdr2_amd", "lib", "libemdr2_hip.so")
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def code_objects(lib_path, arch="gfx950"):
    """The device ELF images of `arch` embedded in a HIP shared library (one clang offload bundle per translation unit)."""
    data = open(lib_path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", data, i + 24)[0]
        o = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if arch in triple and size:
                out.append(data[i + off:i + off + size])
        pos = i + 24


def kernels(lib_path=DEFAULT_LIB):
    """[{name, vgpr_count, ...}] for every kernel of the library (llvm-readelf --notes on each code object)."""
    res = []
    with tempfile.TemporaryDirectory() as tmp:
        for k, img in enumerate(code_objects(lib_path)):
            path = os.path.join(tmp, "co%d.elf" % k)
            open(path, "wb").write(img)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True, check=True).stdout
            for block in notes.split("  - .agpr_count:")[1:]:
                block = ".agpr_count:" + block
                ent = {"object": k}
                m = re.search(r"^\s+\.name:\s+(\S+)", block, re.M)
                ent["name"] = m.group(1).strip("'\"") if m else "?"
                for f in FIELDS:
                    m = re.search(r"\.%s:\s+(\d+)" % f, block)
                    ent[f] = int(m.group(1)) if m else 0
                res.append(ent)
    return res


def count_opcode(lib_path, opcode, symbol_substr):
    """{kernel symbol: occurrences of `opcode`} over the disassembly of the kernels whose symbol contains `symbol_substr`."""
    counts = {}
    with tempfile.TemporaryDirectory() as tmp:
        for k, img in enumerate(code_objects(lib_path)):
            if symbol_substr.encode() not in img:
                continue
            path = os.path.join(tmp, "co%d.elf" % k)
            open(path, "wb").write(img)
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", path], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1) if symbol_substr in m.group(1) else None
                    if cur:
                        counts.setdefault(cur, 0)
                elif cur and re.search(r"\b%s(_e32|_e64|_sdwa|_dpp)?\b" % re.escape(opcode), line):
                    counts[cur] += 1
    return counts


if __name__ == "__main__":
    args = [a for a in sys.argv[1:]]
    lib = DEFAULT_LIB
    if args and not args[0].startswith("--"):
        lib = args.pop(0)
    if "--count" in args:
        op = args[args.index("--count") + 1]
        sub = args[args.index("--in") + 1] if "--in" in args else ""
        for name, c in sorted(count_opcode(lib, op, sub).items()):
            print("%6d  %s" % (c, name))
        sys.exit(0)
    print("%5s %5s %5s %6s %6s %8s %7s  kernel" % ("vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
    for e in sorted(kernels(lib), key=lambda e: e["name"]):
        print("%5d %5d %5d %6d %6d %8d %7d  %s" % (e["vgpr_count"], e["agpr_count"], e["sgpr_count"], e["vgpr_spill_count"], e["sgpr_spill_count"],
                                                    e["private_segment_fixed_size"], e["group_segment_fixed_size"], e["name"]))
