#!/usr/bin/env python3
"""Per-kernel ISA fingerprints of a HIP object: the gfx950 code object is taken out of the .hip_fatbin section, disassembled, and every
function's instruction stream (mnemonics + operands, no addresses, no encodings, no symbol names) is hashed.
    kernel_isa_sha.py <a.o>            -> one line per kernel: sha256[:16]  instructions  name
    kernel_isa_sha.py <before.o> <after.o>
        -> which kernels of `after` have an instruction stream that some kernel of `before` has too (template-parameter lists may have
           changed, so names are not compared), which are new, and which of `before` are gone.
Used by round 6's prune: removing compile-time knobs must leave every surviving kernel's ISA untouched."""
import hashlib
import re
import subprocess
import sys
import tempfile
from collections import Counter
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def kernels(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = f"{tmp}/fat.bin", f"{tmp}/gfx950.co"
        subprocess.run([LLVM / "llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj, "/dev/null"], check=True)
        subprocess.run([LLVM / "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
                        f"--output={co}", "--unbundle"], check=True)
        dis = subprocess.run([LLVM / "llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", "-C", co], capture_output=True,
                             text=True, check=True).stdout
    out, name, body = {}, None, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]* ?<(.*)>:$", line.strip())
        if m:
            if name is not None:
                out[name] = body
            name, body = m.group(1), []
            continue
        if name is None or not line.strip():
            continue
        ins = line.split("//")[0].strip()
        ins = re.sub(r"<[^>]*>", "", ins)                 # symbolic branch targets
        if ins:
            body.append(ins)
    if name is not None:
        out[name] = body
    return {k: (hashlib.sha256("\n".join(v).encode()).hexdigest()[:16], len(v)) for k, v in out.items() if v}


def main():
    if len(sys.argv) == 2:
        for k, (h, n) in sorted(kernels(sys.argv[1]).items()):
            print(h, n, k)
        return 0
    before, after = kernels(sys.argv[1]), kernels(sys.argv[2])
    pool = Counter(h for h, _ in before.values())
    same, new = [], []
    for k, (h, n) in sorted(after.items()):
        if pool[h] > 0:
            pool[h] -= 1
            same.append(k)
        else:
            new.append((k, n))
    gone = []
    left = Counter(pool)
    for k, (h, n) in sorted(before.items()):
        if left[h] > 0:
            left[h] -= 1
            gone.append((k, n))
    print(f"{len(after)} functions after, {len(before)} before: {len(same)} with an instruction stream identical to one of before's")
    for k, n in new:
        print(f"  CHANGED/NEW  {n:7d} instructions  {k}")
    for k, n in gone:
        print(f"  GONE/CHANGED {n:7d} instructions  {k}")
    return 1 if new else 0


if __name__ == "__main__":
    sys.exit(main())
