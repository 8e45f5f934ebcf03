#!/usr/bin/env python
"""Where do a kernel's vector-memory requests sit relative to its waits?  (no GPU needed)

    python tools/isa_scan.py [file.hip ...] [--kernel NAME_PART]

Compiles csrc/*.hip (or the files given) to gfx950 assembly with the product's flags (`hipcc --cuda-device-only -S`) and prints, per kernel, its
vector-memory instructions and `s_waitcnt vmcnt(N)` in program order as one token string:

    L  a global / buffer / flat load        S  a store or global atomic        0..9, (N)  s_waitcnt vmcnt(N)        |  a basic-block label

`LLLL3210` is four requests in flight, consumed one by one; `L0L0L0L0` is the same source after the compiler has moved every load behind the test that
guards its only use (LLVM's sinking) or down to its first use (the machine scheduler): one request in flight whatever the source's unroll depth says.
Round 5, fourth session: k_scatter_binned's "four entries ahead" were `L0` x 4 (LABNOTES 17a); the count in the last column (`L0` pairs) is where to look first.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_signature_amd import build as b      # noqa: E402  (flags_for: the per-file flags of the product build)


def tokens(asm_path):
    name, toks, out = None, [], []
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, toks = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            out.append((name, "".join(toks)))
            name = None
            continue
        if re.search(r"\b(global_load|buffer_load|flat_load)", line):
            toks.append("L")
        elif re.search(r"\b(global_store|buffer_store|flat_store|global_atomic)", line):
            toks.append("S")
        elif "s_waitcnt" in line and "vmcnt" in line:
            n = re.search(r"vmcnt\((\d+)\)", line).group(1)
            toks.append(n if len(n) == 1 else f"({n})")
        elif re.match(r"^\.LBB", line):
            toks.append("|")
    return out


def demangled(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return [re.sub(r"^void ", "", re.sub(r"\(.*", "", x)).replace("nsig::", "") for x in p.stdout.splitlines()]
    except Exception:
        return names


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    part = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else None
    if part in args:
        args.remove(part)
    files = args or b.sources()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        for src in files:
            asm = os.path.join(tmp, os.path.basename(src) + ".s")
            flags = [f for f in b.flags_for(src) if f not in ("-fPIC", "-fvisibility=hidden")]
            subprocess.run([hipcc, *flags, "--cuda-device-only", "-S", src, "-o", asm], check=True, stderr=subprocess.DEVNULL)
            rows = tokens(asm)
            names = demangled([r[0] for r in rows])
            print(f"== {os.path.relpath(src, ROOT)}")
            for (_, t), n in zip(rows, names):
                if part and part not in n:
                    continue
                print(f"{n[:56]:56s} loads {t.count('L'):3d}  L0 pairs {len(re.findall('L0', t)):3d}  {t[:170]}")


if __name__ == "__main__":
    main()
