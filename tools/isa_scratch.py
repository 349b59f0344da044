#!/usr/bin/env python3
"""Where the scratch (spill) instructions of the persistent kernels sit: per kernel, the scratch_load / scratch_store instructions of
the gfx950 ISA grouped by the source region (lambda or function of the .hip file) they were generated for.   No GPU needed.

    python tools/isa_scratch.py [file.hip ...] [--match REGEX] > profiles/rNN_isa_scratch.txt

Each translation unit is compiled to assembly with the Makefile's flags + -g1 (line tables only: same code) and every scratch
instruction is attributed to the innermost line of that .hip file in its inline chain; a line belongs to the innermost enclosing
`auto name = [&]...` lambda (or to the kernel body).  Regions that run once per launch or once per tile hand-off are named in the
source comments; the STEP / ITEM loops are the regions `multiply`, `kstep`, `pool`, `pool_step`, `step_dma`, `finish`, `pair`."""
import os, re, subprocess, sys, tempfile
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vfa_amd", "csrc")
FLAGS = "-O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -std=c++17 -g1 -S --cuda-device-only".split()


def regions_of(path):
    """line number -> name of the innermost enclosing lambda / kernel (by indentation of `auto x = [&]` ... `};`)."""
    lines = open(path).read().split("\n")
    stack, out = [], {}
    for i, line in enumerate(lines, 1):
        ind = len(line) - len(line.lstrip())
        text = line.strip()
        while stack and text.startswith("}") and ind <= stack[-1][1]:
            out[i] = stack[-1][0]
            stack.pop()
            break
        m = re.match(r"auto (\w+) = \[[&=]?\]", text) or re.match(r"(?:template <[^>]*>\s*)?__global__ .* (\w+)\(", text)
        if m and not text.endswith(";"):
            stack.append((m.group(1), ind))
        elif m and text.endswith("};") is False and "{" in text and "}" not in text.split("{", 1)[1]:
            stack.append((m.group(1), ind))
        out.setdefault(i, stack[-1][0] if stack else "(file scope)")
    return out


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()


def analyse(src, match):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I" + os.path.join(ROOT, "include"), "-o", asm, src], check=True, capture_output=True, cwd=CSRC)
        text = open(asm).read().split("\n")
    regions = regions_of(src)
    base = os.path.basename(src)
    cur_fn, cur_line = None, None
    per = defaultdict(Counter)
    lines_of = defaultdict(Counter)
    totals = Counter()
    for line in text:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur_fn, cur_line = m.group(1), None
            continue
        if "\t.loc\t" in line and ";" in line:
            hits = re.findall(re.escape(base) + r":(\d+):", line.split(";", 1)[1])
            cur_line = int(hits[0]) if hits else cur_line  # the innermost frame of the inline chain that lies in this file
            continue
        ins = line.strip().split(" ")[0].split("\t")[0]
        if ins.startswith("scratch_") and cur_fn:
            kind = "load" if "load" in ins else "store"
            per[cur_fn][(regions.get(cur_line, "?"), kind)] += 1
            lines_of[cur_fn][cur_line] += 1
            totals[cur_fn] += 1
    for fn, c in per.items():
        name = re.sub(r"\(anonymous namespace\)::", "", demangle(fn))
        name = re.sub(r"\(.*\)$", "", name).replace("void ", "")
        if match and not re.search(match, name):
            continue
        print(f"{base}: {name}: {totals[fn]} scratch instructions")
        by_region = defaultdict(lambda: [0, 0])
        for (region, kind), n in c.items():
            by_region[region][0 if kind == "load" else 1] += n
        for region, (ld, st) in sorted(by_region.items(), key=lambda kv: -sum(kv[1])):
            print(f"    {region:28s} loads {ld:4d}  stores {st:4d}")
        if "--lines" in sys.argv:
            print("    by source line: " + ", ".join(f"{ln}: {n}" for ln, n in sorted(lines_of[fn].items(), key=lambda kv: (kv[0] is None, kv[0]))))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    for f in args or ["vfa_pipe.hip", "vfa_collapse_gemm.hip", "vfa_fused.hip"]:
        analyse(os.path.join(CSRC, os.path.basename(f)), match)


if __name__ == "__main__":
    main()
