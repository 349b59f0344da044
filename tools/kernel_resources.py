#!/usr/bin/env python3
"""Per-kernel register / scratch figures of the HIP library, from hipcc's own resource remarks (no GPU needed).

    python tools/kernel_resources.py [file.hip ...] [--match REGEX]

Compiles each translation unit of vfa_amd/csrc with -Rpass-analysis=kernel-resource-usage (same flags as the Makefile) and prints
one line per kernel: VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, LDS bytes, waves per SIMD."""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vfa_amd", "csrc")
FLAGS = "-O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -std=c++17".split()


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def resources(path):
    cmd = ["/opt/rocm/bin/hipcc", *FLAGS, "-I" + os.path.join(ROOT, "include"), "-c", "-o", "/dev/null", path,
           "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    kernels, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line)
        if not m:
            m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        text = m.group(1).strip()
        if text.startswith("Function Name:"):
            cur = {"name": text.split(":", 1)[1].strip()}
            kernels.append(cur)
        elif cur is not None and ":" in text:
            k, v = text.split(":", 1)
            cur[k.strip()] = v.strip()
    for k, d in zip(kernels, demangle([k["name"] for k in kernels])):
        k["name"] = re.sub(r"\(anonymous namespace\)::", "", d)
        k["name"] = re.sub(r"\(.*\)$", "", k["name"]).replace("void ", "")
    return kernels


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = re.compile(sys.argv[sys.argv.index("--match") + 1])
        args = [a for a in args if a != sys.argv[sys.argv.index("--match") + 1]]
    files = args or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'LDS':>7s} {'occ':>3s}")
    for f in files:
        for k in resources(os.path.join(CSRC, os.path.basename(f))):
            if match and not match.search(k["name"]):
                continue
            print(f"{k['name'][:70]:70s} {k.get('VGPRs', '?'):>5s} {k.get('AGPRs', '?'):>5s} {k.get('SGPRs', '?'):>5s} "
                  f"{k.get('VGPRs Spill', k.get('VGPR Spill', '?')):>6s} {k.get('SGPRs Spill', k.get('SGPR Spill', '?')):>6s} "
                  f"{k.get('ScratchSize [bytes/lane]', '?'):>7s} {k.get('LDS Size [bytes/block]', '?'):>7s} {k.get('Occupancy [waves/SIMD]', '?'):>3s}")


if __name__ == "__main__":
    main()
