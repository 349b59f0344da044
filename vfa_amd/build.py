"""Compile the HIP library in-tree: ``python -m vfa_amd.build``.

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the build container; the
resulting ``vfa_amd/csrc/libvfa_hip.so`` travels to the GPU box with the source tree.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")


def build(force=False, verbose=False):
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else []) + ["libvfa_hip.so"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stdout.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("hipcc build of libvfa_hip.so failed")
    return os.path.join(CSRC, "libvfa_hip.so")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
