"""Generator of the woven quarter-step body (vfa_weave_body.inc) of the one-wave-per-SIMD frame kernel.

One wave per SIMD (512 registers: the fp16 hi + lo weight of 64 output columns in 256 of them) has nobody to cover its stalls, so
the instruction stream of a quarter-step is laid out BY HAND: 24 MFMAs (k-steps 0..3 of the quarter pooled two steps ago, two
accumulators, three products) with the pooling of the current quarter (this wave's 8 boxes x 64 channels = two lane-passes A and B of
4 boxes x 16 lanes x float4) in their shadows.  This script is the "hand": it holds the stream as a list of operations in issue order,
places the MFMAs evenly among them, counts the LDS operations behind every read (LDS operations return in order) and emits C++ whose
statement order is pinned by `__builtin_amdgcn_sched_barrier(0)` between chunks.

Stream of step n (quarter n = 4 item + q):
    frag(0) R(A0) R(A1) R(A2) | Fq(B') Fs(B') | S(A0) R(A3) S(A1) R(B0) S(A2) R(B1) S(A3) R(B2) Fv(A) R(B3) Fq(A) Fs(A) S(B0..B3) Fv(B)
    R(Xc): the four tap reads of corner c of pass X into tap buffer (corner index mod 4); S: its mul + 3 fma per channel;
    Fv: box sum; Fq: the exact quotient; Fs: fp16 split + the two plane stores; B': pass B of the PREVIOUS step (its last 34
    instructions cover the landing of this step's first reads); frag(k): the A fragments of k-step k, two k-steps ahead.
The emitted text uses names the kernel defines: TAP(buf, i), WT(pass, i), TB(pass, i), lt/rb/rt/lb per pass, vq per pass, etc.
"""
import sys

N_MFMA = 24


def build():
    ops = []  # (kind, payload)

    def R(p, c):  # four tap reads of corner c of pass p
        ops.append(("R", (p, c)))

    def S(p, c):
        ops.append(("W", ("tap", (p, c))))
        for i in range(4):  # tap i of the corner: 4 VALU (one per channel)
            ops.append(("V", f"S_TAP({p}, {c}, {i});", 4))

    def Fv(p):
        ops.append(("V", f"F_SUM({p});", 12))

    def Fq(p):
        for j in range(4):
            ops.append(("V", f"F_QUOT({p}, {j});", 5))

    def Fs(p):
        ops.append(("V", f"F_SPLIT({p});", 12))
        ops.append(("L", f"F_STORE({p});", 2))

    def frag(k):
        ops.append(("G", k))

    # ---- the step (tap buffers: corner index g = 4 pass + c, buffer g % 3: R(g + 2) may go out once S(g - 1) has read its buffer)
    frag(0)
    R(0, 0); R(0, 1)
    Fq(2); Fs(2)            # pass "2" = B of the previous step (its last 34 instructions cover the landing of the first reads)
    frag(1)
    R(0, 2)
    S(0, 0); R(0, 3)
    S(0, 1); R(1, 0)
    frag(2)
    S(0, 2); R(1, 1)
    S(0, 3); R(1, 2)
    Fv(0)
    frag(3)
    S(1, 0); R(1, 3)
    Fq(0); Fs(0)
    S(1, 1); S(1, 2); S(1, 3)
    Fv(1)
    return ops


def emit(ops, out):
    # cost of the fillers in issue slots (4 cycles each); MFMA = 2 slots of issue hold
    cost = 0
    for o in ops:
        if o[0] == "V":
            cost += o[2]
        elif o[0] == "R":
            cost += 5
        elif o[0] == "G":
            cost += 2
        elif o[0] == "L":
            cost += o[2]
    per = cost / N_MFMA
    # the MFMAs of k-step k (6 each) may not start before frag(k) has been waited for; place MFMA m after `per * (m + 1) - per / 2`
    # filler slots, but never before its fragments were read + ~12 slots
    lines = []
    lds_log = []  # issue order of LDS operations: ("tap", (p, c), n) / ("frag", k, n) / ("store", p, n)
    def lds_count_after(tag):
        n = 0
        found = False
        for t in lds_log:
            if found:
                n += t[1]
            if t[0] == tag:
                found = True
                n = 0
        assert found, tag
        return n
    acc = 0.0
    m = 0
    waited_frag = set()
    frag_issued = {}
    def place_mfma():
        nonlocal m
        k = m // 6
        if k not in waited_frag:
            cnt = min(15, lds_count_after(("frag", k)))
            lines.append(f"WAIT_FRAG({k}, {cnt});")
            waited_frag.add(k)
        lines.append(f"MFMA({m});")
        m += 1
    for o in ops:
        kind = o[0]
        if kind == "R":
            p, c = o[1]
            lines.append(f"R_TAPS({p}, {c});")
            lds_log.append((("tap", (p, c)), 5))  # four taps + the four weights of the corner (one 16-byte read of the box record)
            acc += 5
        elif kind == "G":
            lines.append(f"R_FRAG({o[1]});")
            lds_log.append((("frag", o[1]), 2))
            frag_issued[o[1]] = acc
            acc += 2
        elif kind == "W":
            tag = o[1]
            cnt = min(15, lds_count_after(tag))
            lines.append(f"WAIT_TAPS({tag[1][0]}, {tag[1][1]}, {cnt});")
        elif kind == "L":
            lines.append(o[1])
            lds_log.append((("store", o[1]), o[2]))
            acc += o[2]
        else:
            lines.append(o[1])
            acc += o[2]
        # MFMAs due by now
        while m < N_MFMA and acc >= per * (m + 0.5) and (m // 6) in frag_issued and acc >= frag_issued[m // 6] + 16:
            place_mfma()
    while m < N_MFMA:
        place_mfma()
    out.write("// generated by tools/gen_weave.py -- do not edit\n")
    out.write(f"// filler issue slots per step: {cost}, per MFMA: {per:.2f}\n")
    for l in lines:
        out.write(l + "\n")
        out.write("__builtin_amdgcn_sched_barrier(0);\n")


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
    with open(path, "w") as f:
        emit(build(), f)
