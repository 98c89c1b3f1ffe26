"""The slice parse's carried bit window (slice_parse.h jm_win_fetch / jm_win) requests two ring dwords with a ds_read2st64_b32
whose wait is NOT in the same statement: between that request and the next `s_waitcnt lgkmcnt(0)` no instruction may touch
the destination registers (a copy the compiler placed there would read them before the data has landed).  This looks at the
ISA of k_parse for such an instruction, in program order (every path from a request to a wait is a fall-through or a forward
branch over code that does not use the window, so the linear order covers them).
    python tools/check_parse_isa.py            # compiles kernels.hip for gfx950 to assembly and checks k_parse
Exit status 1 and the offending lines if one is found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_isa(defs=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "jsmpeg_amd", "csrc"), "-S", "--cuda-device-only", "-o", out,
                               *defs, os.path.join(ROOT, "jsmpeg_amd", "csrc", "kernels.hip")], stderr=subprocess.DEVNULL)
        text = open(out).read()
    out = {}
    for sym in ("_Z7k_parse11JmParseBufs", "_Z13k_parse_split11JmParseBufs"):
        m = re.search(r"^%s:.*?^\s*\.size\s+%s" % (sym, sym), text, re.S | re.M)
        assert m, sym + " not found in the assembly"
        out[sym] = m.group(0).splitlines()
    return out


def regs_of(line):
    """vector registers an instruction line names: v5, v[2:3] -> {5} / {2, 3}"""
    code = line.split(";")[0]
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", code))
    return out


def dest_regs(code):
    """registers an instruction writes (its first operand), none for stores / LDS writes / compares into scalar registers"""
    op = code.split()[0]
    if op.startswith(("global_store", "ds_write", "v_cmp", "buffer_store", "flat_store", "s_", "global_atomic")):
        return set()
    return regs_of(code.split(",")[0])


def check_uses(lines):
    """the second rule: whoever READS a carried-window register finds, going back through the program, a full LDS wait (or
    another instruction that wrote the register: then it is a temporary there) before it finds a request into it"""
    codes = [(n, l.split(";")[0].strip()) for n, l in enumerate(lines)]
    codes = [(n, c) for n, c in codes if c and not c.endswith(":") and not c.startswith(".")]
    carried = set()
    for i, (n, c) in enumerate(codes):
        if c.startswith("ds_read2st64_b32") and not (i + 1 < len(codes) and codes[i + 1][1].startswith("s_waitcnt") and "lgkmcnt(0)" in codes[i + 1][1]):
            carried |= regs_of(c.split(",")[0])
    bad = []
    for i, (n, c) in enumerate(codes):
        if c.startswith("ds_read2st64_b32"):
            continue
        used = (regs_of(c) - dest_regs(c)) & carried
        for r in used:
            for j in range(i - 1, -1, -1):
                cj = codes[j][1]
                if cj.startswith("s_waitcnt") and "lgkmcnt(0)" in cj:
                    break
                if cj.startswith("ds_read2st64_b32") and r in regs_of(cj.split(",")[0]):
                    bad.append((codes[j][0], n, lines[n].strip()))
                    break
                if r in dest_regs(cj):
                    break
    return carried, bad


def check_chunks(lines):
    """the third rule, for the refill's two halves (jm_lane_request / jm_lane_land, tagged `jm_req` / `jm_land` in the asm): from a
    request to the next landing wait nothing names the requested registers -- in program order to the end of the turn loop,
    and from the loop's head (the target of the backward branch that closes the loop around the requests) to the first wait"""
    req = [n for n, l in enumerate(lines) if "jm_req" in l]
    land = [n for n, l in enumerate(lines) if "jm_land" in l]
    if not req:
        return set(), []
    R = set()
    for n in req:
        R |= regs_of(lines[n].split(";")[0].split(",")[0])
    labels = {l.split(":")[0].strip(): n for n, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    first_land = min(land)
    head = None
    for n in range(max(req) + 1, len(lines)):
        m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[n])
        if m and labels.get(m.group(1), len(lines)) < first_land:
            head = labels[m.group(1)]            # (the innermost such loop is closed first)
            break
    assert head is not None, "no loop around the requests"
    bad, flying = [], False
    order = list(range(min(req), len(lines))) + list(range(head, first_land + 1))
    for n in order:
        l = lines[n]
        if "jm_req" in l:
            flying = True
        elif "jm_land" in l:
            flying = False
        elif flying and regs_of(l) & R and not l.split(";")[0].strip().endswith(":"):
            bad.append((min(req), n, l.strip()))
    return R, bad


def check(lines):
    bad, pending, requests = [], None, 0
    for n, line in enumerate(lines):
        code = line.split(";")[0].strip()
        if not code or code.endswith(":") or code.startswith("."):
            continue
        if code.startswith("s_waitcnt") and "lgkmcnt(0)" in code:
            pending = None
            continue
        if pending is not None and regs_of(line) & pending[1] and not code.startswith("ds_read2st64_b32"):
            bad.append((pending[0], n, line.strip()))
        if code.startswith("ds_read2st64_b32"):
            requests += 1
            pending = (n, regs_of(code.split(",")[0]))
    return requests, bad


if __name__ == "__main__":
    rc = 0
    for sym, lines in kernel_isa(sys.argv[1:]).items():
        requests, bad = check(lines)
        carried, bad2 = check_uses(lines)
        print("%s: %d window requests, %d instructions touch a window register before its wait" % (sym, requests, len(bad)))
        print("  carried window registers: %s; %d readers without a wait behind the request" % (sorted(carried), len(bad2)))
        bad += bad2
        R, bad3 = check_chunks(lines)
        print("  requested chunk registers: v%d..v%d; %d instructions name one between request and landing" % (min(R), max(R), len(bad3)) if R else "  (the refill in one piece)")
        bad += bad3
        for r, n, l in bad[:40]:
            print("  request at line %d: line %d: %s" % (r, n, l))
        if bad or not requests or (("split" in sym) != bool(R)):
            rc = 1
    sys.exit(rc)
