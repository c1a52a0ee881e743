#!/usr/bin/env python3
"""Static safety check of the fused MLP kernels' ISA (gfx950).

The kernels issue their weight-fragment loads as inline asm and wait for them with hand-counted
`s_waitcnt vmcnt(N)` many instructions later (nerf_amd/csrc/mlp_common.hip.h); wgrad.hip does the same with LDS
reads and lgkmcnt.  The compiler does not know
that the destination VGPRs of such a load are "in flight": if its register allocator copies, spills or
re-uses one of them before the load has landed, the kernel computes garbage or -- when the register is
re-used for an address -- faults the GPU.  This tool runs a forward data-flow analysis over the kernel's
control-flow graph with a model of the in-order VMEM counter (state = queue of outstanding VMEM operations,
each with the VGPRs an asm load will write; states are merged aligned at the newest operation, which
over-approximates what is outstanding) and reports every instruction that reads or writes a VGPR while an
asm load into it may still be outstanding.

usage: check_inflight.py <file.s> [kernel-name-substring ...]     (exit status 1 on any violation)
"""
import re
import sys

VMEM = re.compile(r"^(global_|scratch_|buffer_|flat_)(load|store|atomic)")
LGKM = re.compile(r"^(ds_|s_load_|s_buffer_load_)")
RE_RANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
RE_ONE = re.compile(r"\bv(\d+)\b")
RE_VMCNT = re.compile(r"vmcnt\((\d+)\)")
RE_LGKM = re.compile(r"lgkmcnt\((\d+)\)")
RE_LABEL = re.compile(r"^(\.LBB\d+_\d+):")
RE_BRANCH = re.compile(r"^s_(c?branch\S*)\s+(\.LBB\d+_\d+)")
CAP = 64   # vmcnt is a 6-bit counter: operations older than this can only be retired by vmcnt(0)


def vregs(text):
    out = set()
    for m in RE_RANGE.finditer(text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in RE_ONE.finditer(text):
        out.add(int(m.group(1)))
    return out


def parse(body):
    """-> list of blocks: dict(label, ins=[(line_no, text, in_asm)], succ=[block indices], fall=bool)"""
    blocks = [dict(label=None, ins=[], targets=[], fall=True)]
    in_asm = False
    for i, raw in enumerate(body):
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        m = RE_LABEL.match(raw)
        if m:
            blocks.append(dict(label=m.group(1), ins=[], targets=[], fall=True))
            continue
        c = raw.split(";")[0].strip()
        if not c or c.startswith("."):
            continue
        cur = blocks[-1]
        cur["ins"].append((i, c, in_asm))
        b = RE_BRANCH.match(c)
        if b:
            cur["targets"].append(b.group(2))
            cur["fall"] = b.group(1) != "branch"
            blocks.append(dict(label=None, ins=[], targets=[], fall=True))
        elif c.startswith("s_endpgm"):
            cur["fall"] = False
            blocks.append(dict(label=None, ins=[], targets=[], fall=True))
    index = {b["label"]: k for k, b in enumerate(blocks) if b["label"]}
    for k, b in enumerate(blocks):
        succ = [index[t] for t in b["targets"] if t in index]
        if b["fall"] and k + 1 < len(blocks):
            succ.append(k + 1)
        b["succ"] = succ
    return blocks


def merge1(a, b):
    """queues are lists oldest..newest of frozensets; align at the newest end, union element-wise"""
    n = max(len(a), len(b))
    pa = [frozenset()] * (n - len(a)) + a
    pb = [frozenset()] * (n - len(b)) + b
    return [x | y for x, y in zip(pa, pb)]


def merge(a, b):
    """state = (vmcnt queue, lgkmcnt queue)"""
    if a is None:
        return (list(b[0]), list(b[1]))
    return (merge1(a[0], b[0]), merge1(a[1], b[1]))


def retire(q, n):
    return (q[len(q) - n:] if n else []) if len(q) > n else q


def push(q, dst):
    q.append(dst)
    if len(q) > CAP:
        q[:2] = [q[0] | q[1]]


def transfer(block, state, viol):
    qv, ql = list(state[0]), list(state[1])
    for line_no, c, in_asm in block["ins"]:
        if c.startswith("s_waitcnt"):
            m = RE_VMCNT.search(c)
            if m:
                qv = retire(qv, int(m.group(1)))
            m = RE_LGKM.search(c)
            if m:
                ql = retire(ql, int(m.group(1)))
            continue
        if LGKM.match(c) and c.startswith("s_"):
            push(ql, frozenset())
            continue
        if c.startswith("s_"):
            continue
        used = vregs(c)
        busy = set()
        for d in qv:
            busy |= d
        for d in ql:
            busy |= d
        hot = used & busy
        if hot and viol is not None:
            viol[line_no] = (c, sorted(hot))
        if VMEM.match(c):
            push(qv, frozenset(vregs(c.split(",")[0])) if (in_asm and "_load_" in c and "_lds_" not in c) else frozenset())
        elif LGKM.match(c):
            push(ql, frozenset(vregs(c.split(",")[0])) if (in_asm and c.startswith("ds_read")) else frozenset())
    return (qv, ql)


def check_kernel(lines, start):
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start + 1:end + 1]
    blocks = parse(body)
    state_in = [None] * len(blocks)
    state_in[0] = ([], [])
    work = [0]
    rounds = 0
    while work:
        k = work.pop()
        rounds += 1
        if rounds > 200000:
            raise RuntimeError("data-flow did not converge")
        out = transfer(blocks[k], state_in[k], None)
        for s in blocks[k]["succ"]:
            new = merge(state_in[s], out)
            if state_in[s] is None or new != state_in[s]:
                state_in[s] = new
                work.append(s)
    viol = {}
    for k, b in enumerate(blocks):
        if state_in[k] is not None:
            transfer(b, state_in[k], viol)
    n_loads = sum(1 for l in body if "global_load_dwordx4" in l or "ds_read_b128" in l)
    return [(ln,) + viol[ln] for ln in sorted(viol)], n_loads


def main():
    path = sys.argv[1]
    keys = sys.argv[2:]
    lines = open(path).read().split("\n")
    bad = 0
    for i, l in enumerate(lines):
        if not (l.startswith("_Z") and ":" in l and "@" in l):
            continue
        name = l.split(":")[0]
        if keys and not any(k in name for k in keys):
            continue
        viol, nload = check_kernel(lines, i)
        print(f"{name}: {nload} wide loads, {len(viol)} in-flight register violations")
        for idx, c, hot in viol[:12]:
            print(f"    +{idx}: {c}    <- may be in flight: v{hot}")
        bad += len(viol)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
