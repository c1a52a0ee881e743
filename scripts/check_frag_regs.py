"""Static check of the fused MLP kernels' ISA: the VGPRs that the inline-asm fragment loads write (in flight
for hundreds of cycles) must never be touched by compiler-generated instructions other than as MFMA operands.
usage: check_frag_regs.py file.s kernel-name-substring"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().endswith(tuple(":")) or (l.startswith("_ZN") and key in l and ": " in l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out
in_asm = False
frag = set()
for l in body:
    if "#ASMSTART" in l: in_asm = True; continue
    if "#ASMEND" in l: in_asm = False; continue
    if in_asm and "global_load_dwordx4" in l:
        dst = l.split("global_load_dwordx4")[1].split(",")[0]
        frag |= regs(dst)
print("fragment registers:", len(frag), sorted(frag)[:4], "...", sorted(frag)[-4:])
bad = 0
in_asm = False
for i, l in enumerate(body):
    if "#ASMSTART" in l: in_asm = True; continue
    if "#ASMEND" in l: in_asm = False; continue
    if in_asm: continue
    code = l.split(";")[0]
    if not code.strip() or code.strip().startswith("."): continue
    r = regs(code)
    if r & frag:
        if "v_mfma" in code:
            ops = code.split(",")
            # dst (ops[0]) and srcC (ops[3]) must not be fragment registers
            if (regs(ops[0]) | regs(ops[3])) & frag: pass
            else: continue
        bad += 1
        if bad <= 40: print("line", i, ":", code.strip())
print("compiler instructions touching fragment registers:", bad)
