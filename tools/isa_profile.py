"""Static VALU-instruction census of render_kernel<false> by source function (needs a -gline-tables-only -S build).
usage: isa_profile.py dev.s rt_kernels.h"""
import re, sys, collections
asm, src = sys.argv[1], sys.argv[2]
kname = sys.argv[3] if len(sys.argv) > 3 else "_ZN3rtk13render_kernelILb0EE"
kfn = sys.argv[4] if len(sys.argv) > 4 else "render_kernel"
# function line ranges in rt_kernels.h: a line starting a function = "RT_DEV ... name(" or "__global__"
starts = []
lines = open(src).read().split("\n")
for i, l in enumerate(lines, 1):
    m = re.match(r"^(?:RT_DEV|template.*RT_DEV)\s.*?([A-Za-z_0-9]+)\s*\(", l)
    if m and not l.startswith("template"):
        starts.append((i, m.group(1)))
    elif l.startswith("__global__"):
        m = re.search(r"\)\s*([a-z_]+)\(", l)
        starts.append((i, m.group(1) if m else "kernel"))
def fn_of(line):
    name = "?"
    for s, n in starts:
        if s <= line: name = n
        else: break
    return name
inner = collections.Counter(); top = collections.Counter(); allv = 0
in_k = False; chain = []
for l in open(asm):
    if l.startswith(kname): in_k = True; continue
    if in_k and l.startswith(".Lfunc_end"): break
    if not in_k: continue
    if "\t.loc\t" in l:
        chain = []
        for f, ln in re.findall(r"([A-Za-z_0-9.]+):(\d+):\d+", l.split(";", 1)[1] if ";" in l else ""):
            chain.append((f, int(ln)))
        continue
    m = re.match(r"^\t(v_[a-z0-9_]+)", l)
    if not m: continue
    allv += 1
    ks = [fn_of(ln) for f, ln in chain if f == "rt_kernels.h"]
    other = [f for f, ln in chain if f != "rt_kernels.h"]
    innermost = ks[0] if ks else "?"
    if other and other[0] in ("rt_portable_math.h", "rt_seed_schedule.h"): innermost = other[0] + "<-" + innermost
    inner[innermost] += 1
    # top = the callee directly under render_kernel
    t = kfn
    for k in reversed(ks):
        if k != kfn and k != "?": t = k; break
    top[t] += 1
print("VALU instructions in", kname, allv)
print("-- by callee of render_kernel"); [print("%6d %s" % (c, n)) for n, c in top.most_common()]
print("-- by innermost function"); [print("%6d %s" % (c, n)) for n, c in inner.most_common(40)]
