"""Where a kernel's scratch traffic sits: scratch_load / scratch_store of one kernel of a `--save-temps -gline-tables-only` assembly
file, by source line (innermost inlined location) and as a profile along the instruction stream (blocks of 1000 instructions).
Usage: scratch_map.py file.s <kernel-name-substring> [top]"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    s = open(path).read().split("\n")
    starts = [i for i, l in enumerate(s) if key in l.split(":")[0] and re.match(r"^_Z\w+:", l)]
    if not starts:
        raise SystemExit("no kernel label containing %r" % key)
    i0 = starts[0]
    i1 = next(i for i in range(i0, len(s)) if s[i].startswith(".Lfunc_end"))
    print("kernel", s[i0].split(":")[0], "asm lines", i1 - i0)
    cur = None
    st, ld = collections.Counter(), collections.Counter()
    prof = collections.Counter()
    lines_at = {}
    n = 0
    for l in s[i0:i1]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        t = l.strip()
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        n += 1
        lines_at.setdefault(n // 1000, cur)
        if t.startswith("scratch_store"):
            st[cur] += 1
            prof[(n // 1000, "st")] += 1
        elif t.startswith("scratch_load"):
            ld[cur] += 1
            prof[(n // 1000, "ld")] += 1
    print("instructions", n, "scratch stores", sum(st.values()), "loads", sum(ld.values()))
    print("stores by line:", st.most_common(top))
    print("loads by line:", ld.most_common(top))
    print("profile (block of 1000 instr: stores/loads @ first source line):")
    for b in range(n // 1000 + 1):
        a, c = prof.get((b, "st"), 0), prof.get((b, "ld"), 0)
        if a or c:
            print("  %4d: st %3d ld %3d  @%s" % (b, a, c, lines_at.get(b)))


if __name__ == "__main__":
    main()
