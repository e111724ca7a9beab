"""Print the s_waitcnt vmcnt / barrier / scratch lines of one kernel in a hipcc -S listing, marking which sit inside a loop.
   python tools/isa_fn.py listing.s <substring of the mangled kernel name> [more substrings]"""
import re
import sys
src = open(sys.argv[1]).read().split("\n")
subs = sys.argv[2:]
i = 0
while i < len(src):
    l = src[i]
    if l.startswith("_Z") and l.split(":")[0].endswith("E") and all(s in l.split(":")[0] for s in subs) and ":" in l:
        name = l.split(":")[0]
        body = []
        while i < len(src) and ".Lfunc_end" not in src[i]:
            body.append(src[i]); i += 1
        print("==", name, len(body), "lines")
        inloop = False
        for j, b in enumerate(body):
            if "Loop Header" in b: inloop = True
            if re.search(r"s_waitcnt vmcnt|s_barrier|scratch_|Loop Header", b): print("   %5d %s %s" % (j, b.strip()[:80], "(loop)" if inloop else ""))
    i += 1
