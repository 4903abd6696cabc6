#!/usr/bin/env python3
"""What the compiler made of qgemv_lean_kernel's request / wait structure (csrc/qgemv_lean.hip: head): for every kernel and
every pipelined region, the vector-memory instructions between the fence load and its wait, the wait's count, and the
counted waits in front of the items behind it.  Usage: tools/lean_isa_report.py <qgemv_lean...gfx950.s> [kernel substring]
(the .s comes from `hipcc ... -save-temps -c exllamav2_amd/csrc/qgemv_lean.hip`)."""
import re
import sys


def main(path, want=""):
    s = open(path).read()
    for m in re.finditer(r"^(_Z17qgemv_lean_kernel\w+):", s, re.M):
        name = m.group(1)
        if want not in name:
            continue
        body = s[m.end():s.index(".Lfunc_end", m.end())].splitlines()
        print("==", name, len(body), "lines")
        i = 0
        while i < len(body):
            t = body[i].strip()
            # the fence load: a global_load_dword (not nt) right behind an empty asm block
            if t.startswith("global_load_dword ") and " nt" not in t:
                loads, j = [], i + 1
                while j < len(body) and j < i + 80:
                    u = body[j].strip()
                    if u.startswith("global_load") or u.startswith("buffer_load"):
                        loads.append(u.split()[0])
                    if u.startswith("s_waitcnt") and "vmcnt" in u:
                        break
                    j += 1
                wait = body[j].strip() if j < len(body) else "?"
                # counted waits that follow (until the next label of a load region)
                later = []
                for k in range(j + 1, min(len(body), j + 1600)):
                    u = body[k].strip()
                    if u.startswith("s_waitcnt") and "vmcnt" in u:
                        later.append(re.search(r"vmcnt\((\d+)\)", u).group(1))
                    if u.startswith("global_load_dword ") and " nt" not in u:
                        break
                print(f"  fence @{i}: {len(loads)} loads behind it {loads[:12]} -> {wait}; later vmcnt waits: {later[:14]}")
                i = j
            i += 1


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
