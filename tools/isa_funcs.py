#!/usr/bin/env python3
"""Per device function of a -save-temps gfx950 assembly file: VGPRs, private memory, scratch instructions (filter by substring)."""
import re, subprocess, sys
path, pats = sys.argv[1], sys.argv[2:]
s = open(path).read()
for m in re.finditer(r'^(\S+):\s*; @(\S+)\n(.*?); -- End function', s, re.S | re.M):
    name, body = m.group(2), m.group(3)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if pats and not any(p in dn for p in pats):
        continue
    tail = s[m.end():m.end() + 3000]
    nv = re.search(r'; NumVgprs: (\d+)', tail)
    sc = re.search(r'; ScratchSize: (\d+)', tail)
    print("%-120s vgpr %s private %s B scratch-instr %d valu-ish %d" % (dn[:120], nv.group(1) if nv else "?", sc.group(1) if sc else "?",
          len(re.findall(r'scratch_(?:load|store)', body)), len(re.findall(r'^\s+v_', body, re.M))))
