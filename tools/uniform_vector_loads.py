#!/usr/bin/env python3
"""Lists, per kernel of a gfx950 assembly file (hipcc -S --cuda-device-only), the vector loads whose ADDRESS is uniform: the saddr form with a zero vector offset
(global_load vD, vZ, s[a:b]) and loads through a register pair that was just copied from SGPRs.  Every wave that executes one sends a request for the same line to the
same L2 channel (~2 ns each: 11 us of a 300 k-thread launch); the fix is to read the value once at the top of the kernel, before its first store, where the
compiler takes the scalar cache.  usage: tools/uniform_vector_loads.py file.s [...]"""
import re, sys
for fn in sys.argv[1:]:
    s = open(fn).read()
    for m in re.finditer(r'^(_Z\w+):\s*;', s, re.M):
        a = m.start()
        try: b = s.index('.Lfunc_end', a)
        except ValueError: continue
        lines = [l.strip() for l in s[a:b].split('\n')]
        hits = []
        src = {}   # vgpr -> line index where it was last v_mov'ed from an sgpr
        for n, l in enumerate(lines):
            mm = re.match(r'v_mov_b32_e32 v(\d+), s\d+', l)
            if mm: src[int(mm.group(1))] = n; continue
            mm = re.match(r'v_mov_b64_e32 v\[(\d+):(\d+)\], s\[', l)
            if mm: src[int(mm.group(1))] = n; src[int(mm.group(2))] = n; continue
            mm = re.match(r'(global_load_\w+) v\S+, v(\d+), s\[\d+:\d+\]', l)
            if mm: hits.append((n, l)); continue
            mm = re.match(r'(global_load_\w+) v\S+, v\[(\d+):(\d+)\], off', l)
            if mm:
                lo, hi = int(mm.group(2)), int(mm.group(3))
                if lo in src and hi in src and n - src[lo] < 16 and n - src[hi] < 16: hits.append((n, l))
            # any other write to a vgpr invalidates it
            mm = re.match(r'v_\w+ v(\d+),', l)
            if mm and not l.startswith('v_mov_b32_e32') : src.pop(int(mm.group(1)), None)
        if hits:
            print(f"{fn.split('/')[-1]:16s} {m.group(1)[:44]:44s} {len(lines):5d} lines  {len(hits):3d} uniform vector loads, first at {hits[0][0]}: {hits[0][1]}")
