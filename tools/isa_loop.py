"""Prints the VMEM / wait / move instructions of the basic blocks that hold at least N MFMAs / f32 multiply-adds of one kernel.
usage: python tools/isa_loop.py <file.s> <mangled-name-regex> [min_mfma]"""
import re, sys, collections
s = open(sys.argv[1]).read()
pat = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for m in re.finditer(r'^(_Z\S*):[^\n]*\n(.*?)\.Lfunc_end\d+:', s, re.S | re.M):
    if not re.search(pat, m.group(1)): continue
    print("==", m.group(1))
    for b in re.split(r'\n(?=\.LBB\d+_\d+:)', m.group(2)):
        L = b.split('\n')
        ins = [x.strip() for x in L if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
        if sum(('mfma' in x or 'v_fmac' in x or 'v_pk_fma' in x) for x in ins) >= mn:
            c = collections.Counter(x.split()[0] for x in ins)
            print(L[0].split()[0], len(ins), c.most_common(14))
            for n, l in enumerate(ins):
                if 'waitcnt' in l or 'buffer_' in l or 'v_mov' in l or 'global_' in l or l.startswith('ds_'): print("   ", n, l)
