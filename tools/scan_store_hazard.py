#!/usr/bin/env python3
"""Scan compiler output (hipcc -S --cuda-device-only) for the store-data hazard LLVM does not guard on gfx950: a
buffer_store_dwordx3/x4 whose soffset is a REGISTER, followed within two instructions by a vector instruction that writes
one of its data registers (tools/microbench/store_data_hazard.hip shows the corruption).
    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o k.s kernel.hip && python tools/scan_store_hazard.py k.s"""
import re,sys
def regs(tok):
    m=re.match(r'v\[(\d+):(\d+)\]',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r'v(\d+)$',tok)
    if m: return {int(m.group(1))}
    return set()
for fn in sys.argv[1:]:
    lines=open(fn).read().split('\n'); kern=None; hits=0; total=0; ex=[]
    for i,l in enumerate(lines):
        m=re.match(r'^(_Z\S+):',l)
        if m: kern=m.group(1)
        t=l.strip()
        if t.startswith('buffer_store_dwordx4') or t.startswith('buffer_store_dwordx3'):
            ops=[o.strip() for o in t.split(None,1)[1].split(',')]
            data=regs(ops[0]); soff=ops[3].split()[0] if len(ops)>3 else ''
            if not soff.startswith('s'): continue
            total+=1
            k=0;j=i+1
            while k<2 and j<len(lines):
                u=lines[j].strip(); j+=1
                if not u or u.startswith(';') or u.endswith(':') or u.startswith('.'): continue
                k+=1
                if u.startswith('s_nop'): break
                if u.startswith('v_') and not u.startswith('v_cmp') and not u.startswith('v_readlane'):
                    dst=[o.strip() for o in u.split(None,1)[1].split(',')][0]
                    if regs(dst)&data:
                        hits+=1
                        if len(ex)<3: ex.append((kern[:70],t[:50],u[:40]))
    print(fn,'wide buffer stores with register soffset:',total,'| VALU write of their data within 2 instructions:',hits)
    for e in ex: print('   ',e)
