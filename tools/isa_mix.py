# Instruction mix of a line range of a kernel's ISA listing (how the attention forward's register copies were found).
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o k.s emdr2_amd/csrc/attention.hip
#   python tools/isa_mix.py k.s FIRST_LINE LAST_LINE       (line numbers of k.s, e.g. one loop body between two labels)
import re,sys,collections
lines=open(sys.argv[1]).read().splitlines()
a,b=int(sys.argv[2]),int(sys.argv[3])
c=collections.Counter()
for l in lines[a-1:b]:
    l=l.strip()
    if not l or l.startswith(';') or l.startswith('.') : continue
    op=l.split()[0]
    if op=='s_nop': c['s_nop(%s)'%l.split()[1]]+=1
    else: c[op]+=1
tot=sum(c.values())
print('total',tot,'valu',sum(v for k,v in c.items() if k.startswith('v_') and 'mfma' not in k),'mfma',sum(v for k,v in c.items() if 'mfma' in k),'salu',sum(v for k,v in c.items() if k.startswith('s_')),'ds',sum(v for k,v in c.items() if k.startswith('ds_')))
for k,v in c.most_common(60): print('%-28s %4d'%(k,v))
