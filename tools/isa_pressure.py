#!/usr/bin/env python3
"""Approximate VGPR liveness along the instruction stream of one kernel of a built object (round 6: how the register peak of the
packed kernel was found in its D section, not in the contraction).  Straight-line backward liveness over the disassembly in
layout order (branches ignored: the kernels are almost branch-free), printed as max / mean live registers per chunk with tags
(MFMA, lds-reads, dpp, gstore, gload, SCRATCH).  The allocator needs ~10 more than the maximum shown (64-bit pairs, fragmentation).
  tools/isa_pressure.py <object> <mangled-name substring> [chunk = 200]
  tools/isa_pressure.py ergodic_exploration_amd/csrc/build/control_pack_kernel.o control_pack_kernelILi0ELi10ELb0ELi16ELi4 250"""
import re,sys,subprocess,os
obj,pat=os.path.abspath(sys.argv[1]),sys.argv[2]
import tempfile; os.chdir(tempfile.mkdtemp())
subprocess.run(['objcopy','-O','binary','--only-section=.hip_fatbin',obj,'fat.bin'],check=True)
subprocess.run(['/opt/rocm/lib/llvm/bin/clang-offload-bundler','--type=o','--targets=hipv4-amdgcn-amd-amdhsa--gfx950','--input=fat.bin','--output=k.co','--unbundle'],stderr=subprocess.DEVNULL)
txt=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump','-d','k.co'],capture_output=True,text=True).stdout
lines=txt.splitlines()
start=[i for i,l in enumerate(lines) if re.match(r'^[0-9a-f]+ <',l) and pat in l][0]
end=next(i for i in range(start+1,len(lines)) if re.match(r'^[0-9a-f]+ <',lines[i]))
body=lines[start:end]
reg_re=re.compile(r'\bv(\d+)\b|v\[(\d+):(\d+)\]')
ins=[]
for l in body:
    m=re.match(r'\s+(\S+)\s+(.*?)\s*//', l)
    if m: ins.append((m.group(1),m.group(2)))
N=len(ins)
def regs(tok):
    out=[]
    for m in reg_re.finditer(tok):
        if m.group(1): out.append(int(m.group(1)))
        else: out+=list(range(int(m.group(2)),int(m.group(3))+1))
    return out
live=set(); press=[0]*N
for i in range(N-1,-1,-1):
    op,args=ins[i]
    toks=[t.strip() for t in args.split(',')]
    if op.startswith(('global_store','ds_write','buffer_store','flat_store','scratch_store','s_','v_cmp')):
        d=[];u=[r for t in toks for r in regs(t)]
    else:
        d=regs(toks[0]); u=[r for t in toks[1:] for r in regs(t)]
        if op.startswith(('v_fmac','v_mac')): u+=d
    for r in d: live.discard(r)
    for r in u: live.add(r)
    press[i]=len(live)
chunk=int(sys.argv[3]) if len(sys.argv)>3 else 200
for c in range(0,N,chunk):
    seg=press[c:c+chunk]; ops=[ins[i][0] for i in range(c,min(N,c+chunk))]
    tag=''
    n=sum(o.startswith('v_mfma') for o in ops)
    if n: tag+=' MFMA%d'%n
    if sum(o.startswith('ds_read') for o in ops)>20: tag+=' lds-reads'
    if sum('dpp' in ins[i][1] for i in range(c,min(N,c+chunk)))>5: tag+=' dpp'
    if any(o.startswith('global_store') for o in ops): tag+=' gstore'
    if any(o.startswith('global_load') for o in ops): tag+=' gload'
    if any(o.startswith('scratch') for o in ops): tag+=' SCRATCH'
    print('%5d max %3d avg %5.1f %s'%(c,max(seg),sum(seg)/len(seg),tag))
print('N',N)
